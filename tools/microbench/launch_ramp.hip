// Microbenchmark: how fast does the chip start the wavefronts of a launch?  k_step at 16 384 envs is 1024 one-wavefront workgroups of 34.6 KB LDS and
// 422 registers per lane, and its duration grows by ~ 4 us over the 256-workgroup launch of 4096 envs although a wavefront's lifetime barely changes
// (profiles/r04_pmc_cu_sharing.json).  Each wavefront stores the real-time clock (100 MHz) and the shader clock at its first instruction; the ramp is
// last start - first start.  Variants: LDS 0 / 34 KB per workgroup, few registers / the whole register file, 1 / 4 wavefronts per workgroup.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/launch_ramp.hip -o tools/microbench/launch_ramp.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int LDS_BYTES, bool BIG_REGS>
__global__ void __launch_bounds__(256) k_probe(unsigned long long* start, unsigned long long* stop, int spin) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  extern __shared__ char dyn[];
  __shared__ char lds[LDS_BYTES > 0 ? LDS_BYTES : 4];
  if (BIG_REGS) asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a165, v255" ::: "v255", "a165");      // 256 VGPR + 166 AGPR like k_step
  if (threadIdx.x == 0) lds[0] = 1;
  // stay resident for `spin` shader clocks so that the workgroups of the launch coexist the way the step's do
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  while ((long long)(__builtin_amdgcn_s_memtime() - c0) < spin) __builtin_amdgcn_s_sleep(8);
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0) { start[w] = t0; stop[w] = __builtin_amdgcn_s_memrealtime() + (lds[0] == 77); }
}
template <int LDS_BYTES, bool BIG_REGS>
static int run(const char* name, int waves, int wpb, int spin) {
  unsigned long long *d0, *d1; CHECK(hipMalloc(&d0, waves * 8)); CHECK(hipMalloc(&d1, waves * 8));
  std::vector<unsigned long long> h0(waves), h1(waves);
  double ramp = 0, span = 0; const int reps = 20;
  for (int r = 0; r < reps + 3; r++) {
    hipLaunchKernelGGL((k_probe<LDS_BYTES, BIG_REGS>), dim3(waves / wpb), dim3(64 * wpb), 0, 0, d0, d1, spin);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h0.data(), d0, waves * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h1.data(), d1, waves * 8, hipMemcpyDeviceToHost));
    if (r < 3) continue;
    const unsigned long long a = *std::min_element(h0.begin(), h0.end()), b = *std::max_element(h0.begin(), h0.end()), c = *std::max_element(h1.begin(), h1.end());
    ramp += (b - a) * 0.01; span += (c - a) * 0.01;      // 100 MHz ticks -> us
  }
  printf("{\"variant\": \"%s\", \"wavefronts\": %d, \"per_workgroup\": %d, \"first_to_last_start_us\": %.2f, \"first_start_to_last_end_us\": %.2f}\n", name, waves, wpb, ramp / reps, span / reps);
  (void)hipFree(d0); (void)hipFree(d1); return 0;
}
int main() {
  const int spin = 24000;      // 10 us at 2.4 GHz
  for (int waves : {256, 1024}) {
    if (run<0, false>("no LDS, few registers", waves, 1, spin)) return 1;
    if (run<34608, false>("34.6 KB LDS, few registers", waves, 1, spin)) return 1;
    if (run<0, true>("no LDS, 422 registers", waves, 1, spin)) return 1;
    if (run<34608, true>("34.6 KB LDS, 422 registers (k_step's shape)", waves, 1, spin)) return 1;
    if (run<4 * 34608, true>("4 wavefronts per workgroup, 4 x 34.6 KB LDS, 422 registers", waves, 4, spin)) return 1;
    if (run<20464, false>("20 KB LDS, 256 registers at most (k_step_w2's shape)", waves, 1, spin)) return 1;
  }
  return 0;
}
