// Microbenchmark: what does a launch pay at its end for the bytes it wrote?  k_step writes ~ 1 KB per env at the end of every wavefront's life;
// its launch period grows from 33.3 us (4096 envs, 4 MB written) to 38.9 us (16 384 envs, 17 MB) while a wavefront's lifetime grows by 1 us and the
// dispatch ramp by 0.2 us (launch_ramp.hip).  Here every wavefront idles for 10 us and then stores BYTES, with ordinary stores (write-back L2: the dirty
// lines leave at the end-of-kernel release), non-temporal stores, or system-scope write-through stores (sc0 sc1).  Period of back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/writeback_tail.hip -o tools/microbench/writeback_tail.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int MODE>      // 0 plain, 1 non-temporal, 2 sc0 sc1 (system scope, write-through), 3 nt sc0 sc1
__global__ void __launch_bounds__(64) k_tail(v4f* __restrict__ out, int per_lane, int spin, float seed) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  while ((long long)(__builtin_amdgcn_s_memtime() - c0) < spin) __builtin_amdgcn_s_sleep(8);
  v4f* p = out + (size_t)blockIdx.x * 64 * per_lane + threadIdx.x;
  const v4f v = {seed, seed + 1.f, seed + 2.f, (float)blockIdx.x};
  for (int k = 0; k < per_lane; k++) {
    v4f* q = p + 64 * k;
    if (MODE == 0) *q = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, q);
    else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(q), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(q), "v"(v) : "memory");
  }
}
template <int MODE>
static int run(const char* name, int waves, int bytes_per_wave, v4f* buf, hipEvent_t e0, hipEvent_t e1) {
  const int per_lane = bytes_per_wave / (64 * 16), spin = 24000, reps = 300;
  for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k_tail<MODE>, dim3(waves), dim3(64), 0, 0, buf, per_lane, spin, (float)r);
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_tail<MODE>, dim3(waves), dim3(64), 0, 0, buf, per_lane, spin, (float)r);
  CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"stores\": \"%s\", \"wavefronts\": %d, \"MB_per_launch\": %.2f, \"period_us\": %.2f}\n", name, waves, waves * (double)bytes_per_wave / 1e6, ms * 1e3 / reps);
  return 0;
}
int main() {
  v4f* buf; CHECK(hipMalloc(&buf, (size_t)4096 * 65536)); CHECK(hipMemset(buf, 0, (size_t)4096 * 65536));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int waves : {256, 1024})
    for (int bytes : {0, 4096, 16384, 65536}) {
      if (run<0>("plain", waves, bytes, buf, e0, e1)) return 1;
      if (bytes == 0) continue;
      if (run<1>("nontemporal", waves, bytes, buf, e0, e1)) return 1;
      if (run<2>("sc0 sc1", waves, bytes, buf, e0, e1)) return 1;
      if (run<3>("sc0 sc1 nt", waves, bytes, buf, e0, e1)) return 1;
    }
  return 0;
}
