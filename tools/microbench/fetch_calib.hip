// Microbenchmark: what rocprofv3's FETCH_SIZE / WRITE_SIZE report for the access shapes of k_step (MI355X_MICROARCH.md calibrates FETCH_SIZE
// only for 16 B-per-lane streams, where it reads half of the bytes).  Three kernels over a [rows][N] fp32 array, each touching every byte once:
//   k_soa4   one dword per lane per row, consecutive lanes consecutive addresses   (the SoA state loads of the step)
//   k_quad4  one dword per lane, lanes l and l+1..3 of a quad share the address    (per-env values read by the four limb lanes)
//   k_vec16  16 bytes per lane, streaming                                           (the obs / states stores and their read-back shape)
// and the same three as stores.  Run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  --pmc WRITE_SIZE  (separate passes).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/fetch_calib.hip -o tools/microbench/fetch_calib.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ROWS 64
__global__ void __launch_bounds__(64) k_soa4(const float* __restrict__ in, float* out, int N) {
  const int i = blockIdx.x * 64 + threadIdx.x; float s = 0.f;
  for (int r = 0; r < ROWS; r++) s += in[(size_t)r * N + i];
  if (s == 12345.678f) out[i] = s;
}
__global__ void __launch_bounds__(64) k_quad4(const float* __restrict__ in, float* out, int N) {      // 16 addresses per wavefront per row
  const int e = blockIdx.x * 16 + (threadIdx.x >> 2); float s = 0.f;
  for (int r = 0; r < ROWS; r++) s += in[(size_t)r * N + e];
  if (s == 12345.678f) out[e] = s;
}
__global__ void __launch_bounds__(64) k_vec16(const float4* __restrict__ in, float* out, int N4) {
  const int i = blockIdx.x * 64 + threadIdx.x; float s = 0.f;
  for (int r = 0; r < ROWS / 4; r++) { float4 v = in[(size_t)r * N4 + i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) out[i] = s;
}
__global__ void __launch_bounds__(64) k_store4(float* out, int N) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  for (int r = 0; r < ROWS; r++) out[(size_t)r * N + i] = (float)r;
}
__global__ void __launch_bounds__(64) k_store16(float4* out, int N4) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  for (int r = 0; r < ROWS / 4; r++) out[(size_t)r * N4 + i] = make_float4(r, r, r, r);
}
int main() {
  const int N = 1 << 20;      // 64 rows x 1 Mi floats = 256 MiB: far beyond the 4 MB L2 and the 256 MB Infinity Cache of one pass
  float *a, *o;
  if (hipMalloc(&a, (size_t)ROWS * N * 4) != hipSuccess || hipMalloc(&o, (size_t)ROWS * N * 4) != hipSuccess) return 1;
  (void)hipMemset(a, 0, (size_t)ROWS * N * 4);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_soa4, dim3(N / 64), dim3(64), 0, 0, a, o, N);
    hipLaunchKernelGGL(k_quad4, dim3(N / 16), dim3(64), 0, 0, a, o, N);
    hipLaunchKernelGGL(k_vec16, dim3(N / 64), dim3(64), 0, 0, (const float4*)a, o, N);      // 16 rows of N float4 = the same 256 MiB
    hipLaunchKernelGGL(k_store4, dim3(N / 64), dim3(64), 0, 0, o, N);
    hipLaunchKernelGGL(k_store16, dim3(N / 64), dim3(64), 0, 0, (float4*)o, N);
  }
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  printf("bytes touched per kernel: %zu\n", (size_t)ROWS * N * 4);
  return 0;
}
