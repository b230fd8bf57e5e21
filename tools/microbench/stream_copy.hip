// Microbenchmark: the HBM bandwidth this box actually delivers to plain streaming kernels ("measure real peak with a stream-copy kernel
// on the box"), next to the nominal 8 TB/s that bench.py's `roofline.peak` uses.  Three kernels, 16 B per lane, grid-stride, 2048 workgroups of 256
// lanes (8 per CU), buffers of 2 GiB (far beyond the 256 MB Infinity Cache): read (sum into a register), write, copy (read + write).  Best of 10
// launches each, HIP events on the launch stream.  One JSON line.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/stream_copy.hip -o tools/microbench/stream_copy.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_read(const v4f* __restrict__ in, float* sink, size_t n4) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const v4f v = __builtin_nontemporal_load(in + i); s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) sink[0] = s;
}
__global__ void __launch_bounds__(256) k_write(v4f* __restrict__ out, size_t n4) {
  const v4f v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, out + i);
}
__global__ void __launch_bounds__(256) k_copy(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
int main() {
  const size_t bytes = (size_t)2 << 30, n4 = bytes / 16;
  v4f *a, *b; float* sink;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int grid = 2048;
  double best[3] = {0, 0, 0};
  for (int k = 0; k < 3; k++)
    for (int rep = 0; rep < 12; rep++) {
      CHECK(hipEventRecord(e0, 0));
      if (k == 0) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, sink, n4);
      if (k == 1) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n4);
      if (k == 2) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n4);
      CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double gbs = (k == 2 ? 2.0 : 1.0) * (double)bytes / (ms * 1e-3) / 1e9;
      if (rep >= 2 && gbs > best[k]) best[k] = gbs;
    }
  printf("{\"source\": \"tools/microbench/stream_copy.hip\", \"buffer_bytes\": %zu, \"read_GBs\": %.1f, \"write_GBs\": %.1f, \"copy_GBs_read_plus_write\": %.1f, \"nominal_peak_GBs\": 8000}\n",
         bytes, best[0], best[1], best[2]);
  return 0;
}
