// Microbenchmark: issue rate of v_mfma_f32_16x16x4_f32 as a function of the number of independent accumulator chains per wavefront
// and of the wavefronts per SIMD (1 or 2), with and without a VALU instruction feeding each MFMA's B operand.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_chains.hip -o /tmp/mfma_chains && /tmp/mfma_chains
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int CHAINS, int VALU>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  f32x4 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float x = b + threadIdx.x * 1e-6f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int c = 0; c < CHAINS; c++) {
        float bb = x;
        if (VALU) { bb = fmaf(bb, 1.0001f, 0.5f); bb = fmaxf(bb, -3.f); bb = bb * 0.999f; x = bb; }      // three dependent VALU ops ahead of the MFMA
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, acc[c], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS, int VALU>
static void run(int blocks, const char* tag) {
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CHAINS, VALU>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL((k<CHAINS, VALU>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_wave = (double)iters * 8 * CHAINS;
  const double waves_per_simd = blocks / 256.0;                  // 4 wavefronts per block = one per SIMD of a CU; 256 CUs
  const double ns_per_mfma_per_simd = ms * 1e6 / (mfma_per_wave * waves_per_simd);
  const double tflops = mfma_per_wave * (blocks * 4.0) * 2048.0 / (ms * 1e-3) / 1e12;
  printf("%-22s chains %d  valu %d  blocks %4d (%.0f wave/SIMD): %7.3f ms  %6.2f ns per MFMA per SIMD  %6.1f TFLOP/s\n", tag, CHAINS, VALU, blocks, waves_per_simd, ms, ns_per_mfma_per_simd, tflops);
  hipFree(out);
}

int main() {
  run<1, 0>(256, "pure"); run<2, 0>(256, "pure"); run<4, 0>(256, "pure"); run<8, 0>(256, "pure");
  run<1, 0>(512, "pure"); run<2, 0>(512, "pure"); run<4, 0>(512, "pure");
  run<2, 1>(256, "valu-fed"); run<4, 1>(256, "valu-fed"); run<2, 1>(512, "valu-fed"); run<4, 1>(512, "valu-fed");
  return 0;
}
