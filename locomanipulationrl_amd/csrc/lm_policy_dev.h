// lm_policy_dev.h -- device-side pieces of the policy kernels that more than one translation unit needs: the MLP forward of one
// 16-sample tile on the four wavefronts of a block (lm_policy.hip: k_mlp_forward; lm_engine.hip: the persistent rollout kernel, which
// runs it between two physics steps without leaving the kernel), the gaussian action sampling of its epilogue, and the ELU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lm_rng.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// ---- fp32 matrix products on the fp16 matrix pipe (GNN tile, round 4).  v_mfma_f32_16x16x32_f16 contracts K = 32 in ONE instruction of 16
// cycles where v_mfma_f32_16x16x4_f32 needs 8 instructions of 32 (and reaches 46-48 with fresh operands): 16 x the rate.  Every fp32
// operand is split into two halves, x = hi + lo with hi = fp16(x) and lo = fp16(x - hi) (both round-to-nearest; x - hi is exact in fp32), which
// carries 22 bits of x (|x - hi - lo| <= 2^-23 |x|: the size of an fp32 rounding), and a product is the four MFMAs lo.lo + lo.hi + hi.lo + hi.hi
// accumulated in fp32, smallest terms first.  fp16 range: |x| < 65504 (observations are clipped to +-5 by the scaler, activations of the 32-wide
// layers stay orders of magnitude below); a lo below the fp16 subnormals (|x| < 6e-5 or so) loses bits that are < 6e-8 absolute.
// 12 vector instructions per 8 values: v_cvt_pk_f16_f32 for a pair of hi halves, then lo = fp16(x - hi) as ONE mixed-precision FMA per value
// (v_fma_mixlo / mixhi_f16: fp16 source hi, fp32 source x, fp32 arithmetic, result rounded once to fp16 into the low / high half) - written as
// plain conversions the compiler emits 20 (two packed converts, four unpacking converts and a packed subtract per pair).
// HAZARD (gfx940+ "dst-sel forwarding"): a VALU instruction that writes HALF of a register must not be followed directly by an instruction
// that consumes that register - and v_fma_mixhi_f16 consumes its own destination (it preserves the low half).  The hazard recogniser does
// not look inside inline assembly (a first version with mixlo directly followed by mixhi on the same register returned stale halves,
// depending on timing), so the eight instructions are ordered by hand - four mixlo, then four mixhi: every mixhi is four instructions behind
// its mixlo - and the block ends with one wait state before the compiler's consumers.
__device__ __forceinline__ void split_f16(const float (&x)[8], f16x8& hi, f16x8& lo) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  u32x4 H, L;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const f32x2 v = {x[2 * i], x[2 * i + 1]};
    H[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));            // v_cvt_pk_f16_f32 (round to nearest even)
  }
#ifdef LM_GNN_NOASM
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const f32x2 v = {x[2 * i], x[2 * i + 1]};
    const f32x2 r = v - __builtin_convertvector(__builtin_bit_cast(f16x2, H[i]), f32x2);
    L[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
  }
#else
  uint32_t l0, l1, l2, l3;
  asm("v_fma_mixlo_f16 %0, %4, -1.0, %8 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %1, %5, -1.0, %10 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %2, %6, -1.0, %12 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %3, %7, -1.0, %14 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %4, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %1, %5, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %2, %6, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %3, %7, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "s_nop 0"
      : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
      : "v"(H[0]), "v"(H[1]), "v"(H[2]), "v"(H[3]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  L[0] = l0; L[1] = l1; L[2] = l2; L[3] = l3;
#endif
  hi = __builtin_bit_cast(f16x8, H); lo = __builtin_bit_cast(f16x8, L);
}
__device__ __forceinline__ f32x4 mfma_split(const f16x8& ahi, const f16x8& alo, const f16x8& bhi, const f16x8& blo, f32x4 acc) {
#ifndef LM_SPLIT3      // A/B switch (tools/ab_build.py split3=-DLM_SPLIT3): without the lo.lo term (2^-22 relative) - measured: GNN forward unchanged, MLP forward -4 %,
                       // same error against float64 on the test inputs; the four-term product is kept (the tiles are not bound by the matrix pipe)
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, blo, acc, 0, 0, 0);
#endif
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi, acc, 0, 0, 0);
}

// Block barrier for data exchanged through LDS only: waits for this wavefront's LDS operations, not for its global loads and stores
// (__syncthreads() also drains vmcnt, which would stall every barrier on the weight prefetch that is deliberately kept in flight across it).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ELU with exp(x) - 1 on the hardware exponential (v_exp_f32): absolute error < 1e-7, against ~30 instructions for expm1f
// No compare / select: e^x - 1 >= x everywhere, so ELU(x) is the median of (x, 0, e^x - 1) - x < e^x - 1 < 0 below zero, 0 < x < e^x - 1 above
// (v_med3_f32; one instruction less per activation, and the activations are the VALU work that competes with the MFMAs of the policy tiles).
__device__ __forceinline__ float elu(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, __expf(x) - 1.0f); }
// two activations at once, ELU(p + q): the add, the scaling of the exponent and the -1 are packed-fp32 instructions (v_pk_add_f32 / v_pk_mul_f32),
// 7 vector instructions per pair instead of 10
typedef __attribute__((ext_vector_type(2))) float elu_f32x2;
__device__ __forceinline__ void elu_sum2(float p0, float p1, float q0, float q1, float& z0, float& z1) {
  const elu_f32x2 x = (elu_f32x2){p0, p1} + (elu_f32x2){q0, q1};
  const elu_f32x2 t = x * (elu_f32x2){1.4426950408889634f, 1.4426950408889634f};
  const elu_f32x2 e = (elu_f32x2){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} - (elu_f32x2){1.0f, 1.0f};
  z0 = __builtin_amdgcn_fmed3f(x.x, 0.0f, e.x); z1 = __builtin_amdgcn_fmed3f(x.y, 0.0f, e.y);
}


#ifdef LM_GNN_STAMPS
__shared__ unsigned long long lm_gnn_stamp_lds[4][16];      // diagnostic builds only (tools/stamp_profile_gnn.py)
#endif
#ifdef LM_GNN_STAMPS
#define MLP_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    if (lane == 0) { lm_gnn_stamp_lds[wave][k] += t_ - lm_gnn_stamp_lds[wave][15]; lm_gnn_stamp_lds[wave][15] = t_; } } while (0)
#else
#define MLP_STAMP(k) do { } while (0)
#endif

// ---- counter-based standard normal for the fused action sampling (same generator as lm_engine.hip dr_sample, stream 9)
__device__ __forceinline__ float ro_normal(uint32_t seed, uint32_t env, uint32_t key, uint32_t idx) {
  // actions 2p and 2p+1 are the cosine and sine branches of one Box-Muller pair
  float u1, u2; lm_rng_pair(lm_rng_base(seed, LM_RNG_STREAM_ACTION_SAMPLING, env, key), idx >> 1, &u1, &u2);
  float sn, cs; sincosf(6.283185307179586f * u2, &sn, &cs);
  return sqrtf(-2.0f * logf(u1)) * ((idx & 1U) ? sn : cs);
}
// (episode_count, progress_buf) identifies an env-step: progress restarts at every reset and the episode count moves on
__device__ __forceinline__ uint32_t ro_key(const int64_t* __restrict__ cnt, int N, int env) {
  return ((uint32_t)cnt[5 * (size_t)N + env] << 16) + (uint32_t)cnt[4 * (size_t)N + env];
}
struct SampleArgs { const float* log_std; const int64_t* cnt; uint32_t seed; float* actions; float* logp; };      // log_std == nullptr: no sampling


// ------------------------------------------------------------------------------------------------
// MLP policy (scripts/skrl_ppo_locomotion.py:30-40): shared trunk 64 -> 256 -> 128 -> 64 (ELU) -> mean (12) + value (1),
// with the observation preprocessor folded in (skrl RunningStandardScaler: clamp((x - mean) / (sqrt(var) + eps), +-clip),
// passed as mean / inverse-std vectors).
//
// Round 4: fp32 products on the fp16 matrix pipe, as in the GNN tile (split_f16 / mfma_split above).  One v_mfma_f32_16x16x32_f16 contracts a
// K-BLOCK of 32 input features for a 16-row output block; an fp32 product is its four half products.  912 v_mfma_f32_16x16x4_f32 of 32+ cycles per
// 16-sample tile became 456 of 16.  The WEIGHTS are split on the host (policies/mlp_model.py pack_mlp_params): the block keeps its size, every
// 32-bit word now holds two fp16 halves.  The ACTIVATIONS are split once by their producer (the observation loader, the epilogue of a layer)
// and pass through LDS already in B-operand form, hi and lo, one 16-byte read each per K-block.
//   packed block (32-bit words): obs_mean NOBS | obs_inv_std NOBS | clip 1 (+3 pad) | W1q [16][KB1][64][8] | b1 256 | W2q [8][8][64][8] | b2 128 |
//                 W3q [4][4][64][8] | b3 64 | Whq [1][2][64][8] (rows 0..11 mean, 12 value, 13..15 zero) | bh 16
//   Wq [mb][kb][lane = n + 16 g][8]: words 0..3 = the fp16 hi halves of W[16 mb + n][col(kb, g, e)], e = 0..7, two per word (even e in the low
//   half), words 4..7 the lo halves.  col = the input feature that k-slot (g, e) of K-block kb carries:
//     layer 1 (observation, natural order):   col = 32 kb + 8 g + e            (columns >= NOBS are zero: 88 is padded to 96)
//     later layers (a previous layer's output): col = 16 (2 kb + (e >> 2)) + 4 g + (e & 3)   = rows 4 g .. 4 g + 3 of output blocks 2 kb, 2 kb + 1,
//   i.e. exactly what lane (n, g) of the producing MFMAs holds - a producer writes its four values where its consumer's lane (n, g) reads them.
__host__ __device__ constexpr int mlp_kb1(int nobs) { return (nobs + 31) / 32; }
__host__ __device__ constexpr int mlp_off_mean(int) { return 0; }
__host__ __device__ constexpr int mlp_off_istd(int nobs) { return nobs; }
__host__ __device__ constexpr int mlp_off_clip(int nobs) { return 2 * nobs; }
__host__ __device__ constexpr int mlp_off_w1(int nobs) { return 2 * nobs + 4; }
__host__ __device__ constexpr int mlp_off_b1(int nobs) { return mlp_off_w1(nobs) + 16 * mlp_kb1(nobs) * 512; }
__host__ __device__ constexpr int mlp_off_w2(int nobs) { return mlp_off_b1(nobs) + 256; }
__host__ __device__ constexpr int mlp_off_b2(int nobs) { return mlp_off_w2(nobs) + 128 * 256; }
__host__ __device__ constexpr int mlp_off_w3(int nobs) { return mlp_off_b2(nobs) + 128; }
__host__ __device__ constexpr int mlp_off_b3(int nobs) { return mlp_off_w3(nobs) + 64 * 128; }
__host__ __device__ constexpr int mlp_off_wh(int nobs) { return mlp_off_b3(nobs) + 64; }
__host__ __device__ constexpr int mlp_off_bh(int nobs) { return mlp_off_wh(nobs) + 16 * 64; }
__host__ __device__ constexpr int mlp_params(int nobs) { return mlp_off_bh(nobs) + 16; }

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
// Activations in LDS, B-operand form: K-block kb, lane group g, sample n -> 8 words (hi halves of k-slots 0..7 | lo halves); consecutive samples
// are 32 bytes apart, so the 16-byte reads of a group of eight lanes fall on disjoint banks.
__device__ __forceinline__ int mlp_act_word(int kb, int g, int n) { return ((kb * 4 + g) * 16 + n) * 8; }
// four consecutive k-slots (half = 0: e 0..3, half = 1: e 4..7) of one (kb, g, n): split and stored as two 8-byte writes
__device__ __forceinline__ void mlp_store4(uint32_t* sAct, int kb, int g, int n, int half, float v0, float v1, float v2, float v3) {
  const f32x2 a = {v0, v1}, b = {v2, v3};
  const uint32_t h0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, f16x2)), h1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, f16x2));
  uint32_t l0, l1;
  // (dst-sel forwarding hazard, see split_f16: one instruction between a mixlo and the mixhi on the same register, one wait state at the end)
  asm("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %1, %3, -1.0, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %2, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %1, %3, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "s_nop 0"
      : "=&v"(l0), "=&v"(l1) : "v"(h0), "v"(h1), "v"(v0), "v"(v1), "v"(v2), "v"(v3));
  uint32_t* p = sAct + mlp_act_word(kb, g, n) + 2 * half;
  *reinterpret_cast<u32x2*>(p) = (u32x2){h0, h1};
  *reinterpret_cast<u32x2*>(p + 4) = (u32x2){l0, l1};
}
__device__ __forceinline__ void mlp_load_b(const uint32_t* sAct, int kb, int g, int n, f16x8& bh, f16x8& bl) {
  const uint32_t* p = sAct + mlp_act_word(kb, g, n);
  bh = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(p)); bl = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(p + 4));
}
// epilogue of an output block mb (rows 4 g + k of sample n in acc): bias, activation, then into the next layer's B-operand form
__device__ __forceinline__ void mlp_epilogue(uint32_t* sOut, int mb, int g, int n, const f32x4& acc, const float* b4) {
  mlp_store4(sOut, mb >> 1, g, n, mb & 1, elu(acc[0] + b4[0]), elu(acc[1] + b4[1]), elu(acc[2] + b4[2]), elu(acc[3] + b4[3]));
}

#define MLP_O_STRIDE 20      // head outputs sO[row][sample] (fp32)
template <int NOBS>
struct MlpSmem { uint32_t sX[mlp_kb1(NOBS) * 512] __attribute__((aligned(16))), sH1[8 * 512] __attribute__((aligned(16))), sH2[4 * 512] __attribute__((aligned(16))),
                 sH3[2 * 512] __attribute__((aligned(16))); float sO[16 * MLP_O_STRIDE], sLp[12 * 16]; };

// normalised, clipped observation tile of samples s0 .. s0+15 into sX (B-operand form), by NT threads (t = 0 .. NT-1); features >= NOBS are zero
template <int NOBS, bool LDS_OBS, int NT>
__device__ __forceinline__ void mlp_load_obs(const float* obs, float obs_clip, int B, int s0, const float* __restrict__ W, uint32_t* sX, int t) {
  constexpr int KP = 32 * mlp_kb1(NOBS);
  for (int idx = t; idx < 16 * (KP / 4); idx += NT) {
    const int sm = idx / (KP / 4), c0 = (idx - sm * (KP / 4)) * 4, sample = min(s0 + sm, B - 1);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < NOBS) {
      float4 o4;
      if (LDS_OBS) {
        o4 = *reinterpret_cast<const float4*>(obs + sm * NOBS + c0);
        o4.x = fminf(fmaxf(o4.x, -obs_clip), obs_clip); o4.y = fminf(fmaxf(o4.y, -obs_clip), obs_clip);
        o4.z = fminf(fmaxf(o4.z, -obs_clip), obs_clip); o4.w = fminf(fmaxf(o4.w, -obs_clip), obs_clip);
      } else {
        o4 = *reinterpret_cast<const float4*>(obs + (size_t)sample * NOBS + c0);
      }
      const float clip = W[mlp_off_clip(NOBS)], o[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
      for (int i = 0; i < 4; i++) v[i] = fminf(fmaxf((o[i] - W[mlp_off_mean(NOBS) + c0 + i]) * W[mlp_off_istd(NOBS) + c0 + i], -clip), clip);
    }
    mlp_store4(sX, c0 >> 5, (c0 & 31) >> 3, sm, (c0 & 7) >> 2, v[0], v[1], v[2], v[3]);
  }
}

// One block = 16 samples on the 4 wavefronts (= 4 SIMDs) of a CU: every layer's output blocks are dealt round-robin to the wavefronts (block
// mb = wave + 4 j).  A wavefront's weights form one sequence of ITEMS - (output block, K-block) pairs, 8 words per lane, K-block-major within a
// layer so that one B operand read serves all the blocks the wavefront owns - streamed in chunks of MLP_CH items with MLP_RING - 1 chunks in
// flight behind the MFMAs, across the layer boundaries (weights do not depend on activations).
#ifndef MLP_CH
#define MLP_CH 4
#endif
#ifndef MLP_RING
#define MLP_RING 3
#endif
template <int OUT_BLOCKS, int KB>
struct MlpLayer {
  static constexpr int OWNED = (OUT_BLOCKS + 3) / 4, TOTAL = OWNED * KB, NCH = (TOTAL + MLP_CH - 1) / MLP_CH;
  static constexpr bool FULL = (OUT_BLOCKS % 4) == 0;       // every wavefront owns exactly OWNED output blocks (else: only wavefront 0 calls)
  // item s of the wavefront's sequence: K-block kb = s / OWNED, owned block j = s % OWNED
  static __device__ __forceinline__ void issue(const float* __restrict__ Wq, int wave, int lane, int c, u32x4* dst) {
#pragma unroll
    for (int i = 0; i < MLP_CH; i++) {
      const int s = c * MLP_CH + i;
      if (s < TOTAL) {
        const int kb = s / OWNED, j = s - kb * OWNED, mb = wave + 4 * j;
        const u32x4* src = reinterpret_cast<const u32x4*>(Wq + ((size_t)(mb * KB + kb) * 64 + lane) * 8);
        dst[2 * i] = src[0]; dst[2 * i + 1] = src[1];
      }
    }
    __builtin_amdgcn_sched_barrier(0);      // keep the whole chunk's loads ahead of the MFMAs that follow (the scheduler would sink them to their uses)
  }
  static __device__ __forceinline__ void load_bias(const float* __restrict__ bias, int wave, int g, float* dst) {
#pragma unroll
    for (int j = 0; j < OWNED; j++)
#pragma unroll
      for (int k = 0; k < 4; k++) dst[4 * j + k] = bias[16 * (wave + 4 * j) + 4 * g + k];
  }
  // MFMAs of chunk c; acc[j] carries over between the chunks of a layer.  LAST = the head: plain fp32 rows into sO, no activation
  template <bool LAST>
  static __device__ __forceinline__ void compute(const float* breg, const uint32_t* sIn, uint32_t* sOut, float* sO, int wave, int n, int g, int c, const u32x4* a, f32x4* acc) {
    f16x8 bh, bl;
#pragma unroll
    for (int i = 0; i < MLP_CH; i++) {
      const int s = c * MLP_CH + i;
      if (s < TOTAL) {
        const int kb = s / OWNED, j = s - kb * OWNED, mb = wave + 4 * j;
        if (j == 0 || i == 0) mlp_load_b(sIn, kb, g, n, bh, bl);
        if (kb == 0) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[j] = mfma_split(__builtin_bit_cast(f16x8, a[2 * i]), __builtin_bit_cast(f16x8, a[2 * i + 1]), bh, bl, acc[j]);
        if (kb == KB - 1) {
          if (LAST) {
#pragma unroll
            for (int k = 0; k < 4; k++) sO[(4 * g + k) * MLP_O_STRIDE + n] = acc[j][k] + breg[4 * j + k];
          } else mlp_epilogue(sOut, mb, g, n, acc[j], breg + 4 * j);
        }
      }
    }
  }
};

// Forward pass of samples s0 .. s0+15 by the 256 threads of a block (t = thread index in the block; contains block barriers).
// LDS_OBS = false: `obs` is the global (B, NOBS) observation matrix.  LDS_OBS = true: `obs` is a [16][NOBS] tile in LDS holding the
// UNCLIPPED observations of exactly these samples; they are clamped to +-obs_clip first, i.e. the values the step kernel returns.
// mean may be null.  With SA.log_std set, wavefront 0 samples the actions and their log-probabilities in the epilogue.
template <int NOBS, bool LDS_OBS>
__device__ __forceinline__ void mlp_block(const float* obs, float obs_clip, int B, int s0, const float* __restrict__ W, float* __restrict__ mean,
                                          float* __restrict__ value, const SampleArgs& SA, MlpSmem<NOBS>& M, int t) {
  const int wave = t >> 6, lane = t & 63, n = lane & 15, g = lane >> 4;
  typedef MlpLayer<16, mlp_kb1(NOBS)> L1; typedef MlpLayer<8, 8> L2; typedef MlpLayer<4, 4> L3; typedef MlpLayer<1, 2> L4;
  static_assert(L1::FULL && L2::FULL && L3::FULL && !L4::FULL, "wavefront ownership");
  const float *W1 = W + mlp_off_w1(NOBS), *W2 = W + mlp_off_w2(NOBS), *W3 = W + mlp_off_w3(NOBS), *W4 = W + mlp_off_wh(NOBS);
  float br1[4 * L1::OWNED], br2[4 * L2::OWNED], br3[4 * L3::OWNED], br4[4];
  L1::load_bias(W + mlp_off_b1(NOBS), wave, g, br1); L2::load_bias(W + mlp_off_b2(NOBS), wave, g, br2);
  L3::load_bias(W + mlp_off_b3(NOBS), wave, g, br3); L4::load_bias(W + mlp_off_bh(NOBS), 0, g, br4);
  mlp_load_obs<NOBS, LDS_OBS, 256>(obs, obs_clip, B, s0, W, M.sX, t);
  // The chunks of all layers form one sequence sq = 0 .. NB-1 (the head's single chunk last, wavefront 0 only); chunk sq + MLP_RING - 1 is issued
  // before chunk sq is computed, into a ring of register buffers.
  constexpr int N1 = L1::NCH, N2 = L2::NCH, N3 = L3::NCH, NB = N1 + N2 + N3;
  auto issue = [&](int sq, u32x4* dst) {
    if (sq < N1) L1::issue(W1, wave, lane, sq, dst);
    else if (sq < N1 + N2) L2::issue(W2, wave, lane, sq - N1, dst);
    else if (sq < NB) L3::issue(W3, wave, lane, sq - N1 - N2, dst);
    else if (sq == NB && wave == 0) L4::issue(W4, 0, lane, 0, dst);
  };
  MLP_STAMP(0);      // biases, observation tile
  u32x4 abuf[MLP_RING][2 * MLP_CH];
#pragma unroll
  for (int sq = 0; sq < MLP_RING - 1; sq++) issue(sq, abuf[sq]);
  lds_barrier();
  MLP_STAMP(1);      // first chunks issued, barrier
  f32x4 acc[4];
#pragma unroll
  for (int sq = 0; sq < NB; sq++) {
    issue(sq + MLP_RING - 1, abuf[(sq + MLP_RING - 1) % MLP_RING]);
    if (sq < N1) L1::template compute<false>(br1, M.sX, M.sH1, M.sO, wave, n, g, sq, abuf[sq % MLP_RING], acc);
    else if (sq < N1 + N2) L2::template compute<false>(br2, M.sH1, M.sH2, M.sO, wave, n, g, sq - N1, abuf[sq % MLP_RING], acc);
    else L3::template compute<false>(br3, M.sH2, M.sH3, M.sO, wave, n, g, sq - N1 - N2, abuf[sq % MLP_RING], acc);
    MLP_STAMP(2 + (sq < N1 ? 0 : sq < N1 + N2 ? 2 : 4));      // compute of a layer's chunks
    if (sq == N1 - 1 || sq == N1 + N2 - 1 || sq == NB - 1) { lds_barrier(); MLP_STAMP(3 + (sq < N1 ? 0 : sq < N1 + N2 ? 2 : 4)); }      // its barrier
  }
  if (wave != 0) return;
  L4::template compute<true>(br4, M.sH3, nullptr, M.sO, 0, n, g, 0, abuf[NB % MLP_RING], acc);
  __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
  const float* sO = M.sO;
  const int smp = s0 + n;
  const bool valid = smp < B;
  float lp = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int j = 4 * g + i;                      // head output row: 0..11 action means, 12 value
    const float v = sO[j * MLP_O_STRIDE + n];
    if (valid && j < 12 && mean) mean[(size_t)smp * 12 + j] = v;
    if (valid && j == 12) value[smp] = v;
    if (SA.log_std && j < 12 && valid) {
      const float ls = SA.log_std[j], eps = ro_normal(SA.seed, (uint32_t)smp, ro_key(SA.cnt, B, smp), (uint32_t)j);
      SA.actions[(size_t)smp * 12 + j] = fmaf(expf(ls), eps, v);
      lp += -0.5f * eps * eps - ls - 0.9189385332046727f;          // log N(a; mean, std) with (a - mean) / std = eps
    }
  }
  if (SA.log_std) { lp += __shfl_xor(lp, 16); lp += __shfl_xor(lp, 32); if (valid && g == 0) SA.logp[smp] = lp; }
  MLP_STAMP(8);      // head + sampling (wavefront 0)
}

// ------------------------------------------------------------------------------------------------
// MLP tile with RESIDENT weights (persistent rollout kernel, lm_engine.hip k_rollout_mlp).  In the persistent kernel wavefront 0 steps the
// physics and the other three - the policy wavefronts P = 0, 1, 2 - are idle meanwhile with 512 registers per lane each: they load every weight
// ONCE per rollout and keep it in registers (320 / 304 / 288 words per lane), so a forward is MFMAs and LDS traffic only.  Output blocks are dealt
// to the policy wavefronts round-robin (layer 3: {0}, {1}, {2, 3}; the head is computed by each of them, so that the action sampling - a Philox
// draw, a logarithm and a sine / cosine per action - is spread over all 192 lanes, one action each); an output block accumulates its K-blocks in
// the order of the streaming tile, each as lo.lo + lo.hi + hi.lo + hi.hi, so both tiles produce the same bits.
template <int P, int OUT_BLOCKS, bool LAYER3> struct ResOwn {
  static constexpr int CNT = LAYER3 ? (P == 2 ? 2 : 1) : (OUT_BLOCKS == 1 ? 1 : (OUT_BLOCKS - P + 2) / 3);      // the head (one block) is computed by all three
  static __device__ __forceinline__ constexpr int mb(int j) { return LAYER3 ? (P == 2 ? 2 + j : P) : (OUT_BLOCKS == 1 ? 0 : P + 3 * j); }
};
template <int P, int OUT_BLOCKS, int KB, bool LAYER3>
struct ResLayer {
  typedef ResOwn<P, OUT_BLOCKS, LAYER3> O;
  static constexpr int CNT = O::CNT, NW = (CNT * KB > 0) ? 2 * CNT * KB : 1, NBR = CNT > 0 ? 4 * CNT : 1;
  static __device__ __forceinline__ void load(const float* __restrict__ Wq, const float* __restrict__ bias, int lane, int g, u32x4* w, float* br) {
#pragma unroll
    for (int j = 0; j < CNT; j++) {
#pragma unroll
      for (int kb = 0; kb < KB; kb++) {
        const u32x4* src = reinterpret_cast<const u32x4*>(Wq + ((size_t)(O::mb(j) * KB + kb) * 64 + lane) * 8);
        w[2 * (j * KB + kb)] = src[0]; w[2 * (j * KB + kb) + 1] = src[1];
      }
#pragma unroll
      for (int k = 0; k < 4; k++) br[4 * j + k] = bias[16 * O::mb(j) + 4 * g + k];
    }
  }
  static __device__ __forceinline__ void compute(const u32x4* w, const float* br, const uint32_t* sIn, uint32_t* sOut, int n, int g) {
    // the layer's input (the B operands of every output block) in registers first: read where they are used, each LDS round trip is exposed
    f16x8 bh[KB], bl[KB];
#pragma unroll
    for (int kb = 0; kb < KB; kb++) mlp_load_b(sIn, kb, g, n, bh[kb], bl[kb]);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[CNT > 0 ? CNT : 1];
#pragma unroll
    for (int j = 0; j < CNT; j++) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; kb++)
#pragma unroll
      for (int j = 0; j < CNT; j++)
        acc[j] = mfma_split(__builtin_bit_cast(f16x8, w[2 * (j * KB + kb)]), __builtin_bit_cast(f16x8, w[2 * (j * KB + kb) + 1]), bh[kb], bl[kb], acc[j]);
#pragma unroll
    for (int j = 0; j < CNT; j++) mlp_epilogue(sOut, O::mb(j), g, n, acc[j], br + 4 * j);
  }
};
#ifndef MLP_RES_STAMP
#define MLP_RES_STAMP(P, k) do { } while (0)      // diagnostic builds of lm_engine.hip define it (tools/stamp_profile_rollout.py)
#endif
#define MLP_RES_BARRIERS 5      // block barriers inside one resident tile; the physics wavefront executes the same number while the tile runs

template <int NOBS, int P> struct MlpResRegs {
  typedef ResLayer<P, 16, mlp_kb1(NOBS), false> L1; typedef ResLayer<P, 8, 8, false> L2; typedef ResLayer<P, 4, 4, true> L3; typedef ResLayer<P, 1, 2, false> L4;
  u32x4 w1[L1::NW], w2[L2::NW], w3[L3::NW], w4[L4::NW]; float b1[L1::NBR], b2[L2::NBR], b3[L3::NBR], b4[L4::NBR];
  __device__ __forceinline__ void load(const float* __restrict__ W, int lane, int g) {
    L1::load(W + mlp_off_w1(NOBS), W + mlp_off_b1(NOBS), lane, g, w1, b1); L2::load(W + mlp_off_w2(NOBS), W + mlp_off_b2(NOBS), lane, g, w2, b2);
    L3::load(W + mlp_off_w3(NOBS), W + mlp_off_b3(NOBS), lane, g, w3, b3); L4::load(W + mlp_off_wh(NOBS), W + mlp_off_bh(NOBS), lane, g, w4, b4);
  }
};

// one forward of samples s0 .. s0+15 by the policy wavefronts; tp = 0 .. 191 the thread index among them.  Same arithmetic, same order as mlp_block.
template <int NOBS, int P, bool LDS_OBS>
__device__ __forceinline__ void mlp_res_tile(const float* obs, float obs_clip, int B, int s0, const float* __restrict__ W, const MlpResRegs<NOBS, P>& R,
                                             float* __restrict__ value, const SampleArgs& SA, MlpSmem<NOBS>& M, int tp) {
  typedef MlpResRegs<NOBS, P> RG;
  const int lane = tp & 63, n = lane & 15, g = lane >> 4;
  mlp_load_obs<NOBS, LDS_OBS, 192>(obs, obs_clip, B, s0, W, M.sX, tp);
  MLP_RES_STAMP(P, 0);      // wait for the physics + observation tile
  lds_barrier();
  MLP_RES_STAMP(P, 1);
  RG::L1::compute(R.w1, R.b1, M.sX, M.sH1, n, g);
  MLP_RES_STAMP(P, 2);
  lds_barrier();
  MLP_RES_STAMP(P, 3);
  RG::L2::compute(R.w2, R.b2, M.sH1, M.sH2, n, g);
  MLP_RES_STAMP(P, 4);
  lds_barrier();
  MLP_RES_STAMP(P, 5);
  RG::L3::compute(R.w3, R.b3, M.sH2, M.sH3, n, g);
  MLP_RES_STAMP(P, 6);
  lds_barrier();
  MLP_RES_STAMP(P, 7);
  {
    // head: rows 0..11 action means, 12 value; lane (n, g) holds rows 4g .. 4g+3 of sample n
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      f16x8 bh, bl; mlp_load_b(M.sH3, kb, g, n, bh, bl);
      acc = mfma_split(__builtin_bit_cast(f16x8, R.w4[2 * kb]), __builtin_bit_cast(f16x8, R.w4[2 * kb + 1]), bh, bl, acc);
    }
    float hv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) hv[k] = acc[k] + R.b4[k];
    const int smp = s0 + n;
    const bool valid = smp < B;
    if (P == 2 && g == 3 && valid) value[smp] = hv[0];
    // one action per lane: lane group g < 3 of wavefront P samples action 4g + P, lane group 3 samples action 4P + 3 (its mean lives in lane group P)
    const float m3 = __shfl(hv[3], n + 16 * P);
    const int j = (g < 3) ? 4 * g + P : 4 * P + 3;
    const float v = (g < 3) ? hv[P] : m3;
    if (SA.log_std && valid) {
      const float ls = SA.log_std[j], eps = ro_normal(SA.seed, (uint32_t)smp, ro_key(SA.cnt, B, smp), (uint32_t)j);
      SA.actions[(size_t)smp * 12 + j] = fmaf(expf(ls), eps, v);
      M.sLp[j * 16 + n] = -0.5f * eps * eps - ls - 0.9189385332046727f;          // log N(a; mean, std) with (a - mean) / std = eps
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // the sampled actions are in the L2 before the physics wavefront is released to read them
  }
  MLP_RES_STAMP(P, 8);      // head + sampling
  lds_barrier();
  MLP_RES_STAMP(P, 9);
  if (P == 2 && SA.log_std && g == 0 && s0 + n < B) {      // (while the physics runs) the log-probability, summed in the order of the streaming tile / k_sample_actions
    float L[3];
#pragma unroll
    for (int q = 0; q < 3; q++) { float lp = 0.f;
#pragma unroll
      for (int i = 0; i < 4; i++) lp += M.sLp[(4 * q + i) * 16 + n];
      L[q] = lp; }
    SA.logp[s0 + n] = (L[0] + L[1]) + (L[2] + 0.f);
  }
}

// Diagnostic build only (-DLM_GNN_STAMPS, tools/stamp_profile.py --gnn): GNN_STAMP(k) adds the shader cycles since the wavefront's previous stamp to
// bucket k of its row of an LDS array that k_gnn_forward copies out.  No stamp exists in the product build.
#ifdef LM_GNN_STAMPS
#define GNN_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    if ((lane & 63) == 0) { lm_gnn_stamp_lds[WAVE][k] += t_ - lm_gnn_stamp_lds[WAVE][15]; lm_gnn_stamp_lds[WAVE][15] = t_; } } while (0)
#else
#define GNN_STAMP(k) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------ GNN policy tile
// (restates scripts/graph_model_orebot_ov.py:11-241; the mapping is described at the top of lm_policy.hip)
#define GNN_NODES 13
#define GNN_EDGES 24
#define GNN_H 32
#define GNN_SAMPLES 16
#define GNN_Q_STRIDE 36                    // floats per (node, sample) row of Q in LDS: 32 features + 4 pad, the 16-byte accesses of a lane group fall on disjoint banks
#define GNN_Q_BUF (GNN_NODES * GNN_SAMPLES * GNN_Q_STRIDE)

// parameter block offsets (floats)
#define OFF_IN1_W 0                      // (32,16)
#define OFF_IN1_B 512
#define OFF_IN2_W 544                    // (32,4)
#define OFF_IN2_B 672
#define OFF_LAYER0 704
#define LAYER_STRIDE 3136                // W1 (32,64) 2048, b1 32, W2 (32,32) 1024, b2 32
#define OFF_ACT_W (704 + 3 * 3136)       // 32
#define OFF_ACT_B (OFF_ACT_W + 32)
#define OFF_VAL_W (OFF_ACT_B + 1)
#define OFF_VAL_B (OFF_VAL_W + 32)
#define OFF_OBS_MEAN (OFF_VAL_B + 1)   // observation preprocessor (skrl RunningStandardScaler): mean 64, inverse std 64, clip 1
#define OFF_OBS_ISTD (OFF_OBS_MEAN + 64)
#define OFF_OBS_CLIP (OFF_OBS_ISTD + 64)
#define GNN_PARAMS (OFF_OBS_CLIP + 1)

// edges (source -> target): 0->{1..4}, i->i+4 (1..4), i->i+4 (5..8), then the 12 reverses (graph_model_orebot_ov.py:142-159)
__device__ __forceinline__ constexpr int edge_src(int e) { return e < 4 ? 0 : (e < 8 ? e - 3 : (e < 12 ? e - 3 : (e < 16 ? e - 11 : (e < 20 ? e - 11 : e - 11)))); }
__device__ __forceinline__ constexpr int edge_tgt(int e) { return e < 4 ? e + 1 : (e < 8 ? e + 1 : (e < 12 ? e + 1 : (e < 16 ? 0 : (e < 20 ? e - 15 : e - 15)))); }
// obs column of feature k (0..3) of joint node n (1..12): [0.3 q, 0.3 qd, action, last action] of that joint (:115-126)
__device__ __forceinline__ constexpr int joint_col(int n, int k) { return 16 + 12 * k + (n <= 4 ? n - 1 : (n <= 8 ? 4 + 2 * (n - 5) : 5 + 2 * (n - 9))); }

// One block = 16 samples on the 4 wavefronts (= 4 SIMDs) of a CU.  Every wavefront OWNS a set of graph nodes: it keeps their features
// in registers, computes their P / Q projections into LDS, and evaluates the messages along the edges that END in its nodes (max
// aggregation is order independent, so the result is bit-identical to a single-wavefront evaluation).  Ownership balances the matrix work.
// node j of wavefront w; MFMA count per layer = 32 per owned node + 16 per incoming edge: {0,1,9} 208, {2,3,4} 192, {5,6,7} 192, {8,10,11,12} 208
__device__ __forceinline__ constexpr int gnn_count(int w) { return (w == 3) ? 4 : 3; }
__device__ __forceinline__ constexpr int gnn_node(int w, int j) {
  return w == 0 ? (j == 0 ? 0 : (j == 1 ? 1 : 9)) : (w == 1 ? 2 + j : (w == 2 ? 5 + j : (j == 0 ? 8 : 9 + j)));
}
// incoming edges of node t (graph_model_orebot_ov.py:142-159): hub <- its 4 dof1 nodes; dof1 <- hub, dof2; dof2 <- dof1, dof3; dof3 <- dof2
__device__ __forceinline__ constexpr int gnn_nin(int t) { return t == 0 ? 4 : (t <= 8 ? 2 : 1); }
__device__ __forceinline__ constexpr int gnn_in(int t, int k) { return t == 0 ? 1 + k : (t <= 4 ? (k == 0 ? 0 : t + 4) : (t <= 8 ? (k == 0 ? t - 4 : t + 4) : t - 4)); }

// the incoming edges of a wavefront's nodes as one flat list (node after node): count, owning node slot, index among that node's edges
__device__ __forceinline__ constexpr int gnn_ne(int w) { int c = 0; for (int j = 0; j < gnn_count(w); j++) c += gnn_nin(gnn_node(w, j)); return c; }
__device__ __forceinline__ constexpr int gnn_ej(int w, int e) { int c = 0; for (int j = 0; j < gnn_count(w); j++) { const int nn = gnn_nin(gnn_node(w, j)); if (e < c + nn) return j; c += nn; } return 0; }
__device__ __forceinline__ constexpr int gnn_ek(int w, int e) { int c = 0; for (int j = 0; j < gnn_count(w); j++) { const int nn = gnn_nin(gnn_node(w, j)); if (e < c + nn) return e - c; c += nn; } return 0; }

// LDS_OBS = false: `obs` is the global (B, 64) observation matrix.  LDS_OBS = true: `obs` is a [16][64] tile in LDS with the UNCLIPPED
// observations of samples s0 .. s0+15, clamped to +-obs_clip first (= the values the step kernel returns).  mean may be null.
template <int WAVE, bool LDS_OBS>
__device__ __forceinline__ void gnn_body(const float* obs, float obs_clip, int B, int s0, const float* __restrict__ W, float* __restrict__ mean,
                                         float* __restrict__ value, const SampleArgs& SA, float* sPQ, float* sHm, float* sLp, const float* sOb, int lane) {
  constexpr int NC = gnn_count(WAVE);
  const int n = lane & 15, g = lane >> 4;
  // normalised observation column c of this lane's sample: staged once per block by gnn_block (sOb[c][sample])
  auto ob = [&](int c) { return sOb[c * GNN_SAMPLES + n]; };

  float wa[4][8], bias1[2][4];      // stage-1 weights of the current layer (A operand, gathered in the accumulator's k order)
  // (four consecutive k-steps of an output block are four consecutive floats of a weight row: 16-byte loads)
  auto load_stage1 = [&](int layer) {
    const float* W1 = W + OFF_LAYER0 + layer * LAYER_STRIDE; const float* b1 = W1 + 2048;
#pragma unroll
    for (int ob4 = 0; ob4 < 4; ob4++)
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(W1 + (16 * (ob4 & 1) + n) * 64 + 32 * (ob4 >> 1) + 16 * q + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; i++) wa[ob4][4 * q + i] = w4[i];
      }
#pragma unroll
    for (int mb = 0; mb < 2; mb++) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(b1 + 16 * mb + 4 * g);
#pragma unroll
      for (int i = 0; i < 4; i++) bias1[mb][i] = b4[i];
    }
    __builtin_amdgcn_sched_barrier(0);      // issue the loads here, not at their first use
  };
  load_stage1(0);
  f32x4 h[NC][2];            // features of the owned nodes, C layout: h[j][mb][i] = feature 16 mb + 4 g + i of sample n
  // ---- input layers (:97-104)
#pragma unroll
  for (int j = 0; j < NC; j++) {
    const int nd = gnn_node(WAVE, j);
    if (nd == 0) {      // hub node: Linear(16,32) on obs[0:16]
#pragma unroll
      for (int mb = 0; mb < 2; mb++) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; s++)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W[OFF_IN1_W + (16 * mb + n) * 16 + 4 * s + g], ob(4 * s + g), acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] += W[OFF_IN1_B + 16 * mb + 4 * g + i];
        h[j][mb] = acc;
      }
    } else {            // joint nodes: shared Linear(4,32) on [0.3 q, 0.3 qd, action, last action] of the joint
      const float bcol = ob(joint_col(nd, 0) + 12 * g);
#pragma unroll
      for (int mb = 0; mb < 2; mb++) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W[OFF_IN2_W + (16 * mb + n) * 4 + g], bcol, acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] += W[OFF_IN2_B + 16 * mb + 4 * g + i];
        h[j][mb] = acc;
      }
    }
  }
  // ---- three message-passing layers.  Per layer: stage 1 = P = W1[:, 0:32] h + b1 (kept in registers: only the owner needs it) and
  // Q = W1[:, 32:64] h (to LDS: the wavefronts that own the edge targets need it) of the owned nodes; stage 2 = messages along the edges that END
  // in the owned nodes, z = ELU(P_tgt + Q_src), y = W2 z, max-aggregated.  Both stages consume their B operand in the accumulator's own
  // feature order (k-step (mb', i) of lane group g = feature 16 mb' + 4 g + i; the weights are gathered in that order), so h, P and z never
  // leave the registers and Q moves as 16-byte LDS accesses.  Q is double-buffered by layer parity: stage 1 of layer L+1 writes the other
  // buffer while slower wavefronts still read layer L's, so there is ONE block barrier per layer (Q written -> Q read) and what a wavefront
  // does between two barriers is stage 2 of a layer plus stage 1 of the next - 208 / 192 / 192 / 208 MFMAs, against 128 + 112 on the
  // critical path with a barrier between the stages.  Weights are loaded a stage ahead (they do not depend on activations).
  GNN_STAMP(1);      // input layers
  f32x4 P[NC][2];
  // k-slot e (0..7) of lane group g of the 16x16x32 operands = feature 16 (e >> 2) + 4 g + (e & 3): the accumulator's own feature order, for
  // the weights (gathered so by load_stage1 / the W2 loads) and for the activations alike - h, P and z still never leave the registers
  auto stage1 = [&](int layer) {
    float* sQ = sPQ + (layer & 1) * GNN_Q_BUF;
    f16x8 wah[4], wal[4];
#pragma unroll
    for (int ob4 = 0; ob4 < 4; ob4++) split_f16(wa[ob4], wah[ob4], wal[ob4]);
#pragma unroll
    for (int j = 0; j < NC; j++) {
      f32x4 acc[4];
      float hb[8];
#pragma unroll
      for (int st = 0; st < 8; st++) hb[st] = h[j][st >> 2][st & 3];
      f16x8 bh, bl; split_f16(hb, bh, bl);
#pragma unroll
      for (int ob4 = 0; ob4 < 4; ob4++) acc[ob4] = mfma_split(wah[ob4], wal[ob4], bh, bl, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int mb = 0; mb < 2; mb++) {
#pragma unroll
        for (int i = 0; i < 4; i++) P[j][mb][i] = acc[mb][i] + bias1[mb][i];
        *reinterpret_cast<f32x4*>(sQ + (gnn_node(WAVE, j) * GNN_SAMPLES + n) * GNN_Q_STRIDE + 16 * mb + 4 * g) = acc[2 + mb];
      }
    }
  };
  stage1(0);
  GNN_STAMP(2);
  for (int layer = 0; layer < 3; layer++) {
    const float* L = W + OFF_LAYER0 + layer * LAYER_STRIDE;
    const float* W2 = L + 2080; const float* b2 = L + 3104;
    float wb[2][8], bias2[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; mb++) {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(W2 + (16 * mb + n) * 32 + 16 * q + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; i++) wb[mb][4 * q + i] = w4[i];
      }
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(b2 + 16 * mb + 4 * g);
#pragma unroll
      for (int i = 0; i < 4; i++) bias2[mb][i] = b4[i];
    }
    if (layer < 2) load_stage1(layer + 1);      // wa / bias1 of this layer are dead: P and Q are computed
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();                               // every wavefront has written this layer's Q (and is done reading the previous layer's)
    GNN_STAMP(3);
    {
      // Software-pipelined over the flat edge list: while the 16 MFMAs of edge e issue, the activations z of edge e+1 are computed and
      // edge e-1 is max-aggregated (two accumulator sets), so the matrix pipe does not drain at every edge / node boundary.
      // max_e ELU(y_e + b) = ELU(max_e y_e + b): ELU and the bias add are monotonic, so the activation is applied once per target node.
      const float* sQ = sPQ + (layer & 1) * GNN_Q_BUF;
      constexpr int NE = gnn_ne(WAVE);
      f16x8 wbh[2], wbl[2];
      split_f16(wb[0], wbh[0], wbl[0]); split_f16(wb[1], wbh[1], wbl[1]);
      f32x4 acc[2][2], mx[NC][2];
#pragma unroll
      for (int j = 0; j < NC; j++) mx[j][0] = mx[j][1] = (f32x4){-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
      auto q_of = [&](int e, f32x4& q0, f32x4& q1) {
        const int src = gnn_in(gnn_node(WAVE, gnn_ej(WAVE, e)), gnn_ek(WAVE, e));
        const float* qs = sQ + (src * GNN_SAMPLES + n) * GNN_Q_STRIDE + 4 * g;
        q0 = *reinterpret_cast<const f32x4*>(qs); q1 = *reinterpret_cast<const f32x4*>(qs + 16);
      };
      auto aggregate = [&](int e) {      // edge e is complete in acc[e & 1]
        const int j = gnn_ej(WAVE, e);
#pragma unroll
        for (int i = 0; i < 4; i++) { mx[j][0][i] = fmaxf(mx[j][0][i], acc[e & 1][0][i]); mx[j][1][i] = fmaxf(mx[j][1][i], acc[e & 1][1][i]); }
        if (gnn_ek(WAVE, e) == gnn_nin(gnn_node(WAVE, j)) - 1) {
#pragma unroll
          for (int i = 0; i < 4; i++) { h[j][0][i] = elu(mx[j][0][i] + bias2[0][i]); h[j][1][i] = elu(mx[j][1][i] + bias2[1][i]); }
        }
      };
      float zc[8];
      {
        f32x4 q0, q1; q_of(0, q0, q1);
#pragma unroll
        for (int st = 0; st < 8; st += 2) { const f32x4& pp = P[gnn_ej(WAVE, 0)][st >> 2]; const f32x4& qq = (st >> 2) ? q1 : q0;
          elu_sum2(pp[st & 3], pp[(st & 3) + 1], qq[st & 3], qq[(st & 3) + 1], zc[st], zc[st + 1]); }
      }
#pragma unroll
      for (int e = 0; e < NE; e++) {
        f32x4 q0, q1; float zn[8];
        if (e + 1 < NE) q_of(e + 1, q0, q1);
        f16x8 zh, zl; split_f16(zc, zh, zl);
        // the 8 MFMAs of edge e (two output blocks x four half products) issue while the activations of edge e+1 and the aggregation of edge
        // e-1 run on the vector ALU (an MFMA holds the issue port for 8 of its 16 cycles)
        acc[e & 1][0] = mfma_split(wbh[0], wbl[0], zh, zl, (f32x4){0.f, 0.f, 0.f, 0.f});
        acc[e & 1][1] = mfma_split(wbh[1], wbl[1], zh, zl, (f32x4){0.f, 0.f, 0.f, 0.f});
        if (e + 1 < NE) {
#pragma unroll
          for (int st = 0; st < 8; st += 2) { const f32x4& pp = P[gnn_ej(WAVE, e + 1)][st >> 2]; const f32x4& qq = (st >> 2) ? q1 : q0;
            elu_sum2(pp[st & 3], pp[(st & 3) + 1], qq[st & 3], qq[(st & 3) + 1], zn[st], zn[st + 1]); }
        }
        if (e >= 1) aggregate(e - 1);
        if (e + 1 < NE) {
#pragma unroll
          for (int st = 0; st < 8; st++) zc[st] = zn[st];
        }
      }
      aggregate(NE - 1);
    }
    GNN_STAMP(4);
    if (layer < 2) stage1(layer + 1);            // into the other Q buffer: no barrier between a layer's stage 2 and the next layer's stage 1
    GNN_STAMP(2);
  }
  // ---- heads (:215-241): action mean of joint node j = Linear(32,1)(h[1+j]); value = Linear(32,1)(max over nodes)
  float wact[2][4];
#pragma unroll
  for (int mb = 0; mb < 2; mb++)
#pragma unroll
    for (int i = 0; i < 4; i++) wact[mb][i] = W[OFF_ACT_W + 16 * mb + 4 * g + i];
  const bool write = (g == 0) && (s0 + n < B);
  f32x4 hm0 = h[0][0], hm1 = h[0][1];
#pragma unroll
  for (int j = 0; j < NC; j++) {
    const int nd = gnn_node(WAVE, j);
#pragma unroll
    for (int i = 0; i < 4; i++) { hm0[i] = fmaxf(hm0[i], h[j][0][i]); hm1[i] = fmaxf(hm1[i], h[j][1][i]); }
    if (nd == 0) continue;
    float p = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) { p = fmaf(wact[0][i], h[j][0][i], p); p = fmaf(wact[1][i], h[j][1][i], p); }
    p += __shfl_xor(p, 16); p += __shfl_xor(p, 32);
    const float m = p + W[OFF_ACT_B];
    if (write) {
      if (mean) mean[(size_t)(s0 + n) * 12 + (nd - 1)] = m;
      if (SA.log_std) {
        const int a = nd - 1, smp = s0 + n;
        const float ls = SA.log_std[a], eps = ro_normal(SA.seed, (uint32_t)smp, ro_key(SA.cnt, B, smp), (uint32_t)a);
        SA.actions[(size_t)smp * 12 + a] = fmaf(expf(ls), eps, m);
        sLp[a * GNN_SAMPLES + n] = -0.5f * eps * eps - ls - 0.9189385332046727f;
      }
    }
  }
  // value head: max over all nodes = max over the four wavefronts' partial maxima
#pragma unroll
  for (int i = 0; i < 4; i++) { sHm[((WAVE * 32) + 4 * g + i) * GNN_SAMPLES + n] = hm0[i]; sHm[((WAVE * 32) + 16 + 4 * g + i) * GNN_SAMPLES + n] = hm1[i]; }
  // Persistent rollout (observations in LDS): the sampled actions were stored to global memory by ALL FOUR wavefronts and are read back by
  // wavefront 0's physics step right after this barrier.  lds_barrier() orders LDS traffic only, so the stores are drained first
  // (vmcnt(0): written through to the L2 the block's loads are served from); nothing else is in flight at this point of the tile.
  if (LDS_OBS && SA.log_std) __builtin_amdgcn_s_waitcnt(0x0F70);
  GNN_STAMP(6);      // heads, sampling
  lds_barrier();
  GNN_STAMP(7);
  if (WAVE != 0) return;
  float v = 0.f;
#pragma unroll
  for (int mb = 0; mb < 2; mb++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int f = 16 * mb + 4 * g + i;
      const float m4 = fmaxf(fmaxf(sHm[(0 * 32 + f) * GNN_SAMPLES + n], sHm[(1 * 32 + f) * GNN_SAMPLES + n]),
                             fmaxf(sHm[(2 * 32 + f) * GNN_SAMPLES + n], sHm[(3 * 32 + f) * GNN_SAMPLES + n]));
      v = fmaf(W[OFF_VAL_W + f], m4, v);
    }
  v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
  if (write) {
    value[s0 + n] = v + W[OFF_VAL_B];
    if (SA.log_std) {      // summed in groups of four like k_sample_actions (bit-identical log-probs)
      float part[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 12; a++) part[a >> 2] += sLp[a * GNN_SAMPLES + n];
      SA.logp[s0 + n] = (part[0] + part[1]) + (part[2] + 0.f);
    }
  }
}

struct GnnSmem {
  float sPQ[2 * GNN_Q_BUF] __attribute__((aligned(16)));      // Q of the current / the next layer: [layer parity][node][sample][32 features + pad]
  float sHm[4 * 32 * GNN_SAMPLES];                  // per-wavefront node maxima for the value head
  float sLp[12 * GNN_SAMPLES];                      // per-action log-prob terms
  float sOb[64 * GNN_SAMPLES];                      // the normalised, clipped observation tile [column][sample]
  int rot;                                          // rotation of the wavefront -> node-set map of this block (gnn_block)
};
// forward of samples s0 .. s0+15 by the 256 threads of a block (contains block barriers)
template <bool LDS_OBS>
__device__ __forceinline__ void gnn_block(const float* obs, float obs_clip, int B, int s0, const float* __restrict__ W, float* __restrict__ mean,
                                          float* __restrict__ value, const SampleArgs& SA, GnnSmem& G, int t) {
  const int lane = t & 63;
  // Which node set a wavefront takes.  With two blocks on a compute unit (> 4096 samples), the two wavefronts that share a SIMD are the same
  // wavefront index of the two blocks: the same node set in the same phase, the stage-2-heavy set {hub, ..} twice on one matrix pipe and the
  // stage-1-heavy four-node set twice on another.  The second block of a compute unit (odd wave slot of its first wavefront, read from HW_ID)
  // rotates its sets by two, pairing 112 + 96 and 80 + 96 MFMAs per stage on a SIMD instead of 112 + 112 and 128 + 128.  Any rotation is a
  // valid assignment (the result does not depend on it); the block agrees on one through LDS.
  if (t == 0) G.rot = (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1) * 2;      // hwreg(HW_REG_HW_ID, 0, 4) = WAVE_ID: this wavefront's slot on its SIMD
  {
    // The observation tile, normalised and clipped, once per block: one 16-byte load of the observation and two of the scaler per thread
    // (16 samples x 64 columns = 256 x 4), instead of three scattered 4-byte global loads per value in every wavefront that needs it
    const int sm = t >> 4, c0 = (t & 15) * 4, sample = min(s0 + sm, B - 1);
    float4 o4 = LDS_OBS ? *reinterpret_cast<const float4*>(obs + sm * 64 + c0) : *reinterpret_cast<const float4*>(obs + (size_t)sample * 64 + c0);
    static_assert(OFF_OBS_MEAN % 2 == 0 && OFF_OBS_ISTD % 2 == 0, "the scaler vectors are read as 8-byte pairs");
    const float2 ma = *reinterpret_cast<const float2*>(W + OFF_OBS_MEAN + c0), mb = *reinterpret_cast<const float2*>(W + OFF_OBS_MEAN + c0 + 2);
    const float2 ia = *reinterpret_cast<const float2*>(W + OFF_OBS_ISTD + c0), ib = *reinterpret_cast<const float2*>(W + OFF_OBS_ISTD + c0 + 2);
    const float oclip = W[OFF_OBS_CLIP];
    float o[4] = {o4.x, o4.y, o4.z, o4.w}; const float mu[4] = {ma.x, ma.y, mb.x, mb.y}, is[4] = {ia.x, ia.y, ib.x, ib.y};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (LDS_OBS) o[i] = fminf(fmaxf(o[i], -obs_clip), obs_clip);
      G.sOb[(c0 + i) * GNN_SAMPLES + sm] = fminf(fmaxf((o[i] - mu[i]) * is[i], -oclip), oclip);
    }
  }
  lds_barrier();
  const int wave = ((t >> 6) + G.rot) & 3;
  if (wave == 0) gnn_body<0, LDS_OBS>(obs, obs_clip, B, s0, W, mean, value, SA, G.sPQ, G.sHm, G.sLp, G.sOb, lane);
  else if (wave == 1) gnn_body<1, LDS_OBS>(obs, obs_clip, B, s0, W, mean, value, SA, G.sPQ, G.sHm, G.sLp, G.sOb, lane);
  else if (wave == 2) gnn_body<2, LDS_OBS>(obs, obs_clip, B, s0, W, mean, value, SA, G.sPQ, G.sHm, G.sLp, G.sOb, lane);
  else gnn_body<3, LDS_OBS>(obs, obs_clip, B, s0, W, mean, value, SA, G.sPQ, G.sHm, G.sLp, G.sOb, lane);
}

