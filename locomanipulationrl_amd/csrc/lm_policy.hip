// lm_policy.hip -- policy forward passes on the matrix cores (gfx950): the GNN (below), the MLP, the gaussian action sampling
// fused into their epilogues, and the fused rollout (forward -> sampling -> lm_step, T times, one hipGraph) of include/lm_policy.h.
//
// GNN:
// Restates RobotLearning/omniisaacgymenvs/scripts/graph_model_orebot_ov.py:11-241 (GraphNet hidden 32, 13 nodes,
// 24 directed edges, 3 message-passing layers with max aggregation, Action_Layer / Value_Layer heads) as one kernel:
//   obs (B,64)  ->  action means (B,12) in node order [dof1 a1..a4, dof2 a1..a4, dof3 a1..a4], value (B,1).
//
// Mapping: one block = 16 samples on the four wavefronts of a CU (node ownership, see gnn_body).  Every dense product runs as  D(features x samples) = W(features x K) * X(K x samples)
// on v_mfma_f32_16x16x4_f32 (exact fp32, = an fmaf chain), weights as the A operand, activations as the B operand, so an
// accumulator tile (feature rows in the 4 registers / 4 lane groups, sample on the lane) feeds the next product's B operand
// with no lane movement: only the k order inside the dot product is permuted, and the A operand is gathered in the same
// permuted order.  W1 [h_i || h_j] is split into per-node products P = W1a h + b1 (as target) and Q = W1b h (as source),
// staged once per layer in LDS, so the 24 edge messages need only the 32x32 second linear layer:
//   m_e = ELU(W2 ELU(P[tgt] + Q[src]) + b2),  h'[tgt] = max_e m_e.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/lm_policy.h"
#include "../../include/lm_engine.h"
#include "lm_rng.h"
#include <new>

typedef __attribute__((ext_vector_type(4))) float f32x4;

#define GNN_NODES 13
#define GNN_EDGES 24
#define GNN_H 32
#define GNN_SAMPLES 16

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

// ELU with exp(x) - 1 on the hardware exponential (v_exp_f32): absolute error < 1e-7, against ~30 instructions for expm1f
__device__ __forceinline__ float elu(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }

// edges (source -> target): 0->{1..4}, i->i+4 (1..4), i->i+4 (5..8), then the 12 reverses (graph_model_orebot_ov.py:142-159)
__device__ __forceinline__ constexpr int edge_src(int e) { return e < 4 ? 0 : (e < 8 ? e - 3 : (e < 12 ? e - 3 : (e < 16 ? e - 11 : (e < 20 ? e - 11 : e - 11)))); }
__device__ __forceinline__ constexpr int edge_tgt(int e) { return e < 4 ? e + 1 : (e < 8 ? e + 1 : (e < 12 ? e + 1 : (e < 16 ? 0 : (e < 20 ? e - 15 : e - 15)))); }
// obs column of feature k (0..3) of joint node n (1..12): [0.3 q, 0.3 qd, action, last action] of that joint (:115-126)
__device__ __forceinline__ constexpr int joint_col(int n, int k) { return 16 + 12 * k + (n <= 4 ? n - 1 : (n <= 8 ? 4 + 2 * (n - 5) : 5 + 2 * (n - 9))); }

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

template <int WAVE>
__device__ __forceinline__ void gnn_body(const float* __restrict__ obs, int B, const float* __restrict__ W, float* __restrict__ mean,
                                         float* __restrict__ value, const SampleArgs& SA, float* sPQ, float* sHm, float* sLp, int lane) {
  constexpr int NC = gnn_count(WAVE);
  const int n = lane & 15, g = lane >> 4;
  const int s0 = blockIdx.x * GNN_SAMPLES;
  const int sample = min(s0 + n, B - 1);
  const float* obr = obs + (size_t)sample * 64;
  const float oclip = W[OFF_OBS_CLIP];
  // normalised observation column c of this lane's sample
  auto ob = [&](int c) { float v = (obr[c] - W[OFF_OBS_MEAN + c]) * W[OFF_OBS_ISTD + c]; return fminf(fmaxf(v, -oclip), oclip); };

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
  // ---- three message-passing layers
  for (int layer = 0; layer < 3; layer++) {
    const float* L = W + OFF_LAYER0 + layer * LAYER_STRIDE;
    const float* W1 = L; const float* b1 = L + 2048; const float* W2 = L + 2080; const float* b2 = L + 3104;
    // stage 1: P = W1[:, 0:32] h + b1, Q = W1[:, 32:64] h of the owned nodes -> LDS.
    // output feature block ob4 (0,1 = P rows 0..31; 2,3 = Q rows 0..31); k-step (mb', i) reads h[.][mb'][i] = feature 16 mb' + 4 g + i
    {
      float wa[4][8];
#pragma unroll
      for (int ob4 = 0; ob4 < 4; ob4++)
#pragma unroll
        for (int st = 0; st < 8; st++) wa[ob4][st] = W1[(16 * (ob4 & 1) + n) * 64 + 32 * (ob4 >> 1) + 16 * (st >> 2) + 4 * g + (st & 3)];
      float bias1[2][4];
#pragma unroll
      for (int mb = 0; mb < 2; mb++)
#pragma unroll
        for (int i = 0; i < 4; i++) bias1[mb][i] = b1[16 * mb + 4 * g + i];
#pragma unroll
      for (int j = 0; j < NC; j++) {
        f32x4 acc[4];
#pragma unroll
        for (int ob4 = 0; ob4 < 4; ob4++) acc[ob4] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 8; st++) {
          const float bb = h[j][st >> 2][st & 3];
#pragma unroll
          for (int ob4 = 0; ob4 < 4; ob4++) acc[ob4] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ob4][st], bb, acc[ob4], 0, 0, 0);
        }
#pragma unroll
        for (int ob4 = 0; ob4 < 4; ob4++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            sPQ[(gnn_node(WAVE, j) * 64 + 16 * ob4 + 4 * g + i) * GNN_SAMPLES + n] = acc[ob4][i] + ((ob4 < 2) ? bias1[ob4][i] : 0.f);
      }
    }
    __syncthreads();
    // stage 2: messages along the edges that end in the owned nodes, max-aggregated
    {
      float wb[2][8];
#pragma unroll
      for (int mb = 0; mb < 2; mb++)
#pragma unroll
        for (int st = 0; st < 8; st++) wb[mb][st] = W2[(16 * mb + n) * 32 + 4 * st + g];
      float bias2[2][4];
#pragma unroll
      for (int mb = 0; mb < 2; mb++)
#pragma unroll
        for (int i = 0; i < 4; i++) bias2[mb][i] = b2[16 * mb + 4 * g + i];
#pragma unroll
      for (int j = 0; j < NC; j++) {
        const int tgt = gnn_node(WAVE, j);
        h[j][0] = (f32x4){-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f}; h[j][1] = h[j][0];
#pragma unroll
        for (int k = 0; k < gnn_nin(tgt); k++) {
          const int src = gnn_in(tgt, k);
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int st = 0; st < 8; st++) {
            const int kk = 4 * st + g;
            const float z = elu(sPQ[(tgt * 64 + kk) * GNN_SAMPLES + n] + sPQ[(src * 64 + 32 + kk) * GNN_SAMPLES + n]);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[0][st], z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[1][st], z, acc1, 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < 4; i++) {
            h[j][0][i] = fmaxf(h[j][0][i], elu(acc0[i] + bias2[0][i]));
            h[j][1][i] = fmaxf(h[j][1][i], elu(acc1[i] + bias2[1][i]));
          }
        }
      }
    }
    __syncthreads();       // every wavefront is done reading sPQ before the next layer overwrites it
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
      mean[(size_t)(s0 + n) * 12 + (nd - 1)] = m;
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
  __syncthreads();
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

__global__ void __launch_bounds__(256) k_gnn_forward(const float* __restrict__ obs, int B, const float* __restrict__ W,
                                                     float* __restrict__ mean, float* __restrict__ value, SampleArgs SA) {
  __shared__ float sPQ[GNN_NODES * 64 * GNN_SAMPLES];          // [node][feature 0..63][sample]
  __shared__ float sHm[4 * 32 * GNN_SAMPLES];                  // per-wavefront node maxima for the value head
  __shared__ float sLp[12 * GNN_SAMPLES];                      // per-action log-prob terms
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave == 0) gnn_body<0>(obs, B, W, mean, value, SA, sPQ, sHm, sLp, lane);
  else if (wave == 1) gnn_body<1>(obs, B, W, mean, value, SA, sPQ, sHm, sLp, lane);
  else if (wave == 2) gnn_body<2>(obs, B, W, mean, value, SA, sPQ, sHm, sLp, lane);
  else gnn_body<3>(obs, B, W, mean, value, SA, sPQ, sHm, sLp, lane);
}

// ------------------------------------------------------------------------------------------------
// MLP policy (scripts/skrl_ppo_locomotion.py:30-40): shared trunk 64 -> 256 -> 128 -> 64 (ELU) -> mean (12) + value (1),
// with the observation preprocessor folded in (skrl RunningStandardScaler: clamp((x - mean) / (sqrt(var) + eps), +-clip),
// passed as mean / inverse-std vectors).  Same orientation as the GNN: weights are the MFMA A operand, pre-permuted on the
// host into the per-lane order each v_mfma_f32_16x16x4_f32 consumes (one coalesced 256-byte load per MFMA), activations stay
// in accumulator layout from layer to layer.
//   packed block: obs_mean 64 | obs_inv_std 64 | clip 1 (+3 pad) | W1p 256x64 | b1 256 | W2p 128x256 | b2 128 | W3p 64x128 | b3 64 |
//                 Whp 16x64 (rows 0..11 mean, 12 value, 13..15 zero) | bh 16
// offsets as functions of the observation width NOBS (64: velocity-drive / position-control tasks, 88: custom-controller tasks)
__host__ __device__ constexpr int mlp_off_mean(int) { return 0; }
__host__ __device__ constexpr int mlp_off_istd(int nobs) { return nobs; }
__host__ __device__ constexpr int mlp_off_clip(int nobs) { return 2 * nobs; }
__host__ __device__ constexpr int mlp_off_w1(int nobs) { return 2 * nobs + 4; }
__host__ __device__ constexpr int mlp_off_b1(int nobs) { return mlp_off_w1(nobs) + 256 * nobs; }
__host__ __device__ constexpr int mlp_off_w2(int nobs) { return mlp_off_b1(nobs) + 256; }
__host__ __device__ constexpr int mlp_off_b2(int nobs) { return mlp_off_w2(nobs) + 128 * 256; }
__host__ __device__ constexpr int mlp_off_w3(int nobs) { return mlp_off_b2(nobs) + 128; }
__host__ __device__ constexpr int mlp_off_b3(int nobs) { return mlp_off_w3(nobs) + 64 * 128; }
__host__ __device__ constexpr int mlp_off_wh(int nobs) { return mlp_off_b3(nobs) + 64; }
__host__ __device__ constexpr int mlp_off_bh(int nobs) { return mlp_off_wh(nobs) + 16 * 64; }
__host__ __device__ constexpr int mlp_params(int nobs) { return mlp_off_bh(nobs) + 16; }

// One block = 16 samples on the 4 wavefronts (= 4 SIMDs) of a CU: every layer's output blocks are dealt round-robin to the
// wavefronts, activations pass from layer to layer through LDS as [feature][sample] (row stride 20 floats: the B-operand reads of the
// four lane groups then fall on disjoint banks).  Weights are the MFMA A operand, pre-permuted on the host (policies/mlp_model.py).
#define MLP_LDS_STRIDE 20
// k-row of the B operand for k-step `st`, lane group g: layer 1 reads the observation in natural order, the later layers in the order
// the host permutation assumes (16 (st >> 2) + 4 g + (st & 3), i.e. "the previous layer's accumulator tile")
template <bool NATURAL> __device__ __forceinline__ int mlp_krow(int st, int g) { return NATURAL ? 4 * st + g : 16 * (st >> 2) + 4 * g + (st & 3); }

// The k-steps of all output blocks a wavefront owns form one sequence, processed in chunks of 16 with the next chunk's weights
// (one coalesced 256-byte load per k-step) already in flight while the current chunk's MFMAs run: the loop is bound by the matrix
// pipe, not by the L2 round trip of each chunk.
template <int OUT_BLOCKS, int IN_STEPS, bool NATURAL>
__device__ __forceinline__ void mlp_layer4(const float* __restrict__ Wp, const float* __restrict__ bias, const float* sIn, float* sOut,
                                           int wave, int lane, int n, int g, bool act) {
  constexpr int OWNED = (OUT_BLOCKS + 3) / 4, TOTAL = OWNED * IN_STEPS, CH = 16, NCH = (TOTAL + CH - 1) / CH;
  if (wave >= OUT_BLOCKS) return;
  float abuf[2][CH];
  auto issue = [&](int c, float* dst) {
#pragma unroll
    for (int i = 0; i < CH; i++) {
      const int s = c * CH + i;
      if (s < TOTAL) { const int j = s / IN_STEPS, st = s - j * IN_STEPS, mb = wave + 4 * j; dst[i] = (mb < OUT_BLOCKS) ? Wp[((size_t)mb * IN_STEPS + st) * 64 + lane] : 0.f; }
    }
  };
  issue(0, abuf[0]);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    if (c + 1 < NCH) issue(c + 1, abuf[(c + 1) & 1]);
#pragma unroll
    for (int i = 0; i < CH; i++) {
      const int s = c * CH + i;
      if (s < TOTAL) {
        const int j = s / IN_STEPS, st = s - j * IN_STEPS, mb = wave + 4 * j;
        if (mb < OUT_BLOCKS) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[c & 1][i], sIn[mlp_krow<NATURAL>(st, g) * MLP_LDS_STRIDE + n], acc, 0, 0, 0);
          if (st == IN_STEPS - 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) { float v = acc[k] + bias[16 * mb + 4 * g + k]; sOut[(16 * mb + 4 * g + k) * MLP_LDS_STRIDE + n] = act ? elu(v) : v; }
            acc = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    }
  }
}

template <int NOBS>
__global__ void __launch_bounds__(256) k_mlp_forward(const float* __restrict__ obs, int B, const float* __restrict__ W,
                                                     float* __restrict__ mean, float* __restrict__ value, SampleArgs SA) {
  __shared__ float sX[NOBS * MLP_LDS_STRIDE], sH1[256 * MLP_LDS_STRIDE], sH2[128 * MLP_LDS_STRIDE], sH3[64 * MLP_LDS_STRIDE], sO[16 * MLP_LDS_STRIDE];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, n = lane & 15, g = lane >> 4;
  const int s0 = blockIdx.x * 16;
  // normalised, clipped observation tile: one 16-byte load per (sample, 4 features)
  for (int idx = t; idx < 16 * (NOBS / 4); idx += 256) {
    const int sm = idx / (NOBS / 4), c0 = (idx - sm * (NOBS / 4)) * 4, sample = min(s0 + sm, B - 1);
    const float4 o4 = *reinterpret_cast<const float4*>(obs + (size_t)sample * NOBS + c0);
    const float clip = W[mlp_off_clip(NOBS)], o[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float v = (o[i] - W[mlp_off_mean(NOBS) + c0 + i]) * W[mlp_off_istd(NOBS) + c0 + i];
      sX[(c0 + i) * MLP_LDS_STRIDE + sm] = fminf(fmaxf(v, -clip), clip);
    }
  }
  __syncthreads();
  mlp_layer4<16, NOBS / 4, true>(W + mlp_off_w1(NOBS), W + mlp_off_b1(NOBS), sX, sH1, wave, lane, n, g, true);
  __syncthreads();
  mlp_layer4<8, 64, false>(W + mlp_off_w2(NOBS), W + mlp_off_b2(NOBS), sH1, sH2, wave, lane, n, g, true);
  __syncthreads();
  mlp_layer4<4, 32, false>(W + mlp_off_w3(NOBS), W + mlp_off_b3(NOBS), sH2, sH3, wave, lane, n, g, true);
  __syncthreads();
  if (wave != 0) return;
  mlp_layer4<1, 16, false>(W + mlp_off_wh(NOBS), W + mlp_off_bh(NOBS), sH3, sO, 0, lane, n, g, false);
  __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
  const int smp = s0 + n;
  const bool valid = smp < B;
  float lp = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int j = 4 * g + i;                      // head output row: 0..11 action means, 12 value
    const float v = sO[j * MLP_LDS_STRIDE + n];
    if (valid && j < 12) mean[(size_t)smp * 12 + j] = v;
    if (valid && j == 12) value[smp] = v;
    if (SA.log_std && j < 12 && valid) {
      const float ls = SA.log_std[j], eps = ro_normal(SA.seed, (uint32_t)smp, ro_key(SA.cnt, B, smp), (uint32_t)j);
      SA.actions[(size_t)smp * 12 + j] = fmaf(expf(ls), eps, v);
      lp += -0.5f * eps * eps - ls - 0.9189385332046727f;          // log N(a; mean, std) with (a - mean) / std = eps
    }
  }
  if (SA.log_std) { lp += __shfl_xor(lp, 16); lp += __shfl_xor(lp, 32); if (valid && g == 0) SA.logp[smp] = lp; }
}

extern "C" {

static int mlp_launch(const float* obs, int batch, int num_obs, const float* params, float* mean, float* value, const SampleArgs& SA, hipStream_t s) {
  if (num_obs == 64) hipLaunchKernelGGL(k_mlp_forward<64>, dim3((batch + 15) / 16), dim3(256), 0, s, obs, batch, params, mean, value, SA);
  else if (num_obs == 88) hipLaunchKernelGGL(k_mlp_forward<88>, dim3((batch + 15) / 16), dim3(256), 0, s, obs, batch, params, mean, value, SA);
  else return -1;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int lm_mlp_param_count(void) { return mlp_params(64); }
int lm_mlp_param_count_obs(int num_obs) { return (num_obs == 64 || num_obs == 88) ? mlp_params(num_obs) : -1; }
int lm_mlp_forward_obs(const float* obs, int batch, int num_obs, const float* params, float* mean, float* value, void* stream) {
  if (!obs || !params || !mean || !value || batch <= 0) return -1;
  SampleArgs SA{}; return mlp_launch(obs, batch, num_obs, params, mean, value, SA, (hipStream_t)stream);
}

int lm_mlp_forward(const float* obs, int batch, const float* params, float* mean, float* value, void* stream) {
  return lm_mlp_forward_obs(obs, batch, 64, params, mean, value, stream);
}

int lm_gnn_param_count(void) { return GNN_PARAMS; }

int lm_gnn_forward(const float* obs, int batch, const float* params, float* mean, float* value, void* stream) {
  if (!obs || !params || !mean || !value || batch <= 0) return -1;
  int blocks = (batch + GNN_SAMPLES - 1) / GNN_SAMPLES;
  SampleArgs SA{}; hipLaunchKernelGGL(k_gnn_forward, dim3(blocks), dim3(256), 0, (hipStream_t)stream, obs, batch, params, mean, value, SA);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}


// ------------------------------------------------------------------------------------------------ fused rollout (f-2)
__global__ void __launch_bounds__(256) k_sample_actions(const float* __restrict__ mean, const float* __restrict__ log_std, const int64_t* __restrict__ cnt,
                                                        int N, uint32_t seed, float* __restrict__ actions, float* __restrict__ logp) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= N) return;
  // (episode_count, progress_buf) identifies the env-step: progress restarts at every reset and the episode count moves on
  const uint32_t key = ro_key(cnt, N, env);
  float part[3] = {0.f, 0.f, 0.f};      // summed in groups of four like the fused epilogue of k_mlp_forward (bit-identical log-probs)
#pragma unroll
  for (int j = 0; j < 12; j++) {
    const float ls = log_std[j], eps = ro_normal(seed, (uint32_t)env, key, (uint32_t)j);
    actions[(size_t)env * 12 + j] = fmaf(expf(ls), eps, mean[(size_t)env * 12 + j]);
    part[j >> 2] += -0.5f * eps * eps - ls - 0.9189385332046727f;          // log N(a; mean, std) with (a - mean) / std = eps
  }
  const float lp = (part[0] + part[1]) + (part[2] + 0.f);
  logp[env] = lp;
}

struct lm_rollout {
  lm_engine* env; int policy, T, N, nobs; uint32_t seed;
  const float *params, *log_std; float *obs, *actions, *logp, *values, *rewards, *extras; int64_t* dones;
  float* mean_tmp; const int64_t* cnt;
  hipGraphExec_t exec; hipStream_t exec_stream;
};

static int rollout_enqueue(lm_rollout* r, hipStream_t s) {
  const size_t N = (size_t)r->N;
  for (int t = 0; t <= r->T; t++) {
    const float* ob = r->obs + (size_t)t * N * r->nobs;
    float* act = r->actions + (size_t)(t < r->T ? t : 0) * N * 12;
    int rc = 0;
    SampleArgs SA{};       // sampling fused into the forward's epilogue
    if (t < r->T) { SA.log_std = r->log_std; SA.cnt = r->cnt; SA.seed = r->seed; SA.actions = act; SA.logp = r->logp + (size_t)t * N; }
    if (r->policy == LM_POLICY_MLP) { int rcm = mlp_launch(ob, r->N, r->nobs, r->params, r->mean_tmp, r->values + (size_t)t * N, SA, s); if (rcm) return rcm; }
    else
      hipLaunchKernelGGL(k_gnn_forward, dim3((r->N + GNN_SAMPLES - 1) / GNN_SAMPLES), dim3(256), 0, s, ob, r->N, r->params, r->mean_tmp, r->values + (size_t)t * N, SA);
    if (rc) return rc;
    if (t == r->T) break;                                        // the last forward only bootstraps the value
    rc = lm_step(r->env, act, nullptr, r->obs + (size_t)(t + 1) * N * r->nobs, nullptr, r->rewards + (size_t)t * N, r->dones + (size_t)t * N,
                 r->extras ? r->extras + (size_t)t * LM_NUM_EXTRAS : nullptr, s);
    if (rc) return rc;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int lm_sample_actions(const float* mean, const float* log_std, const int64_t* cnt, int n_envs, uint32_t seed, float* actions, float* logp, void* stream) {
  if (!mean || !log_std || !cnt || !actions || !logp || n_envs <= 0) return -1;
  hipLaunchKernelGGL(k_sample_actions, dim3((n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, log_std, cnt, n_envs, seed, actions, logp);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int lm_rollout_create(lm_rollout** out, lm_engine* env, int policy, const float* policy_params, const float* log_std, int T, uint32_t noise_seed,
                      float* obs, float* actions, float* logp, float* values, float* rewards, int64_t* dones, float* extras) {
  if (!out || !env || !policy_params || !log_std || !obs || !actions || !logp || !values || !rewards || !dones || T <= 0) return -1;
  if (policy != LM_POLICY_MLP && policy != LM_POLICY_GNN) return -1;
  const int nobs = lm_num_obs(env);
  if (nobs != 64 && !(nobs == 88 && policy == LM_POLICY_MLP)) return -1;      // the GNN reads the 64-wide layout; the MLP also the 88-wide one
  lm_rollout* r = new (std::nothrow) lm_rollout();
  if (!r) return -3;
  r->env = env; r->policy = policy; r->T = T; r->N = lm_num_envs(env); r->nobs = nobs; r->seed = noise_seed;
  r->params = policy_params; r->log_std = log_std; r->obs = obs; r->actions = actions; r->logp = logp; r->values = values;
  r->rewards = rewards; r->dones = dones; r->extras = extras; r->exec = nullptr; r->exec_stream = nullptr;
  r->cnt = (const int64_t*)lm_ptr(env, LM_PTR_CNT);
  if (hipMalloc((void**)&r->mean_tmp, (size_t)r->N * 12 * sizeof(float)) != hipSuccess) { delete r; return -2; }
  *out = r;
  return 0;
}

int lm_rollout_run(lm_rollout* r, int use_graph, void* stream) {
  if (!r) return -1;
  hipStream_t s = (hipStream_t)stream;
  if (!use_graph) return rollout_enqueue(r, s);
  if (!r->exec) {
    // capture the 4T+1 launches once; every pointer in them is fixed for the lifetime of the plan
    hipGraph_t g = nullptr;
    hipStream_t cs = s;
    bool own = false;
    if (cs == nullptr) { if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) return -2; own = true; }   // the legacy stream cannot capture
    if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) { if (own) (void)hipStreamDestroy(cs); return -2; }
    int rc = rollout_enqueue(r, cs);
    hipError_t e = hipStreamEndCapture(cs, &g);
    if (own) (void)hipStreamDestroy(cs);
    if (rc || e != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); return rc ? rc : -2; }
    e = hipGraphInstantiate(&r->exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { r->exec = nullptr; return -2; }
  }
  return hipGraphLaunch(r->exec, s) == hipSuccess ? 0 : -2;
}

int lm_rollout_destroy(lm_rollout* r) {
  if (!r) return 0;
  if (r->exec) (void)hipGraphExecDestroy(r->exec);
  if (r->mean_tmp) (void)hipFree(r->mean_tmp);
  delete r;
  return 0;
}

}  // extern "C"
