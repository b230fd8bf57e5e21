// lm_engine.hip -- MI355X (gfx950) physics-step engine: kernels + C ABI (include/lm_engine.h).
//
// Mapping: one 64-lane wavefront = 16 environments x 4 limbs ("limb per lane").  Each lane runs the
// limb-aggregate articulated-body sweep for its overconstrained module (5 bodies, one embedded
// Bennett loop, 3 independent joints); the four lanes of an env meet at the hub through DPP
// quad-permute reductions (backward sweep: articulated inertia + bias onto the hub; forward sweep:
// hub acceleration back to the limbs).  State is SoA [row][N] in HBM, the robot table sits in LDS,
// per-task constants are read through scalar loads.  DESIGN.md sections 3-5 derive every formula.
//
// Reference rows replaced: a4-a6 (reset scatter + action scaling), a7 (PhysX world.step x
// controlFrequencyInv), a8 (state read-back), a9-a11 (obs / reward / termination), a13 (plate deltas), a14 (co-train:
// two parameter blocks in one launch); 8f-1 (PD-actuator task families = template parameter VAR of the step), 8f-3 (domain
// randomisation = template parameter DR, kernel k_step_dr); reductions over envs are fused into the step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <new>
#include "lm_math.h"
#include "lm_rng.h"
#include "../../include/lm_engine.h"
#include "../../include/lm_policy.h"
// Diagnostic build only (-DLM_STAMPS, tools/stamp_profile.sh): LM_STAMP(k) adds the shader cycles since the previous stamp to bucket k of
// a per-workgroup LDS array that k_step copies to lm_stamp_out.  No stamp exists in the product build.
#ifdef LM_STAMPS
__device__ unsigned long long lm_stamp_out[1024 * 64];
__shared__ unsigned long long lm_stamp_lds[64];      // 0..15: wavefront 0 (15 = its last stamp time); 16 (w - 1) + 16 ..: policy wavefront w of k_rollout_mlp
#define LM_PSTAMP(w, k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    if ((threadIdx.x & 63) == 0) { lm_stamp_lds[16 * (w) + (k)] += t_ - lm_stamp_lds[16 * (w) + 15]; lm_stamp_lds[16 * (w) + 15] = t_; } } while (0)
#if LM_STAMPS == 2      // only the wavefront's lifetime (two clock reads around the whole body): what the product code takes inside the kernel
#define LM_STAMP(k) do { } while (0)
#else
#define LM_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    if (threadIdx.x == 0) { lm_stamp_lds[k] += t_ - lm_stamp_lds[15]; lm_stamp_lds[15] = t_; } } while (0)
#endif
#define MLP_RES_STAMP(P, k) LM_PSTAMP((P) + 1, k)
#else
#define LM_STAMP(k) do { } while (0)
#endif
#include "lm_policy_dev.h"
#include "lm_internal.h"

#define HUB_FLOATS 10
#define ENVS_PER_WAVE 16
// Device-side table layout (built from the packed public table by permute_table() in lm_create).  A limb is the shell plus two
// structurally identical two-link chains, A = link4 -> link3 (joints dof2, p1) and B = link1 -> link2 (joints dof3, p2); the entries
// of the chains are interleaved (A, B) so that one 8-byte LDS read feeds one packed-fp32 operand (lm_math.h).
#define PUB_LIMB_STRIDE 123   // public: 5 joints x 13, 5 inertias x 10, tip (3), foot-sphere centre (3), foot body flag, pad
#define LIMB_STRIDE 124       // device: even, every pair 8-byte aligned
#define T_J0 0                // shell joint: R (9, row-major), p (3), axis sign
#define T_P1 14               // 13 pairs: first joints of the chains (dof2 | dof3)
#define T_P2 40               // 13 pairs: second joints (p1 | p2)
#define T_I0 66               // shell inertia (m, com 3, I 6)
#define T_Q1 76               // 10 pairs: link4 | link1
#define T_Q2 96               // 10 pairs: link3 | link2
#define T_TIP 116             // fingertip frame on link3
#define T_FOOT 119            // foot-sphere centre in its body's frame
#define T_FLAG 122            // 0: the foot rides on link3, 1: on link2
#define LM_ITAB_FLOATS (HUB_FLOATS + 4 * LIMB_STRIDE)
static void permute_table(const float* pub, float* dev) {
  memset(dev, 0, LM_ITAB_FLOATS * sizeof(float));
  {   // hub body: (m, com, I about COM) -> spatial inertia about the hub origin (m, h = m c, I_O [xx,yy,zz,xy,xz,yz]), constant in hub coordinates
    const double m = pub[0], c[3] = {pub[1], pub[2], pub[3]}, cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    dev[0] = (float)m; dev[1] = (float)(m * c[0]); dev[2] = (float)(m * c[1]); dev[3] = (float)(m * c[2]);
    dev[4] = (float)(pub[4] + m * (cc - c[0] * c[0])); dev[5] = (float)(pub[5] + m * (cc - c[1] * c[1])); dev[6] = (float)(pub[6] + m * (cc - c[2] * c[2]));
    dev[7] = (float)(pub[7] - m * c[0] * c[1]); dev[8] = (float)(pub[8] - m * c[0] * c[2]); dev[9] = (float)(pub[9] - m * c[1] * c[2]);
  }
  for (int l = 0; l < 4; l++) {
    const float* s = pub + HUB_FLOATS + l * PUB_LIMB_STRIDE; float* d = dev + HUB_FLOATS + l * LIMB_STRIDE;
    for (int k = 0; k < 13; k++) { d[T_J0 + k] = s[k]; d[T_P1 + 2 * k] = s[13 + k]; d[T_P1 + 2 * k + 1] = s[39 + k]; d[T_P2 + 2 * k] = s[26 + k]; d[T_P2 + 2 * k + 1] = s[52 + k]; }
    for (int k = 0; k < 10; k++) { d[T_I0 + k] = s[65 + k]; d[T_Q1 + 2 * k] = s[75 + k]; d[T_Q1 + 2 * k + 1] = s[95 + k]; d[T_Q2 + 2 * k] = s[85 + k]; d[T_Q2 + 2 * k + 1] = s[105 + k]; }
    for (int k = 0; k < 7; k++) d[T_TIP + k] = s[115 + k];
  }
}

// state rows
#define R_FB0 0            // base: pos 0..2 quat 3..6 lin 7..9 ang 10..12
#define R_Q 13
#define R_QD 25
#define R_FB1 37           // plate: pos, quat, lin, ang
#define R_LACT 50
#define R_LQD 62
#define R_LTIP 74
#define R_GOAL 86
#define R_SE 90           // swing/extension position targets (custom-controller tasks)
#define R_LTGT 102        // last joint position targets
#define R_LRD 114         // last rot_dist

#define SQRT2F 1.41421356237f

static thread_local char g_err[256] = "";
static int fail(int code, const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(g_err, sizeof(g_err), "%s: %s", #x, hipGetErrorString(e_)); return LM_EHIP; } } while (0)

struct lm_engine {
  int N, n_tasks, split, nblocks, num_obs, device;
  int w2_min_envs;         // lm_step launches k_step_w2 (two wavefronts per SIMD) from this env count on; LM_W2_MIN_ENVS overrides 32769 (tests, A/B)
  uint32_t seed;
  lm_params* d_params;     // [2]
  float* d_table;
  float* d_state; int64_t* d_cnt; int64_t* d_drc; float* d_dr_phys; int dr_enabled;
  float *d_obs, *d_states, *d_rew, *d_extras, *d_terms; long long* d_acc; int acc_rows;
  bool view_obs, view_states, view_terms;      // lm_ptr() handed out obs_buf / states_buf / the reward terms: lm_step keeps them current from then on
  char* d_stats;           // int64 {num_successes, num_resets} x {all, first task, second task}; float success_rate x 3 at byte 48;
                           // uint32 count of contained blow-ups at byte 60
  lm_params h_params[2];
};

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
LM_DEV uint32_t mix32(uint32_t x) { return lm_mix32(x); }
LM_DEV void hash_uniform3(uint32_t seed, uint32_t env, uint32_t episode, float* u) {
  uint32_t base = mix32(seed ^ mix32(env * 0x9E3779B9U + 0x7F4A7C15U) ^ mix32(episode * 0x85EBCA6BU + 0x165667B1U));
#pragma unroll
  for (uint32_t k = 0; k < 3; k++) { uint32_t r = mix32(base + (k + 1U) * 0xC2B2AE35U); u[k] = (float)(r >> 8) * (1.0f / 16777216.0f); }
}
// sin/cos for |x| up to a few turns (joint angles are bounded by +-pi): Cody-Waite reduction to [-pi/4, pi/4] and
// the single-precision minimax polynomials of Cephes sinf/cosf; absolute error ~1e-7, ~25 instructions
// (the libm sincosf carries a large-argument Payne-Hanek path that costs ~150).
LM_DEV void lm_sincos(float x, float* s, float* c) {
  float kf = rintf(x * 0.636619772367581343f);          // 2/pi
  int k = (int)kf;
  float r = fmaf(-kf, 1.57079625129699707031f, x);       // pi/2 split in three parts
  r = fmaf(-kf, 7.54978941586159635335e-8f, r);
  r = fmaf(-kf, 5.39030285815811905290e-15f, r);
  float z = r * r;
  float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
  float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
  float ss = (k & 1) ? cp : sp, cc = (k & 1) ? sp : cp;
  *s = (k & 2) ? -ss : ss;
  *c = ((k + 1) & 2) ? -cc : cc;
}

// ---- domain randomisation: counter-based samples, same bits as oracle lmo_dr_sample up to fp32 rounding of log / cos
LM_DEV float dr_sample(uint32_t seed, uint32_t stream, uint32_t env, uint32_t key, uint32_t idx, int dist, float p0, float p1) {
  // components 2p and 2p+1 share one pair of uniforms (Box-Muller's cosine and sine branches): calls for neighbouring components
  // share the hashes, the logarithm, the square root and the sincos after common-subexpression elimination
  const uint32_t pair = idx >> 1; const bool odd = (idx & 1U) != 0;
  float u1, u2; lm_rng_pair(lm_rng_base(seed, stream, env, key), pair, &u1, &u2);
  if (dist == LM_DR_GAUSSIAN) { float sn, cs; lm_sincos(6.283185307179586f * u2, &sn, &cs); return p0 + p1 * (sqrtf(-2.0f * logf(u1)) * (odd ? sn : cs)); }
  const float u = odd ? u1 - (1.0f / 16777216.0f) : u2;
  if (dist == LM_DR_UNIFORM) return p0 + (p1 - p0) * u;
  return expf(logf(p0) + (logf(p1) - logf(p0)) * u);
}
LM_DEV float dr_apply(int op, float x, float n) { return op == LM_DR_ADDITIVE ? x + n : (op == LM_DR_SCALING ? x * n : n); }
// one randomised physics attribute: on_interval entries are redrawn every `interval` control steps, on_reset entries at the env's
// last gated reset (reset_key 0 = never randomised)
LM_DEV float dr_attr(const lm_dr_channel& ch, uint32_t seed, uint32_t stream, int env, uint32_t dr_step, uint32_t reset_key, int idx, int comp, float base) {
  if (!ch.enabled) return base;
  uint32_t key = ch.interval > 0 ? dr_step / (uint32_t)ch.interval : reset_key;
  if (ch.interval == 0 && key == 0) return base;
  return dr_apply(ch.operation, base, dr_sample(seed, stream, (uint32_t)env, key, (uint32_t)idx, ch.distribution, ch.p0[comp], ch.p1[comp]));
}
struct DrPhys { float tmax[3], vmax[3], cj[3]; V3 g, f; };      // this lane's three joints; gravity (world); base-link force (world)
LM_DEV Q4 quat_from_euler(float roll, float pitch, float yaw) {
  float sy, cy, sr, cr, sp, cp;
  sincosf(yaw * 0.5f, &sy, &cy); sincosf(roll * 0.5f, &sr, &cr); sincosf(pitch * 0.5f, &sp, &cp);
  Q4 q; q.w = cy * cr * cp + sy * sr * sp; q.x = cy * sr * cp - sy * cr * sp; q.y = cy * cr * sp + sy * sr * cp; q.z = sy * cr * cp - cy * sr * sp;
  return q;
}
LM_DEV M3 load_m3_rowmajor(const float* t) {
  M3 R; R.c0 = v3(t[0], t[3], t[6]); R.c1 = v3(t[1], t[4], t[7]); R.c2 = v3(t[2], t[5], t[8]); return R;
}
LM_DEV SI load_si(const float* t) { SI I; I.m = t[0]; I.h = v3(t[1], t[2], t[3]); I.xx = t[4]; I.yy = t[5]; I.zz = t[6]; I.xy = t[7]; I.xz = t[8]; I.yz = t[9]; return I; }
// hub body: the device table holds its spatial inertia about the hub origin (permute_table)
LM_DEV SI hub_inertia(const float* t) { return load_si(t); }
LM_DEV void si_to_66(const SI& I, float A[6][6]) {
  A[0][0] = I.xx; A[0][1] = I.xy; A[0][2] = I.xz; A[1][1] = I.yy; A[1][2] = I.yz; A[2][2] = I.zz;
  A[1][0] = I.xy; A[2][0] = I.xz; A[2][1] = I.yz;
  // M_wv = [h]x, M_vw = -[h]x
  A[0][3] = 0.f;     A[0][4] = -I.h.z; A[0][5] = I.h.y;
  A[1][3] = I.h.z;   A[1][4] = 0.f;    A[1][5] = -I.h.x;
  A[2][3] = -I.h.y;  A[2][4] = I.h.x;  A[2][5] = 0.f;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) { A[3 + i][j] = A[j][3 + i]; A[3 + i][3 + j] = (i == j) ? I.m : 0.f; }
}
LM_DEV void sv_to_arr(SV a, float* o) { o[0] = a.w.x; o[1] = a.w.y; o[2] = a.w.z; o[3] = a.v.x; o[4] = a.v.y; o[5] = a.v.z; }
LM_DEV SV arr_to_sv(const float* o) { return sv(v3(o[0], o[1], o[2]), v3(o[3], o[4], o[5])); }

// one joint of the limb tree: parent pose (Rp, op) + table entry (R 9 row-major, p 3, sign) + rotation (c, s); T = float for the shell
// joint, f2 for a pair of chain joints (table entries interleaved)
template <class T> LM_DEV M3T<T> load_m3_rowmajor_t(const T* t) {
  M3T<T> R; R.c0 = v3t<T>(t[0], t[3], t[6]); R.c1 = v3t<T>(t[1], t[4], t[7]); R.c2 = v3t<T>(t[2], t[5], t[8]); return R;
}
template <class T> LM_DEV void joint_frame(const M3T<T>& Rp, V3T<T> op, const T* tj, T c, T s, M3T<T>& R, V3T<T>& o, V3T<T>& z) {
  M3T<T> A = mul(Rp, load_m3_rowmajor_t<T>(tj));
  T sg = tj[12]; T ss = sg * s;
  R.c0 = fma3(c, A.c0, ss * A.c1);
  R.c1 = fma3(c, A.c1, (-ss) * A.c0);
  R.c2 = A.c2;
  o = op + mul(Rp, v3t<T>(tj[9], tj[10], tj[11]));
  z = sg * A.c2;
}
// the shell joint hangs on the hub frame itself (identity parent): no parent products
LM_DEV void joint_frame_root(const float* tj, float c, float s, M3& R, V3& o, V3& z) {
  M3 A = load_m3_rowmajor_t<float>(tj);
  float sg = tj[12]; float ss = sg * s;
  R.c0 = fma3(c, A.c0, ss * A.c1);
  R.c1 = fma3(c, A.c1, (-ss) * A.c0);
  R.c2 = A.c2;
  o = v3(tj[9], tj[10], tj[11]);
  z = sg * A.c2;
}
// sin/cos of a pair of angles (same polynomial as lm_sincos, both halves in packed instructions where the ISA has them)
LM_DEV void lm_sincos2(f2 x, f2* s, f2* c) {
  const f2 kf = mk2(rintf(x.x * 0.636619772367581343f), rintf(x.y * 0.636619772367581343f));
  const int k0 = (int)kf.x, k1 = (int)kf.y;
  f2 r = fma_(-kf, sp2(1.57079625129699707031f), x);
  r = fma_(-kf, sp2(7.54978941586159635335e-8f), r);
  r = fma_(-kf, sp2(5.39030285815811905290e-15f), r);
  const f2 z = r * r;
  const f2 sp = fma_(fma_(fma_(sp2(-1.9515295891e-4f), z, sp2(8.3321608736e-3f)), z, sp2(-1.6666654611e-1f)), z * r, r);
  const f2 cp = fma_(fma_(fma_(sp2(2.443315711809948e-5f), z, sp2(-1.388731625493765e-3f)), z, sp2(4.166664568298827e-2f)), z * z, fma_(sp2(-0.5f), z, sp2(1.0f)));
  const float ss0 = (k0 & 1) ? cp.x : sp.x, cc0 = (k0 & 1) ? sp.x : cp.x, ss1 = (k1 & 1) ? cp.y : sp.y, cc1 = (k1 & 1) ? sp.y : cp.y;
  *s = mk2((k0 & 2) ? -ss0 : ss0, (k1 & 2) ? -ss1 : ss1);
  *c = mk2(((k0 + 1) & 2) ? -cc0 : cc0, ((k1 + 1) & 2) ? -cc1 : cc1);
}

// kinematics + dynamics terms of one limb, all in hub ("base") coordinates about the hub origin.  Pairs hold (chain A | chain B) =
// (link4 | link1) at the first level and (link3 | link2) at the second.
struct LimbKin {
  M3 Rs; V3 os; SV s1;                     // shell
  M3P R41, R32; V3P o41, o32;              // frames: (link4 | link1), (link3 | link2); tip on link3, knees = origins of link3 and link2
  SVP s23, sp12;                           // joint axes (dof2 | dof3), (p1 | p2)
  float g1, pd, pdd;                       // closure: dp/dD, passive rate, passive vp-acceleration
};

LM_DEV void limb_kinematics(const float* tl, const float q[3], const float qd[3], LimbKin& K) {
  float s1_, c1_; f2 s23_, c23_;
  lm_sincos(q[0], &s1_, &c1_); lm_sincos2(mk2(q[1], q[2]), &s23_, &c23_);
  const float s2_ = s23_.x, c2_ = c23_.x, s3_ = s23_.y, c3_ = c23_.y;
  float cD = c2_ * c3_ + s2_ * s3_, sD = s2_ * c3_ - c2_ * s3_;      // D = q2 - q3
  float inv = 1.0f / (3.0f - cD);
  float cp = (3.0f * cD - 1.0f) * inv, sp = 2.0f * SQRT2F * sD * inv;
  K.g1 = 2.0f * SQRT2F * inv;
  float g2 = -2.0f * SQRT2F * sD * inv * inv;
  float dd = qd[1] - qd[2];
  K.pd = K.g1 * dd; K.pdd = g2 * dd * dd;
  const f2* tp = reinterpret_cast<const f2*>(tl);
  V3 z1; V3P z23, zp12;
  joint_frame_root(tl + T_J0, c1_, s1_, K.Rs, K.os, z1);
  joint_frame<f2>(bc(K.Rs), bc(K.os), tp + T_P1 / 2, c23_, s23_, K.R41, K.o41, z23);          // dof2 | dof3
  joint_frame<f2>(K.R41, K.o41, tp + T_P2 / 2, sp2(cp), mk2(sp, -sp), K.R32, K.o32, zp12);    // p1 = +g(D) | p2 = -g(D)
  K.s1 = axis_sv(z1, K.os); K.s23 = axis_sv(z23, K.o41); K.sp12 = axis_sv(zp12, K.o32);
}
// the fingertip frame (on link3) and the two knee origins of a limb, hub coordinates
LM_DEV void limb_points(const float* tl, const LimbKin& K, V3& tip, V3& knee2, V3& knee3) {
  knee3 = lo(K.o32); knee2 = hi(K.o32);
  tip = knee3 + mul(lo(K.R32), v3(tl[T_TIP], tl[T_TIP + 1], tl[T_TIP + 2]));
}

struct LimbDyn {
  SV Fq0, Fq1, Fq2;       // coupling columns (6x3): hub wrench per unit limb acceleration
  float H[6];             // limb joint-space inertia, packed [00,01,02,11,12,22]
  float hq[3];            // limb bias
  SV fcs;                 // limb bias wrench on the hub
  SI Isc;                 // limb composite inertia
  SV j31, j32;            // motion of the foot's body (link3, or link2 on a right-hand module) per unit rate of (q2, q3); per unit q1 it is s1
  V3 x;                   // centre of the foot sphere (hub coords)
};
LM_DEV V3 sel(bool c, V3 a, V3 b) { return v3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
LM_DEV SV sel(bool c, SV a, SV b) { return sv(sel(c, a.w, b.w), sel(c, a.v, b.v)); }

LM_DEV void limb_dynamics(const float* tl, const LimbKin& K, const float qd[3], SV v0, SV avp0, LimbDyn& D) {
  const f2* tp = reinterpret_cast<const f2*>(tl);
  const SI Is = place_inertia<float>(tl + T_I0, K.Rs, K.os);
  const SIP I41 = place_inertia<f2>(tp + T_Q1 / 2, K.R41, K.o41), I32 = place_inertia<f2>(tp + T_Q2 / 2, K.R32, K.o32);
  // velocities / velocity-product accelerations down the two chains
  const SV j1 = qd[0] * K.s1; const SVP j23 = mk2(qd[1], qd[2]) * K.s23, jp = mk2(K.pd, -K.pd) * K.sp12;
  const SV vs = v0 + j1; const SVP v41 = bc(vs) + j23, v32 = v41 + jp;
  const SV as = avp0 + mcross(v0, j1);
  const SVP a41 = bc(as) + mcross(bc(vs), j23);
  const SVP a32 = fma6(mk2(K.pdd, -K.pdd), K.sp12, a41 + mcross(v41, jp));
  // bias wrenches
  const SV ps = Is * as + fcross(vs, Is * vs);
  const SVP p41 = I41 * a41 + fcross(v41, I41 * v41), p32 = I32 * a32 + fcross(v32, I32 * v32);
  const SVP fc = p41 + p32;                                             // (fc4 | fc1)
  D.fcs = ps + lo(fc) + hi(fc);
  const float h1 = sdot(K.s1, D.fcs); const f2 h23 = sdot(K.s23, fc), hp = sdot(K.sp12, p32);      // (h2 | h3), (hp1 | hp2)
  const float g1 = K.g1, dh = hp.x - hp.y;
  D.hq[0] = h1; D.hq[1] = h23.x + g1 * dh; D.hq[2] = h23.y - g1 * dh;
  // composite inertias and the columns of the coupling / joint-space inertia
  const SIP Ic = I41 + I32;                                             // (I4c | I1c)
  D.Isc = Is + lo(Ic) + hi(Ic);
  const SV F1 = D.Isc * K.s1; const SVP F23 = Ic * K.s23, Fp = I32 * K.sp12;      // (F2 | F3), (Fp1 | Fp2)
  const SV dF = lo(Fp) - hi(Fp);
  D.Fq0 = F1; D.Fq1 = fma6(g1, dF, lo(F23)); D.Fq2 = fma6(-g1, dF, hi(F23));
  const SVP s1p = bc(K.s1);
  const float H11 = sdot(K.s1, F1);
  const f2 H1d = sdot(s1p, F23), H1p = sdot(s1p, Fp);                   // (H12 | H13), (H1p1 | H1p2)
  const f2 Hdd = sdot(K.s23, F23), Hdp = sdot(K.s23, Fp), Hpp = sdot(K.sp12, Fp);      // (H22 | H33), (H2p1 | H3p2), (Hp1p1 | Hp2p2)
  const float gg = g1 * g1 * (Hpp.x + Hpp.y), d1p = H1p.x - H1p.y;
  D.H[0] = H11; D.H[1] = H1d.x + g1 * d1p; D.H[2] = H1d.y - g1 * d1p;
  D.H[3] = Hdd.x + 2.0f * g1 * Hdp.x + gg; D.H[4] = -g1 * (Hdp.x + Hdp.y) - gg; D.H[5] = Hdd.y + 2.0f * g1 * Hdp.y + gg;
  // foot collider = the hemispherical end of the long distal link (robot_model.py FOOT_*): link3 (chain dof2 -> p1 = +g(D)) on a left-hand
  // module, link2 (chain dof3 -> p2 = -g(D)) on a right-hand one
  const bool on2 = tl[T_FLAG] != 0.f;
  const V3 off = v3(tl[T_FOOT], tl[T_FOOT + 1], tl[T_FOOT + 2]);
  const SV s2 = lo(K.s23), s3 = hi(K.s23), sp1 = lo(K.sp12), sp2_ = hi(K.sp12);
  D.j31 = sel(on2, (-g1) * sp2_, fma6(g1, sp1, s2));
  D.j32 = sel(on2, fma6(g1, sp2_, s3), (-g1) * sp1);
  D.x = sel(on2, hi(K.o32), lo(K.o32)) + mul(M3{sel(on2, hi(K.R32.c0), lo(K.R32.c0)), sel(on2, hi(K.R32.c1), lo(K.R32.c1)), sel(on2, hi(K.R32.c2), lo(K.R32.c2))}, off);
}

// Projected Gauss-Seidel over the 4 tip contacts of one env (rows n, t1, t2 per contact, limb order).
// The Delassus operator is  W_ij = delta_ij D_i + T_i^T Phi T_j  (arrowhead structure of the hub + limbs system).
// Each lane keeps the current contact-space velocity c (3) of ITS contact, its own full 3x3 block, and the 3x3
// cross blocks X_K = T_i^T B_K towards the other three contacts; lanes take turns, the lane whose turn it is
// relaxes its normal row and then its two friction rows together (one packed update: both see the state the normal row left,
// their mutual coupling enters at the contact's next turn; the pair is projected onto the friction cone), then its three impulse
// increments are quad-broadcast and every lane updates c.  Identical arithmetic (up to rounding) to the oracle's sweep over the dense 12x12 system.
// Row layout: row 0 (normal) in plain registers, rows 1 | 2 (friction) as one packed pair.
struct PgsData { f2 W0t, nrWt; float X0[4][3]; f2 X12[4][3]; };   // own block: (W01 | W02), (-1/(mu W11) | -1/(mu W22)); columns 1, 2 of the X blocks carry mu, row 0 (X0) is scaled by -1/W00 (the normal residual is carried as the unclamped impulse step);
                                                                                   // block towards contact K (K == own limb: the own block) by columns s: X0[K][s] = X[0][s], X12[K][s] = (X[1][s] | X[2][s])

// ---- "four 6-vectors at once" layout of the pass linear algebra.  Component i of the vectors (v0, v1, v2, v3) is an R4: p = (v0[i] | v1[i]),
// q = (v2[i] | v3[i]), so that an operation applied to all four is two packed-fp32 instructions and any single entry is a free half-register
// read.  The pass uses it for (-bA, T0, T1, T2) - the hub bias and the three contact rows of this lane - and for what the hub solve makes of
// them, (a0, B0, B1, B2).
struct R4 { f2 p, q; };
// component-pair layout of ONE 6-vector (w.x,w.y | w.z,v.x | v.y,v.z): what a float4 stash reload delivers in aligned register pairs
struct S6 { f2 a, b, c; };
LM_DEV S6 operator*(float s, S6 x) { const f2 t = sp2(s); S6 r; r.a = t * x.a; r.b = t * x.b; r.c = t * x.c; return r; }
LM_DEV S6 fma6(float s, S6 x, S6 y) { const f2 t = sp2(s); S6 r; r.a = fma_(t, x.a, y.a); r.b = fma_(t, x.b, y.b); r.c = fma_(t, x.c, y.c); return r; }
template <int I> LM_DEV float comp(const S6& x) { return I == 0 ? x.a.x : I == 1 ? x.a.y : I == 2 ? x.b.x : I == 3 ? x.b.y : I == 4 ? x.c.x : x.c.y; }
template <int J> LM_DEV f2 pairc(const S6& x) { return J == 0 ? x.a : J == 1 ? x.b : x.c; }

// Cholesky factor of a symmetric positive definite 6x6 matrix given by its upper triangle: L (strictly lower) and 1 / diagonal
struct Chol6 { float L[6][6]; float d[6]; };
LM_DEV void chol6(const float A[6][6], Chol6& C) {
#pragma unroll
  for (int j = 0; j < 6; j++) {
    float s = A[j][j];
#pragma unroll
    for (int k = 0; k < j; k++) s = fmaf(-C.L[j][k], C.L[j][k], s);
    const float inv = __builtin_amdgcn_rsqf(s);      // v_rsq_f32 (1 ulp); the pivots are O(1e-3 .. 1): no denormal scaling needed
    C.d[j] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float t = A[j][i];
#pragma unroll
      for (int k = 0; k < j; k++) t = fmaf(-C.L[i][k], C.L[j][k], t);
      C.L[i][j] = t * inv;
    }
  }
}
// A x = b for four right-hand sides at once, in place (forward then backward substitution: 84 packed instructions)
LM_DEV void chol6_solve4(const Chol6& C, R4 x[6]) {
#pragma unroll
  for (int i = 0; i < 6; i++) {
    f2 p = x[i].p, q = x[i].q;
#pragma unroll
    for (int k = 0; k < i; k++) { const f2 l = sp2(-C.L[i][k]); p = fma_(l, x[k].p, p); q = fma_(l, x[k].q, q); }
    const f2 d = sp2(C.d[i]); x[i].p = p * d; x[i].q = q * d;
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    f2 p = x[i].p, q = x[i].q;
#pragma unroll
    for (int k = i + 1; k < 6; k++) { const f2 l = sp2(-C.L[k][i]); p = fma_(l, x[k].p, p); q = fma_(l, x[k].q, q); }
    const f2 d = sp2(C.d[i]); x[i].p = p * d; x[i].q = q * d;
  }
}

// 3x3 block of this lane's contact rows T_i against the B vectors of contact K:  X[r][s] = T_i,r . B_K,s  (K == own limb: the own block)
template <int K>
LM_DEV void pgs_cross_blocks(int limb, const R4 T[6], const R4 X[6], const float Wf[6], float X0[3], f2 X12[3]) {
  float a00 = 0.f; f2 a0t = sp2(0.f), c0 = sp2(0.f), c1 = sp2(0.f), c2 = sp2(0.f);
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const float k0 = quad_bcast<K>(X[i].p.y); const f2 k12 = mk2(quad_bcast<K>(X[i].q.x), quad_bcast<K>(X[i].q.y));
    const float t0 = T[i].p.y; const f2 t12 = T[i].q;
    a00 = fmaf(t0, k0, a00); a0t = fma_(sp2(t0), k12, a0t);                                   // row 0: column 0, columns (1 | 2)
    c0 = fma_(t12, sp2(k0), c0); c1 = fma_(t12, sp2(k12.x), c1); c2 = fma_(t12, sp2(k12.y), c2);      // rows (1 | 2): columns 0, 1, 2
  }
  const bool own = (limb == K);
  X0[0] = own ? Wf[0] : a00; X0[1] = own ? Wf[1] : a0t.x; X0[2] = own ? Wf[2] : a0t.y;
  X12[0] = mk2(own ? Wf[1] : c0.x, own ? Wf[2] : c0.y);
  X12[1] = mk2(own ? Wf[3] : c1.x, own ? Wf[4] : c1.y);
  X12[2] = mk2(own ? Wf[4] : c2.x, own ? Wf[5] : c2.y);
}

// One Gauss-Seidel turn: contact K relaxes its normal row, then its two friction rows as a pair (both from the state the normal row
// left), then its three impulse increments are quad-broadcast and EVERY lane, the owner included, applies c += X[K] * d.  Lanes whose
// turn it is not run the same instructions on their own (discarded) candidates; the impulses move by  m * d  with m = 1 on the owner
// lane and 0 elsewhere (one FMA instead of an add and a select).  The friction impulses are carried divided by mu, so that their bound
// is the normal impulse itself and mu sits in the constants (nrWt, columns 1 and 2 of the X blocks).
// Measured alternatives (tools/ab_build.py, profiles/r03_pgs_turn_ab.json): the relaxation as a real branch under the owner lanes'
// execution mask (18 vector + 3 scalar instructions per turn) is 14 % SLOWER per sweep than selects (20 vector instructions).
template <int K>
LM_DEV void pgs_turn(float m, const PgsData& G, float& lam0, f2& lam12, float& c0, f2& c12) {
  const float d0 = __builtin_amdgcn_fmed3f(c0, -lam0, __builtin_inff());               // max(-lam0, -v_n / W00): c0 is carried as -v_n / W00 (one v_max, no canonicalising copy of -lam0)
  lam0 = fmaf(m, d0, lam0);                                                            // owner: the relaxed normal impulse, which bounds its friction rows
  const f2 u12 = fma_(fma_(G.W0t, sp2(d0), c12), G.nrWt, lam12);
  // projection of the friction pair onto the cone |lam_t| <= mu lam_n (here: |u| <= lam0, the pair being carried divided by mu): scale by
  // min(1, lam0 / |u|).  |u| = 0 gives lam0 * inf = inf (or NaN when lam0 = 0 too), and v_min returns 1 for both
  const float sc = fminf(1.0f, lam0 * __builtin_amdgcn_rsqf(fmaf(u12.x, u12.x, u12.y * u12.y)));
  const f2 d12 = fma_(u12, sp2(sc), -lam12);
  lam12 = fma_(sp2(m), d12, lam12);
  const float b0 = quad_bcast<K>(d0), b1 = quad_bcast<K>(d12.x), b2 = quad_bcast<K>(d12.y);
  // the normal increment's terms first: they are ready before the friction pair's broadcasts and fill those broadcasts' wait states
  c0 = fmaf(G.X0[K][2], b2, fmaf(G.X0[K][1], b1, fmaf(G.X0[K][0], b0, c0)));
  c12 = fma_(G.X12[K][2], sp2(b2), fma_(G.X12[K][1], sp2(b1), fma_(G.X12[K][0], sp2(b0), c12)));
}

// T[i].p.y, T[i].q = this lane's three contact rows (hub / plate wrench per unit impulse); X[i].p.y, X[i].q = B = Phi T
struct PgsState { PgsData G; float lam0; f2 lam12; float c0; f2 c12; float m0, m1, m2, m3; };
LM_DEV void pgs_setup(PgsState& S, int limb, float mu, float bn, const float vf[3], const float Wl[6], const R4 T[6], const R4 X[6]) {
  PgsData& G = S.G; float Wf[6];
  // full own block = limb-local part + hub/plate part T_r^T Phi T_s
  {
    f2 a0 = sp2(0.f), a1 = sp2(0.f), a2 = sp2(0.f); float b0 = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      b0 = fmaf(T[i].p.y, X[i].p.y, b0);
      a0 = fma_(sp2(T[i].p.y), X[i].q, a0); a1 = fma_(sp2(T[i].q.x), X[i].q, a1); a2 = fma_(sp2(T[i].q.y), X[i].q, a2);
    }
    Wf[0] = Wl[0] + b0; Wf[1] = Wl[1] + a0.x; Wf[2] = Wl[2] + a0.y;
    Wf[3] = Wl[3] + a1.x; Wf[4] = Wl[4] + a1.y; Wf[5] = Wl[5] + a2.y;
  }
  // friction impulses are carried as lam_t / mu (bound = the normal impulse); mu = 0 pins them at zero
  const float imu = mu > 0.f ? 1.0f / mu : 0.f;
  const float nrW0 = -1.0f / Wf[0]; G.W0t = mk2(Wf[1], Wf[2]); G.nrWt = mk2(-imu / Wf[3], -imu / Wf[5]);
  pgs_cross_blocks<0>(limb, T, X, Wf, G.X0[0], G.X12[0]); pgs_cross_blocks<1>(limb, T, X, Wf, G.X0[1], G.X12[1]);
  pgs_cross_blocks<2>(limb, T, X, Wf, G.X0[2], G.X12[2]); pgs_cross_blocks<3>(limb, T, X, Wf, G.X0[3], G.X12[3]);
#pragma unroll
  for (int k = 0; k < 4; k++) { G.X0[k][0] *= nrW0; G.X0[k][1] *= mu * nrW0; G.X0[k][2] *= mu * nrW0; G.X12[k][1] = sp2(mu) * G.X12[k][1]; G.X12[k][2] = sp2(mu) * G.X12[k][2]; }
  S.lam0 = 0.f; S.lam12 = sp2(0.f);
  S.c0 = (vf[0] + bn) * nrW0; S.c12 = mk2(vf[1], vf[2]);
  S.m0 = limb == 0 ? 1.f : 0.f; S.m1 = limb == 1 ? 1.f : 0.f; S.m2 = limb == 2 ? 1.f : 0.f; S.m3 = limb == 3 ? 1.f : 0.f;
}
// sweeps it0 (even) ... it1 - 1.  Sweeps alternate direction (contacts 0,1,2,3 then 3,2,1,0): no limb is systematically relaxed first,
// which removes the ordering bias an unconverged Gauss-Seidel solve would otherwise leave between the four limbs
LM_DEV void pgs_sweeps(PgsState& S, int it0, int it1) {
  for (int it = it0; it < it1; it += 2) {
    pgs_turn<0>(S.m0, S.G, S.lam0, S.lam12, S.c0, S.c12);
    pgs_turn<1>(S.m1, S.G, S.lam0, S.lam12, S.c0, S.c12);
    pgs_turn<2>(S.m2, S.G, S.lam0, S.lam12, S.c0, S.c12);
    pgs_turn<3>(S.m3, S.G, S.lam0, S.lam12, S.c0, S.c12);
    if (it + 1 < it1) {
      pgs_turn<3>(S.m3, S.G, S.lam0, S.lam12, S.c0, S.c12);
      pgs_turn<2>(S.m2, S.G, S.lam0, S.lam12, S.c0, S.c12);
      pgs_turn<1>(S.m1, S.G, S.lam0, S.lam12, S.c0, S.c12);
      pgs_turn<0>(S.m0, S.G, S.lam0, S.lam12, S.c0, S.c12);
    }
  }
}
// impulses found so far and the hub / plate velocity change  w = Phi sum_j T_j lam_j = sum_j B_j lam_j
LM_DEV void pgs_finish(const PgsState& S, float mu, const R4 X[6], float lam[3], float w[6]) {
  const f2 l12 = sp2(mu) * S.lam12;
  lam[0] = S.lam0; lam[1] = l12.x; lam[2] = l12.y;
#pragma unroll
  for (int i = 0; i < 6; i++) { const f2 t = l12 * X[i].q; w[i] = quad_sum(fmaf(S.lam0, X[i].p.y, t.x + t.y)); }
}

// free rigid body carried as (position, quaternion, body-coordinate spatial velocity about its origin)
struct FreeBody { V3 p; Q4 q; SV u; };

LM_DEV void integrate_free(FreeBody& F, const M3& R, float dt) {
  V3 ww = mul(R, F.u.w);
  float wn = sqrtf(dot(ww, ww)), th = wn * dt;
  Q4 dq;
  if (th < 1e-8f) { dq.w = 1.f; dq.x = 0.5f * dt * ww.x; dq.y = 0.5f * dt * ww.y; dq.z = 0.5f * dt * ww.z; }
  else { float sh, ch; lm_sincos(0.5f * th, &sh, &ch); float s = sh / wn; dq.w = ch; dq.x = s * ww.x; dq.y = s * ww.y; dq.z = s * ww.z; }
  Q4 qn = qmul(dq, F.q);
  float rn = rsqrtf(qn.w * qn.w + qn.x * qn.x + qn.y * qn.y + qn.z * qn.z);
  F.q.w = qn.w * rn; F.q.x = qn.x * rn; F.q.y = qn.y * rn; F.q.z = qn.z * rn;
  M3 Rn = quat_to_mat(F.q.w, F.q.x, F.q.y, F.q.z);
  F.p = fma3(dt, mul(Rn, F.u.v), F.p);
}

// Per-lane stash in LDS for the pass-invariant terms of a sub-step (they are needed at the top of each of the two
// drive passes but not during the contact iterations; keeping them in registers across the PGS loop spills).
// Layout [slot][lane] as float4 -> conflict-free 16-byte accesses.
#ifdef LM_WAVES2
// k_step compiled for TWO wavefronts per SIMD.  That needs <= 256 registers per lane (amdgpu_waves_per_eu(2, 2): the
// compiler spills the rest to scratch) and <= 20 KB of LDS per wavefront (8 blocks per CU): the stash shrinks to 18 slots (the limb's last two
// inertia entries ride in the free half of slot 4) and the output staging (sObs, sSt: dead until the task layer, when the stash is dead) lives in
// the stash's memory.  Same arithmetic in the same order, so the same bits (tests/test_gpu_full_size.py).  The product compiles it as a SECOND
// translation unit (lm_engine_w2.hip: k_step_w2, the locomotion specialisation only) that lm_step dispatches beyond 32 768 envs - more than two
// generations of one-wavefront workgroups - for un-randomised velocity-drive locomotion engines: slower below (16 384 envs: 45 against 39 us),
// even at 24 576 / 32 768, + 5-8 % from 36 864, + 9 % at 65 536, + 14 % at 131 072, + 18 % at 262 144.
// (-DLM_WAVES2 on the whole library, tools/ab_build.py, is the A/B build of round 4: every step kernel then has this layout.)
#define STASH_SLOTS 18
#define LM_STEP_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define STASH_SLOTS 22
#define LM_STEP_ATTR
#endif
struct Stash {
  float4* base; int lane;
  LM_DEV void put(int slot, float a, float b, float c, float d) const { base[slot * 64 + lane] = make_float4(a, b, c, d); }
  LM_DEV float4 get(int slot) const { return base[slot * 64 + lane]; }
};
LM_DEV void stash_sv3(const Stash& S, int slot, SV a, SV b, SV c, float e0 = 0.f, float e1 = 0.f) {      // 18 floats -> 5 slots (the last one half used: e0, e1 ride there)
  S.put(slot + 0, a.w.x, a.w.y, a.w.z, a.v.x); S.put(slot + 1, a.v.y, a.v.z, b.w.x, b.w.y);
  S.put(slot + 2, b.w.z, b.v.x, b.v.y, b.v.z); S.put(slot + 3, c.w.x, c.w.y, c.w.z, c.v.x); S.put(slot + 4, c.v.y, c.v.z, e0, e1);
}
LM_DEV void unstash_sv3(const Stash& S, int slot, SV& a, SV& b, SV& c) {
  float4 t0 = S.get(slot), t1 = S.get(slot + 1), t2 = S.get(slot + 2), t3 = S.get(slot + 3), t4 = S.get(slot + 4);
  a = sv(v3(t0.x, t0.y, t0.z), v3(t0.w, t1.x, t1.y)); b = sv(v3(t1.z, t1.w, t2.x), v3(t2.y, t2.z, t2.w)); c = sv(v3(t3.x, t3.y, t3.z), v3(t3.w, t4.x, t4.y));
}


#ifdef LM_COUNT_PASS2      // diagnostic builds only (tools/pass2_count.py): wavefront-sub-steps run / of those with a second drive pass
__device__ unsigned int lm_dbg_pass2[2];
extern "C" void lm_dbg_pass2_read(unsigned int* out, int clear) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(lm_dbg_pass2), sizeof(lm_dbg_pass2));
  if (clear) { unsigned int z[2] = {0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(lm_dbg_pass2), z, sizeof(z)); }
}
#endif

// One physics sub-step of one env (4 lanes).  MODE 0: F is the robot base.  MODE 1: F is the plate, the
// robot base is fixed at (Rb, pb).
template <int MODE, int VAR, int DR>
LM_DEV void substep(const lm_params* __restrict__ P, const float* th, const float* tl, int limb, const Stash& St,
                    FreeBody& F, const M3& Rfix, V3 pfix, float q[3], float qd[3], const float tgt[3], float tau_acc[3], const DrPhys& X) {
  const float dt = P->dt, kd = P->kd;
  const float cjv[3] = {VAR ? (DR ? X.cj[0] : P->joint_damping) : 0.f, VAR ? (DR ? X.cj[1] : P->joint_damping) : 0.f, VAR ? (DR ? X.cj[2] : P->joint_damping) : 0.f};
  const float tmax[3] = {DR ? X.tmax[0] : P->tau_max, DR ? X.tmax[1] : P->tau_max, DR ? X.tmax[2] : P->tau_max};
  M3 Rf = quat_to_mat(F.q.w, F.q.x, F.q.y, F.q.z);
  float bn;
  {
    M3 Rb; V3 pb; SV v0;
    if (MODE == 0) { Rb = Rf; pb = F.p; v0 = F.u; } else { Rb = Rfix; pb = pfix; v0 = sv(v3(0, 0, 0), v3(0, 0, 0)); }
    SV avp0 = sv(v3(0, 0, 0), DR ? mulT(Rb, -X.g) : P->gravity * row2(Rb));      // fictitious acceleration = -gravity
    LimbKin K; limb_kinematics(tl, q, qd, K);
    LimbDyn D; limb_dynamics(tl, K, qd, v0, avp0, D);
    LM_STAMP(1);
    if (MODE == 0) {
      // the hub body itself rides on limb 0's contribution to the quad reductions
      const float m0 = (limb == 0) ? 1.f : 0.f;
      SI I0 = hub_inertia(th);
      const V3 com0 = (1.0f / I0.m) * I0.h;
      I0.m *= m0; I0.h = m0 * I0.h; I0.xx *= m0; I0.yy *= m0; I0.zz *= m0; I0.xy *= m0; I0.xz *= m0; I0.yz *= m0;
      D.Isc = D.Isc + I0;
      D.fcs = D.fcs + I0 * avp0 + fcross(v0, I0 * v0);
      if (DR) { V3 fh = m0 * mulT(Rb, X.f); D.fcs = D.fcs - sv(cross(com0, fh), fh); }      // randomised force on the base link, at its COM
    }
    // ---- contact geometry of this limb's tip
    V3 C0, C1, C2;        // contact axes (n, t1, t2) in hub coordinates
    float phi;
    if (MODE == 0) {
      // contact axes in hub coordinates: n = world z; t1 = the base's x axis projected onto the ground plane, t2 = n x t1.  The friction
      // basis turns with the robot, so the dynamics do not depend on its heading (exactly, even with the unconverged solver)
      C0 = row2(Rb);
      C1 = rsqrtf(fmaxf(1.0f - C0.x * C0.x, 1.0e-12f)) * v3(1.0f - C0.x * C0.x, -C0.x * C0.y, -C0.x * C0.z);
      C2 = cross(C0, C1);
      phi = pb.z + dot(C0, D.x) - P->tip_radius;
      D.x = fma3(-P->tip_radius, C0, D.x);      // from here on: the contact point on the sphere's surface (Jacobians are taken there)
    } else {
      V3 xw = pb + mul(Rb, D.x);
      V3 yc = mulT(Rf, xw - F.p);                // sphere centre, plate coordinates
      V3 y = yc - v3(P->plate_center[0], P->plate_center[1], P->plate_center[2]);
      // contact face = the slab face on the robot's side of the plate (robust to deep initial overlap)
      float sg = (mulT(Rf, pb - F.p).z - P->plate_center[2] >= 0.f) ? 1.f : -1.f;
      phi = sg * y.z - P->plate_half[2] - P->tip_radius;
      if (fabsf(y.x) > P->plate_half[0] || fabsf(y.y) > P->plate_half[1]) phi = 1.0e3f;
      // contact axes in plate coords: n=(0,0,sg) t1=(1,0,0) t2=(0,sg,0); in hub coords: Rb^T Rf axis
      M3 Mrp = mulTA(Rb, Rf);
      C0 = sg * Mrp.c2; C1 = Mrp.c0; C2 = sg * Mrp.c1;
      D.x = fma3(-P->tip_radius, C0, D.x);      // the contact point on the sphere's surface, hub coordinates ...
      V3 y0 = v3(yc.x, yc.y, yc.z - sg * P->tip_radius);      // ... and plate coordinates
      V3 a0 = v3(0, 0, sg), a1 = v3(1, 0, 0), a2 = v3(0, sg, 0);
      SV Tp0 = sv(-cross(y0, a0), -a0), Tp1 = sv(-cross(y0, a1), -a1), Tp2 = sv(-cross(y0, a2), -a2);
      // plate free motion
      SI Ip = load_si(P->plate_si);
      SV hp = fcross(F.u, Ip * F.u);
      V3 fg = DR ? P->plate_mass * mulT(Rf, X.g) : (-P->plate_mass * P->gravity) * row2(Rf);
      V3 c = v3(P->plate_com[0], P->plate_com[1], P->plate_com[2]);
      hp = hp - sv(cross(c, fg), fg);
      float Ph[6][6];
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) Ph[i][j] = P->plate_phi[6 * i + j];
      SV up_free = F.u - dt * mul66(Ph, hp);
      // pass-invariant contact rows in the four-vector layout: slots 10-12 (up_free[i] | Tp0[i]), slots 13-15 (Tp1[i] | Tp2[i])
      St.put(10, up_free.w.x, Tp0.w.x, up_free.w.y, Tp0.w.y); St.put(11, up_free.w.z, Tp0.w.z, up_free.v.x, Tp0.v.x); St.put(12, up_free.v.y, Tp0.v.y, up_free.v.z, Tp0.v.z);
      St.put(13, Tp1.w.x, Tp2.w.x, Tp1.w.y, Tp2.w.y); St.put(14, Tp1.w.z, Tp2.w.z, Tp1.v.x, Tp2.v.x); St.put(15, Tp1.v.y, Tp2.v.y, Tp1.v.z, Tp2.v.z);
    }
    bn = (phi >= 0.f) ? phi / dt : fmaxf(P->baumgarte * phi / dt, -P->max_depen_vel);
    // tip linear velocity per unit generalized rate, contact coordinates
    V3 e0 = K.s1.v + cross(K.s1.w, D.x);
    V3 e1 = D.j31.v + cross(D.j31.w, D.x);
    V3 e2 = D.j32.v + cross(D.j32.w, D.x);
#ifdef LM_WAVES2
    stash_sv3(St, 0, D.Fq0, D.Fq1, D.Fq2, D.Isc.xz, D.Isc.yz);
#else
    stash_sv3(St, 0, D.Fq0, D.Fq1, D.Fq2);
#endif
    St.put(5, D.H[0], D.H[1], D.H[2], D.H[3]); St.put(6, D.H[4], D.H[5], D.hq[0], D.hq[1]);
    // Jq = contact-coordinate tip velocity per unit joint rate: row 0 plain, rows 1 and 2 interleaved (Jq[1][c] | Jq[2][c])
    St.put(7, D.hq[2], dot(C0, e0), dot(C0, e1), dot(C0, e2));
    St.put(8, dot(C1, e0), dot(C2, e0), dot(C1, e1), dot(C2, e1)); St.put(9, dot(C1, e2), dot(C2, e2), 0.f, 0.f);
    if (MODE == 0) {
      // hub rows Jb_r = [x x C_r ; C_r] of the three contact axes and the limb's bias wrench, in the four-vector layout:
      // slots 10-12 (fcs[i] | Jb0[i]), slots 13-15 (Jb1[i] | Jb2[i]); slots 16-18 the limb's composite inertia
      const V3 n0 = cross(D.x, C0), n1 = cross(D.x, C1), n2 = cross(D.x, C2);
      St.put(10, D.fcs.w.x, n0.x, D.fcs.w.y, n0.y); St.put(11, D.fcs.w.z, n0.z, D.fcs.v.x, C0.x); St.put(12, D.fcs.v.y, C0.y, D.fcs.v.z, C0.z);
      St.put(13, n1.x, n2.x, n1.y, n2.y); St.put(14, n1.z, n2.z, C1.x, C2.x); St.put(15, C1.y, C2.y, C1.z, C2.z);
      St.put(16, D.Isc.m, D.Isc.h.x, D.Isc.h.y, D.Isc.h.z); St.put(17, D.Isc.xx, D.Isc.yy, D.Isc.zz, D.Isc.xy);
#ifndef LM_WAVES2
      St.put(18, D.Isc.xz, D.Isc.yz, 0.f, 0.f);
#endif
    }
  }

  LM_STAMP(2);
  // effort mode (RobotOmni.take_action, robot.py:455-459): tgt IS the joint torque, gains off = the constant-torque branch from the start
  const bool effort = (VAR == 0) && P->drive_mode == LM_DRIVE_EFFORT;
  bool sat[3] = {effort, effort, effort}; float tsat[3] = {effort ? tgt[0] : 0.f, effort ? tgt[1] : 0.f, effort ? tgt[2] : 0.f};
  // PD-actuator families: the reference evaluates  clamp(kp (q* - q) - kd qd, +-max_effort)  on the state BEFORE the sub-step and holds it
  // (quadruped_pose_control_custom_controller.py:289-293), so which joints sit on the limit is known up front: those get the constant limit
  // torque, the others the implicit form of the same PD law, in ONE pass.  The implicit (end-of-step) torque of an unsaturated joint leaves the
  // limit in 0.02 % of the joint-sub-steps under random actions; lm_params.pd_second_pass = 1 puts those on the limit too and solves again.  That
  // happens in 1-3 % of the wavefront-sub-steps (tools/pass2_count.py), but a step lasts as long as its slowest wavefront and one of the 256
  // nearly always has one: +6 us per step, which is why it is off by default.
  // (Variant 0 with a finite tau_max limits the force of an implicit drive, which only the solve can tell: two passes.)
  if (VAR) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float tau = kd * (tgt[a] - qd[a]);
      const bool hi_ = tau > tmax[a], lo_ = tau < -tmax[a];
      sat[a] = hi_ || lo_; tsat[a] = hi_ ? tmax[a] : -tmax[a];
    }
  }
  float qdn[3]; SV un;
  for (int pass = 0; pass < 2; pass++) {
    asm volatile("" ::: "memory");          // keep the stash reloads inside the pass (no hoisting across the PGS loop)
    S6 Fq0, Fq1, Fq2; float isc_xz = 0.f, isc_yz = 0.f;
    {
      const float4 t0 = St.get(0), t1 = St.get(1), t2 = St.get(2), t3 = St.get(3), t4 = St.get(4);
      isc_xz = t4.z; isc_yz = t4.w;
      Fq0.a = mk2(t0.x, t0.y); Fq0.b = mk2(t0.z, t0.w); Fq0.c = mk2(t1.x, t1.y);
      Fq1.a = mk2(t1.z, t1.w); Fq1.b = mk2(t2.x, t2.y); Fq1.c = mk2(t2.z, t2.w);
      Fq2.a = mk2(t3.x, t3.y); Fq2.b = mk2(t3.z, t3.w); Fq2.c = mk2(t4.x, t4.y);
    }
    const float4 h5 = St.get(5), h6 = St.get(6), h7 = St.get(7), h8 = St.get(8), h9 = St.get(9);
    float Ha[6] = {h5.x, h5.y, h5.z, h5.w, h6.x, h6.y}, r[3];
    const float hq[3] = {h6.z, h6.w, h7.x};
    const float Jq[3][3] = {{h7.y, h7.z, h7.w}, {h8.x, h8.z, h9.x}, {h8.y, h8.w, h9.y}};
    const f2 j12[3] = {mk2(h8.x, h8.y), mk2(h8.z, h8.w), mk2(h9.x, h9.y)};      // (Jq[1][c] | Jq[2][c])
    const int di[3] = {0, 3, 5};
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float cj = cjv[a];
      const float dd = sat[a] ? dt * cj : dt * (kd + cj);                  // viscous joint damping is implicit in both cases
      const float rr = sat[a] ? tsat[a] : kd * (tgt[a] - qd[a]);
      Ha[di[a]] += dd; r[a] = rr - cj * qd[a] - hq[a];
    }
    float Hi[6]; inv3sym(Ha, Hi);
    // K = Fq Hinv  (columns)
    const S6 K0 = fma6(Hi[0], Fq0, fma6(Hi[1], Fq1, Hi[2] * Fq2));
    const S6 K1 = fma6(Hi[1], Fq0, fma6(Hi[3], Fq1, Hi[4] * Fq2));
    const S6 K2 = fma6(Hi[2], Fq0, fma6(Hi[4], Fq1, Hi[5] * Fq2));
    float qdd[3], qdf[3], v0f[6], vf[3];
    R4 T[6], X[6];
    // pass-invariant rows (slots 10-15): PB[i].p = (fcs[i] | Jb0[i]) or (up_free[i] | Tp0[i]),  PB[i].q = (row 1 | row 2)
    R4 PB[6];
    {
      const float4 g0 = St.get(10), g1 = St.get(11), g2 = St.get(12), g3 = St.get(13), g4 = St.get(14), g5 = St.get(15);
      PB[0].p = mk2(g0.x, g0.y); PB[1].p = mk2(g0.z, g0.w); PB[2].p = mk2(g1.x, g1.y); PB[3].p = mk2(g1.z, g1.w); PB[4].p = mk2(g2.x, g2.y); PB[5].p = mk2(g2.z, g2.w);
      PB[0].q = mk2(g3.x, g3.y); PB[1].q = mk2(g3.z, g3.w); PB[2].q = mk2(g4.x, g4.y); PB[3].q = mk2(g4.z, g4.w); PB[4].q = mk2(g5.x, g5.y); PB[5].q = mk2(g5.z, g5.w);
    }
    if (MODE == 0) {
#ifdef LM_WAVES2
      const float4 g16 = St.get(16), g17 = St.get(17), g18 = make_float4(isc_xz, isc_yz, 0.f, 0.f);
#else
      const float4 g16 = St.get(16), g17 = St.get(17), g18 = St.get(18); (void)isc_xz; (void)isc_yz;
#endif
      SI Isc; Isc.m = g16.x; Isc.h = v3(g16.y, g16.z, g16.w); Isc.xx = g17.x; Isc.yy = g17.y; Isc.zz = g17.z; Isc.xy = g17.w; Isc.xz = g18.x; Isc.yz = g18.y;
      // articulated hub inertia  A = sum over the quad of (Isc - K F^T): rows of pairs, upper triangle only
      float A0[6][6]; si_to_66(Isc, A0);
      float A[6][6];
#define LM_AROW(I) { \
        const f2 k0 = sp2(comp<I>(K0)), k1 = sp2(comp<I>(K1)), k2 = sp2(comp<I>(K2)); \
        _Pragma("unroll") for (int jp = (I) / 2; jp < 3; jp++) { \
          const f2 f0 = jp == 0 ? Fq0.a : jp == 1 ? Fq0.b : Fq0.c, f1 = jp == 0 ? Fq1.a : jp == 1 ? Fq1.b : Fq1.c, f2_ = jp == 0 ? Fq2.a : jp == 1 ? Fq2.b : Fq2.c; \
          const f2 v = fma_(-k2, f2_, fma_(-k1, f1, fma_(-k0, f0, mk2(A0[I][2 * jp], A0[I][2 * jp + 1])))); \
          if (2 * jp >= (I)) A[I][2 * jp] = quad_sum(v.x); \
          A[I][2 * jp + 1] = quad_sum(v.y); } }
      LM_AROW(0) LM_AROW(1) LM_AROW(2) LM_AROW(3) LM_AROW(4) LM_AROW(5)
#undef LM_AROW
      Chol6 Ch; chol6(A, Ch);
      // (bias | row 0) and (row 1 | row 2) with the limb's joints eliminated:  P = PB + coefficient x K,  T_r = Jb_r - K Jq_r^T
      const f2 cp[3] = {mk2(r[0], -Jq[0][0]), mk2(r[1], -Jq[0][1]), mk2(r[2], -Jq[0][2])};
#define LM_PROW(I) { \
        const f2 k0 = sp2(comp<I>(K0)), k1 = sp2(comp<I>(K1)), k2 = sp2(comp<I>(K2)); \
        T[I].p = fma_(cp[2], k2, fma_(cp[1], k1, fma_(cp[0], k0, PB[I].p))); \
        T[I].q = fma_(-j12[2], k2, fma_(-j12[1], k1, fma_(-j12[0], k0, PB[I].q))); \
        X[I].p = mk2(-quad_sum(T[I].p.x), T[I].p.y); X[I].q = T[I].q; }
      LM_PROW(0) LM_PROW(1) LM_PROW(2) LM_PROW(3) LM_PROW(4) LM_PROW(5)
#undef LM_PROW
      chol6_solve4(Ch, X);          // X = (a0 | B0), (B1 | B2): hub acceleration and Phi T without forming Phi
      float t0 = r[0], t1 = r[1], t2 = r[2];
#define LM_TROW(I) { const float a = X[I].p.x; t0 = fmaf(-comp<I>(Fq0), a, t0); t1 = fmaf(-comp<I>(Fq1), a, t1); t2 = fmaf(-comp<I>(Fq2), a, t2); }
      LM_TROW(0) LM_TROW(1) LM_TROW(2) LM_TROW(3) LM_TROW(4) LM_TROW(5)
#undef LM_TROW
      qdd[0] = Hi[0] * t0 + Hi[1] * t1 + Hi[2] * t2; qdd[1] = Hi[1] * t0 + Hi[3] * t1 + Hi[4] * t2; qdd[2] = Hi[2] * t0 + Hi[4] * t1 + Hi[5] * t2;
      const float u6[6] = {F.u.w.x, F.u.w.y, F.u.w.z, F.u.v.x, F.u.v.y, F.u.v.z};
#pragma unroll
      for (int i = 0; i < 6; i++) v0f[i] = fmaf(dt, X[i].p.x, u6[i]);
    } else {
      qdd[0] = Hi[0] * r[0] + Hi[1] * r[1] + Hi[2] * r[2]; qdd[1] = Hi[1] * r[0] + Hi[3] * r[1] + Hi[4] * r[2]; qdd[2] = Hi[2] * r[0] + Hi[4] * r[1] + Hi[5] * r[2];
#pragma unroll
      for (int i = 0; i < 6; i++) { v0f[i] = PB[i].p.x; T[i] = PB[i]; }
      // B = Phi T with the plate's constant inverse inertia
#pragma unroll
      for (int i = 0; i < 6; i++) {
        f2 bp = sp2(0.f), bq = sp2(0.f);
#pragma unroll
        for (int j = 0; j < 6; j++) { const f2 ph = sp2(P->plate_phi[6 * i + j]); bp = fma_(ph, T[j].p, bp); bq = fma_(ph, T[j].q, bq); }
        X[i].p = bp; X[i].q = bq;
      }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) qdf[a] = fmaf(dt, qdd[a], qd[a]);
    // contact operator
    float Wl[6];
    float JH[3][3];   // Jq * Hinv
#pragma unroll
    for (int rr = 0; rr < 3; rr++) {
      JH[rr][0] = Jq[rr][0] * Hi[0] + Jq[rr][1] * Hi[1] + Jq[rr][2] * Hi[2];
      JH[rr][1] = Jq[rr][0] * Hi[1] + Jq[rr][1] * Hi[3] + Jq[rr][2] * Hi[4];
      JH[rr][2] = Jq[rr][0] * Hi[2] + Jq[rr][1] * Hi[4] + Jq[rr][2] * Hi[5];
    }
    Wl[0] = JH[0][0] * Jq[0][0] + JH[0][1] * Jq[0][1] + JH[0][2] * Jq[0][2];
    Wl[1] = JH[0][0] * Jq[1][0] + JH[0][1] * Jq[1][1] + JH[0][2] * Jq[1][2];
    Wl[2] = JH[0][0] * Jq[2][0] + JH[0][1] * Jq[2][1] + JH[0][2] * Jq[2][2];
    Wl[3] = JH[1][0] * Jq[1][0] + JH[1][1] * Jq[1][1] + JH[1][2] * Jq[1][2];
    Wl[4] = JH[1][0] * Jq[2][0] + JH[1][1] * Jq[2][1] + JH[1][2] * Jq[2][2];
    Wl[5] = JH[2][0] * Jq[2][0] + JH[2][1] * Jq[2][1] + JH[2][2] * Jq[2][2];
    // free contact-space velocity: hub / plate rows (before the joints were eliminated) on v0f, plus the joint part
    {
      float a = 0.f; f2 b = sp2(0.f);
#pragma unroll
      for (int i = 0; i < 6; i++) { a = fmaf(PB[i].p.y, v0f[i], a); b = fma_(PB[i].q, sp2(v0f[i]), b); }
      vf[0] = a + (Jq[0][0] * qdf[0] + Jq[0][1] * qdf[1] + Jq[0][2] * qdf[2]);
      vf[1] = b.x + (Jq[1][0] * qdf[0] + Jq[1][1] * qdf[1] + Jq[1][2] * qdf[2]);
      vf[2] = b.y + (Jq[2][0] * qdf[0] + Jq[2][1] * qdf[1] + Jq[2][2] * qdf[2]);
    }
    float lam[3], w[6];
    LM_STAMP(3);
    PgsState S; pgs_setup(S, limb, P->mu, bn, vf, Wl, T, X);
    pgs_sweeps(S, 0, P->pgs_iters);
    pgs_finish(S, P->mu, X, lam, w);
    un = sv(v3(v0f[0] + w[0], v0f[1] + w[1], v0f[2] + w[2]), v3(v0f[3] + w[3], v0f[4] + w[4], v0f[5] + w[5]));
#pragma unroll
    for (int a = 0; a < 3; a++) qdn[a] = qdf[a] + JH[0][a] * lam[0] + JH[1][a] * lam[1] + JH[2][a] * lam[2];
    if (MODE == 0) {
#define LM_WROW(I) { qdn[0] = fmaf(-comp<I>(K0), w[I], qdn[0]); qdn[1] = fmaf(-comp<I>(K1), w[I], qdn[1]); qdn[2] = fmaf(-comp<I>(K2), w[I], qdn[2]); }
      LM_WROW(0) LM_WROW(1) LM_WROW(2) LM_WROW(3) LM_WROW(4) LM_WROW(5)
#undef LM_WROW
    }
    LM_STAMP(4);
    if (pass == 0) {
      if (effort || (VAR != 0 && !P->pd_second_pass)) break;
      int any = 0;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const float tau = kd * (tgt[a] - qdn[a]);
        const bool hi_ = tau > tmax[a], lo_ = tau < -tmax[a];
        if (VAR) {                             // joints already on the limit (the pre-step decision) stay there
          const bool nw = !sat[a] && (hi_ || lo_);
          tsat[a] = nw ? (hi_ ? tmax[a] : -tmax[a]) : tsat[a]; sat[a] = sat[a] || nw; any |= nw ? 1 : 0;
        } else {
          sat[a] = hi_ || lo_; tsat[a] = hi_ ? tmax[a] : -tmax[a]; any |= (hi_ || lo_) ? 1 : 0;
        }
      }
      any = quad_sum_i(any);
#ifdef LM_COUNT_PASS2
      { const bool wa = __any(any); if ((threadIdx.x & 63) == 0) { atomicAdd(&lm_dbg_pass2[0], 1u); if (wa) atomicAdd(&lm_dbg_pass2[1], 1u); } }
#endif
      if (!__any(any)) break;                  // wave-uniform: nobody saturated
      // envs without saturation redo the identical unsaturated solve in pass 1 (same result)
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {      // driven joints are speed-limited like PhysX's maxJointVelocity (config_module_joints.py:11,61-69)
    // the LOGGED drive torque (observation 88, mechanical power): clipped like the reference's (…custom_controller.py:289-293); the implicit torque
    // applied to an unsaturated joint can exceed the limit in 0.02 % of the joint-sub-steps when pd_second_pass = 0 - dynamics only, never logged
    if (VAR) tau_acc[a] += sat[a] ? tsat[a] : fminf(fmaxf(kd * (tgt[a] - qdn[a]), -tmax[a]), tmax[a]);
    const float vm = DR ? X.vmax[a] : P->max_joint_vel;
    float v = fminf(fmaxf(qdn[a], -vm), vm);
    qd[a] = v; q[a] = fmaf(dt, v, q[a]);
  }
  F.u = un;
  integrate_free(F, Rf, dt);
  LM_STAMP(5);
}

// ------------------------------------------------------------------------------------------------
// task layer (obs / reward / termination), one env = 4 lanes; restates
// quadruped_pose_control.py:301-426,428-560,562-633 and quadruped_manipulate_plate.py:311-435,569-652
// ------------------------------------------------------------------------------------------------
struct TaskIn {
  float q[3], qd[3], acc[3], act[3];       // this limb's joints (dof1, dof2, dof3)
  float torque[3], tgtq[3];                // custom-controller tasks: logged torque, current joint position targets
  V3 tipw, knee2, knee3;                   // world positions of this limb's tip and knees
  V3 fp; Q4 fq; V3 lin, ang;               // free body (base or plate) world pose / velocity
};
struct TaskState { float lact[3]; V3 ltip; Q4 goal; int succ, consec, greset, reset, progress; float ltgt[3]; float lrd; };
struct TaskOut { float rew; float terms[11]; };

template <int MODE, int VAR>
LM_DEV void task_eval(const lm_params* __restrict__ P, int limb, int envl, const TaskIn& I, TaskState& S, TaskOut& O,
                      float* sObs, float* sSt) {
  S.progress += 1;
  // Every parameter the task layer reads, fetched in ONE batch of loads: read where they are used (inside || chains and after branches) each
  // one was a load the compiler may not hoist, i.e. an exposed cache round trip for the lone wavefront - twenty of them in a row.
  struct { float s_pos, s_lin, s_ang, s_q, s_qd, quat_scale, rot_eps, trans_scale, acc_scale, rate_scale, bonus, limit_pen, fall_pen, succ_thresh, h_base, h_corner, h_knee;
           float d23_pen[2], d23_rst[2], d1_pen[2], d1_rst[2], corner[3]; int max_consec, max_episode; } C;
  C.s_pos = P->s_pos; C.s_lin = P->s_lin; C.s_ang = P->s_ang; C.s_q = P->s_q; C.s_qd = P->s_qd; C.quat_scale = P->quat_scale; C.rot_eps = P->rot_eps;
  C.trans_scale = P->trans_scale; C.acc_scale = P->acc_scale; C.rate_scale = P->rate_scale; C.bonus = P->bonus; C.limit_pen = P->limit_pen;
  C.fall_pen = P->fall_pen; C.succ_thresh = P->succ_thresh; C.h_base = P->h_base; C.h_corner = P->h_corner; C.h_knee = P->h_knee;
  C.d23_pen[0] = P->d23_pen[0]; C.d23_pen[1] = P->d23_pen[1]; C.d23_rst[0] = P->d23_rst[0]; C.d23_rst[1] = P->d23_rst[1];
  C.d1_pen[0] = P->d1_pen[limb][0]; C.d1_pen[1] = P->d1_pen[limb][1]; C.d1_rst[0] = P->d1_rst[limb][0]; C.d1_rst[1] = P->d1_rst[limb][1];
  C.corner[0] = P->corner[limb][0]; C.corner[1] = P->corner[limb][1]; C.corner[2] = P->corner[limb][2];
  C.max_consec = P->max_consec; C.max_episode = P->max_episode;
  V3 opos, olin, oang; Q4 oq; M3 Rr; V3 pr;
  if (MODE == 0) {
    Rr = quat_to_mat(I.fq.w, I.fq.x, I.fq.y, I.fq.z); pr = I.fp;
    opos = mulT(Rr, -I.fp); oq = qconj(I.fq); olin = mulT(Rr, -I.lin); oang = mulT(Rr, -I.ang);
  } else {
    Q4 qr; qr.w = P->fixed_base_quat[0]; qr.x = P->fixed_base_quat[1]; qr.y = P->fixed_base_quat[2]; qr.z = P->fixed_base_quat[3];
    Rr = quat_to_mat(qr.w, qr.x, qr.y, qr.z); pr = v3(P->fixed_base_pos[0], P->fixed_base_pos[1], P->fixed_base_pos[2]);
    opos = mulT(Rr, I.fp - pr);
    oq = qmul(qconj(qr), I.fq);
    if (oq.w < 0.f) { oq.w = -oq.w; oq.x = -oq.x; oq.y = -oq.y; oq.z = -oq.z; }
    olin = mulT(Rr, I.lin); oang = mulT(Rr, I.ang);
  }
  V3 btip = mulT(Rr, I.tipw - pr);
  Q4 qd_ = qmul(oq, qconj(S.goal));
  constexpr bool var1 = (VAR == 1), var2 = (VAR == 2), pd = (VAR >= 1); const int NO = var1 ? LM_MAX_OBS : 64;
  float fl = (qd_.w < 0.f && !var1) ? -1.f : 1.f;        // the custom-controller tasks do not flip the sign (…custom_controller.py:429-431)
  Q4 qf; qf.w = fl * qd_.w; qf.x = fl * qd_.x; qf.y = fl * qd_.y; qf.z = fl * qd_.z;
  M3 Ro = quat_to_mat(oq.w, oq.x, oq.y, oq.z);
  V3 up = Ro.c2;
  const int j1 = limb, j2 = 4 + 2 * limb, j3 = 5 + 2 * limb;
  float* ob = sObs + envl * NO; float* st = sSt + envl * 93;
  if (limb == 0) {
    ob[0] = C.s_pos * opos.x; ob[1] = C.s_pos * opos.y; ob[2] = C.s_pos * opos.z;
    ob[3] = up.x; ob[4] = up.y; ob[5] = up.z;
    ob[6] = qf.w; ob[7] = qf.x; ob[8] = qf.y; ob[9] = qf.z;
    ob[10] = C.s_lin * olin.x; ob[11] = C.s_lin * olin.y; ob[12] = C.s_lin * olin.z;
    ob[13] = C.s_ang * oang.x; ob[14] = C.s_ang * oang.y; ob[15] = C.s_ang * oang.z;
    st[0] = C.s_pos * opos.x; st[1] = C.s_pos * opos.y; st[2] = C.s_pos * opos.z;
    st[3] = C.s_lin * olin.x; st[4] = C.s_lin * olin.y; st[5] = C.s_lin * olin.z;
    st[6] = oq.w; st[7] = oq.x; st[8] = oq.y; st[9] = oq.z;
    st[10] = C.s_ang * oang.x; st[11] = C.s_ang * oang.y; st[12] = C.s_ang * oang.z;
    st[37] = S.goal.w; st[38] = S.goal.x; st[39] = S.goal.y; st[40] = S.goal.z;
    st[41] = qf.w; st[42] = qf.x; st[43] = qf.y; st[44] = qf.z;
  }
  const int jj[3] = {j1, j2, j3};
#pragma unroll
  for (int a = 0; a < 3; a++) {
    int j = jj[a];
    ob[16 + j] = C.s_q * I.q[a]; ob[28 + j] = C.s_qd * I.qd[a]; ob[40 + j] = var2 ? 0.3f * I.tgtq[a] : I.act[a]; ob[52 + j] = var2 ? 0.3f * S.ltgt[a] : S.lact[a];      // position-control tasks: targets replace the actions (…position_control.py:438-453)
    st[13 + j] = C.s_q * I.q[a]; st[25 + j] = C.s_qd * I.qd[a]; st[69 + j] = I.act[a]; st[81 + j] = S.lact[a];
    if (var1) { ob[64 + j] = 0.3f * I.tgtq[a]; ob[76 + j] = 0.3f * S.ltgt[a]; }      // :432-455
  }
  st[45 + 3 * limb] = btip.x; st[46 + 3 * limb] = btip.y; st[47 + 3 * limb] = btip.z;
  st[57 + 3 * limb] = S.ltip.x; st[58 + 3 * limb] = S.ltip.y; st[59 + 3 * limb] = S.ltip.z;
  S.ltip = btip;
  // ---- calculate_metrics
  float vn = fminf(sqrtf(qd_.x * qd_.x + qd_.y * qd_.y + qd_.z * qd_.z), 1.0f);
  float rot_dist = 2.0f * asinf(vn);
  float rot_rew = C.quat_scale / (fabsf(rot_dist) + C.rot_eps);
  float trans = sqrtf(opos.x * opos.x + opos.y * opos.y) * C.trans_scale;
  float accp = quad_sum(fabsf(I.acc[0]) * C.acc_scale + fabsf(I.acc[1]) * C.acc_scale + fabsf(I.acc[2]) * C.acc_scale);
  float rate = quad_sum(var1 ? (fabsf(I.act[0]) + fabsf(I.act[1]) + fabsf(I.act[2]))
                             : (fabsf(S.lact[0] - I.act[0]) + fabsf(S.lact[1] - I.act[1]) + fabsf(S.lact[2] - I.act[2]))) * C.rate_scale;
  float powp = 0.f, terr = 0.f, rdec = 0.f;
  if (var1) {      // mechanical power, position-target error, rot-dist-decreasing terms (:530-545)
    powp = quad_sum(fabsf(I.torque[0] * I.qd[0]) + fabsf(I.torque[1] * I.qd[1]) + fabsf(I.torque[2] * I.qd[2])) * P->power_scale;
    terr = quad_sum(fabsf(S.ltgt[0] - I.q[0]) + fabsf(S.ltgt[1] - I.q[1]) + fabsf(S.ltgt[2] - I.q[2])) * P->target_err_scale;
    rdec = ((rot_dist > P->rot_dec_thresh) ? 1.f : 0.f) * (S.lrd - rot_dist) * P->rot_dec_scale;
    S.lrd = rot_dist;
  }
  int cgr = (S.consec > C.max_consec) ? 1 : 0;
  float bonus = C.bonus * (float)cgr;
  int succ = (fabsf(rot_dist) <= C.succ_thresh) ? 1 : 0;
  float dd = fabsf(I.q[2] - I.q[1]);
  int brk = (int)((dd < C.d23_pen[0]) | (dd > C.d23_pen[1])) + (int)((I.q[0] < C.d1_pen[0]) | (I.q[0] > C.d1_pen[1]));
  int rst = (int)((dd < C.d23_rst[0]) | (dd > C.d23_rst[1])) + (int)((I.q[0] < C.d1_rst[0]) | (I.q[0] > C.d1_rst[1]));
  brk = quad_sum_i(brk); rst = quad_sum_i(rst);
  float limp = (brk > 0) ? C.limit_pen : 0.f;
  float total = rot_rew + trans + accp + rate + bonus + limp + powp + terr + rdec;
  S.greset = cgr;
  int both = (succ && S.succ) ? 1 : 0;
  int consec = both ? (S.consec + 1) : 0;
  if (S.succ == 0 && succ == 1) consec = 1;
  S.consec = consec; S.succ = succ;
#pragma unroll
  for (int a = 0; a < 3; a++) S.lact[a] = I.act[a];
  // ---- is_done
  int reset = S.reset;
  if (opos.z > 0.f) reset = 1;
  M3 Rp; V3 pp;
  if (MODE == 0) { Rp.c0 = v3(1, 0, 0); Rp.c1 = v3(0, 1, 0); Rp.c2 = v3(0, 0, 1); pp = v3(0, 0, 0); }
  else { Rp = quat_to_mat(I.fq.w, I.fq.x, I.fq.y, I.fq.z); pp = I.fp; }
  if (mulT(Rp, pr - pp).z <= C.h_base) reset = 1;
  V3 cw = pr + mul(Rr, v3(C.corner[0], C.corner[1], C.corner[2]));
  int nlow = (mulT(Rp, cw - pp).z < C.h_corner) ? 1 : 0;
  nlow += (mulT(Rp, I.knee2 - pp).z - C.h_knee <= 0.f) ? 1 : 0;
  nlow += (mulT(Rp, I.knee3 - pp).z - C.h_knee <= 0.f) ? 1 : 0;
  nlow = quad_sum_i(nlow);
  if (nlow > 0) reset = 1;
  if (rst > 0) reset = 1;
  float fallp = C.fall_pen * (float)reset;
  total += fallp;
  if (cgr == 1) reset = 1;
  if (S.progress >= C.max_episode - 1) reset = 1;
  S.reset = reset;
  O.rew = total;
  O.terms[0] = rot_rew; O.terms[1] = trans; O.terms[2] = accp; O.terms[3] = rate; O.terms[4] = bonus; O.terms[5] = limp; O.terms[6] = fallp; O.terms[7] = (float)cgr;
  O.terms[8] = powp; O.terms[9] = terr; O.terms[10] = rdec;
  if (pd && P->cc_update_last_tgt) { S.ltgt[0] = I.tgtq[0]; S.ltgt[1] = I.tgtq[1]; S.ltgt[2] = I.tgtq[2]; }      // :723-725
}

// Which 16 envs a workgroup takes.  Workgroups go round-robin to the 8 XCDs, each with its own L2, and one row of the SoA state is 64 bytes
// per wavefront: with the identity map the two wavefronts that share a 128-byte line sit on different XCDs and both L2s fetch the whole
// line (FETCH_SIZE showed 1.8x the bytes read; tools/microbench/fetch_calib.hip reproduces the 2x with 64-byte rows).  Within each group
// of 16 workgroups the map puts blocks 2j and 2j+1 of envs on XCD j; the tail of a grid that is no multiple of 16 keeps the identity.
LM_DEV int lm_block() {
  const int b = (int)blockIdx.x;
  return b < ((int)gridDim.x & ~15) ? ((b & ~15) | ((b & 7) << 1) | ((b >> 3) & 1)) : b;
}

// shared tail: write staged obs / states, reward, counters, per-block partial sums
LM_DEV float clampf(float x, float c) { return fminf(fmaxf(x, -c), c); }

struct OutPtrs { float *obs_buf, *states_buf, *rew_buf, *terms; float *out_obs, *out_states, *out_rew; int64_t* out_resets;
                 long long* acc;          // int64 [16 + acc_rows*16]: row 0 = totals of a launch, rows 1.. = the spread first-level rows (write_outputs);
                                          // the persistent rollout (DEFER) points it at one plain row of 14 fixed-point sums per step
                 char* stats; float* extras; float* out_extras; int split_block; int acc_rows; };
#ifndef LM_ACC_COPIES
#define LM_ACC_COPIES 32         // least number of first-level accumulator rows (lm_create doubles it until a row takes < 4096 wavefronts)
#endif
#define ACC_SCALE 1048576.0f      // 2^20: integer accumulation makes the means independent of the arrival order (bitwise reproducible)
// Counted accumulator words (k_step): bits 0..11 count the arrivals, bits 12..63 hold the sum - a signed 2^20 fixed-point value for the
// reward-term means (words 0..6, 9..11), two unsigned 26-bit counts (goal resets | resets) for the success-rate windows (word 7 all envs,
// 8 first task, 12 second task).  The count never carries into the sum: a word sees at most 4095 arrivals.
#define ACC_CNT_BITS 12
#define ACC_CNT_MASK 4095LL
#define ACC_WIN_BITS 26

LM_DEV void success_window(int64_t* ns, float* rate, int64_t add_succ, int64_t add_rst, int64_t max_cnt) {
  int64_t num_succ = ns[0], num_rst = ns[1]; float sr = *rate;
  if (num_rst > max_cnt) { sr = (float)num_succ / (float)num_rst; num_rst = 0; num_succ = 0; }
  ns[0] = num_succ + add_succ; ns[1] = num_rst + add_rst; *rate = sr;
}

// The lane that completed word k of a launch publishes what depends on it: a mean of a reward term, or one success-rate window
// (quadruped_pose_control.py:560,610,618-633; the co-train task keeps two more windows for its halves, joint_locomanipulation.py:795-859).
// The 13 words are independent of each other, so their last arrivals may be lanes of different wavefronts.
LM_DEV void publish_extra(const lm_params* __restrict__ P, const OutPtrs& W, int N, int k, long long tot) {
  if (k == 7 || k == 8 || k == 12) {
    const int w = (k == 7) ? 0 : (k == 8) ? 1 : 2;
    int64_t* ns = reinterpret_cast<int64_t*>(W.stats); float* rate = reinterpret_cast<float*>(W.stats + 48);
    success_window(ns + 2 * w, rate + w, (int64_t)(tot >> ACC_WIN_BITS), (int64_t)(tot & ((1LL << ACC_WIN_BITS) - 1)), (int64_t)P->max_reset_counts);
    W.extras[7 + w] = rate[w]; if (W.out_extras) W.out_extras[7 + w] = rate[w];
  } else {
    const float m = (float)((double)tot * (1.0 / (double)ACC_SCALE)) / (float)N;
    const int e = (k < 7) ? k : k + 1;      // words 9..11 are extras 10..12
    W.extras[e] = m; if (W.out_extras) W.out_extras[e] = m;
  }
}

struct DrOut { int64_t* drc; uint32_t seed, dr_step; int64_t rand_buf, reset_key; uint32_t* sKey; };      // sKey: LDS [16][2] {corr key, fire}

template <int DR, int DEFER = 0, int NOBS = 0>      // DEFER 1: only accumulate; the extras of this step are published later (persistent rollout kernel); NOBS 0: width from the parameters
LM_DEV void write_outputs(const lm_params* __restrict__ P, const OutPtrs& W, int N, int env0, int lane, int limb, int env, bool active,
                          const TaskState& S, const TaskOut& O, int64_t* cnt, int episode, float* sObs, float* sSt, const DrOut& DO) {
  if (DR) {
    // observation-noise bookkeeping with the flags is_done has just written (vec_env_rlgames.py:70-72; randomize.py:213-216,228-230)
    int64_t oc = S.reset ? 0 : DO.drc[0 * (size_t)N + env];
    oc += 1;
    const lm_dr_channel& ci = P->dr[LM_DR_OBS_INTERVAL];
    const bool fire = ci.enabled && oc >= ci.interval;
    if (fire) oc = 0;
    if (limb == 0) { DO.sKey[2 * (lane >> 2)] = (uint32_t)episode + (S.reset ? 1u : 0u); DO.sKey[2 * (lane >> 2) + 1] = fire ? 1u : 0u; }
    if (active && limb == 0) {
      DO.drc[0 * (size_t)N + env] = oc; DO.drc[2 * (size_t)N + env] = (int64_t)DO.dr_step + 1;
      DO.drc[3 * (size_t)N + env] = DO.rand_buf + 1; DO.drc[4 * (size_t)N + env] = DO.reset_key;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS staging writes landed (single wave per block)
  __builtin_amdgcn_wave_barrier();
  // per-block partial sums in a fixed order (deterministic means).  Every lane of a quad holds its env's terms; lane 3 is the one that counts
  float tot0[12];
  {
    const bool cnts = active && (limb == 3);
#pragma unroll
    for (int k = 0; k < 8; k++) tot0[k] = wave_sum_lane3(cnts ? O.terms[k] : 0.f);
    tot0[8] = wave_sum_lane3(cnts ? (float)S.reset : 0.f);
#pragma unroll
    for (int k = 0; k < 3; k++) tot0[9 + k] = wave_sum_lane3(cnts ? O.terms[8 + k] : 0.f);
  }
  // ---- means of the reward terms + success-rate windows.  Every wavefront adds its partial sums to a first-level row (row = block index
  // mod acc_rows, so that 256 wavefronts do not serialise on one cache line) with ONE returning device-scope atomic per word; the word
  // counts its arrivals, so the lane whose add completes a row's word knows it holds the row's total, adds that to the launch's word
  // (row 0) the same way, and the lane that completes that one publishes the extras entry: two dependent round trips on the critical
  // path, no ticket, no read-back, and every word is left zero for the next launch.  Integer sums: the totals do not depend on the order.
  long long acc_old = 0, acc_add = 0;
  long long* acc_row = W.acc;
  {
    const bool first_task = lm_block() < W.split_block;
    float mine = 0.f;
#pragma unroll
    for (int k = 0; k < 12; k++) mine = (lane == k) ? tot0[k] : mine;
    if (DEFER) {      // per-step rows of the persistent rollout kernel: 14 plain fixed-point sums, read by k_rollout_finalize
      mine = (lane == 12) ? (first_task ? tot0[7] : 0.f) : mine;
      mine = (lane == 13) ? (first_task ? tot0[8] : 0.f) : mine;
      if (lane < 14) acc_old = __hip_atomic_fetch_add(acc_row + lane, (long long)llrintf(mine * ACC_SCALE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const long long win = ((long long)llrintf(tot0[7]) << ACC_WIN_BITS) + (long long)llrintf(tot0[8]);      // (goal resets | resets) of these 16 envs
      long long c = (long long)llrintf(mine * ACC_SCALE);
      c = (lane == 7) ? win : c;
      c = (lane == 8) ? (first_task ? win : 0LL) : c;
      c = (lane == 12) ? (first_task ? 0LL : win) : c;
      acc_add = c * (1LL << ACC_CNT_BITS) + 1LL;
      acc_row = W.acc + 16 + (size_t)((int)blockIdx.x & (W.acc_rows - 1)) * 16;
      if (lane < 13) acc_old = __hip_atomic_fetch_add(acc_row + lane, acc_add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  const float clip = P->clip_obs;
  int nenv = min(ENVS_PER_WAVE, N - env0);
  const int NO = NOBS ? NOBS : P->num_obs;
  auto clamp4 = [&](float4 v) { v.x = clampf(v.x, clip); v.y = clampf(v.y, clip); v.z = clampf(v.z, clip); v.w = clampf(v.w, clip); return v; };
  const bool full = (nenv == ENVS_PER_WAVE) && (reinterpret_cast<uintptr_t>(W.out_states) & 15) == 0;
  if (!DR && NOBS && full) {
    // a full wavefront without observation noise: 16 rows of obs (NOBS floats, a multiple of 4) and of states (93 floats; env0 is a multiple
    // of 16: 5952-byte offsets) are contiguous blocks of float4.  Known trip counts: all LDS reads are issued before the first store
    constexpr int NB = NOBS ? NOBS : 64, NV = ENVS_PER_WAVE * NB / 4, NS = ENVS_PER_WAVE * 93 / 4, KV = (NV + 63) / 64, KS = (NS + 63) / 64;
    float4 vo[KV], vs[KS];
#pragma unroll
    for (int k = 0; k < KV; k++) { const int i = lane + 64 * k; vo[k] = reinterpret_cast<const float4*>(sObs)[(NV % 64 == 0 || i < NV) ? i : 0]; }
#pragma unroll
    for (int k = 0; k < KS; k++) { const int i = lane + 64 * k; vs[k] = reinterpret_cast<const float4*>(sSt)[(NS % 64 == 0 || i < NS) ? i : 0]; }
    float4* ob = reinterpret_cast<float4*>(W.obs_buf + (size_t)env0 * NB); float4* oo = reinterpret_cast<float4*>(W.out_obs + (size_t)env0 * NB);
    float4* sb = reinterpret_cast<float4*>(W.states_buf + (size_t)env0 * 93); float4* so = reinterpret_cast<float4*>(W.out_states + (size_t)env0 * 93);
    if (W.out_obs) {
#pragma unroll
      for (int k = 0; k < KV; k++) { const int i = lane + 64 * k; if (NV % 64 == 0 || i < NV) oo[i] = clamp4(vo[k]); }
    }
    if (W.out_states) {
#pragma unroll
      for (int k = 0; k < KS; k++) { const int i = lane + 64 * k; if (NS % 64 == 0 || i < NS) so[i] = clamp4(vs[k]); }
    }
    if (W.obs_buf) {
#pragma unroll
      for (int k = 0; k < KV; k++) { const int i = lane + 64 * k; if (NV % 64 == 0 || i < NV) ob[i] = vo[k]; }
    }
    if (W.states_buf) {
#pragma unroll
      for (int k = 0; k < KS; k++) { const int i = lane + 64 * k; if (NS % 64 == 0 || i < NS) sb[i] = vs[k]; }
    }
  } else {
  // obs: nenv*NO floats contiguous (NO = 64 or 88, both multiples of 4)
  for (int i = lane; i < nenv * (NO / 4); i += 64) {
    float4 v = reinterpret_cast<const float4*>(sObs)[i];
    if (DR) {      // noise in place on obs_buf, then the clipObservations clamp on the returned copy
      const int el = (4 * i) / NO, col = (4 * i) - el * NO;
      const uint32_t ckey = DO.sKey[2 * el], fire = DO.sKey[2 * el + 1], e = (uint32_t)(env0 + el);
      const lm_dr_channel& cr = P->dr[LM_DR_OBS_RESET]; const lm_dr_channel& ci = P->dr[LM_DR_OBS_INTERVAL];
      float x[4] = {v.x, v.y, v.z, v.w};
      const uint32_t cpair = (uint32_t)col >> 1;          // col is a multiple of 4: components (col, col+1) and (col+2, col+3) are Box-Muller pairs
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (cr.enabled) x[k] = dr_apply(cr.operation, x[k], dr_sample(DO.seed, LM_DR_OBS_RESET, e, ckey, 2u * cpair + (uint32_t)k, cr.distribution, cr.p0[0], cr.p1[0]));
        if (fire) x[k] = dr_apply(ci.operation, x[k], dr_sample(DO.seed, LM_DR_OBS_INTERVAL, e, DO.dr_step, 2u * cpair + (uint32_t)k, ci.distribution, ci.p0[0], ci.p1[0]));
      }
      v.x = x[0]; v.y = x[1]; v.z = x[2]; v.w = x[3];
    }
    if (W.obs_buf) reinterpret_cast<float4*>(W.obs_buf + (size_t)env0 * NO)[i] = v;
    if (W.out_obs) reinterpret_cast<float4*>(W.out_obs + (size_t)env0 * NO)[i] = clamp4(v);
  }
  // states: 16 rows of 93 floats are one contiguous block of 372 float4 (env0 is a multiple of 16: 5952-byte offsets)
  if (full) {
    for (int i = lane; i < ENVS_PER_WAVE * 93 / 4; i += 64) {
      float4 v = reinterpret_cast<const float4*>(sSt)[i];
      if (W.states_buf) reinterpret_cast<float4*>(W.states_buf + (size_t)env0 * 93)[i] = v;
      if (W.out_states) reinterpret_cast<float4*>(W.out_states + (size_t)env0 * 93)[i] = clamp4(v);
    }
  } else {
    for (int i = lane; i < nenv * 93; i += 64) {
      float v = sSt[i];
      if (W.states_buf) W.states_buf[(size_t)env0 * 93 + i] = v;
      if (W.out_states) W.out_states[(size_t)env0 * 93 + i] = clampf(v, clip);
    }
  }
  }
  if (active && limb == 0) {
    W.rew_buf[env] = O.rew;
    if (W.out_rew) W.out_rew[env] = O.rew;
    if (W.out_resets) W.out_resets[env] = (int64_t)S.reset;
    cnt[0 * (size_t)N + env] = S.succ; cnt[1 * (size_t)N + env] = S.consec; cnt[2 * (size_t)N + env] = S.greset;
    cnt[3 * (size_t)N + env] = S.reset; cnt[4 * (size_t)N + env] = S.progress; cnt[5 * (size_t)N + env] = episode;
    if (W.terms) {
#pragma unroll
      for (int k = 0; k < 11; k++) W.terms[(size_t)k * N + env] = O.terms[k];
    }
  }
  LM_STAMP(9);      // partial sums, atomics issued, output stores issued
  if (!DEFER) {
    const int rows = W.acc_rows, r = (int)blockIdx.x & (rows - 1);
    const int in_row = ((int)gridDim.x - 1 - r) / rows + 1;      // wavefronts of this launch that add to row r
    if (lane < 13 && (int)(acc_old & ACC_CNT_MASK) == in_row - 1) {
      const long long t1 = (acc_old + acc_add) >> ACC_CNT_BITS;      // the row's total of word `lane`
      __hip_atomic_store(acc_row + lane, 0LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long add2 = t1 * (1LL << ACC_CNT_BITS) + 1LL;
      const long long old2 = __hip_atomic_fetch_add(W.acc + lane, add2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int)(old2 & ACC_CNT_MASK) == min(rows, (int)gridDim.x) - 1) {
        __hip_atomic_store(W.acc + lane, 0LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        publish_extra(P, W, N, lane, (old2 + add2) >> ACC_CNT_BITS);
      }
    }
  } else {
    asm volatile("" :: "v"(acc_old));
  }
}

LM_DEV void load_table(const float* __restrict__ table, float* sTab, int lane) {
  for (int i = lane; i < LM_ITAB_FLOATS; i += 64) sTab[i] = table[i];
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
}
// the same in two halves so that the table's round trip overlaps the state loads of the step kernel
#define TABLE_REGS ((LM_ITAB_FLOATS + 63) / 64)
struct TableRegs { float v[TABLE_REGS]; };
LM_DEV void table_fetch(const float* __restrict__ table, int lane, TableRegs& T) {
#pragma unroll
  for (int j = 0; j < TABLE_REGS; j++) { int i = lane + 64 * j; T.v[j] = (i < LM_ITAB_FLOATS) ? table[i] : 0.f; }
}
LM_DEV void table_commit(const TableRegs& T, float* sTab, int lane) {
#pragma unroll
  for (int j = 0; j < TABLE_REGS; j++) { int i = lane + 64 * j; if (i < LM_ITAB_FLOATS) sTab[i] = T.v[j]; }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
struct StepArgs {
  const lm_params* params; const float* table; float* state; int64_t* cnt;
  const float* actions; const float* goal_rand; OutPtrs W; int N, split; uint32_t seed;
  int skip_reset;   // 1: leave reset_buf untouched (staged API: resets were applied by lm_apply_resets)
  int nsub;         // < 0: params.substeps, otherwise that many sub-steps (0 = read-back + task layer only)
  int64_t* drc;     // domain-randomisation counters [LM_DR_CNT_ROWS][N] (k_step_dr only)
  float* dr_phys;   // [LM_DR_PHYS_ROWS][N] attributes sampled for this step (k_step_dr only)
  int kind[2];      // variant * 2 + (mode == LM_MODE_MANI) of the two parameter blocks: the kernels pick their specialisation from the kernel
                    // arguments, so the first loads of the step do not wait for a round trip to the parameter block
};

template <int MODE, int VAR, int DR, int DEFER = 0>
LM_DEV void step_body(const StepArgs& A, const lm_params* __restrict__ P, float* sTab, float* sObs, float* sSt, float4* sStash) {
  TableRegs TR; table_fetch(A.table, threadIdx.x, TR);
  const int lane = threadIdx.x, limb = lane & 3, envl = lane >> 2;
  const int env0 = lm_block() * ENVS_PER_WAVE, envr = env0 + envl, N = A.N;
  const bool active = envr < N; const int env = active ? envr : (N - 1);
  const float* tl = sTab + HUB_FLOATS + limb * LIMB_STRIDE;
  const int jj[3] = {limb, 4 + 2 * limb, 5 + 2 * limb};
  float* st = A.state; int64_t* cnt = A.cnt;
  const int fb = (MODE == 0) ? R_FB0 : R_FB1;
  Stash St; St.base = sStash; St.lane = lane;
  // ---- load the physical state (the task-layer state is loaded after the physics to keep registers free)
  const bool do_reset = (cnt[3 * (size_t)N + env] != 0) && !A.skip_reset;
  FreeBody F; V3 lin, ang; float q[3], qd[3], act[3];
  F.p = v3(st[(size_t)(fb + 0) * N + env], st[(size_t)(fb + 1) * N + env], st[(size_t)(fb + 2) * N + env]);
  F.q.w = st[(size_t)(fb + 3) * N + env]; F.q.x = st[(size_t)(fb + 4) * N + env]; F.q.y = st[(size_t)(fb + 5) * N + env]; F.q.z = st[(size_t)(fb + 6) * N + env];
  lin = v3(st[(size_t)(fb + 7) * N + env], st[(size_t)(fb + 8) * N + env], st[(size_t)(fb + 9) * N + env]);
  ang = v3(st[(size_t)(fb + 10) * N + env], st[(size_t)(fb + 11) * N + env], st[(size_t)(fb + 12) * N + env]);
#pragma unroll
  for (int a = 0; a < 3; a++) {
    q[a] = st[(size_t)(R_Q + jj[a]) * N + env]; qd[a] = st[(size_t)(R_QD + jj[a]) * N + env];
    act[a] = A.actions[(size_t)env * 12 + jj[a]];
  }
  DrPhys X; uint32_t dr_step = 0; int64_t dr_rand_buf = 0, dr_reset_key = 0;
  if (DR) {
    // ---- action noise on the raw actions (vec_env_rlgames.py:56-58; randomize.py:237-259): correlated noise keyed by the episode this
    // step belongs to (redrawn exactly when the reset flag is set), uncorrelated noise every frequency_interval calls
    int64_t* dc = A.drc;
    const uint32_t ep_now = (uint32_t)cnt[5 * (size_t)N + env] + (do_reset ? 1u : 0u);
    dr_step = (uint32_t)dc[2 * (size_t)N + env]; dr_rand_buf = dc[3 * (size_t)N + env]; dr_reset_key = dc[4 * (size_t)N + env];
    int64_t ac = do_reset ? 0 : dc[1 * (size_t)N + env];
    ac += 1;
    const lm_dr_channel& cr = P->dr[LM_DR_ACT_RESET]; const lm_dr_channel& ci = P->dr[LM_DR_ACT_INTERVAL];
    const bool fire = ci.enabled && ac >= ci.interval;
    if (fire) ac = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      if (cr.enabled) act[a] = dr_apply(cr.operation, act[a], dr_sample(A.seed, LM_DR_ACT_RESET, (uint32_t)env, ep_now, (uint32_t)jj[a], cr.distribution, cr.p0[0], cr.p1[0]));
      if (fire) act[a] = dr_apply(ci.operation, act[a], dr_sample(A.seed, LM_DR_ACT_INTERVAL, (uint32_t)env, dr_step, (uint32_t)jj[a], ci.distribution, ci.p0[0], ci.p1[0]));
    }
    if (active && limb == 0) dc[1 * (size_t)N + env] = ac;
    // ---- gated on_reset randomisation (quadruped_pose_control.py:224-228), then this control step's physics attributes
    if (do_reset && dr_rand_buf >= P->dr_min_frequency) { dr_reset_key = ep_now; dr_rand_buf = 0; }
    const float g0[3] = {0.f, 0.f, -P->gravity}; float gv[3], fv[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      gv[c] = dr_attr(P->dr[LM_DR_GRAVITY], A.seed, LM_DR_GRAVITY, env, dr_step, (uint32_t)dr_reset_key, c, c, g0[c]);
      fv[c] = dr_attr(P->dr[LM_DR_BASE_FORCE], A.seed, LM_DR_BASE_FORCE, env, dr_step, (uint32_t)dr_reset_key, c, c, 0.f);
      X.tmax[c] = dr_attr(P->dr[LM_DR_MAX_EFFORT], A.seed, LM_DR_MAX_EFFORT, env, dr_step, (uint32_t)dr_reset_key, jj[c], 0, P->tau_max);
      X.vmax[c] = dr_attr(P->dr[LM_DR_MAX_VELOCITY], A.seed, LM_DR_MAX_VELOCITY, env, dr_step, (uint32_t)dr_reset_key, jj[c], 0, P->max_joint_vel);
      X.cj[c] = dr_attr(P->dr[LM_DR_JOINT_DAMPING], A.seed, LM_DR_JOINT_DAMPING, env, dr_step, (uint32_t)dr_reset_key, jj[c], 0, P->joint_damping);
    }
    X.g = v3(gv[0], gv[1], gv[2]); X.f = v3(fv[0], fv[1], fv[2]);
    if (active) {
      float* ph = A.dr_phys;
#pragma unroll
      for (int c = 0; c < 3; c++) { ph[(size_t)jj[c] * N + env] = X.tmax[c]; ph[(size_t)(12 + jj[c]) * N + env] = X.vmax[c]; ph[(size_t)(30 + jj[c]) * N + env] = X.cj[c]; }
      if (limb == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { ph[(size_t)(24 + c) * N + env] = gv[c]; ph[(size_t)(27 + c) * N + env] = fv[c]; }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) act[a] = clampf(act[a], P->clip_actions);
  // ---- reset_idx (quadruped_pose_control.py:230-299), physical part
  if (do_reset) {
#pragma unroll
    for (int a = 0; a < 3; a++) { q[a] = P->init_q[jj[a]]; qd[a] = 0.f; }
    const float* ip = (MODE == 0) ? P->init_base_pos : P->init_plate_pos; const float* iq = (MODE == 0) ? P->init_base_quat : P->init_plate_quat;
    F.p = v3(ip[0], ip[1], ip[2]); F.q.w = iq[0]; F.q.x = iq[1]; F.q.y = iq[2]; F.q.z = iq[3];
    lin = v3(0, 0, 0); ang = v3(0, 0, 0);
  }
  // world -> body-coordinate twist
  {
    M3 R = quat_to_mat(F.q.w, F.q.x, F.q.y, F.q.z);
    F.u = sv(mulT(R, ang), mulT(R, lin));
  }
  M3 Rfix; V3 pfix = v3(P->fixed_base_pos[0], P->fixed_base_pos[1], P->fixed_base_pos[2]);
  Rfix = quat_to_mat(P->fixed_base_quat[0], P->fixed_base_quat[1], P->fixed_base_quat[2], P->fixed_base_quat[3]);
  table_commit(TR, sTab, lane);
  LM_STAMP(0);
  float tau_acc[3] = {0.f, 0.f, 0.f}, tgtq[3] = {0.f, 0.f, 0.f}, qda[3] = {0.f, 0.f, 0.f}; bool qda_set = false;
  constexpr bool pd = (VAR >= 1);
  const int nsub = (A.nsub < 0) ? P->substeps : A.nsub;
  if (!pd) {
    // ---- take_action (robot.py:452-454): velocity targets
    // velocity mode (every task of the path): the drive's velocity target; effort mode: the torque; position mode (robot.py:448-450):
    // q* = a * act_scale, tau = kp (q* - q) - kd qd = kd (v* - qd) with v* = kp / kd (q* - q), re-evaluated every sub-step
    const float a0[3] = {act[0] * P->act_scale, act[1] * P->act_scale, act[2] * P->act_scale};
    const bool posm = P->drive_mode == LM_DRIVE_POSITION; const float gp = posm ? P->pd_kp / P->kd : 0.f;
    for (int s = 0; s < nsub; s++) {
      const float tgt[3] = {posm ? gp * (a0[0] - q[0]) : a0[0], posm ? gp * (a0[1] - q[1]) : a0[1], posm ? gp * (a0[2] - q[2]) : a0[2]};
      substep<MODE, VAR, DR>(P, sTab, tl, limb, St, F, Rfix, pfix, q, qd, tgt, tau_acc, X);
    }
  } else {
    // ---- custom-controller tasks (quadruped_pose_control_custom_controller.py:255-307): the action integrates the swing / extension
    // position targets; the actuator torque  clamp(kp (q* - q) - kd qd, +-tau_max)  is re-evaluated every sub-step.  It is the same drive
    // as above with damping gain kd and the position-derived velocity target  v* = kp / kd (q* - q)  (implicit in qd, 2-pass clamp).
    float se[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      float v = do_reset ? P->init_se[jj[a]] : st[(size_t)(R_SE + jj[a]) * N + env];
      if (!A.skip_reset || A.nsub != 0) v = fminf(fmaxf(v + act[a] * P->act_scale_se, P->se_lo[jj[a]]), P->se_hi[jj[a]]);
      se[a] = v;
      if (active) st[(size_t)(R_SE + jj[a]) * N + env] = v;
    }
    tgtq[0] = se[0]; tgtq[1] = se[1] + 0.5f * se[2]; tgtq[2] = se[1] - 0.5f * se[2];      // dof1, dof2 = swing + ext/2, dof3 = swing - ext/2
    const float g = P->pd_kp / P->kd;
    for (int s = 0; s < nsub; s++) {
      // update_joint_states() runs after every in-task sub-step (…custom_controller.py:296-297): the joint acceleration spans only the
      // trailing acc_substeps (= controlFrequencyInv) sub-steps (robot.py:289-291)
      if (s == nsub - P->acc_substeps) { qda[0] = qd[0]; qda[1] = qd[1]; qda[2] = qd[2]; qda_set = true; }
      float tgt[3] = {g * (tgtq[0] - q[0]), g * (tgtq[1] - q[1]), g * (tgtq[2] - q[2])};
      substep<MODE, VAR, DR>(P, sTab, tl, limb, St, F, Rfix, pfix, q, qd, tgt, tau_acc, X);
    }
  }
  // ---- task-layer state (loaded after the physics: issuing these loads at the top of the last sub-step was measured and gains nothing)
  TaskState S; int episode; float lqd[3];
  S.succ = (int)cnt[0 * (size_t)N + env]; S.consec = (int)cnt[1 * (size_t)N + env]; S.greset = (int)cnt[2 * (size_t)N + env];
  S.reset = (int)cnt[3 * (size_t)N + env]; S.progress = (int)cnt[4 * (size_t)N + env]; episode = (int)cnt[5 * (size_t)N + env];
#pragma unroll
  for (int a = 0; a < 3; a++) { S.lact[a] = st[(size_t)(R_LACT + jj[a]) * N + env]; lqd[a] = st[(size_t)(R_LQD + jj[a]) * N + env]; }
  S.ltip = v3(st[(size_t)(R_LTIP + 3 * limb) * N + env], st[(size_t)(R_LTIP + 3 * limb + 1) * N + env], st[(size_t)(R_LTIP + 3 * limb + 2) * N + env]);
  S.goal.w = st[(size_t)(R_GOAL + 0) * N + env]; S.goal.x = st[(size_t)(R_GOAL + 1) * N + env]; S.goal.y = st[(size_t)(R_GOAL + 2) * N + env]; S.goal.z = st[(size_t)(R_GOAL + 3) * N + env];
  S.lrd = 0.f; S.ltgt[0] = S.ltgt[1] = S.ltgt[2] = 0.f;
  if (pd) {
#pragma unroll
    for (int a = 0; a < 3; a++) S.ltgt[a] = st[(size_t)(R_LTGT + jj[a]) * N + env];
    S.lrd = st[(size_t)R_LRD * N + env];
  }
  if (do_reset) {
    float u3[3];
    if (A.goal_rand) { u3[0] = A.goal_rand[(size_t)env * 3]; u3[1] = A.goal_rand[(size_t)env * 3 + 1]; u3[2] = A.goal_rand[(size_t)env * 3 + 2]; }
    else hash_uniform3(A.seed, (uint32_t)env, (uint32_t)episode, u3);
    S.goal = quat_from_euler(P->goal_lo[0] + (P->goal_hi[0] - P->goal_lo[0]) * u3[0], P->goal_lo[1] + (P->goal_hi[1] - P->goal_lo[1]) * u3[1],
                             P->goal_lo[2] + (P->goal_hi[2] - P->goal_lo[2]) * u3[2]);
#pragma unroll
    for (int a = 0; a < 3; a++) { S.lact[a] = 0.f; lqd[a] = 0.f; }
    S.ltip = v3(P->default_tip[3 * limb], P->default_tip[3 * limb + 1], P->default_tip[3 * limb + 2]);
    S.succ = 0; S.consec = 0; S.greset = 0; S.reset = 0; S.progress = 0; episode += 1;
    if (pd) {      // :371-384
#pragma unroll
      for (int a = 0; a < 3; a++) S.ltgt[a] = P->init_q[jj[a]];
      Q4 qb; qb.w = (MODE == 0) ? P->init_base_quat[0] : 1.f; qb.x = (MODE == 0) ? -P->init_base_quat[1] : 0.f;
      qb.y = (MODE == 0) ? -P->init_base_quat[2] : 0.f; qb.z = (MODE == 0) ? -P->init_base_quat[3] : 0.f;
      Q4 d4 = qmul(qb, qconj(S.goal));
      S.lrd = 2.0f * asinf(fminf(sqrtf(d4.x * d4.x + d4.y * d4.y + d4.z * d4.z), 1.0f));
    }
  }
  // ---- blow-up guard: the reference only prints NaNs and asserts (quadruped_pose_control.py:550-558); here a non-finite or
  // exploding state is replaced by the reset pose and the env is flagged for reset, so one bad env cannot poison a batch
  int blown = 0;
  {
    float chk = F.p.x + F.p.y + F.p.z + F.q.w + F.q.x + F.q.y + F.q.z + F.u.w.x + F.u.w.y + F.u.w.z + F.u.v.x + F.u.v.y + F.u.v.z
              + q[0] + q[1] + q[2] + qd[0] + qd[1] + qd[2];
    float big = fmaxf(fmaxf(fabsf(qd[0]), fabsf(qd[1])), fmaxf(fabsf(qd[2]), fabsf(F.u.v.x) + fabsf(F.u.v.y) + fabsf(F.u.v.z)));
    blown = quad_sum_i((!(fabsf(chk) < 1.0e30f) || big > 1.0e4f) ? 1 : 0);
    if (blown) {
#pragma unroll
      for (int a = 0; a < 3; a++) { q[a] = P->init_q[jj[a]]; qd[a] = 0.f; }
      const float* ip = (MODE == 0) ? P->init_base_pos : P->init_plate_pos; const float* iq = (MODE == 0) ? P->init_base_quat : P->init_plate_quat;
      F.p = v3(ip[0], ip[1], ip[2]); F.q.w = iq[0]; F.q.x = iq[1]; F.q.y = iq[2]; F.q.z = iq[3];
      F.u = sv(v3(0, 0, 0), v3(0, 0, 0));
      if (active && limb == 0) atomicAdd(reinterpret_cast<unsigned int*>(A.W.stats + 60), 1u);      // contained blow-ups since creation (LM_PTR_STATS)
    }
  }
  LM_STAMP(6);      // task-state loads, reset scatter, blow-up guard
  // ---- read-back (robot.py:276-321)
  TaskIn I;
  M3 Rf = quat_to_mat(F.q.w, F.q.x, F.q.y, F.q.z);
  I.fp = F.p; I.fq = F.q; I.lin = mul(Rf, F.u.v); I.ang = mul(Rf, F.u.w);
  {
    LimbKin K; float z3[3] = {0.f, 0.f, 0.f};
    limb_kinematics(tl, q, z3, K);
    V3 x, k2, k3; limb_points(tl, K, x, k2, k3);
    M3 Rb = (MODE == 0) ? Rf : Rfix; V3 pb = (MODE == 0) ? F.p : pfix;
    I.tipw = pb + mul(Rb, x); I.knee2 = pb + mul(Rb, k2); I.knee3 = pb + mul(Rb, k3);
  }
  const float acc_dt_inv = P->acc_dt_inv, ctrl_dt_inv = P->ctrl_dt_inv, torque_div = pd ? P->torque_div : 1.f;      // one batch of loads (see task_eval)
#pragma unroll
  for (int a = 0; a < 3; a++) { I.q[a] = q[a]; I.qd[a] = qd[a]; I.acc[a] = (pd && qda_set) ? (qd[a] - qda[a]) * acc_dt_inv : (qd[a] - lqd[a]) * ctrl_dt_inv; I.act[a] = act[a];
    I.torque[a] = pd ? tau_acc[a] / torque_div : 0.f; I.tgtq[a] = tgtq[a]; }      // logged torque = sum over sub-steps / control_decimal (:307)
  TaskOut O;
  task_eval<MODE, VAR>(P, limb, envl, I, S, O, sObs, sSt);
  LM_STAMP(7);      // read-back kinematics + task layer
  if (blown) S.reset = 1;
  // ---- store state
  if (active) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      st[(size_t)(R_Q + jj[a]) * N + env] = q[a]; st[(size_t)(R_QD + jj[a]) * N + env] = qd[a];
      st[(size_t)(R_LACT + jj[a]) * N + env] = S.lact[a]; st[(size_t)(R_LQD + jj[a]) * N + env] = qd[a];
    }
    st[(size_t)(R_LTIP + 3 * limb) * N + env] = S.ltip.x; st[(size_t)(R_LTIP + 3 * limb + 1) * N + env] = S.ltip.y; st[(size_t)(R_LTIP + 3 * limb + 2) * N + env] = S.ltip.z;
    if (pd) {
#pragma unroll
      for (int a = 0; a < 3; a++) st[(size_t)(R_LTGT + jj[a]) * N + env] = S.ltgt[a];
      if (limb == 0) st[(size_t)R_LRD * N + env] = S.lrd;
    }
    if (limb == 0) {
      st[(size_t)(fb + 0) * N + env] = F.p.x; st[(size_t)(fb + 1) * N + env] = F.p.y; st[(size_t)(fb + 2) * N + env] = F.p.z;
      st[(size_t)(fb + 3) * N + env] = F.q.w; st[(size_t)(fb + 4) * N + env] = F.q.x; st[(size_t)(fb + 5) * N + env] = F.q.y; st[(size_t)(fb + 6) * N + env] = F.q.z;
      st[(size_t)(fb + 7) * N + env] = I.lin.x; st[(size_t)(fb + 8) * N + env] = I.lin.y; st[(size_t)(fb + 9) * N + env] = I.lin.z;
      st[(size_t)(fb + 10) * N + env] = I.ang.x; st[(size_t)(fb + 11) * N + env] = I.ang.y; st[(size_t)(fb + 12) * N + env] = I.ang.z;
      st[(size_t)(R_GOAL + 0) * N + env] = S.goal.w; st[(size_t)(R_GOAL + 1) * N + env] = S.goal.x; st[(size_t)(R_GOAL + 2) * N + env] = S.goal.y; st[(size_t)(R_GOAL + 3) * N + env] = S.goal.z;
    }
  }
  DrOut DO; DO.drc = A.drc; DO.seed = A.seed; DO.dr_step = dr_step; DO.rand_buf = dr_rand_buf; DO.reset_key = dr_reset_key;
#ifdef LM_WAVES2
  DO.sKey = reinterpret_cast<uint32_t*>(sSt + ENVS_PER_WAVE * 93);      // behind the output staging, which lives in the stash's memory in this build
#else
  DO.sKey = reinterpret_cast<uint32_t*>(sStash);      // the stash is dead after the last sub-step
#endif
  LM_STAMP(8);      // state stores issued
  write_outputs<DR, DEFER, (VAR == 1) ? LM_MAX_OBS : 64>(P, A.W, N, env0, lane, limb, env, active, S, O, cnt, episode, sObs, sSt, DO);
  LM_STAMP(10);     // the reduction's round trips
}

// The step kernels.  One launch per step(); a wavefront picks its specialisation (task mode x actuator family) from the kernel arguments.  The
// velocity-drive tasks (k_step: the headline) and the PD-actuator families (k_step_pd) are separate kernels, so that the register allocation
// and code layout of the one do not move when the other is edited; both blocks of a co-training engine are of one actuator family (lm_create).
#ifdef LM_STAMPS
#define LM_STEP_PROLOGUE \
  if (threadIdx.x < 64) lm_stamp_lds[threadIdx.x] = 0; \
  __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); \
  { unsigned long long t0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_) :: "memory"); if ((threadIdx.x & 63) == 0) lm_stamp_lds[16 * (threadIdx.x >> 6) + 15] = t0_; } \
  LM_STAMP(11); LM_STAMP(12);      /* two stamps back to back: bucket 12 = the cost of a stamp */ \
  const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime(), mt0_ = __builtin_amdgcn_s_memtime();
#define LM_STEP_EPILOGUE \
  { const unsigned long long rt1_ = __builtin_amdgcn_s_memrealtime(), mt1_ = __builtin_amdgcn_s_memtime(); \
    if (threadIdx.x == 0) { lm_stamp_lds[13] = mt1_ - mt0_; lm_stamp_lds[14] = rt1_ - rt0_; } } \
  __builtin_amdgcn_s_waitcnt(0xc07f); \
  if (threadIdx.x < 64 && blockIdx.x < 1024) lm_stamp_out[blockIdx.x * 64 + threadIdx.x] = lm_stamp_lds[threadIdx.x];
#else
#define LM_STEP_PROLOGUE
#define LM_STEP_EPILOGUE
#endif
#ifdef LM_WAVES2
#define LM_STEP_SMEM(NOBS) \
  __shared__ __attribute__((aligned(16))) float sTab[LM_ITAB_FLOATS + 2]; \
  __shared__ float4 sStash[STASH_SLOTS * 64]; \
  static_assert(ENVS_PER_WAVE * ((NOBS) + 93 + 3) * 4 <= STASH_SLOTS * 64 * 16, "output staging must fit in the stash"); \
  float* sObs = reinterpret_cast<float*>(sStash); float* sSt = sObs + ENVS_PER_WAVE * (NOBS); \
  const int env0 = lm_block() * ENVS_PER_WAVE; \
  const lm_params* P = A.params + ((env0 >= A.split) ? 1 : 0); \
  const int kind = A.kind[(env0 >= A.split) ? 1 : 0];
#else
#define LM_STEP_SMEM(NOBS) \
  __shared__ __attribute__((aligned(16))) float sTab[LM_ITAB_FLOATS + 2]; \
  __shared__ __attribute__((aligned(16))) float sObs[ENVS_PER_WAVE * (NOBS)]; \
  __shared__ __attribute__((aligned(16))) float sSt[ENVS_PER_WAVE * 93]; \
  __shared__ float4 sStash[STASH_SLOTS * 64]; \
  const int env0 = lm_block() * ENVS_PER_WAVE; \
  const lm_params* P = A.params + ((env0 >= A.split) ? 1 : 0); \
  const int kind = A.kind[(env0 >= A.split) ? 1 : 0];
#endif

#ifdef LM_W2_UNIT
#define k_step k_step_w2
#endif
__global__ void __launch_bounds__(64) LM_STEP_ATTR k_step(StepArgs A) {                 // velocity-drive tasks (kinds 0, 1)
  LM_STEP_SMEM(64)
  LM_STEP_PROLOGUE
#ifdef LM_W2_UNIT
  // locomotion only: the plate specialisation does not live in 256 registers (measured with both in this kernel: 300 against 533 M env-steps/s on
  // the manipulation task at 65 536 envs), so manipulation and co-training engines stay on the one-wavefront kernel at every size
  step_body<0, 0, 0>(A, P, sTab, sObs, sSt, sStash);
#else
  if (kind == 0) step_body<0, 0, 0>(A, P, sTab, sObs, sSt, sStash); else step_body<1, 0, 0>(A, P, sTab, sObs, sSt, sStash);
#endif
  LM_STEP_EPILOGUE
}
#ifdef LM_W2_UNIT
#undef k_step
// the second translation unit (lm_engine_w2.hip) ends here: the kernel above and its launcher
extern "C" __attribute__((visibility("hidden"))) void lm_internal_launch_step_w2(const StepArgs* A, int nblocks, hipStream_t s) {
  hipLaunchKernelGGL(k_step_w2, dim3(nblocks), dim3(64), 0, s, *A);
}
#else
extern "C" __attribute__((visibility("hidden"))) void lm_internal_launch_step_w2(const StepArgs* A, int nblocks, hipStream_t s);      // lm_engine_w2.hip

__global__ void __launch_bounds__(64) k_step_pd(StepArgs A) {              // PD-actuator families (kinds 2 ... 5)
  LM_STEP_SMEM(LM_MAX_OBS)
  LM_STEP_PROLOGUE
  if (kind == 2) step_body<0, 1, 0>(A, P, sTab, sObs, sSt, sStash); else if (kind == 3) step_body<1, 1, 0>(A, P, sTab, sObs, sSt, sStash);
  else if (kind == 4) step_body<0, 2, 0>(A, P, sTab, sObs, sSt, sStash); else step_body<1, 2, 0>(A, P, sTab, sObs, sSt, sStash);
  LM_STEP_EPILOGUE
}

// the same steps with domain randomisation (separate kernels so that the un-randomised ones above are untouched)
__global__ void __launch_bounds__(64) k_step_dr(StepArgs A) {
  LM_STEP_SMEM(64)
  if (kind == 0) step_body<0, 0, 1>(A, P, sTab, sObs, sSt, sStash); else step_body<1, 0, 1>(A, P, sTab, sObs, sSt, sStash);
}

__global__ void __launch_bounds__(64) k_step_dr_pd(StepArgs A) {
  LM_STEP_SMEM(LM_MAX_OBS)
  if (kind == 2) step_body<0, 1, 1>(A, P, sTab, sObs, sSt, sStash); else if (kind == 3) step_body<1, 1, 1>(A, P, sTab, sObs, sSt, sStash);
  else if (kind == 4) step_body<0, 2, 1>(A, P, sTab, sObs, sSt, sStash); else step_body<1, 2, 1>(A, P, sTab, sObs, sSt, sStash);
}


// ---- persistent rollout: T x (policy forward -> action sampling -> physics step) + the bootstrap forward without leaving
// the kernel.  A block owns its 16 envs for the whole rollout: wavefront 0 runs the step exactly as k_step does (same step_body), the
// observations it stages in LDS feed the next forward directly, and all four wavefronts run the policy tile (MLP or GNN, lm_policy_dev.h).  Blocks
// never wait for each other, so a step costs a block its own time rather than the slowest block's, and no launch boundary is paid.
// The per-step reductions go to per-step accumulators (blocks drift apart); k_rollout_finalize publishes the extras afterwards, in
// step order, with the same arithmetic as the last-arriver path of write_outputs.
struct RolloutDev {
  const float* params; const float* log_std;
  float *obs, *actions, *logp, *values, *rewards; int64_t* dones;
  long long* acc_steps; int T; uint32_t noise_seed;
};

// the step of the persistent kernel as a real call: its ~350 registers are then allocated separately from the policy tile's
LM_DEV void step_dispatch(const StepArgs& B, const lm_params* P, float* sTab, float* sObs, float* sSt, float4* sStash) {
  if (P->variant == 0) { if (P->mode == LM_MODE_LOCO) step_body<0, 0, 0, 1>(B, P, sTab, sObs, sSt, sStash); else step_body<1, 0, 0, 1>(B, P, sTab, sObs, sSt, sStash); }
  else if (P->variant == 1) { if (P->mode == LM_MODE_LOCO) step_body<0, 1, 0, 1>(B, P, sTab, sObs, sSt, sStash); else step_body<1, 1, 0, 1>(B, P, sTab, sObs, sSt, sStash); }
  else { if (P->mode == LM_MODE_LOCO) step_body<0, 2, 0, 1>(B, P, sTab, sObs, sSt, sStash); else step_body<1, 2, 0, 1>(B, P, sTab, sObs, sSt, sStash); }
}

template <int NOBS, int POLICY> struct PolicySmem { MlpSmem<NOBS> M; };
template <int NOBS> struct PolicySmem<NOBS, LM_POLICY_GNN> { GnnSmem M; };

template <int NOBS, int POLICY>
__global__ void __launch_bounds__(256) k_rollout(StepArgs A, RolloutDev R) {
  __shared__ __attribute__((aligned(16))) float sTab[LM_ITAB_FLOATS + 2];
  __shared__ __attribute__((aligned(16))) float sObs[ENVS_PER_WAVE * LM_MAX_OBS];
  __shared__ __attribute__((aligned(16))) float sSt[ENVS_PER_WAVE * 93];
  __shared__ float4 sStash[STASH_SLOTS * 64];
  __shared__ PolicySmem<NOBS, POLICY> PS;
  const int t = threadIdx.x, env0 = lm_block() * ENVS_PER_WAVE;
  const lm_params* P = A.params + ((env0 >= A.split) ? 1 : 0);
#ifdef LM_STAMPS
  if (threadIdx.x < 64) lm_stamp_lds[threadIdx.x] = 0;
  __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
  { unsigned long long t0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_) :: "memory"); if ((threadIdx.x & 63) == 0) lm_stamp_lds[16 * (threadIdx.x >> 6) + 15] = t0_; }
  __syncthreads();
#endif
  for (int k = 0; k <= R.T; k++) {
    // The loop body must be compiled like a stand-alone kernel: without these opaque copies the compiler hoists every loop-invariant
    // address (one 64-bit pointer per state row and per weight chunk) out of the loop and spills hundreds of registers to scratch.
    int z = 0; asm volatile("" : "+s"(z));      // an opaque zero, new in every iteration
    StepArgs B = A; B.state = A.state + z; B.cnt = A.cnt + z; B.N = A.N + z;
    const float* Wk = R.params + z; const lm_params* Pk = P + z;
    const size_t Nk = (size_t)B.N;
    SampleArgs SA{};
    if (k < R.T) { SA.log_std = R.log_std; SA.cnt = B.cnt; SA.seed = R.noise_seed; SA.actions = R.actions + (size_t)k * Nk * 12; SA.logp = R.logp + (size_t)k * Nk; }
    if (POLICY == LM_POLICY_GNN) {
      if (k == 0) gnn_block<false>(R.obs, 0.f, B.N, env0, Wk, nullptr, R.values, SA, reinterpret_cast<GnnSmem&>(PS.M), t);
      else gnn_block<true>(sObs, Pk->clip_obs, B.N, env0, Wk, nullptr, R.values + (size_t)k * Nk, SA, reinterpret_cast<GnnSmem&>(PS.M), t);
    } else {
      if (k == 0) mlp_block<NOBS, false>(R.obs, 0.f, B.N, env0, Wk, nullptr, R.values, SA, reinterpret_cast<MlpSmem<NOBS>&>(PS.M), t);
      else mlp_block<NOBS, true>(sObs, Pk->clip_obs, B.N, env0, Wk, nullptr, R.values + (size_t)k * Nk, SA, reinterpret_cast<MlpSmem<NOBS>&>(PS.M), t);
    }
    if (t < 64) LM_STAMP(13);      // diagnostic build: the policy tile as wavefront 0 sees it
    if (k == R.T) break;
    if (t < 64) {
      // (the counters and the state are written and read back by this same wavefront: program order.  The sampled actions too with the MLP;
      // the GNN tile stores them from all four wavefronts and drains those stores before its closing barrier, lm_policy_dev.h gnn_body)
      B.actions = SA.actions; B.goal_rand = nullptr;
      B.W.out_obs = R.obs + (size_t)(k + 1) * Nk * NOBS; B.W.out_states = nullptr; B.W.out_rew = R.rewards + (size_t)k * Nk;
      B.W.out_resets = R.dones + (size_t)k * Nk; B.W.out_extras = nullptr; B.W.acc = R.acc_steps + 16 * k;
      step_dispatch(B, Pk, sTab, sObs, sSt, sStash);
    }
    lds_barrier();          // the observations staged in LDS are visible to the other wavefronts; global data is private to wavefront 0
    if (t < 64) LM_STAMP(14);
  }
#ifdef LM_STAMPS
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x < 1024) lm_stamp_out[blockIdx.x * 64 + threadIdx.x] = lm_stamp_lds[threadIdx.x];
#endif
}

// ---- persistent rollout with the MLP policy: wavefront-specialised.  Wavefront 0 steps the physics; wavefronts 1..3 hold the policy's weights
// in registers for the whole rollout and run the forward + sampling between two physics steps (lm_policy_dev.h mlp_res_tile).  The two roles
// are separate loops (separate live ranges: the 300 weight registers of a policy wavefront never meet the physics' 440) that meet at block
// barriers: MLP_RES_BARRIERS inside a forward, one after the physics step (observations staged in LDS).
template <int NOBS, int P>
LM_DEV void rollout_policy_loop(const StepArgs& A, const RolloutDev& R, const lm_params* P_, float* sObs, MlpSmem<NOBS>& M, int env0, int tp) {
  MlpResRegs<NOBS, P> RG; RG.load(R.params, tp & 63, (tp & 63) >> 4);
  const size_t N = (size_t)A.N;
  const float clip_obs = P_->clip_obs;
  for (int k = 0; k <= R.T; k++) {
    SampleArgs SA{};
    if (k < R.T) { SA.log_std = R.log_std; SA.cnt = A.cnt; SA.seed = R.noise_seed; SA.actions = R.actions + (size_t)k * N * 12; SA.logp = R.logp + (size_t)k * N; }
    if (k == 0) mlp_res_tile<NOBS, P, false>(R.obs, 0.f, A.N, env0, R.params, RG, R.values, SA, M, tp);
    else mlp_res_tile<NOBS, P, true>(sObs, clip_obs, A.N, env0, R.params, RG, R.values + (size_t)k * N, SA, M, tp);
    if (k == R.T) break;
    lds_barrier();          // the physics step of this iteration is done: observations in LDS, counters in memory
  }
}

template <int NOBS>
__global__ void __launch_bounds__(256) k_rollout_mlp(StepArgs A, RolloutDev R) {
  __shared__ __attribute__((aligned(16))) float sTab[LM_ITAB_FLOATS + 2];
  __shared__ __attribute__((aligned(16))) float sObs[ENVS_PER_WAVE * LM_MAX_OBS];
  __shared__ __attribute__((aligned(16))) float sSt[ENVS_PER_WAVE * 93];
  __shared__ float4 sStash[STASH_SLOTS * 64];
  __shared__ MlpSmem<NOBS> M;
  const int t = threadIdx.x, env0 = lm_block() * ENVS_PER_WAVE;
  const lm_params* P = A.params + ((env0 >= A.split) ? 1 : 0);
#ifdef LM_STAMPS
  if (threadIdx.x < 64) lm_stamp_lds[threadIdx.x] = 0;
  __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
  { unsigned long long t0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_) :: "memory"); if ((threadIdx.x & 63) == 0) lm_stamp_lds[16 * (threadIdx.x >> 6) + 15] = t0_; }
  __syncthreads();
#endif
  if (t < 64) {
    for (int k = 0; k <= R.T; k++) {
#pragma unroll
      for (int b = 0; b < MLP_RES_BARRIERS; b++) lds_barrier();      // the policy wavefronts' forward k
      LM_STAMP(13);
      if (k == R.T) break;
      // (as in k_rollout: the loop body compiled like a stand-alone kernel)
      int z = 0; asm volatile("" : "+s"(z));
      StepArgs B = A; B.state = A.state + z; B.cnt = A.cnt + z; B.N = A.N + z;
      const lm_params* Pk = P + z; const size_t Nk = (size_t)B.N;
      B.actions = R.actions + (size_t)k * Nk * 12; B.goal_rand = nullptr;
      B.W.out_obs = R.obs + (size_t)(k + 1) * Nk * NOBS; B.W.out_states = nullptr; B.W.out_rew = R.rewards + (size_t)k * Nk;
      B.W.out_resets = R.dones + (size_t)k * Nk; B.W.out_extras = nullptr; B.W.acc = R.acc_steps + 16 * k;
      step_dispatch(B, Pk, sTab, sObs, sSt, sStash);
      __builtin_amdgcn_s_waitcnt(0x0F70);      // the counters that key the next action noise are in the L2 before the sampling wavefront reads them
      lds_barrier();
      LM_STAMP(14);
    }
  } else {
    const int tp = t - 64;
    if (tp < 64) rollout_policy_loop<NOBS, 0>(A, R, P, sObs, M, env0, tp);
    else if (tp < 128) rollout_policy_loop<NOBS, 1>(A, R, P, sObs, M, env0, tp);
    else rollout_policy_loop<NOBS, 2>(A, R, P, sObs, M, env0, tp);
  }
#ifdef LM_STAMPS
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x < 1024) lm_stamp_out[blockIdx.x * 64 + threadIdx.x] = lm_stamp_lds[threadIdx.x];
#endif
}

// extras and success windows of the T steps of a persistent rollout, in step order: the accumulator rows are fetched (and cleared) 64 steps
// at a time, the window counters live in registers across the steps; same arithmetic as publish_extra
__global__ void __launch_bounds__(64) k_rollout_finalize(const lm_params* params, OutPtrs W, int N, long long* acc_steps, float* extras, int T) {
  __shared__ long long sAcc[64 * 16];
  __shared__ int sCnt[64 * 4];          // per step: goal resets, resets (all envs), goal resets, resets (first task)
  const int lane = threadIdx.x;
  int64_t* ns_g = reinterpret_cast<int64_t*>(W.stats); float* rate_g = reinterpret_cast<float*>(W.stats + 48);
  int64_t ns[6]; float rate[3];
#pragma unroll
  for (int i = 0; i < 6; i++) ns[i] = ns_g[i];
#pragma unroll
  for (int i = 0; i < 3; i++) rate[i] = rate_g[i];
  const int64_t max_cnt = (int64_t)params->max_reset_counts;
  for (int k0 = 0; k0 < T; k0 += 64) {
    const int nk = min(64, T - k0);
    {
      // all 16 loads of a lane in flight at once (they bypass the L2: issued one by one, each was a full round trip), then the clears
      long long v[16];
#pragma unroll
      for (int j = 0; j < 16; j++) { const int i = lane + 64 * j; v[j] = (i < nk * 16) ? __hip_atomic_load(acc_steps + 16 * (size_t)k0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0LL; }
#pragma unroll
      for (int j = 0; j < 16; j++) { const int i = lane + 64 * j; if (i < nk * 16) { sAcc[i] = v[j]; __hip_atomic_store(acc_steps + 16 * (size_t)k0 + i, 0LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
    // the means of a step do not depend on the other steps: one lane per step (same arithmetic as publish_extra)
    if (lane < nk) {
      const bool last = (k0 + lane == T - 1);
      float* ex = extras ? extras + (size_t)(k0 + lane) * LM_NUM_EXTRAS : nullptr;
#pragma unroll
      for (int j = 0; j < 14; j++) {
        const float sum = (float)((double)sAcc[16 * lane + j] * (1.0 / (double)ACC_SCALE));
        if (j < 7) { const float m = sum / (float)N; if (last) W.extras[j] = m; if (ex) ex[j] = m; }
        else if (j >= 9 && j < 12) { const float m = sum / (float)N; if (last) W.extras[j + 1] = m; if (ex) ex[j + 1] = m; }
        else sCnt[4 * lane + (j == 7 ? 0 : j == 8 ? 1 : j == 12 ? 2 : 3)] = (int)(sum + 0.5f);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
    // the success-rate windows are a recurrence over the steps: one lane walks them in order
    if (lane == 0) {
      for (int k = 0; k < nk; k++) {
        const bool last = (k0 + k == T - 1);
        float* ex = extras ? extras + (size_t)(k0 + k) * LM_NUM_EXTRAS : nullptr;
        const int64_t gs = sCnt[4 * k], rs = sCnt[4 * k + 1], gl = sCnt[4 * k + 2], rl = sCnt[4 * k + 3];
        success_window(ns + 0, rate + 0, gs, rs, max_cnt);
        success_window(ns + 2, rate + 1, gl, rl, max_cnt);
        success_window(ns + 4, rate + 2, gs - gl, rs - rl, max_cnt);
#pragma unroll
        for (int c = 0; c < 3; c++) { if (last) W.extras[7 + c] = rate[c]; if (ex) ex[7 + c] = rate[c]; }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; i++) ns_g[i] = ns[i];
#pragma unroll
    for (int i = 0; i < 3; i++) rate_g[i] = rate[i];
  }
}

__global__ void __launch_bounds__(64) k_reset_all(int64_t* cnt, int N) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i < N) cnt[3 * (size_t)N + i] = 1;
}

// ---- test / tooling kernels ---------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_apply_resets(StepArgs A) {
  const int lane = threadIdx.x, limb = lane & 3, envl = lane >> 2;
  const int envr = lm_block() * ENVS_PER_WAVE + envl, N = A.N; if (envr >= N) return; const int env = envr;
  const lm_params* P = A.params + ((lm_block() * ENVS_PER_WAVE >= A.split) ? 1 : 0);
  const int jj[3] = {limb, 4 + 2 * limb, 5 + 2 * limb};
  float* st = A.state; int64_t* cnt = A.cnt;
  if (cnt[3 * (size_t)N + env] == 0) return;
  int episode = (int)cnt[5 * (size_t)N + env];
  float u3[3];
  if (A.goal_rand) { u3[0] = A.goal_rand[(size_t)env * 3]; u3[1] = A.goal_rand[(size_t)env * 3 + 1]; u3[2] = A.goal_rand[(size_t)env * 3 + 2]; }
  else hash_uniform3(A.seed, (uint32_t)env, (uint32_t)episode, u3);
  Q4 g = quat_from_euler(P->goal_lo[0] + (P->goal_hi[0] - P->goal_lo[0]) * u3[0], P->goal_lo[1] + (P->goal_hi[1] - P->goal_lo[1]) * u3[1],
                         P->goal_lo[2] + (P->goal_hi[2] - P->goal_lo[2]) * u3[2]);
  for (int a = 0; a < 3; a++) {
    st[(size_t)(R_Q + jj[a]) * N + env] = P->init_q[jj[a]]; st[(size_t)(R_QD + jj[a]) * N + env] = 0.f;
    st[(size_t)(R_LACT + jj[a]) * N + env] = 0.f; st[(size_t)(R_LQD + jj[a]) * N + env] = 0.f;
    st[(size_t)(R_LTIP + 3 * limb + a) * N + env] = P->default_tip[3 * limb + a];
    st[(size_t)(R_SE + jj[a]) * N + env] = P->init_se[jj[a]]; st[(size_t)(R_LTGT + jj[a]) * N + env] = P->init_q[jj[a]];
  }
  __builtin_amdgcn_wave_barrier();
  if (limb == 0) {
    for (int k = 0; k < 3; k++) {
      st[(size_t)(R_FB0 + k) * N + env] = P->init_base_pos[k]; st[(size_t)(R_FB0 + 7 + k) * N + env] = 0.f; st[(size_t)(R_FB0 + 10 + k) * N + env] = 0.f;
      st[(size_t)(R_FB1 + k) * N + env] = P->init_plate_pos[k]; st[(size_t)(R_FB1 + 7 + k) * N + env] = 0.f; st[(size_t)(R_FB1 + 10 + k) * N + env] = 0.f;
    }
    for (int k = 0; k < 4; k++) { st[(size_t)(R_FB0 + 3 + k) * N + env] = P->init_base_quat[k]; st[(size_t)(R_FB1 + 3 + k) * N + env] = P->init_plate_quat[k]; }
    st[(size_t)(R_GOAL + 0) * N + env] = g.w; st[(size_t)(R_GOAL + 1) * N + env] = g.x; st[(size_t)(R_GOAL + 2) * N + env] = g.y; st[(size_t)(R_GOAL + 3) * N + env] = g.z;
    const bool fixedb = (P->mode == LM_MODE_MANI);
    Q4 qb; qb.w = fixedb ? 1.f : P->init_base_quat[0]; qb.x = fixedb ? 0.f : -P->init_base_quat[1]; qb.y = fixedb ? 0.f : -P->init_base_quat[2]; qb.z = fixedb ? 0.f : -P->init_base_quat[3];
    Q4 d4 = qmul(qb, qconj(g));
    st[(size_t)R_LRD * N + env] = 2.0f * asinf(fminf(sqrtf(d4.x * d4.x + d4.y * d4.y + d4.z * d4.z), 1.0f));
  }
}
__global__ void __launch_bounds__(64) k_apply_resets_cnt(int64_t* cnt, int N) {
  int i = blockIdx.x * 64 + threadIdx.x; if (i >= N) return;
  if (cnt[3 * (size_t)N + i] != 0) { cnt[0 * (size_t)N + i] = 0; cnt[1 * (size_t)N + i] = 0; cnt[2 * (size_t)N + i] = 0; cnt[3 * (size_t)N + i] = 0; cnt[4 * (size_t)N + i] = 0; cnt[5 * (size_t)N + i] += 1; }
}

template <int MODE>
LM_DEV void load_phys(const float* st, int N, int env, int limb, FreeBody& F, float q[3], float qd[3]) {
  const int fb = (MODE == 0) ? R_FB0 : R_FB1; const int jj[3] = {limb, 4 + 2 * limb, 5 + 2 * limb};
  F.p = v3(st[(size_t)(fb + 0) * N + env], st[(size_t)(fb + 1) * N + env], st[(size_t)(fb + 2) * N + env]);
  F.q.w = st[(size_t)(fb + 3) * N + env]; F.q.x = st[(size_t)(fb + 4) * N + env]; F.q.y = st[(size_t)(fb + 5) * N + env]; F.q.z = st[(size_t)(fb + 6) * N + env];
  V3 lin = v3(st[(size_t)(fb + 7) * N + env], st[(size_t)(fb + 8) * N + env], st[(size_t)(fb + 9) * N + env]);
  V3 ang = v3(st[(size_t)(fb + 10) * N + env], st[(size_t)(fb + 11) * N + env], st[(size_t)(fb + 12) * N + env]);
  M3 R = quat_to_mat(F.q.w, F.q.x, F.q.y, F.q.z); F.u = sv(mulT(R, ang), mulT(R, lin));
  for (int a = 0; a < 3; a++) { q[a] = st[(size_t)(R_Q + jj[a]) * N + env]; qd[a] = st[(size_t)(R_QD + jj[a]) * N + env]; }
}
template <int MODE>
LM_DEV void store_phys(float* st, int N, int env, int limb, const FreeBody& F, const float q[3], const float qd[3]) {
  const int fb = (MODE == 0) ? R_FB0 : R_FB1; const int jj[3] = {limb, 4 + 2 * limb, 5 + 2 * limb};
  for (int a = 0; a < 3; a++) { st[(size_t)(R_Q + jj[a]) * N + env] = q[a]; st[(size_t)(R_QD + jj[a]) * N + env] = qd[a]; }
  if (limb == 0) {
    M3 R = quat_to_mat(F.q.w, F.q.x, F.q.y, F.q.z); V3 lin = mul(R, F.u.v), ang = mul(R, F.u.w);
    st[(size_t)(fb + 0) * N + env] = F.p.x; st[(size_t)(fb + 1) * N + env] = F.p.y; st[(size_t)(fb + 2) * N + env] = F.p.z;
    st[(size_t)(fb + 3) * N + env] = F.q.w; st[(size_t)(fb + 4) * N + env] = F.q.x; st[(size_t)(fb + 5) * N + env] = F.q.y; st[(size_t)(fb + 6) * N + env] = F.q.z;
    st[(size_t)(fb + 7) * N + env] = lin.x; st[(size_t)(fb + 8) * N + env] = lin.y; st[(size_t)(fb + 9) * N + env] = lin.z;
    st[(size_t)(fb + 10) * N + env] = ang.x; st[(size_t)(fb + 11) * N + env] = ang.y; st[(size_t)(fb + 12) * N + env] = ang.z;
  }
}

template <int MODE, int VAR>
LM_DEV void substeps_body(const StepArgs& A, const lm_params* P, const float* sTab, const float* targets, int n, float4* sStash) {
  const int lane = threadIdx.x, limb = lane & 3, envl = lane >> 2;
  Stash St; St.base = sStash; St.lane = lane;
  const int envr = lm_block() * ENVS_PER_WAVE + envl, N = A.N; const bool active = envr < N; const int env = active ? envr : N - 1;
  const float* tl = sTab + HUB_FLOATS + limb * LIMB_STRIDE; const int jj[3] = {limb, 4 + 2 * limb, 5 + 2 * limb};
  FreeBody F; float q[3], qd[3], tgt[3];
  load_phys<MODE>(A.state, N, env, limb, F, q, qd);
  for (int a = 0; a < 3; a++) tgt[a] = targets[(size_t)env * 12 + jj[a]];
  M3 Rfix = quat_to_mat(P->fixed_base_quat[0], P->fixed_base_quat[1], P->fixed_base_quat[2], P->fixed_base_quat[3]);
  V3 pfix = v3(P->fixed_base_pos[0], P->fixed_base_pos[1], P->fixed_base_pos[2]);
  float tau_acc[3] = {0.f, 0.f, 0.f};
  DrPhys X;
  for (int s = 0; s < n; s++) substep<MODE, VAR, 0>(P, sTab, tl, limb, St, F, Rfix, pfix, q, qd, tgt, tau_acc, X);
  if (active) store_phys<MODE>(A.state, N, env, limb, F, q, qd);
}
__global__ void __launch_bounds__(64) k_substeps(StepArgs A, const float* targets, int n) {
  __shared__ __attribute__((aligned(16))) float sTab[LM_ITAB_FLOATS + 2];
  __shared__ float4 sStash[STASH_SLOTS * 64];
  load_table(A.table, sTab, threadIdx.x);
  const lm_params* P = A.params + ((lm_block() * ENVS_PER_WAVE >= A.split) ? 1 : 0);
  if (P->variant == 0) { if (P->mode == LM_MODE_LOCO) substeps_body<0, 0>(A, P, sTab, targets, n, sStash); else substeps_body<1, 0>(A, P, sTab, targets, n, sStash); }
  else { if (P->mode == LM_MODE_LOCO) substeps_body<0, 1>(A, P, sTab, targets, n, sStash); else substeps_body<1, 1>(A, P, sTab, targets, n, sStash); }
}

__global__ void __launch_bounds__(64) k_fk(StepArgs A, float* tips, float* knees) {
  __shared__ __attribute__((aligned(16))) float sTab[LM_ITAB_FLOATS + 2];
  load_table(A.table, sTab, threadIdx.x);
  const lm_params* P = A.params + ((lm_block() * ENVS_PER_WAVE >= A.split) ? 1 : 0);
  const int lane = threadIdx.x, limb = lane & 3, envl = lane >> 2;
  const int env = lm_block() * ENVS_PER_WAVE + envl, N = A.N; if (env >= N) return;
  const float* tl = sTab + HUB_FLOATS + limb * LIMB_STRIDE;
  FreeBody F; float q[3], qd[3];
  M3 Rb; V3 pb;
  if (P->mode == LM_MODE_LOCO) { load_phys<0>(A.state, N, env, limb, F, q, qd); Rb = quat_to_mat(F.q.w, F.q.x, F.q.y, F.q.z); pb = F.p; }
  else { load_phys<1>(A.state, N, env, limb, F, q, qd); Rb = quat_to_mat(P->fixed_base_quat[0], P->fixed_base_quat[1], P->fixed_base_quat[2], P->fixed_base_quat[3]);
         pb = v3(P->fixed_base_pos[0], P->fixed_base_pos[1], P->fixed_base_pos[2]); }
  LimbKin K; float z3[3] = {0, 0, 0}; limb_kinematics(tl, q, z3, K);
  V3 xh, k2h, k3h; limb_points(tl, K, xh, k2h, k3h);
  V3 x = pb + mul(Rb, xh);
  V3 k2 = pb + mul(Rb, k2h), k3 = pb + mul(Rb, k3h);
  float* t = tips + ((size_t)env * 4 + limb) * 3; t[0] = x.x; t[1] = x.y; t[2] = x.z;
  float* kk = knees + ((size_t)env * 8 + 2 * limb) * 3; kk[0] = k2.x; kk[1] = k2.y; kk[2] = k2.z; kk[3] = k3.x; kk[4] = k3.y; kk[5] = k3.z;
}

// dense M (18x18) and h (18) of the loco system from the limb-aggregate terms (bring-up / parity tests)
__global__ void __launch_bounds__(64) k_debug_dyn(StepArgs A, float* Mout, float* hout) {
  __shared__ __attribute__((aligned(16))) float sTab[LM_ITAB_FLOATS + 2];
  load_table(A.table, sTab, threadIdx.x);
  const lm_params* P = A.params;
  const int lane = threadIdx.x, limb = lane & 3, envl = lane >> 2;
  const int envr = lm_block() * ENVS_PER_WAVE + envl, N = A.N; const bool active = envr < N; const int env = active ? envr : N - 1;
  const float* tl = sTab + HUB_FLOATS + limb * LIMB_STRIDE; const int jj[3] = {limb, 4 + 2 * limb, 5 + 2 * limb};
  FreeBody F; float q[3], qd[3];
  load_phys<0>(A.state, N, env, limb, F, q, qd);
  M3 Rb = quat_to_mat(F.q.w, F.q.x, F.q.y, F.q.z);
  SV avp0 = sv(v3(0, 0, 0), P->gravity * row2(Rb));
  LimbKin K; limb_kinematics(tl, q, qd, K);
  LimbDyn D; limb_dynamics(tl, K, qd, F.u, avp0, D);
  float Ac[6][6]; si_to_66(D.Isc, Ac);
  SI I0 = hub_inertia(sTab); float A0[6][6]; si_to_66(I0, A0);
  SV b = quad_sum(D.fcs) + I0 * avp0 + fcross(F.u, I0 * F.u);
  float* M = Mout + (size_t)env * 324; float* h = hout + (size_t)env * 18;
  float bb[6]; sv_to_arr(b, bb);
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { float v = quad_sum(Ac[i][j]) + A0[i][j]; if (active && limb == 0) M[i * 18 + j] = v; }
  if (active) {
    if (limb == 0) for (int i = 0; i < 6; i++) h[i] = bb[i];
    float f[3][6]; sv_to_arr(D.Fq0, f[0]); sv_to_arr(D.Fq1, f[1]); sv_to_arr(D.Fq2, f[2]);
    const int hi[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    for (int a = 0; a < 3; a++) {
      for (int i = 0; i < 6; i++) { M[i * 18 + 6 + jj[a]] = f[a][i]; M[(6 + jj[a]) * 18 + i] = f[a][i]; }
      for (int c = 0; c < 3; c++) M[(6 + jj[a]) * 18 + 6 + jj[c]] = D.H[hi[a][c]];
      h[6 + jj[a]] = D.hq[a];
    }
  }
}

template <int MODE>
LM_DEV void task_only_body(const StepArgs& A, const lm_params* P, const float* rb_all, float* sObs, float* sSt) {
  const int lane = threadIdx.x, limb = lane & 3, envl = lane >> 2;
  const int env0 = lm_block() * ENVS_PER_WAVE, envr = env0 + envl, N = A.N; const bool active = envr < N; const int env = active ? envr : N - 1;
  const int jj[3] = {limb, 4 + 2 * limb, 5 + 2 * limb};
  float* st = A.state; int64_t* cnt = A.cnt;
  const float* rb = rb_all + (size_t)env * 99;
  TaskIn I; TaskState S; TaskOut O;
  for (int a = 0; a < 3; a++) { I.q[a] = rb[jj[a]]; I.qd[a] = rb[12 + jj[a]]; I.acc[a] = rb[24 + jj[a]];
    I.act[a] = clampf(A.actions[(size_t)env * 12 + jj[a]], P->clip_actions); S.lact[a] = st[(size_t)(R_LACT + jj[a]) * N + env];
    I.torque[a] = rb[87 + jj[a]]; S.ltgt[a] = st[(size_t)(R_LTGT + jj[a]) * N + env]; }
  {
    // custom-controller tasks: integrate the swing / extension targets like pre_physics_step (:255-276)
    float se[3];
    for (int a = 0; a < 3; a++) { se[a] = st[(size_t)(R_SE + jj[a]) * N + env];
      if (P->variant >= 1) { se[a] = fminf(fmaxf(se[a] + I.act[a] * P->act_scale_se, P->se_lo[jj[a]]), P->se_hi[jj[a]]); if (active) st[(size_t)(R_SE + jj[a]) * N + env] = se[a]; } }
    I.tgtq[0] = se[0]; I.tgtq[1] = se[1] + 0.5f * se[2]; I.tgtq[2] = se[1] - 0.5f * se[2];
    S.lrd = st[(size_t)R_LRD * N + env];
  }
  I.fp = v3(rb[36], rb[37], rb[38]); I.fq.w = rb[39]; I.fq.x = rb[40]; I.fq.y = rb[41]; I.fq.z = rb[42];
  I.lin = v3(rb[43], rb[44], rb[45]); I.ang = v3(rb[46], rb[47], rb[48]);
  I.tipw = v3(rb[49 + 3 * limb], rb[50 + 3 * limb], rb[51 + 3 * limb]);
  I.knee2 = v3(rb[61 + 6 * limb], rb[62 + 6 * limb], rb[63 + 6 * limb]); I.knee3 = v3(rb[64 + 6 * limb], rb[65 + 6 * limb], rb[66 + 6 * limb]);
  S.ltip = v3(st[(size_t)(R_LTIP + 3 * limb) * N + env], st[(size_t)(R_LTIP + 3 * limb + 1) * N + env], st[(size_t)(R_LTIP + 3 * limb + 2) * N + env]);
  S.goal.w = st[(size_t)(R_GOAL + 0) * N + env]; S.goal.x = st[(size_t)(R_GOAL + 1) * N + env]; S.goal.y = st[(size_t)(R_GOAL + 2) * N + env]; S.goal.z = st[(size_t)(R_GOAL + 3) * N + env];
  S.succ = (int)cnt[0 * (size_t)N + env]; S.consec = (int)cnt[1 * (size_t)N + env]; S.greset = (int)cnt[2 * (size_t)N + env];
  S.reset = (int)cnt[3 * (size_t)N + env]; S.progress = (int)cnt[4 * (size_t)N + env]; int episode = (int)cnt[5 * (size_t)N + env];
  if (P->variant == 1) task_eval<MODE, 1>(P, limb, envl, I, S, O, sObs, sSt); else if (P->variant == 2) task_eval<MODE, 2>(P, limb, envl, I, S, O, sObs, sSt);
  else task_eval<MODE, 0>(P, limb, envl, I, S, O, sObs, sSt);
  if (active) {
    for (int a = 0; a < 3; a++) { st[(size_t)(R_LACT + jj[a]) * N + env] = S.lact[a]; st[(size_t)(R_LTGT + jj[a]) * N + env] = S.ltgt[a]; }
    if (limb == 0) st[(size_t)R_LRD * N + env] = S.lrd;
    st[(size_t)(R_LTIP + 3 * limb) * N + env] = S.ltip.x; st[(size_t)(R_LTIP + 3 * limb + 1) * N + env] = S.ltip.y; st[(size_t)(R_LTIP + 3 * limb + 2) * N + env] = S.ltip.z;
  }
  DrOut DO{};
  write_outputs<0>(P, A.W, N, env0, lane, limb, env, active, S, O, cnt, episode, sObs, sSt, DO);
}
__global__ void __launch_bounds__(64) k_task_eval(StepArgs A, const float* readback) {
  __shared__ __attribute__((aligned(16))) float sObs[ENVS_PER_WAVE * LM_MAX_OBS];
  __shared__ __attribute__((aligned(16))) float sSt[ENVS_PER_WAVE * 93];
  const lm_params* P = A.params + ((lm_block() * ENVS_PER_WAVE >= A.split) ? 1 : 0);
  if (P->mode == LM_MODE_LOCO) task_only_body<0>(A, P, readback, sObs, sSt); else task_only_body<1>(A, P, readback, sObs, sSt);
}

// ------------------------------------------------------------------------------------------------
// host side (C ABI)
// ------------------------------------------------------------------------------------------------
static void derive_params(lm_params* p) {
  // plate spatial inertia about its origin (plate coordinates) and its inverse (double precision Gauss-Jordan)
  double m = p->plate_mass, c[3] = {p->plate_com[0], p->plate_com[1], p->plate_com[2]};
  double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
  double IO[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) IO[i][j] = ((i == j) ? (p->plate_inertia[i] + m * cc) : 0.0) - m * c[i] * c[j];
  p->plate_si[0] = (float)m; p->plate_si[1] = (float)(m * c[0]); p->plate_si[2] = (float)(m * c[1]); p->plate_si[3] = (float)(m * c[2]);
  p->plate_si[4] = (float)IO[0][0]; p->plate_si[5] = (float)IO[1][1]; p->plate_si[6] = (float)IO[2][2];
  p->plate_si[7] = (float)IO[0][1]; p->plate_si[8] = (float)IO[0][2]; p->plate_si[9] = (float)IO[1][2];
  double h[3] = {m * c[0], m * c[1], m * c[2]};
  double hx[3][3] = {{0, -h[2], h[1]}, {h[2], 0, -h[0]}, {-h[1], h[0], 0}};
  double A[6][12];
  for (int i = 0; i < 6; i++) for (int j = 0; j < 12; j++) A[i][j] = (j - 6 == i) ? 1.0 : 0.0;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { A[i][j] = IO[i][j]; A[i][3 + j] = hx[i][j]; A[3 + i][j] = -hx[i][j]; A[3 + i][3 + j] = (i == j) ? m : 0.0; }
  for (int col = 0; col < 6; col++) {
    int piv = col; for (int r = col + 1; r < 6; r++) if (fabs(A[r][col]) > fabs(A[piv][col])) piv = r;
    if (piv != col) for (int j = 0; j < 12; j++) { double t = A[col][j]; A[col][j] = A[piv][j]; A[piv][j] = t; }
    double d = A[col][col]; if (d == 0.0) d = 1e-30;
    for (int j = 0; j < 12; j++) A[col][j] /= d;
    for (int r = 0; r < 6; r++) if (r != col) { double f = A[r][col]; for (int j = 0; j < 12; j++) A[r][j] -= f * A[col][j]; }
  }
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) p->plate_phi[6 * i + j] = (float)A[i][6 + j];
  p->ctrl_dt_inv = (float)(1.0 / ((double)p->dt * (double)p->substeps));
  p->acc_dt_inv = (float)(1.0 / ((double)p->dt * (double)(p->acc_substeps > 0 ? p->acc_substeps : 1)));
}


extern "C" {

const char* lm_last_error(void) { return g_err; }
const char* lm_version(void) { return "lm_engine 0.5 (gfx950, abi 4, policy tiles on the fp16 matrix pipe)"; }
int lm_abi_version(void) { return LM_ABI_VERSION; }

int lm_create(lm_engine** out, int n_envs, const float* table, const lm_params* params, int n_tasks, int split_env, uint32_t seed) {
  if (!out || !table || !params) return fail(LM_EINVAL, "lm_create: null argument");
  if (n_envs <= 0) return fail(LM_EINVAL, "lm_create: n_envs must be positive");
  if (n_envs >= (1 << (ACC_WIN_BITS - 1))) return fail(LM_EINVAL, "lm_create: n_envs must be below 2^25 per engine (width of the reset counts in the extras reduction)");
  if (n_tasks != 1 && n_tasks != 2) return fail(LM_EINVAL, "lm_create: n_tasks must be 1 or 2");
  // the stamp of the FIRST block is read before anything else of the struct: a caller built against another header passes a shifted layout
  if (params[0].abi_version != LM_ABI_VERSION || params[0].params_size != (int32_t)sizeof(lm_params) || params[0].table_floats != LM_TABLE_FLOATS)
    return fail(LM_EINVAL, "lm_create: ABI mismatch (lm_params.abi_version / params_size / table_floats differ from the library's LM_ABI_VERSION, sizeof(lm_params), LM_TABLE_FLOATS)");
  if (n_tasks == 2 && (params[1].abi_version != LM_ABI_VERSION || params[1].params_size != (int32_t)sizeof(lm_params) || params[1].table_floats != LM_TABLE_FLOATS))
    return fail(LM_EINVAL, "lm_create: ABI mismatch in the second parameter block");
  if (n_tasks == 2 && (split_env <= 0 || split_env >= n_envs || (split_env % ENVS_PER_WAVE) != 0))
    return fail(LM_EINVAL, "lm_create: split_env must be a multiple of 16 inside (0, n_envs)");
  for (int t = 0; t < n_tasks; t++) {
    const lm_params& p = params[t];
    if (!(p.dt > 0) || p.substeps <= 0 || p.pgs_iters < 0 || (p.mode != LM_MODE_LOCO && p.mode != LM_MODE_MANI))
      return fail(LM_EINVAL, "lm_create: invalid dt / substeps / pgs_iters / mode");
    if (p.pd_second_pass < 0 || p.pd_second_pass > 1) return fail(LM_EINVAL, "lm_create: pd_second_pass must be 0 or 1");
    if (p.drive_mode < 0 || p.drive_mode > 2 || (p.drive_mode != 0 && p.variant != 0) || (p.drive_mode == LM_DRIVE_POSITION && !(p.kd > 0)))
      return fail(LM_EINVAL, "lm_create: drive_mode must be 0 (velocity), 1 (position: kd > 0) or 2 (effort), and 0 for the PD-actuator variants");
    if ((p.num_obs != 64 && p.num_obs != LM_MAX_OBS) || p.num_obs != params[0].num_obs || p.variant < 0 || p.variant > 2 ||
        (p.variant == 1) != (p.num_obs == LM_MAX_OBS) || (p.variant >= 1 && !(p.kd > 0 && p.torque_div > 0 && p.acc_substeps >= 1 && p.acc_substeps <= p.substeps)))
      return fail(LM_EINVAL, "lm_create: invalid variant / num_obs (64 for velocity-drive and position-control tasks, 88 for custom-controller tasks, equal across tasks) or acc_substeps");
    if ((p.dr_enabled != 0) != (params[0].dr_enabled != 0)) return fail(LM_EINVAL, "lm_create: dr_enabled must be equal across tasks");
    if ((p.variant != 0) != (params[0].variant != 0)) return fail(LM_EINVAL, "lm_create: both parameter blocks must be of one actuator family (velocity drive: variant 0; PD actuator: variants 1 / 2)");
    if (p.dr_enabled) for (int c = 0; c < LM_DR_CHANNELS; c++) {
      const lm_dr_channel& ch = p.dr[c];
      if (!ch.enabled) continue;
      const bool noise_reset = (c == LM_DR_OBS_RESET || c == LM_DR_ACT_RESET), noise_interval = (c == LM_DR_OBS_INTERVAL || c == LM_DR_ACT_INTERVAL);
      if (ch.operation < 0 || ch.operation > 2 || ch.distribution < 0 || ch.distribution > 2 || ch.interval < 0 ||
          (noise_reset && ch.interval != 0) || (noise_interval && ch.interval < 1) || ((noise_reset || noise_interval) && ch.operation == LM_DR_DIRECT) ||
          (ch.distribution == LM_DR_LOGUNIFORM && !(ch.p0[0] > 0 && ch.p1[0] > 0)) || p.dr_min_frequency < 0)
        return fail(LM_EINVAL, "lm_create: invalid domain-randomisation channel (operation / distribution / interval / parameters)");
    }
  }
  int device = 0;
  HIPCHK(hipGetDevice(&device));
  lm_engine* h = new (std::nothrow) lm_engine();
  if (!h) return fail(LM_ENOMEM, "lm_create: host allocation failed");
  memset(h, 0, sizeof(*h));
  h->device = device;          // every buffer lives on the device current at creation; launches check it (on_device)
  h->N = n_envs; h->n_tasks = n_tasks; h->split = (n_tasks == 2) ? split_env : n_envs; h->seed = seed;
  h->nblocks = (n_envs + ENVS_PER_WAVE - 1) / ENVS_PER_WAVE;
  { const char* e = getenv("LM_W2_MIN_ENVS"); h->w2_min_envs = e ? atoi(e) : 32769; }
  h->num_obs = params[0].num_obs; h->dr_enabled = params[0].dr_enabled != 0;
  h->h_params[0] = params[0]; h->h_params[1] = params[n_tasks - 1];
  derive_params(&h->h_params[0]); derive_params(&h->h_params[1]);
  size_t N = (size_t)n_envs;
#define ALLOC(ptr, bytes) do { hipError_t e_ = hipMalloc((void**)&(ptr), (bytes)); if (e_ != hipSuccess) { snprintf(g_err, sizeof(g_err), "hipMalloc(%zu): %s", (size_t)(bytes), hipGetErrorString(e_)); lm_destroy(h); return LM_EHIP; } \
    e_ = hipMemset((ptr), 0, (bytes)); if (e_ != hipSuccess) { snprintf(g_err, sizeof(g_err), "hipMemset: %s", hipGetErrorString(e_)); lm_destroy(h); return LM_EHIP; } } while (0)
  ALLOC(h->d_params, 2 * sizeof(lm_params));
  ALLOC(h->d_table, LM_ITAB_FLOATS * sizeof(float));
  ALLOC(h->d_state, LM_STATE_ROWS * N * sizeof(float));
  ALLOC(h->d_cnt, LM_CNT_ROWS * N * sizeof(int64_t));
  ALLOC(h->d_drc, LM_DR_CNT_ROWS * N * sizeof(int64_t));
  ALLOC(h->d_dr_phys, LM_DR_PHYS_ROWS * N * sizeof(float));
  ALLOC(h->d_obs, N * (size_t)h->num_obs * sizeof(float));
  ALLOC(h->d_states, N * 93 * sizeof(float));
  ALLOC(h->d_rew, N * sizeof(float));
  ALLOC(h->d_extras, 16 * sizeof(float));
  ALLOC(h->d_terms, LM_TERM_ROWS * N * sizeof(float));
  h->acc_rows = LM_ACC_COPIES;      // a counted accumulator word takes at most 4095 arrivals (write_outputs)
  while (((N + ENVS_PER_WAVE - 1) / ENVS_PER_WAVE + h->acc_rows - 1) / h->acc_rows > (int)ACC_CNT_MASK) h->acc_rows *= 2;
  ALLOC(h->d_acc, (16 + (size_t)h->acc_rows * 16) * sizeof(long long));      // row 0: totals of a launch; rows 1..: first-level rows
  ALLOC(h->d_stats, 64);
#undef ALLOC
  float itab[LM_ITAB_FLOATS]; permute_table(table, itab);      // public packed layout -> the device's chain-interleaved layout
  if (hipMemcpy(h->d_params, h->h_params, 2 * sizeof(lm_params), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(h->d_table, itab, LM_ITAB_FLOATS * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    lm_destroy(h); return fail(LM_EHIP, "lm_create: parameter / table upload failed");
  }
  // identity quaternions so that an un-reset state is still valid; reset_buf = 1 (rl_task.py:111)
  {
    float* tmp = new float[LM_STATE_ROWS * N](); int64_t* ct = new int64_t[LM_CNT_ROWS * N]();
    for (size_t e = 0; e < N; e++) { tmp[(size_t)(R_FB0 + 3) * N + e] = 1.f; tmp[(size_t)(R_FB1 + 3) * N + e] = 1.f; tmp[(size_t)R_GOAL * N + e] = 1.f; ct[3 * N + e] = 1; }
    hipError_t e1 = hipMemcpy(h->d_state, tmp, LM_STATE_ROWS * N * sizeof(float), hipMemcpyHostToDevice);
    hipError_t e2 = hipMemcpy(h->d_cnt, ct, LM_CNT_ROWS * N * sizeof(int64_t), hipMemcpyHostToDevice);
    delete[] tmp; delete[] ct;
    if (e1 != hipSuccess || e2 != hipSuccess) { lm_destroy(h); return fail(LM_EHIP, "lm_create: initial upload failed"); }
  }
  *out = h;
  return LM_OK;
}

int lm_destroy(lm_engine* h) {
  if (!h) return LM_OK;
  void* ptrs[] = {h->d_params, h->d_table, h->d_state, h->d_cnt, h->d_drc, h->d_dr_phys, h->d_obs, h->d_states, h->d_rew, h->d_extras, h->d_terms, h->d_acc, h->d_stats};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  delete h;
  return LM_OK;
}

// The engine's buffers belong to the device that was current in lm_create: a launch from a thread whose current
// device differs would run on the wrong GPU (multi-GPU hosts run one process per GPU, so this is a caller bug).
static bool on_device(const lm_engine* h) {
  int d = -1;
  return hipGetDevice(&d) == hipSuccess && d == h->device;
}
#define CHECK_DEVICE(h, fn) do { if (!on_device(h)) return fail(LM_EINVAL, fn ": the calling thread's current device is not the engine's device"); } while (0)

static StepArgs make_args(lm_engine* h, const float* actions, const float* goal_rand, float* out_obs, float* out_states, float* out_rew, int64_t* out_resets) {
  StepArgs A;
  A.params = h->d_params; A.table = h->d_table; A.state = h->d_state; A.cnt = h->d_cnt; A.actions = actions; A.goal_rand = goal_rand;
  A.W.obs_buf = h->d_obs; A.W.states_buf = h->d_states; A.W.rew_buf = h->d_rew; A.W.terms = h->d_terms; A.W.acc = h->d_acc;
  A.W.stats = (char*)h->d_stats; A.W.extras = h->d_extras; A.W.out_extras = nullptr; A.W.split_block = h->split / ENVS_PER_WAVE; A.W.acc_rows = h->acc_rows;
  A.W.out_obs = out_obs; A.W.out_states = out_states; A.W.out_rew = out_rew; A.W.out_resets = out_resets;
  A.N = h->N; A.split = h->split; A.seed = h->seed; A.skip_reset = 0; A.nsub = -1; A.drc = h->d_drc; A.dr_phys = h->d_dr_phys;
  for (int t = 0; t < 2; t++) A.kind[t] = h->h_params[t].variant * 2 + (h->h_params[t].mode == LM_MODE_MANI ? 1 : 0);
  return A;
}

// The engine's own unclipped buffers (task.obs_buf / states_buf, rl_task.py:104-113, and the per-env reward terms) are a second copy of
// what the caller's out_obs / out_states receive clipped: a step writes them only once somebody has asked lm_ptr() for them (the Python
// task does, at construction), or when the caller passes no output buffer of that kind.  628 + 44 B per env-step of stores otherwise.
static void drop_unrequested_views(const lm_engine* h, StepArgs& A) {
  if (!h->view_obs && A.W.out_obs) A.W.obs_buf = nullptr;
  if (!h->view_states && A.W.out_states) A.W.states_buf = nullptr;
  if (!h->view_terms) A.W.terms = nullptr;
}

#ifdef LM_STAMPS
extern "C" int lm_debug_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(lm_stamp_out), sizeof(lm_stamp_out)) == hipSuccess ? 0 : -1; }
#endif

int lm_step(lm_engine* h, const float* actions, const float* goal_rand, float* out_obs, float* out_states, float* out_rew,
            int64_t* out_resets, float* out_extras, void* stream) {
  if (!h || !actions) return fail(LM_EINVAL, "lm_step: null handle or actions");
  CHECK_DEVICE(h, "lm_step");
  hipStream_t s = (hipStream_t)stream;
  StepArgs A = make_args(h, actions, goal_rand, out_obs, out_states, out_rew, out_resets); A.W.out_extras = out_extras;
  drop_unrequested_views(h, A);
  const bool pd = A.kind[0] >= 2;                                  // both blocks are of one actuator family (lm_create)
  if (!h->dr_enabled && A.kind[0] == 0 && A.kind[1] == 0 && h->N >= h->w2_min_envs) lm_internal_launch_step_w2(&A, h->nblocks, s);      // locomotion, two wavefronts per SIMD: ahead beyond 32 768 envs
  else {
    void (*kern)(StepArgs) = h->dr_enabled ? (pd ? k_step_dr_pd : k_step_dr) : (pd ? k_step_pd : k_step);
    hipLaunchKernelGGL(kern, dim3(h->nblocks), dim3(64), 0, s, A);
  }
  HIPCHK(hipGetLastError());
  return LM_OK;
}

int lm_post_physics(lm_engine* h, const float* actions, float* out_obs, float* out_states, float* out_rew,
                    int64_t* out_resets, float* out_extras, void* stream) {
  if (!h || !actions) return fail(LM_EINVAL, "lm_post_physics: null handle or actions");
  CHECK_DEVICE(h, "lm_post_physics");
  if (h->dr_enabled) return fail(LM_EINVAL, "lm_post_physics: a randomised engine runs through lm_step only");
  hipStream_t s = (hipStream_t)stream;
  StepArgs A = make_args(h, actions, nullptr, out_obs, out_states, out_rew, out_resets); A.W.out_extras = out_extras;
  A.skip_reset = 1; A.nsub = 0;
  void (*kern)(StepArgs) = A.kind[0] >= 2 ? k_step_pd : k_step;
  hipLaunchKernelGGL(kern, dim3(h->nblocks), dim3(64), 0, s, A);
  HIPCHK(hipGetLastError());
  return LM_OK;
}

int lm_reset_all(lm_engine* h, void* stream) {
  if (!h) return fail(LM_EINVAL, "lm_reset_all: null handle");
  CHECK_DEVICE(h, "lm_reset_all");
  hipLaunchKernelGGL(k_reset_all, dim3((h->N + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->d_cnt, h->N);
  HIPCHK(hipGetLastError());
  return LM_OK;
}

int lm_task_eval(lm_engine* h, const float* readback, const float* actions, float* out_obs, float* out_states, float* out_rew,
                 int64_t* out_resets, float* out_extras, void* stream) {
  if (!h || !readback || !actions) return fail(LM_EINVAL, "lm_task_eval: null argument");
  CHECK_DEVICE(h, "lm_task_eval");
  hipStream_t s = (hipStream_t)stream;
  StepArgs A = make_args(h, actions, nullptr, out_obs, out_states, out_rew, out_resets); A.W.out_extras = out_extras;
  hipLaunchKernelGGL(k_task_eval, dim3(h->nblocks), dim3(64), 0, s, A, readback);
  HIPCHK(hipGetLastError());
  return LM_OK;
}

int lm_apply_resets(lm_engine* h, const float* goal_rand, void* stream) {
  if (!h) return fail(LM_EINVAL, "lm_apply_resets: null handle");
  CHECK_DEVICE(h, "lm_apply_resets");
  hipStream_t s = (hipStream_t)stream;
  StepArgs A = make_args(h, nullptr, goal_rand, nullptr, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_apply_resets, dim3(h->nblocks), dim3(64), 0, s, A);
  hipLaunchKernelGGL(k_apply_resets_cnt, dim3((h->N + 63) / 64), dim3(64), 0, s, h->d_cnt, h->N);
  HIPCHK(hipGetLastError());
  return LM_OK;
}

int lm_substeps(lm_engine* h, const float* targets, int n, void* stream) {
  if (!h || !targets || n < 0) return fail(LM_EINVAL, "lm_substeps: bad argument");
  CHECK_DEVICE(h, "lm_substeps");
  StepArgs A = make_args(h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_substeps, dim3(h->nblocks), dim3(64), 0, (hipStream_t)stream, A, targets, n);
  HIPCHK(hipGetLastError());
  return LM_OK;
}

int lm_forward_kinematics(lm_engine* h, float* tips, float* knees, void* stream) {
  if (!h || !tips || !knees) return fail(LM_EINVAL, "lm_forward_kinematics: null argument");
  CHECK_DEVICE(h, "lm_forward_kinematics");
  StepArgs A = make_args(h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_fk, dim3(h->nblocks), dim3(64), 0, (hipStream_t)stream, A, tips, knees);
  HIPCHK(hipGetLastError());
  return LM_OK;
}

int lm_debug_dynamics(lm_engine* h, float* M, float* hvec, void* stream) {
  if (!h || !M || !hvec) return fail(LM_EINVAL, "lm_debug_dynamics: null argument");
  CHECK_DEVICE(h, "lm_debug_dynamics");
  if (h->h_params[0].mode != LM_MODE_LOCO || h->n_tasks != 1) return fail(LM_EINVAL, "lm_debug_dynamics: loco engines only");
  StepArgs A = make_args(h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_debug_dyn, dim3(h->nblocks), dim3(64), 0, (hipStream_t)stream, A, M, hvec);
  HIPCHK(hipGetLastError());
  return LM_OK;
}

void* lm_ptr(lm_engine* h, int kind) {
  if (!h) return nullptr;
  switch (kind) {
    case LM_PTR_STATE: return h->d_state;
    case LM_PTR_CNT: return h->d_cnt;
    case LM_PTR_DR_CNT: return h->d_drc;
    case LM_PTR_DR_PHYS: return h->d_dr_phys;
    case LM_PTR_OBS_BUF: h->view_obs = true; return h->d_obs;
    case LM_PTR_STATES_BUF: h->view_states = true; return h->d_states;
    case LM_PTR_REW_BUF: return h->d_rew;
    case LM_PTR_EXTRAS: return h->d_extras;
    case LM_PTR_STATS: return h->d_stats;
    case LM_PTR_TERMS: h->view_terms = true; return h->d_terms;
    default: return nullptr;
  }
}
int lm_num_envs(const lm_engine* h) { return h ? h->N : 0; }
int lm_num_obs(const lm_engine* h) { return h ? h->num_obs : 0; }
int lm_set_seed(lm_engine* h, uint32_t seed) { if (!h) return fail(LM_EINVAL, "lm_set_seed: null handle"); h->seed = seed; return LM_OK; }

}  // extern "C"

// (lm_internal.h) the persistent rollout launch used by lm_rollout_run (lm_policy.hip)
int lm_internal_rollout(lm_engine* h, int policy, const LmRolloutArgs& R, hipStream_t s) {
  if (!h || !R.params || !R.log_std || !R.obs || !R.actions || !R.logp || !R.values || !R.rewards || !R.dones || !R.acc_steps || R.T <= 0) return -1;
  if (h->dr_enabled || R.nobs != h->num_obs) return fail(-1, "persistent rollout: domain-randomised engines and foreign observation widths run through the graph mode");
  if (!on_device(h)) return fail(-1, "persistent rollout: the calling thread's current device is not the engine's device");
  StepArgs A = make_args(h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (!h->view_obs) A.W.obs_buf = nullptr;           // as the graph mode's lm_step(out_obs = the rollout's slot, out_states = NULL) does
  if (!h->view_terms) A.W.terms = nullptr;
  RolloutDev D; D.params = R.params; D.log_std = R.log_std; D.obs = R.obs; D.actions = R.actions; D.logp = R.logp; D.values = R.values;
  D.rewards = R.rewards; D.dones = R.dones; D.acc_steps = R.acc_steps; D.T = R.T; D.noise_seed = R.noise_seed;
  static const bool streaming = getenv("LM_ROLLOUT_STREAMING_MLP") != nullptr;      // kernel experiments: the four-wavefront tile with streamed weights
  if (policy == LM_POLICY_MLP && R.nobs == 64 && !streaming) hipLaunchKernelGGL(k_rollout_mlp<64>, dim3(h->nblocks), dim3(256), 0, s, A, D);
  else if (policy == LM_POLICY_MLP && R.nobs == LM_MAX_OBS && !streaming) hipLaunchKernelGGL(k_rollout_mlp<LM_MAX_OBS>, dim3(h->nblocks), dim3(256), 0, s, A, D);
  else if (policy == LM_POLICY_MLP && R.nobs == 64) hipLaunchKernelGGL((k_rollout<64, LM_POLICY_MLP>), dim3(h->nblocks), dim3(256), 0, s, A, D);
  else if (policy == LM_POLICY_MLP && R.nobs == LM_MAX_OBS) hipLaunchKernelGGL((k_rollout<LM_MAX_OBS, LM_POLICY_MLP>), dim3(h->nblocks), dim3(256), 0, s, A, D);
  else if (policy == LM_POLICY_GNN && R.nobs == 64) hipLaunchKernelGGL((k_rollout<64, LM_POLICY_GNN>), dim3(h->nblocks), dim3(256), 0, s, A, D);
  else return -1;
  hipLaunchKernelGGL(k_rollout_finalize, dim3(1), dim3(64), 0, s, h->d_params, A.W, h->N, R.acc_steps, R.extras, R.T);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int lm_internal_rollout_supported(const lm_engine* h, int policy, int nobs) {
  if (!h || h->dr_enabled || nobs != h->num_obs) return 0;
  return (policy == LM_POLICY_MLP && (nobs == 64 || nobs == LM_MAX_OBS)) || (policy == LM_POLICY_GNN && nobs == 64);
}

uint64_t lm_internal_args_key(const lm_engine* h) {
  return h ? ((uint64_t)h->seed | ((uint64_t)((h->view_obs ? 1 : 0) | (h->view_states ? 2 : 0) | (h->view_terms ? 4 : 0)) << 32)) : 0ull;
}
int lm_internal_fail(int code, const char* msg) { return fail(code, msg); }
#endif      // !LM_W2_UNIT
