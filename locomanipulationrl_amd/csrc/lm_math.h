// lm_math.h -- device-side vector / quaternion / spatial algebra for the CDNA4 step kernel.
// All quantities are fp32 in registers; small structs with named members so that nothing is
// runtime-indexed (runtime-indexed arrays would go to scratch on gfx950).
//
// Every type is a template over its scalar T in {float, f2}.  f2 = two floats in an aligned VGPR pair: arithmetic on it compiles to the
// packed fp32 instructions of gfx90a+ (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 results per lane per issue slot, the rate the
// 157 TFLOP/s vector peak is quoted at), and a plain float used as an operand of a packed op is broadcast by the instruction's op_sel bits,
// without a move.  The step kernel uses the pair to run the two structurally identical chains of a limb
// (shell -> link4 -> link3 and shell -> link1 -> link2) in the same instructions (lm_engine.hip, limb_kinematics / limb_dynamics).
#pragma once
#include <hip/hip_runtime.h>

#define LM_DEV __device__ __forceinline__

typedef float f2 __attribute__((ext_vector_type(2)));
LM_DEV f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
LM_DEV f2 sp2(float a) { return mk2(a, a); }
LM_DEV float fma_(float a, float b, float c) { return fmaf(a, b, c); }
LM_DEV f2 fma_(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

template <class T> struct V3T { T x, y, z; };
template <class T> struct M3T { V3T<T> c0, c1, c2; };          // columns
template <class T> struct SVT { V3T<T> w, v; };                // spatial motion [omega; v_O]  or force [n_O; f]
template <class T> struct SIT { T m; V3T<T> h; T xx, yy, zz, xy, xz, yz; };   // spatial inertia about O: mass, h = m*c, I_O
typedef V3T<float> V3; typedef M3T<float> M3; typedef SVT<float> SV; typedef SIT<float> SI;
typedef V3T<f2> V3P; typedef M3T<f2> M3P; typedef SVT<f2> SVP; typedef SIT<f2> SIP;

template <class T> LM_DEV V3T<T> v3t(T x, T y, T z) { V3T<T> r; r.x = x; r.y = y; r.z = z; return r; }
LM_DEV V3 v3(float x, float y, float z) { return v3t<float>(x, y, z); }
LM_DEV V3P v3(f2 x, f2 y, f2 z) { return v3t<f2>(x, y, z); }
template <class T> LM_DEV V3T<T> operator+(V3T<T> a, V3T<T> b) { return v3t<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> LM_DEV V3T<T> operator-(V3T<T> a, V3T<T> b) { return v3t<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> LM_DEV V3T<T> operator-(V3T<T> a) { return v3t<T>(-a.x, -a.y, -a.z); }
template <class T> LM_DEV V3T<T> operator*(T s, V3T<T> a) { return v3t<T>(s * a.x, s * a.y, s * a.z); }
template <class T> LM_DEV T dot(V3T<T> a, V3T<T> b) { return fma_(a.x, b.x, fma_(a.y, b.y, a.z * b.z)); }
template <class T> LM_DEV V3T<T> cross(V3T<T> a, V3T<T> b) { return v3t<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
template <class T> LM_DEV V3T<T> fma3(T s, V3T<T> a, V3T<T> b) { return v3t<T>(fma_(s, a.x, b.x), fma_(s, a.y, b.y), fma_(s, a.z, b.z)); }   // s*a + b

template <class T> LM_DEV V3T<T> mul(const M3T<T>& A, V3T<T> v) { return fma3(v.x, A.c0, fma3(v.y, A.c1, v.z * A.c2)); }
template <class T> LM_DEV V3T<T> mulT(const M3T<T>& A, V3T<T> v) { return v3t<T>(dot(A.c0, v), dot(A.c1, v), dot(A.c2, v)); }
template <class T> LM_DEV M3T<T> mul(const M3T<T>& A, const M3T<T>& B) { M3T<T> r; r.c0 = mul(A, B.c0); r.c1 = mul(A, B.c1); r.c2 = mul(A, B.c2); return r; }
template <class T> LM_DEV M3T<T> mulTA(const M3T<T>& A, const M3T<T>& B) { M3T<T> r; r.c0 = mulT(A, B.c0); r.c1 = mulT(A, B.c1); r.c2 = mulT(A, B.c2); return r; }  // A^T B
LM_DEV V3 row0(const M3& A) { return v3(A.c0.x, A.c1.x, A.c2.x); }
LM_DEV V3 row1(const M3& A) { return v3(A.c0.y, A.c1.y, A.c2.y); }
LM_DEV V3 row2(const M3& A) { return v3(A.c0.z, A.c1.z, A.c2.z); }

// pair <-> scalar: bc = the same value in both halves (free as an operand of a packed instruction), lo / hi = the halves (free)
LM_DEV V3P bc(V3 a) { return v3(sp2(a.x), sp2(a.y), sp2(a.z)); }
LM_DEV M3P bc(const M3& A) { M3P r; r.c0 = bc(A.c0); r.c1 = bc(A.c1); r.c2 = bc(A.c2); return r; }
LM_DEV V3 lo(V3P a) { return v3(a.x.x, a.y.x, a.z.x); }
LM_DEV V3 hi(V3P a) { return v3(a.x.y, a.y.y, a.z.y); }
LM_DEV M3 lo(const M3P& A) { M3 r; r.c0 = lo(A.c0); r.c1 = lo(A.c1); r.c2 = lo(A.c2); return r; }
LM_DEV M3 hi(const M3P& A) { M3 r; r.c0 = hi(A.c0); r.c1 = hi(A.c1); r.c2 = hi(A.c2); return r; }

// quaternion (w,x,y,z) -> rotation matrix (columns)
LM_DEV M3 quat_to_mat(float w, float x, float y, float z) {
  M3 R;
  R.c0 = v3(1.f - 2.f * (y * y + z * z), 2.f * (x * y + w * z), 2.f * (x * z - w * y));
  R.c1 = v3(2.f * (x * y - w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + w * x));
  R.c2 = v3(2.f * (x * z + w * y), 2.f * (y * z - w * x), 1.f - 2.f * (x * x + y * y));
  return R;
}
struct Q4 { float w, x, y, z; };
LM_DEV Q4 qmul(Q4 a, Q4 b) {
  Q4 r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  return r;
}
LM_DEV Q4 qconj(Q4 a) { Q4 r; r.w = a.w; r.x = -a.x; r.y = -a.y; r.z = -a.z; return r; }

// ---- spatial algebra (everything expressed in one frame, about one origin O)
template <class T> LM_DEV SVT<T> sv(V3T<T> w, V3T<T> v) { SVT<T> r; r.w = w; r.v = v; return r; }
template <class T> LM_DEV SVT<T> operator+(SVT<T> a, SVT<T> b) { return sv(a.w + b.w, a.v + b.v); }
template <class T> LM_DEV SVT<T> operator-(SVT<T> a, SVT<T> b) { return sv(a.w - b.w, a.v - b.v); }
template <class T> LM_DEV SVT<T> operator*(T s, SVT<T> a) { return sv(s * a.w, s * a.v); }
template <class T> LM_DEV SVT<T> fma6(T s, SVT<T> a, SVT<T> b) { return sv(fma3(s, a.w, b.w), fma3(s, a.v, b.v)); }
template <class T> LM_DEV T sdot(SVT<T> a, SVT<T> b) { return dot(a.w, b.w) + dot(a.v, b.v); }
template <class T> LM_DEV SVT<T> mcross(SVT<T> a, SVT<T> b) { return sv(cross(a.w, b.w), cross(a.w, b.v) + cross(a.v, b.w)); }        // motion x motion
template <class T> LM_DEV SVT<T> fcross(SVT<T> a, SVT<T> f) { return sv(cross(a.w, f.w) + cross(a.v, f.v), cross(a.w, f.v)); }        // motion x* force
// unit revolute axis through point o with direction z: [z; o x z]
template <class T> LM_DEV SVT<T> axis_sv(V3T<T> z, V3T<T> o) { return sv(z, cross(o, z)); }
LM_DEV SVP bc(SV a) { return sv(bc(a.w), bc(a.v)); }
LM_DEV SV lo(SVP a) { return sv(lo(a.w), lo(a.v)); }
LM_DEV SV hi(SVP a) { return sv(hi(a.w), hi(a.v)); }

template <class T> LM_DEV V3T<T> symmul(const SIT<T>& I, V3T<T> w) {
  return v3t<T>(I.xx * w.x + I.xy * w.y + I.xz * w.z, I.xy * w.x + I.yy * w.y + I.yz * w.z, I.xz * w.x + I.yz * w.y + I.zz * w.z);
}
template <class T> LM_DEV SVT<T> operator*(const SIT<T>& I, SVT<T> x) { return sv(symmul(I, x.w) + cross(I.h, x.v), fma3(I.m, x.v, cross(x.w, I.h))); }
template <class T> LM_DEV SIT<T> operator+(const SIT<T>& a, const SIT<T>& b) {
  SIT<T> r; r.m = a.m + b.m; r.h = a.h + b.h; r.xx = a.xx + b.xx; r.yy = a.yy + b.yy; r.zz = a.zz + b.zz;
  r.xy = a.xy + b.xy; r.xz = a.xz + b.xz; r.yz = a.yz + b.yz; return r;
}
LM_DEV SI lo(const SIP& a) { SI r; r.m = a.m.x; r.h = lo(a.h); r.xx = a.xx.x; r.yy = a.yy.x; r.zz = a.zz.x; r.xy = a.xy.x; r.xz = a.xz.x; r.yz = a.yz.x; return r; }
LM_DEV SI hi(const SIP& a) { SI r; r.m = a.m.y; r.h = hi(a.h); r.xx = a.xx.y; r.yy = a.yy.y; r.zz = a.zz.y; r.xy = a.xy.y; r.xz = a.xz.y; r.yz = a.yz.y; return r; }
// body inertia (m, com in body frame, I about COM in body axes [xx,yy,zz,xy,xz,yz]) placed at pose (R,o) -> about O
template <class T> LM_DEV SIT<T> place_inertia(const T* t, const M3T<T>& R, V3T<T> o) {
  T m = t[0];
  V3T<T> c = o + mul(R, v3t<T>(t[1], t[2], t[3]));
  // T = R * Ib (Ib symmetric), Iw = T * R^T
  V3T<T> t0 = fma3(t[4], R.c0, fma3(t[7], R.c1, t[8] * R.c2));   // T col0 = R * Ib col0 = xx*c0 + xy*c1 + xz*c2
  V3T<T> t1 = fma3(t[7], R.c0, fma3(t[5], R.c1, t[9] * R.c2));   // xy, yy, yz
  V3T<T> t2 = fma3(t[8], R.c0, fma3(t[9], R.c1, t[6] * R.c2));   // xz, yz, zz
  // Iw[i][j] = sum_k T[i][k] R[j][k]
  SIT<T> I; I.m = m; I.h = m * c;
  T cc = dot(c, c);
  I.xx = t0.x * R.c0.x + t1.x * R.c1.x + t2.x * R.c2.x + m * (cc - c.x * c.x);
  I.yy = t0.y * R.c0.y + t1.y * R.c1.y + t2.y * R.c2.y + m * (cc - c.y * c.y);
  I.zz = t0.z * R.c0.z + t1.z * R.c1.z + t2.z * R.c2.z + m * (cc - c.z * c.z);
  I.xy = t0.x * R.c0.y + t1.x * R.c1.y + t2.x * R.c2.y - m * c.x * c.y;
  I.xz = t0.x * R.c0.z + t1.x * R.c1.z + t2.x * R.c2.z - m * c.x * c.z;
  I.yz = t0.y * R.c0.z + t1.y * R.c1.z + t2.y * R.c2.z - m * c.y * c.z;
  return I;
}

// ---- 4-lane (one env) cross-lane helpers: DPP quad permutes, no LDS traffic
LM_DEV float quad_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
LM_DEV float quad_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
LM_DEV float quad_sum(float v) { v += quad_xor1(v); v += quad_xor2(v); return v; }
template <int K> LM_DEV float quad_bcast(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), K * 0x55, 0xF, 0xF, true)); }
LM_DEV int quad_sum_i(int v) {
  v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);
  v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);
  return v;
}
// Sum over the wavefront of a value that is non-zero only in lane 3 of every quad (one value per env), returned wave-uniform.  Row shifts by
// 4 and 8 lanes leave each 16-lane row's sum in its lane 15, row_bcast:15 / row_bcast:31 carry it across the rows into lane 63: four DPP
// adds and a v_readlane, no LDS.  The additions pair up exactly as in the xor-butterfly (4, 8, 16, 32) the oracle restates:
// ((v3 + v7) + (v11 + v15)) per row, then (R0 + R1) + (R2 + R3).
#define LM_DPP_ADD(v, ctrl, row_mask) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xF, true))
LM_DEV float wave_sum_lane3(float v) {
  LM_DPP_ADD(v, 0x114, 0xF);      // row_shr:4
  LM_DPP_ADD(v, 0x118, 0xF);      // row_shr:8
  LM_DPP_ADD(v, 0x142, 0xA);      // row_bcast:15 into rows 1 and 3
  LM_DPP_ADD(v, 0x143, 0xC);      // row_bcast:31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef LM_DPP_ADD
LM_DEV V3 quad_sum(V3 a) { return v3(quad_sum(a.x), quad_sum(a.y), quad_sum(a.z)); }
LM_DEV SV quad_sum(SV a) { return sv(quad_sum(a.w), quad_sum(a.v)); }

// ---- small dense inverses
// symmetric 3x3 inverse, packed [00,01,02,11,12,22]
LM_DEV void inv3sym(const float* a, float* o) {
  float c00 = a[3] * a[5] - a[4] * a[4];
  float c01 = a[2] * a[4] - a[1] * a[5];
  float c02 = a[1] * a[4] - a[2] * a[3];
  float det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  float id = 1.0f / det;
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = (a[0] * a[5] - a[2] * a[2]) * id;
  o[4] = (a[1] * a[2] - a[0] * a[4]) * id;
  o[5] = (a[0] * a[3] - a[1] * a[1]) * id;
}
// symmetric positive definite 6x6 inverse via Cholesky; A is full [6][6] (only upper used), result full symmetric
LM_DEV void inv6spd(float A[6][6], float P[6][6]) {
  float L[6][6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    float s = A[j][j];
#pragma unroll
    for (int k = 0; k < j; k++) s = fmaf(-L[j][k], L[j][k], s);
    float inv = rsqrtf(s);            // 1/L[j][j]
    L[j][j] = inv;                    // store the reciprocal on the diagonal
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float t = A[j][i];
#pragma unroll
      for (int k = 0; k < j; k++) t = fmaf(-L[i][k], L[j][k], t);
      L[i][j] = t * inv;
    }
  }
  // Li = L^{-1} (lower)
  float Li[6][6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    Li[j][j] = L[j][j];
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float s = 0.f;
#pragma unroll
      for (int k = j; k < i; k++) s = fmaf(L[i][k], Li[k][j], s);
      Li[i][j] = -s * L[i][i];
    }
  }
  // P = Li^T Li
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++) {
      float s = 0.f;
#pragma unroll
      for (int k = j; k < 6; k++) s = fmaf(Li[k][i], Li[k][j], s);
      P[i][j] = s; P[j][i] = s;
    }
}
LM_DEV SV mul66(const float P[6][6], SV x) {
  float in[6] = {x.w.x, x.w.y, x.w.z, x.v.x, x.v.y, x.v.z}, o[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 6; j++) s = fmaf(P[i][j], in[j], s);
    o[i] = s;
  }
  return sv(v3(o[0], o[1], o[2]), v3(o[3], o[4], o[5]));
}
