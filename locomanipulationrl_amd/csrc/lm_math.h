// lm_math.h -- device-side vector / quaternion / spatial algebra for the CDNA4 step kernel.
// All quantities are fp32 in registers; small structs with named members so that nothing is
// runtime-indexed (runtime-indexed arrays would go to scratch on gfx950).
#pragma once
#include <hip/hip_runtime.h>

#define LM_DEV __device__ __forceinline__

struct V3 { float x, y, z; };
struct M3 { V3 c0, c1, c2; };          // columns
struct SV { V3 w, v; };                // spatial motion [omega; v_O]  or force [n_O; f]
struct SI { float m; V3 h; float xx, yy, zz, xy, xz, yz; };   // spatial inertia about O: mass, h = m*c, I_O

LM_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
LM_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
LM_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
LM_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
LM_DEV V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
LM_DEV float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
LM_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
LM_DEV V3 fma3(float s, V3 a, V3 b) { return v3(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z)); }   // s*a + b

LM_DEV V3 mul(const M3& A, V3 v) { return fma3(v.x, A.c0, fma3(v.y, A.c1, v.z * A.c2)); }
LM_DEV V3 mulT(const M3& A, V3 v) { return v3(dot(A.c0, v), dot(A.c1, v), dot(A.c2, v)); }
LM_DEV M3 mul(const M3& A, const M3& B) { M3 r; r.c0 = mul(A, B.c0); r.c1 = mul(A, B.c1); r.c2 = mul(A, B.c2); return r; }
LM_DEV M3 mulTA(const M3& A, const M3& B) { M3 r; r.c0 = mulT(A, B.c0); r.c1 = mulT(A, B.c1); r.c2 = mulT(A, B.c2); return r; }  // A^T B
LM_DEV V3 row0(const M3& A) { return v3(A.c0.x, A.c1.x, A.c2.x); }
LM_DEV V3 row1(const M3& A) { return v3(A.c0.y, A.c1.y, A.c2.y); }
LM_DEV V3 row2(const M3& A) { return v3(A.c0.z, A.c1.z, A.c2.z); }

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
LM_DEV SV sv(V3 w, V3 v) { SV r; r.w = w; r.v = v; return r; }
LM_DEV SV operator+(SV a, SV b) { return sv(a.w + b.w, a.v + b.v); }
LM_DEV SV operator-(SV a, SV b) { return sv(a.w - b.w, a.v - b.v); }
LM_DEV SV operator*(float s, SV a) { return sv(s * a.w, s * a.v); }
LM_DEV SV fma6(float s, SV a, SV b) { return sv(fma3(s, a.w, b.w), fma3(s, a.v, b.v)); }
LM_DEV float sdot(SV a, SV b) { return dot(a.w, b.w) + dot(a.v, b.v); }
LM_DEV SV mcross(SV a, SV b) { return sv(cross(a.w, b.w), cross(a.w, b.v) + cross(a.v, b.w)); }        // motion x motion
LM_DEV SV fcross(SV a, SV f) { return sv(cross(a.w, f.w) + cross(a.v, f.v), cross(a.w, f.v)); }        // motion x* force
// unit revolute axis through point o with direction z: [z; o x z]
LM_DEV SV axis_sv(V3 z, V3 o) { return sv(z, cross(o, z)); }

LM_DEV V3 symmul(const SI& I, V3 w) {
  return v3(I.xx * w.x + I.xy * w.y + I.xz * w.z, I.xy * w.x + I.yy * w.y + I.yz * w.z, I.xz * w.x + I.yz * w.y + I.zz * w.z);
}
LM_DEV SV operator*(const SI& I, SV x) { return sv(symmul(I, x.w) + cross(I.h, x.v), fma3(I.m, x.v, cross(x.w, I.h))); }
LM_DEV SI operator+(const SI& a, const SI& b) {
  SI r; r.m = a.m + b.m; r.h = a.h + b.h; r.xx = a.xx + b.xx; r.yy = a.yy + b.yy; r.zz = a.zz + b.zz;
  r.xy = a.xy + b.xy; r.xz = a.xz + b.xz; r.yz = a.yz + b.yz; return r;
}
// body inertia (m, com in body frame, I about COM in body axes [xx,yy,zz,xy,xz,yz]) placed at pose (R,o) -> about O
LM_DEV SI place_inertia(const float* t, const M3& R, V3 o) {
  float m = t[0];
  V3 c = o + mul(R, v3(t[1], t[2], t[3]));
  // T = R * Ib (Ib symmetric), Iw = T * R^T
  V3 t0 = fma3(t[4], R.c0, fma3(t[7], R.c1, t[8] * R.c2));   // T col0 = R * Ib col0 = xx*c0 + xy*c1 + xz*c2
  V3 t1 = fma3(t[7], R.c0, fma3(t[5], R.c1, t[9] * R.c2));   // xy, yy, yz
  V3 t2 = fma3(t[8], R.c0, fma3(t[9], R.c1, t[6] * R.c2));   // xz, yz, zz
  // Iw[i][j] = sum_k T[i][k] R[j][k]
  SI I; I.m = m; I.h = m * c;
  float cc = dot(c, c);
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
