/*
 * lm_oracle.c -- CPU oracle (see lm_oracle.h for scope, citations and the
 * "parity unpinned vs PhysX" statement).  TEST INFRASTRUCTURE ONLY.
 *
 * Physics algorithm (deliberately NOT the kernel's algorithm):
 *   dense world-axes body Jacobians -> M = sum J^T I J, h = sum J^T (I a_vp + w x I w, m(a_vp+g))
 *   -> loop closure by coordinate projection G (20 tree DoF -> 12 independent)
 *   -> implicit velocity drive as a diagonal augmentation dt*kd, 2-pass torque clamp
 *   -> dense Cholesky, dense Delassus W = Jc Minv Jc^T, projected Gauss-Seidel
 *   -> semi-implicit Euler.
 */
#include "lm_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

#define NU 18

/* ------------------------------------------------------------------ small math */
static void m3mul(const real* A, const real* B, real* C) {
  real T[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    real s = 0; for (int k = 0; k < 3; k++) s += A[3*i+k] * B[3*k+j]; T[3*i+j] = s; }
  memcpy(C, T, sizeof(T));
}
static void m3T(const real* A, real* C) { real T[9]; for (int i=0;i<3;i++) for(int j=0;j<3;j++) T[3*i+j]=A[3*j+i]; memcpy(C,T,sizeof(T)); }
static void m3v(const real* A, const real* v, real* o) { real t[3]; for (int i=0;i<3;i++) t[i]=A[3*i]*v[0]+A[3*i+1]*v[1]+A[3*i+2]*v[2]; o[0]=t[0];o[1]=t[1];o[2]=t[2]; }
static void m3Tv(const real* A, const real* v, real* o) { real t[3]; for (int i=0;i<3;i++) t[i]=A[i]*v[0]+A[3+i]*v[1]+A[6+i]*v[2]; o[0]=t[0];o[1]=t[1];o[2]=t[2]; }
static void cross(const real* a, const real* b, real* o) { real t0=a[1]*b[2]-a[2]*b[1], t1=a[2]*b[0]-a[0]*b[2], t2=a[0]*b[1]-a[1]*b[0]; o[0]=t0;o[1]=t1;o[2]=t2; }
static real dot3(const real* a, const real* b) { return a[0]*b[0]+a[1]*b[1]+a[2]*b[2]; }

static void quat_to_mat(const real* q, real* R) { /* wxyz */
  real w=q[0],x=q[1],y=q[2],z=q[3];
  R[0]=1-2*(y*y+z*z); R[1]=2*(x*y-w*z);   R[2]=2*(x*z+w*y);
  R[3]=2*(x*y+w*z);   R[4]=1-2*(x*x+z*z); R[5]=2*(y*z-w*x);
  R[6]=2*(x*z-w*y);   R[7]=2*(y*z+w*x);   R[8]=1-2*(x*x+y*y);
}
static void quat_mul(const real* a, const real* b, real* o) { /* Hamilton, wxyz */
  real w=a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3];
  real x=a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2];
  real y=a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1];
  real z=a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0];
  o[0]=w;o[1]=x;o[2]=y;o[3]=z;
}
static void quat_conj(const real* a, real* o) { o[0]=a[0]; o[1]=-a[1]; o[2]=-a[2]; o[3]=-a[3]; }
/* rotation about a unit axis (Rodrigues) */
static void axis_rot(const real* a, real th, real* R) {
  real c = cos(th), s = sin(th), v = 1 - c;
  R[0]=c+a[0]*a[0]*v;      R[1]=a[0]*a[1]*v-a[2]*s; R[2]=a[0]*a[2]*v+a[1]*s;
  R[3]=a[1]*a[0]*v+a[2]*s; R[4]=c+a[1]*a[1]*v;      R[5]=a[1]*a[2]*v-a[0]*s;
  R[6]=a[2]*a[0]*v-a[1]*s; R[7]=a[2]*a[1]*v+a[0]*s; R[8]=c+a[2]*a[2]*v;
}

/* loop-closure function g(D) = 2 atan(sqrt2 tan(D/2)) and derivatives */
static void closure_g(real D, real* g, real* g1, real* g2) {
  const real s2 = sqrt((real)2);
  real den = 3 - cos(D);
  *g = 2 * atan2(s2 * sin(D / 2), cos(D / 2));
  *g1 = 2 * s2 / den;
  *g2 = -2 * s2 * sin(D) / (den * den);
}

void lmo_quat_from_euler(real roll, real pitch, real yaw, real* q) {
  real cy=cos(yaw*0.5), sy=sin(yaw*0.5), cr=cos(roll*0.5), sr=sin(roll*0.5), cp=cos(pitch*0.5), sp=sin(pitch*0.5);
  q[0]=cy*cr*cp+sy*sr*sp; q[1]=cy*sr*cp-sy*cr*sp; q[2]=cy*cr*sp+sy*sr*cp; q[3]=sy*cr*cp-cy*sr*sp;
}

static uint32_t mix32(uint32_t x) { /* lowbias32 finalizer */
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
void lmo_hash_uniform3(uint32_t seed, uint32_t env, uint32_t episode, real* u3) {
  uint32_t base = mix32(seed ^ mix32(env * 0x9E3779B9U + 0x7F4A7C15U) ^ mix32(episode * 0x85EBCA6BU + 0x165667B1U));
  for (uint32_t k = 0; k < 3; k++) {
    uint32_t r = mix32(base + (k + 1U) * 0xC2B2AE35U);
    u3[k] = (real)(r >> 8) * (real)(1.0 / 16777216.0);
  }
}
int lmo_sizeof_real(void) { return (int)sizeof(real); }

/* ------------------------------------------------------------------ domain randomisation: counter-based samples */
real lmo_dr_sample(uint32_t seed, uint32_t stream, uint32_t env, uint32_t key, uint32_t idx, int distribution, real p0, real p1) {
  /* components 2p and 2p+1 share one pair of uniforms: Box-Muller's cosine and sine branches (two normals per logarithm) */
  const uint32_t pair = idx >> 1, odd = idx & 1U;
  uint32_t base = mix32(seed ^ mix32(env * 0x9E3779B9U + 0x7F4A7C15U) ^ mix32(key * 0x85EBCA6BU + 0x165667B1U) ^ mix32(stream * 0x27D4EB2FU + 0x632BE5ABU));
  uint32_t r1 = mix32(base + (2U * pair + 1U) * 0xC2B2AE35U), r2 = mix32(base + (2U * pair + 2U) * 0xC2B2AE35U);
  real u1 = ((real)(r1 >> 8) + (real)1) * (real)(1.0 / 16777216.0);     /* (0, 1] */
  real u2 = (real)(r2 >> 8) * (real)(1.0 / 16777216.0);                 /* [0, 1) */
  if (distribution == 0) { real rad = sqrt(-2 * log(u1)), ang = (real)6.283185307179586 * u2; return p0 + p1 * (real)(rad * (odd ? sin(ang) : cos(ang))); }   /* torch.normal(mean, std) */
  real u = odd ? u1 - (real)(1.0 / 16777216.0) : u2;                     /* both in [0, 1) */
  if (distribution == 1) return p0 + (p1 - p0) * u;                                                       /* (hi-lo)*rand + lo */
  return (real)exp(log(p0) + (log(p1) - log(p0)) * u);                                                      /* loguniform */
}

static uint32_t g_env0 = 0;      /* global id of the first env of the block being processed (co-training: the second half starts at N/2) */
void lmo_set_env_offset(uint32_t env0) { g_env0 = env0; }
static real dr_apply(int operation, real x, real n) { return operation == 0 ? x + n : (operation == 1 ? x * n : n); }

void lmo_dr_noise(const lmo_dr_channel* on_reset, const lmo_dr_channel* on_interval, uint32_t seed, uint32_t stream, int N, int D,
                  real* buf, const int64_t* reset_flags, int64_t* counter, const int64_t* corr_key, const int64_t* step_key) {
  for (int e = 0; e < N; e++) {
    real* b = buf + (size_t)e * D;
    /* randomize.py:213-216 / 237-240: counter[reset ids] = 0; counter += 1 */
    if (reset_flags[e]) counter[e] = 0;
    counter[e] += 1;
    /* :218-226: correlated noise, redrawn for the envs whose reset flag is set, applied to every env on every call */
    if (on_reset && on_reset->enabled)
      for (int j = 0; j < D; j++)
        b[j] = dr_apply(on_reset->operation, b[j], lmo_dr_sample(seed, stream, g_env0 + (uint32_t)e, (uint32_t)corr_key[e], (uint32_t)j, on_reset->distribution, (real)on_reset->p0[0], (real)on_reset->p1[0]));
    /* :228-236: uncorrelated noise for the envs whose counter reached frequency_interval */
    if (on_interval && on_interval->enabled && counter[e] >= on_interval->interval) {
      counter[e] = 0;
      for (int j = 0; j < D; j++)
        b[j] = dr_apply(on_interval->operation, b[j], lmo_dr_sample(seed, stream + 1U, g_env0 + (uint32_t)e, (uint32_t)step_key[e], (uint32_t)j, on_interval->distribution, (real)on_interval->p0[0], (real)on_interval->p1[0]));
    }
  }
}

typedef struct { real tmax[12], vmax[12], g[3], f[3], cj[12]; } env_dr_t;
static __thread const env_dr_t* g_dr = 0;
      /* per-env physics overrides while lmo_step_dr runs a sub-step */

/* ------------------------------------------------------------------ kinematics */
typedef struct {
  real Rw[LMO_MAXB][9], ow[LMO_MAXB][3], zw[LMO_MAXB][3], cw[LMO_MAXB][3], Iw[LMO_MAXB][9];
  real qt[LMO_NTREE];               /* tree joint angles */
  real G[LMO_NTREE][LMO_NQ];        /* d qtree / d q */
  real qdd_vp[LMO_NTREE];           /* passive joint accel at qdd_indep = 0 */
  real qdt[LMO_NTREE];              /* tree joint rates */
} kin_t;

static void kin_compute(const lmo_model* m, const real* R0, const real* p0, const real* q, const real* qd, kin_t* K) {
  memset(K->G, 0, sizeof(K->G)); memset(K->qdd_vp, 0, sizeof(K->qdd_vp));
  for (int i = 0; i < LMO_NQ; i++) { K->qt[i] = q[i]; K->G[i][i] = 1; }
  for (int c = 0; c < m->nclos; c++) {
    int pd = m->clos_p[c], a = m->clos_a[c], b = m->clos_b[c]; real s = (real)m->clos_s[c];
    real g, g1, g2; closure_g(q[a] - q[b], &g, &g1, &g2);
    K->qt[pd] = s * g; K->G[pd][a] = s * g1; K->G[pd][b] = -s * g1;
    real dd = qd ? (qd[a] - qd[b]) : 0;
    K->qdd_vp[pd] = s * g2 * dd * dd;
  }
  for (int t = 0; t < LMO_NTREE; t++) { real s = 0; if (qd) for (int j = 0; j < LMO_NQ; j++) s += K->G[t][j] * qd[j]; K->qdt[t] = s; }
  for (int k = 0; k < m->nb; k++) {
    if (m->parent[k] < 0) { memcpy(K->Rw[k], R0, 9*sizeof(real)); memcpy(K->ow[k], p0, 3*sizeof(real)); }
    else {
      int par = m->parent[k]; real Rt[9], ax[3], pt[3], Rj[9], Rq[9], t[3];
      for (int i = 0; i < 9; i++) Rt[i] = (real)m->Rt[k][i];
      for (int i = 0; i < 3; i++) { ax[i] = (real)m->axis[k][i]; pt[i] = (real)m->pt[k][i]; }
      axis_rot(ax, K->qt[m->dof[k]], Rq);
      m3mul(K->Rw[par], Rt, Rj); m3mul(Rj, Rq, K->Rw[k]);
      m3v(K->Rw[par], pt, t); for (int i = 0; i < 3; i++) K->ow[k][i] = K->ow[par][i] + t[i];
      m3v(K->Rw[k], ax, K->zw[k]);
    }
    real c[3], Ib[9], T[9], RT[9];
    for (int i = 0; i < 3; i++) c[i] = (real)m->com[k][i];
    for (int i = 0; i < 9; i++) Ib[i] = (real)m->inertia[k][i];
    m3v(K->Rw[k], c, c); for (int i = 0; i < 3; i++) K->cw[k][i] = K->ow[k][i] + c[i];
    m3mul(K->Rw[k], Ib, T); m3T(K->Rw[k], RT); m3mul(T, RT, K->Iw[k]);
  }
}

void lmo_fk(const lmo_model* m, const lmo_params* p, const real* phys, real* tips, real* knees) {
  real R0[9], p0[3], q0[4]; kin_t K;
  if (p->mode == 0) { quat_to_mat(phys + 3, R0); memcpy(p0, phys, 3*sizeof(real)); }
  else { for (int i=0;i<4;i++) q0[i]=(real)p->fixed_base_quat[i]; quat_to_mat(q0, R0); for (int i=0;i<3;i++) p0[i]=(real)p->fixed_base_pos[i]; }
  kin_compute(m, R0, p0, phys + 13, NULL, &K);
  for (int i = 0; i < 4; i++) { real off[3], t[3]; for (int j=0;j<3;j++) off[j]=(real)m->tip_off[i][j];
    m3v(K.Rw[m->tip_body[i]], off, t); for (int j=0;j<3;j++) tips[3*i+j] = K.ow[m->tip_body[i]][j] + t[j]; }
  for (int i = 0; i < 8; i++) for (int j=0;j<3;j++) knees[3*i+j] = K.ow[m->knee_body[i]][j];
}

/* ------------------------------------------------------------------ dense dynamics
 * Generalised velocity u[18]:
 *   mode 0: [w_b(3), v_b(3)] = base spatial velocity in BASE coordinates, then qd(12)
 *   mode 1: [w_b(3), v_b(3)] = plate spatial velocity (at plate origin) in PLATE coordinates, then qd(12)
 */
typedef struct {
  real M[NU][NU], h[NU];
  real tip[4][3];          /* world centres of the foot spheres */
  real cpt[4][3];          /* world contact points = centre - tip_radius * n (on the sphere's surface) */
  real Jt[4][3][NU];       /* world-axes linear Jacobian of the contact points wrt u (robot side) */
  real R0[9], p0[3];       /* robot base pose */
  real Rf[9], pf[3];       /* free body pose (= base in mode 0, plate in mode 1) */
} dyn_t;

static void dyn_compute(const lmo_model* m, const lmo_params* p, const real* phys, dyn_t* D) {
  kin_t K; real wb[3] = {0,0,0}, vb[3] = {0,0,0};
  const real* q = phys + 13; const real* qd = phys + 25;
  if (p->mode == 0) {
    quat_to_mat(phys + 3, D->R0); memcpy(D->p0, phys, 3*sizeof(real));
    m3Tv(D->R0, phys + 10, wb); m3Tv(D->R0, phys + 7, vb);
    memcpy(D->Rf, D->R0, sizeof(D->R0)); memcpy(D->pf, D->p0, sizeof(D->p0));
  } else {
    real q0[4]; for (int i=0;i<4;i++) q0[i]=(real)p->fixed_base_quat[i]; quat_to_mat(q0, D->R0);
    for (int i=0;i<3;i++) D->p0[i]=(real)p->fixed_base_pos[i];
    quat_to_mat(phys + 40, D->Rf); memcpy(D->pf, phys + 37, 3*sizeof(real));
  }
  kin_compute(m, D->R0, D->p0, q, qd, &K);
  memset(D->M, 0, sizeof(D->M)); memset(D->h, 0, sizeof(D->h));
  const int floating = (p->mode == 0);
  /* per-body classical velocity-product kinematics */
  real w[LMO_MAXB][3], al[LMO_MAXB][3], ao[LMO_MAXB][3];
  for (int k = 0; k < m->nb; k++) {
    int par = m->parent[k];
    if (par < 0) {
      real vo[3]; m3v(D->R0, wb, w[k]); m3v(D->R0, vb, vo);
      al[k][0]=al[k][1]=al[k][2]=0; cross(w[k], vo, ao[k]);
    } else {
      real zq[3], t[3], d[3], t2[3]; real qdk = K.qdt[m->dof[k]];
      for (int i=0;i<3;i++) { zq[i] = K.zw[k][i]*qdk; w[k][i] = w[par][i] + zq[i]; }
      cross(w[par], zq, t);
      for (int i=0;i<3;i++) al[k][i] = al[par][i] + t[i] + K.zw[k][i]*K.qdd_vp[m->dof[k]];
      for (int i=0;i<3;i++) d[i] = K.ow[k][i]-K.ow[par][i];
      cross(al[par], d, t); cross(w[par], d, t2); cross(w[par], t2, t2);
      for (int i=0;i<3;i++) ao[k][i] = ao[par][i] + t[i] + t2[i];
    }
  }
  /* Jacobians + accumulation */
  for (int k = 0; k < m->nb; k++) {
    real JA[3][NU], JL[3][NU]; memset(JA,0,sizeof(JA)); memset(JL,0,sizeof(JL));
    real JAt[3][LMO_NTREE], JLt[3][LMO_NTREE]; memset(JAt,0,sizeof(JAt)); memset(JLt,0,sizeof(JLt));
    for (int j = k; m->parent[j] >= 0; j = m->parent[j]) {
      int dj = m->dof[j]; real r[3], t[3];
      for (int i=0;i<3;i++) r[i] = K.cw[k][i]-K.ow[j][i];
      cross(K.zw[j], r, t);
      for (int i=0;i<3;i++) { JAt[i][dj] = K.zw[j][i]; JLt[i][dj] = t[i]; }
    }
    for (int i=0;i<3;i++) for (int c=0;c<LMO_NQ;c++) { real sa=0, sl=0;
      for (int t=0;t<LMO_NTREE;t++) { sa += JAt[i][t]*K.G[t][c]; sl += JLt[i][t]*K.G[t][c]; }
      JA[i][6+c]=sa; JL[i][6+c]=sl; }
    if (floating) {
      real r[3]; for (int i=0;i<3;i++) r[i]=K.cw[k][i]-D->p0[i];
      for (int c=0;c<3;c++) { real col[3]={D->R0[c],D->R0[3+c],D->R0[6+c]}, t[3]; cross(col, r, t); /* (R e_c) x r = -[r]x R e_c */
        for (int i=0;i<3;i++) { JA[i][c]=col[i]; JL[i][c]=t[i]; JL[i][3+c]=col[i]; } }
    }
    /* forces */
    real mass=(real)m->mass[k], rc[3], t[3], t2[3], ac[3], Iw_[3], nA[3], fL[3];
    for (int i=0;i<3;i++) rc[i]=K.cw[k][i]-K.ow[k][i];
    cross(al[k], rc, t); cross(w[k], rc, t2); cross(w[k], t2, t2);
    for (int i=0;i<3;i++) ac[i]=ao[k][i]+t[i]+t2[i];
    m3v(K.Iw[k], w[k], Iw_); cross(w[k], Iw_, t); m3v(K.Iw[k], al[k], nA);
    for (int i=0;i<3;i++) { nA[i]+=t[i]; fL[i]=mass*ac[i]; }
    if (!g_dr) fL[2]+=mass*(real)p->gravity;
    else { for (int i=0;i<3;i++) fL[i]-=mass*g_dr->g[i];                 /* randomised gravity vector */
           if (m->parent[k]<0) for (int i=0;i<3;i++) fL[i]-=g_dr->f[i]; }   /* external force on the base link, at its COM, world axes */
    for (int a=0;a<NU;a++) {
      real IJ[3]; real ja[3]={JA[0][a],JA[1][a],JA[2][a]}; m3v(K.Iw[k], ja, IJ);
      for (int b=a;b<NU;b++) { real s=0; for (int i=0;i<3;i++) s += IJ[i]*JA[i][b] + mass*JL[i][a]*JL[i][b]; D->M[a][b]+=s; }
      real s=0; for (int i=0;i<3;i++) s += JA[i][a]*nA[i] + JL[i][a]*fL[i]; D->h[a]+=s;
    }
  }
  for (int a=0;a<NU;a++) for (int b=0;b<a;b++) D->M[a][b]=D->M[b][a];
  /* tips */
  for (int i=0;i<4;i++) {
    int tb=m->contact_body[i]; real off[3], t[3]; for (int j=0;j<3;j++) off[j]=(real)m->contact_off[i][j];
    m3v(K.Rw[tb], off, t); for (int j=0;j<3;j++) D->tip[i][j]=K.ow[tb][j]+t[j];
    /* contact normal (from the surface towards the sphere): world z on the ground; the plate's face normal on the robot's side */
    { real n[3]={0,0,1};
      if (!floating) { real db[3], yb[3]; for (int a=0;a<3;a++) db[a]=D->p0[a]-D->pf[a]; m3Tv(D->Rf, db, yb);
        real sgn=(yb[2]-(real)p->plate_center[2]>=0)?1:-1; real ez[3]={0,0,sgn}; m3v(D->Rf, ez, n); }
      for (int a=0;a<3;a++) D->cpt[i][a]=D->tip[i][a]-(real)p->tip_radius*n[a]; }
    real JLt[3][LMO_NTREE]; memset(JLt,0,sizeof(JLt)); memset(D->Jt[i],0,sizeof(D->Jt[i]));
    for (int j=tb; m->parent[j]>=0; j=m->parent[j]) { real r[3], tt[3]; for (int a=0;a<3;a++) r[a]=D->cpt[i][a]-K.ow[j][a];
      cross(K.zw[j], r, tt); for (int a=0;a<3;a++) JLt[a][m->dof[j]]=tt[a]; }
    for (int a=0;a<3;a++) for (int c=0;c<LMO_NQ;c++) { real s=0; for (int tt=0;tt<LMO_NTREE;tt++) s+=JLt[a][tt]*K.G[tt][c]; D->Jt[i][a][6+c]=s; }
    if (floating) { real r[3]; for (int a=0;a<3;a++) r[a]=D->cpt[i][a]-D->p0[a];
      for (int c=0;c<3;c++) { real col[3]={D->R0[c],D->R0[3+c],D->R0[6+c]}, tt[3]; cross(col, r, tt);
        for (int a=0;a<3;a++) { D->Jt[i][a][c]=tt[a]; D->Jt[i][a][3+c]=col[a]; } } }
  }
  if (!floating) {
    /* plate: spatial inertia about plate origin in plate coordinates, u = [w_b, v_b] */
    real mp=(real)p->plate_mass, c[3], Ic[3]; for (int i=0;i<3;i++){c[i]=(real)p->plate_com[i]; Ic[i]=(real)p->plate_inertia[i];}
    real IO[3][3]; real cc=dot3(c,c);
    for (int i=0;i<3;i++) for (int j=0;j<3;j++) IO[i][j]=(i==j?Ic[i]+mp*cc:0)-mp*c[i]*c[j];
    real cx[3][3]={{0,-c[2],c[1]},{c[2],0,-c[0]},{-c[1],c[0],0}};
    for (int i=0;i<3;i++) for (int j=0;j<3;j++) { D->M[i][j]=IO[i][j]; D->M[i][3+j]=mp*cx[i][j]; D->M[3+i][j]=-mp*cx[i][j]; D->M[3+i][3+j]=(i==j)?mp:0; }
    real wpl[3], vpl[3]; m3Tv(D->Rf, phys+47, wpl); m3Tv(D->Rf, phys+44, vpl);
    /* momentum: n = IO w + m c x v ; f = m v - m c x w */
    real n[3], f[3], t[3], t2[3];
    for (int i=0;i<3;i++) n[i]=IO[i][0]*wpl[0]+IO[i][1]*wpl[1]+IO[i][2]*wpl[2];
    cross(c, vpl, t); for (int i=0;i<3;i++) n[i]+=mp*t[i];
    cross(c, wpl, t); for (int i=0;i<3;i++) f[i]=mp*(vpl[i]-t[i]);
    /* v x* [n;f] = [w x n + v x f ; w x f] */
    cross(wpl, n, t); cross(vpl, f, t2); for (int i=0;i<3;i++) D->h[i]=t[i]+t2[i];
    cross(wpl, f, t); for (int i=0;i<3;i++) D->h[3+i]=t[i];
    /* gravity force in plate coords at COM */
    real gw[3]={0,0,-(real)p->gravity}, gb[3]; if (g_dr) for (int i=0;i<3;i++) gw[i]=g_dr->g[i]; m3Tv(D->Rf, gw, gb);
    real fg[3]={mp*gb[0],mp*gb[1],mp*gb[2]}, ng[3]; cross(c, fg, ng);
    for (int i=0;i<3;i++) { D->h[i]-=ng[i]; D->h[3+i]-=fg[i]; }
  }
}

void lmo_dyn_terms(const lmo_model* m, const lmo_params* p, const real* phys, real* M, real* h) {
  dyn_t* D = (dyn_t*)malloc(sizeof(dyn_t)); dyn_compute(m, p, phys, D);
  for (int a=0;a<NU;a++) { for (int b=0;b<NU;b++) M[a*NU+b]=D->M[a][b]; h[a]=D->h[a]; }
  free(D);
}

/* Cholesky (lower) in place; returns 0 on success */
static int chol(real A[NU][NU]) {
  for (int j=0;j<NU;j++) { real s=A[j][j]; for (int k=0;k<j;k++) s-=A[j][k]*A[j][k]; if (s<=0) return 1; real d=sqrt(s); A[j][j]=d;
    for (int i=j+1;i<NU;i++) { real t=A[i][j]; for (int k=0;k<j;k++) t-=A[i][k]*A[j][k]; A[i][j]=t/d; } }
  return 0;
}
static void chol_solve(real L[NU][NU], const real* b, real* x) {
  real y[NU]; for (int i=0;i<NU;i++) { real s=b[i]; for (int k=0;k<i;k++) s-=L[i][k]*y[k]; y[i]=s/L[i][i]; }
  for (int i=NU-1;i>=0;i--) { real s=y[i]; for (int k=i+1;k<NU;k++) s-=L[k][i]*x[k]; x[i]=s/L[i][i]; }
}

/* debug capture of the contact problem of the LAST pass executed (tests / solver studies) */
static __thread real* g_cap_W = 0; static __thread real* g_cap_vf = 0; static __thread real* g_cap_bn = 0; static __thread real* g_cap_lam = 0;

/* contact rows of one env: Jc[3i .. 3i+2] = (normal, t1, t2) rows of contact i with respect to u, phi[i] = signed gap */
static void contact_rows(const lmo_params* p, const dyn_t* D, real Jc[12][NU], real* phi_out) {
  memset(Jc,0,sizeof(real)*12*NU);
  for (int i=0;i<4;i++) {
    real n[3], t1[3], t2[3], phi; real Jrel[3][NU]; memcpy(Jrel, D->Jt[i], sizeof(Jrel));
    if (p->mode==0) {
      /* n = world z; t1 = the base's x axis projected onto the ground plane, t2 = n x t1: the friction basis turns with the robot,
         so the dynamics do not depend on its heading */
      real hx=D->R0[0], hy=D->R0[3], hn=sqrt(hx*hx+hy*hy); if (hn<(real)1e-6) hn=(real)1e-6;
      n[0]=0;n[1]=0;n[2]=1; t1[0]=hx/hn;t1[1]=hy/hn;t1[2]=0; t2[0]=-t1[1];t2[1]=t1[0];t2[2]=0; phi=D->tip[i][2]-(real)p->tip_radius; }
    else {
      real d[3], y0[3], y[3], yc[3];
      for (int a=0;a<3;a++) d[a]=D->tip[i][a]-D->pf[a];
      m3Tv(D->Rf, d, yc);                                                                                              /* sphere centre, plate coords */
      for (int a=0;a<3;a++) d[a]=D->cpt[i][a]-D->pf[a];
      m3Tv(D->Rf, d, y0);                                                                                              /* contact point, plate coords */
      for (int a=0;a<3;a++) y[a]=yc[a]-(real)p->plate_center[a];
      /* contact face = the slab face on the robot's side of the plate (robust to deep initial overlap) */
      real db[3], yb[3]; for (int a=0;a<3;a++) db[a]=D->p0[a]-D->pf[a]; m3Tv(D->Rf, db, yb);
      real sgn = (yb[2]-(real)p->plate_center[2]>=0)?1:-1; real ez[3]={0,0,sgn}, ex[3]={1,0,0};
      m3v(D->Rf, ez, n); m3v(D->Rf, ex, t1); cross(n, t1, t2);
      phi = sgn*y[2]-(real)p->plate_half[2]-(real)p->tip_radius;
      if (fabs(y[0])>(real)p->plate_half[0] || fabs(y[1])>(real)p->plate_half[1]) phi = 1.0e3;
      /* subtract plate point velocity: R_p (v_b + w_b x y0) */
      for (int c=0;c<3;c++) { real col[3]={D->Rf[c],D->Rf[3+c],D->Rf[6+c]}; real ec[3]={0,0,0}; ec[c]=1; real cy[3], t[3]; cross(ec, y0, cy); m3v(D->Rf, cy, t);
        for (int a=0;a<3;a++) { Jrel[a][c]-=t[a]; Jrel[a][3+c]-=col[a]; } }
    }
    for (int a=0;a<NU;a++) { real jn=0,j1=0,j2=0; for (int r=0;r<3;r++){ jn+=n[r]*Jrel[r][a]; j1+=t1[r]*Jrel[r][a]; j2+=t2[r]*Jrel[r][a]; }
      Jc[3*i][a]=jn; Jc[3*i+1][a]=j1; Jc[3*i+2][a]=j2; }
    phi_out[i]=phi;
  }
}

static void substep_tgs(const lmo_model* m, const lmo_params* p, real* phys, const real* target, real* tau_out);

static void substep_one(const lmo_model* m, const lmo_params* p, real* phys, const real* target, real* tau_out) {
  if (p->solver==1 && p->variant==0 && p->drive_mode==0) { substep_tgs(m, p, phys, target, tau_out); return; }
  dyn_t* D = (dyn_t*)malloc(sizeof(dyn_t));
  dyn_compute(m, p, phys, D);
  const real dt=(real)p->dt, kd=(real)p->kd, tmax=(real)p->tau_max, mu=(real)p->mu, cj=(real)p->joint_damping;
  real u[NU];
  if (p->mode==0) { m3Tv(D->R0, phys+10, u); m3Tv(D->R0, phys+7, u+3); }
  else { m3Tv(D->Rf, phys+47, u); m3Tv(D->Rf, phys+44, u+3); }
  for (int j=0;j<12;j++) u[6+j]=phys[25+j];
  /* contact geometry */
  real Jc[12][NU], bn[4], phi4[4]; contact_rows(p, D, Jc, phi4);
  for (int i=0;i<4;i++) { real phi=phi4[i];
    if (phi>=0) bn[i]=phi/dt; else { real b=(real)p->baumgarte*phi/dt; if (b<-(real)p->max_depen_vel) b=-(real)p->max_depen_vel; bn[i]=b; }
  }
  /* effort mode (robot.py:455-459): the action is the joint torque itself, gains off - the constant-torque branch of the drive from the start */
  const int effort = (p->variant==0 && p->drive_mode==2);
  int sat[12]; real tsat[12]; for (int j=0;j<12;j++){sat[j]=effort;tsat[j]=effort?target[j]:0;}
  /* PD-actuator families (variants 1 / 2): the reference evaluates  clamp(kp (q* - q) - kd qd, +-max_effort)  on the state BEFORE the
     sub-step and holds it (quadruped_pose_control_custom_controller.py:289-293), so which joints sit on the limit is known up front:
     those get the constant limit torque, the others the implicit form of the same PD law (end-of-step velocity: the stable choice at
     kd dt / I of order 1), in one pass.  The implicit torque of an unsaturated joint leaves the limit in 0.02 % of the joint-sub-steps under
     random actions; pd_second_pass = 1 puts those on the limit too and solves again (the applied torque then never exceeds max_effort).
     (PhysX's own drives - variant 0 with a finite tau_max - limit the force of an implicit drive, which only the solve can tell: two passes.) */
  const int pd = (p->variant!=0);
  if (pd) for (int j=0;j<12;j++) { real tau=kd*(target[j]-u[6+j]); real tm=g_dr?g_dr->tmax[j]:tmax; if (tau>tm){sat[j]=1;tsat[j]=tm;} else if (tau<-tm){sat[j]=1;tsat[j]=-tm;} }
  real un[NU];
  for (int pass=0; pass<2; pass++) {
    real L[NU][NU]; memcpy(L, D->M, sizeof(L)); real rhs[NU];
    for (int a=0;a<NU;a++) rhs[a]=-D->h[a];
    for (int j=0;j<12;j++) { if (!sat[j]) { L[6+j][6+j]+=dt*kd; rhs[6+j]+=kd*(target[j]-u[6+j]); } else rhs[6+j]+=tsat[j];
      { real cjj=g_dr?g_dr->cj[j]:cj; L[6+j][6+j]+=dt*cjj; rhs[6+j]-=cjj*u[6+j]; } }       /* viscous joint damping, implicit */
    chol(L);
    real acc[NU], uf[NU]; chol_solve(L, rhs, acc); for (int a=0;a<NU;a++) uf[a]=u[a]+dt*acc[a];
    real MiJ[12][NU], W[12][12], vf[12], lam[12];
    for (int r=0;r<12;r++) { chol_solve(L, Jc[r], MiJ[r]); }
    for (int r=0;r<12;r++) { for (int c=0;c<12;c++){ real s=0; for (int a=0;a<NU;a++) s+=Jc[r][a]*MiJ[c][a]; W[r][c]=s; }
      real s=0; for (int a=0;a<NU;a++) s+=Jc[r][a]*uf[a]; vf[r]=s; lam[r]=0; }
    if (g_cap_W) { for (int r=0;r<12;r++) { for (int c=0;c<12;c++) g_cap_W[12*r+c]=W[r][c]; g_cap_vf[r]=vf[r]; } for (int i=0;i<4;i++) g_cap_bn[i]=bn[i]; }
    for (int it=0; it<p->pgs_iters; it++) for (int ii=0;ii<4;ii++) {
      const int i=(it&1)?3-ii:ii;      /* sweeps alternate direction: no limb is systematically relaxed first */
      int r=3*i; real v=vf[r]+bn[i]; for (int c=0;c<12;c++) v+=W[r][c]*lam[c];
      real ln=lam[r]-v/W[r][r]; if (ln<0) ln=0; lam[r]=ln;
      /* the two friction rows of a contact relax together, both from the state the normal row left (one packed update in the kernel);
         their mutual coupling W[t1][t2] enters at the contact's next turn.  The pair is then projected onto the friction CONE
         |lam_t| <= mu lam_n (isotropic Coulomb friction; DESIGN.md 2.1: the axis-aligned pyramid of rounds 1-2 made the replay outcomes depend
         on the sweep count and missed PhysX's rows; lmo_params.pyramid = 1 keeps it for the evidence tables) */
      real vt[2]; for (int k=1;k<3;k++) { int rr=r+k; vt[k-1]=vf[rr]; for (int c=0;c<12;c++) vt[k-1]+=W[rr][c]*lam[c]; }
      real l1=lam[r+1]-vt[0]/W[r+1][r+1], l2=lam[r+2]-vt[1]/W[r+2][r+2]; const real lim=mu*lam[r];
      if (p->pyramid) { if (l1>lim) l1=lim; if (l1<-lim) l1=-lim; if (l2>lim) l2=lim; if (l2<-lim) l2=-lim; }
      else { real n2=l1*l1+l2*l2; if (n2>lim*lim) { real sc=lim/sqrt(n2); l1*=sc; l2*=sc; } }
      lam[r+1]=l1; lam[r+2]=l2;
    }
    for (int a=0;a<NU;a++) { real s=uf[a]; for (int r=0;r<12;r++) s+=MiJ[r][a]*lam[r]; un[a]=s; }
    if (g_cap_lam) for (int r=0;r<12;r++) g_cap_lam[r]=lam[r];
    if (pass==0) { if (effort || (pd && !p->pd_second_pass)) break;
      int any=0; for (int j=0;j<12;j++) { if (sat[j]) continue; real tau=kd*(target[j]-un[6+j]); real tm=g_dr?g_dr->tmax[j]:tmax; if (tau>tm){sat[j]=1;tsat[j]=tm;any=1;} else if (tau<-tm){sat[j]=1;tsat[j]=-tm;any=1;} }
      if (!any) break; }
  }
  /* the drive torque LOGGED for this sub-step (observation 88 and the mechanical-power term of the PD-actuator tasks): the reference clips the torque
     before applying and logging it (quadruped_pose_control_custom_controller.py:289-293), so the log never leaves +-max_effort.  With pd_second_pass = 0
     the implicit torque APPLIED to an unsaturated joint, kd (v* - qd_end), can exceed the limit (0.02 % of the joint-sub-steps, by at most kd x the
     sub-step's velocity change): that excess is dynamics-only, the logged value is clipped */
  if (tau_out) for (int j=0;j<12;j++) { real tm=g_dr?g_dr->tmax[j]:tmax, t = sat[j] ? tsat[j] : kd*(target[j]-un[6+j]); tau_out[j] = t>tm?tm:(t<-tm?-tm:t); }
  /* integrate; driven joints are speed-limited like PhysX's maxJointVelocity (Design/Scripts/config_module_joints.py:11,61-69) */
  for (int j=0;j<12;j++) { real v=un[6+j]; real vm=g_dr?g_dr->vmax[j]:(real)p->max_joint_vel; if (v>vm) v=vm; if (v<-vm) v=-vm; phys[25+j]=v; phys[13+j]+=dt*v; }
  {
    real* pos  = (p->mode==0)? phys   : phys+37;
    real* quat = (p->mode==0)? phys+3 : phys+40;
    real* lin  = (p->mode==0)? phys+7 : phys+44;
    real* ang  = (p->mode==0)? phys+10: phys+47;
    real ww[3]; m3v(D->Rf, un, ww);
    real th = sqrt(dot3(ww,ww))*dt, dq[4];
    if (th < (real)1e-8) { dq[0]=1; for (int a=0;a<3;a++) dq[1+a]=(real)0.5*dt*ww[a]; }
    else { real s=sin(th*(real)0.5)/(th/dt); dq[0]=cos(th*(real)0.5); for (int a=0;a<3;a++) dq[1+a]=s*ww[a]; }
    real qn[4]; quat_mul(dq, quat, qn); real nn=sqrt(qn[0]*qn[0]+qn[1]*qn[1]+qn[2]*qn[2]+qn[3]*qn[3]);
    for (int a=0;a<4;a++) quat[a]=qn[a]/nn;
    real Rn[9], vw[3]; quat_to_mat(quat, Rn); m3v(Rn, un+3, vw);
    for (int a=0;a<3;a++) { ang[a]=ww[a]; lin[a]=vw[a]; pos[a]+=dt*vw[a]; }
  }
  free(D);
}

/* lmo_params.solver = 1 (round 4): the velocity drive as rows of the same iteration as the contacts, on one linearisation
 * per sub-step.  Only numbers the reference holds: pgs_iters = solver_position_iteration_count 16 and vel_iters = solver_velocity_iteration_count 2
 * (cfg/task/QuadrupedPoseControl.yaml:41-42), the per-iteration drive impulse max_effort x dt (robot/base/robot.py:347-355), the joint speed
 * bound 450 deg/s (Design/Scripts/config_module_joints.py:11), max_depenetration_velocity 100 (YAML :50).  M is NOT augmented by the drive:
 * a drive row is the damper  dP = h kd (v* - qd - W dP)  over the iteration's share h = dt / pgs_iters of the step. */
static void advance_free(const real* Rf, real* pos, real* quat, const real* v, real h) {
  real ww[3], vw[3]; m3v(Rf, v, ww); m3v(Rf, v+3, vw);
  real th = sqrt(dot3(ww,ww))*h, dq[4];
  if (th < (real)1e-8) { dq[0]=1; for (int a=0;a<3;a++) dq[1+a]=(real)0.5*h*ww[a]; }
  else { real s=sin(th*(real)0.5)/(th/h); dq[0]=cos(th*(real)0.5); for (int a=0;a<3;a++) dq[1+a]=s*ww[a]; }
  real qn[4]; quat_mul(dq, quat, qn); real nn=sqrt(qn[0]*qn[0]+qn[1]*qn[1]+qn[2]*qn[2]+qn[3]*qn[3]);
  for (int a=0;a<4;a++) quat[a]=qn[a]/nn;
  for (int a=0;a<3;a++) pos[a]+=h*vw[a];
}

static void substep_tgs(const lmo_model* m, const lmo_params* p, real* phys, const real* target, real* tau_out) {
  dyn_t* D = (dyn_t*)malloc(sizeof(dyn_t));
  dyn_compute(m, p, phys, D);
  const int K = p->pgs_iters>0 ? p->pgs_iters : 1, F = p->tgs_flags;
  const real dt=(real)p->dt, h=dt/(real)K, kd=(real)p->kd, mu=(real)p->mu, cj=(real)p->joint_damping;
  real v[NU];
  if (p->mode==0) { m3Tv(D->R0, phys+10, v); m3Tv(D->R0, phys+7, v+3); }
  else { m3Tv(D->Rf, phys+47, v); m3Tv(D->Rf, phys+44, v+3); }
  for (int j=0;j<12;j++) v[6+j]=phys[25+j];
  real Jc[12][NU], phi[4]; contact_rows(p, D, Jc, phi);
  real L[NU][NU]; memcpy(L, D->M, sizeof(L)); real rhs[NU], acc[NU];
  for (int a=0;a<NU;a++) rhs[a]=-D->h[a];
  for (int j=0;j<12;j++) { real cjj=g_dr?g_dr->cj[j]:cj; L[6+j][6+j]+=dt*cjj; rhs[6+j]-=cjj*v[6+j]; }
  chol(L); chol_solve(L, rhs, acc); for (int a=0;a<NU;a++) v[a]+=dt*acc[a];      /* external forces over the whole step, once, before the iterations */
  real MiJ[12][NU], MiD[12][NU], Wc[12], Wd[12], lam[12], P[12], dq[12];
  for (int r=0;r<12;r++) { chol_solve(L, Jc[r], MiJ[r]); real s=0; for (int a=0;a<NU;a++) s+=Jc[r][a]*MiJ[r][a]; Wc[r]=s; lam[r]=0; }
  for (int j=0;j<12;j++) { real e[NU]; memset(e,0,sizeof(e)); e[6+j]=1; chol_solve(L, e, MiD[j]); Wd[j]=MiD[j][6+j]; P[j]=0; dq[j]=0; }
  real* pos  = (p->mode==0)? phys   : phys+37;
  real* quat = (p->mode==0)? phys+3 : phys+40;
  real* lin  = (p->mode==0)? phys+7 : phys+44;
  real* ang  = (p->mode==0)? phys+10: phys+47;
  const real lim0 = (real)p->drive_iter_impulse;
  for (int it=0; it<K+p->vel_iters; it++) {
    const int posit = it<K;
    if (F&32) for (int r=0;r<12;r++) lam[r]=0;                  /* every iteration its own impulses (no release of an earlier iteration's push) */
    for (int half=0; half<2; half++) {
      const int drives = (F&1) ? (half==0) : (half==1);
      if (!drives) {
        for (int ii=0;ii<4;ii++) {
          const int i=((it&1)&&!(F&256))?3-ii:ii, r=3*i;
          if (phi[i]>(real)1e2) continue;                      /* off the plate */
          real vn=0; for (int a=0;a<NU;a++) vn+=Jc[r][a]*v[a];
          real b;
          if (phi[i]>=0) b = phi[i]/((F&64) ? dt : (F&8) ? (dt-(real)(posit?it:K-1)*h) : h);          /* a gap may close, not more */
          else if (posit) { b=(real)p->baumgarte*phi[i]/h; if (b<-(real)p->max_depen_vel) b=-(real)p->max_depen_vel; }
          else b=0;                                                                   /* velocity iterations: no penetration bias */
          real ln=lam[r]-(vn+b)/Wc[r]; if (ln<0) ln=0; real d=ln-lam[r]; lam[r]=ln;
          for (int a=0;a<NU;a++) v[a]+=MiJ[r][a]*d;
          real vt1=0, vt2=0; for (int a=0;a<NU;a++) { vt1+=Jc[r+1][a]*v[a]; vt2+=Jc[r+2][a]*v[a]; }
          real l1=lam[r+1]-vt1/Wc[r+1], l2=lam[r+2]-vt2/Wc[r+2]; const real lm=mu*lam[r];
          real n2=l1*l1+l2*l2; if (n2>lm*lm) { real sc=lm/sqrt(n2); l1*=sc; l2*=sc; }
          real d1=l1-lam[r+1], d2=l2-lam[r+2]; lam[r+1]=l1; lam[r+2]=l2;
          for (int a=0;a<NU;a++) v[a]+=MiJ[r+1][a]*d1+MiJ[r+2][a]*d2;
        }
      } else if (F&512) {
        /* the 12 drive rows as ONE block: (I / (h kd) + W_dd) dP = v* - qd, solved exactly, then each row's impulse bounded */
        real A[NU][NU], b12[NU], x[NU]; memset(A,0,sizeof(A)); memset(b12,0,sizeof(b12));
        for (int j=0;j<12;j++) { for (int k=0;k<12;k++) A[j][k]=MiD[k][6+j]; A[j][j]+=1/(h*kd); b12[j]=target[j]-v[6+j]; }
        for (int j=12;j<NU;j++) A[j][j]=1;
        chol(A); chol_solve(A, b12, x);
        for (int j=0;j<12;j++) { real lim = g_dr ? g_dr->tmax[j]*dt : lim0, dP=x[j];
          if (lim>0) { if (dP>lim) dP=lim; if (dP<-lim) dP=-lim; }
          P[j]+=dP; for (int a=0;a<NU;a++) v[a]+=MiD[j][a]*dP; }
      } else {
        for (int jj=0;jj<12;jj++) {
          static const int limb_order[12]={0,4,5,1,6,7,2,8,9,3,10,11};
          const int j=(F&128)?limb_order[jj]:jj;
          real lim = g_dr ? g_dr->tmax[j]*dt : lim0;
          real dP = h*kd*(target[j]-v[6+j])/(1+h*kd*Wd[j]);
          if (lim>0) {
            if (F&16) { real Pn=P[j]+dP, cap=lim*(real)K; if (Pn>cap) Pn=cap; if (Pn<-cap) Pn=-cap; dP=Pn-P[j]; }
            else { if (dP>lim) dP=lim; if (dP<-lim) dP=-lim; }
          }
          P[j]+=dP; for (int a=0;a<NU;a++) v[a]+=MiD[j][a]*dP;
        }
      }
    }
    for (int j=0;j<12;j++) {                                    /* joint speed bound, every iteration */
      real vm=g_dr?g_dr->vmax[j]:(real)p->max_joint_vel, x=v[6+j], c=x>vm?vm:(x<-vm?-vm:x);
      if (c!=x) { if (F&4) { real dP=(c-x)/Wd[j]; for (int a=0;a<NU;a++) v[a]+=MiD[j][a]*dP; } else v[6+j]=c; }
    }
    if (posit && !(F&2)) {                                       /* TGS: gaps, joint angles and the free body advance by the iteration's share */
      for (int i=0;i<4;i++) { real vn=0; for (int a=0;a<NU;a++) vn+=Jc[3*i][a]*v[a]; phi[i]+=h*vn; }
      for (int j=0;j<12;j++) dq[j]+=h*v[6+j];
      advance_free(D->Rf, pos, quat, v, h);
    }
  }
  if (F&2) { for (int j=0;j<12;j++) dq[j]=dt*v[6+j]; advance_free(D->Rf, pos, quat, v, dt); }
  if (tau_out) for (int j=0;j<12;j++) tau_out[j]=P[j]/dt;
  for (int j=0;j<12;j++) { phys[25+j]=v[6+j]; phys[13+j]+=dq[j]; }
  { real Rn[9], ww[3], vw[3]; m3v(D->Rf, v, ww); quat_to_mat(quat, Rn); m3v(Rn, v+3, vw);
    for (int a=0;a<3;a++) { ang[a]=ww[a]; lin[a]=vw[a]; } }
  free(D);
}

/* one env, one sub-step: the contact problem (W 12x12, vf 12, bn 4) of its last pass and the impulses found (solver studies) */
void lmo_contact_problem(const lmo_model* m, const lmo_params* p, const real* phys, const real* target, real* W, real* vf, real* bn, real* lam) {
  real ph[LMO_PHYS]; memcpy(ph, phys, sizeof(ph));
  g_cap_W=W; g_cap_vf=vf; g_cap_bn=bn; g_cap_lam=lam;
  substep_one(m, p, ph, target, NULL);
  g_cap_W=0; g_cap_vf=0; g_cap_bn=0; g_cap_lam=0;
}

void lmo_substep(const lmo_model* m, const lmo_params* p, int N, real* phys, const real* targets) {
  #pragma omp parallel for schedule(static)
  for (int e=0;e<N;e++) substep_one(m, p, phys+(size_t)e*LMO_PHYS, targets+(size_t)e*12, NULL);
}

void lmo_substep_tau(const lmo_model* m, const lmo_params* p, int N, real* phys, const real* targets, real* tau) {
  #pragma omp parallel for schedule(static)
  for (int e=0;e<N;e++) substep_one(m, p, phys+(size_t)e*LMO_PHYS, targets+(size_t)e*12, tau+(size_t)e*12);
}

/* ------------------------------------------------------------------ task layer */
static void quat_rotate_inverse(const real* q, const real* v, real* o) { real R[9]; quat_to_mat(q,R); m3Tv(R,v,o); }

void lmo_task_eval(const lmo_params* p, int N, const real* readback, const real* actions,
                   real* task, int64_t* cnt, real* obs, real* states, real* rew, real* terms) {
  #pragma omp parallel for schedule(static)
  for (int e=0;e<N;e++) {
    const real* rb=readback+(size_t)e*LMO_READBACK; const real* act=actions+(size_t)e*12;
    real* tk=task+(size_t)e*LMO_TASK; int64_t* c=cnt+(size_t)e*LMO_CNT;
    real* ob=obs+(size_t)e*p->num_obs; real* st=states+(size_t)e*93; real* tr=terms+(size_t)e*LMO_TERMS;
    const int var1=(p->variant==1), var2=(p->variant==2), pd=(p->variant>=1); const real* torque=rb+87; real* se=tk+40; real* last_tgt=tk+52;
    real tgtq[12];   /* current joint position targets from the swing/extension targets (:267-276) */
    for (int l=0;l<4;l++) { tgtq[l]=se[l]; tgtq[4+2*l]=se[4+2*l]+se[5+2*l]/2; tgtq[5+2*l]=se[4+2*l]-se[5+2*l]/2; }
    const real *q=rb, *qd=rb+12, *acc=rb+24, *bp=rb+36, *bq=rb+39, *lv=rb+43, *av=rb+46, *tips=rb+49, *knees=rb+61;
    real* last_act=tk; real* last_tip=tk+24; real* goal=tk+36;
    /* post_physics_step: progress_buf += 1 (rl_task.py:251) */
    c[4]+=1;
    real opos[3], oquat[4], olin[3], oang[3], btip[12];
    real Rr[9], pr[3];
    if (p->mode==0) {
      /* quadruped_pose_control.py:319-335 */
      real nb[3]={-bp[0],-bp[1],-bp[2]}, nl[3]={-lv[0],-lv[1],-lv[2]}, na[3]={-av[0],-av[1],-av[2]};
      quat_rotate_inverse(bq, nb, opos); quat_conj(bq, oquat); quat_rotate_inverse(bq, nl, olin); quat_rotate_inverse(bq, na, oang);
      quat_to_mat(bq, Rr); for (int i=0;i<3;i++) pr[i]=bp[i];
    } else {
      /* quadruped_manipulate_plate.py:323-343 */
      real qr[4]; for (int i=0;i<4;i++) qr[i]=(real)p->fixed_base_quat[i]; for (int i=0;i<3;i++) pr[i]=(real)p->fixed_base_pos[i];
      quat_to_mat(qr, Rr);
      real d[3]={bp[0]-pr[0],bp[1]-pr[1],bp[2]-pr[2]}; m3Tv(Rr, d, opos);
      real qc[4]; quat_conj(qr, qc); quat_mul(qc, bq, oquat); if (oquat[0]<0) for (int i=0;i<4;i++) oquat[i]=-oquat[i];
      m3Tv(Rr, lv, olin); m3Tv(Rr, av, oang);
    }
    for (int i=0;i<4;i++) { real d[3]={tips[3*i]-pr[0],tips[3*i+1]-pr[1],tips[3*i+2]-pr[2]}; m3Tv(Rr, d, btip+3*i); }
    real gc[4], qdiff[4]; quat_conj(goal, gc); quat_mul(oquat, gc, qdiff);
    real qdf[4]; for (int i=0;i<4;i++) qdf[i]=(qdiff[0]<0 && !var1)?-qdiff[i]:qdiff[i];   /* the custom-controller tasks do not flip the sign (:429-431) */
    real Ro[9]; quat_to_mat(oquat, Ro); real up[3]={Ro[2],Ro[5],Ro[8]};
    int k=0;
    for (int i=0;i<3;i++) ob[k++]=(real)p->s_pos*opos[i];
    for (int i=0;i<3;i++) ob[k++]=up[i];
    for (int i=0;i<4;i++) ob[k++]=qdf[i];
    for (int i=0;i<3;i++) ob[k++]=(real)p->s_lin*olin[i];
    for (int i=0;i<3;i++) ob[k++]=(real)p->s_ang*oang[i];
    for (int i=0;i<12;i++) ob[k++]=(real)p->s_q*q[i];
    for (int i=0;i<12;i++) ob[k++]=(real)p->s_qd*qd[i];
    if (!var2) { for (int i=0;i<12;i++) ob[k++]=act[i]; for (int i=0;i<12;i++) ob[k++]=last_act[i]; }
    else { for (int i=0;i<12;i++) ob[k++]=(real)0.3*tgtq[i]; for (int i=0;i<12;i++) ob[k++]=(real)0.3*last_tgt[i]; }   /* position-control tasks:
                                                   the joint position targets replace the actions (quadruped_pose_control_position_control.py:438-453) */
    if (var1) { for (int i=0;i<12;i++) ob[k++]=(real)0.3*tgtq[i]; for (int i=0;i<12;i++) ob[k++]=(real)0.3*last_tgt[i]; }   /* :432-455 */
    k=0;
    for (int i=0;i<3;i++) st[k++]=(real)p->s_pos*opos[i];
    for (int i=0;i<3;i++) st[k++]=(real)p->s_lin*olin[i];
    for (int i=0;i<4;i++) st[k++]=oquat[i];
    for (int i=0;i<3;i++) st[k++]=(real)p->s_ang*oang[i];
    for (int i=0;i<12;i++) st[k++]=(real)p->s_q*q[i];
    for (int i=0;i<12;i++) st[k++]=(real)p->s_qd*qd[i];
    for (int i=0;i<4;i++) st[k++]=goal[i];
    for (int i=0;i<4;i++) st[k++]=qdf[i];
    for (int i=0;i<12;i++) st[k++]=btip[i];
    for (int i=0;i<12;i++) st[k++]=last_tip[i];
    for (int i=0;i<12;i++) st[k++]=act[i];
    for (int i=0;i<12;i++) st[k++]=last_act[i];
    for (int i=0;i<12;i++) last_tip[i]=btip[i];
    /* calculate_metrics (quadruped_pose_control.py:428-548) */
    real vn=sqrt(qdiff[1]*qdiff[1]+qdiff[2]*qdiff[2]+qdiff[3]*qdiff[3]); if (vn>1) vn=1;
    real rot_dist=2*asin(vn);
    real rot_rew=(real)p->quat_scale/(fabs(rot_dist)+(real)p->rot_eps);
    real trans=sqrt(opos[0]*opos[0]+opos[1]*opos[1])*(real)p->trans_scale;
    real accp=0, rate=0; for (int i=0;i<12;i++){ accp+=fabs(acc[i])*(real)p->acc_scale; rate+=var1?fabs(act[i]):fabs(last_act[i]-act[i]); } rate*=(real)p->rate_scale;
    real powp=0, terr=0, rdec=0;
    if (var1) {   /* :530-545 */
      for (int i=0;i<12;i++){ powp+=fabs(torque[i]*qd[i]); terr+=fabs(last_tgt[i]-q[i]); }
      powp*=(real)p->power_scale; terr*=(real)p->target_err_scale;
      rdec=((rot_dist>(real)p->rot_dec_thresh)?(real)1:(real)0)*(tk[64]-rot_dist)*(real)p->rot_dec_scale; tk[64]=rot_dist;
    }
    int64_t cgr=(c[1]>p->max_consec)?1:0;
    real bonus=(real)p->bonus*(real)cgr;
    int64_t succ=(fabs(rot_dist)<=(real)p->succ_thresh)?1:0;
    int brk=0, rst=0;
    for (int l=0;l<4;l++) {
      real dd=fabs(q[5+2*l]-q[4+2*l]);
      if (dd<(real)p->d23_pen[0]||dd>(real)p->d23_pen[1]) brk++;
      if (dd<(real)p->d23_rst[0]||dd>(real)p->d23_rst[1]) rst++;
      if (q[l]<(real)p->d1_pen[l][0]||q[l]>(real)p->d1_pen[l][1]) brk++;
      if (q[l]<(real)p->d1_rst[l][0]||q[l]>(real)p->d1_rst[l][1]) rst++;
    }
    real limp=(brk>0)?(real)p->limit_pen:0;
    real total=rot_rew+trans+accp+rate+bonus+limp+powp+terr+rdec;
    c[2]=cgr;
    int64_t both=(succ&&c[0])?1:0;
    if (!both) c[1]=0; else c[1]=c[1]+1;
    if (c[0]==0 && succ==1) c[1]=1;
    c[0]=succ;
    for (int i=0;i<12;i++) last_act[i]=act[i];
    /* is_done (quadruped_pose_control.py:562-616 / quadruped_manipulate_plate.py:569-640) */
    int64_t reset=c[3];
    if (opos[2]>0) reset=1;
    {
      /* heights are measured in the "object" frame: ground (identity) for loco, plate for mani */
      real Rp[9], pp[3];
      if (p->mode==0) { for (int i=0;i<9;i++) Rp[i]=(i%4==0)?1:0; pp[0]=pp[1]=pp[2]=0; }
      else { quat_to_mat(bq, Rp); for (int i=0;i<3;i++) pp[i]=bp[i]; }
      real d[3], o[3];
      for (int i=0;i<3;i++) d[i]=pr[i]-pp[i];
      m3Tv(Rp,d,o); if (o[2]<=(real)p->h_base) reset=1;
      int nc=0, nk=0;
      for (int i=0;i<4;i++) { real cb[3], cw[3];
        for (int a=0;a<3;a++) cb[a]=(real)p->corner[i][a];
        m3v(Rr, cb, cw); for (int a=0;a<3;a++) d[a]=cw[a]+pr[a]-pp[a];
        m3Tv(Rp,d,o); if (o[2]<(real)p->h_corner) nc++; }
      for (int i=0;i<8;i++) { for (int a=0;a<3;a++) d[a]=knees[3*i+a]-pp[a];
        m3Tv(Rp,d,o); if (o[2]-(real)p->h_knee<=0) nk++; }
      if (nc>0) reset=1;
      if (nk>0) reset=1;
    }
    if (rst>0) reset=1;
    real fallp=(real)p->fall_pen*(real)reset;
    total+=fallp;
    if (cgr==1) reset=1;
    if (c[4]>=p->max_episode-1) reset=1;
    c[3]=reset;
    rew[e]=total;
    tr[0]=rot_rew; tr[1]=trans; tr[2]=accp; tr[3]=rate; tr[4]=bonus; tr[5]=limp; tr[6]=fallp; tr[7]=(real)cgr; tr[8]=powp; tr[9]=terr; tr[10]=rdec;
    if (pd && p->cc_update_last_tgt) for (int i=0;i<12;i++) last_tgt[i]=tgtq[i];     /* :723-725, end of is_done */
  }
}

/* ------------------------------------------------------------------ reset + full step */
void lmo_reset(const lmo_params* p, int N, real* phys, real* task, int64_t* cnt,
               const real* goal_rand, uint32_t seed) {
  for (int e=0;e<N;e++) {
    int64_t* c=cnt+(size_t)e*LMO_CNT; if (c[3]==0) continue;
    real* ph=phys+(size_t)e*LMO_PHYS; real* tk=task+(size_t)e*LMO_TASK;
    /* quadruped_pose_control.py:230-299 */
    real u3[3];
    if (goal_rand) { for (int i=0;i<3;i++) u3[i]=goal_rand[3*e+i]; }
    else lmo_hash_uniform3(seed, (uint32_t)e, (uint32_t)c[5], u3);
    real eul[3]; for (int i=0;i<3;i++) eul[i]=(real)p->goal_lo[i]+((real)p->goal_hi[i]-(real)p->goal_lo[i])*u3[i];
    lmo_quat_from_euler(eul[0], eul[1], eul[2], tk+36);
    for (int i=0;i<12;i++) { ph[13+i]=(real)p->init_q[i]; ph[25+i]=0; tk[i]=0; tk[12+i]=0; tk[24+i]=(real)p->default_tip[i]; }
    for (int i=0;i<3;i++) { ph[i]=(real)p->init_base_pos[i]; ph[7+i]=0; ph[10+i]=0; ph[37+i]=(real)p->init_plate_pos[i]; ph[44+i]=0; ph[47+i]=0; }
    for (int i=0;i<4;i++) { ph[3+i]=(real)p->init_base_quat[i]; ph[40+i]=(real)p->init_plate_quat[i]; }
    if (p->variant>=1) {   /* :371-384 */
      for (int i=0;i<12;i++) { tk[40+i]=(real)p->init_se[i]; tk[52+i]=(real)p->init_q[i]; }
      real qb[4]={(real)p->init_base_quat[0],-(real)p->init_base_quat[1],-(real)p->init_base_quat[2],-(real)p->init_base_quat[3]}, gc[4], qd4[4];
      if (p->mode==1) { qb[0]=1; qb[1]=qb[2]=qb[3]=0; }
      quat_conj(tk+36, gc); quat_mul(qb, gc, qd4);
      real vn=sqrt(qd4[1]*qd4[1]+qd4[2]*qd4[2]+qd4[3]*qd4[3]); if (vn>1) vn=1; tk[64]=2*asin(vn);
    }
    c[0]=0; c[1]=0; c[2]=0; c[3]=0; c[4]=0; c[5]+=1;
  }
}

static void substeps_all(const lmo_model* m, const lmo_params* p, int N, real* phys, const real* targets, real* tau, const env_dr_t* drs) {
  #pragma omp parallel for schedule(static)
  for (int e=0;e<N;e++) { g_dr = drs ? drs+e : 0; substep_one(m, p, phys+(size_t)e*LMO_PHYS, targets+(size_t)e*12, tau?tau+(size_t)e*12:NULL); g_dr = 0; }
}

static void step_core(const lmo_model* m, const lmo_params* p, int N, real* phys, real* task, int64_t* cnt,
                      const real* actions, const real* goal_rand, uint32_t seed,
                      real* obs, real* states, real* rew, real* terms, const env_dr_t* drs) {
  lmo_reset(p, N, phys, task, cnt, goal_rand, seed);
  real* targets=(real*)malloc(sizeof(real)*12*(size_t)N);
  real* rb=(real*)malloc(sizeof(real)*LMO_READBACK*(size_t)N);
  real* tausum=(real*)calloc((size_t)N*12, sizeof(real));
  if (p->variant==0) {
    /* robot.py:452-454: velocity mode, unscale_transform(a, -lim, +lim) = a*lim */
    for (size_t i=0;i<(size_t)N*12;i++) targets[i]=actions[i]*(real)p->act_scale;
    if (p->drive_mode==1) {
      /* position mode (robot.py:448-450): q* = a * pi; tau = kp (q* - q) - kd qd = kd (v* - qd) with v* = kp/kd (q* - q), refreshed every sub-step */
      real* vt=(real*)malloc(sizeof(real)*12*(size_t)N);
      for (int s=0;s<p->substeps;s++) {
        for (int e=0;e<N;e++) for (int i=0;i<12;i++) vt[(size_t)e*12+i]=(real)p->pd_kp/(real)p->kd*(targets[(size_t)e*12+i]-phys[(size_t)e*LMO_PHYS+13+i]);
        substeps_all(m, p, N, phys, vt, NULL, drs);
      }
      free(vt);
    } else
    for (int s=0;s<p->substeps;s++) substeps_all(m, p, N, phys, targets, NULL, drs);
  } else {
    /* quadruped_pose_control_custom_controller.py:255-307: integrate swing/extension targets, PD torque every sub-step */
    real* tau=(real*)malloc(sizeof(real)*12*(size_t)N);
    for (int e=0;e<N;e++) { real* se=task+(size_t)e*LMO_TASK+40;
      for (int i=0;i<12;i++) { real v=se[i]+actions[(size_t)e*12+i]*(real)p->act_scale_se; if (v<(real)p->se_lo[i]) v=(real)p->se_lo[i]; if (v>(real)p->se_hi[i]) v=(real)p->se_hi[i]; se[i]=v; } }
    for (int s=0;s<p->substeps;s++) {
      /* update_joint_states() runs after every in-task sub-step (…custom_controller.py:296-297), so the joint acceleration the reward sees
         spans only the trailing acc_substeps (= controlFrequencyInv) sub-steps: remember the velocity they start from (robot.py:289-291) */
      if (s==p->substeps-p->acc_substeps) for (int e=0;e<N;e++) for (int i=0;i<12;i++) task[(size_t)e*LMO_TASK+12+i]=phys[(size_t)e*LMO_PHYS+25+i];
      for (int e=0;e<N;e++) { const real* se=task+(size_t)e*LMO_TASK+40; const real* q=phys+(size_t)e*LMO_PHYS+13; real* tg=targets+(size_t)e*12; real qs[12];
        for (int l=0;l<4;l++) { qs[l]=se[l]; qs[4+2*l]=se[4+2*l]+se[5+2*l]/2; qs[5+2*l]=se[4+2*l]-se[5+2*l]/2; }
        /* tau = kp (q* - q) - kd qd  ==  kd (v* - qd)  with  v* = kp/kd (q* - q): the velocity-drive solver with a position-derived target */
        for (int i=0;i<12;i++) tg[i]=(real)p->pd_kp/(real)p->kd*(qs[i]-q[i]); }
      substeps_all(m, p, N, phys, targets, tau, drs);
      for (size_t i=0;i<(size_t)N*12;i++) tausum[i]+=tau[i];
    }
    free(tau);
  }
  #pragma omp parallel for schedule(static)
  for (int e=0;e<N;e++) {
    real* ph=phys+(size_t)e*LMO_PHYS; real* tk=task+(size_t)e*LMO_TASK; real* r=rb+(size_t)e*LMO_READBACK;
    for (int i=0;i<12;i++) { r[i]=ph[13+i]; r[12+i]=ph[25+i];
      r[24+i]=(ph[25+i]-tk[12+i])/((p->variant==0)?(real)p->ctrl_dt:(real)(p->dt*p->acc_substeps));   /* robot.py:290 */
      tk[12+i]=ph[25+i]; }
    const real* src=(p->mode==0)?ph:ph+37;
    for (int i=0;i<13;i++) r[36+i]=src[i];
    lmo_fk(m, p, ph, r+49, r+61); r[85]=r[86]=0;
    for (int i=0;i<12;i++) r[87+i]=(p->variant>=1)?tausum[(size_t)e*12+i]/(real)p->torque_div:0;
  }
  lmo_task_eval(p, N, rb, actions, task, cnt, obs, states, rew, terms);
  free(targets); free(rb); free(tausum);
}

void lmo_step(const lmo_model* m, const lmo_params* p, int N, real* phys, real* task, int64_t* cnt,
              const real* actions, const real* goal_rand, uint32_t seed,
              real* obs, real* states, real* rew, real* terms) {
  step_core(m, p, N, phys, task, cnt, actions, goal_rand, seed, obs, states, rew, terms, NULL);
}

/* one randomised physics attribute of one env: on_interval entries are redrawn every `interval` control steps
 * (key = dr_step / interval), on_reset entries at the env's last gated reset (key = dr_reset_key; 0 = never randomised) */
static real dr_attr(const lmo_dr_channel* ch, uint32_t seed, uint32_t stream, int e, const int64_t* d, int idx, int comp, real base) {
  if (!ch->enabled) return base;
  int64_t key = ch->interval > 0 ? d[2] / ch->interval : d[4];
  if (ch->interval == 0 && key == 0) return base;
  real n = lmo_dr_sample(seed, stream, g_env0 + (uint32_t)e, (uint32_t)key, (uint32_t)idx, ch->distribution, (real)ch->p0[comp], (real)ch->p1[comp]);
  return dr_apply(ch->operation, base, n);
}

void lmo_step_dr(const lmo_model* m, const lmo_params* p, int N, real* phys, real* task, int64_t* cnt, int64_t* drc,
                 const real* actions_raw, real clip_actions, const real* goal_rand, uint32_t seed,
                 real* obs, real* states, real* rew, real* terms, real* actions_used, real* physdr) {
  int64_t* flags=(int64_t*)calloc((size_t)N*4, sizeof(int64_t)); int64_t *rf=flags, *key=flags+N, *stepk=flags+2*(size_t)N, *cntr=flags+3*(size_t)N;
  real* act=(real*)malloc(sizeof(real)*12*(size_t)N); memcpy(act, actions_raw, sizeof(real)*12*(size_t)N);
  env_dr_t* drs=(env_dr_t*)malloc(sizeof(env_dr_t)*(size_t)N);
  /* 1. action noise on the raw actions, then the clipActions clamp (vec_env_rlgames.py:56-60) */
  for (int e=0;e<N;e++) { rf[e]=cnt[(size_t)e*LMO_CNT+3]!=0; key[e]=cnt[(size_t)e*LMO_CNT+5]+rf[e]; stepk[e]=drc[(size_t)e*LMO_DR_CNT+2]; cntr[e]=drc[(size_t)e*LMO_DR_CNT+1]; }
  lmo_dr_noise(&p->dr[LMO_DR_ACT_RESET], &p->dr[LMO_DR_ACT_INTERVAL], seed, LMO_DR_ACT_RESET, N, 12, act, rf, cntr, key, stepk);
  for (int e=0;e<N;e++) drc[(size_t)e*LMO_DR_CNT+1]=cntr[e];
  for (size_t i=0;i<(size_t)N*12;i++) { if (act[i]>clip_actions) act[i]=clip_actions; if (act[i]<-clip_actions) act[i]=-clip_actions; }
  if (actions_used) memcpy(actions_used, act, sizeof(real)*12*(size_t)N);
  /* 2. gated on_reset randomisation (quadruped_pose_control.py:224-228) and 3. this control step's physics attributes */
  for (int e=0;e<N;e++) {
    int64_t* d=drc+(size_t)e*LMO_DR_CNT;
    if (rf[e] && d[3]>=p->dr_min_frequency) { d[4]=key[e]; d[3]=0; }
    env_dr_t* x=drs+e;
    const real g0[3]={0,0,-(real)p->gravity};
    for (int c=0;c<3;c++) { x->g[c]=dr_attr(&p->dr[LMO_DR_GRAVITY], seed, LMO_DR_GRAVITY, e, d, c, c, g0[c]);
                            x->f[c]=dr_attr(&p->dr[LMO_DR_BASE_FORCE], seed, LMO_DR_BASE_FORCE, e, d, c, c, 0); }
    for (int j=0;j<12;j++) { x->tmax[j]=dr_attr(&p->dr[LMO_DR_MAX_EFFORT], seed, LMO_DR_MAX_EFFORT, e, d, j, 0, (real)p->tau_max);
                             x->vmax[j]=dr_attr(&p->dr[LMO_DR_MAX_VELOCITY], seed, LMO_DR_MAX_VELOCITY, e, d, j, 0, (real)p->max_joint_vel);
                             x->cj[j]=dr_attr(&p->dr[LMO_DR_JOINT_DAMPING], seed, LMO_DR_JOINT_DAMPING, e, d, j, 0, (real)p->joint_damping); }
    if (physdr) { real* o=physdr+(size_t)e*42; for (int j=0;j<12;j++){o[j]=x->tmax[j];o[12+j]=x->vmax[j];o[30+j]=x->cj[j];} for (int c=0;c<3;c++){o[24+c]=x->g[c];o[27+c]=x->f[c];} }
  }
  /* 4./5. the step itself */
  step_core(m, p, N, phys, task, cnt, act, goal_rand, seed, obs, states, rew, terms, drs);
  /* 6. observation noise in place on obs_buf, keyed by the flags is_done has just written (vec_env_rlgames.py:70-72) */
  for (int e=0;e<N;e++) { rf[e]=cnt[(size_t)e*LMO_CNT+3]!=0; key[e]=cnt[(size_t)e*LMO_CNT+5]+rf[e]; cntr[e]=drc[(size_t)e*LMO_DR_CNT+0]; }
  lmo_dr_noise(&p->dr[LMO_DR_OBS_RESET], &p->dr[LMO_DR_OBS_INTERVAL], seed, LMO_DR_OBS_RESET, N, p->num_obs, obs, rf, cntr, key, stepk);
  for (int e=0;e<N;e++) { int64_t* d=drc+(size_t)e*LMO_DR_CNT; d[0]=cntr[e]; d[2]+=1; d[3]+=1; }
  free(flags); free(act); free(drs);
}
