// lm_policy.hip -- policy forward passes on the matrix cores (gfx950): the GNN (below), the MLP, the gaussian action sampling
// fused into their epilogues, and the fused rollout (forward -> sampling -> lm_step, T times, one hipGraph) of include/lm_policy.h.
//
// GNN:
// Restates RobotLearning/omniisaacgymenvs/scripts/graph_model_orebot_ov.py:11-241 (GraphNet hidden 32, 13 nodes,
// 24 directed edges, 3 message-passing layers with max aggregation, Action_Layer / Value_Layer heads) as one kernel:
//   obs (B,64)  ->  action means (B,12) in node order [dof1 a1..a4, dof2 a1..a4, dof3 a1..a4], value (B,1).
//
// Mapping: one block = 16 samples on the four wavefronts of a CU (node ownership, see gnn_body).  Every dense product runs as  D(features x samples) = W(features x K) * X(K x samples)
// with the weights as the A operand and the activations as the B operand, so an accumulator tile (feature rows in the 4 registers / 4 lane
// groups, sample on the lane) feeds the next product's B operand with no lane movement: only the k order inside the dot product is
// permuted, and the A operand is gathered in the same permuted order.  Rounds 1-3 ran the 32-wide products on v_mfma_f32_16x16x4_f32 (8
// instructions each); since round 4 they run on v_mfma_f32_16x16x32_f16 (ONE instruction per K = 32) with every fp32 operand split into two fp16
// halves and the four half products accumulated in fp32 (lm_policy_dev.h split_f16 / mfma_split: fp32-level accuracy, 16 x the matrix rate);
// the two small input layers (K = 16 and K = 4) stay on the fp32 instruction.  W1 [h_i || h_j] is split into per-node products P = W1a h + b1 (as target) and Q = W1b h (as source),
// staged once per layer in LDS, so the 24 edge messages need only the 32x32 second linear layer:
//   m_e = ELU(W2 ELU(P[tgt] + Q[src]) + b2),  h'[tgt] = max_e m_e.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/lm_policy.h"
#include "../../include/lm_engine.h"
#include "lm_rng.h"
#include "lm_policy_dev.h"
#include "lm_internal.h"
#include <new>

#ifdef LM_GNN_STAMPS
__device__ unsigned long long lm_gnn_stamp_out[512 * 64];
extern "C" int lm_debug_gnn_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(lm_gnn_stamp_out), sizeof(lm_gnn_stamp_out)) == hipSuccess ? 0 : -1; }
#endif
__global__ void __launch_bounds__(256) k_gnn_forward(const float* __restrict__ obs, int B, const float* __restrict__ W,
                                                     float* __restrict__ mean, float* __restrict__ value, SampleArgs SA) {
  __shared__ GnnSmem G;
#ifdef LM_GNN_STAMPS
  if (threadIdx.x < 64) reinterpret_cast<unsigned long long*>(lm_gnn_stamp_lds)[threadIdx.x] = 0;
  __syncthreads();
  { unsigned long long t0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_) :: "memory"); if ((threadIdx.x & 63) == 0) lm_gnn_stamp_lds[threadIdx.x >> 6][15] = t0_; }
#endif
  gnn_block<false>(obs, 0.f, B, blockIdx.x * GNN_SAMPLES, W, mean, value, SA, G, threadIdx.x);
#ifdef LM_GNN_STAMPS
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x < 512) lm_gnn_stamp_out[blockIdx.x * 64 + threadIdx.x] = reinterpret_cast<unsigned long long*>(lm_gnn_stamp_lds)[threadIdx.x];
#endif
}

// MLP policy forward (device code in lm_policy_dev.h): one block = 16 samples on the 4 wavefronts of a CU
template <int NOBS>
__global__ void __launch_bounds__(256) k_mlp_forward(const float* __restrict__ obs, int B, const float* __restrict__ W,
                                                     float* __restrict__ mean, float* __restrict__ value, SampleArgs SA) {
  __shared__ MlpSmem<NOBS> M;
#ifdef LM_GNN_STAMPS
  if (threadIdx.x < 64) reinterpret_cast<unsigned long long*>(lm_gnn_stamp_lds)[threadIdx.x] = 0;
  __syncthreads();
  { unsigned long long t0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_) :: "memory"); if ((threadIdx.x & 63) == 0) lm_gnn_stamp_lds[threadIdx.x >> 6][15] = t0_; }
#endif
  mlp_block<NOBS, false>(obs, 0.f, B, blockIdx.x * 16, W, mean, value, SA, M, threadIdx.x);
#ifdef LM_GNN_STAMPS
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x < 512) lm_gnn_stamp_out[blockIdx.x * 64 + threadIdx.x] = reinterpret_cast<unsigned long long*>(lm_gnn_stamp_lds)[threadIdx.x];
#endif
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
  if (reinterpret_cast<uintptr_t>(params) & 15) return lm_internal_fail(-1, "lm_gnn_forward: the parameter block must be 16-byte aligned (its weight rows are read with 16-byte loads)");
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
  float* mean_tmp; const int64_t* cnt; long long* acc_steps;      // acc_steps: [T][16] accumulators of the persistent kernel
  int n_cu; bool persistent_ok;                                   // compute units of the device; engine can run the persistent kernel
  hipGraphExec_t exec; hipStream_t exec_stream;
  uint64_t exec_env_key;                                          // engine seed + requested views baked into the captured kernel arguments
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
  if (!mean || !log_std || !cnt || !actions || !logp || n_envs <= 0) return lm_internal_fail(-1, "lm_sample_actions: null argument or n_envs <= 0");
  hipLaunchKernelGGL(k_sample_actions, dim3((n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, log_std, cnt, n_envs, seed, actions, logp);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int lm_rollout_create(lm_rollout** out, lm_engine* env, int policy, const float* policy_params, const float* log_std, int T, uint32_t noise_seed,
                      float* obs, float* actions, float* logp, float* values, float* rewards, int64_t* dones, float* extras) {
  if (!out || !env || !policy_params || !log_std || !obs || !actions || !logp || !values || !rewards || !dones || T <= 0) return lm_internal_fail(-1, "lm_rollout_create: null argument or T <= 0");
  if (policy != LM_POLICY_MLP && policy != LM_POLICY_GNN) return lm_internal_fail(-1, "lm_rollout_create: policy must be LM_POLICY_MLP or LM_POLICY_GNN");
  if (policy == LM_POLICY_GNN && (reinterpret_cast<uintptr_t>(policy_params) & 15)) return lm_internal_fail(-1, "lm_rollout_create: the GNN parameter block must be 16-byte aligned");
  const int nobs = lm_num_obs(env);
  if (nobs != 64 && !(nobs == 88 && policy == LM_POLICY_MLP)) return lm_internal_fail(-1, "lm_rollout_create: the GNN needs 64-wide observations (the MLP takes 64 or 88)");      // the GNN reads the 64-wide layout; the MLP also the 88-wide one
  lm_rollout* r = new (std::nothrow) lm_rollout();
  if (!r) return lm_internal_fail(-3, "lm_rollout_create: host allocation failed");
  r->env = env; r->policy = policy; r->T = T; r->N = lm_num_envs(env); r->nobs = nobs; r->seed = noise_seed;
  r->params = policy_params; r->log_std = log_std; r->obs = obs; r->actions = actions; r->logp = logp; r->values = values;
  r->rewards = rewards; r->dones = dones; r->extras = extras; r->exec = nullptr; r->exec_stream = nullptr;
  r->cnt = (const int64_t*)lm_ptr(env, LM_PTR_CNT);
  if (hipMalloc((void**)&r->mean_tmp, (size_t)r->N * 12 * sizeof(float)) != hipSuccess) { delete r; return lm_internal_fail(-2, "lm_rollout_create: hipMalloc failed"); }
  r->acc_steps = nullptr;
  { int dev = 0, cus = 0; r->n_cu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) ? cus : 0; }
  r->persistent_ok = lm_internal_rollout_supported(env, policy, nobs) != 0;
  if (hipMalloc((void**)&r->acc_steps, (size_t)T * 16 * sizeof(long long)) != hipSuccess ||
      hipMemset(r->acc_steps, 0, (size_t)T * 16 * sizeof(long long)) != hipSuccess) { (void)hipFree(r->mean_tmp); if (r->acc_steps) (void)hipFree(r->acc_steps); delete r; return lm_internal_fail(-2, "lm_rollout_create: hipMalloc / hipMemset failed"); }
  *out = r;
  return 0;
}

int lm_rollout_run(lm_rollout* r, int use_graph, void* stream) {
  if (!r) return lm_internal_fail(-1, "lm_rollout_run: null plan");
  hipStream_t s = (hipStream_t)stream;
  if (use_graph == LM_ROLLOUT_AUTO)      // a persistent block holds a whole CU (512 registers per lane): one resident generation of blocks, or the graph
    use_graph = (r->persistent_ok && (r->N + 15) / 16 <= r->n_cu) ? LM_ROLLOUT_PERSISTENT : LM_ROLLOUT_GRAPH;
  if (!use_graph) return rollout_enqueue(r, s);
  if (use_graph == LM_ROLLOUT_PERSISTENT) {
    // the whole rollout in one kernel (un-randomised engines); same results as the other two modes
    LmRolloutArgs R; R.params = r->params; R.log_std = r->log_std; R.obs = r->obs; R.actions = r->actions; R.logp = r->logp; R.values = r->values;
    R.rewards = r->rewards; R.extras = r->extras; R.dones = r->dones; R.acc_steps = r->acc_steps; R.T = r->T; R.nobs = r->nobs; R.noise_seed = r->seed;
    return lm_internal_rollout(r->env, r->policy, R, s);
  }
  if (r->exec && r->exec_env_key != lm_internal_args_key(r->env)) {      // lm_set_seed / a first lm_ptr() for a view since the capture: the graph's kernel arguments are stale
    (void)hipGraphExecDestroy(r->exec); r->exec = nullptr;
  }
  if (!r->exec) {
    // capture the 4T+1 launches once; every pointer in them is fixed for the lifetime of the plan
    r->exec_env_key = lm_internal_args_key(r->env);
    hipGraph_t g = nullptr;
    hipStream_t cs = s;
    bool own = false;
    if (cs == nullptr) { if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) return lm_internal_fail(-2, "lm_rollout_run: hipStreamCreate failed"); own = true; }   // the legacy stream cannot capture
    if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) { if (own) (void)hipStreamDestroy(cs); return lm_internal_fail(-2, "lm_rollout_run: stream capture could not begin (is the stream already capturing?)"); }
    int rc = rollout_enqueue(r, cs);
    hipError_t e = hipStreamEndCapture(cs, &g);
    if (own) (void)hipStreamDestroy(cs);
    if (rc || e != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); return lm_internal_fail(rc ? rc : -2, "lm_rollout_run: capturing the rollout failed"); }
    e = hipGraphInstantiate(&r->exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { r->exec = nullptr; return lm_internal_fail(-2, "lm_rollout_run: hipGraphInstantiate failed"); }
  }
  return hipGraphLaunch(r->exec, s) == hipSuccess ? 0 : lm_internal_fail(-2, "lm_rollout_run: hipGraphLaunch failed");
}

int lm_rollout_destroy(lm_rollout* r) {
  if (!r) return 0;
  if (r->exec) (void)hipGraphExecDestroy(r->exec);
  if (r->mean_tmp) (void)hipFree(r->mean_tmp);
  if (r->acc_steps) (void)hipFree(r->acc_steps);
  delete r;
  return 0;
}

}  // extern "C"
