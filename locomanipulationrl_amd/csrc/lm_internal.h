// lm_internal.h -- host-side hooks between lm_engine.hip and lm_policy.hip (one shared library; not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct lm_engine;

// buffers of a rollout plan (include/lm_policy.h lm_rollout_create), as the persistent rollout kernel needs them
struct LmRolloutArgs {
  const float* params; const float* log_std;
  float *obs, *actions, *logp, *values, *rewards, *extras; int64_t* dones;
  long long* acc_steps;        // [T][16] int64 accumulators, zero on entry; left zero
  int T, nobs; uint32_t noise_seed;
};

// T x (policy forward -> sampling -> step) + the bootstrap forward in ONE kernel launch (lm_engine.hip); policy = LM_POLICY_MLP / _GNN.
// Returns 0, or a negative code when the engine cannot run it (domain-randomised engines; the GNN on 88-wide observations).
int lm_internal_rollout(lm_engine* h, int policy, const LmRolloutArgs& R, hipStream_t s);
// 1 when lm_internal_rollout can run this engine / policy / observation width
int lm_internal_rollout_supported(const lm_engine* h, int policy, int nobs);
// what a captured lm_step launch has baked into its kernel arguments and the engine may change afterwards: the goal / domain-randomisation seed
// (lm_set_seed) and which of the unclipped views are kept current (lm_ptr).  A hipGraph captured under another key is re-captured.
uint64_t lm_internal_args_key(const lm_engine* h);
// records a message for lm_last_error() (thread-local, lm_engine.hip) and returns `code`
int lm_internal_fail(int code, const char* msg);
