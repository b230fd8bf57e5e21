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
// the engine's current goal / domain-randomisation seed (lm_set_seed changes it; a captured hipGraph holds the value it was captured with)
uint32_t lm_internal_seed(const lm_engine* h);
// records a message for lm_last_error() (thread-local, lm_engine.hip) and returns `code`
int lm_internal_fail(int code, const char* msg);
