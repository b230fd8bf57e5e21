/*
 * lm_policy.h -- C ABI of the policy forward passes (GNN, MLP), the action sampling and the fused rollout (part of liblm_engine.so).
 *
 * Replaces, for inference, the torch modules of RobotLearning/omniisaacgymenvs/scripts/graph_model_orebot_ov.py
 * (GraphNet :82-110, GraphLayer :11-80, Action_Layer :215-226, Value_Layer :228-241) as instantiated by
 * scripts/skrl_ppo_locomanipulation_vertical.py:44-49 (hidden_features = out_features = 32).
 */
#ifndef LM_POLICY_H
#define LM_POLICY_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Number of floats in the packed parameter block:
 *   input_layer1.weight (32,16) .bias (32) | input_layer2.weight (32,4) .bias (32) |
 *   3 x { linear1.weight (32,64) .bias (32) | linear2.weight (32,32) .bias (32) } |
 *   action_layer.weight (32) .bias (1) | value action_layer.weight (32) .bias (1)          (all row-major, torch layout) |
 *   observation preprocessor: mean (64) | 1/(sqrt(var)+eps) (64) | clip (1)   (identity: 0, 1, +inf) */
int lm_gnn_param_count(void);

/* obs: device float [batch][64] (the env's observation layout, quadruped_pose_control.py:358-371);
 * params: device float [lm_gnn_param_count()]; mean: device float [batch][12] (node order = dof1 a1..a4, dof2 a1..a4,
 * dof3 a1..a4); value: device float [batch].  Returns 0, -1 (bad argument) or -2 (launch failure). */
int lm_gnn_forward(const float* obs, int batch, const float* params, float* mean, float* value, void* stream);

/* MLP policy of the locomotion / manipulation scripts (scripts/skrl_ppo_locomotion.py:30-40: shared trunk
 * Linear(64,256) ELU Linear(256,128) ELU Linear(128,64) ELU, mean_layer Linear(64,12), value_layer Linear(64,1)) with the
 * observation preprocessor (skrl RunningStandardScaler, :96-99) folded in.  `params` is the block produced by
 * locomanipulationrl_amd.policies.mlp_model.pack_mlp_params: biases and the scaler as floats; every weight matrix as 32-bit words
 * [out/16][K-blocks of 32][64 lanes][8] in the operand order of v_mfma_f32_16x16x32_f16, each weight split into two fp16 halves
 * (w = hi + lo; words 0..3 the hi halves, 4..7 the lo halves) - the tile computes fp32 products as four half products, fp32-level accuracy
 * (layout and column map: csrc/lm_policy_dev.h; round 4 - rounds 1-3 packed plain floats in v_mfma_f32_16x16x4_f32 order, same word counts for
 * num_obs 64).  lm_mlp_param_count() is its length.  Outputs as lm_gnn_forward (mean in the env's action order). */
int lm_mlp_param_count(void);
int lm_mlp_forward(const float* obs, int batch, const float* params, float* mean, float* value, void* stream);
/* The same network on the 88-wide observation of the custom-controller tasks (…custom_controller.py:432-455) or the 64-wide one:
 * num_obs in {64, 88}; the packed block is  mean num_obs | 1/std num_obs | clip (+3 pad) | W1q 256 x (num_obs padded to a multiple of 32) | ... as above. */
int lm_mlp_param_count_obs(int num_obs);
int lm_mlp_forward_obs(const float* obs, int batch, int num_obs, const float* params, float* mean, float* value, void* stream);

/* ---- fused rollout: policy forward -> gaussian action sampling -> lm_step, T times, as ONE hipGraph launch.
 * Replaces the per-step Python of the reference's trainer loop (skrl SequentialTrainer / scripts/random_policy.py:52-61:
 * agent.act -> env.step -> agent.record_transition) for the duration of a rollout, during which the policy is constant. */
struct lm_engine;
typedef struct lm_rollout lm_rollout;
enum { LM_POLICY_MLP = 0, LM_POLICY_GNN = 1 };

/* Gaussian policy sampling (skrl GaussianMixin.act): actions = mean + exp(log_std) * eps, logp = sum_j log N(actions_j; mean_j, std_j).
 * eps is a counter-based normal keyed by (seed, env, episode_count, progress_buf) read from the engine's counters
 * (cnt = lm_ptr(h, LM_PTR_CNT)), so no generator state is kept and no step repeats a draw.
 *   mean device [n_envs][12], log_std device [12], actions device [n_envs][12] (not clamped: lm_step clamps), logp device [n_envs] */
int lm_sample_actions(const float* mean, const float* log_std, const int64_t* cnt, int n_envs, uint32_t seed,
                      float* actions, float* logp, void* stream);

/* Plan a rollout of T steps on `env`.  All buffers are device memory owned by the caller and must stay valid:
 *   obs [T+1][N][lm_num_obs(env)]   obs[0] = the current (clipped) observations on entry; obs[t+1] = those returned by step t
 *                      (88-wide observations: MLP policy only)
 *   actions [T][N][12], logp [T][N], values [T+1][N] (value head output; values[T] bootstraps), rewards [T][N], dones int64 [T][N],
 *   extras [T][LM_NUM_EXTRAS] (may be NULL)
 *   policy_params / log_std: device blocks read at run time (update them in place between runs) */
int lm_rollout_create(lm_rollout** out, struct lm_engine* env, int policy, const float* policy_params, const float* log_std, int T,
                      uint32_t noise_seed, float* obs, float* actions, float* logp, float* values, float* rewards, int64_t* dones,
                      float* extras);
/* Enqueue the rollout on `stream`.  use_graph: LM_ROLLOUT_ENQUEUE enqueues the 2T+1 kernels, LM_ROLLOUT_GRAPH replays a hipGraph of them
 * captured on first use (one launch), LM_ROLLOUT_PERSISTENT runs the whole rollout inside ONE kernel (every block keeps its 16 envs for
 * the T steps, the observations go from the step to the next forward through LDS; un-randomised engines only, -1 otherwise).
 * A persistent block occupies a whole compute unit, so the mode pays up to 16 x (compute units) envs = 4096 on MI355X and serialises beyond;
 * LM_ROLLOUT_AUTO picks it within that size on engines that support it and the graph otherwise.
 * All modes write bit-identical buffers. */
#define LM_ROLLOUT_ENQUEUE 0
#define LM_ROLLOUT_GRAPH 1
#define LM_ROLLOUT_PERSISTENT 2
#define LM_ROLLOUT_AUTO 3
int lm_rollout_run(lm_rollout* r, int use_graph, void* stream);
int lm_rollout_destroy(lm_rollout* r);

#ifdef __cplusplus
}
#endif
#endif
