/*
 * lm_policy.h -- C ABI of the GNN policy forward pass (part of liblm_engine.so).
 *
 * Replaces, for inference, the torch modules of RobotLearning/omniisaacgymenvs/scripts/graph_model_orebot_ov.py
 * (GraphNet :82-110, GraphLayer :11-80, Action_Layer :215-226, Value_Layer :228-241) as instantiated by
 * scripts/skrl_ppo_locomanipulation_vertical.py:44-49 (hidden_features = out_features = 32).
 */
#ifndef LM_POLICY_H
#define LM_POLICY_H
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
 * locomanipulationrl_amd.policies.mlp_model.pack_mlp_params (weights pre-permuted into MFMA operand order);
 * lm_mlp_param_count() is its length.  Outputs as lm_gnn_forward (mean in the env's action order). */
int lm_mlp_param_count(void);
int lm_mlp_forward(const float* obs, int batch, const float* params, float* mean, float* value, void* stream);

#ifdef __cplusplus
}
#endif
#endif
