/*
 * lm_engine.h -- C ABI of the MI355X loco-manipulation physics-step engine (liblm_engine.so).
 *
 * The reference has no FFI on this path: the boundary is the Python object protocol
 *   VecEnvRLGames.step / reset / set_task      RobotLearning/omniisaacgymenvs/envs/vec_env_rlgames.py:48-90
 *   RLTask buffers + post_physics_step         RobotLearning/omniisaacgymenvs/tasks/base/rl_task.py:104-113,240-260
 * below which it calls closed-source PhysX (World.step, ArticulationView / RigidPrimView tensor API,
 * vec_env_rlgames.py:65; robot/base/robot.py:276-321,357-461; objects/base/rigid_object.py:30-58).
 * These entry points are what a binding for that path binds instead (INTEGRATION.md shows the ctypes
 * stub).  Plain pointers and sizes only; no torch types; no exceptions cross the ABI; every call
 * returns 0 on success or a negative LM_E* code (lm_last_error() gives the text).
 *
 * Memory: all device buffers are owned by the handle (hipMalloc) and exposed through lm_ptr();
 * output buffers for step() are caller-provided device pointers (so a host binding can hand out
 * fresh tensors per step, matching the "returned tensors are clones" contract of
 * vec_env_rlgames.py:41-46 without extra copy kernels).  The stream is the caller's.
 */
#ifndef LM_ENGINE_H
#define LM_ENGINE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LM_OK 0
#define LM_EINVAL (-1)
#define LM_EHIP (-2)
#define LM_ENOMEM (-3)

#define LM_MODE_LOCO 0   /* free base on a ground plane   (QuadrupedPoseControl)      */
#define LM_MODE_MANI 1   /* fixed inverted base + plate    (QuadrupedManipulatePlate)  */

#define LM_STATE_ROWS 115  /* float state, SoA [row][N]: see DESIGN.md 4.1 */
#define LM_CNT_ROWS 6      /* int64 counters, SoA [row][N] */
#define LM_NUM_OBS 64        /* velocity-drive tasks; the custom-controller tasks have 88 (lm_num_obs) */
#define LM_MAX_OBS 88
#define LM_TERM_ROWS 11
#define LM_NUM_STATES 93
#define LM_NUM_ACTIONS 12
#define LM_NUM_EXTRAS 13  /* 7 reward-term means, success_rate, success_rate of task 0 / task 1 (co-train),
                             custom-controller means: mechanical_power, position_target_error, rot_dist_decreasing */
#define LM_TABLE_FLOATS 502  /* 10 hub + 4 x 123 limb (RobotModel.packed_table) */
/* Bumped whenever lm_params, the table layout or the meaning of an entry point changes.  Every lm_params block carries it together with the
 * caller's sizeof(lm_params) and LM_TABLE_FLOATS (first three fields); lm_create returns LM_EINVAL when any of them differs from what the
 * library was built with, instead of reading a shifted struct or past the end of a shorter table.
 *   1  round 1     2  round 2 (drive_mode, 502-float table; not stamped)     3  round 3 (the stamp itself; pgs_iters per contact surface)
 *   4  round 3 (pd_second_pass replaces sat_probe: the PD-actuator families decide their clamp on the pre-step state, one pass) */
#define LM_ABI_VERSION 4

/* Task / simulation constants for one task family.  Mirrors EngineParams (engine_config.py);
 * sources in the reference are cited there. */
/* Domain randomisation.  One channel = one entry of the YAML block cfg/task/QuadrupedPoseControl.yaml:116-173
 * (operation / distribution / distribution_parameters [/ frequency_interval]).  Observation and action noise follow
 * utils/domain_randomization/randomize.py:212-306 (wrapper call sites vec_env_rlgames.py:56-58,70-72); the physics attributes are
 * sampled per env inside lm_step (the reference delegates them to omni.replicator.isaac; semantics in DESIGN.md 3.6). */
#define LM_DR_CHANNELS 9
enum { LM_DR_OBS_RESET = 0, LM_DR_OBS_INTERVAL = 1, LM_DR_ACT_RESET = 2, LM_DR_ACT_INTERVAL = 3,
       LM_DR_GRAVITY = 4, LM_DR_BASE_FORCE = 5, LM_DR_MAX_EFFORT = 6, LM_DR_MAX_VELOCITY = 7,
       LM_DR_JOINT_DAMPING = 8 /* articulation `damping` of the PD-actuator tasks (variants 1 / 2): scales joint_damping */ };
enum { LM_DR_ADDITIVE = 0, LM_DR_SCALING = 1, LM_DR_DIRECT = 2 };
enum { LM_DRIVE_VELOCITY = 0, LM_DRIVE_POSITION = 1, LM_DRIVE_EFFORT = 2 };
enum { LM_DR_GAUSSIAN = 0, LM_DR_UNIFORM = 1, LM_DR_LOGUNIFORM = 2 };
typedef struct lm_dr_channel {
  int32_t enabled;
  int32_t operation;       /* LM_DR_ADDITIVE / SCALING / DIRECT */
  int32_t distribution;    /* LM_DR_GAUSSIAN / UNIFORM / LOGUNIFORM */
  int32_t interval;        /* frequency_interval of an on_interval entry (>= 1); 0 = on_reset entry */
  float p0[3], p1[3];      /* distribution_parameters: mean / std or low / high (one pair per component for gravity and force) */
} lm_dr_channel;
#define LM_DR_PHYS_ROWS 42
#define LM_DR_CNT_ROWS 5   /* int64 [row][N]: observation noise counter, action noise counter, dr_step, randomization_buf, dr_reset_key */

typedef struct lm_params {
  int32_t abi_version;     /* LM_ABI_VERSION of the header the caller was compiled against */
  int32_t params_size;     /* the caller's sizeof(lm_params) */
  int32_t table_floats;    /* the caller's LM_TABLE_FLOATS = the length of the table it passes to lm_create */
  int32_t reserved0;       /* 0 */
  float dt, kd, tau_max, act_scale, mu, tip_radius, baumgarte, max_depen_vel, max_joint_vel, gravity;
  int32_t substeps, pgs_iters, mode;
  float fixed_base_pos[3], fixed_base_quat[4];
  float plate_mass, plate_com[3], plate_inertia[3], plate_half[3], plate_center[3];
  float init_q[12], init_base_pos[3], init_base_quat[4], init_plate_pos[3], init_plate_quat[4];
  float default_tip[12], goal_lo[3], goal_hi[3];
  float s_pos, s_lin, s_ang, s_q, s_qd;
  float quat_scale, rot_eps, trans_scale, acc_scale, rate_scale, bonus, limit_pen, fall_pen, succ_thresh;
  int32_t max_consec, max_episode;
  float d23_pen[2], d23_rst[2], d1_pen[4][2], d1_rst[4][2];
  float h_base, h_corner, h_knee, corner[4][3];
  float clip_obs, clip_actions;
  int32_t max_reset_counts;
  /* custom-controller task family (quadruped_pose_control_custom_controller.py:24-52,88-97,255-307) */
  int32_t variant;           /* 0 velocity-drive tasks, 1 custom-controller tasks, 2 position-control tasks (same PD actuator and swing/extension
                                actions; 64-wide observation with the scaled joint targets in place of the actions; reward of variant 0;
                                quadruped_pose_control_position_control.py:24-118,438-455, joint_locomanipulation_position_control.py:37-44,365-408) */
  int32_t num_obs;           /* 64 / 88 */
  float pd_kp, joint_damping, act_scale_se;
  float se_lo[12], se_hi[12], init_se[12];
  float torque_div, power_scale, target_err_scale, rot_dec_scale, rot_dec_thresh;
  int32_t cc_update_last_tgt;
  int32_t acc_substeps;      /* variants 1/2: trailing sub-steps the joint acceleration spans (controlFrequencyInv; robot.py:289-291) */
  int32_t dr_enabled;        /* domain_randomization.randomize */
  int32_t dr_min_frequency;  /* domain_randomization.min_frequency: gate of the on_reset physics attributes (quadruped_pose_control.py:224-228) */
  lm_dr_channel dr[LM_DR_CHANNELS];
  int32_t drive_mode;        /* variant 0 only - RobotOmni.take_action's control modes (robot/base/robot.py:444-461):
                                LM_DRIVE_VELOCITY  target = a * act_scale [rad/s]           implicit damper kd (the mode every task of the path uses)
                                LM_DRIVE_POSITION  target = a * act_scale [rad] (act_scale = pi); tau = pd_kp (q* - q) - kd qd, re-evaluated per sub-step
                                LM_DRIVE_EFFORT    tau = a * act_scale [N m] (act_scale = torque limit), gains off */
  int32_t pd_second_pass;    /* variants 1 / 2.  0 (default): which joints sit on the +-tau_max limit is decided from the PD torque on the state BEFORE
                                the sub-step, as the reference's explicit clamp decides it (quadruped_pose_control_custom_controller.py:289-293); the others
                                get the implicit form of the law in ONE pass, and the 0.02 % of joint-sub-steps whose implicit torque then leaves the
                                limit keep it.  1: those joints are put on the limit too and the sub-step is solved a second time (the applied
                                torque never exceeds tau_max; a step then takes as long as its slowest wavefront: +6 us at 4096 envs).  DESIGN.md 3.3 */
  /* derived by lm_create (callers leave zero) */
  float plate_si[10];      /* plate spatial inertia about its origin */
  float plate_phi[36];     /* its inverse */
  float ctrl_dt_inv, acc_dt_inv;
} lm_params;

typedef struct lm_engine lm_engine;   /* opaque */

/* Pointers the host side may wrap zero-copy.  LM_PTR_OBS_BUF, LM_PTR_STATES_BUF and LM_PTR_TERMS are the engine's own unclipped copies of
 * what lm_step's out_obs / out_states deliver clipped: lm_step keeps each of them current from the first lm_ptr() call for it on (ask before
 * the step whose values you want), and in any case when the matching out_* argument is NULL; a caller that only consumes the out_* buffers
 * does not pay for the second copy.  The staged entry points (lm_post_physics ...) always write them.
 * CAVEATS of the on-demand copies: (1) the first lm_ptr() call for one of the three changes what LATER lm_step launches write - the buffer it
 * returns is zeros / an older step's values until the next step (the Python mirror warns when a view is first requested after stepping);
 * (2) a hipGraph the CALLER captured around lm_step before that first lm_ptr() call keeps the kernel arguments of capture time (NULL view
 * pointers) for ever: request the views you need before capturing.  lm_rollout's own graph is re-captured when the set of views changed. */
typedef enum {
  LM_PTR_STATE = 0,     /* float [LM_STATE_ROWS][N]                                  */
  LM_PTR_CNT = 1,       /* int64 [LM_CNT_ROWS][N]: successes, consecutive_successes,
                           goal_reset_buf, reset_buf, progress_buf, episode_count    */
  LM_PTR_OBS_BUF = 2,   /* float [N][lm_num_obs()]  task.obs_buf (unclipped)   rl_task.py:107  */
  LM_PTR_STATES_BUF = 3,/* float [N][93]  task.states_buf                            */
  LM_PTR_REW_BUF = 4,   /* float [N]      task.rew_buf                               */
  LM_PTR_EXTRAS = 5,    /* float [LM_NUM_EXTRAS]  reward-term means + success rates     */
  LM_PTR_STATS = 6,     /* int64 [6] {num_successes, num_resets} x {all, task 0, task 1}; float [3] rates at byte 48;
                           uint32 at byte 60: envs whose state became non-finite / exploded and was replaced by the reset pose */
  LM_PTR_TERMS = 7,     /* float [LM_TERM_ROWS][N]  per-env reward terms of the last step */
  LM_PTR_DR_CNT = 8,    /* int64 [LM_DR_CNT_ROWS][N]  domain-randomisation counters */
  LM_PTR_DR_PHYS = 9    /* float [LM_DR_PHYS_ROWS][N]  attributes sampled for the last step: max efforts 12, max joint velocities 12,
                           gravity 3, base force 3, joint damping 12 */
} lm_ptr_kind;

/* Create an engine for n_envs environments on the current HIP device.
 *   table      : LM_TABLE_FLOATS packed robot model (host pointer)
 *   params     : n_tasks (1 or 2) parameter blocks (host pointer); with 2 tasks, envs [0, split_env)
 *                use params[0] and [split_env, n_envs) use params[1] (co-train layout,
 *                joint_locomanipulation.py:25-34); split_env must be a multiple of 16.
 *   seed       : stream seed of the in-kernel goal sampler (replaces torch.rand in utils/math.py:184)
 * n_envs must be positive and below 2^25 (33 554 432) per engine: the width of the reset counts in the fused extras reduction; LM_EINVAL otherwise.
 * All envs start with reset_buf = 1 (rl_task.py:111).
 * The engine belongs to the device that is current in the calling thread here: every later call on the handle must be made with the
 * same device current (one process per GPU) and returns LM_EINVAL otherwise instead of launching on another GPU. */
int lm_create(lm_engine** out, int n_envs, const float* table, const lm_params* params, int n_tasks,
              int split_env, uint32_t seed);
int lm_destroy(lm_engine* h);

/* One VecEnvRLGames.step(): reset flagged envs, clamp + apply actions, controlFrequencyInv physics
 * sub-steps, observations / reward / termination.  (vec_env_rlgames.py:56-79)
 *   actions     device float [N][12]
 *   goal_rand   device float [N][3] uniforms for goal sampling, or NULL for the in-kernel hash RNG
 *   out_*       device buffers receiving clipped copies for the caller (any may be NULL; see lm_ptr_kind for the unclipped copies):
 *               obs [N][lm_num_obs], states [N][93], rew [N], resets int64 [N], extras float [LM_NUM_EXTRAS]
 *   stream      hipStream_t (void* here so the header needs no HIP include)
 * With params.dr_enabled the same launch also applies the action noise (before the clamp), samples this step's physics attributes
 * and adds the observation noise (obs_buf and out_obs); the staged entry points below refuse a randomised engine.
 * One kernel launch.  Un-randomised velocity-drive locomotion engines of more than 32 768 envs get the build of the same kernel for two
 * wavefronts per SIMD (k_step_w2: same bits, faster beyond two generations of workgroups; the environment variable LM_W2_MIN_ENVS, read
 * by lm_create, moves that threshold). */
int lm_step(lm_engine* h, const float* actions, const float* goal_rand, float* out_obs, float* out_states,
            float* out_rew, int64_t* out_resets, float* out_extras, void* stream);

/* Staged form of lm_step for callers that drive the reference's three phases themselves
 * (scripts/random_policy.py:57-61): lm_apply_resets = the reset part of pre_physics_step,
 * lm_substeps(...,1) = one world.step, lm_post_physics = post_physics_step (state read-back,
 * observations, reward, termination; rl_task.py:240-260).  Running apply_resets, controlFrequencyInv
 * sub-steps and post_physics gives the same state as one lm_step. */
int lm_post_physics(lm_engine* h, const float* actions, float* out_obs, float* out_states, float* out_rew,
                    int64_t* out_resets, float* out_extras, void* stream);

/* RLTask.reset(): flag every env for reset (rl_task.py:227-230). */
int lm_reset_all(lm_engine* h, void* stream);

/* Task layer alone on explicit read-back inputs (device float [N][99], layout of oracle LMO_READBACK);
 * uses and updates the handle's task state / counters exactly like the tail of lm_step.
 * Test entry point for the golden vectors captured from the reference's Python. */
int lm_task_eval(lm_engine* h, const float* readback, const float* actions, float* out_obs, float* out_states,
                 float* out_rew, int64_t* out_resets, float* out_extras, void* stream);

/* Only the reset scatter of lm_step (reset_idx, quadruped_pose_control.py:230-299). */
int lm_apply_resets(lm_engine* h, const float* goal_rand, void* stream);

/* n physics sub-steps with given joint-velocity targets (device float [N][12]); no task layer. */
int lm_substeps(lm_engine* h, const float* targets, int n, void* stream);

/* World tip (device float [N][4][3]) and knee ([N][8][3]) positions of the current state. */
int lm_forward_kinematics(lm_engine* h, float* tips, float* knees, void* stream);

/* Debug: dense mass matrix [N][18][18] and bias [N][18] assembled from the limb-aggregate terms. */
int lm_debug_dynamics(lm_engine* h, float* M, float* hvec, void* stream);

void* lm_ptr(lm_engine* h, int kind);
int lm_num_envs(const lm_engine* h);
int lm_num_obs(const lm_engine* h);      /* observation width of this engine (64 or 88) */
int lm_set_seed(lm_engine* h, uint32_t seed);
const char* lm_last_error(void);
const char* lm_version(void);
int lm_abi_version(void);                /* LM_ABI_VERSION the library was built with */

#ifdef __cplusplus
}
#endif
#endif
