/*
 * lm_oracle.h -- CPU oracle for the loco-manipulation physics-step engine.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import, link or
 * execute this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * What it restates:
 *   - task layer (obs / reward / termination / reset bookkeeping): the reference's
 *     RobotLearning/omniisaacgymenvs/tasks/quadruped_pose_control_tasks/quadruped_pose_control.py
 *     :230-299 (reset_idx), :301-426 (get_observations), :428-560 (calculate_metrics),
 *     :562-633 (is_done); manipulation deltas quadruped_manipulate_plate.py:311-435,569-652;
 *     robot/base/robot.py:276-321 (state read-back), :444-461 (take_action);
 *     utils/math.py:33-193.  PINNED by golden vectors generated from the reference's
 *     own Python (tests/golden/task_*.npz, tools/gen_golden.py).
 *   - physics (reference row a7 = closed-source PhysX, absent from /root/reference): a float64 restatement of this repo's own physics
 *     specification written with a deliberately different algorithm from the HIP kernel (dense Jacobian
 *     projection + Cholesky + dense Delassus matrix instead of limb-aggregate articulated-body sweeps).  PhysX itself cannot be run or
 *     read; what pins the SPECIFICATION to it: the FK known answer, loop closure, mass, analytic cases and - the only
 *     simulator-derived numbers the reference holds - its 13 recorded PhysX joint trajectories, replayed open loop
 *     (tests/test_reference_npy_replay.py, DESIGN.md 2.1: per-kind entry rows into PhysX's success windows, shared window rows, the fall
 *     row, episode reward, negative controls).  PARITY UNPINNED and labelled so: the reading of the drive limit (tau_max), the effective
 *     friction coefficient (0.8 x nominal, fitted), row 0's deflection magnitude, the physics attributes of the domain
 *     randomisation.
 *
 * Build:  make -C oracle   (gcc -O2 -shared; -DLMO_FLOAT for the fp32 variant)
 */
#ifndef LM_ORACLE_H
#define LM_ORACLE_H
#include <stdint.h>

#ifdef LMO_FLOAT
typedef float real;
#else
typedef double real;
#endif

#define LMO_MAXB 24      /* bodies incl. root */
#define LMO_NTREE 20     /* tree DoF (12 driven + 8 passive) */
#define LMO_NQ 12        /* independent joint coordinates */

typedef struct {
  int32_t nb;                    /* number of bodies incl. root (21) */
  int32_t parent[LMO_MAXB];      /* parent body index, -1 for root */
  int32_t dof[LMO_MAXB];         /* tree DoF index moved by this body's joint, -1 root */
  double Rt[LMO_MAXB][9];        /* parent body frame -> joint frame at q=0 (row major) */
  double pt[LMO_MAXB][3];
  double axis[LMO_MAXB][3];      /* joint axis in joint frame */
  double mass[LMO_MAXB];
  double com[LMO_MAXB][3];       /* in body frame */
  double inertia[LMO_MAXB][9];   /* about COM, body axes */
  /* loop-closure embedding: passive tree dof p = sgn * g(q[a]-q[b]), g(D)=2 atan(sqrt2 tan(D/2)) */
  int32_t nclos;
  int32_t clos_p[8], clos_a[8], clos_b[8];
  double clos_s[8];
  int32_t tip_body[4];
  double tip_off[4][3];          /* fingertip_frame: the point the task layer reads back (robot.py:300-310) */
  int32_t knee_body[8];
  /* foot collider: the hemispherical end of the long distal link (Design/RobotURDF/mesh/collision/overconstrained/link3.obj,
     link2_right.obj: exact hemisphere, radius 5 mm, centre 5 mm behind the fingertip frame); radius = lmo_params.tip_radius */
  int32_t contact_body[4];
  double contact_off[4][3];      /* sphere centre in the contact body's frame */
} lmo_model;

#define LMO_DR_CHANNELS 9
enum { LMO_DR_OBS_RESET = 0, LMO_DR_OBS_INTERVAL, LMO_DR_ACT_RESET, LMO_DR_ACT_INTERVAL, LMO_DR_GRAVITY, LMO_DR_BASE_FORCE, LMO_DR_MAX_EFFORT,
       LMO_DR_MAX_VELOCITY, LMO_DR_JOINT_DAMPING };
typedef struct {
  int32_t enabled;
  int32_t operation;       /* 0 additive, 1 scaling, 2 direct */
  int32_t distribution;    /* 0 gaussian, 1 uniform, 2 loguniform */
  int32_t interval;        /* frequency_interval of an on_interval entry; 0 = on_reset entry */
  double p0[3], p1[3];     /* distribution_parameters: mean/std or low/high (per component for vector attributes) */
} lmo_dr_channel;

typedef struct {
  /* physics */
  double dt;                 /* 0.0083 */
  int32_t substeps;          /* controlFrequencyInv = 4 */
  int32_t pgs_iters;         /* contact sweeps per solve */
  double gravity;            /* 9.81 (acts along -z world) */
  double kd;                 /* velocity-drive damping 100 */
  double tau_max;            /* 1.5 */
  double act_scale;          /* velocity limit used for action scaling, 3.0 */
  double mu;                 /* tip friction coefficient */
  double tip_radius;
  double baumgarte;          /* penetration push-out fraction per step */
  double max_depen_vel;      /* m/s cap on push-out */
  double max_joint_vel;      /* rad/s clamp on the driven joints' speed (USD maxJointVelocity 450 deg/s) */
  int32_t mode;              /* 0 = free base on ground (loco), 1 = fixed inverted base + plate (mani) */
  int32_t pyramid;           /* 0 (shipped): friction cone; 1: the axis-aligned friction pyramid of rounds 1-2 (evidence tables only; the kernel has no such switch) */
  double fixed_base_pos[3];
  double fixed_base_quat[4];
  double plate_mass;
  double plate_com[3];
  double plate_inertia[3];   /* diagonal about COM, plate axes */
  double plate_half[3];      /* collider half extents */
  double plate_center[3];    /* collider centre in plate frame */
  /* reset (reset_idx) */
  double init_q[12];
  double init_base_pos[3];
  double init_base_quat[4];
  double init_plate_pos[3];
  double init_plate_quat[4];
  double default_tip[12];    /* default_base_tip_positions */
  double goal_lo[3], goal_hi[3];   /* roll/pitch/yaw ranges */
  /* observation scales */
  double s_pos, s_lin, s_ang, s_q, s_qd;
  /* reward */
  double quat_scale, rot_eps, trans_scale, acc_scale, rate_scale, bonus, limit_pen, fall_pen;
  double succ_thresh;
  int32_t max_consec;        /* 15 */
  int32_t max_episode;       /* 300 */
  double d23_pen[2], d23_rst[2];   /* |dof3-dof2| penalty / reset windows */
  double d1_pen[4][2], d1_rst[4][2]; /* per-limb dof1 windows */
  double h_base, h_corner, h_knee;
  double corner[4][3];       /* frame corner points in the robot base frame */
  double ctrl_dt;            /* dt*substeps, for joint acceleration */
  /* ---- custom-controller task family (quadruped_pose_control_custom_controller.py:24-52,255-307,530-545):
   * actions integrate swing/extension position targets, the actuator is  tau = clamp(kp (q* - q) - kd qd, +-tau_max)
   * re-evaluated every sub-step (kd above is its damping gain), plus viscous joint damping */
  int32_t variant;           /* 0 = velocity-drive tasks, 1 = custom-controller tasks, 2 = position-control tasks (same PD actuator and
                                swing/extension actions; 64-wide observation whose last 24 entries are the scaled joint position targets; base reward) */
  int32_t num_obs;           /* 64 or 88 */
  double pd_kp;              /* 4.5 */
  double joint_damping;      /* 0.008 (0 for variant 0) */
  double act_scale_se;       /* 0.1 */
  double se_lo[12], se_hi[12], init_se[12];
  double torque_div;         /* control_decimal: the logged torque is sum over sub-steps / control_decimal (:307) */
  double power_scale, target_err_scale, rot_dec_scale, rot_dec_thresh;
  int32_t cc_update_last_tgt;  /* 1: last_joint_position_targets follows the targets (loco, :723-725); 0: stays at its reset value
                                  (quadruped_manipulate_plate_custom_controller.py never updates it after :378) */
  int32_t acc_substeps;      /* variants 1/2: trailing sub-steps spanned by the joint acceleration (controlFrequencyInv; robot.py:289-291) */
  /* ---- domain randomisation: utils/domain_randomization/randomize.py:212-306 (observation / action noise, pinned by
   * tests/golden/dr_noise.npz) and the physics attributes of cfg/task/QuadrupedPoseControl.yaml:136-173 (gravity, base-link force,
   * max efforts, max joint velocities; their sampling lives in omni.replicator.isaac, absent: this repo's specification) */
  int32_t dr_enabled, dr_min_frequency;
  lmo_dr_channel dr[LMO_DR_CHANNELS];
  int32_t drive_mode;        /* variant 0: RobotOmni.take_action's control mode (robot/base/robot.py:444-461): 0 velocity target a*act_scale,
                                1 position target a*act_scale with tau = pd_kp (q* - q) - kd qd per sub-step, 2 effort tau = a*act_scale */
  int32_t pd_second_pass;    /* variants 1 / 2: 1 = unsaturated joints whose implicit torque left the limit are put on it and the sub-step is solved again (engine_config.py) */
  /* ---- round 4: the drive as rows of the SAME velocity-level iteration as the contacts (variant 0, velocity drive).
   * solver 0 = the drive solved exactly (diagonal augmentation of M) and Gauss-Seidel on the contact rows only;
   * solver 1 = `pgs_iters` position iterations (the reference's solver_position_iteration_count 16, cfg/task/QuadrupedPoseControl.yaml:41) over
   * [4 contacts x (normal, friction pair), 12 drive rows] on ONE linearisation per sub-step, every drive row's impulse bounded PER ITERATION by
   * drive_iter_impulse (max effort 1.5 N m x dt: robot/base/robot.py:347-355), joint speeds bounded by max_joint_vel in every iteration
   * (Design/Scripts/config_module_joints.py:11), gaps and joint angles advanced by dt / pgs_iters after every iteration (TGS), then `vel_iters`
   * iterations without the penetration bias (solver_velocity_iteration_count 2, YAML :42). */
  int32_t solver;
  int32_t vel_iters;
  double drive_iter_impulse;   /* N m s per iteration and drive row; <= 0: unbounded */
  int32_t tgs_flags;           /* experiment switches of solver 1 (tests/npy_replay_evidence.py): 1 = drive rows before the contact rows; 2 = positions advance
                                  with the END velocity over the whole dt instead of per iteration; 4 = the speed bound as an impulse through M^-1 instead
                                  of a direct clamp; 8 = a gap's speculative bias over the whole dt instead of the iteration's share;
                                  16 = the drive's impulse bound on the accumulated impulse (x pgs_iters) instead of per iteration;
                                  64 = a gap's speculative bias is (current gap) / dt;
                                  128 = drive rows limb by limb (hip, dof2, dof3) instead of the DoF order; 256 = contacts always in limb order;
                                  512 = the 12 drive rows solved as one block per iteration (exact 12 x 12 solve, then the per-row bound);
                                  32 = contact impulses accumulate (and are clamped / released) within one iteration only */
} lmo_params;

/* per-env physical state, env-major */
#define LMO_PHYS 50
/*  0:3 base_pos  3:7 base_quat(wxyz)  7:10 base_linvel(world) 10:13 base_angvel(world)
 * 13:25 q  25:37 qd  37:40 plate_pos 40:44 plate_quat 44:47 plate_linvel 47:50 plate_angvel */
#define LMO_TASK 65
/*  0:12 last_actions 12:24 last_qd 24:36 last_base_tip 36:40 goal_quat
 *  40:52 swing/extension targets 52:64 last joint position targets 64 last_rot_dist   (variant 1) */
#define LMO_CNT 6
/*  successes, consecutive_successes, goal_reset_buf, reset_buf, progress_buf, episode_count */
#define LMO_READBACK 99
/*  0:12 q 12:24 qd 24:36 acc 36:39 base_pos 39:43 base_quat 43:46 linvel 46:49 angvel
 *  49:61 tips(4x3) 61:85 knees(8x3) 85:87 unused 87:99 logged joint torque (variant 1)
 *  (mani: base_* slots hold the plate pose/velocity; robot pose is params.fixed_base_*) */
#define LMO_TERMS 11
/*  rot_rew, translation_penalty, joint_acc_penalty, action_rate_penalty,
 *  consecutive_successes_rew, joint_limit_penalty, fall_penalty, goal_reset (as real),
 *  mechanical_power_penalty, position_target_error_penalty, rot_dist_decreasing_reward */

#define LMO_DR_CNT 5
/*  per-env randomisation counters (int64): observation noise counter, action noise counter (randomize.py:214-216,238-240),
 *  dr_step (control steps since creation), randomization_buf (steps since the last gated on_reset randomisation,
 *  quadruped_pose_control.py:224-228), dr_reset_key (episode index at that randomisation) */

#ifdef __cplusplus
extern "C" {
#endif

/* kinematics of one env: world tips (4x3) and knees (8x3) */
void lmo_fk(const lmo_model* m, const lmo_params* p, const real* phys, real* tips, real* knees);

/* one physics sub-step for N envs; targets = joint velocity targets (N x 12) */
void lmo_substep(const lmo_model* m, const lmo_params* p, int N, real* phys, const real* targets);
/* same, also returning the drive torque applied in that sub-step (N x 12) */
void lmo_substep_tau(const lmo_model* m, const lmo_params* p, int N, real* phys, const real* targets, real* tau);

/* contact problem of one env's next sub-step: Delassus W (12x12), free contact velocities vf (12), normal biases bn (4)
 * and the PGS impulses lam (12) -- for solver-convergence studies and tests */
void lmo_contact_problem(const lmo_model* m, const lmo_params* p, const real* phys, const real* target, real* W, real* vf, real* bn, real* lam);

/* dense dynamics terms of one env (debug/tests): M (18x18 row major), h (18) */
void lmo_dyn_terms(const lmo_model* m, const lmo_params* p, const real* phys, real* M, real* h);

/* task layer on explicit read-back inputs (golden-vector entry point) */
void lmo_task_eval(const lmo_params* p, int N, const real* readback, const real* actions,
                   real* task, int64_t* cnt, real* obs, real* states, real* rew, real* terms);

/* reset of flagged envs (cnt reset_buf != 0); goal_rand may be NULL -> hash RNG(seed) */
void lmo_reset(const lmo_params* p, int N, real* phys, real* task, int64_t* cnt,
               const real* goal_rand, uint32_t seed);

/* full env step: reset flagged -> targets -> substeps -> readback -> task_eval */
void lmo_step(const lmo_model* m, const lmo_params* p, int N, real* phys, real* task, int64_t* cnt,
              const real* actions, const real* goal_rand, uint32_t seed,
              real* obs, real* states, real* rew, real* terms);

/* goal sampling helpers exposed for tests */
/* counter-based sample of one randomised scalar: stream = quantity, key = episode / step / interval index, idx = component */
real lmo_dr_sample(uint32_t seed, uint32_t stream, uint32_t env, uint32_t key, uint32_t idx, int distribution, real p0, real p1);
/* randomize.py:212-306 restated on an (N x D) buffer: correlated noise drawn when an env's reset flag is set (keyed by corr_key) and
 * applied every call, plus uncorrelated noise (keyed by step_key) every `interval` calls; counter is the reference's per-env counter */
void lmo_dr_noise(const lmo_dr_channel* on_reset, const lmo_dr_channel* on_interval, uint32_t seed, uint32_t stream, int N, int D,
                  real* buf, const int64_t* reset_flags, int64_t* counter, const int64_t* corr_key, const int64_t* step_key);
/* lmo_step with domain randomisation: raw (unclamped) actions in; obs_noisy = what VecEnvRLGames.step hands to the policy before
 * clamping (vec_env_rlgames.py:56-72).  drc = (N x LMO_DR_CNT) counters.  physdr (N x 42, may be NULL) receives the sampled
 * max efforts (12), max velocities (12), gravity (3), base force (3), joint damping (12) */
/* env ids feed the random stream: a block that is not at the start of the env range (the manipulation half of a co-training engine)
 * sets the global id of its first env before calling lmo_dr_noise / lmo_step_dr (not thread safe: test infrastructure) */
void lmo_set_env_offset(uint32_t env0);
void lmo_step_dr(const lmo_model* m, const lmo_params* p, int N, real* phys, real* task, int64_t* cnt, int64_t* drc,
                 const real* actions_raw, real clip_actions, const real* goal_rand, uint32_t seed,
                 real* obs, real* states, real* rew, real* terms, real* actions_used, real* physdr);
void lmo_hash_uniform3(uint32_t seed, uint32_t env, uint32_t episode, real* u3);
void lmo_quat_from_euler(real roll, real pitch, real yaw, real* q);
int lmo_sizeof_real(void);

#ifdef __cplusplus
}
#endif
#endif
