/*
 * legged_hip.h -- C-ABI of the MI355X-native legged-robot hot path.
 *
 * This is the *inner* drop-in boundary of SURVEY.md section 8(b): it replaces
 * the Isaac Gym tensor API calls the reference environment makes from
 *   legged_gym/envs/base/legged_robot.py:80-137   (step / post_physics_step)
 * plus the torch-side arithmetic between those calls, with one fused device
 * launch per policy step.  Each entry point cites the reference call sites it
 * stands in for.  No torch types appear here: all arrays are raw device
 * pointers to buffers the caller (PyTorch-ROCm) allocated.
 *
 * The CPU oracle under oracle/ exports the same functions with the prefix
 * `lgo_` over host pointers; it is test infrastructure only.
 *
 * Conventions: extern "C", opaque handle, fp32 / int32 / uint8 arrays,
 * 0 = success, negative = error (text via lg_last_error()), no allocation
 * inside lg_step, one handle per GPU, a handle is not thread-safe (the
 * reference is single-threaded), the HIP stream is passed explicitly.
 */
#ifndef LEGGED_HIP_H
#define LEGGED_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LG_ABI_VERSION        21

#define LG_MAX_LIMBS          4
#define LG_MAX_CHAIN          6
#define LG_MAX_DOF            12
#define LG_MAX_LIMB_POINTS    8
#define LG_MAX_BASE_POINTS    4
#define LG_MAX_BODIES         20
#define LG_MAX_HEIGHT_POINTS  192
#define LG_ACTUATOR_FLOATS    972

/* Reward-term ids, ALPHABETICAL by name: the reference sums terms in dir()
 * order (helpers.py:45, legged_robot.py:199-203).  `termination` is added
 * after the only_positive clip (legged_robot.py:206-210). */
enum lg_reward_term {
    LG_REW_ACTION_RATE = 0, LG_REW_ANG_VEL_XY, LG_REW_BASE_HEIGHT, LG_REW_COLLISION,
    LG_REW_DOF_ACC, LG_REW_DOF_POS_LIMITS, LG_REW_DOF_VEL, LG_REW_DOF_VEL_LIMITS,
    LG_REW_FEET_AIR_TIME, LG_REW_FEET_CONTACT_FORCES, LG_REW_LIN_VEL_Z, LG_REW_NO_FLY,
    LG_REW_ORIENTATION, LG_REW_STAND_STILL, LG_REW_STUMBLE, LG_REW_TERMINATION,
    LG_REW_TORQUE_LIMITS, LG_REW_TORQUES, LG_REW_TRACKING_ANG_VEL, LG_REW_TRACKING_LIN_VEL,
    LG_NUM_REWARD_TERMS
};

enum lg_control_type { LG_CTRL_P = 0, LG_CTRL_V = 1, LG_CTRL_T = 2, LG_CTRL_ACTUATOR_NET = 3 };
enum lg_terrain_type { LG_TERRAIN_PLANE = 0, LG_TERRAIN_HEIGHTFIELD = 1 };

/* A collision sphere rigidly attached to a dynamic body (a capsule = 2). */
typedef struct lg_point {
    float   pos[3];        /* in the carrying dynamic body's frame */
    float   radius;
    int32_t report_body;   /* row of contact_forces this point reports to */
    int32_t joint;         /* index in the chain (0..L-1) of the carrying body; -1 = base */
} lg_point;

/* Robot = floating base + K serial limbs of L revolute joints (K*L == 12).
 * Output of the model compiler (legged_games_gym_amd/utils/model_compiler.py),
 * which stands in for gym.load_asset + get_asset_* (legged_robot.py:685-702). */
typedef struct lg_robot_model {
    int32_t num_limbs, chain_len, num_bodies, _pad0;
    float   base_mass, base_com[3], base_inertia[6];      /* xx,xy,xz,yy,yz,zz at COM */
    float   joint_pos[LG_MAX_DOF][3];                     /* joint origin in parent body frame */
    float   joint_rot[LG_MAX_DOF][9];                     /* joint frame -> parent frame (q=0), row-major */
    float   joint_axis[LG_MAX_DOF][3];                    /* unit axis in the joint frame */
    float   body_mass[LG_MAX_DOF], body_com[LG_MAX_DOF][3], body_inertia[LG_MAX_DOF][6];
    float   dof_lower[LG_MAX_DOF], dof_upper[LG_MAX_DOF]; /* hard limits (URDF); lower>upper = none */
    float   dof_vel_limit[LG_MAX_DOF];
    float   dof_armature[LG_MAX_DOF], dof_damping[LG_MAX_DOF];
    int32_t num_base_points, num_limb_points[LG_MAX_LIMBS], _pad1[3];
    lg_point base_points[LG_MAX_BASE_POINTS];
    lg_point limb_points[LG_MAX_LIMBS][LG_MAX_LIMB_POINTS];
    int32_t foot_body[LG_MAX_LIMBS];                      /* report body of each limb's foot (feet_indices) */
    uint32_t penalised_mask, termination_mask;            /* bit b = report body b */
} lg_robot_model;

/* Everything LeggedRobotCfg / sim_params contribute to the hot path. */
typedef struct lg_params {
    int32_t abi_version, num_envs, decimation, control_type;
    float   sim_dt, gravity[3];
    /* contact + joint-limit model of the built-in rigid-body engine */
    float   contact_stiffness, contact_damping, friction_damping, contact_margin;
    float   ground_friction, limit_stiffness, limit_damping, stick_velocity; /* Coulomb regularisation speed [m/s] */
    /* control: legged_robot.py:371-395, anymal.py:71-81 */
    float   action_scale, clip_actions, clip_observations, _padf1;
    float   p_gains[LG_MAX_DOF], d_gains[LG_MAX_DOF], default_dof_pos[LG_MAX_DOF], torque_limits[LG_MAX_DOF];
    /* soft limits for the reward terms: legged_robot.py:310-313, 914-930 */
    float   soft_pos_lower[LG_MAX_DOF], soft_pos_upper[LG_MAX_DOF], dof_vel_limits[LG_MAX_DOF];
    float   soft_dof_vel_limit, soft_torque_limit, tracking_sigma, base_height_target;
    float   max_contact_force, dt_policy, max_push_vel, _padf2;
    /* episode / commands: legged_robot.py:329-369, 781-791 */
    int32_t max_episode_length, push_interval, resample_interval, heading_command;
    float   cmd_lin_vel_x[2], cmd_lin_vel_y[2], cmd_ang_vel_yaw[2], cmd_heading[2];
    /* observations: legged_robot.py:212-230, 485-508 */
    float   obs_scale_lin_vel, obs_scale_ang_vel, obs_scale_dof_pos, obs_scale_dof_vel, obs_scale_height;
    float   noise_lin_vel, noise_ang_vel, noise_gravity, noise_dof_pos, noise_dof_vel, noise_height; /* already x level x obs scale */
    int32_t add_noise, measure_heights, num_height_points, num_obs;
    float   height_points[LG_MAX_HEIGHT_POINTS][2];
    /* rewards: legged_robot.py:193-210, 583-607; scale already multiplied by dt, 0 = term absent */
    float   reward_scale[LG_NUM_REWARD_TERMS];
    int32_t only_positive_rewards, reward_slot[LG_NUM_REWARD_TERMS]; /* row of episode_sums, -1 = absent */
    int32_t num_reward_slots;
    /* terrain: legged_robot.py:609-637, 831-869; terrain.py */
    int32_t terrain_type, hf_rows, hf_cols, custom_origins;
    float   hf_horizontal_scale, hf_vertical_scale, hf_border;
    float   hf_step_threshold;         /* > 0 (mesh_type 'trimesh': slope_treshold x horizontal_scale, legged_robot_config.py:66, terrain.py:69-73): a height
                                          difference across a cell beyond this [m] is a vertical face at the high side, not a ramp; 0: bilinear patches */
    int32_t terrain_curriculum, terrain_num_rows, terrain_num_cols;
    int32_t self_collision;            /* 1: links of one robot collide with each other (asset.self_collisions == 0, legged_robot.py:683; anymal_c_flat_config.py:42) */
    float   terrain_env_length, max_episode_length_s;
    /* reset: legged_robot.py:397-436 */
    float   base_init_state[13], _padf4;
    uint64_t seed;
} lg_params;

/* Raw device pointers to caller-owned (torch-allocated) buffers.  Layouts are
 * the reference's so the Python views stay valid. */
typedef struct lg_buffers {
    float   *root_states;       /* [N,13] pos, quat xyzw, lin vel, ang vel (world)       T1 */
    float   *dof_state;         /* [N*ndof,2] (pos,vel) interleaved                      T2 */
    float   *contact_forces;    /* [N,num_bodies,3]                                      T3 */
    float   *obs_buf;           /* [N,num_obs]                                           T4 */
    float   *rew_buf;           /* [N] */
    uint8_t *reset_buf;         /* [N] bool */
    uint8_t *time_out_buf;      /* [N] bool */
    int64_t *episode_length_buf;/* [N] */
    float   *torques, *actions, *last_actions, *last_dof_vel;   /* [N,ndof]              T5 */
    float   *last_root_vel;     /* [N,6] */
    float   *commands;          /* [N,4] */
    float   *feet_air_time;     /* [N,K] */
    uint8_t *last_contacts;     /* [N,K] bool */
    float   *base_lin_vel, *base_ang_vel, *projected_gravity;   /* [N,3] */
    float   *measured_heights;  /* [N,num_height_points] or NULL */
    float   *sea_hidden_state, *sea_cell_state;                 /* [2,N*ndof,8] or NULL  T6 */
    float   *episode_sums;      /* [num_reward_slots,N]                                  T7 */
    float   *episode_means;     /* [num_reward_slots+1] extras["episode"]: mean episode sum / max_episode_length_s over the
                                   envs that reset in the latest step that had resets (legged_robot.py:179-183), stale
                                   otherwise; last entry = mean terrain level (:186) */
    float   *extras_accum;      /* [num_reward_slots+1] scratch: per-step sums + count, zeroed by the library */
    int64_t *step_counter;      /* [1] device copy of common_step_counter (legged_robot.py:115); lets a captured HIP graph
                                   replay lg_step without new kernel arguments (pass common_step_counter = -1) */
    float   *env_origins;       /* [N,3] */
    int32_t *terrain_levels, *terrain_types;                    /* [N] or NULL           T9 */
    const float   *terrain_origins;   /* [rows,cols,3] or NULL */
    const int16_t *height_samples;    /* [hf_rows,hf_cols] or NULL                       T8 */
    const float   *friction_coeffs;   /* [N] per-env shape friction (legged_robot.py:261-285) */
    const float   *base_mass_delta;   /* [N] added base mass (legged_robot.py:316-327) */
} lg_buffers;

typedef struct lg_sim lg_sim;

/* create_sim + prepare_sim (base_task.py:42,88; legged_robot.py:232-251, 657-750):
 * copies model, params and the actuator weights to the device. */
int  lg_create(const lg_params *params, const lg_robot_model *model,
               const float *actuator_weights /* LG_ACTUATOR_FLOATS or NULL */,
               int device_id, lg_sim **out);
void lg_destroy(lg_sim *sim);

/* acquire_*_tensor + wrap_tensor (legged_robot.py:515-529): bind caller buffers. */
int  lg_bind(lg_sim *sim, const lg_buffers *buffers);

/* LeggedRobot.step (legged_robot.py:80-104) for all envs, fused:
 *   clip actions; decimation x (_compute_torques -> set_dof_actuation_force ->
 *   simulate -> refresh_dof_state); post_physics_step (incl. reset_idx for
 *   terminated envs and compute_observations); clip obs.
 * `actions` is [N,ndof] on the device.  `common_step_counter` is the value
 * AFTER the increment of legged_robot.py:115, or -1 to use (device step_counter + 1): the
 * library stores the value it used back into step_counter, so graph replays self-advance.
 * Asynchronous on `stream`. */
int  lg_step(lg_sim *sim, const float *actions, int64_t common_step_counter, void *stream);

/* Deferred extras (rollout graphs).  lg_step turns the finished episodes' sums into extras["episode"] (`episode_means`,
 * legged_robot.py:179-186) before it returns, which costs every launch a serial tail: the workgroup that finishes last
 * reads the accumulators past the caches.  With lg_set_deferred_extras(sim, 1) a step leaves its sums in one of two
 * accumulator slots (step parity) and the NEXT step's launch publishes them while it runs; lg_extras_flush publishes what the
 * last step left (pass the same common_step_counter convention as lg_step: the last step's value, or -1 = device counter).
 * Callers that read extras after every step (the eager env.step) keep the default (0); a captured rollout graph switches
 * it on for its steps and ends with one lg_extras_flush node. */
int  lg_set_deferred_extras(lg_sim *sim, int32_t on);
int  lg_extras_flush(lg_sim *sim, int64_t common_step_counter, void *stream);

/* reset_idx on an explicit env list (base_task.py:114-118 reset(); device int32 ids). */
int  lg_reset_idx(lg_sim *sim, const int32_t *env_ids, int32_t count,
                  int64_t common_step_counter, void *stream);

/* Sub-path entry points (parity tests drive them one at a time). */
int  lg_actuator_forward(lg_sim *sim, const float *pos_err, const float *vel, float *torques,
                         float *hidden, float *cell, int32_t rows, void *stream);       /* anymal.py:71-78 */
int  lg_physics_substep(lg_sim *sim, const float *torques, int32_t write_contacts, void *stream); /* legged_robot.py:92-96 */
int  lg_compute_observations_only(lg_sim *sim, int64_t common_step_counter, void *stream);        /* legged_robot.py:212-230 */
/* reset_idx evaluates update_command_curriculum BEFORE _resample_commands (legged_robot.py:159-176).  The fused step resets inside the
 * launch; when the host rule then widened the ranges (lg_set_params), this re-draws the commands of the envs whose reset_buf the step set
 * from the new ranges -- same Philox block as the in-step draw -- and patches the command slots of the bound observation buffer. */
int  lg_resample_reset_commands(lg_sim *sim, int64_t common_step_counter, void *stream);

/* Redirect where the next lg_step / lg_compute_observations_only writes observations ([N,num_obs] device buffer).
 * The reference re-creates obs_buf every step (legged_robot.py:215) and rsl_rl keeps a reference to the previous
 * one until process_env_step; the Python surface therefore ping-pongs two buffers through this call. */
int  lg_set_obs_buffer(lg_sim *sim, float *obs_buf);

/* Update params that the Python surface may change between steps
 * (command ranges by the curriculum, legged_robot.py:471-483). */
int  lg_set_params(lg_sim *sim, const lg_params *params);

/* Fused rollout-time actor: actions = MLP(obs) + std * eps (rsl_rl ActorCritic.act, [EXTERNAL]; widths from
 * legged_robot_config.py:204-209 / anymal_c_flat_config.py:62-65), ELU hidden activations, 3 hidden layers.
 * `weights[i]`/`biases[i]` are HOST arrays in torch.nn.Linear layout ([out,in] row-major / [out]); they are repacked into
 * the MFMA operand layout and copied to the device.  Supported widths: hidden multiples of 16, <= 512. */
typedef struct lg_policy lg_policy;
int  lg_policy_create(const int32_t dims[5] /* obs, h1, h2, h3, actions */, const float *const weights[4],
                      const float *const biases[4], const float *std, int device_id, lg_policy **out);
void lg_policy_destroy(lg_policy *p);
/* obs [N,dims[0]] -> actions [N,dims[4]] (and mean if non-null), device pointers.  `step` selects the noise stream
 * (>= 0) or, when -1, (*step_counter + 1) is read on the device (graph replay).  deterministic != 0 returns the mean. */
int  lg_policy_act(lg_policy *p, const float *obs, float *actions, float *mean, int32_t num_envs, uint64_t seed,
                   int64_t step, const int64_t *step_counter, int32_t deterministic, void *stream);

/* Refresh an existing policy from DEVICE tensors (torch Linear layout [out, in], biases [out], std [num_actions]) without a host
 * round trip: one pack kernel per layer on `stream`.  Same dims as at lg_policy_create.  For training loops that re-use
 * the MFMA actor for every rollout (rl/runner.py). */
int  lg_policy_load_device(lg_policy *p, const float *const weights[4], const float *const biases[4], const float *std, void *stream);

/* Fused rollout step: actions = actor(obs) + std * eps (as lg_policy_act, same noise stream keyed by the step counter)
 * immediately followed by lg_step on those actions, in ONE launch.  `obs` is the observation tensor of the previous step
 * (it may be the sim's own obs output buffer: a workgroup reads only its envs' rows, and before it rewrites them),
 * `actions` / `mean` receive what rsl_rl's storage needs.
 * Compiled for the flat actor (48-128-64-32) on the quadruped actuator-net plane kernel; other combinations return -4 and
 * the caller uses lg_policy_act + lg_step (reference: ActorCritic.act + LeggedRobot.step, legged_robot.py:80-104). */
int  lg_step_policy(lg_sim *sim, lg_policy *p, const float *obs, float *actions, float *mean, uint64_t seed,
                    int32_t deterministic, int64_t common_step_counter, void *stream);

/* A whole rollout segment in ONE launch: `steps` consecutive fused policy steps (actor + sampling + lg_step) -- what the caller's
 * rollout loop does step by step ([EXTERNAL] rsl_rl OnPolicyRunner.learn: `for i in range(num_steps_per_env): actions =
 * alg.act(obs, ...); obs, ..., = env.step(actions)`, entered from legged_gym/scripts/train.py:43), with identical results: the same
 * state, observations, actions, rewards and dones as `steps` lg_step_policy calls, bit for bit.  Envs never interact, so every
 * workgroup walks through the steps of its own envs without meeting the others at each step boundary: the launch ends with the
 * slowest SUM over the steps instead of the sum of the slowest workgroups.  Per-step outputs go to caller-owned rollout storage:
 * obs[steps + 1][N][num_obs] (obs[0] = the input of the first step, step t writes obs[t + 1]), actions / mean[steps][N][num_actions]
 * (mean may be null), rew[steps][N], dones / time_outs[steps][N] (uint8).  The bound state buffers hold the state after the last
 * step; extras["episode"] (episode_means) is that of the last step of the segment in which an env was reset (legged_robot.py:179-183);
 * the device step counter advances by `steps`.  Compiled for the shape lg_step_policy is compiled for (48-128-64-32 actor,
 * quadruped actuator-net kernel on the plane, no height measurements); -4 otherwise.  The first call on a handle allocates a small
 * workspace (make it outside stream capture). */
#define LG_MAX_ROLL_STEPS 256
typedef struct {
    int32_t  steps;
    float   *obs;
    const float *obs0;     /* optional: the input of the first step lives HERE (typically obs[steps] of the previous segment in the same storage);
                              every workgroup copies its envs' rows to obs[0] while it reads them, so no copy kernel is needed between segments */
    float   *actions;
    float   *mean;
    float   *rew;
    uint8_t *dones;
    uint8_t *time_outs;
} lg_rollout_buffers;
int  lg_rollout_policy(lg_sim *sim, lg_policy *p, const lg_rollout_buffers *out, uint64_t seed, int32_t deterministic,
                       int64_t common_step_counter, void *stream);

/* GAE(gamma, lambda) scan over a rollout, one thread per env (rsl_rl RolloutStorage.compute_returns, [EXTERNAL]; hyper-
 * parameters legged_robot_config.py:222-223).  rewards, values, returns, advantages: float [T, N]; dones: uint8 [T, N];
 * last_values: float [N].  returns[t] = A_t + V_t with A_t = delta_t + gamma lam (1 - done_t) A_{t+1}; advantages = returns -
 * values (un-normalised: the caller normalises over the global batch).  Device pointers, asynchronous on `stream`. */
int  lg_gae_returns(const float *rewards, const float *values, const uint8_t *dones, const float *last_values, float gamma, float lam,
                    float *returns, float *advantages, int32_t num_steps, int32_t num_envs, void *stream);

/* One ActorCritic MLP (Linear/ELU x3 + Linear; rsl_rl ActorCritic.actor / .critic, [EXTERNAL], widths from
 * legged_robot_config.py:204-209) for the learner kernels below.  weights[l] is torch's nn.Linear weight [dims[l+1], dims[l]]
 * row-major, biases[l] its bias; all device pointers. */
typedef struct lg_mlp_net {
    const float *weights[4];
    const float *biases[4];
    float       *grad_weights[4];   /* lg_mlp_backward outputs (same layouts), overwritten */
    float       *grad_biases[4];
    const float *input;             /* [R, dims[0]] rows (e.g. the flattened rollout storage) */
    float       *output;            /* lg_mlp_forward: [mb, dims[4]] */
    const float *grad_output;       /* lg_mlp_backward: dL/d output [mb, dims[4]] */
    int32_t      dims[5];
} lg_mlp_net;

/* Mini-batch forward of 1 or 2 MLPs on the matrix cores: output[i] = net(input[rows[i]]) (rows == NULL: input[i]), i < mb.
 * This is what `actor(obs_batch)` / `critic(critic_obs_batch)` compute inside rsl_rl PPO.update() ([EXTERNAL]).
 * Returns -4 for MLP shapes that are not built (callers then keep their autograd path).  Asynchronous, capturable. */
int  lg_mlp_forward(const lg_mlp_net *nets, int32_t n_nets, const int64_t *rows, int32_t mb, void *stream);
/* Gradients of all weights and biases given dL/d output, i.e. what loss.backward() leaves in .grad for these modules
 * (forward activations are recomputed in LDS, nothing is stored between the two calls).  `workspace` holds per-workgroup
 * partial sums (lg_mlp_workspace_bytes); the reduction order is fixed, so results are bit-reproducible. */
size_t lg_mlp_workspace_bytes(const lg_mlp_net *nets, int32_t n_nets);
int  lg_mlp_backward(const lg_mlp_net *nets, int32_t n_nets, const int64_t *rows, int32_t mb, float *workspace, size_t workspace_bytes,
                     void *stream);

/* The same two passes for MLPs of ANY widths (the 512-256-128 networks of anymal_c_rough / cassie / a1 / anymal_b, reference
 * legged_robot_config.py:205-208).  Backward: every layer is a tiled MFMA GEMM with ELU' / bias sums fused into its epilogue
 * (csrc/lg_gemm.h).  Forward: the same per-layer GEMMs at precision 0; at precision 1 the [235 | 169]-512-256-128-[<= 16] shapes
 * run as ONE chain kernel (csrc/lg_policy.h: k_mlp_chain_fwd64, 64 rows per workgroup through all four layers).  The activations of
 * the forward pass stay in `workspace` (lg_mlp_wide_workspace_bytes(nets, n_nets, mb) bytes) for the backward pass that follows;
 * weight gradients are summed in a fixed order (bit-reproducible). */
size_t lg_mlp_wide_workspace_bytes(const lg_mlp_net *nets, int32_t n_nets, int32_t mb);
/* Arithmetic of the lg_mlp_wide_* kernels AND of lg_policy_act's wide actor kernel: 0 = exact f32 MFMA (bitwise a k-ordered fmaf
 * chain); 1 (default) = split-bf16 on the bf16 matrix cores, every f32 operand as hi + lo bf16 and every product as
 * hi*hi + hi*lo + lo*hi with f32 accumulation (relative error of a product ~2^-15).  Process-wide; returns the previous setting. */
int  lg_mlp_wide_set_precision(int mode);
int  lg_mlp_wide_forward(const lg_mlp_net *nets, int32_t n_nets, const int64_t *rows, int32_t mb, float *workspace, size_t workspace_bytes,
                         void *stream);
int  lg_mlp_wide_backward(const lg_mlp_net *nets, int32_t n_nets, const int64_t *rows, int32_t mb, float *workspace, size_t workspace_bytes,
                          void *stream);

/* What one PPO mini-batch step reads besides the observations (rsl_rl PPO.update [EXTERNAL]; hyper-parameters
 * legged_robot_config.py:215-228): the rollout storage (flattened [B, .] device arrays, gathered through `rows`), the policy's std
 * parameter and the loss coefficients; and what it produces besides the network gradients. */
typedef struct lg_ppo_batch {
    const float *actions, *old_log_prob, *old_mu, *old_sigma;    /* [B, A], [B], [B, A], [B, A] */
    const float *advantages, *old_values, *returns;              /* [B] */
    const float *std;                                            /* [A] */
    float        clip, value_coef, entropy_coef;
    int32_t      use_clipped_value;
    float       *d_std;                                          /* out [A]: dL/dstd */
    float       *stats;                                          /* out [4]: surrogate mean, value-loss mean, KL mean, entropy */
    float       *loss_acc;                                       /* optional [2]: += {value-loss mean, surrogate mean} (running sums of an update) */
} lg_ppo_batch;
/* lg_mlp_forward + lg_ppo_loss + lg_mlp_backward in ONE pass over the mini-batch: nets[0] = actor, nets[1] = critic (one output).
 * The networks' outputs never leave the kernel: after the forward pass of a 16-row tile the loss gradient w.r.t. mu / value is
 * evaluated in registers and fed to the backward pass.  Writes every weight / bias gradient, d_std and stats; `output` and
 * `grad_output` of the nets are not used.  Same loss expressions and same results as the three separate calls (tests). */
int  lg_ppo_minibatch(const lg_mlp_net *nets, const int64_t *rows, int32_t mb, const lg_ppo_batch *batch, float *workspace,
                      size_t workspace_bytes, void *stream);

/* Diagnostic (tools/mlp_probe.py): while `buf` (device, uint64[60]) is set, lg_mlp_forward / lg_mlp_backward launches write the
 * s_memtime stamps of their phases as seen by thread 0 of workgroup (0, 0); NULL stops. */
void lg_mlp_trace(unsigned long long *buf);

/* One transition of the PPO rollout, as rsl_rl's RolloutStorage.add_transitions() + OnPolicyRunner.learn()'s episode
 * bookkeeping consume it ([EXTERNAL]): the observation the policy saw, its sampled action and mean, and the step's reward /
 * done / time-out flags (LeggedRobot.step outputs, reference legged_robot.py:106-127).  Device pointers. */
typedef struct lg_rollout_step {
    const float   *obs;               /* [N, num_obs]  observation BEFORE the step */
    const float   *actions;           /* [N, num_actions] */
    const float   *mean;              /* [N, num_actions] */
    const float   *rewards;           /* [N] */
    const uint8_t *dones;             /* [N] reset_buf */
    const uint8_t *time_outs;         /* [N] time_out_buf, may be NULL */
    float   *storage_obs, *storage_actions, *storage_mu, *storage_rewards;   /* slices [t] of the rollout storage */
    uint8_t *storage_dones;
    float   *storage_time_outs;       /* [N] 0/1, may be NULL */
    float   *cur_return, *cur_length; /* [N] running episode return / length, may both be NULL */
    float   *sums;                    /* [3] += {return, length, 1} of every episode that ended this step */
    const float *std;                 /* [num_actions] policy std, may be NULL; with it the kernel also stores ...           */
    float   *storage_sigma;           /* ... [N, num_actions] the broadcast std and                                          */
    float   *storage_log_prob;        /* ... [N] log N(action; mean, std) summed over actions (rsl_rl PPO.act, [EXTERNAL])   */
    int32_t  num_envs, num_obs, num_actions;
} lg_rollout_step;
/* Store the transition and update the episode statistics in one launch.  Asynchronous, capturable. */
int  lg_rollout_record(const lg_rollout_step *step, void *stream);

/* The same bookkeeping for a whole segment that lg_rollout_policy wrote straight into the rollout storage (observations, actions,
 * means, rewards, dones are already in place): the broadcast std and log N(action; mean, std) of every transition, the 0/1 time-out
 * floats PPO bootstraps with, and the episode statistics (each env's transitions walked in step order) -- ONE launch per segment
 * instead of one lg_rollout_record per step.  Arrays are [steps][N][...] as lg_rollout_buffers.  Asynchronous, capturable. */
typedef struct lg_rollout_post {
    const float   *actions, *mean;    /* [steps][N][num_actions] */
    const float   *rewards;           /* [steps][N] */
    const uint8_t *dones, *time_outs; /* [steps][N]; time_outs may be NULL */
    const float   *std;               /* [num_actions] */
    float   *sigma;                   /* out [steps][N][num_actions] */
    float   *log_prob;                /* out [steps][N] */
    float   *time_outs_f;             /* out [steps][N] 0/1, may be NULL */
    float   *cur_return, *cur_length; /* [N] running episode return / length, may both be NULL */
    float   *sums;                    /* [3] += {return, length, 1} of every episode that ended inside the segment */
    int32_t  steps, num_envs, num_actions;
} lg_rollout_post;
int  lg_rollout_finish(const lg_rollout_post *post, void *stream);

/* One parameter tensor of torch.optim.Adam(capturable=True): the parameter, its .grad and the optimiser state
 * (state['exp_avg'], state['exp_avg_sq'], state['step'] -- a float32 device scalar).  All updated in place. */
typedef struct lg_adam_tensor {
    float       *param;
    const float *grad;
    float       *exp_avg;
    float       *exp_avg_sq;
    float       *step;
    int64_t      numel;
} lg_adam_tensor;

/* The tail of rsl_rl PPO.update()'s mini-batch step ([EXTERNAL]) in three launches: nn.utils.clip_grad_norm_(params, max_grad_norm)
 * over all `tensors` together, the adaptive-KL learning-rate rule (kl == NULL or desired_kl <= 0: fixed schedule;
 * legged_robot_config.py:221-226) on the device scalar `lr`, and torch.optim.Adam.step() (weight_decay 0, amsgrad off) on the
 * clipped gradients.  scratch [LG_ADAM_SCRATCH_FLOATS] is work space; its first two floats receive {total gradient norm, clip
 * coefficient}.  n_tensors <= 32.  Three launches, fixed summation order (bit-reproducible).  Capturable. */
#define LG_ADAM_SCRATCH_FLOATS 2050
int  lg_adam_step(const lg_adam_tensor *tensors, int32_t n_tensors, float *lr, float beta1, float beta2, float eps, float max_grad_norm,
                  const float *kl, float desired_kl, float *scratch, void *stream);

/* Fused PPO loss and its gradient w.r.t. the network outputs for one mini-batch (rsl_rl PPO.update's surrogate / clipped value /
 * entropy terms and the adaptive-KL statistic, [EXTERNAL]; hyper-parameters legged_robot_config.py:215-228):
 *   loss = mean(max(-A r, -A clamp(r, 1-c, 1+c))) + value_coef mean(max((v-R)^2, (v_clip-R)^2)) - entropy_coef mean(H),
 *   r = exp(log pi(a) - old_log_prob), pi = N(mu, std), v_clip = v_old + clamp(v - v_old, -c, c).
 * mu [mb, A] / value [mb] are the actor / critic outputs of mini-batch row i; `rows[i]` is its row in the rollout storage, from
 * which actions, old_log_prob, old_mu, old_sigma [B, A] and advantages, old_values, returns [B] are gathered.  Outputs:
 * d_mu [mb, A], d_value [mb], d_std [A] = d loss / d(.)  and  stats[4] = {surrogate mean, value-loss mean, KL mean, entropy}.
 * Device pointers; asynchronous on `stream` (capturable).  use_clipped_value != 0 selects the clipped value loss. */
int  lg_ppo_loss(const float *mu, const float *std, const float *value, const int64_t *rows, const float *actions,
                 const float *old_log_prob, const float *old_mu, const float *old_sigma, const float *advantages,
                 const float *old_values, const float *returns, float clip, float value_coef, float entropy_coef,
                 int32_t use_clipped_value, float *d_mu, float *d_std, float *d_value, float *stats, int32_t mb, int32_t num_actions,
                 void *stream);

/* Sticky device status.  The workgroups of lg_step hand data between their waves through LDS flags with BOUNDED polls; a poll
 * that runs out (never observed; it would mean a lost store or a scheduling fault) lets the wave go on -- a hung CU is worse --
 * but ORs a bit into a host-mapped status word of the handle.  From then on EVERY call on the handle returns -20 (text in
 * lg_last_error()) until lg_clear_device_status(): rc 0 never covers a step whose physics missed a hand-over once the host has
 * seen the word, i.e. at the latest on the first call after the next synchronisation.  lg_device_status() returns the word
 * (>= 0; 0 = clean), optionally after hipDeviceSynchronize().  No counterpart in the reference (PhysX reports nothing).
 * lg_debug_handover() is the test hook that provokes the condition (skip: 1 = frame flag withheld, 2 = self-collision flags
 * withheld; spin_limit: poll bound, <= 0 restores 2^22). */
#define LG_STATUS_FRAME_HANDOVER_TIMEOUT  0x1u
#define LG_STATUS_SELF_COLLISION_TIMEOUT  0x2u
int  lg_device_status(lg_sim *sim, int32_t synchronize);
int  lg_clear_device_status(lg_sim *sim);
int  lg_debug_handover(lg_sim *sim, int32_t skip, int32_t spin_limit);

const char *lg_last_error(void);
int  lg_abi_version(void);
/* sizeof of the ABI structs (0 params, 1 robot_model, 2 buffers, 3 point, 4 mlp_net, 5 adam_tensor, 6 rollout_step, 7 ppo_batch;
 * the CPU oracle knows 0-3) so a binding can verify its layout. */
int  lg_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif /* LEGGED_HIP_H */
