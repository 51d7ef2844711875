/*
 * hgym.h -- C-ABI of libhgym_hip.so: the MI355X (gfx950) hot path of roboterax/humanoid-gym.
 *
 * The reference (pure Python/PyTorch) has no FFI of its own; its replaceable seam is the Python API
 * of humanoid.envs / humanoid.algo.ppo.  Each entry point below states the reference
 * interface (file:line under /root/reference/humanoid) it stands in for.  INTEGRATION.md shows the
 * ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns 0 (HGYM_OK) or a negative HGYM_E_* code; hgym_last_error() gives the
 *     thread-local message of the last failure.
 *   - the CALLER owns every buffer (PyTorch allocates them); the library never allocates or frees
 *     device memory and keeps no pointer past the call.
 *   - all device work is enqueued on the hipStream_t passed as `void* stream`; no call synchronises,
 *     all calls are hipGraph-capturable, and per-step counters live in device memory so that a
 *     replayed graph advances them.
 *   - plain pointers and sizes only: no torch types cross this boundary.
 *   - per-env state is env-major SoA: a field with C components is a [C][N] fp32 array, so that
 *     lane i of a wavefront touches env i of every component (coalesced).  The (N,C) tensors of the
 *     reference API are transposed views of these arrays.
 */
#ifndef HGYM_H_
#define HGYM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HGYM_VERSION 9      /* 2: HgymEnvOut carries the logging sink; hgym_rollout_*, hgym_ppo_grad_part, hgym_net_param_offset
                             * 3: the rollout scratch block grows with the env count (HGYM_ROLLOUT_SCRATCH_BYTES(num_envs))
                             * 4: bf16 observation shadow: HgymObsShadow argument of the policy launches, HgymBatch.obs_bf16 /
                             *    priv_bf16, hgym_net_shadow_ld
                             * 5: HgymEnvOut.obs_ahead / obs_older_ready (hgym_rollout_step writes the older frames one launch ahead)
                             * 6: HgymEnvOut.l0_ahead / l0_ready / obs_bf16_ahead (the actor's first layer carried across the launches of a rollout);
                             *    HGYM_MAX_CUSTOM_REWARDS 8 -> 24, custom_reward_pos = 23: after the clip (`termination`);
                             *    (same version, later: frames written into obs_ahead / priv_ahead for an env that resets in the same step are
                             *    zeroed by the NEXT hgym_rollout_step call -- see HgymEnvOut.obs_ahead; layouts and call sequence unchanged)
                             * 7: HgymComm.wait_ticks (bound of the direct exchange's waits) + hgym_comm_status; a communicator stays usable
                             *    after an expired wait (the done counter is per call); HgymEnvOut.t_time_outs, hgym_critic_values,
                             *    hgym_gae_bootstrap and hgym_rollout_step with values = NULL (the critic run once after the rollout)
                             * 8: hgym_gae / hgym_gae_bootstrap: `stats` is HGYM_GAE_STATS_DOUBLES(n) doubles (per-workgroup partial sums + arrival
                             *    counter behind the three results), the advantage statistics are summed in a fixed order
                             * 9: hgym_randperm_dev (draw number on the device); hgym_comm_allreduce(seq = 0): the call number kept on the device
                             *    (status[2]); hgym_comm_sum64 -- what a captured update needs: no launch argument changes between iterations */

enum {
    HGYM_OK = 0,
    HGYM_E_BADARG = -1,      /* null pointer, negative size ... */
    HGYM_E_SHAPE = -2,       /* sizes inconsistent with the configuration */
    HGYM_E_LAUNCH = -3,      /* HIP reported an error at launch */
    HGYM_E_UNSUPPORTED = -4, /* configuration outside what the kernels were built for */
    HGYM_E_NODEVICE = -5     /* no gfx950 device visible */
};

#define HGYM_NUM_DOF 12
#define HGYM_NUM_ACTIONS 12
#define HGYM_NUM_BODIES 13
#define HGYM_NUM_REWARDS 22
#define HGYM_OBS_FRAME 47   /* num_single_obs,           envs/custom/humanoid_config.py:41 */
#define HGYM_PRIV_FRAME 73  /* single_num_privileged_obs, envs/custom/humanoid_config.py:43 */
#define HGYM_MAX_LAYERS 8
#define HGYM_MAX_CUSTOM_REWARDS 24  /* user-defined reward terms per env (HgymEnvConfig.num_custom_rewards): room for a task class that
                                      * defines every one of its terms itself, the reference's way */

int32_t hgym_version(void);
const char* hgym_last_error(void);
/* number of compute units of the current device (0 if none); also a cheap "is there a GPU" probe */
int32_t hgym_device_cus(void);
/* sizeof() of a struct of this header by name ("HgymEnvConfig", ...), -1 if unknown: lets a binding in
 * another language verify its mirror of the layouts */
int64_t hgym_sizeof(const char* name);

/* ------------------------------------------------------------------------------------------------
 * Env configuration = the constants of XBotLCfg the hot path reads
 * (envs/custom/humanoid_config.py:34-227, envs/base/legged_robot.py:710-720).
 * ---------------------------------------------------------------------------------------------- */
typedef struct HgymEnvConfig {
    int32_t num_envs;
    int32_t frame_stack;          /* 15 */
    int32_t c_frame_stack;        /* 3  */
    int32_t decimation;           /* 10 */
    float sim_dt;                 /* 0.001 */
    float dt;                     /* decimation*sim_dt as fp32(python double product) = 0.01f */
    int32_t max_episode_length;   /* 2400 */
    int32_t resample_steps;       /* 800 */
    int32_t push_interval;        /* 400 */
    int32_t push_robots;          /* 1 */
    int32_t add_noise;            /* 1 */
    int32_t heading_command;      /* 1 for XBot-L: the yaw-rate command follows a sampled heading (legged_robot.py:311-314); 0: the yaw
                                     rate itself is sampled from cmd_yaw_lo/span (:333-334) -- one of the generic options below */
    int32_t use_ref_actions;      /* 0 for XBot-L (humanoid_config.py:49); 1: actions += 2 * ref_dof_pos, IN PLACE, before the clip
                                     (humanoid_env.py:190-191) */
    float clip_actions, clip_obs; /* 18, 18 */
    float action_scale;           /* 0.25 */
    float action_delay, action_noise; /* 0.5, 0.02 */
    float noise_level;            /* 0.6 */
    float obs_noise[HGYM_OBS_FRAME];   /* noise_scale_vec, humanoid_env.py:166-186 */
    float scale_lin_vel, scale_ang_vel, scale_dof_pos, scale_dof_vel, scale_quat;
    /* torch_rand_float(lo,hi) = (hi-lo)*u + lo with (hi-lo) formed in python double: carry lo and the
       double-rounded span so the fp32 arithmetic is the reference's (legged_robot.py:328-331,367) */
    float cmd_x_lo, cmd_x_span, cmd_y_lo, cmd_y_span, cmd_h_lo, cmd_h_span;
    float dof_reset_lo, dof_reset_span;       /* -0.1, 0.2 */
    float push_vel_lo, push_vel_span;         /* -0.2, 0.4   humanoid_env.py:86-89 */
    float push_ang_lo, push_ang_span;         /* -0.4, 0.8   humanoid_env.py:92-93 */
    float p_gains[HGYM_NUM_DOF], d_gains[HGYM_NUM_DOF], torque_limits[HGYM_NUM_DOF];
    float default_dof_pos[HGYM_NUM_DOF];
    float dof_lower[HGYM_NUM_DOF], dof_upper[HGYM_NUM_DOF]; /* synthetic physics only */
    float base_init_state[13];
    int32_t base_body, feet_bodies[2], knee_bodies[2];
    /* rewards (humanoid_config.py:174-216); scales already multiplied by dt, alphabetical order */
    float reward_scales[HGYM_NUM_REWARDS];
    int32_t only_positive_rewards;
    float base_height_target, min_dist, max_dist, target_joint_pos_scale, target_feet_height;
    float cycle_time, tracking_sigma, max_contact_force, episode_length_s;
    uint64_t seed;                /* Philox key for internally generated noise */
    /* ---- generic LeggedRobot options XBot-L leaves off.  All zero = plane terrain, fixed command
     * ranges: the XBot-L default, and the configuration the headline numbers are measured on. ---- */
    int32_t custom_origins;       /* terrain.mesh_type in {heightfield, trimesh} (legged_robot.py:687-697): a reset spawns the robot at
                                     its tile origin + U(-1,1) m in x and y (:382-385) */
    int32_t terrain_curriculum;   /* terrain.curriculum: on reset an env that walked more than half a tile is promoted one terrain
                                     level, one that covered less than half its commanded distance demoted; solving the last level
                                     redraws a random one (:176-177,400-420).  Needs custom_origins. */
    int32_t terrain_rows, terrain_cols;   /* levels x types of HgymEnvState.terrain_origins; max_terrain_level = terrain_rows (:695) */
    float terrain_env_length;     /* Terrain.env_length (utils/terrain.py:46) */
    int32_t num_height_points;    /* > 0: terrain.measure_heights with that many sample points per env (:316-317,743-795) */
    int32_t height_rows, height_cols;     /* shape of HgymEnvState.height_samples (Terrain.tot_rows, tot_cols) */
    float terrain_border, terrain_hscale, terrain_vscale;   /* terrain.border_size / horizontal_scale / vertical_scale */
    float cmd_yaw_lo, cmd_yaw_span;   /* commands.ranges.ang_vel_yaw, used when heading_command = 0 */
    int32_t command_curriculum;   /* commands.curriculum: every max_episode_length common steps, if the resetting envs' mean
                                     tracking_lin_vel episode sum exceeds 80 % of its maximum, lin_vel_x widens by 0.5 each way
                                     up to +-max_curriculum (:179-180,422-431).  The live range is HgymEnvState.command_range_x. */
    float max_curriculum;
    /* User-defined reward terms (legged_robot.py:518-541: every non-zero entry of cfg.rewards.scales names a method
     * `_reward_<name>`, found by name).  The 22 XBot-L terms are built in; any OTHER name is evaluated by the caller between
     * hgym_env_step_begin and hgym_env_step_end (what the reference has computed by the time compute_reward runs is then in the
     * state) and handed in through HgymEnvState.custom_rew.  custom_reward_pos[j] = how many built-in terms precede custom
     * term j in the alphabetical order the reference sums in (0 .. 22): the fp32 sum is formed in that merged order.
     * 23 (= HGYM_NUM_REWARDS + 1): the term is added AFTER the only_positive_rewards clip -- the reference's `termination`
     * (legged_robot.py:229-235; it is skipped in the function list, :533-534). */
    int32_t num_custom_rewards;
    int32_t custom_reward_pos[HGYM_MAX_CUSTOM_REWARDS];
} HgymEnvConfig;

/* fills *cfg with the XBot-L values (the reference defaults) for num_envs environments */
int32_t hgym_env_config_default(HgymEnvConfig* cfg, int32_t num_envs);

/* A strided view of one simulator tensor: element (env, comp) is base[env*env_stride + comp*comp_stride].
 * Isaac Gym's AoS buffers (legged_robot.py:449-457) and the library's own SoA synthetic backend are both
 * expressible.  comps: root 13 | dof_pos 12 | dof_vel 12 | contact body*3+axis (39) | rigid body*13+c (169). */
typedef struct HgymStrided {
    float* base;
    int64_t env_stride;
    int64_t comp_stride;
} HgymStrided;

typedef struct HgymSimTensors {
    HgymStrided root, dof_pos, dof_vel, contact, rigid;
} HgymSimTensors;

/* Per-env state, every field [C][N] fp32 SoA unless noted (legged_robot.py:434-516, base_task.py:71-94,
 * humanoid_env.py:78-79). */
typedef struct HgymEnvState {
    int64_t* episode_length;   /* [N] int64, VecEnv.episode_length_buf */
    int64_t* counters;         /* [4] int64 device scalars: [0] common_step_counter, [1] resets this step,
                                  [2] ring step (frames pushed so far), [3] reserved */
    float* commands;           /* 4 */
    float* actions;            /* 12 */
    float* last_actions;       /* 12 */
    float* last_last_actions;  /* 12 */
    float* last_dof_vel;       /* 12 */
    float* last_root_vel;      /* 6 */
    float* torques;            /* 12 */
    float* feet_air_time;      /* 2 */
    float* last_contacts;      /* 2 (0/1) */
    float* feet_height;        /* 2 */
    float* last_feet_z;        /* 2 */
    float* ref_dof_pos;        /* 12 */
    float* push_force;         /* 3 */
    float* push_torque;        /* 3 */
    float* episode_sums;       /* 22 */
    float* base_lin_vel;       /* 3 */
    float* base_ang_vel;       /* 3 */
    float* projected_gravity;  /* 3 */
    float* base_euler;         /* 3 */
    float* friction;           /* 1 */
    float* body_mass;          /* 1 */
    float* env_origins;        /* 3 */
    float* obs_ring;           /* [N][frame_stack][47]   unclipped noisy frames, ring slot = ring step % frame_stack */
    float* priv_ring;          /* [N][c_frame_stack][73] */
    float* episode_acc;        /* [24] fp32 device scalars: sum over resetting envs of episode_sums[k]; [22] unused */
    /* generic options (HgymEnvConfig tail); NULL when the option is off */
    int64_t* terrain_levels;        /* (N,) int64  LeggedRobot.terrain_levels */
    const int64_t* terrain_types;   /* (N,) int64  LeggedRobot.terrain_types */
    const float* terrain_origins;   /* (terrain_rows, terrain_cols, 3) fp32 = Terrain.env_origins */
    const int16_t* height_samples;  /* (height_rows, height_cols) int16 = Terrain.heightsamples */
    const float* height_points;     /* (num_height_points, 3) base-frame sample points, the same for every env (:743-759) */
    float* height_pose;             /* (N, 7) scratch: base position + quaternion at the point of the step where the reference
                                       samples the heights (before the reset overwrites them) */
    float* measured_heights;        /* (N, num_height_points) fp32 = LeggedRobot.measured_heights */
    double* command_range_x;        /* [2] device doubles: command_ranges["lin_vel_x"] = [lo, hi]; moved by the command curriculum */
    /* user-defined reward terms (HgymEnvConfig.num_custom_rewards = K > 0; NULL otherwise) */
    const float* custom_rew;        /* (K, N) this step's terms, already times scale * dt: written by the caller between begin and end */
    float* custom_sums;             /* (K, N) their episode sums (episode_sums of the reference, legged_robot.py:225-227) */
    float* custom_acc;              /* (K,)   sum over the envs resetting this step of their episode sums (-> HgymEnvOut.extras_custom) */
} HgymEnvState;

/* Outputs of one env step = the 5-tuple of VecEnv.step (algo/vec_env.py:50-51) plus the extras tensors. */
typedef struct HgymEnvOut {
    float* obs;                /* (N, frame_stack*47) row-major, clipped */
    float* priv_obs;           /* (N, c_frame_stack*73) row-major, clipped */
    float* rew;                /* (N,) */
    uint8_t* reset;            /* (N,) bool */
    uint8_t* time_out;         /* (N,) bool: this step's time_out_buf */
    uint8_t* extras_time_outs; /* (N,) bool: extras["time_outs"], refreshed only on steps with >=1 reset
                                  (legged_robot.py:173-174,209-210) */
    float* extras_episode;     /* (22,) extras["episode"]["rew_<term>"], same staleness rule */
    /* Optional transition sink (t_rewards == NULL: off).  When set, the step finaliser also does what
     * PPO.process_env_step + RolloutStorage.add_transitions do for the scalar columns (ppo.py:103-113,
     * rollout_storage.py:91-94; same arithmetic as hgym_store_step) from this step's rew / reset and the
     * just-refreshed extras["time_outs"], and bumps *t_step: one launch less per vec-step. */
    const float* t_values;     /* (N,) critic values of the transition being stored */
    float* t_rewards;          /* (N,) storage.rewards[step]  = rew + t_gamma * (values * extras_time_outs) */
    uint8_t* t_dones;          /* (N,) storage.dones[step]    = reset */
    int64_t* t_step;           /* device scalar += 1 per step (the policy's sampling-step counter), or NULL */
    float t_gamma;
    /* 1: the env-step entry points do NOT launch the step finaliser; the caller runs it before the next env step, either as
     * one extra workgroup of the next policy launch (hgym_policy_act_fin) or on its own (hgym_env_finalize).  *t_step is
     * then bumped by the env kernel itself.  Until the finaliser has run, the extras_* outputs, the transition-sink slots
     * and the step counters are those of the previous step. */
    int32_t defer_finalize;
    /* Optional logging sink (log_stats == NULL: off): the step finaliser also keeps OnPolicyRunner.learn's per-step book-keeping
     * (algo/ppo/on_policy_runner.py:143-156) on the device, so that a logging run needs no host work between vec-steps:
     *   log_cur   (2, N) fp32   cur_reward_sum | cur_episode_length of every env (+= rew, += 1; zeroed when the env is done)
     *   log_stats HGYM_LOG_STATS floats:
     *     [0, 22)    sum over the steps since the caller last cleared it of extras["episode"][k] (what ep_infos.append collects)
     *     [22]       number of those steps
     *     [24], [25] head / fill count of the two rings below
     *     [32, 132)  returns of the last 100 finished episodes (rewbuffer, a deque(maxlen=100)), in env order within a step
     *     [132, 232) their lengths (lenbuffer)
     * The caller zero-fills both once and clears log_stats[0, 23) after reading it. */
    float* log_cur;
    float* log_stats;
    float* extras_custom;      /* (K,) extras["episode"]["rew_<name>"] of the user-defined terms, same staleness rule as extras_episode */
    /* hgym_rollout_step only (NULL / 0 everywhere else).  obs_ahead: rows of the observation AFTER `obs` ((N, frame_stack*47): the
     * rollout storage's slot after next).  The launch then also writes their frames 0 .. frame_stack-2 -- the frame_stack-2 older
     * frames it knows when it starts (copied by the tile's critic workgroup behind its tile) and this step's own frame -- so that
     * the NEXT launch, called with obs = this pointer and obs_older_ready = 1, does not copy them on its critical path: between the
     * first and second layer of the actor tile that copy costs 4.4 us of a 42 us launch.  The rows come out bit-identical.
     * For an env that resets in THIS step the copied frames are pre-reset history: the next call (obs_older_ready = 1, prev_out = this
     * call's out -- both already required) zeroes them before anybody reads the rows; a next call WITHOUT obs_older_ready writes all
     * older frames itself.  Do not read obs_ahead / priv_ahead rows between the two calls. */
    float* obs_ahead;
    float* priv_ahead;         /* the same for priv_obs ((N, c_frame_stack*73)); both or neither */
    int32_t obs_older_ready;   /* 1: frames 0 .. frame_stack-2 of `obs` (and 0 .. c_frame_stack-2 of `priv_obs`) were written by the
                                  previous launch (as its obs_ahead / priv_ahead) */
    /* hgym_rollout_step only, header v6 (NULL / 0 everywhere else): the actor's FIRST LAYER carried across launches.  The next row is
     * this row shifted by one frame plus the frame this step produces, so up to 20 of the first layer's 24 k-steps (columns [0, 640)
     * of the next row; 12 are taken by default) can be formed while this launch runs: its critic workgroups, idle for the second half of the launch, do that and
     * leave the fp32 partial pre-activations in l0_ahead ((num_envs, 512) floats, caller-owned, 16-byte aligned); the NEXT launch,
     * called with l0_ready = that buffer (and obs_older_ready = 1), starts its actor tile from them and reads only the remaining
     * columns of its rows.  Same fragments, same k order, same accumulator chain: bit-identical outputs.  Partial sums of rows whose
     * env was reset in between are dropped (their older frames are zero).  obs_bf16_ahead (optional, with l0_ahead): the bf16 shadow
     * rows of the NEXT step's observation (HgymObsShadow.obs of the next call): the columns formed ahead are written by this launch,
     * the next launch writes the rest and zeroes the former in reset rows. */
    float* l0_ahead;
    const float* l0_ready;
    void* obs_bf16_ahead;
    int64_t ld_obs_bf16_ahead;
    /* header v7, transition sink with DEFERRED values (t_rewards set, t_values NULL): nothing inside a rollout consumes V(s_t) except the
     * time-out bootstrap r += gamma * V * time_outs (ppo.py:107-108) and the GAE that follows the rollout, and the weights do not change
     * while it is collected -- so the critic may run ONCE over all stored rows afterwards (hgym_critic_values) instead of once per
     * step.  The finaliser then stores the RAW reward in t_rewards and, here, the flags the bootstrap would have used (the stale-by-design
     * extras["time_outs"] of this step, (N,) uint8); hgym_gae_bootstrap applies the same fp32 expression when it loads r_t. */
    uint8_t* t_time_outs;
} HgymEnvOut;
#define HGYM_LOG_STATS 256

/* Optional externally supplied random draws (parity mode), row-major (N,k) tables indexed by env id.
 * A NULL member means "draw it from the internal Philox4x32-10 stream keyed by (seed, step, env, slot)". */
typedef struct HgymEnvNoise {
    const float* u_delay;      /* (N,)    U[0,1)  humanoid_env.py:194 */
    const float* z_act;        /* (N,12)  N(0,1)  humanoid_env.py:196 */
    const float* u_cmd;        /* (N,6)   U[0,1)  [0:3] callback resample, [3:6] reset resample (legged_robot.py:328-331) */
    const float* u_dof;        /* (N,12)  U[0,1)  legged_robot.py:367 */
    const float* u_push;       /* (N,5)   U[0,1)  humanoid_env.py:88-93 */
    const float* z_obs;        /* (N,47)  N(0,1)  humanoid_env.py:251 */
    const float* u_xy;         /* (N,2)   U[0,1)  legged_robot.py:385 (custom origins) */
    const int64_t* r_level;    /* (N,)    integers in [0, terrain_rows)  legged_robot.py:418 (terrain curriculum) */
} HgymEnvNoise;

/* XBotLFreeEnv.__init__ tail (humanoid_env.py:78-81): state defaults, reset_idx(all), compute_observations. */
int32_t hgym_env_prime(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st,
                       const HgymEnvOut* out, const HgymEnvNoise* noise, void* stream);

/* LeggedRobot.reset's reset_idx(all) (legged_robot.py:112-114): the reset branch for every env, history
 * cleared, no observation pushed (the caller follows with a zero-action step, :115-116). */
int32_t hgym_env_reset_all(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st,
                           const HgymEnvOut* out, const HgymEnvNoise* noise, void* stream);

/* XBotLFreeEnv.step head + LeggedRobot.step clip (humanoid_env.py:189-197, legged_robot.py:90-91):
 * st->actions <- clip(blend(clip(actions_in), st->actions) * (1 + noise)).  actions_in (N,12) row-major; read-only
 * unless cfg->use_ref_actions, in which case the reference pose is added to it in place first (as the reference does to
 * the caller's tensor). */
int32_t hgym_pre_physics(const HgymEnvConfig* cfg, const HgymEnvState* st, float* actions_in,
                         const HgymEnvNoise* noise, void* stream);

/* LeggedRobot._compute_torques (legged_robot.py:340-356) on the current dof state -> st->torques. */
int32_t hgym_pd_torques(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, void* stream);

/* Stands where gym.simulate x decimation is (legged_robot.py:94-101,124-126): the synthetic physics of
 * SURVEY.md §8d (benchmark backend).  Runs `decimation` PD + semi-implicit Euler substeps and draws the
 * root / contact / rigid-body tensors from Philox. */
int32_t hgym_synth_physics(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, void* stream);

/* LeggedRobot.post_physics_step + obs clip (legged_robot.py:119-151,105-108) with XBotLFreeEnv's
 * callbacks, 22 reward terms, mask-driven reset_idx and observation stacking. */
int32_t hgym_post_physics(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st,
                          const HgymEnvOut* out, const HgymEnvNoise* noise, void* stream);

/* Fast path: pre_physics + synth_physics + post_physics in ONE launch, then the step finaliser. */
int32_t hgym_env_step_synth(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st,
                            const HgymEnvOut* out, float* actions_in, void* stream);

/* One env step in TWO launches, for tasks with user-defined reward terms (HgymEnvConfig.num_custom_rewards > 0):
 *   hgym_env_step_begin  action processing (+ synthetic physics when actions_in != NULL; NULL: an external simulator has written the
 *                        sim tensors, as for hgym_post_physics), episode length + 1, derived state, command resampling, pushes,
 *                        termination flags -> out->reset / out->time_out -- legged_robot.py:128-137,156-161: the state
 *                        compute_reward (:142) starts from;
 *   [the caller evaluates its `_reward_<name>` terms on that state and writes term * scale * dt into st->custom_rew]
 *   hgym_env_step_end    compute_reward with the caller's terms merged in at their alphabetical positions, reset_idx,
 *                        observations, tail, finaliser (:142-151).
 * With num_custom_rewards == 0 the pair computes exactly what hgym_env_step_synth / hgym_post_physics do in one launch. */
int32_t hgym_env_step_begin(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                            const HgymEnvNoise* noise, float* actions_in, void* stream);
int32_t hgym_env_step_end(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                          const HgymEnvNoise* noise, void* stream);

/* LeggedRobot._get_heights (legged_robot.py:761-795) for the poses the last env step left in st->height_pose ->
 * st->measured_heights.  The env-step entry points call it themselves when cfg->num_height_points > 0. */
int32_t hgym_measure_heights(const HgymEnvConfig* cfg, const HgymEnvState* st, void* stream);

/* The step finaliser of the last env step on its own (flushes HgymEnvOut.defer_finalize; a no-op to call twice it is NOT:
 * the counters advance each time). */
int32_t hgym_env_finalize(const HgymEnvConfig* cfg, const HgymEnvState* st, const HgymEnvOut* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Rollout storage side (algo/ppo/rollout_storage.py, algo/ppo/ppo.py:103-117)
 * ---------------------------------------------------------------------------------------------- */

/* PPO.process_env_step + RolloutStorage.add_transitions for the scalar columns (ppo.py:103-113,
 * rollout_storage.py:87-100): rewards_slot = rew + gamma*values*time_outs ; dones_slot = dones. */
int32_t hgym_store_step(int32_t n, const float* rew, const float* values, const uint8_t* time_outs,
                        const uint8_t* dones, float gamma, float* rewards_slot, uint8_t* dones_slot, void* stream);

/* The minibatch permutation of RolloutStorage.mini_batch_generator (rollout_storage.py:149, torch.randperm(T*N)): out[i], i in
 * [0, n), is a bijection of [0, n) keyed by (seed, draw) -- a 6-round Feistel network, cycle-walked; one launch, no scratch
 * (torch.randperm on the device: a sort, 125 us per iteration at the XBot-L batch). */
int32_t hgym_randperm(int64_t n, uint64_t seed, uint64_t draw, int64_t* out, void* stream);
/* header v9: the same permutation with the draw number read from DEVICE memory (*draw, one int64 the caller advances with a device
 * operation): the launch's arguments do not change from one learning iteration to the next, so an update captured into a HIP graph
 * (OnPolicyRunner: the whole iteration is two graph replays) draws a new permutation at every replay. */
int32_t hgym_randperm_dev(int64_t n, uint64_t seed, const int64_t* draw, int64_t* out, void* stream);

/* RolloutStorage.compute_returns (rollout_storage.py:122-136): GAE(lambda) as a wavefront suffix scan.
 * rewards/values/returns/advantages are (T,N) time-major fp32, dones (T,N) uint8, last_values (N,).
 * stats: HGYM_GAE_STATS_DOUBLES(n) doubles on the device, ZERO-FILLED ONCE by the caller: [0] sum adv, [1] sum adv^2, [2] count are
 * WRITTEN by the call; [3] is the arrival counter of the call's workgroups (left zero again), [4 ..) their partial sums, which the
 * last workgroup to arrive adds up in workgroup order -- the statistics have the same bits in every run and on every rank (header v8;
 * until v7 three doubles accumulated with fp64 atomics in arrival order).  One call at a time per stats buffer. */
#define HGYM_GAE_STATS_DOUBLES(n) (4 + 2 * (((n) + 15) / 16))
int32_t hgym_gae(int32_t T, int32_t n, const float* rewards, const float* values, const uint8_t* dones,
                 const float* last_values, float gamma, float lam, float* returns, float* advantages,
                 double* stats, void* stream);

/* advantages = (adv - mean) / (std_unbiased + 1e-8) from stats (possibly all-reduced by the caller). */
int32_t hgym_adv_normalize(int64_t count, float* advantages, const double* stats, void* stream);
/* hgym_gae for a rollout collected with deferred values (HgymEnvOut.t_time_outs): rewards (T, n) holds the RAW rewards and is
 * overwritten with r_t + gamma * (V_t * time_outs_t) -- PPO.process_env_step's bootstrap (ppo.py:107-108), the three fp32 roundings of
 * hgym_store_step -- before the scan uses it, so that afterwards every storage column is what the per-step path leaves. */
int32_t hgym_gae_bootstrap(int32_t T, int32_t n, float* rewards, const float* values, const uint8_t* dones, const uint8_t* time_outs,
                           const float* last_values, float gamma, float lam, float* returns, float* advantages, double* stats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Actor / critic (algo/ppo/actor_critic.py) and PPO update (algo/ppo/ppo.py:119-184)
 * ---------------------------------------------------------------------------------------------- */
enum { HGYM_F32 = 0, HGYM_BF16 = 1 };

typedef struct HgymNetConfig {
    int32_t num_obs, num_priv, num_actions;
    int32_t actor_layers, critic_layers;                 /* number of Linear layers (4, 4) */
    int32_t actor_dims[HGYM_MAX_LAYERS + 1];             /* 705,512,256,128,12 */
    int32_t critic_dims[HGYM_MAX_LAYERS + 1];            /* 219,768,256,128,1  */
    int32_t precision;                                   /* HGYM_F32 (parity) or HGYM_BF16 (MFMA fast path) */
    int32_t max_batch;                                   /* largest M any call will use */
    /* Optional auxiliary head trained jointly with PPO (BASELINE configs[4], SURVEY.md 8f item 4: the denoising world-model
     * head -- NO reference code exists for it; design: DESIGN.md section 9).  An MLP obs -> aux_dims[1..] (ELU between) that
     * regresses columns [aux_target_offset, aux_target_offset + aux_dims[aux_layers]) of the privileged observation row
     * with a mean-squared error weighted by HgymPPOConfig.aux_coef.  aux_layers = 0: absent.  Its parameters follow the
     * critic's in the flat vector ("denoiser.{0,2,..}.{weight,bias}"); hgym_mlp_forward(which = 2) evaluates it. */
    int32_t aux_layers;
    int32_t aux_dims[HGYM_MAX_LAYERS + 1];               /* aux_dims[0] = num_obs */
    int32_t aux_target_offset;
} HgymNetConfig;

typedef struct HgymPPOConfig {
    float clip_param, value_loss_coef, entropy_coef, max_grad_norm, desired_kl;
    float beta1, beta2, adam_eps;
    double lr_min, lr_max;          /* 1e-5, 1e-2 (ppo.py:142-145) */
    int32_t adaptive_lr;            /* schedule == 'adaptive' */
    int32_t world_size;             /* >1: between hgym_ppo_grad and hgym_ppo_apply the caller all-reduces (SUM) the P+1 floats of
                                       net->grads across ranks; apply forms the means (gradient and KL) itself */
    float aux_coef;                 /* weight of the auxiliary head's MSE in the total loss (0 with aux_layers = 0) */
    int32_t grad_norm_ready;        /* 1: net->grads is exactly what the preceding hgym_ppo_grad left (one rank, nothing touched it),
                                       so its squared norm is already in opt_state[9] and apply skips its own pass over the
                                       gradient; 0 (or world_size > 1): apply computes the norm itself.  The flag is a PROMISE that every
                                       hgym_ppo_grad is followed by exactly one hgym_ppo_apply with the same configuration: on the fused
                                       bf16 path hgym_ppo_grad then also takes the adaptive-KL learning-rate decision and advances Adam's
                                       step count (the work of apply's prologue, done beside the weight-gradient launch).  A marker in
                                       opt_state[13] keeps the two honest: a second hgym_ppo_grad before the apply does not advance the step
                                       again, and an apply under another configuration does not repeat a prologue already taken.  (Not
                                       covered: a gradient call WITHOUT the flag followed by an apply WITH it -- no prologue runs.) */
} HgymPPOConfig;

/* Sizes (bytes) of the caller-allocated blocks, as functions of the configuration. */
int64_t hgym_net_param_count(const HgymNetConfig* net);          /* 926105 for XBot-L */
int64_t hgym_net_workspace_bytes(const HgymNetConfig* net);

/* Master parameters are ONE flat fp32 array in state_dict order
 * (std, actor.{0,2,4,6}.{weight,bias}, critic.{0,2,4,6}.{weight,bias}; SURVEY.md §5 checkpoint row);
 * grads/adam_m/adam_v have the same layout.  opt_state: 16 doubles on the device
 * [0] learning rate (python-double semantics of ppo.py:142-148)   [1] Adam step count
 * [2] sum of minibatch mean KL  [3] sum of surrogate losses  [4] sum of value losses  [5] sum of mean entropies
 * [6] gradient norm of the last step (before clipping)  [7] minibatches accumulated in [2..5]
 * [8] mean KL of the last minibatch (average it across ranks before hgym_ppo_apply when world_size > 1)
 * [9] internal  [10] sum of the auxiliary head's minibatch MSE losses  [11] Adam step size lr / (1 - beta1^t) and
 * [12] sqrt(1 - beta2^t) of the current step (as floats; written by hgym_ppo_apply)  [13..15] internal (the step whose beta^t lie in [14..15] / the
 * "prologue done, not applied" marker: Adam's betas must not change between a gradient call and the apply that follows it).
 * workspace: hgym_net_workspace_bytes() bytes, 256-byte aligned, ZERO-FILLED once by the caller before first use
 * (padding rows/columns of the operand buffers rely on it). */
typedef struct HgymNet {
    float* params;
    float* grads;          /* hgym_net_param_count() + 1 floats: the flat gradient, then ONE slot carrying the minibatch mean KL */
    float* adam_m;
    float* adam_v;
    double* opt_state;     /* [16] */
    void* workspace;       /* hgym_net_workspace_bytes() bytes, 256-byte aligned */
} HgymNet;

/* re-derive the compute-precision operand copies (padded W and W^T) from the fp32 master parameters;
 * call after loading a checkpoint.  The Adam kernel keeps them current afterwards. */
int32_t hgym_net_sync_shadow(const HgymNetConfig* cfg, const HgymNet* net, void* stream);

/* ActorCritic.act_inference / evaluate (actor_critic.py:122-128): which = 0 actor, 1 critic.
 * x (M, in) row-major fp32 with leading dimension ldx; y (M, out) row-major fp32. */
int32_t hgym_mlp_forward(const HgymNetConfig* cfg, const HgymNet* net, int32_t which, int32_t M,
                         const float* x, int64_t ldx, float* y, void* stream);

/* bf16 shadow of the observation rows (optional; the update's fast path).  The update reads every stored observation row twice per
 * epoch -- the forward gathers it, the first layer's weight-gradient product gathers it again -- and only ever in bf16.  A policy
 * launch has each row in hand as bf16 anyway (it converts the fp32 row for its first layer), so it can leave that copy behind:
 * `shadow->obs` / `shadow->priv` receive, row m, the bf16 of row m of `obs` / `priv` (row-major, leading dimensions ld_obs /
 * ld_priv in elements = hgym_net_shadow_ld(cfg, 0 / 1): the input width padded to 128, pad columns written as zero).  Handed
 * back through HgymBatch.obs_bf16 / priv_bf16, the update gathers 2 bytes per element instead of 4 and writes no operand copy
 * of its own.  NULL: no shadow is written.  Only the fused bf16 path writes / reads it (hgym_net_shadow_ld returns 0 otherwise). */
typedef struct HgymObsShadow {
    void* obs;       /* (M, ld_obs) bf16 */
    int64_t ld_obs;
    void* priv;      /* (M, ld_priv) bf16 */
    int64_t ld_priv;
} HgymObsShadow;
/* leading dimension (elements) of the bf16 shadow of net `which`'s input rows (0 actor: obs, 1 critic: privileged obs), or 0 when
 * this configuration does not take the fused bf16 path */
int64_t hgym_net_shadow_ld(const HgymNetConfig* cfg, int32_t which);
/* The critic over M stored privileged-observation rows in one call (ActorCritic.evaluate, actor_critic.py:127-129, on every slot of a
 * finished rollout): values (M,) fp32; shadow->priv (optional) receives the bf16 of the rows, as the per-step policy launches leave it.
 * M may exceed HgymNetConfig.max_batch (the rows are walked in pieces). */
int32_t hgym_critic_values(const HgymNetConfig* cfg, const HgymNet* net, int64_t M, const float* priv, float* values,
                           const HgymObsShadow* shadow, void* stream);

/* PPO.act (ppo.py:91-101): mu = actor(obs); sigma = std; a = mu + sigma*z; V = critic(priv);
 * logp = sum log N(a; mu, sigma).  z (M,12) standard normal draws or NULL -> Philox(seed, *step_counter).
 * Outputs row-major: actions/mu/sigma (M,12), logp (M,), values (M,). */
int32_t hgym_policy_act(const HgymNetConfig* cfg, const HgymNet* net, int32_t M, const float* obs,
                        const float* priv, const float* z, uint64_t seed, const int64_t* step_counter,
                        float* actions, float* mu, float* sigma, float* logp, float* values,
                        const HgymObsShadow* shadow, void* stream);

/* hgym_policy_act + the postponed step finaliser of the PREVIOUS env step (env_cfg / env_st / env_out as that step got them,
 * env_out->defer_finalize = 1) as one extra workgroup of the same launch: one launch less per vec-step. */
int32_t hgym_policy_act_fin(const HgymNetConfig* cfg, const HgymNet* net, int32_t M, const float* obs,
                            const float* priv, const float* z, uint64_t seed, const int64_t* step_counter,
                            float* actions, float* mu, float* sigma, float* logp, float* values,
                            const HgymEnvConfig* env_cfg, const HgymEnvState* env_st, const HgymEnvOut* env_out,
                            const HgymObsShadow* shadow, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused rollout step: PPO.act + XBotLFreeEnv.step (synthetic-physics backend) + the previous step's finaliser in ONE launch
 * (algo/ppo/on_policy_runner.py:129-141 is the loop body this stands for: act -> env.step -> process_env_step).
 * Workgroup b computes the actor tile of envs [32 b, 32 b + 32), samples their actions and runs their env step right behind it;
 * a second grid row holds the critic tiles, a third the finaliser of the PREVIOUS step (HgymEnvOut.defer_finalize semantics:
 * the transition sink of `prev_out`, extras, counters).  Results are those of hgym_policy_act_fin followed by
 * hgym_env_step_synth: same kernels' source, same arithmetic.
 *
 * Calling sequence for a rollout of T steps, parity(t) alternating 0 / 1:
 *     hgym_rollout_begin(st, step_counter, scratch, parity(0))
 *     for t in 0..T-1:  hgym_rollout_step(..., out[t], t ? out[t-1] : NULL, obs[t], priv[t], ..., scratch, parity(t))
 *     hgym_rollout_end(env_cfg, st, out[T-1], scratch, parity(T-1))
 * where out[t] has obs / priv_obs = where the observations of step t+1 go, the transition sink of step t (t_values = the
 * `values` buffer of the same call, t_rewards / t_dones, t_step = step_counter), defer_finalize = 1, and -- because the
 * finaliser of step t-1 runs concurrently with the env phase of step t -- rew / reset / time_out buffers DISTINCT from
 * out[t-1]'s (two sets, alternating).  Between begin and end no other env entry point may run on this state.
 * scratch: HGYM_ROLLOUT_SCRATCH_BYTES(num_envs) bytes owned by the caller, 16-byte aligned, zero-filled once (ping-pong step
 * counters, per-parity reset count and episode-sum accumulators; behind that header the env step's draw tables, which the
 * critic workgroups of step t compute for step t + 1).  Supported: the XBot-L default options (none of the generic ones, no use_ref_actions),
 * 15 / 3 history, contiguous [136][N] state, SoA sim tensors, N a multiple of 32, the bf16 fused net path;
 * HGYM_E_UNSUPPORTED otherwise (callers fall back to hgym_policy_act_fin + hgym_env_step_synth).
 * values = NULL (header v7, deferred values): no critic tiles -- the grid is the N / 32 actor + env workgroups and the finaliser, every
 * workgroup draws its own step's random numbers and copies its own history rows (the critic workgroups' side jobs), and the
 * sinks of out / prev_out must be of the deferred kind (t_values NULL, t_time_outs set); obs_ahead / obs_older_ready / l0_* stay
 * NULL / 0; `priv` is not read and the shadow's priv member is not written (hgym_critic_values does both after the rollout).  This is
 * the form for more than half a chip of envs (N / 32 > CUs / 2: 8192 envs per MI355X), where the critic's tiles would need the
 * compute units of a second round.
 * ---------------------------------------------------------------------------------------------- */
#define HGYM_ROLLOUT_SCRATCH_HEADER_BYTES 512
#define HGYM_ROLLOUT_DRAW_BYTES_PER_ENV 1000      /* two parities x 125 floats */
#define HGYM_ROLLOUT_SCRATCH_BYTES(num_envs) (HGYM_ROLLOUT_SCRATCH_HEADER_BYTES + (size_t)HGYM_ROLLOUT_DRAW_BYTES_PER_ENV * (size_t)(num_envs))
int32_t hgym_rollout_begin(const HgymEnvState* st, const int64_t* step_counter, void* scratch, int32_t parity, void* stream);
int32_t hgym_rollout_step(const HgymNetConfig* cfg, const HgymNet* net, const HgymEnvConfig* env_cfg, const HgymSimTensors* sim,
                          const HgymEnvState* st, const HgymEnvOut* out, const HgymEnvOut* prev_out, const float* obs,
                          const float* priv, uint64_t seed, float* actions, float* mu, float* sigma, float* logp, float* values,
                          void* scratch, int32_t parity, const HgymObsShadow* shadow, void* stream);
int32_t hgym_rollout_end(const HgymEnvConfig* env_cfg, const HgymEnvState* st, const HgymEnvOut* last_out, void* scratch,
                         int32_t parity, void* stream);

/* One minibatch of PPO.update up to and including backward (ppo.py:128-171), device side only:
 * gathers rows `idx[0..B)` (indices into the flattened (T*N) storage, rollout_storage.py:151-182) of the
 * nine storage tensors, forward, KL -> learning-rate adaptation (written to opt_state[0]), clipped
 * surrogate + clipped value loss + entropy bonus, hand-written backward -> net->grads (un-clipped). */
typedef struct HgymBatch {
    const float* obs;        /* (T*N, num_obs)   */
    const float* priv;       /* (T*N, num_priv)  */
    const float* actions;    /* (T*N, 12) */
    const float* values;     /* (T*N,)    */
    const float* advantages; /* (T*N,)    */
    const float* returns;    /* (T*N,)    */
    const float* logp;       /* (T*N,)    */
    const float* mu;         /* (T*N, 12) */
    const float* sigma;      /* (T*N, 12) */
    const int64_t* idx;      /* (B,) */
    int32_t B;
    /* optional bf16 shadows of obs / priv, same rows (HgymObsShadow: written by the policy launches that read those rows), leading
     * dimensions hgym_net_shadow_ld(cfg, 0 / 1); both or neither.  NULL: the fp32 rows are gathered and converted (and a bf16
     * operand copy of the gathered rows is kept for the weight-gradient kernel). */
    const void* obs_bf16;
    const void* priv_bf16;
    int64_t num_rows;        /* rows of the storage tensors (T*N), required (> 0) with the shadows: the weight-gradient kernel addresses a
                                shadow with 32-bit byte offsets, so num_rows * hgym_net_shadow_ld(cfg, 0) * 2 must stay below 4 GiB (2.79 M
                                rows for XBot-L) -- beyond that the fp32 rows are used */
} HgymBatch;

int32_t hgym_ppo_grad(const HgymNetConfig* cfg, const HgymPPOConfig* ppo, const HgymNet* net,
                      const HgymBatch* batch, void* stream);

/* hgym_ppo_grad in two halves (one process per GPU; the reference is single-process, its `--horovod` flag is dead:
 * utils/helpers.py:207-212), for a caller that wants to start exchanging the first gradient bucket a few microseconds early:
 *   part 0: forward, loss, dZ chain and ALL weight-gradient products (of every net, the auxiliary head included), then the slab
 *           sums of [std | actor]: on return (stream order) grads[0 .. hgym_net_param_offset(cfg, 1)) -- 527 256 of 926 106
 *           floats -- are final;
 *   part 1: the slab sums of the rest: grads[hgym_net_param_offset(cfg, 1) .. P] -- critic | auxiliary head | the KL slot.
 * (Until round 3 part 1 also launched the critic's weight-gradient products on their own, so that the first bucket travelled
 * under them: two launches of 144 and 112 workgroups on 256 CUs took 2 x 142 us against 178 us for the one launch, more than the
 * exchange they hid.  PPO.update now calls hgym_ppo_grad and exchanges ONE bucket.)
 * part 0 followed by part 1 leaves net->grads exactly as hgym_ppo_grad does (bit-identical), and opt_state[9] (the squared norm
 * hgym_ppo_apply may reuse with grad_norm_ready) complete: part 0 zeroes it and adds its bucket's share, part 1 adds the rest.
 * Between part 0 and part 1 it is partial; with world_size > 1 apply recomputes the norm of the rank mean regardless. */
int32_t hgym_ppo_grad_part(const HgymNetConfig* cfg, const HgymPPOConfig* ppo, const HgymNet* net,
                           const HgymBatch* batch, int32_t part, void* stream);
/* offset (floats) in the flat parameter / gradient vector of the first parameter of net `which` (0 actor, 1 critic, 2 auxiliary
 * head; which == number of nets: the parameter count); -1 on a bad argument */
int64_t hgym_net_param_offset(const HgymNetConfig* cfg, int32_t which);

/* clip_grad_norm_ + Adam.step (ppo.py:173-174) on net->grads (the rank-SUM when world_size > 1: divided by
 * world_size here, as is the KL in grads[P] before the learning-rate decision), refreshes the
 * compute-precision shadows, bumps opt_state. */
int32_t hgym_ppo_apply(const HgymNetConfig* cfg, const HgymPPOConfig* ppo, const HgymNet* net, void* stream);

/* ---- direct gradient exchange over peer mappings (header v6 / v7; the alternative to the RCCL all-reduce of the data-parallel update
 * that HGYM_COMM=auto probes and picks at start-up:
 * reference has nothing here, /root/reference/humanoid/utils/helpers.py:207-212 is a dead flag).  One process per GPU.  Every rank
 * allocates ONE fine-grained buffer with hgym_comm_alloc, exports its handle (hgym_comm_ipc_export, 64 bytes, exchanged by the
 * caller over whatever it has -- torch.distributed's all_gather_object here), opens the peers' (hgym_comm_ipc_open) and fills an
 * HgymComm: data[q] = rank q's payload (count floats, count % 4 == 0; data[rank] is the caller's own -- the flat [gradient | KL] vector
 * lives there, HgymNet.grads points into it), flags[q] = rank q's flag block (HGYM_COMM_FLAG_WORDS zero-filled uint32), status = 16
 * int64 of the caller's own buffer.  hgym_comm_allreduce(seq = 1, 2, 3, ... -- the same sequence on every rank; or 0, below) enqueues ONE kernel that
 * leaves the rank-ordered fp32 SUM in every rank's data (bit-identical on all ranks): arrival flags, each rank sums its 1 / world shard
 * from all buffers and stores the result into all buffers, completion flags.  Waits are bounded (wait_ticks of the 100 MHz wall clock,
 * 0 = 15 s): on expiry status[0] = 1 (sticky until the caller clears it), status[1] = the seq of the call, that call's payload is
 * garbage and the communicator stays usable for later calls.  hgym_comm_status synchronises `stream` and copies the 16 status words
 * to the host: call it where the host synchronises anyway.  status[8 .. 10] = 100 MHz timestamps of the last call: start, every
 * rank arrived, every shard delivered. */
#define HGYM_COMM_MAX_RANKS 8
#define HGYM_IPC_HANDLE_BYTES 64
#define HGYM_COMM_FLAG_WORDS 32
typedef struct HgymComm {
    int32_t world, rank;
    float* data[HGYM_COMM_MAX_RANKS];
    uint32_t* flags[HGYM_COMM_MAX_RANKS];
    int64_t count;
    int64_t* status;
    int64_t wait_ticks;     /* v7: bound of every wait inside hgym_comm_allreduce, 100 MHz ticks; 0 = the default (15 s) */
    double* aux[HGYM_COMM_MAX_RANKS];   /* v9: aux[q] = rank q's block of HGYM_COMM_AUX_DOUBLES zero-filled doubles in the same fine-grained
                                         * allocation (hgym_comm_sum64's slots); NULL everywhere if hgym_comm_sum64 is not used */
} HgymComm;
int32_t hgym_comm_alloc(int64_t bytes, void** dev_ptr);
int32_t hgym_comm_free(void* dev_ptr);
int32_t hgym_comm_ipc_export(void* dev_ptr, void* handle_out);
int32_t hgym_comm_ipc_open(const void* handle, void** dev_ptr);
int32_t hgym_comm_ipc_close(void* dev_ptr);
int32_t hgym_comm_allreduce(const HgymComm* comm, uint32_t seq, void* stream);
/* header v9 -- what lets a data-parallel update be captured into a HIP graph (no launch argument changes from call to call, no
 * torch.distributed call inside):
 *   hgym_comm_allreduce(seq = 0): the call number is kept on the device -- status[2] = number of the last call, read by the kernel on
 *   entry and advanced by it; all ranks must use the same form for the same call (a communicator is driven either with explicit numbers
 *   or with 0 throughout);
 *   hgym_comm_sum64: in-place SUM over the ranks of n <= 3 doubles at `vals` (device memory of the caller) -- the per-iteration
 *   (sum adv, sum adv^2, count) of RolloutStorage.compute_returns' normalisation (rollout_storage.py:132-136 over the global batch) --
 *   through aux[]: one wavefront, tagged slots, summed in rank order (bit-identical on all ranks); its call number is status[3];
 *   bounded waits as above (status[0] = 1 on expiry). */
#define HGYM_COMM_AUX_DOUBLES (2 * HGYM_COMM_MAX_RANKS * 4)
int32_t hgym_comm_sum64(const HgymComm* comm, double* vals, int32_t n, void* stream);
int32_t hgym_comm_status(const HgymComm* comm, int64_t* host16, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py's roofline leg).  PROCESS-GLOBAL state, unlike the rest of this interface: one event list and one
 * phase buffer per process, not per device or per stream, and not thread-safe -- enable them from one thread, for one device at
 * a time.  (Everything else keeps no state between calls except the per-(kernel, device) record of how much dynamic LDS has been
 * reserved, which is mutex-guarded: one process may drive several devices and streams.)  When enabled, every launch of a profiled kernel class is
 * bracketed by a pair of HIP events recorded on the launch stream; the summary synchronises those events and
 * returns launches, summed duration and summed ALGORITHMIC work (flops for the GEMM class, bytes for the
 * HBM-bound classes; DESIGN.md states the per-unit figures).  Off by default; not capturable in a hipGraph.
 * ---------------------------------------------------------------------------------------------- */
enum {
    HGYM_PROF_GEMM = 0,      /* generic MFMA GEMM launches (fp32 parity path / unsupported layer shapes) */
    HGYM_PROF_ENV_STEP = 1, HGYM_PROF_GAE = 2, HGYM_PROF_LOSS = 3,
    HGYM_PROF_MLP_FWD = 4,   /* the update's fused forward + loss + dZ chain (64-row tiles, writes the activations and their gradients) */
    HGYM_PROF_MLP_BWD = 5,   /* (unused since the dZ chain runs inside the class-4 kernel; the value is kept) */
    HGYM_PROF_DW = 6,        /* all weight-gradient products */
    HGYM_PROF_REDUCE = 7,    /* split-K slab reduction */
    HGYM_PROF_APPLY = 8,     /* grad-norm + clip + Adam */
    HGYM_PROF_POLICY = 9,    /* fused forward, rollout / inference (32-row tiles, sampling epilogue) */
    HGYM_PROF_ROLLOUT = 10,  /* fused rollout step (policy + env step + previous finaliser); work = algorithmic HBM bytes */
    HGYM_PROF_COMM = 11,     /* hgym_comm_allreduce (the direct gradient exchange); work = bytes this rank moves over its links */
    HGYM_PROF_CLASSES = 12
};
int32_t hgym_prof_enable(int32_t on);  /* 1: start collecting (clears previous events), 0: stop */
int32_t hgym_prof_summary(int32_t cls, int64_t* launches, double* total_ms, double* work);
/* Kernel-internal phase clock for tuning the fused MLP kernels: while a caller-owned device buffer of `slots` int64 is
 * set, thread 0 of workgroup b of every fused MLP launch (forward, update, rollout step) writes the 100 MHz wall clock at up to 8 phase
 * boundaries into dev[b*8 + phase] (launches with more than slots/8 workgroups are not instrumented).  NULL: off. */
int32_t hgym_prof_phase_buffer(void* dev, int64_t slots);

#ifdef __cplusplus
}
#endif
#endif /* HGYM_H_ */
