// hgym_env.hip -- env-side kernels of the XBot-L hot path for gfx950.
//
// One launch per vec-step.  A workgroup owns a contiguous slice of `envs_per_block` envs:
//   phase A  one lane per env: (optionally) action processing + synthetic physics, then the whole
//            post-physics pipeline -- derived state, command resampling, termination, the 22 reward
//            terms, mask-driven reset -- on env-major SoA state (lane i <-> env i: every load/store of a
//            state component is one coalesced 256-B wavefront access).  The clean new 47/73-float
//            frames are staged in LDS.
//   phase B  all lanes: the 15x47 / 3x73 history stack is produced as ONE contiguous run of the
//            row-major output per workgroup (coalesced stores), reading the older frames from an
//            HBM ring buffer, adding observation noise to the newest frame, pushing it into the
//            ring, zeroing the history of envs that reset, clipping to +-18.
// HBM traffic per env-step is the algorithmic minimum of SURVEY.md §8d: read 14x47+2x73 old frames,
// write 15x47+3x73 stacked outputs + 47+73 ring frames, ~250 floats of sim input / env state.
// The kernel is bandwidth/latency bound; no MFMA here by design.
#include <stdlib.h>

#include "hgym_env_math.hpp"


namespace hgym {

// kStep: the instantiation for plain steps (A.mode == MODE_STEP) -- the split per-env chain is then the only one compiled in, which
// is what sets the kernel's register count (and with it how many workgroups share a CU when there are more envs than one round).
template <int H_T, int HC_T, int E_T, bool kGeneric, bool kStep = false>
__global__ __launch_bounds__(256) void env_step_kernel(const EnvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t csc0 = A.st.counters[0];
    const int64_t ring_step = A.st.counters[2];
    const int t = threadIdx.x;
    // compiled-in geometry: wavefronts 1..3 prefetch the observation history into registers here, so the whole step makes
    // ONE round trip to memory for its inputs, and store it (clipped) while wavefront 0 runs the per-env scalar chains
    constexpr bool kPrefetch = E_T > 0 && H_T > 1 && HC_T > 1;
    constexpr int NTH = 192;                              // history lanes: wavefronts 1..3
    constexpr int HP = kPrefetch ? H_T : 2, HCP = kPrefetch ? HC_T : 2, EP = kPrefetch ? E_T : 4;
    constexpr int NIO = kPrefetch ? hist_ni<HP, HGYM_OBS_FRAME, EP, NTH>() : 0;
    constexpr int NIP = kPrefetch ? hist_ni<HCP, HGYM_PRIV_FRAME, EP, NTH>() : 0;
    float hist_o[NIO > 0 ? NIO : 1][4], hist_p[NIP > 0 ? NIP : 1][4];
    const StackGeom geom = stack_geom<H_T, HC_T, E_T>(A, blockIdx.x);
    const bool stack_on = A.mode != MODE_RESET_ALL && A.phase != 1;      // a derive launch writes no observations
    if (kPrefetch && stack_on && t >= 64) {
        hist_load<HP, HGYM_OBS_FRAME, NIO>(A.st.obs_ring, geom.e0, geom.nE, (int)(ring_step % HP), t - 64, NTH, hist_o);
        hist_load<HCP, HGYM_PRIV_FRAME, NIP>(A.st.priv_ring, geom.e0, geom.nE, (int)(ring_step % HCP), t - 64, NTH, hist_p);
    }
    env_stage_in<E_T>(A, blockIdx.x, t, blockDim.x, smem);
    env_fill_draws<E_T>(A, blockIdx.x, t, blockDim.x, smem, csc0);
    env_reset_pose<E_T>(A, t, blockDim.x, smem);
    __syncthreads();
    // The XBot-L instantiation splits the per-env chain of a plain step: what is the same few instructions
    // for each of the 12 joints runs one (env, joint) pair per lane before (phase J) and after (phase F) a shorter chain, and
    // the synthetic physics' per-env remainder runs on two otherwise idle wavefronts of phase J.
    const bool split = !kGeneric && E_T > 0 && (kStep || A.mode == MODE_STEP);
    // (this kernel keeps the chain on ONE wavefront -- its other three store the observation history meanwhile; the fused rollout
    // launch runs it on four by role, hgym_rollout.hip)
    if (split) env_step_phase_j<E_T, false>(A, blockIdx.x, t, blockDim.x, smem);
    else env_step_joints<E_T>(A, blockIdx.x, t, blockDim.x, smem);
    __syncthreads();
    // wavefront 0 runs the per-env scalar chains (one lane per env); the other wavefronts meanwhile move the
    // older frames of the observation history, which depend on nothing this step computes (reset envs are fixed up in phase B) --
    // their stores are issued first
    if (t >= 64) {
        if (kPrefetch) {
            if (stack_on) {
                hist_store<HP, HGYM_OBS_FRAME, NIO>(A.out.obs, geom.e0, geom.nE, (int)(ring_step % HP), t - 64, NTH, nullptr, A.cfg.clip_obs, hist_o);
                hist_store<HCP, HGYM_PRIV_FRAME, NIP>(A.out.priv_obs, geom.e0, geom.nE, (int)(ring_step % HCP), t - 64, NTH, nullptr, A.cfg.clip_obs,
                                                      hist_p);
            }
        } else if (A.phase != 1) {
            env_step_stack_old<H_T, HC_T, E_T>(A, blockIdx.x, t - 64, blockDim.x - 64, ring_step);
        }
    }
    if (t < 64) {
        if (split) env_step_phase_a<E_T, kGeneric, true>(A, blockIdx.x, t, smem, csc0);
        else env_step_phase_a<E_T, kGeneric>(A, blockIdx.x, t, smem, csc0);
    }
    __syncthreads();
    if (split) {
        env_step_phase_f<E_T>(A, blockIdx.x, t, blockDim.x, smem);
        __syncthreads();
    }
    env_stage_out<E_T>(A, blockIdx.x, t, blockDim.x, smem);
    if (A.phase != 1) env_step_phase_b<H_T, HC_T, E_T>(A, blockIdx.x, t, blockDim.x, smem, csc0, ring_step, false);
    // postponed finaliser (HgymEnvOut.defer_finalize): the sampling step the NEXT policy launch reads is bumped here -- no policy
    // kernel is running now, and the finaliser will be one of that launch's workgroups
    if (A.out.defer_finalize && blockIdx.x == 0 && t == 0 && A.out.t_rewards && A.out.t_step) A.out.t_step[0] += 1;
}

__global__ __launch_bounds__(1024) void env_finalize_kernel(const EnvArgs A) { fin_block(fin_of(A), threadIdx.x, blockDim.x); }

// generic options; neither kernel is launched in the XBot-L default configuration
__global__ __launch_bounds__(256) void measure_heights_kernel(const EnvArgs A) {
    const int P = A.cfg.num_height_points;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)A.cfg.num_envs * P) return;
    const int e = (int)(i / P);
    measure_height_point(A, e, (int)(i - (int64_t)e * P));
}

__global__ __launch_bounds__(1024) void command_curriculum_kernel(const EnvArgs A) {
    __shared__ int due;
    __shared__ float xr[2];
    const int64_t csc0 = A.st.counters[0];
    if (threadIdx.x == 0) {
        due = command_curriculum_due(A, A.mode == MODE_STEP ? csc0 + 1 : csc0) ? 1 : 0;   // the finaliser has not bumped the counter yet
        if (due) {
            double lo, hi;
            command_curriculum_move(A, A.st.command_range_x[0], A.st.command_range_x[1], lo, hi);
            A.st.command_range_x[0] = lo;
            A.st.command_range_x[1] = hi;
            xr[0] = (float)lo;
            xr[1] = (float)(hi - lo);
        }
    }
    __syncthreads();
    if (!due) return;
    const RngKey rk = make_rng_key(A, csc0);
    const int64_t ring_step = A.st.counters[2];
    for (int e = threadIdx.x; e < A.cfg.num_envs; e += blockDim.x) command_curriculum_fix_env(A, rk, e, xr[0], xr[1], ring_step);
}

__global__ __launch_bounds__(256) void pre_physics_kernel(const EnvArgs A) {
    const int N = A.cfg.num_envs;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const int64_t csc0 = A.st.counters[0];
    RngKey rk = {(uint32_t)A.cfg.seed, (uint32_t)(A.cfg.seed >> 32), (uint32_t)csc0, (uint32_t)(csc0 >> 32)};
    pre_physics_env(A, rk, e, N);
}

__global__ __launch_bounds__(256) void pd_torques_kernel(const EnvArgs A) {
    const int N = A.cfg.num_envs;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    pd_torques_env(A, e, N);
}

__global__ __launch_bounds__(256) void synth_physics_kernel(const EnvArgs A) {
    const int N = A.cfg.num_envs;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const int64_t csc0 = A.st.counters[0];
    RngKey rk = {(uint32_t)A.cfg.seed, (uint32_t)(A.cfg.seed >> 32), (uint32_t)csc0, (uint32_t)(csc0 >> 32)};
    synth_physics_env(A, rk, e, N);
}

// ------------------------------------------------------------------------------------------------ host side
int device_cus() {
    constexpr int kMaxDev = 64;
    static int cus[kMaxDev];           // 0: not asked yet (a benign race: every thread writes the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev < 0 || dev >= kMaxDev || cus[dev] == 0) {
        hipDeviceProp_t p;
        const int n = hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 0;
        if (dev < 0 || dev >= kMaxDev) return n;
        cus[dev] = n;
    }
    return cus[dev];
}

static int pick_envs_per_block(int N) {
    // 16 envs per workgroup at every size: one workgroup per CU at 4096 envs, two co-resident ones (240 VGPRs) beyond.
    // (32-env workgroups needed 256 + 69 registers, ran one per CU and measured 80 us against 67 us at 16 384 envs.)
    // Multiples of 4 keep every slice of the row-major outputs 16-byte aligned.
    (void)N;
    return 16;
}

static int32_t check_common(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st) {
    HG_REQUIRE(cfg && st, HGYM_E_BADARG, "null cfg/state");
    HG_REQUIRE(cfg->num_envs > 0 && cfg->num_envs <= (1 << 22), HGYM_E_SHAPE, "num_envs=%d (1 .. 4 194 304: the kernels index with 32 bits)",
               cfg->num_envs);
    HG_REQUIRE(cfg->frame_stack >= 1 && cfg->c_frame_stack >= 1, HGYM_E_SHAPE, "frame_stack/c_frame_stack must be >= 1");
    if (sim) HG_REQUIRE(sim->root.base && sim->dof_pos.base && sim->dof_vel.base && sim->contact.base && sim->rigid.base,
                        HGYM_E_BADARG, "null sim tensor");
    HG_REQUIRE(st->episode_length && st->counters && st->commands && st->actions, HGYM_E_BADARG, "null env state field");
    if (cfg->terrain_curriculum) {
        HG_REQUIRE(cfg->custom_origins, HGYM_E_BADARG, "terrain_curriculum needs custom_origins (a height-field / trimesh terrain)");
        HG_REQUIRE(st->terrain_levels && st->terrain_types && st->terrain_origins && cfg->terrain_rows > 0 && cfg->terrain_cols > 0,
                   HGYM_E_BADARG, "terrain_curriculum needs terrain_levels / terrain_types / terrain_origins and their shape");
    }
    if (cfg->num_height_points > 0)
        HG_REQUIRE(st->height_samples && st->height_points && st->height_pose && st->measured_heights && cfg->height_rows > 1 &&
                       cfg->height_cols > 1 && cfg->terrain_hscale > 0.f,
                   HGYM_E_BADARG, "height measurements need height_samples / height_points / height_pose / measured_heights");
    if (cfg->command_curriculum) HG_REQUIRE(st->command_range_x, HGYM_E_BADARG, "command_curriculum needs command_range_x");
    return HGYM_OK;
}

static void launch_measure_heights(const EnvArgs& A, hipStream_t s) {
    const int64_t total = (int64_t)A.cfg.num_envs * A.cfg.num_height_points;
    hipLaunchKernelGGL(measure_heights_kernel, dim3((unsigned)ceil_div(total, (int64_t)256)), dim3(256), 0, s, A);
}

static int32_t launch_step(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                           const HgymEnvNoise* noise, float* actions_in, int mode, int fused, hipStream_t s, int phase = 0) {
    int32_t rc = check_common(cfg, sim, st);
    if (rc) return rc;
    HG_REQUIRE(cfg->num_custom_rewards >= 0 && cfg->num_custom_rewards <= HGYM_MAX_CUSTOM_REWARDS, HGYM_E_SHAPE, "num_custom_rewards=%d",
               cfg->num_custom_rewards);
    if (cfg->num_custom_rewards > 0) {
        HG_REQUIRE(mode != MODE_STEP || phase != 0, HGYM_E_BADARG,
                   "user-defined reward terms need the two-launch step (hgym_env_step_begin / hgym_env_step_end)");
        HG_REQUIRE(st->custom_rew && st->custom_sums && st->custom_acc && out->extras_custom, HGYM_E_BADARG,
                   "user-defined reward terms need custom_rew / custom_sums / custom_acc / extras_custom");
        for (int j = 0; j < cfg->num_custom_rewards; ++j)
            HG_REQUIRE(cfg->custom_reward_pos[j] >= 0 && cfg->custom_reward_pos[j] <= HGYM_NUM_REWARDS + 1, HGYM_E_SHAPE, "custom_reward_pos[%d]=%d",
                       j, cfg->custom_reward_pos[j]);
    }
    HG_REQUIRE(sim && out, HGYM_E_BADARG, "null sim/out");
    HG_REQUIRE(out->obs && out->priv_obs && out->rew && out->reset && out->time_out && out->extras_time_outs && out->extras_episode,
               HGYM_E_BADARG, "null output buffer");
    HG_REQUIRE(st->obs_ring && st->priv_ring && st->episode_acc, HGYM_E_BADARG, "null ring/episode_acc");
    HG_REQUIRE(!out->t_rewards || ((out->t_values || out->t_time_outs) && out->t_dones), HGYM_E_BADARG,
               "transition sink needs t_values (or, deferred values, t_time_outs) and t_dones");
    EnvArgs A;
    memset(&A, 0, sizeof(A));
    A.cfg = *cfg;
    A.sim = *sim;
    A.st = *st;
    A.out = *out;
    A.out.obs_ahead = A.out.priv_ahead = nullptr;      // the rows-ahead protocol exists in hgym_rollout_step only (a stand-alone step
    A.out.obs_older_ready = 0;                          // would write one frame of those rows and nothing else)
    if (noise) A.noise = *noise;
    A.actions_in = actions_in;
    A.origins_hbm = st->env_origins;
    A.mode = mode;
    A.phase = phase;
    A.fused = fused;
    A.envs_per_block = pick_envs_per_block(cfg->num_envs);
    set_body_offsets(A);
    {   // fast staging when the state fields are one contiguous [136][N] allocation
        bool contig = true;
        float* const* f = &st->commands;
        for (int i = 0; i + 1 < kNumStateFields; ++i) contig = contig && (f[i + 1] == f[i] + (int64_t)state_field_comps(i) * cfg->num_envs);
        A.state_contig = contig ? 1 : 0;
    }
    const int blocks = ceil_div(cfg->num_envs, A.envs_per_block);
    const size_t lds = step_smem_bytes(A.envs_per_block);
    prof_begin(HGYM_PROF_ENV_STEP, s);
    const bool std_stack = cfg->frame_stack == 15 && cfg->c_frame_stack == 3;
    // the generic LeggedRobot options (HgymEnvConfig tail) have their own instantiation: off, none of their code is compiled in
    const bool generic = cfg->custom_origins || cfg->terrain_curriculum || cfg->num_height_points > 0 || cfg->command_curriculum ||
                         !cfg->heading_command || phase != 0 || cfg->num_custom_rewards > 0;
    if (std_stack && A.envs_per_block == 16 && !generic && mode == MODE_STEP)
        hipLaunchKernelGGL((env_step_kernel<15, 3, 16, false, true>), dim3(blocks), dim3(256), lds, s, A);
    else if (std_stack && A.envs_per_block == 16 && !generic)
        hipLaunchKernelGGL((env_step_kernel<15, 3, 16, false>), dim3(blocks), dim3(256), lds, s, A);
    else if (std_stack && A.envs_per_block == 16)
        hipLaunchKernelGGL((env_step_kernel<15, 3, 16, true>), dim3(blocks), dim3(256), lds, s, A);
    else
        hipLaunchKernelGGL((env_step_kernel<0, 0, 0, true>), dim3(blocks), dim3(256), lds, s, A);
    {   // algorithmic bytes per env-step, SURVEY.md §8d: 4*[245 + (H-1)*47 + (Hc-1)*73 + H*47 + Hc*73] + 6
        const double H = cfg->frame_stack, HC = cfg->c_frame_stack;
        prof_end(HGYM_PROF_ENV_STEP, s, (double)cfg->num_envs * (4.0 * (245 + (H - 1) * 47 + (HC - 1) * 73 + H * 47 + HC * 73) + 6));
    }
    HG_CHECK_LAUNCH("env_step_kernel");
    if (cfg->num_height_points > 0 && mode == MODE_STEP && phase != 2) {     // (a derive launch samples them: the caller's terms may read them)
        launch_measure_heights(A, s);
        HG_CHECK_LAUNCH("measure_heights_kernel");
    }
    if (phase == 1) return HGYM_OK;                           // the step is finished by hgym_env_step_end
    if (cfg->command_curriculum && mode != MODE_PRIME) {     // before the finaliser consumes the episode-sum accumulators
        hipLaunchKernelGGL(command_curriculum_kernel, dim3(1), dim3(cfg->num_envs > 256 ? 1024 : 256), 0, s, A);
        HG_CHECK_LAUNCH("command_curriculum_kernel");
    }
    if (!out->defer_finalize) {
        hipLaunchKernelGGL(env_finalize_kernel, dim3(1), dim3(cfg->num_envs > 256 ? 1024 : 256), 0, s, A);
        HG_CHECK_LAUNCH("env_finalize_kernel");
    }
    return HGYM_OK;
}

// For hgym_rollout.hip: the EnvArgs record of one synthetic-physics env step in 32-env workgroups, as launch_step builds it,
// after checking everything the fused rollout kernel compiles in (XBot-L default options, 15 / 3 history, contiguous SoA state
// and sim tensors, a whole number of 32-env blocks).
int32_t rollout_env_args(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                         float* actions, EnvArgs* A) {
    int32_t rc = check_common(cfg, sim, st);
    if (rc) return rc;
    HG_REQUIRE(sim && out && actions, HGYM_E_BADARG, "null sim / out / actions");
    HG_REQUIRE(out->obs && out->priv_obs && out->rew && out->reset && out->time_out && out->extras_time_outs && out->extras_episode,
               HGYM_E_BADARG, "null output buffer");
    HG_REQUIRE(st->obs_ring && st->priv_ring && st->episode_acc, HGYM_E_BADARG, "null ring/episode_acc");
    HG_REQUIRE(out->t_rewards && (out->t_values || out->t_time_outs) && out->t_dones && out->t_step && out->defer_finalize, HGYM_E_BADARG,
               "the fused rollout step stores the transition itself: transition sink (immediate: t_values; deferred: t_time_outs) + defer_finalize required");
    const bool generic = cfg->custom_origins || cfg->terrain_curriculum || cfg->num_height_points > 0 || cfg->command_curriculum ||
                         !cfg->heading_command || cfg->num_custom_rewards > 0;
    HG_REQUIRE(!generic && !cfg->use_ref_actions && cfg->frame_stack == 15 && cfg->c_frame_stack == 3, HGYM_E_UNSUPPORTED,
               "fused rollout step: XBot-L default options only");
    HG_REQUIRE(cfg->num_envs % 32 == 0, HGYM_E_UNSUPPORTED, "fused rollout step: num_envs must be a multiple of 32");
    memset(A, 0, sizeof(*A));
    A->cfg = *cfg;
    A->sim = *sim;
    A->st = *st;
    A->out = *out;
    A->actions_in = actions;
    A->origins_hbm = st->env_origins;
    A->mode = MODE_STEP;
    A->fused = 1;
    A->envs_per_block = 32;
    set_body_offsets(*A);
    bool contig = true;
    float* const* f = &st->commands;
    for (int i = 0; i + 1 < kNumStateFields; ++i) contig = contig && (f[i + 1] == f[i] + (int64_t)state_field_comps(i) * cfg->num_envs);
    A->state_contig = contig ? 1 : 0;
    HG_REQUIRE(contig && sim->root.env_stride == 1 && sim->dof_pos.env_stride == 1 && sim->dof_vel.env_stride == 1 &&
                   sim->contact.env_stride == 1 && sim->rigid.env_stride == 1, HGYM_E_UNSUPPORTED,
               "fused rollout step: contiguous [136][N] state and SoA sim tensors required");
    return HGYM_OK;
}

}  // namespace hgym

using namespace hgym;

extern "C" {

int32_t hgym_env_config_default(HgymEnvConfig* c, int32_t num_envs) {
    HG_REQUIRE(c, HGYM_E_BADARG, "null cfg");
    memset(c, 0, sizeof(*c));
    // envs/custom/humanoid_config.py:34-227 ; python-double arithmetic first, fp32 rounding last
    c->num_envs = num_envs;
    c->frame_stack = 15;
    c->c_frame_stack = 3;
    c->decimation = 10;
    c->sim_dt = 0.001f;
    c->dt = (float)(10 * 0.001);
    c->max_episode_length = 2400;   // ceil(24 / 0.01), legged_robot.py:717-718
    c->resample_steps = 800;        // int(8. / 0.01), legged_robot.py:309
    c->push_interval = 400;         // ceil(4 / 0.01), legged_robot.py:720
    c->push_robots = 1;
    c->add_noise = 1;
    c->heading_command = 1;
    c->clip_actions = 18.f;
    c->clip_obs = 18.f;
    c->action_scale = 0.25f;
    c->action_delay = 0.5f;
    c->action_noise = 0.02f;
    c->noise_level = 0.6f;
    for (int k = 5; k < 17; ++k) c->obs_noise[k] = (float)(0.05 * 1.0);
    for (int k = 17; k < 29; ++k) c->obs_noise[k] = (float)(0.5 * 0.05);
    for (int k = 41; k < 44; ++k) c->obs_noise[k] = (float)(0.1 * 1.0);
    for (int k = 44; k < 47; ++k) c->obs_noise[k] = (float)(0.03 * 1.0);
    c->scale_lin_vel = 2.f;
    c->scale_ang_vel = 1.f;
    c->scale_dof_pos = 1.f;
    c->scale_dof_vel = 0.05f;
    c->scale_quat = 1.f;
    c->cmd_x_lo = -0.3f;  c->cmd_x_span = (float)(0.6 - (-0.3));
    c->cmd_y_lo = -0.3f;  c->cmd_y_span = (float)(0.3 - (-0.3));
    c->cmd_h_lo = -3.14f; c->cmd_h_span = (float)(3.14 - (-3.14));
    c->cmd_yaw_lo = -0.3f; c->cmd_yaw_span = (float)(0.3 - (-0.3));
    c->dof_reset_lo = -0.1f; c->dof_reset_span = (float)(0.1 - (-0.1));
    c->push_vel_lo = -0.2f;  c->push_vel_span = (float)(0.2 - (-0.2));
    c->push_ang_lo = -0.4f;  c->push_ang_span = (float)(0.4 - (-0.4));
    const float kp[6] = {200.f, 200.f, 350.f, 350.f, 15.f, 15.f};
    const float eff[6] = {100.f, 100.f, 250.f, 250.f, 100.f, 100.f};
    const float lo[12] = {-0.44f, -1.05f, -1.57f, -1.05f, -0.70f, -0.44f, -1.57f, -1.05f, -1.31f, -1.10f, -0.87f, -0.44f};
    const float hi[12] = {1.57f, 1.05f, 1.31f, 1.10f, 0.87f, 0.44f, 0.44f, 1.05f, 1.57f, 1.05f, 0.70f, 0.44f};
    for (int j = 0; j < 12; ++j) {
        c->p_gains[j] = kp[j % 6];
        c->d_gains[j] = 10.f;
        c->torque_limits[j] = eff[j % 6] * 0.85f;   // fp32 product, legged_robot.py:293
        c->default_dof_pos[j] = 0.f;
        c->dof_lower[j] = lo[j];
        c->dof_upper[j] = hi[j];
    }
    const float init[13] = {0.f, 0.f, 0.95f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 13; ++i) c->base_init_state[i] = init[i];
    c->base_body = 0;
    c->feet_bodies[0] = 6;  c->feet_bodies[1] = 12;
    c->knee_bodies[0] = 4;  c->knee_bodies[1] = 10;
    const double raw[HGYM_NUM_REWARDS] = {-0.002, 0.2, 0.2, -1.0, 0.5, -1e-7, -5e-4, 1.0, 1.0, -0.01, 1.2, 0.2, -0.05, 1.6,
                                          0.2, 0.2, 1.0, -1e-5, 0.5, 1.1, 1.2, 0.5};
    for (int k = 0; k < HGYM_NUM_REWARDS; ++k) c->reward_scales[k] = (float)(raw[k] * (10 * 0.001));
    c->only_positive_rewards = 1;
    c->base_height_target = 0.89f;
    c->min_dist = 0.2f;
    c->max_dist = 0.5f;
    c->target_joint_pos_scale = 0.17f;
    c->target_feet_height = 0.06f;
    c->cycle_time = 0.64f;
    c->tracking_sigma = 5.f;
    c->max_contact_force = 700.f;
    c->episode_length_s = 24.f;
    c->seed = 5;
    return HGYM_OK;
}

int32_t hgym_env_prime(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                       const HgymEnvNoise* noise, void* stream) {
    return launch_step(cfg, sim, st, out, noise, nullptr, MODE_PRIME, 0, (hipStream_t)stream);
}

int32_t hgym_env_reset_all(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                           const HgymEnvNoise* noise, void* stream) {
    return launch_step(cfg, sim, st, out, noise, nullptr, MODE_RESET_ALL, 0, (hipStream_t)stream);
}

int32_t hgym_post_physics(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                          const HgymEnvNoise* noise, void* stream) {
    return launch_step(cfg, sim, st, out, noise, nullptr, MODE_STEP, 0, (hipStream_t)stream);
}

int32_t hgym_env_step_synth(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                            float* actions_in, void* stream) {
    HG_REQUIRE(actions_in, HGYM_E_BADARG, "null actions");
    return launch_step(cfg, sim, st, out, nullptr, actions_in, MODE_STEP, 1, (hipStream_t)stream);
}

int32_t hgym_env_step_begin(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                            const HgymEnvNoise* noise, float* actions_in, void* stream) {
    return launch_step(cfg, sim, st, out, noise, actions_in, MODE_STEP, actions_in ? 1 : 0, (hipStream_t)stream, 1);
}

int32_t hgym_env_step_end(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                          const HgymEnvNoise* noise, void* stream) {
    HG_REQUIRE(out && !out->defer_finalize, HGYM_E_UNSUPPORTED, "the two-launch step runs its finaliser itself (defer_finalize must be 0)");
    return launch_step(cfg, sim, st, out, noise, nullptr, MODE_STEP, 0, (hipStream_t)stream, 2);
}

int32_t hgym_measure_heights(const HgymEnvConfig* cfg, const HgymEnvState* st, void* stream) {
    int32_t rc = check_common(cfg, nullptr, st);
    if (rc) return rc;
    HG_REQUIRE(cfg->num_height_points > 0, HGYM_E_BADARG, "num_height_points is 0");
    EnvArgs A;
    memset(&A, 0, sizeof(A));
    A.cfg = *cfg;
    A.st = *st;
    launch_measure_heights(A, (hipStream_t)stream);
    HG_CHECK_LAUNCH("measure_heights_kernel");
    return HGYM_OK;
}

int32_t hgym_env_finalize(const HgymEnvConfig* cfg, const HgymEnvState* st, const HgymEnvOut* out, void* stream) {
    HG_REQUIRE(cfg && st && out, HGYM_E_BADARG, "null cfg/state/out");
    HG_REQUIRE(st->counters && st->episode_acc && out->time_out && out->extras_time_outs && out->extras_episode, HGYM_E_BADARG,
               "null finaliser buffer");
    EnvArgs A;
    memset(&A, 0, sizeof(A));
    A.cfg = *cfg;
    A.st = *st;
    A.out = *out;
    A.mode = MODE_STEP;
    hipLaunchKernelGGL(env_finalize_kernel, dim3(1), dim3(cfg->num_envs > 256 ? 1024 : 256), 0, (hipStream_t)stream, A);
    HG_CHECK_LAUNCH("env_finalize_kernel");
    return HGYM_OK;
}

static int32_t launch_simple(void (*kern)(const EnvArgs), const char* name, const HgymEnvConfig* cfg, const HgymSimTensors* sim,
                             const HgymEnvState* st, const HgymEnvNoise* noise, float* actions_in, void* stream) {
    int32_t rc = check_common(cfg, sim, st);
    if (rc) return rc;
    EnvArgs A;
    memset(&A, 0, sizeof(A));
    A.cfg = *cfg;
    if (sim) A.sim = *sim;
    A.st = *st;
    if (noise) A.noise = *noise;
    A.actions_in = actions_in;
    set_body_offsets(A);
    hipLaunchKernelGGL(kern, dim3(ceil_div(cfg->num_envs, 256)), dim3(256), 0, (hipStream_t)stream, A);
    HG_CHECK_LAUNCH(name);
    return HGYM_OK;
}

int32_t hgym_pre_physics(const HgymEnvConfig* cfg, const HgymEnvState* st, float* actions_in, const HgymEnvNoise* noise,
                         void* stream) {
    HG_REQUIRE(actions_in, HGYM_E_BADARG, "null actions");
    return launch_simple(pre_physics_kernel, "pre_physics_kernel", cfg, nullptr, st, noise, actions_in, stream);
}

int32_t hgym_pd_torques(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, void* stream) {
    HG_REQUIRE(sim, HGYM_E_BADARG, "null sim");
    return launch_simple(pd_torques_kernel, "pd_torques_kernel", cfg, sim, st, nullptr, nullptr, stream);
}

int32_t hgym_synth_physics(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, void* stream) {
    HG_REQUIRE(sim, HGYM_E_BADARG, "null sim");
    return launch_simple(synth_physics_kernel, "synth_physics_kernel", cfg, sim, st, nullptr, nullptr, stream);
}

}  // extern "C"
