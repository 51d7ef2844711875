// tests/hostcheck/hostcheck.hip -- TEST TOOL, not product code.
// Builds the product's __host__ __device__ env arithmetic (humanoid-gym_amd/csrc/hgym_env_math.hpp) for the
// HOST and drives it with plain loops that emulate the launch geometry of env_step_kernel (phase A over the
// block's first lanes, barrier, phase B over all lanes, then the finaliser).  It lets the CPU-only test suite
// compare the kernels' source-level arithmetic with the oracle before any GPU time is spent.  It is never
// linked into libhgym_hip.so and nothing in the product path loads it.
#include <vector>

#include "../../humanoid-gym_amd/csrc/hgym_env_math.hpp"

namespace hgym {
char* last_error_buf() {
    static thread_local char buf[512];
    return buf;
}
}  // namespace hgym
using namespace hgym;

static EnvArgs make_args(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                         const HgymEnvNoise* noise, float* actions_in, int mode, int fused, int epb, int phase = 0) {
    EnvArgs A;
    memset(&A, 0, sizeof(A));
    A.cfg = *cfg;
    if (sim) A.sim = *sim;
    A.st = *st;
    if (out) A.out = *out;
    if (noise) A.noise = *noise;
    A.actions_in = actions_in;
    A.origins_hbm = st->env_origins;
    A.mode = mode;
    A.phase = phase;
    A.fused = fused;
    A.envs_per_block = epb;
    set_body_offsets(A);
    {   // as launch_step: the fast staging paths when the state fields are one contiguous [136][N] allocation
        bool contig = true;
        float* const* f = &st->commands;
        for (int i = 0; i + 1 < kNumStateFields; ++i) contig = contig && (f[i + 1] == f[i] + (int64_t)state_field_comps(i) * cfg->num_envs);
        A.state_contig = contig ? 1 : 0;
    }
    return A;
}

extern "C" {

// split != 0 (1: chain on one wavefront, 2 / 3: on three, see below): the phase sequence of the XBot-L fast kernels (per-joint work on (env, joint) lanes around a shorter per-env chain;
// env_step_kernel<15, 3, 16, false> and rollout_step_kernel), plain steps of the default options only
int hc_env_step_ex(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                   const HgymEnvNoise* noise, float* actions_in, int mode, int fused, int epb, int nthreads, int split);
// phase 1 / 2: the derive / finish launches of the two-launch step (hgym_env_step_begin / hgym_env_step_end: user-defined reward terms)
int hc_env_step_phase(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                      const HgymEnvNoise* noise, float* actions_in, int fused, int epb, int nthreads, int phase) {
    const EnvArgs A = make_args(cfg, sim, st, out, noise, actions_in, MODE_STEP, phase == 2 ? 0 : fused, epb, phase);
    const int N = cfg->num_envs;
    const int blocks = (N + epb - 1) / epb;
    const int64_t csc0 = st->counters[0], ring = st->counters[2];
    std::vector<float> smem(step_smem_bytes(epb) / sizeof(float));
    for (int b = 0; b < blocks; ++b) {
        for (int t = 0; t < nthreads; ++t) env_stage_in<0>(A, b, t, nthreads, smem.data());
        for (int t = 0; t < nthreads; ++t) env_fill_draws<0>(A, b, t, nthreads, smem.data(), csc0);
        for (int t = 0; t < nthreads; ++t) env_reset_pose<0>(A, t, nthreads, smem.data());
        for (int t = 0; t < nthreads; ++t) env_step_joints<0>(A, b, t, nthreads, smem.data());
        for (int t = 0; t < nthreads; ++t) env_step_phase_a<0, true>(A, b, t, smem.data(), csc0);
        if (phase == 2)
            for (int t = 0; t < nthreads; ++t) {
                if (cfg->frame_stack == 15 && cfg->c_frame_stack == 3) env_step_stack_old<15, 3, 0>(A, b, t, nthreads, ring);
                else env_step_stack_old<0, 0, 0>(A, b, t, nthreads, ring);
            }
        for (int t = 0; t < nthreads; ++t) env_stage_out<0>(A, b, t, nthreads, smem.data());
        if (phase == 2)
            for (int t = 0; t < nthreads; ++t) {
                if (cfg->frame_stack == 15 && cfg->c_frame_stack == 3) env_step_phase_b<15, 3, 0>(A, b, t, nthreads, smem.data(), csc0, ring);
                else env_step_phase_b<0, 0, 0>(A, b, t, nthreads, smem.data(), csc0, ring);
            }
    }
    if (cfg->num_height_points > 0 && phase == 1)
        for (int e = 0; e < N; ++e)
            for (int p = 0; p < cfg->num_height_points; ++p) measure_height_point(A, e, p);
    if (phase == 1) return 0;
    if (cfg->command_curriculum && command_curriculum_due(A, csc0 + 1)) {
        double lo, hi;
        command_curriculum_move(A, st->command_range_x[0], st->command_range_x[1], lo, hi);
        st->command_range_x[0] = lo;
        st->command_range_x[1] = hi;
        const RngKey rk = make_rng_key(A, csc0);
        for (int e = 0; e < N; ++e) command_curriculum_fix_env(A, rk, e, (float)lo, (float)(hi - lo), ring);
    }
    for (int t = 0; t < nthreads; ++t) env_finalize_part1(A, t, nthreads);
    env_finalize_part2(A);
    return 0;
}
int hc_env_step(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                const HgymEnvNoise* noise, float* actions_in, int mode, int fused, int epb, int nthreads) {
    return hc_env_step_ex(cfg, sim, st, out, noise, actions_in, mode, fused, epb, nthreads, 0);
}
int hc_env_step_ex(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                   const HgymEnvNoise* noise, float* actions_in, int mode, int fused, int epb, int nthreads, int split) {
    const EnvArgs A = make_args(cfg, sim, st, out, noise, actions_in, mode, fused, epb);
    const int N = cfg->num_envs;
    const int blocks = (N + epb - 1) / epb;
    const int64_t csc0 = st->counters[0], ring = st->counters[2];
    std::vector<float> smem(step_smem_bytes(epb) / sizeof(float));
    // the previous step's reset flags, as hgym_rollout_step is handed them (prev_out->reset): the rows-ahead protocol below zeroes the
    // frames that step copied ahead for an env it then reset in the NEXT call, as the device does since round 4 (HGYM_RO_AHEAD_CRITIC)
    std::vector<uint8_t> prev_reset(N, 0);
    if (A.out.reset && mode == MODE_STEP) memcpy(prev_reset.data(), A.out.reset, N);
    for (int b = 0; b < blocks; ++b) {
        for (int t = 0; t < nthreads; ++t) env_stage_in<0>(A, b, t, nthreads, smem.data());
        for (int t = 0; t < nthreads; ++t) env_fill_draws<0>(A, b, t, nthreads, smem.data(), csc0);
        for (int t = 0; t < nthreads; ++t) env_reset_pose<0>(A, t, nthreads, smem.data());
        const bool generic = cfg->custom_origins || cfg->terrain_curriculum || cfg->num_height_points > 0 || cfg->command_curriculum ||
                             !cfg->heading_command;
        const bool sp = split && mode == MODE_STEP && !generic;       // as launch_step picks the instantiation
        if (sp && split >= 2) {
            // the chain on four wavefronts (state / reward terms A / reward terms B / frames): no role may read what another writes during the
            // phase, so the order the emulation runs them in must not matter -- split = 2: lanes ascending, split = 3: descending
            for (int t = 0; t < nthreads; ++t) env_step_phase_j<0, true>(A, b, t, nthreads, smem.data());
            if (split == 2)
                for (int t = 0; t < nthreads; ++t) env_step_phase_a3<0>(A, b, t, nthreads, smem.data(), csc0);
            else
                for (int t = nthreads - 1; t >= 0; --t) env_step_phase_a3<0>(A, b, t, nthreads, smem.data(), csc0);
            for (int t = 0; t < nthreads; ++t) env_step_phase_f<0>(A, b, t, nthreads, smem.data());
            for (int t = 0; t < nthreads; ++t) env_step_reward_sum<0>(A, b, t, nthreads, smem.data());
        } else if (sp) {
            for (int t = 0; t < nthreads; ++t) env_step_phase_j<0>(A, b, t, nthreads, smem.data());
            for (int t = 0; t < nthreads; ++t) env_step_phase_a<0, false, true>(A, b, t, smem.data(), csc0);
            for (int t = 0; t < nthreads; ++t) env_step_phase_f<0>(A, b, t, nthreads, smem.data());
        } else {
            for (int t = 0; t < nthreads; ++t) env_step_joints<0>(A, b, t, nthreads, smem.data());
            for (int t = 0; t < nthreads; ++t) env_step_phase_a<0, true>(A, b, t, smem.data(), csc0);
        }
        // The rows-ahead protocol of hgym_rollout_step (HgymEnvOut.obs_ahead / priv_ahead / obs_older_ready; XBot-L geometry): the
        // frames this launch already finds in the ring go into the rows AFTER next through the device's item geometry
        // (hist_slot<H, F, 2>) -- on the device by the wavefronts idle during phase A -- and a launch whose rows were prepared this
        // way copies no older frames itself.
        const bool xbot = cfg->frame_stack == 15 && cfg->c_frame_stack == 3;
        if (xbot && mode == MODE_STEP && A.out.obs_ahead && A.out.priv_ahead) {
            const StackGeom g = stack_geom<15, 3, 0>(A, b);
            auto copy_ahead = [&](auto geom, const float* ring_base, float* dst_base, int H, int F, int slots) {
                for (int le = 0; le < g.nE; ++le)
                    for (int j = 0; j < slots; ++j) {
                        int so, d_o;
                        geom((int)(ring % H), j, so, d_o);
                        for (int k = 0; k < 4; ++k)
                            dst_base[(int64_t)(g.e0 + le) * H * F + d_o + k] =
                                clampf(ring_base[(int64_t)(g.e0 + le) * H * F + so + k], -A.cfg.clip_obs, A.cfg.clip_obs);
                    }
            };
            copy_ahead([](int s, int j, int& so, int& d_o) { hist_slot<15, HGYM_OBS_FRAME, 2>(s, j, so, d_o); }, A.st.obs_ring, A.out.obs_ahead,
                       15, HGYM_OBS_FRAME, HistGeom<15, HGYM_OBS_FRAME, 2>::kSlots);
            copy_ahead([](int s, int j, int& so, int& d_o) { hist_slot<3, HGYM_PRIV_FRAME, 2>(s, j, so, d_o); }, A.st.priv_ring, A.out.priv_ahead,
                       3, HGYM_PRIV_FRAME, HistGeom<3, HGYM_PRIV_FRAME, 2>::kSlots);
        }
        if (!(xbot && A.out.obs_older_ready))
            for (int t = 0; t < nthreads; ++t) {   // the device runs this on its idle wavefronts, concurrently with phase A
                if (xbot) env_step_stack_old<15, 3, 0>(A, b, t, nthreads, ring);
                else env_step_stack_old<0, 0, 0>(A, b, t, nthreads, ring);
            }
        // (the block sizes the device compiles in take their instantiation: its fast path -- env_stage_out_fast -- where the layout allows)
        for (int t = 0; t < nthreads; ++t) {
            if (epb == 16) env_stage_out<16>(A, b, t, nthreads, smem.data());
            else if (epb == 32) env_stage_out<32>(A, b, t, nthreads, smem.data());
            else env_stage_out<0>(A, b, t, nthreads, smem.data());
        }
        const bool ahead = xbot && mode == MODE_STEP && A.out.obs_ahead && A.out.priv_ahead;
        for (int t = 0; t < nthreads; ++t) {
            // (rows ahead: this step does NOT zero what it copied ahead for the envs it resets -- on the device another workgroup copied it)
            if (xbot) env_step_phase_b<15, 3, 0>(A, b, t, nthreads, smem.data(), csc0, ring, false, ahead);
            else env_step_phase_b<0, 0, 0>(A, b, t, nthreads, smem.data(), csc0, ring);
        }
        if (xbot && mode == MODE_STEP && A.out.obs_older_ready) {      // ... the next step does, for ITS next-observation rows
            const StackGeom g = stack_geom<15, 3, 0>(A, b);
            for (int le = 0; le < g.nE; ++le)
                if (prev_reset[g.e0 + le]) {
                    for (int i = 0; i < 13 * HGYM_OBS_FRAME; ++i) A.out.obs[(int64_t)(g.e0 + le) * 15 * HGYM_OBS_FRAME + i] = 0.0f;
                    for (int i = 0; i < HGYM_PRIV_FRAME; ++i) A.out.priv_obs[(int64_t)(g.e0 + le) * 3 * HGYM_PRIV_FRAME + i] = 0.0f;
                }
        }
    }
    if (cfg->num_height_points > 0 && mode == MODE_STEP)
        for (int e = 0; e < N; ++e)
            for (int p = 0; p < cfg->num_height_points; ++p) measure_height_point(A, e, p);
    if (cfg->command_curriculum && mode != MODE_PRIME && command_curriculum_due(A, mode == MODE_STEP ? csc0 + 1 : csc0)) {
        double lo, hi;                                      // command_curriculum_kernel, one lane
        command_curriculum_move(A, st->command_range_x[0], st->command_range_x[1], lo, hi);
        st->command_range_x[0] = lo;
        st->command_range_x[1] = hi;
        const RngKey rk = make_rng_key(A, csc0);
        for (int e = 0; e < N; ++e) command_curriculum_fix_env(A, rk, e, (float)lo, (float)(hi - lo), ring);
    }
    for (int t = 0; t < nthreads; ++t) env_finalize_part1(A, t, nthreads);
    env_finalize_part2(A);
    return 0;
}

// The finaliser's two forms on the same inputs: fused != 0 runs fin_fused (one pass, a lane owns 8 envs, packed loads / stores) for
// every lane, otherwise fin_part1 then fin_store as the general path does.  Returns 1 if the fused form declined (N % 8, alignment).
int hc_finalize_forms(const HgymEnvConfig* cfg, const HgymEnvState* st, const HgymEnvOut* out, int nthreads, int fused) {
    const FinArgs F = make_fin_args(*cfg, *st, *out, FIN_MODE_STEP);
    if (fused) {
        for (int t = 0; t < nthreads; ++t)
            if (!fin_fused(F, t, nthreads)) return 1;
        return 0;
    }
    for (int t = 0; t < nthreads; ++t) fin_part1(F, t, nthreads);
    for (int t = 0; t < nthreads; ++t) fin_store(F, t, nthreads);
    return 0;
}

int hc_pre_physics(const HgymEnvConfig* cfg, const HgymEnvState* st, float* actions_in, const HgymEnvNoise* noise) {
    const EnvArgs A = make_args(cfg, nullptr, st, nullptr, noise, actions_in, MODE_STEP, 0, 4);
    const RngKey rk = make_rng_key(A, st->counters[0]);
    for (int e = 0; e < cfg->num_envs; ++e) pre_physics_env(A, rk, e, cfg->num_envs);
    return 0;
}

int hc_pd_torques(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st) {
    const EnvArgs A = make_args(cfg, sim, st, nullptr, nullptr, nullptr, MODE_STEP, 0, 4);
    for (int e = 0; e < cfg->num_envs; ++e) pd_torques_env(A, e, cfg->num_envs);
    return 0;
}

int hc_synth_physics(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st) {
    const EnvArgs A = make_args(cfg, sim, st, nullptr, nullptr, nullptr, MODE_STEP, 0, 4);
    const RngKey rk = make_rng_key(A, st->counters[0]);
    for (int e = 0; e < cfg->num_envs; ++e) synth_physics_env(A, rk, e, cfg->num_envs);
    return 0;
}

// raw Philox / normal streams for pinning oracle/philox.py
void hc_philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t* out4) {
    U4 c = {c0, c1, c2, c3};
    const U4 r = philox4x32_10(c, k0, k1);
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}
void hc_streams(uint64_t seed, int64_t step, uint32_t env, uint32_t slot, int n, float* uniforms, float* normals) {
    RngKey rk = {(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    for (int i = 0; i < n; ++i) {
        uniforms[i] = uniform_at(rk, env, slot, i);
        normals[i] = normal_at(rk, env, slot, i);
    }
}
}
