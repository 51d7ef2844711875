// hgym_env_math.hpp -- per-env arithmetic of the XBot-L env step.
//
// Every function is __host__ __device__: the HIP kernels in hgym_env.hip call them with one thread per
// env, and tests/hostcheck builds the very same source for the host so the arithmetic can be compared
// with the oracle without a GPU.  fp32 operation order follows the reference's torch expressions
// (cited per block, paths under /root/reference/humanoid) so masks come out bit-identical and floats
// agree to the last ulp or two (libm vs ocml transcendental rounding).
//
// Layout: state fields are [C][N] fp32 (env-major SoA): component c of env e is p[c*N + e].
#pragma once
#include "hgym_common.hpp"
#include "hgym_finalize.hpp"

namespace hgym {

constexpr float kTwoPi = 6.2831855f;   // fp32(2*pi)
constexpr float kPi = 3.1415927f;      // fp32(pi)
constexpr float kHalfPi = 1.5707964f;  // fp32(pi/2)

enum { MODE_STEP = 0, MODE_PRIME = 1, MODE_RESET_ALL = 2 };

struct EnvArgs {
    HgymEnvConfig cfg;
    HgymSimTensors sim;
    HgymEnvState st;
    HgymEnvOut out;
    HgymEnvNoise noise;
    float* actions_in;        // (N,12) row-major or null; written only when cfg.use_ref_actions (humanoid_env.py:190-191)
    float* origins_hbm;       // st.env_origins as the caller gave it (the LDS shadow replaces st.env_origins by its staged copy)
    int64_t* reset_count;     // where resetting envs count themselves for the step finaliser; null: &st.counters[1]
    int mode;
    int phase;                // user-defined reward terms (cfg.num_custom_rewards > 0) split the step in two launches: 1 = derive (action
                              // processing, physics, derived state, commands, pushes, termination flags -- what the reference has
                              // done when compute_reward starts), 2 = finish (the reward sum with the caller's terms merged in,
                              // reset, observations, tail); 0 = the whole step in one launch
    int fused;                // 1: pre_physics + synthetic physics run inside the step kernel
    int envs_per_block;
    int state_contig;         // 1: the 22 [C][N] state fields are adjacent in memory, in HgymEnvState order
    int env_base;             // global id of env index 0 of the state arrays (0, or the block's first env for an LDS shadow)
    int contact_comp[3];      // component offset of the xyz triple of {base, foot L, foot R} in sim.contact
    int rigid_comp[4];        // component offset of the 13-vector of {foot L, foot R, knee L, knee R} in sim.rigid
};

HG_HD void set_body_offsets(EnvArgs& A) {   // full Isaac-Gym-shaped tensors: body-major components
    A.env_base = 0;
    A.contact_comp[0] = A.cfg.base_body * 3;
    A.contact_comp[1] = A.cfg.feet_bodies[0] * 3;
    A.contact_comp[2] = A.cfg.feet_bodies[1] * 3;
    A.rigid_comp[0] = A.cfg.feet_bodies[0] * 13;
    A.rigid_comp[1] = A.cfg.feet_bodies[1] * 13;
    A.rigid_comp[2] = A.cfg.knee_bodies[0] * 13;
    A.rigid_comp[3] = A.cfg.knee_bodies[1] * 13;
}

// 32-bit index arithmetic (check_common bounds num_envs so that every product fits): with 64-bit indices the per-env chain, which
// runs against an LDS image through these accessors, spent a third of its ~4200 instructions on 64-bit address arithmetic
// (v_mad_u64_u32 / v_lshl_add_u64) for addresses that end up as 32-bit LDS offsets
HG_HD float sget(const HgymStrided& s, int env, int comp) { return s.base[env * (int)s.env_stride + comp * (int)s.comp_stride]; }
HG_HD void sset(const HgymStrided& s, int env, int comp, float v) { s.base[env * (int)s.env_stride + comp * (int)s.comp_stride] = v; }

#define FG(p, c) (p)[(c) * N + e]

HG_HD void hg_atomic_add(float* p, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(p, v);
#else
    *p += v;
#endif
}
HG_HD int hg_atomic_inc_int(int* p) {      // returns the previous value
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, 1);
#else
    return (*p)++;
#endif
}
HG_HD void hg_atomic_inc(int64_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd((unsigned long long*)p, 1ull);
#else
    *p += 1;
#endif
}

// ------------------------------------------------------------------------------------------------
// isaacgym.torch_utils restated as xyzw quaternion math (third-party, absent from /root/reference;
// call sites legged_robot.py:133-135,215,312).
HG_HD void quat_rotate_inverse(const float q[4], const float v[3], float o[3]) {
    const float w = q[3];
    const float s = 2.0f * (w * w) - 1.0f;
    const float cx = q[1] * v[2] - q[2] * v[1];
    const float cy = q[2] * v[0] - q[0] * v[2];
    const float cz = q[0] * v[1] - q[1] * v[0];
    const float d = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
    o[0] = v[0] * s - cx * w * 2.0f + q[0] * d * 2.0f;
    o[1] = v[1] * s - cy * w * 2.0f + q[1] * d * 2.0f;
    o[2] = v[2] * s - cz * w * 2.0f + q[2] * d * 2.0f;
}

HG_HD void quat_apply(const float q[4], const float v[3], float o[3]) {
    const float tx = (q[1] * v[2] - q[2] * v[1]) * 2.0f;
    const float ty = (q[2] * v[0] - q[0] * v[2]) * 2.0f;
    const float tz = (q[0] * v[1] - q[1] * v[0]) * 2.0f;
    o[0] = v[0] + q[3] * tx + (q[1] * tz - q[2] * ty);
    o[1] = v[1] + q[3] * ty + (q[2] * tx - q[0] * tz);
    o[2] = v[2] + q[3] * tz + (q[0] * ty - q[1] * tx);
}

// torch.remainder(x, 2pi) for fp32 (sign of the divisor), then the (-pi, pi] fold of legged_robot.py:54
// and utils/math.py:46-49.  The reference really does add and subtract fp32(2pi) for negative angles, which
// quantises them to the fp32 grid near 2pi; reproduce it rather than "fixing" it.
HG_HD float wrap_mod_2pi(float x) {
    float r = fmodf(x, kTwoPi);
    if (r != 0.0f && r < 0.0f) r += kTwoPi;
    return r;
}
HG_HD float wrap_like_reference(float x) {
    float r = wrap_mod_2pi(x);
    if (r > kPi) r -= kTwoPi;
    return r;
}

HG_HD void euler_xyz_wrapped(const float q[4], float e[3]) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float roll = atan2f(2.0f * (w * x + y * z), w * w - x * x - y * y + z * z);
    const float sp = 2.0f * (w * y - z * x);
    const float sgn = (sp > 0.0f) ? 1.0f : ((sp < 0.0f) ? -1.0f : 0.0f);
    const float pitch = (fabsf(sp) >= 1.0f) ? sgn * kHalfPi : asinf(sp);
    const float yaw = atan2f(2.0f * (w * z + x * y), w * w + x * x - y * y - z * z);
    e[0] = wrap_like_reference(roll);
    e[1] = wrap_like_reference(pitch);
    e[2] = wrap_like_reference(yaw);
}

// ------------------------------------------------------------------------------------------------ noise
// `e` indexes the (possibly LDS-shadowed) table row, `ge` is the global env id that keys the Philox stream
HG_HD float nz_uniform(const float* tab, int width, int col, const RngKey& k, int e, int ge, uint32_t slot, int i) {
    return tab ? tab[(int64_t)e * width + col] : uniform_at(k, (uint32_t)ge, slot, i);
}
HG_HD float nz_normal(const float* tab, int width, int col, const RngKey& k, int e, int ge, uint32_t slot, int i) {
    return tab ? tab[(int64_t)e * width + col] : normal_at(k, (uint32_t)ge, slot, i);
}

// ------------------------------------------------------------------------------------------------ gait clock
// humanoid_env.py:100-118: phase = int64 * fp32(dt) / fp32(cycle_time); s = sin(fp32(2pi) * phase).
HG_HD float gait_phase(const HgymEnvConfig& c, int64_t ep) { return (float)ep * c.dt / c.cycle_time; }

HG_HD void stance_from_sin(float s, float st[2]) {
    st[0] = (s >= 0.0f) ? 1.0f : 0.0f;
    st[1] = (s < 0.0f) ? 1.0f : 0.0f;
    if (fabsf(s) < 0.1f) st[0] = st[1] = 1.0f;
}

// ------------------------------------------------------------------------------------------------ E1/E2
// humanoid_env.py:189-197 + legged_robot.py:90-91, one joint: u = delay draw, z = action-noise normal
HG_HD float filter_action(const HgymEnvConfig& c, float a_in, float a_prev, float u, float z) {
    const float delay = u * c.action_delay;
    float a = clampf(a_in, -c.clip_actions, c.clip_actions);
    a = (1.0f - delay) * a + delay * a_prev;
    a = a + c.action_noise * z * a;
    return clampf(a, -c.clip_actions, c.clip_actions);
}
HG_HD void pre_physics_joint(const EnvArgs& A, int e, int N, int j, float u, float z) {
    float a_in = A.actions_in[(int64_t)e * 12 + j];
    if (A.cfg.use_ref_actions) {          // actions += self.ref_action (= 2 * ref_dof_pos of the last compute_observations), in place
        a_in = a_in + 2.0f * FG(A.st.ref_dof_pos, j);
        A.actions_in[(int64_t)e * 12 + j] = a_in;
    }
    FG(A.st.actions, j) = filter_action(A.cfg, a_in, FG(A.st.actions, j), u, z);
}

HG_HD void pre_physics_env(const EnvArgs& A, const RngKey& rk, int e, int N) {
    const int ge = A.env_base + e;
    const float u = nz_uniform(A.noise.u_delay, 1, 0, rk, e, ge, SLOT_DELAY_CMD, 0);
    float zn[12];
    if (!A.noise.z_act) normals_block<3>(rk, (uint32_t)ge, SLOT_ACT, zn);
#pragma unroll
    for (int j = 0; j < 12; ++j) pre_physics_joint(A, e, N, j, u, A.noise.z_act ? A.noise.z_act[(int64_t)e * 12 + j] : zn[j]);
}

// legged_robot.py:340-356
// The six per-joint constant arrays of the configuration as ONE block of 72 floats (p_gains, d_gains, torque_limits, default_dof_pos,
// dof_lower, dof_upper).  Code that indexes them with a lane-varying joint takes the block as a pointer `jc`: the workgroup kernels
// pass their LDS copy (LdsMap::jcfg, filled by the stage-in) -- indexing the kernel ARGUMENT by a lane-varying j is a load from
// memory with a full round trip in front of the arithmetic, once per phase that does it.
constexpr int kJcP = 0, kJcD = 12, kJcTq = 24, kJcDef = 36, kJcLo = 48, kJcHi = 60, kJointConsts = 72;
static_assert(offsetof(HgymEnvConfig, d_gains) == offsetof(HgymEnvConfig, p_gains) + 48 &&
              offsetof(HgymEnvConfig, torque_limits) == offsetof(HgymEnvConfig, p_gains) + 96 &&
              offsetof(HgymEnvConfig, default_dof_pos) == offsetof(HgymEnvConfig, p_gains) + 144 &&
              offsetof(HgymEnvConfig, dof_lower) == offsetof(HgymEnvConfig, p_gains) + 192 &&
              offsetof(HgymEnvConfig, dof_upper) == offsetof(HgymEnvConfig, p_gains) + 240, "the per-joint constants are one block");
HG_HD const float* joint_consts(const HgymEnvConfig& c) { return reinterpret_cast<const float*>(&c) + offsetof(HgymEnvConfig, p_gains) / 4; }
#define HGYM_JC(A, smem, m) ((const float*)((smem) + (m).jcfg))

HG_HD float pd_torque(const HgymEnvConfig& c, const float* jc, int j, float a, float q, float qd) {
    const float t = jc[kJcP + j] * (a * c.action_scale + jc[kJcDef + j] - q) - jc[kJcD + j] * qd;
    return clampf(t, -jc[kJcTq + j], jc[kJcTq + j]);
}

HG_HD void pd_torques_env(const EnvArgs& A, int e, int N) {
#pragma unroll
    for (int j = 0; j < 12; ++j)
        FG(A.st.torques, j) = pd_torque(A.cfg, joint_consts(A.cfg), j, FG(A.st.actions, j), sget(A.sim.dof_pos, e, j), sget(A.sim.dof_vel, e, j));
}

// ------------------------------------------------------------------------------------------------ synthetic physics
// Stands where PhysX is (legged_robot.py:94-101,124-126).  SURVEY.md §8d: unit-inertia joints under the PD
// torque, `decimation` semi-implicit Euler substeps with URDF joint limits; root / contact / rigid-body
// tensors drawn from Philox.  This is the benchmark backend only -- it has no reference counterpart.
// Split in three so the step kernel can spread it over lanes: the random draws (one Philox call per work item),
// the joint integration (one (env, joint) pair per lane) and the per-env remainder.
constexpr int kPhysDraws = 36;   // r0: 4 uniforms | n: 12 normals | r1: 4 uniforms | m: 12 normals | r2: 4 uniforms
constexpr int kPhysCalls = 9;    // Philox calls SLOT_PHYS + 0 .. 8

// draws of call c (0..8) -> tab[4c .. 4c+3]
HG_HD void phys_draw_call(const RngKey& rk, uint32_t ue, int c, float* tab) {
    const U4 r = rng4(rk, ue, SLOT_PHYS + (uint32_t)c);
    const bool normal = (c >= 1 && c <= 3) || (c >= 5 && c <= 7);
    if (normal) {
        box_muller(r.x, r.y, tab[4 * c + 0], tab[4 * c + 1]);
        box_muller(r.z, r.w, tab[4 * c + 2], tab[4 * c + 3]);
    } else {
        tab[4 * c + 0] = u01(r.x);
        tab[4 * c + 1] = u01(r.y);
        tab[4 * c + 2] = u01(r.z);
        tab[4 * c + 3] = u01(r.w);
    }
}

// `decimation` PD + semi-implicit Euler substeps of one unit-inertia joint; t = the last torque evaluation
HG_HD void integrate_joint(const HgymEnvConfig& c, const float* jc, int j, float a, float& q, float& qd, float& t) {
    t = 0.0f;
    const float lo = jc[kJcLo + j], hi = jc[kJcHi + j];
    for (int s = 0; s < c.decimation; ++s) {
        t = pd_torque(c, jc, j, a, q, qd);
        qd = qd + c.sim_dt * t;
        q = q + c.sim_dt * qd;
        if (q < lo) { q = lo; qd = 0.0f; }
        if (q > hi) { q = hi; qd = 0.0f; }
    }
}
HG_HD void synth_joint(const EnvArgs& A, int e, int N, int j) {
    float q = sget(A.sim.dof_pos, e, j), qd = sget(A.sim.dof_vel, e, j), t;
    integrate_joint(A.cfg, joint_consts(A.cfg), j, FG(A.st.actions, j), q, qd, t);
    // the reference evaluates the torque before each substep; the last evaluation is what rewards see
    FG(A.st.torques, j) = t;
    sset(A.sim.dof_pos, e, j, q);
    sset(A.sim.dof_vel, e, j, qd);
}

// tab: this env's kPhysDraws draws.  Two independent halves (the step kernel runs them on two different wavefronts):
// the root pose / velocities + the rare base-link hit (draws 0 .. 15), and the feet contact loads + feet / knee rigid-body
// entries (draws 16 .. 35, gait clock).  Each reads every input before its first store: the stores go through the same untyped
// float pointers, so a read placed after one of them cannot be moved above it by the compiler and costs its own LDS round trip.
HG_HD void synth_root_env(const EnvArgs& A, const float* tab, int e, int N) {
    float d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = tab[i];
    const float r3 = sget(A.sim.root, e, 3), r4 = sget(A.sim.root, e, 4), r5 = sget(A.sim.root, e, 5);
    const float* n = d + 4;
    // root: mean-reverting orientation walk, small height jitter, gaussian velocities
    float qx = 0.9f * r3 + 0.05f * n[0];
    float qy = 0.9f * r4 + 0.05f * n[1];
    float qz = 0.9f * r5 + 0.05f * n[2];
    const float inv = 1.0f / sqrtf(qx * qx + qy * qy + qz * qz + 1.0f);
    sset(A.sim.root, e, 3, qx * inv);
    sset(A.sim.root, e, 4, qy * inv);
    sset(A.sim.root, e, 5, qz * inv);
    sset(A.sim.root, e, 6, inv);
    sset(A.sim.root, e, 2, 0.9f + 0.02f * (2.0f * d[0] - 1.0f));
#pragma unroll
    for (int i = 0; i < 6; ++i) sset(A.sim.root, e, 7 + i, 0.3f * n[3 + i]);
    // rare base-link hits end episodes (~ every 500 steps)
    const float hit = (d[1] < 0.002f) ? 2.0f : 0.0f;
    sset(A.sim.contact, e, A.contact_comp[0] + 0, hit * n[9]);
    sset(A.sim.contact, e, A.contact_comp[0] + 1, hit * n[10]);
    sset(A.sim.contact, e, A.contact_comp[0] + 2, hit * n[11]);
}
HG_HD void synth_feet_env(const EnvArgs& A, const float* tab, int e, int N) {
    const HgymEnvConfig& c = A.cfg;
    float d[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) d[i] = tab[16 + i];
    const int64_t ep_next = A.st.episode_length[e] + 1;
    const float* m = d + 4;
    // contacts: feet load follows the gait clock (all other contact entries stay at their initial zero)
    const float s = sinf(kTwoPi * gait_phase(c, ep_next));
    float stance[2];
    stance_from_sin(s, stance);
    const float uf[2] = {d[0], d[1]};
    const float ug[2] = {d[2], d[3]};
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const float on = (stance[f] > 0.5f || ug[f] > 0.4f) ? 1.0f : 0.0f;
        sset(A.sim.contact, e, A.contact_comp[1 + f] + 2, 600.0f * uf[f] * on);
    }
    // rigid bodies: only the entries the rewards read (feet x,y,z,vx,vy ; knees x,y)
    const float uz[2] = {d[16], d[17]};
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int fb = A.rigid_comp[f], kb = A.rigid_comp[2 + f];
        const float side = f == 0 ? 0.15f : -0.15f;
        sset(A.sim.rigid, e, fb + 0, 0.2f * m[f * 6 + 0]);
        sset(A.sim.rigid, e, fb + 1, side + 0.05f * m[f * 6 + 1]);
        sset(A.sim.rigid, e, fb + 2, 0.03f + 0.09f * uz[f]);
        sset(A.sim.rigid, e, fb + 7, 0.2f * m[f * 6 + 2]);
        sset(A.sim.rigid, e, fb + 8, 0.2f * m[f * 6 + 3]);
        sset(A.sim.rigid, e, kb + 0, 0.2f * m[f * 6 + 4]);
        sset(A.sim.rigid, e, kb + 1, 0.8f * side + 0.05f * m[f * 6 + 5]);
    }
}
HG_HD void synth_rest_env(const EnvArgs& A, const float* tab, int e, int N) {
    synth_root_env(A, tab, e, N);
    synth_feet_env(A, tab, e, N);
}

HG_HD void synth_physics_env(const EnvArgs& A, const RngKey& rk, int e, int N) {
#pragma unroll
    for (int j = 0; j < 12; ++j) synth_joint(A, e, N, j);
    float tab[kPhysDraws];
#pragma unroll
    for (int c = 0; c < kPhysCalls; ++c) phys_draw_call(rk, (uint32_t)(A.env_base + e), c, tab);
    synth_rest_env(A, tab, e, N);
}

// ------------------------------------------------------------------------------------------------ commands
// legged_robot.py:322-336 for one env; u[3] = draws for x, y, heading
// lin_vel_x range: the configuration's, or (command curriculum) the live device-resident [lo, hi] python doubles
template <bool kGeneric>
HG_HD void cmd_x_range(const EnvArgs& A, float& lo, float& span) {
    lo = A.cfg.cmd_x_lo;
    span = A.cfg.cmd_x_span;
    if (kGeneric && A.cfg.command_curriculum && A.st.command_range_x) {
        const double l = A.st.command_range_x[0], h = A.st.command_range_x[1];
        lo = (float)l;
        span = (float)(h - l);      // torch_rand_float: (hi - lo) in python double, then against the fp32 tensor
    }
}

// heading = true: the third draw is the heading target (:331-332); false: the yaw rate itself (:333-334)
HG_HD void resample_commands(const HgymEnvConfig& c, float x_lo, float x_span, float cmd[4], const float u[3], bool heading = true) {
    cmd[0] = x_span * u[0] + x_lo;
    cmd[1] = c.cmd_y_span * u[1] + c.cmd_y_lo;
    if (heading) cmd[3] = c.cmd_h_span * u[2] + c.cmd_h_lo;
    else cmd[2] = c.cmd_yaw_span * u[2] + c.cmd_yaw_lo;
    const float keep = (sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) > 0.2f) ? 1.0f : 0.0f;
    cmd[0] *= keep;
    cmd[1] *= keep;
}

// exp / sqrt of the reward terms.  On the device: v_exp_f32(x * log2 e) and v_sqrt_f32 (1 ulp; the exponent scaling adds
// <= |x| * 1e-7 relative, i.e. < 2e-6 for the arguments that do not underflow) instead of libm's ~15 / ~10 instruction
// sequences -- 16 exponentials and 6 roots sit on the single-wave per-env chain of the step kernel.  Nothing thresholded
// (masks, termination, the command dead-band) goes through these.  The host emulation keeps libm.  HGYM_ENV_FAST=0: libm.
HG_HD float r_exp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
#else
    return expf(x);
#endif
}
HG_HD float r_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
HG_HD float dist_reward(const HgymEnvConfig& c, float ax, float ay, float bx, float by, float max_df) {
    const float dx = ax - bx, dy = ay - by;
    const float d = r_sqrt(dx * dx + dy * dy);
    const float d_min = clampf(d - c.min_dist, -0.5f, 0.0f);
    const float d_max = clampf(d - max_df, 0.0f, 0.5f);
    return (r_exp(-fabsf(d_min) * 100.0f) + r_exp(-fabsf(d_max) * 100.0f)) / 2.0f;
}

struct StepFlags {
    int reset;     // env resets this step (history must be zeroed before the push)
};

// The per-joint products of the reward terms that sum over the 12 joints (humanoid_env.py: 0 action_smoothness :530-540,
// 4 default_joint_pos :362-372, 5 dof_acc :516-521, 6 dof_vel :509-514, 17 torques :502-507, 13 joint_pos :272-280), joint j:
// a = this step's action, la / lla = the previous two, ldv = last joint velocity, tq = torque, rdp = reference pose of the
// previous step.
HG_HD void joint_terms(const HgymEnvConfig& c, const float* jc, int j, float a, float la, float lla, float ldv, float q, float qd, float tq,
                       float rdp, float (&o)[8]) {
    const float d1 = la - a;
    o[0] = d1 * d1;
    const float d2 = a + lla - 2.0f * la;
    o[1] = d2 * d2;
    o[2] = fabsf(a);
    const float jd = q - jc[kJcDef + j];
    o[3] = jd * jd;
    const float ac = (ldv - qd) / c.dt;
    o[4] = ac * ac;
    o[5] = qd * qd;
    o[6] = tq * tq;
    const float d = q - rdp;
    o[7] = d * d;
}

// ------------------------------------------------------------------------------------------------ E4-E12
// LeggedRobot.post_physics_step for ONE env (legged_robot.py:119-151) with XBotLFreeEnv's reward terms
// (humanoid_env.py:272-540, alphabetical order), mask-driven reset_idx (legged_robot.py:163-215) and the
// clean observation frames (humanoid_env.py:200-244).  frame47 / priv73 receive the UN-noised new frames
// (LDS on the device); noise, history stacking and clipping happen in the cooperative phase.
// kGeneric = false compiles the generic LeggedRobot options (terrain map, curricula, height measurements) OUT: the XBot-L
// default configuration runs the instantiation that has none of their branches on its per-env latency chain.
// kSplit (the compiled-in fast kernels, MODE_STEP only): everything that is the same few instructions for each of the 12 joints
// has been taken off this single-wavefront chain and runs one (env, joint) pair per lane around it -- the per-joint products of
// reward terms 0 / 4 / 5 / 6 / 13 / 17 before it (env_step_joint_terms -> `jpart`, summed here in the reference's order), the
// per-joint part of the reset, the reference pose, the per-joint frame entries and the last_* write-back after it
// (env_step_phase_f, which takes this step's reset flag and gait-clock sine from `cscal`).  Same arithmetic, same order.
constexpr int kJointTerms = 8;     // per-joint products: d1^2, d2^2, |a| (term 0), jd^2 (4), acc^2 (5), qd^2 (6), tq^2 (17), (q - ref)^2 (13)
// ROLE (split chain only): the chain is ONE wavefront issuing ~2 500 instructions once, and a lone wavefront issues one instruction
// (of any kind) per four cycles; four wavefronts of the workgroup -- on the CU's four SIMDs -- can share it.
//   ROLE_ALL    everything (one wavefront, the form above);
//   ROLE_MAIN   the state: derived state, commands, push, termination, reset, write-back -- no reward terms, no frames;
//   ROLE_REW_A / ROLE_REW_B   the derived quantities their terms need (re-derived from a SNAPSHOT of root state / commands / episode
//               length / last root velocity taken before the phase: ROLE_MAIN rewrites those while the others run) and their share of
//               the 22 terms, left times their scale in `tscr` [22][N]; B owns the stateful terms 7 / 8 and their four state fields;
//   ROLE_FRAMES the same re-derivation, the values a reset changes, and the non-joint entries of the two clean observation frames.
// env_step_reward_sum then forms the reward and the episode sums in the reference's order.  No wavefront reads what another writes
// during the phase; tests/hostcheck runs the roles in both orders.
constexpr int ROLE_ALL = 0, ROLE_MAIN = 1, ROLE_REW_A = 2, ROLE_REW_B = 3, ROLE_FRAMES = 4;
constexpr int kChainRoles = 4;       // wavefronts of env_step_phase_a3
HG_HD constexpr bool term_in_role(int k, int role) {
    // A: the terms of the base velocities / orientation / commands (they need the quaternion prefix); B: feet, gait clock, joint sums
    const bool a = k == 1 || k == 15 || k == 16 || k == 18 || k == 19 || k == 20 || k == 21;
    return role == ROLE_ALL || (role == ROLE_REW_A && a) || (role == ROLE_REW_B && !a);
}
template <bool kGeneric, bool kSplit = false, int ROLE = ROLE_ALL>
HG_HD StepFlags post_physics_env(const EnvArgs& A, const RngKey& rk, int64_t csc, int e, int N, float* frame47,
                                 float* priv73, const float* jpart = nullptr, float* cscal = nullptr, const float* reset_pose = nullptr,
                                 float* tscr = nullptr) {
    static_assert(ROLE == ROLE_ALL || (kSplit && !kGeneric), "the chain by roles exists for the split XBot-L chain only");
    constexpr bool kMain = ROLE == ROLE_ALL || ROLE == ROLE_MAIN;      // state write-back, reset
    constexpr bool kFrames = ROLE == ROLE_ALL || ROLE == ROLE_FRAMES;  // the clean observation frames
    constexpr bool kFeet = ROLE == ROLE_ALL || ROLE == ROLE_REW_B;     // owner of last_contacts / feet_air_time / feet_height / last_feet_z
    const HgymEnvConfig& c = A.cfg;
    const HgymEnvState& S = A.st;
    const int mode = A.mode;
    const int ge = A.env_base + e;
    StepFlags fl;
    fl.reset = 0;

    int64_t ep = S.episode_length[e];
    float root[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) root[i] = sget(A.sim.root, e, i);
    float q[12], qd[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const bool need = !kSplit || j == 0 || j == 1 || j == 6 || j == 7;     // split: only the four hip joints of term 4
        q[j] = need ? sget(A.sim.dof_pos, e, j) : 0.0f;
        qd[j] = kSplit ? 0.0f : sget(A.sim.dof_vel, e, j);
    }
    float cmd[4] = {FG(S.commands, 0), FG(S.commands, 1), FG(S.commands, 2), FG(S.commands, 3)};
    float act[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) act[j] = kSplit ? 0.0f : FG(S.actions, j);
    float blv[3], bav[3], grav[3], eul[3];
    float fz[2], contact[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        fz[f] = sget(A.sim.contact, e, A.contact_comp[1 + f] + 2);
        contact[f] = fz[f] > 5.0f ? 1.0f : 0.0f;
    }
    int reset = 0, time_out = 0;
    float rew = 0.0f;
    // Everything the step arithmetic reads from the state / sim tensors is fetched HERE, before the first store.  All of it
    // goes through untyped float pointers (on the device: into the LDS shadow), so a read placed after a store cannot be
    // hoisted above it and pays its own LDS round trip; this function is one quarter-filled wavefront's serial chain.
    float la[12], lla[12], ldv[12], tqv[12], rdp[12], lrv[6], esum[HGYM_NUM_REWARDS];
    float lcon[2], fat[2], fhs[2], lfz[2], pf[2], pt[3], org[3], bxyz[3];
    float fpos[2][3], fvxy[2][2], kxy[2][2], fxyz[2][3];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        la[j] = kSplit ? 0.0f : FG(S.last_actions, j);
        lla[j] = kSplit ? 0.0f : FG(S.last_last_actions, j);
        ldv[j] = kSplit ? 0.0f : FG(S.last_dof_vel, j);
        tqv[j] = kSplit ? 0.0f : FG(S.torques, j);
        rdp[j] = kSplit ? 0.0f : FG(S.ref_dof_pos, j);
    }
    // split: the sums over the joints of the per-joint products, accumulated j = 0 .. 11 like the loops they replace
    float jsum[kJointTerms];
#pragma unroll
    for (int k = 0; k < kJointTerms; ++k) {
        jsum[k] = 0.0f;
        if (kSplit && kFeet) {
#pragma unroll
            for (int j = 0; j < 12; ++j) jsum[k] += jpart[(k * 12 + j) * N + e];
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) lrv[i] = FG(S.last_root_vel, i);
#pragma unroll
    for (int k = 0; k < HGYM_NUM_REWARDS; ++k) esum[k] = FG(S.episode_sums, k);
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        lcon[f] = FG(S.last_contacts, f);
        fat[f] = FG(S.feet_air_time, f);
        fhs[f] = FG(S.feet_height, f);
        lfz[f] = FG(S.last_feet_z, f);
        pf[f] = FG(S.push_force, f);
        const int fb = A.rigid_comp[f], kb = A.rigid_comp[2 + f];
        fpos[f][0] = sget(A.sim.rigid, e, fb + 0);
        fpos[f][1] = sget(A.sim.rigid, e, fb + 1);
        fpos[f][2] = sget(A.sim.rigid, e, fb + 2);
        fvxy[f][0] = sget(A.sim.rigid, e, fb + 7);
        fvxy[f][1] = sget(A.sim.rigid, e, fb + 8);
        kxy[f][0] = sget(A.sim.rigid, e, kb + 0);
        kxy[f][1] = sget(A.sim.rigid, e, kb + 1);
        fxyz[f][0] = sget(A.sim.contact, e, A.contact_comp[1 + f] + 0);
        fxyz[f][1] = sget(A.sim.contact, e, A.contact_comp[1 + f] + 1);
        fxyz[f][2] = fz[f];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        pt[i] = FG(S.push_torque, i);
        org[i] = FG(S.env_origins, i);
        bxyz[i] = sget(A.sim.contact, e, A.contact_comp[0] + i);
    }
    const float friction0 = FG(S.friction, 0), body_mass0 = FG(S.body_mass, 0);

    const int phase = kGeneric ? A.phase : 0;
    if (mode == MODE_STEP && phase == 2) {
        // finish launch: the derive launch left the incremented episode length, the derived state, the (possibly resampled)
        // commands and the pushes in the state / sim tensors; the termination flags are re-derived from the same inputs below
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            blv[i] = FG(S.base_lin_vel, i);
            bav[i] = FG(S.base_ang_vel, i);
            grav[i] = FG(S.projected_gravity, i);
            eul[i] = FG(S.base_euler, i);
        }
    }
    if (mode == MODE_STEP) {
      if (phase != 2) {
        ep += 1;                                                     // legged_robot.py:128
        // derived state :132-136
        const float gvec[3] = {0.0f, 0.0f, -1.0f};
        quat_rotate_inverse(root + 3, root + 7, blv);
        quat_rotate_inverse(root + 3, root + 10, bav);
        quat_rotate_inverse(root + 3, gvec, grav);
        euler_xyz_wrapped(root + 3, eul);
        // _post_physics_step_callback :304-320
        if (ep % c.resample_steps == 0) {
            const float u[3] = {nz_uniform(A.noise.u_cmd, 6, 0, rk, e, ge, SLOT_DELAY_CMD, 1),
                                nz_uniform(A.noise.u_cmd, 6, 1, rk, e, ge, SLOT_DELAY_CMD, 2),
                                nz_uniform(A.noise.u_cmd, 6, 2, rk, e, ge, SLOT_DELAY_CMD, 3)};
            float x_lo, x_span;
            cmd_x_range<kGeneric>(A, x_lo, x_span);
            resample_commands(c, x_lo, x_span, cmd, u, !kGeneric || c.heading_command);
        }
        if (!kGeneric || c.heading_command) {
            const float fv[3] = {1.0f, 0.0f, 0.0f};
            float fw[3];
            quat_apply(root + 3, fv, fw);
            const float heading = atan2f(fw[1], fw[0]);
            cmd[2] = clampf(0.5f * wrap_like_reference(cmd[3] - heading), -1.0f, 1.0f);
        }
        if (kGeneric && c.num_height_points > 0 && S.height_pose) {   // :316-317 samples the terrain HERE, on the pre-reset base pose
#pragma unroll
            for (int i = 0; i < 7; ++i) S.height_pose[(int64_t)ge * 7 + i] = root[i];
        }
        if (c.push_robots && (csc % c.push_interval == 0)) {          // humanoid_env.py:83-98
            const float px = c.push_vel_span * nz_uniform(A.noise.u_push, 5, 0, rk, e, ge, SLOT_PUSH, 0) + c.push_vel_lo;
            const float py = c.push_vel_span * nz_uniform(A.noise.u_push, 5, 1, rk, e, ge, SLOT_PUSH, 1) + c.push_vel_lo;
            if (kMain) {
                FG(S.push_force, 0) = px;
                FG(S.push_force, 1) = py;
            }
            pf[0] = px;
            pf[1] = py;
            root[7] = px;
            root[8] = py;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float t = c.push_ang_span * nz_uniform(A.noise.u_push, 5, 2 + i, rk, e, ge, SLOT_PUSH, 2 + i) + c.push_ang_lo;
                if (kMain) FG(S.push_torque, i) = t;
                pt[i] = t;
                root[10 + i] = t;
            }
            if (kMain) {
                sset(A.sim.root, e, 7, root[7]);
                sset(A.sim.root, e, 8, root[8]);
                sset(A.sim.root, e, 10, root[10]);
                sset(A.sim.root, e, 11, root[11]);
                sset(A.sim.root, e, 12, root[12]);
            }
        }
      }
        // check_termination :156-161
        {
            const float bx = bxyz[0], by = bxyz[1], bz = bxyz[2];
            const float bn = sqrtf(bx * bx + by * by + bz * bz);
            time_out = ep > (int64_t)c.max_episode_length;
            reset = (bn > 1.0f) || time_out;
          if (phase != 1 && ROLE != ROLE_MAIN && ROLE != ROLE_FRAMES) {
            // ---------------- compute_reward :217-235, 22 terms in alphabetical order ----------------
            const float s = sinf(kTwoPi * gait_phase(c, ep));
            float stance[2];
            stance_from_sin(s, stance);
            float term[HGYM_NUM_REWARDS];
            // 0 action_smoothness :530-540
            {
                float t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
                if (kSplit) {
                    t1 = jsum[0]; t2 = jsum[1]; t3 = jsum[2];
                } else {
#pragma unroll
                    for (int j = 0; j < 12; ++j) {
                        float jt[kJointTerms];
                        joint_terms(c, joint_consts(c), j, act[j], la[j], lla[j], ldv[j], q[j], qd[j], tqv[j], rdp[j], jt);
                        t1 += jt[0];
                        t2 += jt[1];
                        t3 += jt[2];
                    }
                }
                term[0] = t1 + t2 + 0.05f * t3;
            }
            // 1 base_acc :386-393
            {
                float a2 = 0.0f;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const float d = lrv[i] - root[7 + i];
                    a2 += d * d;
                }
                term[1] = r_exp(-r_sqrt(a2) * 3.0f);
            }
            // 2 base_height :374-384
            {
                const float mh = (fpos[0][2] * stance[0] + fpos[1][2] * stance[1]) / (stance[0] + stance[1]);
                const float bh = root[2] - (mh - 0.05f);
                term[2] = r_exp(-fabsf(bh - c.base_height_target) * 100.0f);
            }
            // 3 collision :523-528
            term[3] = (bn > 0.1f) ? 1.0f : 0.0f;
            // 4 default_joint_pos :362-372
            {
                float jd[12], all2 = 0.0f;
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    jd[j] = q[j] - c.default_dof_pos[j];
                    if (!kSplit) all2 += jd[j] * jd[j];
                }
                if (kSplit) all2 = jsum[3];
                float yr = r_sqrt(jd[0] * jd[0] + jd[1] * jd[1]) + r_sqrt(jd[6] * jd[6] + jd[7] * jd[7]);
                yr = clampf(yr - 0.1f, 0.0f, 50.0f);
                term[4] = r_exp(-yr * 100.0f) - 0.01f * r_sqrt(all2);
            }
            // 5 dof_acc :516-521 ; 6 dof_vel :509-514 ; 17 torques :502-507
            {
                float acc = 0.0f, vel = 0.0f, tq = 0.0f;
                if (kSplit) {
                    acc = jsum[4]; vel = jsum[5]; tq = jsum[6];
                } else {
#pragma unroll
                    for (int j = 0; j < 12; ++j) {
                        const float a = (ldv[j] - qd[j]) / c.dt;
                        acc += a * a;
                        vel += qd[j] * qd[j];
                        const float t = tqv[j];
                        tq += t * t;
                    }
                }
                term[5] = acc;
                term[6] = vel;
                term[17] = tq;
            }
            // 7 feet_air_time :320-334 (stateful) ; 8 feet_clearance :446-467 (stateful)
            {
                float r7 = 0.0f, r8 = 0.0f;
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const float lc = lcon[f];
                    const int filt = (contact[f] > 0.5f) || (stance[f] > 0.5f) || (lc > 0.5f);
                    if (kFeet) FG(S.last_contacts, f) = contact[f];
                    float air = fat[f];
                    const int first = (air > 0.0f) && filt;
                    air += c.dt;
                    r7 += clampf(air, 0.0f, 0.5f) * (first ? 1.0f : 0.0f);
                    // (ROLE_REW_B: reset_idx's feet_air_time = 0 lands here, ROLE_MAIN does not touch the field)
                    if (kFeet) FG(S.feet_air_time, f) = (ROLE == ROLE_REW_B && reset) ? 0.0f : air * (filt ? 0.0f : 1.0f);

                    const float z = fpos[f][2] - 0.05f;
                    float fh = fhs[f] + (z - lfz[f]);
                    if (kFeet) FG(S.last_feet_z, f) = z;
                    const float swing = 1.0f - stance[f];
                    r8 += ((fabsf(fh - c.target_feet_height) < 0.01f) ? 1.0f : 0.0f) * swing;
                    if (kFeet) FG(S.feet_height, f) = fh * (contact[f] > 0.5f ? 0.0f : 1.0f);
                }
                term[7] = r7;
                term[8] = r8;
            }
            // 9 feet_contact_forces :355-360 ; 10 feet_contact_number :336-344 ; 12 foot_slip :308-318
            {
                float r9 = 0.0f, r10 = 0.0f, r12 = 0.0f;
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const float fn = r_sqrt(fxyz[f][0] * fxyz[f][0] + fxyz[f][1] * fxyz[f][1] + fxyz[f][2] * fxyz[f][2]);
                    r9 += clampf(fn - c.max_contact_force, 0.0f, 400.0f);
                    r10 += (contact[f] == stance[f]) ? 1.0f : -0.3f;
                    r12 += r_sqrt(r_sqrt(fvxy[f][0] * fvxy[f][0] + fvxy[f][1] * fvxy[f][1])) * contact[f];
                }
                term[9] = r9;
                term[10] = r10 / 2.0f;
                term[12] = r12;
            }
            // 11 feet_distance :282-292 ; 14 knee_distance :295-305
            term[11] = dist_reward(c, fpos[0][0], fpos[0][1], fpos[1][0], fpos[1][1], c.max_dist);
            term[14] = dist_reward(c, kxy[0][0], kxy[0][1], kxy[1][0], kxy[1][1], c.max_dist / 2.0f);
            // 13 joint_pos :272-280 -- the PREVIOUS step's reference pose
            {
                float e2 = 0.0f;
                if (kSplit) {
                    e2 = jsum[7];
                } else {
#pragma unroll
                    for (int j = 0; j < 12; ++j) {
                        const float d = q[j] - rdp[j];
                        e2 += d * d;
                    }
                }
                const float en = r_sqrt(e2);
                term[13] = r_exp(-2.0f * en) - 0.2f * clampf(en, 0.0f, 0.5f);
            }
            // 15 low_speed :469-500
            {
                const float vx = blv[0], cx = cmd[0];
                const float av = fabsf(vx), ac = fabsf(cx);
                const int low = av < 0.5f * ac, high = av > 1.2f * ac;
                const float sv = (vx > 0.0f) ? 1.0f : ((vx < 0.0f) ? -1.0f : 0.0f);
                const float sc = (cx > 0.0f) ? 1.0f : ((cx < 0.0f) ? -1.0f : 0.0f);
                float r = 0.0f;
                if (low) r = -1.0f;
                if (high) r = 0.0f;
                if (!(low || high)) r = 1.2f;
                if (sv != sc) r = -2.0f;
                term[15] = r * ((ac > 0.1f) ? 1.0f : 0.0f);
            }
            // 16 orientation :346-353
            term[16] = (r_exp(-(fabsf(eul[0]) + fabsf(eul[1])) * 10.0f) +
                        r_exp(-r_sqrt(grav[0] * grav[0] + grav[1] * grav[1]) * 20.0f)) / 2.0f;
            // 18 track_vel_hard :408-425 ; 19 tracking_ang_vel :436-444 ; 20 tracking_lin_vel :427-434
            {
                const float ex = cmd[0] - blv[0], ey = cmd[1] - blv[1];
                const float le2 = ex * ex + ey * ey;
                const float le = r_sqrt(le2);
                const float ae = fabsf(cmd[2] - bav[2]);
                term[18] = (r_exp(-le * 10.0f) + r_exp(-ae * 10.0f)) / 2.0f - 0.2f * (le + ae);
                const float d = cmd[2] - bav[2];
                term[19] = r_exp(-(d * d) * c.tracking_sigma);
                term[20] = r_exp(-le2 * c.tracking_sigma);
            }
            // 21 vel_mismatch_exp :396-406
            term[21] = (r_exp(-(blv[2] * blv[2]) * 10.0f) + r_exp(-r_sqrt(bav[0] * bav[0] + bav[1] * bav[1]) * 5.0f)) / 2.0f;

            // user-defined terms (legged_robot.py:518-541 discovers `_reward_<name>` by name; the caller evaluated them between the
            // derive and the finish launch, already times scale * dt): merged into the sum at their place in the alphabetical order
            // -- custom term j comes right before built-in term custom_reward_pos[j] (22: after all of them; 23: after the clip)
            if (ROLE == ROLE_REW_A || ROLE == ROLE_REW_B) {     // this wavefront's terms, times their scale: env_step_reward_sum adds them up
#pragma unroll
                for (int k = 0; k < HGYM_NUM_REWARDS; ++k)
                    if (term_in_role(k, ROLE)) tscr[k * N + e] = term[k] * c.reward_scales[k];
                return fl;
            }
            const int ncust = (kGeneric && phase == 2) ? c.num_custom_rewards : 0;
            auto add_custom = [&](int k) {
                for (int j = 0; j < ncust; ++j)
                    if (c.custom_reward_pos[j] == k) {
                        const float t = S.custom_rew[(int64_t)j * c.num_envs + ge];
                        rew += t;
                        S.custom_sums[(int64_t)j * c.num_envs + ge] += t;
                    }
            };
#pragma unroll
            for (int k = 0; k < HGYM_NUM_REWARDS; ++k) {
                if (kGeneric && ncust > 0) add_custom(k);
                const float t = term[k] * c.reward_scales[k];
                rew += t;
                esum[k] += t;
                FG(S.episode_sums, k) = esum[k];
            }
            if (kGeneric && ncust > 0) add_custom(HGYM_NUM_REWARDS);
            if (c.only_positive_rewards) rew = fmaxf(rew, 0.0f);
            // "termination" is not in the reference's function list: it is added AFTER the clip (legged_robot.py:229-235, :533-534);
            // a user-defined `_reward_termination` arrives with custom_reward_pos = HGYM_NUM_REWARDS + 1
            if (kGeneric && ncust > 0) add_custom(HGYM_NUM_REWARDS + 1);
          }
        }
    } else {
        // PRIME / RESET_ALL: reset_idx(all); derived velocities keep their current values
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            blv[i] = FG(S.base_lin_vel, i);
            bav[i] = FG(S.base_ang_vel, i);
            grav[i] = FG(S.projected_gravity, i);
            eul[i] = FG(S.base_euler, i);
        }
        if (mode == MODE_PRIME) {   // _init_buffers (legged_robot.py:478-480) on the initial sim state
            const float gvec[3] = {0.0f, 0.0f, -1.0f};
            quat_rotate_inverse(root + 3, root + 7, blv);
            quat_rotate_inverse(root + 3, root + 10, bav);
            quat_rotate_inverse(root + 3, gvec, grav);
            FG(S.last_feet_z, 0) = 0.05f;  // humanoid_env.py:78 (python scalar, broadcast)
            FG(S.last_feet_z, 1) = 0.05f;
        }
        reset = 1;
    }

    if (kGeneric && mode == MODE_STEP && phase == 1) {
        // derive launch: leave what compute_reward (and the caller's reward terms) will read, touch nothing else
        S.episode_length[e] = ep;
#pragma unroll
        for (int i = 0; i < 4; ++i) FG(S.commands, i) = cmd[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            FG(S.base_lin_vel, i) = blv[i];
            FG(S.base_ang_vel, i) = bav[i];
            FG(S.projected_gravity, i) = grav[i];
            FG(S.base_euler, i) = eul[i];
        }
        A.out.reset[e] = (uint8_t)reset;
        A.out.time_out[e] = (uint8_t)time_out;
        return fl;
    }
    // ---------------- reset_idx :163-215 (+ humanoid_env.py:264-269), mask-driven ----------------
    if (reset) {      // (ROLE_FRAMES: only the values its frames read -- commands, episode length, euler angles)
        fl.reset = 1;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            if (kSplit) break;        // env_step_phase_f, one (env, joint) pair per lane
            q[j] = c.default_dof_pos[j] + (c.dof_reset_span * nz_uniform(A.noise.u_dof, 12, j, rk, e, ge, SLOT_DOF, j) + c.dof_reset_lo);
            qd[j] = 0.0f;
            sset(A.sim.dof_pos, e, j, q[j]);
            sset(A.sim.dof_vel, e, j, 0.0f);
            act[j] = 0.0f;
            FG(S.last_actions, j) = 0.0f;
            FG(S.last_last_actions, j) = 0.0f;
            FG(S.last_dof_vel, j) = 0.0f;
        }
        if (kGeneric && c.terrain_curriculum && S.terrain_levels) {
            // _update_terrain_curriculum legged_robot.py:400-420 (runs in every reset_idx once init_done, i.e. in all three modes):
            // on the pre-reset base position and the commands this step ends with
            const float dx = root[0] - org[0], dy = root[1] - org[1];
            const float distance = sqrtf(dx * dx + dy * dy);
            const bool up = distance > c.terrain_env_length / 2.0f;
            const float cn = sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]);
            const bool down = (distance < cn * c.episode_length_s * 0.5f) && !up;
            int64_t lv = S.terrain_levels[ge] + (up ? 1 : 0) - (down ? 1 : 0);
            if (lv >= c.terrain_rows) {      // solved the last level: a random one (torch.randint_like)
                if (A.noise.r_level) lv = A.noise.r_level[ge];
                else {
                    const int r = (int)(uniform_at(rk, (uint32_t)ge, SLOT_TERRAIN, 2) * (float)c.terrain_rows);
                    lv = r < c.terrain_rows ? r : c.terrain_rows - 1;
                }
            } else if (lv < 0) lv = 0;
            S.terrain_levels[ge] = lv;
            const float* o = S.terrain_origins + ((int64_t)lv * c.terrain_cols + S.terrain_types[ge]) * 3;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                FG(S.env_origins, i) = o[i];
                org[i] = o[i];
                A.origins_hbm[(int64_t)i * c.num_envs + ge] = o[i];      // env_origins is a read-only field of the staged state
            }
        }
#pragma unroll
        for (int i = 0; i < 13; ++i) root[i] = c.base_init_state[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) root[i] += org[i];
        if (kGeneric && c.custom_origins) {  // xy within 1 m of the tile centre, legged_robot.py:382-385
            root[0] += 2.0f * (A.noise.u_xy ? A.noise.u_xy[(int64_t)ge * 2 + 0] : uniform_at(rk, (uint32_t)ge, SLOT_TERRAIN, 0)) + -1.0f;
            root[1] += 2.0f * (A.noise.u_xy ? A.noise.u_xy[(int64_t)ge * 2 + 1] : uniform_at(rk, (uint32_t)ge, SLOT_TERRAIN, 1)) + -1.0f;
        }
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            if (kMain) sset(A.sim.root, e, i, root[i]);
        }
        {
            const float u[3] = {nz_uniform(A.noise.u_cmd, 6, 3, rk, e, ge, SLOT_CMD_RESET, 0),
                                nz_uniform(A.noise.u_cmd, 6, 4, rk, e, ge, SLOT_CMD_RESET, 1),
                                nz_uniform(A.noise.u_cmd, 6, 5, rk, e, ge, SLOT_CMD_RESET, 2)};
            float x_lo, x_span;
            cmd_x_range<kGeneric>(A, x_lo, x_span);       // a command-curriculum move this step is applied by command_curriculum_fix
            resample_commands(c, x_lo, x_span, cmd, u, !kGeneric || c.heading_command);
        }
        if (ROLE == ROLE_ALL) {        // (chain by roles: ROLE_REW_B zeroes the field it owns)
            FG(S.feet_air_time, 0) = 0.0f;
            FG(S.feet_air_time, 1) = 0.0f;
        }
        ep = 0;
        // extras["episode"]: mean over resetting envs, finished by the step finaliser
        if (kMain) hg_atomic_inc(A.reset_count ? A.reset_count : &S.counters[1]);
#pragma unroll
        for (int k = 0; k < HGYM_NUM_REWARDS; ++k) {
            if (ROLE != ROLE_ALL) break;          // env_step_reward_sum, once the terms of this step are in
            hg_atomic_add(&S.episode_acc[k], esum[k]);
            FG(S.episode_sums, k) = 0.0f;
        }
        if (kGeneric && c.num_custom_rewards > 0 && S.custom_sums) {
            for (int j = 0; j < c.num_custom_rewards; ++j) {
                float* cs = S.custom_sums + (int64_t)j * c.num_envs + ge;
                hg_atomic_add(&S.custom_acc[j], *cs);
                *cs = 0.0f;
            }
        }
        if (kSplit) {       // the reset orientation is cfg.base_init_state's for every env: evaluated once per block (env_step_phase_j)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                eul[i] = reset_pose[i];
                grav[i] = reset_pose[3 + i];
            }
        } else {
            euler_xyz_wrapped(root + 3, eul);
            const float gvec[3] = {0.0f, 0.0f, -1.0f};
            quat_rotate_inverse(root + 3, gvec, grav);
        }
    }

    if (mode != MODE_RESET_ALL && kFrames) {
        // ---------------- compute_observations humanoid_env.py:200-244 (clean frames) ----------------
        const float phase = gait_phase(c, ep);
        const float s = sinf(kTwoPi * phase);
        const float co = cosf(kTwoPi * phase);
        float stance[2];
        stance_from_sin(s, stance);
        const float sl = (s > 0.0f) ? 0.0f : s, sr = (s < 0.0f) ? 0.0f : s;
        const float s1 = c.target_joint_pos_scale, s2 = 2.0f * c.target_joint_pos_scale;
        float ref[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) ref[j] = 0.0f;
        ref[2] = sl * s1; ref[3] = sl * s2; ref[4] = sl * s1;
        ref[8] = sr * s1; ref[9] = sr * s2; ref[10] = sr * s1;
        if (fabsf(s) < 0.1f) {
#pragma unroll
            for (int j = 0; j < 12; ++j) ref[j] = 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            if (!kSplit) FG(S.ref_dof_pos, j) = ref[j];
        }
        if (kSplit) cscal[e] = s;     // the gait-clock sine the per-joint lanes rebuild the reference pose from
        float ci[5] = {s, co, cmd[0] * c.scale_lin_vel, cmd[1] * c.scale_lin_vel, cmd[2] * c.scale_ang_vel};
#pragma unroll
        for (int i = 0; i < 5; ++i) { frame47[i] = ci[i]; priv73[i] = ci[i]; }
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            if (kSplit) break;
            const float qq = (q[j] - c.default_dof_pos[j]) * c.scale_dof_pos;
            const float dq = qd[j] * c.scale_dof_vel;
            frame47[5 + j] = qq;   priv73[5 + j] = qq;
            frame47[17 + j] = dq;  priv73[17 + j] = dq;
            frame47[29 + j] = act[j]; priv73[29 + j] = act[j];
            priv73[41 + j] = q[j] - ref[j];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            frame47[41 + i] = bav[i] * c.scale_ang_vel;
            frame47[44 + i] = eul[i] * c.scale_quat;
            priv73[53 + i] = blv[i] * c.scale_lin_vel;
            priv73[56 + i] = bav[i] * c.scale_ang_vel;
            priv73[59 + i] = eul[i] * c.scale_quat;
            priv73[64 + i] = pt[i];
        }
        priv73[62] = pf[0];
        priv73[63] = pf[1];
        priv73[67] = friction0;
        priv73[68] = body_mass0 / 30.0f;
        priv73[69] = stance[0];
        priv73[70] = stance[1];
        priv73[71] = contact[0];
        priv73[72] = contact[1];
    }

    if (ROLE == ROLE_FRAMES) return fl;
    // ---------------- write-back + tail of post_physics_step :147-151 ----------------
    S.episode_length[e] = ep;
#pragma unroll
    for (int i = 0; i < 4; ++i) FG(S.commands, i) = cmd[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        FG(S.base_lin_vel, i) = blv[i];
        FG(S.base_ang_vel, i) = bav[i];
        FG(S.projected_gravity, i) = grav[i];
        FG(S.base_euler, i) = eul[i];
    }
    if (mode == MODE_STEP) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            if (kSplit) break;
            FG(S.last_last_actions, j) = reset ? 0.0f : la[j];
            FG(S.last_actions, j) = act[j];
            FG(S.last_dof_vel, j) = qd[j];
            FG(S.actions, j) = act[j];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) FG(S.last_root_vel, i) = root[7 + i];
        if (ROLE == ROLE_ALL) A.out.rew[e] = rew;
        A.out.reset[e] = (uint8_t)reset;
        A.out.time_out[e] = (uint8_t)time_out;
    } else {
#pragma unroll
        for (int j = 0; j < 12; ++j) FG(S.actions, j) = 0.0f;
        A.out.reset[e] = 1;
    }
    return fl;
}

// ------------------------------------------------------------------------------------------------ workgroup phases
// The step kernel runs, per workgroup of E envs (256 lanes):
//   stage-in   all lanes: every per-env input (state, the needed sim components, actions, noise rows) is copied into
//              LDS with coalesced 16-byte loads -- ONE global-memory round trip instead of ~250 dependent ones;
//   draws      all lanes: every random number the step may consume that was not supplied as a table is produced here,
//              one Philox call per work item, into the same LDS tables the parity mode fills from HBM -- the arithmetic
//              phases below never run a Philox round;
//   joints     one lane per (env, joint): action filter + the synthetic joint integration (fused backend only);
//   per-env    one lane per env: everything that is a scalar chain per env (quaternions, commands, termination, the 22
//              reward terms, reset, clean observation frames), executed against an LDS "shadow" of EnvArgs (same code,
//              pointers re-aimed at LDS, component stride E instead of N);
//   stage-out  all lanes: modified state / sim components / step outputs written back, coalesced;
//   stacking   all lanes: observation history -> the row-major outputs, 16 bytes per lane.
// Every phase is a plain function of (block, thread) so tests/hostcheck runs the identical code on the host.
HG_HD RngKey make_rng_key(const EnvArgs& A, int64_t csc0) {
    RngKey rk;
    rk.k0 = (uint32_t)A.cfg.seed;
    rk.k1 = (uint32_t)(A.cfg.seed >> 32);
    rk.s0 = (uint32_t)csc0;
    rk.s1 = (uint32_t)(csc0 >> 32) ^ (A.mode == MODE_STEP ? 0u : 0x80000000u);
    return rk;
}

struct __attribute__((packed, aligned(4))) EnvF4 {
    float v[4];
};
// The observation history (14 + 2 older frames per env read, the stacked rows and the ring slot written) is touched once per
// step: with the non-temporal hint it does not push the policy's weight fragments -- which every actor / critic tile of the XCD
// re-reads at the next launch -- out of the 4 MB L2.
constexpr int kEnvNT = 7;      // bit 0: loads of the older frames, bit 1: their stores into the stacked rows, bit 2: newest frame (ring slot + row)
typedef float envf4_nt __attribute__((ext_vector_type(4), aligned(4)));
template <bool NT = true>
HG_HD EnvF4 ld_stream4(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (NT) {
        const envf4_nt v = __builtin_nontemporal_load(reinterpret_cast<const envf4_nt*>(p));
        EnvF4 q = {{v[0], v[1], v[2], v[3]}};
        return q;
    }
#endif
    return *reinterpret_cast<const EnvF4*>(p);
}
template <bool NT = true>
HG_HD void st_stream4(float* p, const EnvF4& q) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (NT) {
        const envf4_nt v = {q.v[0], q.v[1], q.v[2], q.v[3]};
        __builtin_nontemporal_store(v, reinterpret_cast<envf4_nt*>(p));
        return;
    }
#endif
    *reinterpret_cast<EnvF4*>(p) = q;
}

// components of every [C][N] state field, in HgymEnvState order (commands ... env_origins)
constexpr int kNumStateFields = 22;
HG_HD int state_field_comps(int f) {
    constexpr int c[kNumStateFields] = {4, 12, 12, 12, 12, 6, 12, 2, 2, 2, 2, 12, 3, 3, HGYM_NUM_REWARDS, 3, 3, 3, 3, 1, 1, 3};
    return c[f];
}
constexpr int kStateComps = 4 + 12 * 6 + 6 + 2 * 4 + 3 * 2 + HGYM_NUM_REWARDS + 3 * 4 + 1 + 1 + 3;   // 136
constexpr int kFirstConstField = 19;    // friction, body_mass, env_origins are read-only
constexpr int kMutableComps = kStateComps - 5;
HG_HD float** state_field_ptr(HgymEnvState& S, int f) { return (&S.commands) + f; }
HG_HD float* const* state_field_ptr(const HgymEnvState& S, int f) { return (&S.commands) + f; }
// flat state component index (0..135) -> its [N] row in HBM
HG_HD float* state_comp_row(const EnvArgs& A, int comp, int N) {
    const HgymEnvState& S = A.st;
    if (A.state_contig) return S.commands + (int64_t)comp * N;   // one [136][N] allocation (hgym.EnvBuffers)
    int f = 0;
    while (comp >= state_field_comps(f)) {
        comp -= state_field_comps(f);
        ++f;
    }
    return *state_field_ptr(S, f) + (int64_t)comp * N;
}

// snapshot rows of the chain by roles: root state 0..12, commands 13..16, last root velocity 17..22, (23 unused), episode length
// (int64[E]) from row 24
constexpr int kSnapRoot = 0, kSnapCmd = 13, kSnapLrv = 17, kSnapEp = 24, kSnapComps = 26;
// LDS carve (float offsets) for a block of E envs
struct LdsMap {
    int state, root, dof_pos, dof_vel, contact, rigid, actions_in, u_delay, z_act, u_cmd, u_dof, u_push, z_obs, phys, frame, priv, rew,
        noise_vec, jcfg, ep_len, flags, reset_i, reset_cnt, reset_list, reset_pose, jpart, cscal, terms, snap, total;
};
HG_HD LdsMap lds_map(int E) {
    LdsMap m;
    int o = 0;
    m.state = o;      o += kStateComps * E;
    m.root = o;       o += 13 * E;
    m.dof_pos = o;    o += 12 * E;
    m.dof_vel = o;    o += 12 * E;
    m.contact = o;    o += 9 * E;
    m.rigid = o;      o += 52 * E;
    m.actions_in = o; o += 12 * E;
    m.u_delay = o;    o += E;
    m.z_act = o;      o += 12 * E;
    m.u_cmd = o;      o += 8 * E;            // 6 used (row width 6), sized for whole Philox calls
    m.u_dof = o;      o += 12 * E;
    m.u_push = o;     o += 8 * E;            // 5 used (row width 5)
    m.z_obs = o;      o += 48 * E;           // 47 used (row width 47)
    m.phys = o;       o += kPhysDraws * E;
    m.frame = o;      o += HGYM_OBS_FRAME * E;
    m.priv = o;       o += HGYM_PRIV_FRAME * E;
    m.rew = o;        o += E;
    m.noise_vec = o;  o += 48;             // obs_noise[47] (a lane-varying index must not go through the kernel argument)
    m.jcfg = o;       o += kJointConsts;   // the six per-joint constant arrays (joint_consts), for the same reason
    o = (o + 1) & ~1;
    m.ep_len = o;     o += 2 * E;          // int64[E]
    m.flags = o;      o += (2 * E + 3) / 4;  // uint8 reset[E], time_out[E]
    m.reset_i = o;    o += E;              // int[E]: "history must be cleared" flags for the stacking phase
    m.reset_cnt = o;  o += 1;              // int: how many of them are set (zeroed by the stage-in, counted by the per-env phase)
    m.reset_list = o; o += E;              // int[reset_cnt]: the local ids of the envs that reset, in arrival order
    m.reset_pose = o; o += 8;              // euler angles [0..2] and projected gravity [3..5] of the reset orientation (split chain)
    m.jpart = o;      o += kJointTerms * 12 * E;   // [8][12][E] per-joint reward products (split per-env chain)
    m.cscal = o;      o += E;              // [E] gait-clock sine of the new observation (split per-env chain -> per-joint lanes)
    m.terms = o;      o += HGYM_NUM_REWARDS * E;   // [22][E] reward terms times their scale (chain by roles -> env_step_reward_sum)
    o = (o + 1) & ~1;
    m.snap = o;       o += kSnapComps * E;   // the reward wavefronts' inputs that ROLE_MAIN rewrites during the phase (env_snapshot)
    m.total = o;
    return m;
}
inline size_t step_smem_bytes(int E) { return (size_t)lds_map(E).total * sizeof(float); }

constexpr int kFootRigidComps[5] = {0, 1, 2, 7, 8};
constexpr int kKneeRigidComps[2] = {0, 1};

// The LDS shadow of EnvArgs for block `block`: same configuration, pointers into smem, component stride E.
// Every noise table points into LDS: it was either staged from the caller's table or filled by env_fill_draws.
HG_HD EnvArgs make_shadow(const EnvArgs& A, float* smem, int block, int E) {
    const LdsMap m = lds_map(E);
    EnvArgs S = A;
    S.env_base = block * E;
    int o = m.state;
    for (int f = 0; f < kNumStateFields; ++f) {
        *state_field_ptr(S.st, f) = smem + o;
        o += state_field_comps(f) * E;
    }
    S.st.episode_length = reinterpret_cast<int64_t*>(smem + m.ep_len);
    S.sim.root = HgymStrided{smem + m.root, 1, E};
    S.sim.dof_pos = HgymStrided{smem + m.dof_pos, 1, E};
    S.sim.dof_vel = HgymStrided{smem + m.dof_vel, 1, E};
    S.sim.contact = HgymStrided{smem + m.contact, 1, E};
    S.sim.rigid = HgymStrided{smem + m.rigid, 1, E};
    S.contact_comp[0] = 0; S.contact_comp[1] = 3; S.contact_comp[2] = 6;
    S.rigid_comp[0] = 0; S.rigid_comp[1] = 13; S.rigid_comp[2] = 26; S.rigid_comp[3] = 39;
    if (A.actions_in) S.actions_in = smem + m.actions_in;
    S.noise.u_delay = smem + m.u_delay;
    S.noise.z_act = smem + m.z_act;
    S.noise.u_cmd = smem + m.u_cmd;
    S.noise.u_dof = smem + m.u_dof;
    S.noise.u_push = smem + m.u_push;
    S.noise.z_obs = smem + m.z_obs;
    S.out.rew = smem + m.rew;
    S.out.reset = reinterpret_cast<uint8_t*>(smem + m.flags);
    S.out.time_out = S.out.reset + E;
    return S;
}

HG_HD void copy_rows_in(const float* g, float* l, int width, int e0, int nE, int t, int nthreads) {   // (N,width) row-major table
    if (!g) return;
    const float* src = g + (int64_t)e0 * width;
    for (int i = t; i < nE * width; i += nthreads) l[i] = src[i];
}

// [C][N] rows <-> [C][E] LDS rows, 4 envs (16 bytes) per lane; the partial last block goes element-wise
template <bool kIn>
HG_HD void copy_comp_rows(float* const* rows_unused, const EnvArgs& A, int first_comp, int ncomp, float* l, int E, int e0, int nE, int N,
                          int t, int nthreads) {
    (void)rows_unused;
    if (nE == E && (E & 3) == 0) {
        const int qpr = E >> 2;
        for (int i = t; i < ncomp * qpr; i += nthreads) {
            const int c = i / qpr, qd = i - c * qpr;
            float* g = state_comp_row(A, first_comp + c, N) + e0 + 4 * qd;
            EnvF4* lp = reinterpret_cast<EnvF4*>(l + c * E + 4 * qd);
            if (kIn) *lp = *reinterpret_cast<const EnvF4*>(g);
            else *reinterpret_cast<EnvF4*>(g) = *lp;
        }
    } else {
        for (int i = t; i < ncomp * E; i += nthreads) {
            const int c = i / E, le = i - c * E;
            float* g = state_comp_row(A, first_comp + c, N) + e0 + le;
            if (kIn) l[i] = (le < nE) ? *g : 0.0f;
            else if (le < nE) *g = l[i];
        }
    }
}

// one strided sim tensor (ncomp components listed by comp_of(k)) <-> LDS [k][E]
template <bool kIn, typename CompOf>
HG_HD void copy_sim_rows(const HgymStrided& sv, int ncomp, CompOf comp_of, float* l, int lstride_comp_of_unused, int E, int e0, int nE, int t,
                         int nthreads) {
    (void)lstride_comp_of_unused;
    if (sv.env_stride == 1 && nE == E && (E & 3) == 0) {
        const int qpr = E >> 2;
        for (int i = t; i < ncomp * qpr; i += nthreads) {
            const int k = i / qpr, qd = i - k * qpr;
            float* g = sv.base + (int64_t)comp_of(k).x * sv.comp_stride + e0 + 4 * qd;
            EnvF4* lp = reinterpret_cast<EnvF4*>(l + comp_of(k).y * E + 4 * qd);
            if (kIn) *lp = *reinterpret_cast<const EnvF4*>(g);
            else *reinterpret_cast<EnvF4*>(g) = *lp;
        }
    } else {
        for (int i = t; i < ncomp * E; i += nthreads) {
            const int k = i / E, le = i - k * E;
            if (kIn) l[comp_of(k).y * E + le] = (le < nE) ? sget(sv, e0 + le, comp_of(k).x) : 0.0f;
            else if (le < nE) sset(sv, e0 + le, comp_of(k).x, l[comp_of(k).y * E + le]);
        }
    }
}
struct CompPair {
    int x, y;   // component in the sim tensor, row in the LDS image
};

template <bool kIn>
HG_HD void stage_sim(const EnvArgs& A, const LdsMap& m, float* smem, int E, int e0, int nE, int t, int nthreads, bool contact_rigid) {
    auto ident = [](int k) { return CompPair{k, k}; };
    copy_sim_rows<kIn>(A.sim.root, 13, ident, smem + m.root, 0, E, e0, nE, t, nthreads);
    copy_sim_rows<kIn>(A.sim.dof_pos, 12, ident, smem + m.dof_pos, 0, E, e0, nE, t, nthreads);
    copy_sim_rows<kIn>(A.sim.dof_vel, 12, ident, smem + m.dof_vel, 0, E, e0, nE, t, nthreads);
    if (!contact_rigid) return;
    const int cc0 = A.contact_comp[0], cc1 = A.contact_comp[1], cc2 = A.contact_comp[2];
    auto cmap = [=](int k) { return CompPair{(k < 3 ? cc0 : (k < 6 ? cc1 : cc2)) + k % 3, k}; };
    copy_sim_rows<kIn>(A.sim.contact, 9, cmap, smem + m.contact, 0, E, e0, nE, t, nthreads);
    const int r0 = A.rigid_comp[0], r1 = A.rigid_comp[1], r2 = A.rigid_comp[2], r3 = A.rigid_comp[3];
    auto rmap = [=](int k) {   // feet {x,y,z,vx,vy}, knees {x,y}
        const int body = k < 10 ? k / 5 : 2 + (k - 10) / 2;
        const int comp = k < 10 ? kFootRigidComps[k % 5] : kKneeRigidComps[(k - 10) % 2];
        const int base = body == 0 ? r0 : (body == 1 ? r1 : (body == 2 ? r2 : r3));
        return CompPair{base + comp, body * 13 + comp};
    };
    copy_sim_rows<kIn>(A.sim.rigid, 14, rmap, smem + m.rigid, 0, E, e0, nE, t, nthreads);
}

// Fast stage-in (contiguous state, SoA sim tensors, full block, 256 lanes): every tensor is a list of 16-byte items
// (4 envs of one component row).  Each lane issues one UNCONDITIONAL load per tensor (index clamped into range, surplus
// lanes re-read the last item) plus its share of the state rows, all before its first LDS write: the stage costs one
// memory round trip instead of one per field.  Every base pointer is a fixed struct member -- a lane-varying choice
// between tensors would make the compiler index the kernel argument dynamically and spill it to scratch.
// The loads of the fast path live in a register record so that a caller can put other independent loads (the history
// prefetch, whose addresses wait for the ring-step counter) between their issue and their LDS writes: issue first, wait last.
HG_HD void stage_ld(float (&r)[4], const float* p) {
    const EnvF4 q = *reinterpret_cast<const EnvF4*>(p);
    r[0] = q.v[0]; r[1] = q.v[1]; r[2] = q.v[2]; r[3] = q.v[3];
}
HG_HD void stage_st(float* p, const float (&r)[4]) {
    const EnvF4 q = {{r[0], r[1], r[2], r[3]}};
    *reinterpret_cast<EnvF4*>(p) = q;
}
template <int E_T>
struct StageRegs {
    static constexpr int EE = E_T > 0 ? E_T : 4;
    static constexpr int Q = EE / 4;                                 // 16-byte items per component row
    static constexpr int NS = (kStateComps * Q + 255) / 256;
    // plain float quads (as hist_load keeps its prefetch): records of the packed 16-byte struct were left in private memory by the
    // compiler, and a kernel with a private segment is dispatched differently from its neighbours
    float rs[NS][4], r_root[4], r_dp[4], r_dv[4], r_ct[4], r_rg[4], r_act[4], r_ep[4];
    int fast;
};
template <int E_T>
HG_HD bool env_stage_fast(const EnvArgs& A, int block, int nthreads) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs, e0 = block * E;
    const int nE = (E < N - e0) ? E : (N - e0);
    constexpr int Q = StageRegs<E_T>::Q;
    return E_T > 0 && (E_T & 3) == 0 && 14 * Q <= 256 && nthreads == 256 && nE == E && A.state_contig &&
           A.sim.root.env_stride == 1 && A.sim.dof_pos.env_stride == 1 && A.sim.dof_vel.env_stride == 1 &&
           A.sim.contact.env_stride == 1 && A.sim.rigid.env_stride == 1;
}
// kAssume (hgym_rollout_step, whose host side refuses every other layout: rollout_env_args): the fast layout is a fact of the launch, the general
// paths of the staging phases -- never taken there -- are not compiled into it
template <int E_T, bool kAssume = false>
HG_HD void env_stage_in_load(const EnvArgs& A, int block, int t, int nthreads, StageRegs<E_T>& R) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs, e0 = block * E;
    constexpr int EE = StageRegs<E_T>::EE, Q = StageRegs<E_T>::Q, NS = StageRegs<E_T>::NS;
    R.fast = kAssume ? 1 : (env_stage_fast<E_T>(A, block, nthreads) ? 1 : 0);
    if (!kAssume && !R.fast) return;
    auto cl = [](int i, int n) { return i < n ? i : n - 1; };
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int i = cl(t + u * 256, kStateComps * Q);
        stage_ld(R.rs[u], A.st.commands + (int64_t)(i / Q) * N + e0 + 4 * (i % Q));
    }
    const int i13 = cl(t, 13 * Q), i12 = cl(t, 12 * Q), i9 = cl(t, 9 * Q), i14 = cl(t, 14 * Q);
    stage_ld(R.r_root, A.sim.root.base + (int64_t)(i13 / Q) * A.sim.root.comp_stride + e0 + 4 * (i13 % Q));
    stage_ld(R.r_dp, A.sim.dof_pos.base + (int64_t)(i12 / Q) * A.sim.dof_pos.comp_stride + e0 + 4 * (i12 % Q));
    stage_ld(R.r_dv, A.sim.dof_vel.base + (int64_t)(i12 / Q) * A.sim.dof_vel.comp_stride + e0 + 4 * (i12 % Q));
    const int cc0 = A.contact_comp[0], cc1 = A.contact_comp[1], cc2 = A.contact_comp[2];
    const int k9 = i9 / Q;
    const int ccomp = (k9 < 3 ? cc0 : (k9 < 6 ? cc1 : cc2)) + k9 % 3;
    stage_ld(R.r_ct, A.sim.contact.base + (int64_t)ccomp * A.sim.contact.comp_stride + e0 + 4 * (i9 % Q));
    const int rc0 = A.rigid_comp[0], rc1 = A.rigid_comp[1], rc2 = A.rigid_comp[2], rc3 = A.rigid_comp[3];
    const int k14 = i14 / Q;                              // feet {x,y,z,vx,vy}, knees {x,y}
    const int body = k14 < 10 ? k14 / 5 : 2 + (k14 - 10) / 2;
    const int c5 = k14 % 5;
    const int rcomp = k14 < 10 ? (c5 < 3 ? c5 : c5 + 4) : (k14 - 10) % 2;
    const int rbase = body == 0 ? rc0 : (body == 1 ? rc1 : (body == 2 ? rc2 : rc3));
    stage_ld(R.r_rg, A.sim.rigid.base + (int64_t)(rbase + rcomp) * A.sim.rigid.comp_stride + e0 + 4 * (i14 % Q));
    const int ia = cl(t, 3 * EE), ie = cl(t, EE / 2);
    stage_ld(R.r_ep, reinterpret_cast<const float*>(A.st.episode_length + e0) + 4 * ie);
    // unconditional load (from the episode-length row when there are no actions): a conditionally initialised vector ends
    // up in scratch memory, and a kernel with a private segment is dispatched differently from its neighbours
    stage_ld(R.r_act, A.actions_in ? A.actions_in + (int64_t)e0 * 12 + 4 * ia
                                                            : reinterpret_cast<const float*>(A.st.episode_length + e0) + 4 * ie);
}
template <int E_T, bool kAssume = false>
HG_HD void env_stage_in_store(const EnvArgs& A, int block, int t, int nthreads, float* smem, const StageRegs<E_T>& R) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs, e0 = block * E;
    const int nE = kAssume ? E : ((E < N - e0) ? E : (N - e0));
    const LdsMap m = lds_map(E);
    constexpr int EE = StageRegs<E_T>::EE, Q = StageRegs<E_T>::Q, NS = StageRegs<E_T>::NS;
    if (kAssume || R.fast) {
        auto cl = [](int i, int n) { return i < n ? i : n - 1; };
        const int i13 = cl(t, 13 * Q), i12 = cl(t, 12 * Q), i9 = cl(t, 9 * Q), i14 = cl(t, 14 * Q);
        const int k9 = i9 / Q, k14 = i14 / Q;
        const int body = k14 < 10 ? k14 / 5 : 2 + (k14 - 10) / 2;
        const int c5 = k14 % 5;
        const int rcomp = k14 < 10 ? (c5 < 3 ? c5 : c5 + 4) : (k14 - 10) % 2;
        const int ia = cl(t, 3 * EE), ie = cl(t, EE / 2);
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int i = t + u * 256;
            if (i < kStateComps * Q) stage_st(smem + m.state + (i / Q) * E + 4 * (i % Q), R.rs[u]);
        }
        if (t < 13 * Q) stage_st(smem + m.root + (i13 / Q) * E + 4 * (i13 % Q), R.r_root);
        if (t < 12 * Q) {
            stage_st(smem + m.dof_pos + (i12 / Q) * E + 4 * (i12 % Q), R.r_dp);
            stage_st(smem + m.dof_vel + (i12 / Q) * E + 4 * (i12 % Q), R.r_dv);
        }
        if (t < 9 * Q) stage_st(smem + m.contact + k9 * E + 4 * (i9 % Q), R.r_ct);
        if (t < 14 * Q) stage_st(smem + m.rigid + (body * 13 + rcomp) * E + 4 * (i14 % Q), R.r_rg);
        if (A.actions_in && t < 3 * EE) stage_st(smem + m.actions_in + 4 * ia, R.r_act);
        if (t < EE / 2) stage_st(smem + m.ep_len + 4 * ie, R.r_ep);
    } else {
        copy_comp_rows<true>(nullptr, A, 0, kStateComps, smem + m.state, E, e0, nE, N, t, nthreads);
        stage_sim<true>(A, m, smem, E, e0, nE, t, nthreads, true);
        for (int i = t; i < nE; i += nthreads) reinterpret_cast<int64_t*>(smem + m.ep_len)[i] = A.st.episode_length[e0 + i];
        copy_rows_in(A.actions_in, smem + m.actions_in, 12, e0, nE, t, nthreads);
    }
    copy_rows_in(A.noise.u_delay, smem + m.u_delay, 1, e0, nE, t, nthreads);
    copy_rows_in(A.noise.z_act, smem + m.z_act, 12, e0, nE, t, nthreads);
    copy_rows_in(A.noise.u_cmd, smem + m.u_cmd, 6, e0, nE, t, nthreads);
    copy_rows_in(A.noise.u_dof, smem + m.u_dof, 12, e0, nE, t, nthreads);
    copy_rows_in(A.noise.u_push, smem + m.u_push, 5, e0, nE, t, nthreads);
    copy_rows_in(A.noise.z_obs, smem + m.z_obs, HGYM_OBS_FRAME, e0, nE, t, nthreads);
    for (int i = t; i < HGYM_OBS_FRAME; i += nthreads) smem[m.noise_vec + i] = A.cfg.obs_noise[i];
    for (int i = t; i < kJointConsts; i += nthreads) smem[m.jcfg + i] = joint_consts(A.cfg)[i];
    if (t == 0) reinterpret_cast<int*>(smem + m.reset_cnt)[0] = 0;
}
template <int E_T, bool kAssume = false>
HG_HD void env_stage_in(const EnvArgs& A, int block, int t, int nthreads, float* smem) {
    StageRegs<E_T> R;
    env_stage_in_load<E_T, kAssume>(A, block, t, nthreads, R);
    env_stage_in_store<E_T, kAssume>(A, block, t, nthreads, smem, R);
}

// Random draws not supplied by the caller, one Philox call per work item, into the LDS tables (row-major per env,
// same widths as the caller's tables).  Values are exactly uniform_at / normal_at of hgym_common.hpp.
// Work item = (call c, env): consecutive lanes take consecutive envs of the SAME call, and every call runs the same
// instruction stream (Philox, then uniforms or a Box-Muller pair), so a wavefront never serialises over call kinds.
constexpr int kDrawCalls = 1 + 1 + 3 + 3 + 2 + 12 + kPhysCalls;   // delay+cmd | cmd reset | act | dof | push | obs | phys
struct DrawCall {
    uint32_t slot;   // Philox counter word
    int normal;      // 1: two Box-Muller pairs, 0: four uniforms
    int active;      // 0: every output of this call was supplied by the caller (or is not needed in this mode)
};
HG_HD DrawCall draw_call(const EnvArgs& A, int c, bool phys) {
    DrawCall d;
    if (c == 0) d = DrawCall{SLOT_DELAY_CMD, 0, !(A.noise.u_delay && A.noise.u_cmd)};
    else if (c == 1) d = DrawCall{SLOT_CMD_RESET, 0, !A.noise.u_cmd};
    else if (c < 5) d = DrawCall{SLOT_ACT + (uint32_t)(c - 2), 1, !A.noise.z_act && A.actions_in != nullptr};
    else if (c < 8) d = DrawCall{SLOT_DOF + (uint32_t)(c - 5), 0, !A.noise.u_dof};
    else if (c < 10) d = DrawCall{SLOT_PUSH + (uint32_t)(c - 8), 0, !A.noise.u_push};
    else if (c < 22) d = DrawCall{SLOT_OBS + (uint32_t)(c - 10), 1, !A.noise.z_obs && A.cfg.add_noise && A.mode != MODE_RESET_ALL};
    else {
        const int p = c - 22;
        d = DrawCall{SLOT_PHYS + (uint32_t)p, (p >= 1 && p <= 3) || (p >= 5 && p <= 7), phys};
    }
    return d;
}
// LDS float index of output k (0..3) of call c for local env le, or -1 if that output has no consumer
HG_HD int draw_dest(const EnvArgs& A, const LdsMap& m, int c, int le, int k) {
    if (c == 0) return k == 0 ? (A.noise.u_delay ? -1 : m.u_delay + le) : (A.noise.u_cmd ? -1 : m.u_cmd + le * 6 + k - 1);
    if (c == 1) return k < 3 ? m.u_cmd + le * 6 + 3 + k : -1;
    if (c < 5) return m.z_act + le * 12 + 4 * (c - 2) + k;
    if (c < 8) return m.u_dof + le * 12 + 4 * (c - 5) + k;
    if (c < 10) return (4 * (c - 8) + k < 5) ? m.u_push + le * 5 + 4 * (c - 8) + k : -1;
    if (c < 22) return (4 * (c - 10) + k < HGYM_OBS_FRAME) ? m.z_obs + le * HGYM_OBS_FRAME + 4 * (c - 10) + k : -1;
    return m.phys + le * kPhysDraws + 4 * (c - 22) + k;
}

// euler angles / projected gravity of a freshly reset env (cfg.base_init_state's orientation: the same for all of them), evaluated
// by ONE lane per block while the block waits for its input loads, instead of inside the divergent reset branch of the per-env
// chain (about one block in ten has a resetting env in any step, and the launch ends with its slowest block)
template <int E_T>
HG_HD void env_reset_pose(const EnvArgs& A, int t, int nthreads, float* smem) {
    if (t != nthreads - 1) return;
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const LdsMap m = lds_map(E);
    float q[4], eul[3], grav[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = A.cfg.base_init_state[3 + i];
    euler_xyz_wrapped(q, eul);
    const float gvec[3] = {0.0f, 0.0f, -1.0f};
    quat_rotate_inverse(q, gvec, grav);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        smem[m.reset_pose + i] = eul[i];
        smem[m.reset_pose + 3 + i] = grav[i];
    }
}

template <int E_T>
HG_HD void env_fill_draws(const EnvArgs& A, int block, int t, int nthreads, float* smem, int64_t csc0) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs, e0 = block * E;
    const int nE = (E < N - e0) ? E : (N - e0);
    const LdsMap m = lds_map(E);
    const RngKey rk = make_rng_key(A, csc0);
    const bool phys = A.mode == MODE_STEP && A.fused;
    for (int i = t; i < E * kDrawCalls; i += nthreads) {
        const int c = i / E, le = i - c * E;
        const DrawCall d = draw_call(A, c, phys);
        if (!d.active || le >= nE) continue;
        const U4 r = rng4(rk, (uint32_t)(e0 + le), d.slot);
        float v[4];
        if (d.normal) {
            box_muller(r.x, r.y, v[0], v[1]);
            box_muller(r.z, r.w, v[2], v[3]);
        } else {
            v[0] = u01(r.x); v[1] = u01(r.y); v[2] = u01(r.z); v[3] = u01(r.w);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int dst = draw_dest(A, m, c, le, k);
            if (dst >= 0) smem[dst] = v[k];
        }
    }
}

// action filter + synthetic joint integration, one (env, joint) pair per lane (fused backend only)
constexpr int kStateOffActions = 4;     // component offsets of state fields inside the [136][E] LDS image
constexpr int kStateOffTorques = 4 + 12 * 4 + 6;
constexpr int kStateOffRefPos = kStateOffTorques + 12 + 2 * 4;     // ... torques, feet_air_time, last_contacts, feet_height, last_feet_z
template <int E_T>
HG_HD void env_step_joints(const EnvArgs& A, int block, int t, int nthreads, float* smem) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs, e0 = block * E;
    const int nE = (E < N - e0) ? E : (N - e0);
    if (!(A.mode == MODE_STEP && A.fused)) return;
    const LdsMap m = lds_map(E);
    // works on the LDS image directly (no shadow struct): the per-joint constants are indexed by a lane-varying j and
    // must be read from the kernel argument itself, not from a per-lane copy of it
    for (int i = t; i < 12 * E; i += nthreads) {
        const int j = i / E, le = i - j * E;
        if (le >= nE) continue;
        float* act = smem + m.state + (kStateOffActions + j) * E + le;
        float a_in = smem[m.actions_in + le * 12 + j];
        if (A.cfg.use_ref_actions) {      // see pre_physics_joint; the LDS row goes back to the caller's tensor in env_stage_out
            a_in = a_in + 2.0f * smem[m.state + (kStateOffRefPos + j) * E + le];
            smem[m.actions_in + le * 12 + j] = a_in;
        }
        const float a = filter_action(A.cfg, a_in, *act, smem[m.u_delay + le], smem[m.z_act + le * 12 + j]);
        *act = a;
        float q = smem[m.dof_pos + j * E + le], qd = smem[m.dof_vel + j * E + le], tq;
        integrate_joint(A.cfg, HGYM_JC(A, smem, m), j, a, q, qd, tq);
        smem[m.state + (kStateOffTorques + j) * E + le] = tq;
        smem[m.dof_pos + j * E + le] = q;
        smem[m.dof_vel + j * E + le] = qd;
    }
}

// ---- the split per-env chain (post_physics_env<.., kSplit = true>), MODE_STEP only ---------------------------------------
constexpr int kStateOffLastActions = 4 + 12;
constexpr int kStateOffLastLastActions = 4 + 12 * 2;
constexpr int kStateOffLastDofVel = 4 + 12 * 3;
constexpr int kStateOffLastRootVel = 4 + 12 * 4;
constexpr int kStateOffEpisodeSums = 4 + 12 * 4 + 6 + 12 + 2 * 4 + 12 + 3 * 2;     // = 96

// the per-joint reward products, one (env, joint) pair per lane, same item mapping as env_step_joints (a lane reads what it has
// just written there: no barrier in between): jpart[(k * 12 + j) * E + le]
template <int E_T>
HG_HD void env_step_joint_terms(const EnvArgs& A, int block, int t, int nthreads, float* smem) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs, e0 = block * E;
    const int nE = (E < N - e0) ? E : (N - e0);
    const LdsMap m = lds_map(E);
    const float* st = smem + m.state;
    for (int i = t; i < 12 * E; i += nthreads) {
        const int j = i / E, le = i - j * E;
        if (le >= nE) continue;
        float jt[kJointTerms];
        joint_terms(A.cfg, HGYM_JC(A, smem, m), j, st[(kStateOffActions + j) * E + le], st[(kStateOffLastActions + j) * E + le],
                    st[(kStateOffLastLastActions + j) * E + le], st[(kStateOffLastDofVel + j) * E + le], smem[m.dof_pos + j * E + le],
                    smem[m.dof_vel + j * E + le], st[(kStateOffTorques + j) * E + le], st[(kStateOffRefPos + j) * E + le], jt);
#pragma unroll
        for (int k = 0; k < kJointTerms; ++k) smem[m.jpart + (k * 12 + j) * E + le] = jt[k];
    }
}

// everything between the draws and the per-env chain: action filter + joint integration (fused backend), the per-joint reward
// products, and -- on the last two wavefronts, next to the joint lanes -- the two halves of the synthetic physics' per-env
// remainder (with a single wavefront, as in the host emulation, they simply follow)
// kSnap: the chain by roles follows -- the same two wavefronts leave the snapshot its reward and frame roles read (each lane its own
// env: the root lane after it has moved the root state)
template <int E_T, bool kSnap = false>
HG_HD void env_step_phase_j(const EnvArgs& A, int block, int t, int nthreads, float* smem) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs, e0 = block * E;
    const int nE = (E < N - e0) ? E : (N - e0);
    env_step_joints<E_T>(A, block, t, nthreads, smem);
    env_step_joint_terms<E_T>(A, block, t, nthreads, smem);
    const LdsMap m = lds_map(E);
    const int nw = nthreads >= 64 ? nthreads / 64 : 1;
    const bool synth = A.mode == MODE_STEP && A.fused;
    if (synth || kSnap) {
        const int w_root = nw - 1, w_feet = nw >= 2 ? nw - 2 : nw - 1;
        const int lr = t - 64 * w_root, lf = t - 64 * w_feet;
        const bool r_ok = lr >= 0 && lr < nE, f_ok = lf >= 0 && lf < nE;
        if (r_ok || f_ok) {
            if (synth) {
                const EnvArgs S = make_shadow(A, smem, block, E);
                if (r_ok) synth_root_env(S, smem + m.phys + lr * kPhysDraws, lr, E);
                if (f_ok) synth_feet_env(S, smem + m.phys + lf * kPhysDraws, lf, E);
            }
            if (kSnap) {
                float* sn = smem + m.snap;
                if (r_ok) {
#pragma unroll
                    for (int i = 0; i < 13; ++i) sn[(kSnapRoot + i) * E + lr] = smem[m.root + i * E + lr];
                }
                if (f_ok) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) sn[(kSnapCmd + i) * E + lf] = smem[m.state + i * E + lf];
#pragma unroll
                    for (int i = 0; i < 6; ++i) sn[(kSnapLrv + i) * E + lf] = smem[m.state + (kStateOffLastRootVel + i) * E + lf];
                    reinterpret_cast<int64_t*>(sn + kSnapEp * E)[lf] = reinterpret_cast<const int64_t*>(smem + m.ep_len)[lf];
                }
            }
        }
    }
}

// after the per-env chain, one (env, joint) pair per lane: the per-joint part of reset_idx (legged_robot.py:358-371 + the last_*
// buffers, :189-195), the reference pose (humanoid_env.py:100-118), the per-joint entries of the two clean observation frames
// (:200-244) and the tail copies of post_physics_step (legged_robot.py:147-151)
template <int E_T>
HG_HD void env_step_phase_f(const EnvArgs& A, int block, int t, int nthreads, float* smem) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs, e0 = block * E;
    const int nE = (E < N - e0) ? E : (N - e0);
    const LdsMap m = lds_map(E);
    const HgymEnvConfig& c = A.cfg;
    float* st = smem + m.state;
    const int* s_reset = reinterpret_cast<const int*>(smem + m.reset_i);
    for (int i = t; i < 12 * E; i += nthreads) {
        const int j = i / E, le = i - j * E;
        if (le >= nE) continue;
        const int reset = s_reset[le];
        const float s = smem[m.cscal + le];
        float q = smem[m.dof_pos + j * E + le], qd = smem[m.dof_vel + j * E + le];
        float act = st[(kStateOffActions + j) * E + le];
        const float la = st[(kStateOffLastActions + j) * E + le];
        const float dflt = HGYM_JC(A, smem, m)[kJcDef + j];
        if (reset) {
            q = dflt + (c.dof_reset_span * smem[m.u_dof + le * 12 + j] + c.dof_reset_lo);
            qd = 0.0f;
            act = 0.0f;
            smem[m.dof_pos + j * E + le] = q;
            smem[m.dof_vel + j * E + le] = 0.0f;
        }
        const float sl = (s > 0.0f) ? 0.0f : s, sr = (s < 0.0f) ? 0.0f : s;
        const float s1 = c.target_joint_pos_scale, s2 = 2.0f * c.target_joint_pos_scale;
        float ref = 0.0f;
        if (j == 2 || j == 4) ref = sl * s1;
        if (j == 3) ref = sl * s2;
        if (j == 8 || j == 10) ref = sr * s1;
        if (j == 9) ref = sr * s2;
        if (fabsf(s) < 0.1f) ref = 0.0f;
        st[(kStateOffRefPos + j) * E + le] = ref;
        float* f47 = smem + m.frame + le * HGYM_OBS_FRAME;
        float* p73 = smem + m.priv + le * HGYM_PRIV_FRAME;
        const float qq = (q - dflt) * c.scale_dof_pos;
        const float dq = qd * c.scale_dof_vel;
        f47[5 + j] = qq;   p73[5 + j] = qq;
        f47[17 + j] = dq;  p73[17 + j] = dq;
        f47[29 + j] = act; p73[29 + j] = act;
        p73[41 + j] = q - ref;
        st[(kStateOffLastLastActions + j) * E + le] = reset ? 0.0f : la;
        st[(kStateOffLastActions + j) * E + le] = act;
        st[(kStateOffLastDofVel + j) * E + le] = qd;
        st[(kStateOffActions + j) * E + le] = act;
    }
}

template <int E_T, bool kGeneric, bool kSplit = false>
HG_HD void env_step_phase_a(const EnvArgs& A, int block, int t, float* smem, int64_t csc0) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs, e0 = block * E;
    const int nE = (E < N - e0) ? E : (N - e0);
    if (t >= nE) return;
    const LdsMap m = lds_map(E);
    const EnvArgs S = make_shadow(A, smem, block, E);
    const RngKey rk = make_rng_key(A, csc0);
    if (!kSplit && A.mode == MODE_STEP && A.fused) synth_rest_env(S, smem + m.phys + t * kPhysDraws, t, E);
    const StepFlags fl = post_physics_env<kGeneric, kSplit>(S, rk, csc0 + 1, t, E, smem + m.frame + t * HGYM_OBS_FRAME, smem + m.priv + t * HGYM_PRIV_FRAME,
                                                            smem + m.jpart, smem + m.cscal, smem + m.reset_pose);
    reinterpret_cast<int*>(smem + m.reset_i)[t] = fl.reset;
    if (fl.reset) reinterpret_cast<int*>(smem + m.reset_list)[hg_atomic_inc_int(reinterpret_cast<int*>(smem + m.reset_cnt))] = t;
}

// The four-wavefront form of the split chain (post_physics_env's ROLE): lanes [0, W) run ROLE_MAIN, [W, 2 W) ROLE_REW_A, [2 W, 3 W)
// ROLE_REW_B, [3 W, 4 W) ROLE_FRAMES, one env per lane, W = 64 on the device (a wavefront each, on the four SIMDs of the CU).  Needs env_step_phase_j<.., true>
// before it (the snapshot) and env_step_reward_sum after the barrier behind it.
template <int E_T>
HG_HD void env_step_phase_a3(const EnvArgs& A, int block, int t, int nthreads, float* smem, int64_t csc0) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs, e0 = block * E;
    const int nE = (E < N - e0) ? E : (N - e0);
    const int W = nthreads >= 64 * kChainRoles ? 64 : nthreads / kChainRoles;        // (the host emulation's small launches: nthreads >= 4 nE)
    const int role = t / (W > 0 ? W : 1), le = t - role * W;
    if (role >= kChainRoles || le >= nE) return;
    const LdsMap m = lds_map(E);
    const RngKey rk = make_rng_key(A, csc0);
    if (role == 0) {
        const EnvArgs S = make_shadow(A, smem, block, E);
        const StepFlags fl = post_physics_env<false, true, ROLE_MAIN>(S, rk, csc0 + 1, le, E, smem + m.frame + le * HGYM_OBS_FRAME,
                                                                     smem + m.priv + le * HGYM_PRIV_FRAME, smem + m.jpart, smem + m.cscal,
                                                                     smem + m.reset_pose, nullptr);
        reinterpret_cast<int*>(smem + m.reset_i)[le] = fl.reset;
        if (fl.reset) reinterpret_cast<int*>(smem + m.reset_list)[hg_atomic_inc_int(reinterpret_cast<int*>(smem + m.reset_cnt))] = le;
        return;
    }
    EnvArgs S = make_shadow(A, smem, block, E);     // ... with what ROLE_MAIN rewrites re-aimed at the snapshot
    float* sn = smem + m.snap;
    S.sim.root = HgymStrided{sn + kSnapRoot * E, 1, E};
    S.st.commands = sn + kSnapCmd * E;
    S.st.last_root_vel = sn + kSnapLrv * E;
    S.st.episode_length = reinterpret_cast<int64_t*>(sn + kSnapEp * E);
    if (role == 1) {
        post_physics_env<false, true, ROLE_REW_A>(S, rk, csc0 + 1, le, E, nullptr, nullptr, smem + m.jpart, nullptr, nullptr, smem + m.terms);
    } else if (role == 2)
        post_physics_env<false, true, ROLE_REW_B>(S, rk, csc0 + 1, le, E, nullptr, nullptr, smem + m.jpart, nullptr, nullptr, smem + m.terms);
    else
        post_physics_env<false, true, ROLE_FRAMES>(S, rk, csc0 + 1, le, E, smem + m.frame + le * HGYM_OBS_FRAME, smem + m.priv + le * HGYM_PRIV_FRAME,
                                                   smem + m.jpart, smem + m.cscal, smem + m.reset_pose, nullptr);
}

// compute_reward's sum (legged_robot.py:217-235) and reset_idx's episode sums (:197-204) for the chain by roles, one env per
// lane of the LAST wavefront (env_step_phase_f, next to it, occupies the first ones): the same additions in the same order as
// post_physics_env<.., ROLE_ALL>
template <int E_T>
HG_HD void env_step_reward_sum(const EnvArgs& A, int block, int t, int nthreads, float* smem) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs, e0 = block * E;
    const int nE = (E < N - e0) ? E : (N - e0);
    const int le = t - (nthreads >= 128 ? nthreads - 64 : 0);
    if (le < 0 || le >= nE) return;
    const LdsMap m = lds_map(E);
    const int reset = reinterpret_cast<const int*>(smem + m.reset_i)[le];
    float* es = smem + m.state + kStateOffEpisodeSums * E + le;
    const float* ts = smem + m.terms + le;
    // every load before the first store or atomic (untyped float pointers: a load behind one would wait for it -- 22 LDS round trips)
    float tk[HGYM_NUM_REWARDS], sum[HGYM_NUM_REWARDS];
#pragma unroll
    for (int k = 0; k < HGYM_NUM_REWARDS; ++k) {
        tk[k] = ts[k * E];
        sum[k] = es[k * E];
    }
    float rew = 0.0f;
#pragma unroll
    for (int k = 0; k < HGYM_NUM_REWARDS; ++k) {
        rew += tk[k];
        sum[k] += tk[k];
    }
    if (A.cfg.only_positive_rewards) rew = fmaxf(rew, 0.0f);
    smem[m.rew + le] = rew;
    if (reset) {
#pragma unroll
        for (int k = 0; k < HGYM_NUM_REWARDS; ++k) {
            hg_atomic_add(&A.st.episode_acc[k], sum[k]);
            sum[k] = 0.0f;
        }
    }
#pragma unroll
    for (int k = 0; k < HGYM_NUM_REWARDS; ++k) es[k * E] = sum[k];
}

// Fast stage-out, the mirror of the fast stage-in (contiguous state, SoA sim tensors, full block of a compiled-in size): every tensor
// is a list of 16-byte items (4 envs of one component row) with an ARITHMETIC item -> (tensor row, LDS row) mapping.  The general
// form below reaches the same rows through component tables and per-field pointers, which the compiler turns into dependent loads
// from memory in front of the stores; this phase is executed by every wavefront of the workgroup and is paced by instruction issue.
// The five sim tensors take consecutive lane ranges, so that a wavefront runs the code of the one or two it has items of.
template <int E_T, bool kAssume = false>
HG_HD bool env_stage_out_fast(const EnvArgs& A, int block, int t, int nthreads, float* smem) {
    if (E_T <= 0 || (E_T & 3) != 0) return false;
    constexpr int E = E_T > 0 ? E_T : 4, Q = E / 4;
    const int N = A.cfg.num_envs, e0 = block * E;
    // every field of the (1.7 KB) kernel argument this phase needs, read up front and the layout test without short-circuits: read where
    // they are used they are a dozen scalar loads each waited for on its own, in a phase all wavefronts execute
    const int64_t es0 = A.sim.root.env_stride, es1 = A.sim.dof_pos.env_stride, es2 = A.sim.dof_vel.env_stride, es3 = A.sim.contact.env_stride,
                  es4 = A.sim.rigid.env_stride;
    const int64_t cs0 = A.sim.root.comp_stride, cs1 = A.sim.dof_pos.comp_stride, cs2 = A.sim.dof_vel.comp_stride, cs3 = A.sim.contact.comp_stride,
                  cs4 = A.sim.rigid.comp_stride;
    float* const b0 = A.sim.root.base;
    float* const b1 = A.sim.dof_pos.base;
    float* const b2 = A.sim.dof_vel.base;
    float* const b3 = A.sim.contact.base;
    float* const b4 = A.sim.rigid.base;
    float* const bs = A.st.commands;
    const int contig = A.state_contig, mode = A.mode, fused = A.fused;
    const int cc0 = A.contact_comp[0], cc1 = A.contact_comp[1], cc2 = A.contact_comp[2];
    const int rc0 = A.rigid_comp[0], rc1 = A.rigid_comp[1], rc2 = A.rigid_comp[2], rc3 = A.rigid_comp[3];
    const bool ok = (N - e0 >= E) & (contig != 0) & (es0 == 1) & (es1 == 1) & (es2 == 1) & (es3 == 1) & (es4 == 1);
    if (!kAssume && !ok) return false;
    const LdsMap m = lds_map(E);
    auto put = [&](float* g, const float* l) { *reinterpret_cast<EnvF4*>(g) = *reinterpret_cast<const EnvF4*>(l); };
    for (int i = t; i < kMutableComps * Q; i += nthreads) {
        const int c = i / Q, qd = i - c * Q;
        put(bs + (int64_t)c * N + e0 + 4 * qd, smem + m.state + c * E + 4 * qd);
    }
    auto first = [&](int lane0) {        // first item of the lane in a tensor whose items start at lane `lane0`
        const int i = t - lane0;
        return i < 0 ? i + nthreads : i;
    };
    int lane0 = 0;
    for (int i = first(lane0); i < 13 * Q; i += nthreads) {
        const int k = i / Q, qd = i - k * Q;
        put(b0 + (int64_t)k * cs0 + e0 + 4 * qd, smem + m.root + k * E + 4 * qd);
    }
    lane0 = (lane0 + 13 * Q) % nthreads;
    for (int i = first(lane0); i < 12 * Q; i += nthreads) {
        const int k = i / Q, qd = i - k * Q;
        put(b1 + (int64_t)k * cs1 + e0 + 4 * qd, smem + m.dof_pos + k * E + 4 * qd);
    }
    lane0 = (lane0 + 12 * Q) % nthreads;
    for (int i = first(lane0); i < 12 * Q; i += nthreads) {
        const int k = i / Q, qd = i - k * Q;
        put(b2 + (int64_t)k * cs2 + e0 + 4 * qd, smem + m.dof_vel + k * E + 4 * qd);
    }
    if (!(mode == MODE_STEP && fused)) return true;       // the synthetic physics wrote contacts / rigid bodies
    lane0 = (lane0 + 12 * Q) % nthreads;
    for (int i = first(lane0); i < 9 * Q; i += nthreads) {
        const int k = i / Q, qd = i - k * Q;
        const int comp = (k < 3 ? cc0 : (k < 6 ? cc1 : cc2)) + k % 3;
        put(b3 + (int64_t)comp * cs3 + e0 + 4 * qd, smem + m.contact + k * E + 4 * qd);
    }
    lane0 = (lane0 + 9 * Q) % nthreads;
    for (int i = first(lane0); i < 14 * Q; i += nthreads) {      // feet {x, y, z, vx, vy}, knees {x, y}
        const int k = i / Q, qd = i - k * Q;
        const int body = k < 10 ? k / 5 : 2 + (k - 10) / 2;
        const int c5 = k % 5;
        const int comp = k < 10 ? (c5 < 3 ? c5 : c5 + 4) : (k - 10) % 2;
        const int base = body == 0 ? rc0 : (body == 1 ? rc1 : (body == 2 ? rc2 : rc3));
        put(b4 + (int64_t)(base + comp) * cs4 + e0 + 4 * qd, smem + m.rigid + (body * 13 + comp) * E + 4 * qd);
    }
    return true;
}

template <int E_T, bool kAssume = false>
HG_HD void env_stage_out(const EnvArgs& A, int block, int t, int nthreads, float* smem) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs, e0 = block * E;
    const int nE = kAssume ? E : ((E < N - e0) ? E : (N - e0));
    const LdsMap m = lds_map(E);
    if (kAssume) {
        env_stage_out_fast<E_T, true>(A, block, t, nthreads, smem);
    } else if (!env_stage_out_fast<E_T>(A, block, t, nthreads, smem)) {
        copy_comp_rows<false>(nullptr, A, 0, kMutableComps, smem + m.state, E, e0, nE, N, t, nthreads);
        stage_sim<false>(A, m, smem, E, e0, nE, t, nthreads, A.mode == MODE_STEP && A.fused);   // the synthetic physics wrote contacts / rigid bodies
    }
    if (A.cfg.use_ref_actions && A.mode == MODE_STEP && A.fused && A.actions_in)       // the in-place `actions += ref_action`
        for (int i = t; i < nE * 12; i += nthreads) A.actions_in[(int64_t)e0 * 12 + i] = smem[m.actions_in + i];
    const uint8_t* fl = reinterpret_cast<const uint8_t*>(smem + m.flags);
    for (int i = t; i < nE; i += nthreads) {
        A.st.episode_length[e0 + i] = reinterpret_cast<const int64_t*>(smem + m.ep_len)[i];
        A.out.reset[e0 + i] = fl[i];
        if (A.mode == MODE_STEP) {
            A.out.rew[e0 + i] = smem[m.rew + i];
            A.out.time_out[e0 + i] = fl[E + i];
        }
    }
}

// ------------------------------------------------------------------------------------------------ generic options
// LeggedRobot._get_heights (legged_robot.py:761-795) for sample point p of env e: the point, rotated by the yaw part of the
// base quaternion (utils/math.py:39-43) and shifted to the base position, indexes the int16 height field; the height is the
// minimum of the cell and its +x / +y neighbours.  Same fp32 operation order as the torch expressions.
HG_HD void measure_height_point(const EnvArgs& A, int e, int p) {
    const HgymEnvConfig& c = A.cfg;
    const float* pose = A.st.height_pose + (int64_t)e * 7;
    float q[4] = {0.0f, 0.0f, pose[5], pose[6]};
    float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    n = n < 1e-9f ? 1e-9f : n;                                // isaacgym.torch_utils.normalize: x / norm.clamp(min=eps)
    q[2] = q[2] / n;
    q[3] = q[3] / n;
    const float* hp = A.st.height_points + (int64_t)p * 3;
    const float v[3] = {hp[0], hp[1], hp[2]};
    float w[3];
    quat_apply(q, v, w);
    const float fx = ((w[0] + pose[0]) + c.terrain_border) / c.terrain_hscale;
    const float fy = ((w[1] + pose[1]) + c.terrain_border) / c.terrain_hscale;
    int64_t px = (int64_t)fx, py = (int64_t)fy;               // .long(): truncation towards zero
    px = px < 0 ? 0 : (px > c.height_rows - 2 ? c.height_rows - 2 : px);
    py = py < 0 ? 0 : (py > c.height_cols - 2 ? c.height_cols - 2 : py);
    const int16_t* hs = A.st.height_samples;
    const int16_t h1 = hs[px * c.height_cols + py], h2 = hs[(px + 1) * c.height_cols + py], h3 = hs[px * c.height_cols + py + 1];
    int16_t h = h1 < h2 ? h1 : h2;
    h = h < h3 ? h : h3;
    A.st.measured_heights[(int64_t)e * c.num_height_points + p] = (float)h * c.terrain_vscale;
}

// LeggedRobot.update_command_curriculum (legged_robot.py:179-180,422-431) AFTER the step kernel and BEFORE the step finaliser.
// The reference decides inside reset_idx, between the termination check and the command resample, from a mean over the
// resetting envs: a cross-env dependency in the middle of the per-env chain, met on one step in max_episode_length.  The step
// kernel therefore resamples with the range it found, and on that one step this single-workgroup pass (a) takes the decision
// from the episode-sum accumulators the finaliser is about to consume, (b) moves the device-resident range, and (c) redoes the
// reset envs' x / y commands from the SAME uniform draws, patching the four places the two command columns went to: the
// state, the newest ring frames and the stacked outputs (those columns carry no observation noise, humanoid_env.py:176).
HG_HD bool command_curriculum_due(const EnvArgs& A, int64_t csc) {
    const HgymEnvConfig& c = A.cfg;
    if (!c.command_curriculum || !A.st.command_range_x || A.mode == MODE_PRIME) return false;
    if (csc % c.max_episode_length != 0) return false;
    const int64_t cnt = A.reset_count ? A.reset_count[0] : A.st.counters[1];
    if (cnt <= 0) return false;
    constexpr int kTrack = 20;                       // "tracking_lin_vel" in the alphabetical reward order
    const float mean_sum = A.st.episode_acc[kTrack] / (float)cnt;
    return mean_sum / (float)c.max_episode_length > (float)(0.8 * (double)c.reward_scales[kTrack]);
}

HG_HD void command_curriculum_move(const EnvArgs& A, double lo, double hi, double& nlo, double& nhi) {
    const double m = (double)A.cfg.max_curriculum;
    nlo = lo - 0.5;
    nlo = nlo < -m ? -m : (nlo > 0.0 ? 0.0 : nlo);
    nhi = hi + 0.5;
    nhi = nhi < 0.0 ? 0.0 : (nhi > m ? m : nhi);
}

// the redo for env e (global id); ring slots as in the stacking phase of the same step
HG_HD void command_curriculum_fix_env(const EnvArgs& A, const RngKey& rk, int e, float x_lo, float x_span, int64_t ring_step) {
    const HgymEnvConfig& c = A.cfg;
    const int N = c.num_envs;
    if (!A.out.reset[e]) return;
    const float u[3] = {A.noise.u_cmd ? A.noise.u_cmd[(int64_t)e * 6 + 3] : uniform_at(rk, (uint32_t)e, SLOT_CMD_RESET, 0),
                        A.noise.u_cmd ? A.noise.u_cmd[(int64_t)e * 6 + 4] : uniform_at(rk, (uint32_t)e, SLOT_CMD_RESET, 1), 0.0f};
    float cmd[4];
    resample_commands(c, x_lo, x_span, cmd, u);
    A.st.commands[e] = cmd[0];
    A.st.commands[(int64_t)N + e] = cmd[1];
    if (A.mode != MODE_STEP) return;                 // reset_all pushes no observation
    const int H = c.frame_stack, HC = c.c_frame_stack;
    const float lim = c.clip_obs;
    for (int k = 0; k < 2; ++k) {
        const float v = cmd[k] * c.scale_lin_vel;
        const float vc = clampf(v, -lim, lim);
        A.st.obs_ring[((int64_t)e * H + (int)(ring_step % H)) * HGYM_OBS_FRAME + 2 + k] = v;
        A.st.priv_ring[((int64_t)e * HC + (int)(ring_step % HC)) * HGYM_PRIV_FRAME + 2 + k] = v;
        A.out.obs[(int64_t)e * H * HGYM_OBS_FRAME + (H - 1) * HGYM_OBS_FRAME + 2 + k] = vc;
        A.out.priv_obs[(int64_t)e * HC * HGYM_PRIV_FRAME + (HC - 1) * HGYM_PRIV_FRAME + 2 + k] = vc;
    }
}

// ------------------------------------------------------------------------------------------------ history stacking
// Stacked, clipped observation rows (humanoid_env.py:250-262, legged_robot.py:105-108), oldest -> newest.
//   ring      [N][H][F] unclipped frames, the newest goes to slot `slot_new`
//   clean     the block's clean new frames (LDS, F floats per env); z the block's noise normals (LDS) or null
// Pass 1 adds the noise to the newest frame and pushes it into the ring.  Pass 2 copies the H-1 older frames: in ring
// order they are ONE circular run of (H-1)*F floats starting at slot_new+1, so each lane moves 16 bytes (the rows are
// only 4-byte aligned; gfx950 global accesses need no more) and issues kStackBatch independent loads before its first store.
struct StackGeom {
    int E, N, H, HC, e0, nE;
};
template <int H_T, int HC_T, int E_T>
HG_HD StackGeom stack_geom(const EnvArgs& A, int block) {
    StackGeom g;
    g.E = E_T > 0 ? E_T : A.envs_per_block;
    g.N = A.cfg.num_envs;
    g.H = H_T > 0 ? H_T : A.cfg.frame_stack;
    g.HC = HC_T > 0 ? HC_T : A.cfg.c_frame_stack;
    g.e0 = block * g.E;
    g.nE = (g.E < g.N - g.e0) ? g.E : (g.N - g.e0);
    return g;
}

constexpr int kStackBatch = 6;

// Pass "old": the H-1 older frames, oldest -> newest, copied + clipped unconditionally.  It depends on nothing this
// step computes, so the kernel runs it on the otherwise idle wavefronts WHILE the per-env scalar phase runs; envs that
// turn out to reset are fixed up afterwards (stack_reset_fix).
HG_HD void stack_old(const EnvArgs& A, const float* __restrict__ ring, float* __restrict__ dst, int e0, int nE, int H, int F, int slot_new,
                     int t, int nthreads) {
    const int row = H * F;
    const float lim = A.cfg.clip_obs;
    const int hrow = (H - 1) * F;
    if (hrow == 0) return;
    const int ipr = (hrow + 3) >> 2;              // 16-byte items per env row
    const int total = nE * ipr;
    const int start = (slot_new + 1) * F;         // first (oldest) element of the circular run
    for (int base = t; base < total; base += nthreads * kStackBatch) {
        float v[kStackBatch][4];
#pragma unroll
        for (int u = 0; u < kStackBatch; ++u) {
            const int i = base + u * nthreads;
            if (i >= total) continue;
            const int le = i / ipr, d0 = (i - le * ipr) << 2;
            const int n = hrow - d0 < 4 ? hrow - d0 : 4;
            int s0 = start + d0;
            if (s0 >= row) s0 -= row;
            const float* src = ring + (int64_t)(e0 + le) * row;
            if (n == 4 && s0 + 3 < row) {
                const EnvF4 q = *reinterpret_cast<const EnvF4*>(src + s0);
                v[u][0] = q.v[0]; v[u][1] = q.v[1]; v[u][2] = q.v[2]; v[u][3] = q.v[3];
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int sk = s0 + k;
                    if (sk >= row) sk -= row;
                    v[u][k] = k < n ? src[sk] : 0.0f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kStackBatch; ++u) {
            const int i = base + u * nthreads;
            if (i >= total) continue;
            const int le = i / ipr, d0 = (i - le * ipr) << 2;
            const int n = hrow - d0 < 4 ? hrow - d0 : 4;
            float* dp = dst + (int64_t)le * row + d0;
            if (n == 4) {
                const EnvF4 q = {{clampf(v[u][0], -lim, lim), clampf(v[u][1], -lim, lim), clampf(v[u][2], -lim, lim), clampf(v[u][3], -lim, lim)}};
                *reinterpret_cast<EnvF4*>(dp) = q;
            } else {
                for (int k = 0; k < n; ++k) dp[k] = clampf(v[u][k], -lim, lim);
            }
        }
    }
}

// Pass "new": the newest frame gets its noise, is pushed into the ring slot and written (clipped) as the last row.
template <bool kNoisy>
HG_HD void stack_new(const EnvArgs& A, float* __restrict__ ring, const float* clean_all, const float* z_all, const float* noise_vec,
                     float* __restrict__ dst, int e0, int nE, int H, int F, int slot_new, int t, int nthreads) {
    const int row = H * F;
    const float lim = A.cfg.clip_obs;
    for (int i = t; i < nE * F; i += nthreads) {
        const int le = i / F, k = i - le * F;
        const int e = e0 + le;
        float v = clean_all[i];
        if (kNoisy && A.cfg.add_noise) {
            const float ns = noise_vec[k];
            if (ns != 0.0f) v = v + z_all[i] * ns * A.cfg.noise_level;
            else v = v + 0.0f;   // clean + z*0*level in the reference
        }
        ring[((int64_t)e * H + slot_new) * F + k] = v;
        dst[(int64_t)le * row + (H - 1) * F + k] = clampf(v, -lim, lim);
    }
}

// The same pass in 16-byte pieces (compiled-in frame width): a work item is (env, quad of 4 consecutive frame entries) -- the last
// quad of a frame is shifted back to end at the frame's end and rewrites identical values -- so the ring slot and the output
// row's last frame each receive F / 4 (+1) unaligned 16-byte stores per env instead of F 4-byte ones.  Same arithmetic per entry.
template <bool kNoisy, int F>
HG_HD void stack_new_vec(const EnvArgs& A, float* __restrict__ ring, const float* clean_all, const float* z_all, const float* noise_vec,
                         float* __restrict__ dst, int e0, int nE, int H, int slot_new, int t, int nthreads, float* __restrict__ ahead = nullptr) {
    // ahead (HgymEnvOut.obs_ahead): this step's frame is frame H - 2 of the rows after next
    constexpr int Q = (F + 3) / 4;
    const int row = H * F;
    const float lim = A.cfg.clip_obs;
    for (int i = t; i < nE * Q; i += nthreads) {
        const int le = i / Q, j = i - le * Q;
        int off = 4 * j;
        off = off < F - 4 ? off : F - 4;
        EnvF4 r, o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = le * F + off + k;
            float v = clean_all[idx];
            if (kNoisy && A.cfg.add_noise) {
                const float ns = noise_vec[off + k];
                if (ns != 0.0f) v = v + z_all[idx] * ns * A.cfg.noise_level;
                else v = v + 0.0f;   // clean + z*0*level in the reference
            }
            r.v[k] = v;
            o.v[k] = clampf(v, -lim, lim);
        }
        st_stream4<(kEnvNT & 4) != 0>(ring + ((int64_t)(e0 + le) * H + slot_new) * F + off, r);
        st_stream4<(kEnvNT & 4) != 0>(dst + (int64_t)le * row + (H - 1) * F + off, o);
        if (ahead) st_stream4<(kEnvNT & 4) != 0>(ahead + (int64_t)le * row + (H - 2) * F + off, o);
    }
}

// Envs that reset this step: their history is cleared (humanoid_env.py:264-269) -- older frames of the output row and
// of the ring become zero.  Rare (a handful of envs per step), so a plain element loop per flagged env.
HG_HD void stack_reset_fix(float* __restrict__ ring, const int* s_reset, float* __restrict__ dst, int e0, int nE, int H, int F, int slot_new,
                           int t, int nthreads, bool zero_dst, int reset_count = -1, const int* reset_list = nullptr) {
    // reset_count / reset_list: the envs of this block that reset, when the caller has them as a compact list (the per-env phase
    // appends to it): about one workgroup in ten has one in any given step, and walking the 32 flags instead is a serial chain
    // of LDS reads on every lane (2.7 us) -- on the slowest workgroups of the launch, i.e. on its critical path.  -1: walk the flags.
    if (reset_count == 0) return;
    const int row = H * F, hrow = (H - 1) * F;
    const int start = (slot_new + 1) * F;
    const int n = reset_count > 0 ? reset_count : nE;
    for (int r = 0; r < n; ++r) {
        int le = r;
        if (reset_count > 0) le = reset_list[r];
        else if (!s_reset[le]) continue;
        float* rp = ring + (int64_t)(e0 + le) * row;
        float* dp = dst + (int64_t)le * row;
        for (int i = t; i < hrow; i += nthreads) {
            int sk = start + i;
            if (sk >= row) sk -= row;
            rp[sk] = 0.0f;
            if (zero_dst) dp[i] = 0.0f;
        }
    }
}

// The same for rows written one launch ahead (HgymEnvOut.obs_ahead): their frames 0 .. H-3 were copied from the ring before this
// step knew which envs reset.
HG_HD void stack_reset_ahead(float* __restrict__ dst, const int* s_reset, int nE, int H, int F, int t, int nthreads, int reset_count,
                             const int* reset_list) {
    if (reset_count == 0) return;
    const int row = H * F, hrow = (H - 2) * F;
    const int n = reset_count > 0 ? reset_count : nE;
    for (int r = 0; r < n; ++r) {
        int le = r;
        if (reset_count > 0) le = reset_list[r];
        else if (!s_reset[le]) continue;
        float* dp = dst + (int64_t)le * row;
        for (int i = t; i < hrow; i += nthreads) dp[i] = 0.0f;
    }
}

// Register-prefetched form of the older-frames copy for the compiled-in geometry: the lanes of wavefronts 1..3 issue ALL
// of the history loads at kernel entry, back to back, together with the state staging loads -- one memory round trip for
// the whole step -- and store them, clipped, WHILE wavefront 0 runs the per-env scalar phase; the few envs that turn out
// to reset are fixed up afterwards (stack_reset_fix), exactly as in the unprefetched form (stack_old).
// The H-1 older frames are, in ring order, two straight segments: A = [(slot+1)*F, H*F) and B = [0, slot*F).  Each is cut
// into 16-byte items; the last item of a segment is shifted back to END at the segment end (it overlaps its neighbour and
// rewrites identical values), so every item is one unconditional unaligned 16-byte load and one 16-byte store: no
// branches, nothing for the compiler to serialise.  Surplus item slots repeat the last item.
// EXCL = 1: the H - 1 frames older than the one this step writes into ring slot slot_new.  EXCL = 2 (rows written one launch ahead,
// HgymEnvOut.obs_ahead): the H - 2 frames that are older than the NEXT step's too -- slot slot_new + 1, the oldest, is left out as well.
template <int H, int F, int EXCL = 1>
struct HistGeom {
    static constexpr int kRow = H * F;
    static constexpr int kSlots = ((H - EXCL) * F + 3) / 4 + 1;    // item slots per env row (covers any split into A and B)
};
template <int H, int F, int E, int NT, int EXCL = 1>
HG_HD constexpr int hist_ni() { return H > EXCL ? (E * HistGeom<H, F, EXCL>::kSlots + NT - 1) / NT : 0; }

// item slot j of an env row -> (ring offset, row offset); slot_new = ring slot that receives the newest frame.  In ring order the
// frames kept are slots slot_new + EXCL .. slot_new + H - 1 (mod H): segment A up to the end of the ring, segment B from its start.
template <int H, int F, int EXCL = 1>
HG_HD void hist_slot(int slot_new, int j, int& src_off, int& dst_off) {
    const int a0 = slot_new + EXCL;
    const int LA = a0 < H ? (H - a0) * F : 0;
    const int LB = (H - EXCL) * F - LA;
    const int sA = a0 * F, sB = a0 < H ? 0 : (a0 - H) * F;
    const int nA = (LA + 3) >> 2, nB = (LB + 3) >> 2;
    j = j < nA + nB - 1 ? j : nA + nB - 1;                       // surplus slots repeat the last item
    const bool inA = j < nA;
    const int k = inA ? j : j - nA;
    const int L = inA ? LA : LB;
    int o = 4 * k;
    o = o < L - 4 ? o : L - 4;                                   // last item of a segment ends at the segment end
    src_off = (inA ? sA : sB) + o;
    dst_off = inA ? o : LA + o;
}
template <int H, int F, int NI, int EXCL = 1>
HG_HD void hist_load(const float* __restrict__ ring, int e0, int nE, int slot_new, int t, int nthreads, float (&v)[NI > 0 ? NI : 1][4]) {
    constexpr int S = HistGeom<H, F, EXCL>::kSlots, ROW = HistGeom<H, F, EXCL>::kRow;
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        int i = t + u * nthreads;
        i = i < nE * S ? i : nE * S - 1;
        const int le = i / S;
        int so, d_o;
        hist_slot<H, F, EXCL>(slot_new, i - le * S, so, d_o);
        const EnvF4 q = ld_stream4<(kEnvNT & 1) != 0>(ring + (int64_t)(e0 + le) * ROW + so);
        v[u][0] = q.v[0]; v[u][1] = q.v[1]; v[u][2] = q.v[2]; v[u][3] = q.v[3];
    }
}
template <int H, int F, int NI, int EXCL = 1>
HG_HD void hist_store(float* __restrict__ dst, int e0, int nE, int slot_new, int t, int nthreads, const int* s_reset, float lim,
                      const float (&v)[NI > 0 ? NI : 1][4]) {
    constexpr int S = HistGeom<H, F, EXCL>::kSlots, ROW = HistGeom<H, F, EXCL>::kRow;
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        int i = t + u * nthreads;
        i = i < nE * S ? i : nE * S - 1;
        const int le = i / S;
        int so, d_o;
        hist_slot<H, F, EXCL>(slot_new, i - le * S, so, d_o);
        const bool rs = s_reset ? s_reset[le] != 0 : false;     // null: reset envs are fixed up later (stack_reset_fix)
        EnvF4 q;
#pragma unroll
        for (int k = 0; k < 4; ++k) q.v[k] = rs ? 0.0f : clampf(v[u][k], -lim, lim);
        st_stream4<(kEnvNT & 2) != 0>(dst + (int64_t)(e0 + le) * ROW + d_o, q);
    }
}

// older frames of both outputs; (t, nthreads) may be any subset of the workgroup's lanes
template <int H_T, int HC_T, int E_T>
HG_HD void env_step_stack_old(const EnvArgs& A, int block, int t, int nthreads, int64_t ring_step) {
    if (A.mode == MODE_RESET_ALL) return;
    const StackGeom g = stack_geom<H_T, HC_T, E_T>(A, block);
    stack_old(A, A.st.obs_ring, A.out.obs + (int64_t)g.e0 * g.H * HGYM_OBS_FRAME, g.e0, g.nE, g.H, HGYM_OBS_FRAME, (int)(ring_step % g.H), t,
              nthreads);
    stack_old(A, A.st.priv_ring, A.out.priv_obs + (int64_t)g.e0 * g.HC * HGYM_PRIV_FRAME, g.e0, g.nE, g.HC, HGYM_PRIV_FRAME,
              (int)(ring_step % g.HC), t, nthreads);
}

template <int H_T, int HC_T, int E_T>
HG_HD void env_step_phase_b(const EnvArgs& A, int block, int t, int nthreads, float* smem, int64_t csc0, int64_t ring_step,
                            bool old_rows_final = false, bool ahead_rows_fixed = false) {
    const StackGeom g = stack_geom<H_T, HC_T, E_T>(A, block);
    const int H = g.H, HC = g.HC, e0 = g.e0, nE = g.nE;
    const LdsMap m = lds_map(g.E);
    const int* s_reset = reinterpret_cast<const int*>(smem + m.reset_i);
    (void)csc0;
    float* ro = A.st.obs_ring + (int64_t)e0 * H * HGYM_OBS_FRAME;
    float* rp = A.st.priv_ring + (int64_t)e0 * HC * HGYM_PRIV_FRAME;
    if (A.mode == MODE_RESET_ALL) {   // reset_idx(all) without compute_observations: just clear the history
        const int64_t no = (int64_t)nE * H * HGYM_OBS_FRAME, np = (int64_t)nE * HC * HGYM_PRIV_FRAME;
        for (int64_t i = t; i < no; i += nthreads) ro[i] = 0.0f;
        for (int64_t i = t; i < np; i += nthreads) rp[i] = 0.0f;
        return;
    }
    float* dobs = A.out.obs + (int64_t)e0 * H * HGYM_OBS_FRAME;
    float* dpriv = A.out.priv_obs + (int64_t)e0 * HC * HGYM_PRIV_FRAME;
    float* dahead = (H_T > 0 && HC_T > 0 && A.out.obs_ahead) ? A.out.obs_ahead + (int64_t)e0 * H * HGYM_OBS_FRAME : nullptr;      // fused rollout step only
    float* pahead = (H_T > 0 && HC_T > 0 && A.out.priv_ahead) ? A.out.priv_ahead + (int64_t)e0 * HC * HGYM_PRIV_FRAME : nullptr;
    if (H_T > 0 && HC_T > 0) {
        stack_new_vec<true, HGYM_OBS_FRAME>(A, A.st.obs_ring, smem + m.frame, smem + m.z_obs, smem + m.noise_vec, dobs, e0, nE, H,
                                            (int)(ring_step % H), t, nthreads, dahead);
        stack_new_vec<false, HGYM_PRIV_FRAME>(A, A.st.priv_ring, smem + m.priv, nullptr, nullptr, dpriv, e0, nE, HC, (int)(ring_step % HC), t,
                                              nthreads, pahead);
    } else {
        stack_new<true>(A, A.st.obs_ring, smem + m.frame, smem + m.z_obs, smem + m.noise_vec, dobs, e0, nE, H, HGYM_OBS_FRAME,
                        (int)(ring_step % H), t, nthreads);
        stack_new<false>(A, A.st.priv_ring, smem + m.priv, nullptr, nullptr, dpriv, e0, nE, HC, HGYM_PRIV_FRAME, (int)(ring_step % HC), t,
                         nthreads);
    }
    const int nreset = reinterpret_cast<const int*>(smem + m.reset_cnt)[0];
    const int* rlist = reinterpret_cast<const int*>(smem + m.reset_list);
    stack_reset_fix(A.st.obs_ring, s_reset, dobs, e0, nE, H, HGYM_OBS_FRAME, (int)(ring_step % H), t, nthreads, !old_rows_final, nreset, rlist);
    stack_reset_fix(A.st.priv_ring, s_reset, dpriv, e0, nE, HC, HGYM_PRIV_FRAME, (int)(ring_step % HC), t, nthreads, !old_rows_final, nreset,
                    rlist);
    // (ahead_rows_fixed: the reset envs' older frames of the rows after next are zeroed elsewhere -- by the next launch, for the
    // fused rollout launch: the kernel boundary orders the two stores)
    if (dahead && !ahead_rows_fixed) stack_reset_ahead(dahead, s_reset, nE, H, HGYM_OBS_FRAME, t, nthreads, nreset, rlist);
    if (pahead && !ahead_rows_fixed) stack_reset_ahead(pahead, s_reset, nE, HC, HGYM_PRIV_FRAME, t, nthreads, nreset, rlist);
}

// Step finaliser (hgym_finalize.hpp) on an EnvArgs record.
HG_HD FinArgs fin_of(const EnvArgs& A) {
    FinArgs f = make_fin_args(A.cfg, A.st, A.out, A.mode);
    if (A.reset_count) f.reset_count = A.reset_count;
    return f;
}
HG_HD void env_finalize_part1(const EnvArgs& A, int t, int nthreads) { fin_part1(fin_of(A), t, nthreads); }
HG_HD void env_finalize_store(const EnvArgs& A, int t, int nthreads) { fin_store(fin_of(A), t, nthreads); }
HG_HD void env_finalize_part2(const EnvArgs& A) { fin_part2(fin_of(A)); }

#undef FG
}  // namespace hgym
