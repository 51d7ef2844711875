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
    const float* actions_in;  // (N,12) row-major or null
    int mode;
    int fused;                // 1: pre_physics + synthetic physics run inside the step kernel
    int envs_per_block;
    int ablate;               // debug/profiling only: bit0 skip stage-in, bit1 skip phase A, bit2 skip stage-out, bit3 skip phase B,
                              // bit4 skip pre_physics+synthetic physics, bit5 skip the per-env post-physics arithmetic
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

HG_HD float sget(const HgymStrided& s, int env, int comp) { return s.base[(int64_t)env * s.env_stride + (int64_t)comp * s.comp_stride]; }
HG_HD void sset(const HgymStrided& s, int env, int comp, float v) { s.base[(int64_t)env * s.env_stride + (int64_t)comp * s.comp_stride] = v; }

#define FG(p, c) (p)[(int64_t)(c) * N + e]

HG_HD void hg_atomic_add(float* p, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(p, v);
#else
    *p += v;
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
// humanoid_env.py:189-197 + legged_robot.py:90-91
HG_HD void pre_physics_env(const EnvArgs& A, const RngKey& rk, int e, int N) {
    const HgymEnvConfig& c = A.cfg;
    const int ge = A.env_base + e;
    const float u = nz_uniform(A.noise.u_delay, 1, 0, rk, e, ge, SLOT_DELAY_CMD, 0);
    const float delay = u * c.action_delay;
    float zn[12];
    if (!A.noise.z_act) normals_block<3>(rk, (uint32_t)ge, SLOT_ACT, zn);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        float a = clampf(A.actions_in[(int64_t)e * 12 + j], -c.clip_actions, c.clip_actions);
        a = (1.0f - delay) * a + delay * FG(A.st.actions, j);
        const float z = A.noise.z_act ? A.noise.z_act[(int64_t)e * 12 + j] : zn[j];
        a = a + c.action_noise * z * a;
        FG(A.st.actions, j) = clampf(a, -c.clip_actions, c.clip_actions);
    }
}

// legged_robot.py:340-356
HG_HD float pd_torque(const HgymEnvConfig& c, int j, float a, float q, float qd) {
    const float t = c.p_gains[j] * (a * c.action_scale + c.default_dof_pos[j] - q) - c.d_gains[j] * qd;
    return clampf(t, -c.torque_limits[j], c.torque_limits[j]);
}

HG_HD void pd_torques_env(const EnvArgs& A, int e, int N) {
#pragma unroll
    for (int j = 0; j < 12; ++j)
        FG(A.st.torques, j) = pd_torque(A.cfg, j, FG(A.st.actions, j), sget(A.sim.dof_pos, e, j), sget(A.sim.dof_vel, e, j));
}

// ------------------------------------------------------------------------------------------------ synthetic physics
// Stands where PhysX is (legged_robot.py:94-101,124-126).  SURVEY.md §8d: unit-inertia joints under the PD
// torque, `decimation` semi-implicit Euler substeps with URDF joint limits; root / contact / rigid-body
// tensors drawn from Philox.  This is the benchmark backend only -- it has no reference counterpart.
HG_HD void synth_physics_env(const EnvArgs& A, const RngKey& rk, int e, int N) {
    const HgymEnvConfig& c = A.cfg;
    const uint32_t ue = (uint32_t)(A.env_base + e);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        float q = sget(A.sim.dof_pos, e, j), qd = sget(A.sim.dof_vel, e, j);
        const float a = FG(A.st.actions, j);
        float t = 0.0f;
        for (int s = 0; s < c.decimation; ++s) {
            t = pd_torque(c, j, a, q, qd);
            qd = qd + c.sim_dt * t;
            q = q + c.sim_dt * qd;
            if (q < c.dof_lower[j]) { q = c.dof_lower[j]; qd = 0.0f; }
            if (q > c.dof_upper[j]) { q = c.dof_upper[j]; qd = 0.0f; }
        }
        // the reference evaluates the torque before each substep; the last evaluation is what rewards see
        FG(A.st.torques, j) = t;
        sset(A.sim.dof_pos, e, j, q);
        sset(A.sim.dof_vel, e, j, qd);
    }
    // root: mean-reverting orientation walk, small height jitter, gaussian velocities
    const U4 r0 = rng4(rk, ue, SLOT_PHYS + 0);
    float n[12];
    normals_block<3>(rk, ue, SLOT_PHYS + 1, n);
    float qx = 0.9f * sget(A.sim.root, e, 3) + 0.05f * n[0];
    float qy = 0.9f * sget(A.sim.root, e, 4) + 0.05f * n[1];
    float qz = 0.9f * sget(A.sim.root, e, 5) + 0.05f * n[2];
    const float inv = 1.0f / sqrtf(qx * qx + qy * qy + qz * qz + 1.0f);
    sset(A.sim.root, e, 3, qx * inv);
    sset(A.sim.root, e, 4, qy * inv);
    sset(A.sim.root, e, 5, qz * inv);
    sset(A.sim.root, e, 6, inv);
    sset(A.sim.root, e, 2, 0.9f + 0.02f * (2.0f * u01(r0.x) - 1.0f));
#pragma unroll
    for (int i = 0; i < 6; ++i) sset(A.sim.root, e, 7 + i, 0.3f * n[3 + i]);
    // contacts: feet load follows the gait clock, rare base-link hits end episodes (~ every 500 steps)
    const float s = sinf(kTwoPi * gait_phase(c, A.st.episode_length[e] + 1));
    float stance[2];
    stance_from_sin(s, stance);
    const U4 r1 = rng4(rk, ue, SLOT_PHYS + 4);
    const float uf[2] = {u01(r1.x), u01(r1.y)};
    const float ug[2] = {u01(r1.z), u01(r1.w)};
    // (all other contact entries stay at their initial zero)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const float on = (stance[f] > 0.5f || ug[f] > 0.4f) ? 1.0f : 0.0f;
        sset(A.sim.contact, e, A.contact_comp[1 + f] + 2, 600.0f * uf[f] * on);
    }
    const float hit = (u01(r0.y) < 0.002f) ? 2.0f : 0.0f;
    sset(A.sim.contact, e, A.contact_comp[0] + 0, hit * n[9]);
    sset(A.sim.contact, e, A.contact_comp[0] + 1, hit * n[10]);
    sset(A.sim.contact, e, A.contact_comp[0] + 2, hit * n[11]);
    // rigid bodies: only the entries the rewards read (feet x,y,z,vx,vy ; knees x,y)
    float m[12];
    normals_block<3>(rk, ue, SLOT_PHYS + 5, m);
    const U4 r2 = rng4(rk, ue, SLOT_PHYS + 8);
    const float uz[2] = {u01(r2.x), u01(r2.y)};
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

// ------------------------------------------------------------------------------------------------ commands
// legged_robot.py:322-336 for one env; u[3] = draws for x, y, heading
HG_HD void resample_commands(const HgymEnvConfig& c, float cmd[4], const float u[3]) {
    cmd[0] = c.cmd_x_span * u[0] + c.cmd_x_lo;
    cmd[1] = c.cmd_y_span * u[1] + c.cmd_y_lo;
    cmd[3] = c.cmd_h_span * u[2] + c.cmd_h_lo;
    const float keep = (sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) > 0.2f) ? 1.0f : 0.0f;
    cmd[0] *= keep;
    cmd[1] *= keep;
}

HG_HD float dist_reward(const HgymEnvConfig& c, float ax, float ay, float bx, float by, float max_df) {
    const float dx = ax - bx, dy = ay - by;
    const float d = sqrtf(dx * dx + dy * dy);
    const float d_min = clampf(d - c.min_dist, -0.5f, 0.0f);
    const float d_max = clampf(d - max_df, 0.0f, 0.5f);
    return (expf(-fabsf(d_min) * 100.0f) + expf(-fabsf(d_max) * 100.0f)) / 2.0f;
}

struct StepFlags {
    int reset;     // env resets this step (history must be zeroed before the push)
};

// ------------------------------------------------------------------------------------------------ E4-E12
// LeggedRobot.post_physics_step for ONE env (legged_robot.py:119-151) with XBotLFreeEnv's reward terms
// (humanoid_env.py:272-540, alphabetical order), mask-driven reset_idx (legged_robot.py:163-215) and the
// clean observation frames (humanoid_env.py:200-244).  frame47 / priv73 receive the UN-noised new frames
// (LDS on the device); noise, history stacking and clipping happen in the cooperative phase.
HG_HD StepFlags post_physics_env(const EnvArgs& A, const RngKey& rk, int64_t csc, int e, int N, float* frame47,
                                 float* priv73) {
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
        q[j] = sget(A.sim.dof_pos, e, j);
        qd[j] = sget(A.sim.dof_vel, e, j);
    }
    float cmd[4] = {FG(S.commands, 0), FG(S.commands, 1), FG(S.commands, 2), FG(S.commands, 3)};
    float act[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) act[j] = FG(S.actions, j);
    float blv[3], bav[3], grav[3], eul[3];
    float fz[2], contact[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        fz[f] = sget(A.sim.contact, e, A.contact_comp[1 + f] + 2);
        contact[f] = fz[f] > 5.0f ? 1.0f : 0.0f;
    }
    int reset = 0, time_out = 0;
    float rew = 0.0f;

    if (mode == MODE_STEP) {
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
            resample_commands(c, cmd, u);
        }
        {
            const float fv[3] = {1.0f, 0.0f, 0.0f};
            float fw[3];
            quat_apply(root + 3, fv, fw);
            const float heading = atan2f(fw[1], fw[0]);
            cmd[2] = clampf(0.5f * wrap_like_reference(cmd[3] - heading), -1.0f, 1.0f);
        }
        if (c.push_robots && (csc % c.push_interval == 0)) {          // humanoid_env.py:83-98
            const float px = c.push_vel_span * nz_uniform(A.noise.u_push, 5, 0, rk, e, ge, SLOT_PUSH, 0) + c.push_vel_lo;
            const float py = c.push_vel_span * nz_uniform(A.noise.u_push, 5, 1, rk, e, ge, SLOT_PUSH, 1) + c.push_vel_lo;
            FG(S.push_force, 0) = px;
            FG(S.push_force, 1) = py;
            root[7] = px;
            root[8] = py;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float t = c.push_ang_span * nz_uniform(A.noise.u_push, 5, 2 + i, rk, e, ge, SLOT_PUSH, 2 + i) + c.push_ang_lo;
                FG(S.push_torque, i) = t;
                root[10 + i] = t;
            }
            sset(A.sim.root, e, 7, root[7]);
            sset(A.sim.root, e, 8, root[8]);
            sset(A.sim.root, e, 10, root[10]);
            sset(A.sim.root, e, 11, root[11]);
            sset(A.sim.root, e, 12, root[12]);
        }
        // check_termination :156-161
        {
            const float bx = sget(A.sim.contact, e, A.contact_comp[0] + 0);
            const float by = sget(A.sim.contact, e, A.contact_comp[0] + 1);
            const float bz = sget(A.sim.contact, e, A.contact_comp[0] + 2);
            const float bn = sqrtf(bx * bx + by * by + bz * bz);
            time_out = ep > (int64_t)c.max_episode_length;
            reset = (bn > 1.0f) || time_out;

            // ---------------- compute_reward :217-235, 22 terms in alphabetical order ----------------
            const float s = sinf(kTwoPi * gait_phase(c, ep));
            float stance[2];
            stance_from_sin(s, stance);
            float fpos[2][3], fvxy[2][2], kxy[2][2], fxyz[2][3];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
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
            float term[HGYM_NUM_REWARDS];
            // 0 action_smoothness :530-540
            {
                float t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const float la = FG(S.last_actions, j), lla = FG(S.last_last_actions, j);
                    const float d1 = la - act[j];
                    t1 += d1 * d1;
                    const float d2 = act[j] + lla - 2.0f * la;
                    t2 += d2 * d2;
                    t3 += fabsf(act[j]);
                }
                term[0] = t1 + t2 + 0.05f * t3;
            }
            // 1 base_acc :386-393
            {
                float a2 = 0.0f;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const float d = FG(S.last_root_vel, i) - root[7 + i];
                    a2 += d * d;
                }
                term[1] = expf(-sqrtf(a2) * 3.0f);
            }
            // 2 base_height :374-384
            {
                const float mh = (fpos[0][2] * stance[0] + fpos[1][2] * stance[1]) / (stance[0] + stance[1]);
                const float bh = root[2] - (mh - 0.05f);
                term[2] = expf(-fabsf(bh - c.base_height_target) * 100.0f);
            }
            // 3 collision :523-528
            term[3] = (bn > 0.1f) ? 1.0f : 0.0f;
            // 4 default_joint_pos :362-372
            {
                float jd[12], all2 = 0.0f;
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    jd[j] = q[j] - c.default_dof_pos[j];
                    all2 += jd[j] * jd[j];
                }
                float yr = sqrtf(jd[0] * jd[0] + jd[1] * jd[1]) + sqrtf(jd[6] * jd[6] + jd[7] * jd[7]);
                yr = clampf(yr - 0.1f, 0.0f, 50.0f);
                term[4] = expf(-yr * 100.0f) - 0.01f * sqrtf(all2);
            }
            // 5 dof_acc :516-521 ; 6 dof_vel :509-514 ; 17 torques :502-507
            {
                float acc = 0.0f, vel = 0.0f, tq = 0.0f;
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const float a = (FG(S.last_dof_vel, j) - qd[j]) / c.dt;
                    acc += a * a;
                    vel += qd[j] * qd[j];
                    const float t = FG(S.torques, j);
                    tq += t * t;
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
                    const float lc = FG(S.last_contacts, f);
                    const int filt = (contact[f] > 0.5f) || (stance[f] > 0.5f) || (lc > 0.5f);
                    FG(S.last_contacts, f) = contact[f];
                    float air = FG(S.feet_air_time, f);
                    const int first = (air > 0.0f) && filt;
                    air += c.dt;
                    r7 += clampf(air, 0.0f, 0.5f) * (first ? 1.0f : 0.0f);
                    FG(S.feet_air_time, f) = air * (filt ? 0.0f : 1.0f);

                    const float z = fpos[f][2] - 0.05f;
                    float fh = FG(S.feet_height, f) + (z - FG(S.last_feet_z, f));
                    FG(S.last_feet_z, f) = z;
                    const float swing = 1.0f - stance[f];
                    r8 += ((fabsf(fh - c.target_feet_height) < 0.01f) ? 1.0f : 0.0f) * swing;
                    FG(S.feet_height, f) = fh * (contact[f] > 0.5f ? 0.0f : 1.0f);
                }
                term[7] = r7;
                term[8] = r8;
            }
            // 9 feet_contact_forces :355-360 ; 10 feet_contact_number :336-344 ; 12 foot_slip :308-318
            {
                float r9 = 0.0f, r10 = 0.0f, r12 = 0.0f;
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const float fn = sqrtf(fxyz[f][0] * fxyz[f][0] + fxyz[f][1] * fxyz[f][1] + fxyz[f][2] * fxyz[f][2]);
                    r9 += clampf(fn - c.max_contact_force, 0.0f, 400.0f);
                    r10 += (contact[f] == stance[f]) ? 1.0f : -0.3f;
                    r12 += sqrtf(sqrtf(fvxy[f][0] * fvxy[f][0] + fvxy[f][1] * fvxy[f][1])) * contact[f];
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
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const float d = q[j] - FG(S.ref_dof_pos, j);
                    e2 += d * d;
                }
                const float en = sqrtf(e2);
                term[13] = expf(-2.0f * en) - 0.2f * clampf(en, 0.0f, 0.5f);
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
            term[16] = (expf(-(fabsf(eul[0]) + fabsf(eul[1])) * 10.0f) +
                        expf(-sqrtf(grav[0] * grav[0] + grav[1] * grav[1]) * 20.0f)) / 2.0f;
            // 18 track_vel_hard :408-425 ; 19 tracking_ang_vel :436-444 ; 20 tracking_lin_vel :427-434
            {
                const float ex = cmd[0] - blv[0], ey = cmd[1] - blv[1];
                const float le2 = ex * ex + ey * ey;
                const float le = sqrtf(le2);
                const float ae = fabsf(cmd[2] - bav[2]);
                term[18] = (expf(-le * 10.0f) + expf(-ae * 10.0f)) / 2.0f - 0.2f * (le + ae);
                const float d = cmd[2] - bav[2];
                term[19] = expf(-(d * d) * c.tracking_sigma);
                term[20] = expf(-le2 * c.tracking_sigma);
            }
            // 21 vel_mismatch_exp :396-406
            term[21] = (expf(-(blv[2] * blv[2]) * 10.0f) + expf(-sqrtf(bav[0] * bav[0] + bav[1] * bav[1]) * 5.0f)) / 2.0f;

#pragma unroll
            for (int k = 0; k < HGYM_NUM_REWARDS; ++k) {
                const float t = term[k] * c.reward_scales[k];
                rew += t;
                FG(S.episode_sums, k) += t;
            }
            if (c.only_positive_rewards) rew = fmaxf(rew, 0.0f);
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

    // ---------------- reset_idx :163-215 (+ humanoid_env.py:264-269), mask-driven ----------------
    if (reset) {
        fl.reset = 1;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            q[j] = c.default_dof_pos[j] + (c.dof_reset_span * nz_uniform(A.noise.u_dof, 12, j, rk, e, ge, SLOT_DOF, j) + c.dof_reset_lo);
            qd[j] = 0.0f;
            sset(A.sim.dof_pos, e, j, q[j]);
            sset(A.sim.dof_vel, e, j, 0.0f);
            act[j] = 0.0f;
            FG(S.last_actions, j) = 0.0f;
            FG(S.last_last_actions, j) = 0.0f;
            FG(S.last_dof_vel, j) = 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 13; ++i) root[i] = c.base_init_state[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) root[i] += FG(S.env_origins, i);
#pragma unroll
        for (int i = 0; i < 13; ++i) sset(A.sim.root, e, i, root[i]);
        {
            const float u[3] = {nz_uniform(A.noise.u_cmd, 6, 3, rk, e, ge, SLOT_CMD_RESET, 0),
                                nz_uniform(A.noise.u_cmd, 6, 4, rk, e, ge, SLOT_CMD_RESET, 1),
                                nz_uniform(A.noise.u_cmd, 6, 5, rk, e, ge, SLOT_CMD_RESET, 2)};
            resample_commands(c, cmd, u);
        }
        FG(S.feet_air_time, 0) = 0.0f;
        FG(S.feet_air_time, 1) = 0.0f;
        ep = 0;
        // extras["episode"]: mean over resetting envs, finished by the step finaliser
        hg_atomic_inc(&S.counters[1]);
#pragma unroll
        for (int k = 0; k < HGYM_NUM_REWARDS; ++k) {
            hg_atomic_add(&S.episode_acc[k], FG(S.episode_sums, k));
            FG(S.episode_sums, k) = 0.0f;
        }
        euler_xyz_wrapped(root + 3, eul);
        const float gvec[3] = {0.0f, 0.0f, -1.0f};
        quat_rotate_inverse(root + 3, gvec, grav);
    }

    if (mode != MODE_RESET_ALL) {
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
        for (int j = 0; j < 12; ++j) FG(S.ref_dof_pos, j) = ref[j];
        float ci[5] = {s, co, cmd[0] * c.scale_lin_vel, cmd[1] * c.scale_lin_vel, cmd[2] * c.scale_ang_vel};
#pragma unroll
        for (int i = 0; i < 5; ++i) { frame47[i] = ci[i]; priv73[i] = ci[i]; }
#pragma unroll
        for (int j = 0; j < 12; ++j) {
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
            priv73[64 + i] = FG(S.push_torque, i);
        }
        priv73[62] = FG(S.push_force, 0);
        priv73[63] = FG(S.push_force, 1);
        priv73[67] = FG(S.friction, 0);
        priv73[68] = FG(S.body_mass, 0) / 30.0f;
        priv73[69] = stance[0];
        priv73[70] = stance[1];
        priv73[71] = contact[0];
        priv73[72] = contact[1];
    }

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
            FG(S.last_last_actions, j) = reset ? 0.0f : FG(S.last_actions, j);
            FG(S.last_actions, j) = act[j];
            FG(S.last_dof_vel, j) = qd[j];
            FG(S.actions, j) = act[j];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) FG(S.last_root_vel, i) = root[7 + i];
        A.out.rew[e] = rew;
        A.out.reset[e] = (uint8_t)reset;
        A.out.time_out[e] = (uint8_t)time_out;
    } else {
#pragma unroll
        for (int j = 0; j < 12; ++j) FG(S.actions, j) = 0.0f;
        A.out.reset[e] = 1;
    }
    return fl;
}

// ------------------------------------------------------------------------------------------------ phase B
// One element of the stacked, clipped observation (humanoid_env.py:250-262, legged_robot.py:105-108):
// rows are oldest -> newest; the newest frame gets its noise here and is pushed into the ring.
//   ring      [N][H][F] unclipped frames, newest at slot `slot_new`
//   clean     this env's clean new frame (LDS), F floats
// Two uniform passes per output tensor: (1) the newest frame (noise, ring push), (2) the H-1 older frames, a pure
// shifted copy in which every lane issues kStackBatch independent, UNCONDITIONAL ring loads before its first store
// (the output and the ring may alias as far as the compiler knows, so a load behind a store would serialise).
constexpr int kStackBatch = 8;

template <bool kNoisy>
HG_HD void stack_rows(const EnvArgs& A, const RngKey& rk, float* ring, const float* clean_all, const int* s_reset, float* dst, int e0,
                      int nE, int H, int F, int slot_new, int t, int nthreads) {
    const int row = H * F;
    const float lim = A.cfg.clip_obs;
    // pass 1: newest frame
    for (int i = t; i < nE * F; i += nthreads) {
        const int le = i / F, k = i - le * F;
        const int e = e0 + le;
        float v = clean_all[i];
        if (kNoisy && A.cfg.add_noise) {
            if (A.cfg.obs_noise[k] != 0.0f) {
                const float z = nz_normal(A.noise.z_obs, HGYM_OBS_FRAME, k, rk, e, e, SLOT_OBS, k);
                v = v + z * A.cfg.obs_noise[k] * A.cfg.noise_level;
            } else {
                v = v + 0.0f;   // clean + z*0*level in the reference
            }
        }
        ring[((int64_t)e * H + slot_new) * F + k] = v;
        dst[(int64_t)le * row + (H - 1) * F + k] = clampf(v, -lim, lim);
    }
    // pass 2: older frames, oldest -> newest
    const int hrow = (H - 1) * F;
    const int total = nE * hrow;
    for (int base = t; base < total; base += nthreads * kStackBatch) {
        float* p[kStackBatch];
        float old[kStackBatch];
        int di[kStackBatch];
        bool rs[kStackBatch];
#pragma unroll
        for (int u = 0; u < kStackBatch; ++u) {
            int i = base + u * nthreads;
            i = i < total ? i : total - 1;             // clamp: surplus lanes redo the last element (same value)
            const int le = i / hrow;
            const int rem = i - le * hrow;
            const int j = rem / F;
            const int k = rem - j * F;
            int slot = slot_new + 1 + j;
            if (slot >= H) slot -= H;
            p[u] = ring + ((int64_t)(e0 + le) * H + slot) * F + k;
            di[u] = le * row + rem;
            rs[u] = s_reset[le] != 0;
        }
#pragma unroll
        for (int u = 0; u < kStackBatch; ++u) old[u] = *p[u];
#pragma unroll
        for (int u = 0; u < kStackBatch; ++u) {
            if (rs[u]) *p[u] = 0.0f;
            dst[di[u]] = rs[u] ? 0.0f : clampf(old[u], -lim, lim);
        }
    }
}

// ------------------------------------------------------------------------------------------------ workgroup phases
// The step kernel runs, per workgroup of E envs:
//   stage-in   all lanes: every per-env input (state, the needed sim components, actions, noise rows) is copied
//              into LDS with coalesced loads -- ONE global-memory round trip instead of ~250 dependent ones;
//   phase A    one lane per env: the per-env arithmetic above, executed against an LDS "shadow" of EnvArgs
//              (same code, pointers re-aimed at LDS, component stride E instead of N);
//   stage-out  all lanes: modified state / sim components / step outputs written back, coalesced;
//   phase B    all lanes: history stacking (stack_element) straight to the row-major outputs.
// Every phase is a plain function of (block, thread) so tests/hostcheck runs the identical code on the host.
HG_HD RngKey make_rng_key(const EnvArgs& A, int64_t csc0) {
    RngKey rk;
    rk.k0 = (uint32_t)A.cfg.seed;
    rk.k1 = (uint32_t)(A.cfg.seed >> 32);
    rk.s0 = (uint32_t)csc0;
    rk.s1 = (uint32_t)(csc0 >> 32) ^ (A.mode == MODE_STEP ? 0u : 0x80000000u);
    return rk;
}

// components of every [C][N] state field, in HgymEnvState order (commands ... env_origins)
constexpr int kNumStateFields = 22;
HG_HD int state_field_comps(int f) {
    constexpr int c[kNumStateFields] = {4, 12, 12, 12, 12, 6, 12, 2, 2, 2, 2, 12, 3, 3, HGYM_NUM_REWARDS, 3, 3, 3, 3, 1, 1, 3};
    return c[f];
}
constexpr int kStateComps = 4 + 12 * 6 + 6 + 2 * 4 + 3 * 2 + HGYM_NUM_REWARDS + 3 * 4 + 1 + 1 + 3;   // 136
constexpr int kFirstConstField = 19;    // friction, body_mass, env_origins are read-only
HG_HD float** state_field_ptr(HgymEnvState& S, int f) { return (&S.commands) + f; }
HG_HD float* const* state_field_ptr(const HgymEnvState& S, int f) { return (&S.commands) + f; }

// LDS carve (float offsets) for a block of E envs
struct LdsMap {
    int state, root, dof_pos, dof_vel, contact, rigid, actions_in, u_delay, z_act, u_cmd, u_dof, u_push, frame, priv, rew, ep_len,
        flags, reset_i, total;
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
    m.u_cmd = o;      o += 6 * E;
    m.u_dof = o;      o += 12 * E;
    m.u_push = o;     o += 5 * E;
    m.frame = o;      o += HGYM_OBS_FRAME * E;
    m.priv = o;       o += HGYM_PRIV_FRAME * E;
    m.rew = o;        o += E;
    o = (o + 1) & ~1;
    m.ep_len = o;     o += 2 * E;          // int64[E]
    m.flags = o;      o += (2 * E + 3) / 4;  // uint8 reset[E], time_out[E]
    m.reset_i = o;    o += E;              // int[E]: "history must be cleared" flags for phase B
    m.total = o;
    return m;
}
inline size_t step_smem_bytes(int E) { return (size_t)lds_map(E).total * sizeof(float); }

constexpr int kFootRigidComps[5] = {0, 1, 2, 7, 8};
constexpr int kKneeRigidComps[2] = {0, 1};

// The LDS shadow of EnvArgs for block `block`: same configuration, pointers into smem, component stride E.
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
    if (A.noise.u_delay) S.noise.u_delay = smem + m.u_delay;
    if (A.noise.z_act) S.noise.z_act = smem + m.z_act;
    if (A.noise.u_cmd) S.noise.u_cmd = smem + m.u_cmd;
    if (A.noise.u_dof) S.noise.u_dof = smem + m.u_dof;
    if (A.noise.u_push) S.noise.u_push = smem + m.u_push;
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

template <int E_T>
HG_HD void env_stage_in(const EnvArgs& A, int block, int t, int nthreads, float* smem) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs, e0 = block * E;
    const int nE = (E < N - e0) ? E : (N - e0);
    const LdsMap m = lds_map(E);
    int o = m.state;
    for (int f = 0; f < kNumStateFields; ++f) {
        const int nc = state_field_comps(f);
        const float* g = *state_field_ptr(A.st, f);
        for (int i = t; i < nc * E; i += nthreads) {
            const int c = i / E, le = i - c * E;
            smem[o + i] = (le < nE) ? g[(int64_t)c * N + e0 + le] : 0.0f;
        }
        o += nc * E;
    }
    {   // sim tensors through their strides; only the components the step reads
        const HgymStrided* sv[3] = {&A.sim.root, &A.sim.dof_pos, &A.sim.dof_vel};
        const int nc[3] = {13, 12, 12};
        const int off[3] = {m.root, m.dof_pos, m.dof_vel};
        for (int k = 0; k < 3; ++k)
            for (int i = t; i < nc[k] * E; i += nthreads) {
                const int c = i / E, le = i - c * E;
                smem[off[k] + i] = (le < nE) ? sget(*sv[k], e0 + le, c) : 0.0f;
            }
        for (int i = t; i < 9 * E; i += nthreads) {
            const int c = i / E, le = i - c * E;
            smem[m.contact + i] = (le < nE) ? sget(A.sim.contact, e0 + le, A.contact_comp[c / 3] + c % 3) : 0.0f;
        }
        for (int i = t; i < 14 * E; i += nthreads) {   // feet {x,y,z,vx,vy}, knees {x,y}
            const int q = i / E, le = i - q * E;
            const int body = q < 10 ? q / 5 : 2 + (q - 10) / 2;
            const int comp = q < 10 ? kFootRigidComps[q % 5] : kKneeRigidComps[(q - 10) % 2];
            smem[m.rigid + (body * 13 + comp) * E + le] = (le < nE) ? sget(A.sim.rigid, e0 + le, A.rigid_comp[body] + comp) : 0.0f;
        }
    }
    for (int i = t; i < nE; i += nthreads) reinterpret_cast<int64_t*>(smem + m.ep_len)[i] = A.st.episode_length[e0 + i];
    copy_rows_in(A.actions_in, smem + m.actions_in, 12, e0, nE, t, nthreads);
    copy_rows_in(A.noise.u_delay, smem + m.u_delay, 1, e0, nE, t, nthreads);
    copy_rows_in(A.noise.z_act, smem + m.z_act, 12, e0, nE, t, nthreads);
    copy_rows_in(A.noise.u_cmd, smem + m.u_cmd, 6, e0, nE, t, nthreads);
    copy_rows_in(A.noise.u_dof, smem + m.u_dof, 12, e0, nE, t, nthreads);
    copy_rows_in(A.noise.u_push, smem + m.u_push, 5, e0, nE, t, nthreads);
}

template <int E_T>
HG_HD void env_step_phase_a(const EnvArgs& A, int block, int t, float* smem, int64_t csc0) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs, e0 = block * E;
    const int nE = (E < N - e0) ? E : (N - e0);
    if (t >= nE) return;
    const LdsMap m = lds_map(E);
    const EnvArgs S = make_shadow(A, smem, block, E);
    const RngKey rk = make_rng_key(A, csc0);
    if (A.mode == MODE_STEP && A.fused && !(A.ablate & 16)) {
        pre_physics_env(S, rk, t, E);
        synth_physics_env(S, rk, t, E);
    }
    StepFlags fl;
    fl.reset = 0;
    if (!(A.ablate & 32))
        fl = post_physics_env(S, rk, csc0 + 1, t, E, smem + m.frame + t * HGYM_OBS_FRAME, smem + m.priv + t * HGYM_PRIV_FRAME);
    reinterpret_cast<int*>(smem + m.reset_i)[t] = fl.reset;
}

template <int E_T>
HG_HD void env_stage_out(const EnvArgs& A, int block, int t, int nthreads, float* smem) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs, e0 = block * E;
    const int nE = (E < N - e0) ? E : (N - e0);
    const LdsMap m = lds_map(E);
    int o = m.state;
    for (int f = 0; f < kFirstConstField; ++f) {
        const int nc = state_field_comps(f);
        float* g = *state_field_ptr(A.st, f);
        for (int i = t; i < nc * E; i += nthreads) {
            const int c = i / E, le = i - c * E;
            if (le < nE) g[(int64_t)c * N + e0 + le] = smem[o + i];
        }
        o += nc * E;
    }
    {
        const HgymStrided* sv[3] = {&A.sim.root, &A.sim.dof_pos, &A.sim.dof_vel};
        const int nc[3] = {13, 12, 12};
        const int off[3] = {m.root, m.dof_pos, m.dof_vel};
        for (int k = 0; k < 3; ++k)
            for (int i = t; i < nc[k] * E; i += nthreads) {
                const int c = i / E, le = i - c * E;
                if (le < nE) sset(*sv[k], e0 + le, c, smem[off[k] + i]);
            }
        if (A.mode == MODE_STEP && A.fused) {   // the synthetic physics wrote contacts and rigid-body entries
            for (int i = t; i < 9 * E; i += nthreads) {
                const int c = i / E, le = i - c * E;
                if (le < nE) sset(A.sim.contact, e0 + le, A.contact_comp[c / 3] + c % 3, smem[m.contact + i]);
            }
            for (int i = t; i < 14 * E; i += nthreads) {
                const int q = i / E, le = i - q * E;
                const int body = q < 10 ? q / 5 : 2 + (q - 10) / 2;
                const int comp = q < 10 ? kFootRigidComps[q % 5] : kKneeRigidComps[(q - 10) % 2];
                if (le < nE) sset(A.sim.rigid, e0 + le, A.rigid_comp[body] + comp, smem[m.rigid + (body * 13 + comp) * E + le]);
            }
        }
    }
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

template <int H_T, int HC_T, int E_T>
HG_HD void env_step_phase_b(const EnvArgs& A, int block, int t, int nthreads, float* smem, int64_t csc0, int64_t ring_step) {
    const int E = E_T > 0 ? E_T : A.envs_per_block;
    const int N = A.cfg.num_envs;
    const int H = H_T > 0 ? H_T : A.cfg.frame_stack;
    const int HC = HC_T > 0 ? HC_T : A.cfg.c_frame_stack;
    const int e0 = block * E;
    const int nE = (E < N - e0) ? E : (N - e0);
    const LdsMap m = lds_map(E);
    const float* s_frame = smem + m.frame;
    const float* s_priv = smem + m.priv;
    const int* s_reset = reinterpret_cast<const int*>(smem + m.reset_i);
    const RngKey rk = make_rng_key(A, csc0);
    if (A.mode == MODE_RESET_ALL) {   // reset_idx(all) without compute_observations: just clear the history
        const int64_t no = (int64_t)nE * H * HGYM_OBS_FRAME, np = (int64_t)nE * HC * HGYM_PRIV_FRAME;
        float* ro = A.st.obs_ring + (int64_t)e0 * H * HGYM_OBS_FRAME;
        float* rp = A.st.priv_ring + (int64_t)e0 * HC * HGYM_PRIV_FRAME;
        for (int64_t i = t; i < no; i += nthreads) ro[i] = 0.0f;
        for (int64_t i = t; i < np; i += nthreads) rp[i] = 0.0f;
        return;
    }
    // actor observations (N, H*47) and privileged observations (N, HC*73): one contiguous run per workgroup
    stack_rows<true>(A, rk, A.st.obs_ring, s_frame, s_reset, A.out.obs + (int64_t)e0 * H * HGYM_OBS_FRAME, e0, nE, H, HGYM_OBS_FRAME,
                     (int)(ring_step % H), t, nthreads);
    stack_rows<false>(A, rk, A.st.priv_ring, s_priv, s_reset, A.out.priv_obs + (int64_t)e0 * HC * HGYM_PRIV_FRAME, e0, nE, HC,
                      HGYM_PRIV_FRAME, (int)(ring_step % HC), t, nthreads);
}

// Step finaliser: the cross-env pieces of reset_idx (legged_robot.py:199-210) -- means of the episode sums
// over the envs that reset, extras["time_outs"] refreshed only when >= 1 env reset (the reference's
// stale-extras behaviour) -- then (after a barrier) the device-resident counters.
HG_HD void env_finalize_part1(const EnvArgs& A, int t, int nthreads) {
    const int N = A.cfg.num_envs;
    const int64_t cnt = A.st.counters[1];
    if (cnt > 0) {
        if (t < HGYM_NUM_REWARDS) {
            A.out.extras_episode[t] = A.st.episode_acc[t] / (float)cnt / A.cfg.episode_length_s;
            A.st.episode_acc[t] = 0.0f;
        }
        for (int i = t; i < N; i += nthreads) A.out.extras_time_outs[i] = A.out.time_out[i];
    }
}
HG_HD void env_finalize_part2(const EnvArgs& A) {
    A.st.counters[1] = 0;
    if (A.mode == MODE_STEP) A.st.counters[0] += 1;
    if (A.mode != MODE_RESET_ALL) A.st.counters[2] += 1;
}

#undef FG
}  // namespace hgym
