// hgym_finalize.hpp -- the step finaliser: the cross-env pieces of one env step, which need every env's result.
//   * extras["episode"][k] = mean over the envs that reset this step of their episode sums / episode_length_s, and
//     extras["time_outs"] = this step's time_out_buf -- both refreshed ONLY when at least one env reset (the reference
//     re-assigns them inside reset_idx: legged_robot.py:173-174, 199-210);
//   * optionally (HgymEnvOut transition sink) PPO.process_env_step's scalar columns, hgym_store_step's arithmetic;
//   * the device-resident step counters.
// Shared by two hosts: env_finalize_kernel (its own launch, hgym_env.hip) and the fused-forward kernel, where the finaliser
// of vec-step t rides as ONE extra workgroup of the policy launch of step t+1 (HgymEnvOut.defer_finalize): the kernel
// boundary between the env step and that launch is the only synchronisation it needs, and a 5 us single-workgroup launch
// leaves the rollout's critical path.  (Folding it into the env kernel itself with a last-workgroup-done ticket needs an
// agent-scope release per workgroup -- an L2 write-back on every XCD -- and measured 52 us against 31.5 us.)
#pragma once
#include "hgym_common.hpp"

namespace hgym {

enum { FIN_MODE_STEP = 0, FIN_MODE_PRIME = 1, FIN_MODE_RESET_ALL = 2 };   // = MODE_* of hgym_env_math.hpp

struct FinArgs {
    int N;
    int mode;
    float episode_length_s;
    int64_t* counters;       // HgymEnvState::counters
    int64_t* reset_count;    // envs that reset in the step being finalised: &counters[1], or the step's own slot when the finaliser
                             // runs concurrently with the NEXT step's env phase (rollout_step_kernel)
    float* episode_acc;      // HgymEnvState::episode_acc (same remark)
    int ncustom;             // user-defined reward terms (HgymEnvConfig::num_custom_rewards)
    float* custom_acc;       // HgymEnvState::custom_acc
    HgymEnvOut out;
};

HG_HD FinArgs make_fin_args(const HgymEnvConfig& cfg, const HgymEnvState& st, const HgymEnvOut& out, int mode) {
    FinArgs f;
    f.N = cfg.num_envs;
    f.mode = mode;
    f.episode_length_s = cfg.episode_length_s;
    f.counters = st.counters;
    f.reset_count = st.counters + 1;
    f.episode_acc = st.episode_acc;
    f.ncustom = (st.custom_acc && out.extras_custom) ? cfg.num_custom_rewards : 0;
    f.custom_acc = st.custom_acc;
    f.out = out;
    return f;
}

HG_HD void fin_part1(const FinArgs& F, int t, int nthreads) {
    const int64_t cnt = F.reset_count[0];
    if (cnt > 0) {
        if (t < HGYM_NUM_REWARDS) {
            F.out.extras_episode[t] = F.episode_acc[t] / (float)cnt / F.episode_length_s;
            F.episode_acc[t] = 0.0f;
        }
        if (t < F.ncustom) {
            F.out.extras_custom[t] = F.custom_acc[t] / (float)cnt / F.episode_length_s;
            F.custom_acc[t] = 0.0f;
        }
        for (int i = t; i < F.N; i += nthreads) F.out.extras_time_outs[i] = F.out.time_out[i];
    }
}
// optional transition sink (HgymEnvOut::t_*).  Thread t touches exactly the elements it refreshed in part 1.
HG_HD void fin_store(const FinArgs& F, int t, int nthreads) {
#pragma clang fp contract(off)      // rew + gamma * (V * to) as three fp32 roundings, in every translation unit
    if (!F.out.t_rewards) return;
    for (int i = t; i < F.N; i += nthreads) {
        const float to = (float)(F.out.extras_time_outs[i] != 0);
        const float boot = F.out.t_values[i] * to;
        const float gb = F.out.t_gamma * boot;
        F.out.t_rewards[i] = F.out.rew[i] + gb;
        F.out.t_dones[i] = F.out.reset[i] != 0;
    }
}
HG_HD void fin_part2(const FinArgs& F) {
    // with a deferred finaliser the env kernel itself bumps the policy's sampling step (the policy launch this rides in
    // reads it at entry)
    if (F.out.t_rewards && F.out.t_step && !F.out.defer_finalize) F.out.t_step[0] += 1;
    F.reset_count[0] = 0;
    if (F.mode == FIN_MODE_STEP) F.counters[0] += 1;
    if (F.mode != FIN_MODE_RESET_ALL) F.counters[2] += 1;
}

// Optional logging sink (HgymEnvOut::log_*): OnPolicyRunner.learn's per-step book-keeping (on_policy_runner.py:143-156).
// Thread t refreshed extras_episode[t] in part 1, so it may add it up here without a barrier.  Finished episodes are appended to
// the two 100-entry rings in env order (the reference extends its deques with cur_reward_sum[new_ids], ascending ids): envs are
// walked in rows of `nthreads`, a workgroup-wide exclusive scan of the done flags gives each its place.
__device__ __forceinline__ void fin_log(const FinArgs& F, int t, int nthreads) {
    float* LS = F.out.log_stats;
    if (!LS) return;
    float* cur = F.out.log_cur;
    __shared__ int s_wave[32];
    __shared__ int s_head;
    if (t < HGYM_NUM_REWARDS) LS[t] += F.out.extras_episode[t];
    if (t == 0) {
        LS[22] += 1.0f;
        s_head = (int)LS[24];
    }
    const int lane = t & 63, wave = t >> 6, nw = (nthreads + 63) >> 6;
    int appended = 0;                    // finished episodes of the rows walked so far (uniform)
    for (int row0 = 0; row0 < F.N; row0 += nthreads) {
        const int i = row0 + t;
        const bool in = i < F.N;
        const bool done = in && F.out.reset[i] != 0;
        float r = 0.0f, l = 0.0f;
        if (in) {
            r = cur[i] + F.out.rew[i];
            l = cur[F.N + i] + 1.0f;
        }
        const unsigned long long b = __ballot(done);
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(b);
        __syncthreads();
        int off = before, total = 0;
        for (int w = 0; w < nw; ++w) {
            const int c = s_wave[w];
            if (w < wave) off += c;
            total += c;
        }
        if (done) {
            if (off >= total - 100) {    // more than 100 in one row: only the last 100 can survive (and keep their slots distinct)
                const int pos = (s_head + appended + off) % 100;
                LS[32 + pos] = r;
                LS[132 + pos] = l;
            }
            r = 0.0f;
            l = 0.0f;
        }
        if (in) {
            cur[i] = r;
            cur[F.N + i] = l;
        }
        appended += total;
        __syncthreads();
    }
    if (t == 0) {
        LS[24] = (float)((s_head + appended) % 100);
        const float filled = LS[25] + (float)appended;
        LS[25] = filled > 100.0f ? 100.0f : filled;
    }
}

// all of it, for one workgroup of `nthreads` lanes
__device__ __forceinline__ void fin_block(const FinArgs& F, int t, int nthreads) {
    fin_part1(F, t, nthreads);
    __syncthreads();
    fin_store(F, t, nthreads);
    fin_log(F, t, nthreads);
    if (t == 0) fin_part2(F);
}

}  // namespace hgym
