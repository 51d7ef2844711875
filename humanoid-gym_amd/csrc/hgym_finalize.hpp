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
    if (F.out.t_time_outs) {        // deferred values: raw reward + the flags of the bootstrap (hgym_gae_bootstrap applies it)
        for (int i = t; i < F.N; i += nthreads) {
            F.out.t_time_outs[i] = F.out.extras_time_outs[i] != 0;
            F.out.t_rewards[i] = F.out.rew[i];
            F.out.t_dones[i] = F.out.reset[i] != 0;
        }
        return;
    }
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

// fin_log in ONE memory round trip (round 5; N a multiple of 8, 16-byte aligned columns: otherwise false, nothing touched).  The loop above
// walks the envs in rows of `nthreads` with two barriers and a dependent load -> scan -> store chain per row: 8 rows for 4096 envs on 512
// lanes, ~9 us of the ONE workgroup that rides in the third grid row of the next rollout launch -- and that workgroup starts late (it
// needs a compute unit one of the launch's 256 tiles has left), so with the log sink bound every launch of the rollout ended ~6.7 us later
// (BENCH_r04 configs[logging_on]: collection 2.59 vs 2.19 ms).  Here a lane owns 8 CONSECUTIVE envs, issues every load it needs up front
// and the workgroup scans the per-lane counts once: same ring order (ascending env index), same values, same roundings.
__device__ __forceinline__ bool fin_log_fused(const FinArgs& F, int t, int nthreads) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned long long u64;
    const HgymEnvOut& O = F.out;
    float* LS = O.log_stats;
    if (!LS) return true;
    float* cur = O.log_cur;
    auto al = [](const void* p, int a) { return ((uintptr_t)p & (uintptr_t)(a - 1)) == 0; };
    if ((F.N & 7) != 0 || !al(O.reset, 8) || !al(O.rew, 16) || !al(cur, 16)) return false;
    __shared__ int s_wave[32];
    __shared__ int s_head;
    if (t < HGYM_NUM_REWARDS) LS[t] += O.extras_episode[t];      // (thread t refreshed extras_episode[t] itself: no barrier)
    if (t == 0) {
        LS[22] += 1.0f;
        s_head = (int)LS[24];
    }
    const int lane = t & 63, wave = t >> 6, nw = (nthreads + 63) >> 6;
    const int G = F.N >> 3;
    int appended = 0;
    for (int g0 = 0; g0 < G; g0 += nthreads) {
        const int g = g0 + t;
        const bool in = g < G;
        u64 rs = 0;
        f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        f4 c[2] = {z, z}, l[2] = {z, z}, rw[2] = {z, z};
        if (in) {
            rs = reinterpret_cast<const u64*>(O.reset)[g];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                c[h] = reinterpret_cast<const f4*>(cur)[2 * g + h];
                l[h] = reinterpret_cast<const f4*>(cur + F.N)[2 * g + h];
                rw[h] = reinterpret_cast<const f4*>(O.rew)[2 * g + h];
            }
        }
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            c[k >> 2][k & 3] = c[k >> 2][k & 3] + rw[k >> 2][k & 3];
            l[k >> 2][k & 3] = l[k >> 2][k & 3] + 1.0f;
            cnt += (int)(((rs >> (8 * k)) & 0xffull) != 0);
        }
        int incl = cnt;                                        // inclusive scan of the lanes' counts over the wavefront ...
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int off = incl - cnt, total = 0;                       // ... and over the workgroup
        for (int w = 0; w < nw; ++w) {
            const int cw = s_wave[w];
            if (w < wave) off += cw;
            total += cw;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (((rs >> (8 * k)) & 0xffull) != 0) {
                if (off >= total - 100) {                      // more than 100 in one trip: only the last 100 can survive (distinct slots)
                    const int pos = (s_head + appended + off) % 100;
                    LS[32 + pos] = c[k >> 2][k & 3];
                    LS[132 + pos] = l[k >> 2][k & 3];
                }
                ++off;
                c[k >> 2][k & 3] = 0.0f;
                l[k >> 2][k & 3] = 0.0f;
            }
        }
        if (in) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                reinterpret_cast<f4*>(cur)[2 * g + h] = c[h];
                reinterpret_cast<f4*>(cur + F.N)[2 * g + h] = l[h];
            }
        }
        appended += total;
        __syncthreads();
    }
    if (t == 0) {
        LS[24] = (float)((s_head + appended) % 100);
        const float filled = LS[25] + (float)appended;
        LS[25] = filled > 100.0f ? 100.0f : filled;
    }
    return true;
}

// fin_part1 + fin_store in ONE memory round trip (N a multiple of 8, 16-byte aligned columns; otherwise returns false and nothing is
// touched).  The two functions above are loops of load -> store over byte pointers: every load waits for the store before it (a
// uint8_t store may alias anything), 8 + 8 dependent round trips for 4096 envs on 512 lanes -- 8 us of a single workgroup that
// rides behind the critic tiles of the rollout launch and ended that launch.  Here a lane owns 8 consecutive envs, issues every
// load it needs (the reset count, its 8 time-out / reset / stale time-out bytes, its 8 values and rewards) before its first store,
// and thread t still refreshes exactly the elements it reads back (no barrier).  Same arithmetic, same roundings.
HG_HD bool fin_fused(const FinArgs& F, int t, int nthreads) {      // (host-compilable: tests/hostcheck runs it against fin_part1 + fin_store)
#pragma clang fp contract(off)
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned long long u64;
    const HgymEnvOut& O = F.out;
    auto al = [](const void* p, int a) { return ((uintptr_t)p & (uintptr_t)(a - 1)) == 0; };
    const bool sink = O.t_rewards != nullptr;
    const bool deferred = sink && O.t_time_outs != nullptr;       // raw reward + bootstrap flags instead of the bootstrapped reward
    if ((F.N & 7) != 0 || !al(O.time_out, 8) || !al(O.extras_time_outs, 8) ||
        (sink && (!al(O.reset, 8) || !al(O.t_dones, 8) || !al(O.rew, 16) || !al(O.t_rewards, 16))) ||
        (sink && !deferred && !al(O.t_values, 16)) || (deferred && !al(O.t_time_outs, 8)))
        return false;
    const int G = F.N >> 3;
    const int64_t cnt = F.reset_count[0];
    float acc = 0.0f, cacc = 0.0f;
    if (t < HGYM_NUM_REWARDS) acc = F.episode_acc[t];
    if (t < F.ncustom) cacc = F.custom_acc[t];
    for (int g0 = 0; g0 < G; g0 += nthreads) {          // (4096 envs on 512 lanes: one trip)
        const int g = g0 + t;
        if (g >= G) break;
        const u64 to_new = reinterpret_cast<const u64*>(O.time_out)[g];
        const u64 to_old = reinterpret_cast<const u64*>(O.extras_time_outs)[g];
        u64 rs = 0;
        f4 v0 = {0.0f, 0.0f, 0.0f, 0.0f}, v1 = v0, r0 = v0, r1 = v0;
        if (sink) {
            rs = reinterpret_cast<const u64*>(O.reset)[g];
            if (!deferred) {
                v0 = reinterpret_cast<const f4*>(O.t_values)[2 * g];
                v1 = reinterpret_cast<const f4*>(O.t_values)[2 * g + 1];
            }
            r0 = reinterpret_cast<const f4*>(O.rew)[2 * g];
            r1 = reinterpret_cast<const f4*>(O.rew)[2 * g + 1];
        }
        const u64 to = cnt > 0 ? to_new : to_old;
        if (cnt > 0) reinterpret_cast<u64*>(O.extras_time_outs)[g] = to_new;
        if (sink) {
            f4 o0, o1;
            u64 dn = 0, tn = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool tob = ((to >> (8 * k)) & 0xffull) != 0;
                const float tof = (float)tob;
                const float boot = (k < 4 ? v0[k & 3] : v1[k & 3]) * tof;
                const float gb = O.t_gamma * boot;
                const float raw = k < 4 ? r0[k & 3] : r1[k & 3];
                const float r = deferred ? raw : raw + gb;
                if (k < 4) o0[k & 3] = r;
                else o1[k & 3] = r;
                dn |= (u64)(((rs >> (8 * k)) & 0xffull) != 0) << (8 * k);
                tn |= (u64)tob << (8 * k);
            }
            if (deferred) reinterpret_cast<u64*>(O.t_time_outs)[g] = tn;
            reinterpret_cast<f4*>(O.t_rewards)[2 * g] = o0;
            reinterpret_cast<f4*>(O.t_rewards)[2 * g + 1] = o1;
            reinterpret_cast<u64*>(O.t_dones)[g] = dn;
        }
    }
    if (cnt > 0) {
        if (t < HGYM_NUM_REWARDS) {
            O.extras_episode[t] = acc / (float)cnt / F.episode_length_s;
            F.episode_acc[t] = 0.0f;
        }
        if (t < F.ncustom) {
            O.extras_custom[t] = cacc / (float)cnt / F.episode_length_s;
            F.custom_acc[t] = 0.0f;
        }
    }
    return true;
}

// all of it, for one workgroup of `nthreads` lanes
__device__ __forceinline__ void fin_block(const FinArgs& F, int t, int nthreads) {
    // fin_part2's counters travel with the first loads (nothing in between writes them): no round trip of their own at the end
    const bool bump_step = F.out.t_rewards && F.out.t_step && !F.out.defer_finalize;
    int64_t c0 = 0, c2 = 0, ts = 0;
    if (t == 0) {
        c0 = F.counters[0];
        c2 = F.counters[2];
        if (bump_step) ts = F.out.t_step[0];
    }
    if (!fin_fused(F, t, nthreads)) {
        fin_part1(F, t, nthreads);
        __syncthreads();
        fin_store(F, t, nthreads);
    }
    if (!fin_log_fused(F, t, nthreads)) fin_log(F, t, nthreads);
    // every lane of the workgroup has loaded reset_count[0] (fin_fused's / fin_part1's first load) before thread 0 zeroes it: without
    // this barrier a late wavefront of a 512 / 1024-lane workgroup could read 0 and keep the stale time-out bytes
    __syncthreads();
    if (t == 0) {
        // = fin_part2, on the values loaded above
        if (bump_step) F.out.t_step[0] = ts + 1;
        F.reset_count[0] = 0;
        if (F.mode == FIN_MODE_STEP) F.counters[0] = c0 + 1;
        if (F.mode != FIN_MODE_RESET_ALL) F.counters[2] = c2 + 1;
    }
}

}  // namespace hgym
