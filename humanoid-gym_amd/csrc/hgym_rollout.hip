// hgym_rollout.hip -- ONE launch per vec-step of the rollout: PPO.act, the env step and the previous step's finaliser.
//
// The rollout is a latency chain: [policy step -> env step] x 60, each link a kernel whose first act is
// a round trip to memory for its inputs and whose launch costs ~4 us before any wave runs.  Nothing in step t of env e depends
// on another env's step t -- the only cross-env pieces are the step finaliser's means, which already ride one launch behind
// (HgymEnvOut.defer_finalize).  So the two links are fused per env slice:
//
//   grid (N / 32, 2 [+ 1])    blockIdx.y = 0: ACTOR tile of 32 rows (mlp_fwd body, 8 wavefronts) and then the ENV STEP of the same
//                                            32 envs, fed with the tile's sampled actions through LDS;
//                             blockIdx.y = 1: CRITIC tile (values of the transition being collected);
//                             blockIdx.y = 2: the finaliser of the PREVIOUS env step, one workgroup.
//
//   actor workgroup timeline  issue {bias, first weight k-steps, first input chunk} | issue the env step's state / sim loads |
//   Philox draws of the env step (ALU under all of those loads) | layer 0 | issue the loads of the 14 + 2 older history frames
//   of the 32 envs (registers; they have the rest of the tile to arrive) | layers 1, 2, head, sampling -> actions to HBM
//   (storage slot) and to LDS | older frames -> stacked outputs | joints | per-env chain (wavefront 0) | state write-back,
//   newest frame, reset fix-ups.
//
// The env phases are the functions of hgym_env_math.hpp that env_step_kernel runs (32 envs per workgroup, 512 lanes), so the
// arithmetic -- and hence every mask and every float -- is the unfused kernel's.  This translation unit is built with
// -ffp-contract=off like hgym_env.hip; hgym_fused.hpp restores the policy kernels' own setting for its part.
//
// Step counters.  env_step_kernel reads the common step counter / ring step from HgymEnvState::counters and the policy reads its
// sampling step from *step_counter; here the finaliser that bumps the former runs CONCURRENTLY (it belongs to the previous
// step), so the launch takes its three counters from a ping-pong record in the caller's scratch block: launch t reads
// pp[parity], workgroup (0, 0) writes pp[parity ^ 1] = pp[parity] + 1 for launch t + 1.  For the same reason the reset count
// and the episode-sum accumulators the finaliser consumes, and the rew / reset / time_out outputs it reads, are per parity:
// the caller hands two HgymEnvOut records (this step's, the previous step's) with distinct rew / reset / time_out buffers.
#define HGYM_TU_CONTRACT_OFF 1
#include <stdlib.h>

#include <algorithm>

#include "hgym_env_math.hpp"
#include "hgym_fused.hpp"


namespace hgym {

int32_t rollout_fwd_args(const HgymNetConfig* cfg, const HgymNet* net, int M, const float* obs, const float* priv, uint64_t seed,
                         const int64_t* step, float* actions, float* mu, float* sigma, float* logp, float* values, FwdArgs* out,
                         size_t* lds_bytes, const HgymObsShadow* sh);
int32_t rollout_env_args(const HgymEnvConfig* cfg, const HgymSimTensors* sim, const HgymEnvState* st, const HgymEnvOut* out,
                         float* actions, EnvArgs* A);

constexpr int RO_E = 32;       // envs (= policy rows) per workgroup
constexpr int RO_NT = 512;     // lanes per workgroup: 8 wavefronts, as mlp_fwd_kernel<32, 8, 4>

// caller's scratch block (HGYM_ROLLOUT_SCRATCH_BYTES(num_envs), zero-filled once): this header, then two per-parity images of the
// env step's draw tables ([tile][125 floats x 32 envs], the layout of the LDS noise tables u_delay .. phys)
struct RolloutScratch {
    int64_t pp[2][4];          // [parity]{common step counter, ring step, sampling step, -}
    int64_t reset_cnt[2];      // [parity] envs that reset in the step of that parity
    int64_t pad[6];
    float acc[2][24];          // [parity] episode-sum accumulators of that step (HgymEnvState::episode_acc layout)
};
static_assert(sizeof(RolloutScratch) <= HGYM_ROLLOUT_SCRATCH_HEADER_BYTES, "scratch block too small");

// The three argument records are SEPARATE kernel parameters: as members of one 3 KB struct the compiler, past some size of the
// kernel body, stopped seeing that the argument block is only read and kept a private-memory copy of all of it.
struct RolloutPP {
    const int64_t* in;         // {common step counter, ring step, sampling step} this launch works with
    int64_t* out;              // the same + 1, written by workgroup (0, 0) for the next launch
    int env_lds_off;           // byte offset of the env image in dynamic LDS (behind the policy tile's buffers)
    // The env step's Philox draws depend on (seed, step, env) only, so the draws of step t + 1 are computed during step t by the
    // CRITIC workgroup of the same tile -- idle for the second half of the launch -- and handed over through global memory:
    // draws_out = where this launch leaves the tables of the next step, draws_in = the tables of this step (null: the first step
    // of a rollout, the actor workgroup computes them itself on its idle wavefronts).  Layout: tile-major, each tile the
    // contiguous LDS noise-table region [u_delay .. phys] of lds_map(32).
    const float* draws_in;
    float* draws_out;
    int draws_len;             // floats per tile
    // The actor's first layer carried across launches (hgym_fused.hpp: L0Part / L0Ahead): l0.acc = what the previous launch's critic
    // workgroups left for this step (PART instantiation), ah.acc_out = where this launch's leave the next step's (null: not).
    L0Part l0;
    L0Ahead ah;
    const uint8_t* prev_reset;   // reset flags of the previous step (prev_out->reset; null: first step of a rollout), see the rows-after-next note below
};

constexpr int RO_NIO = hist_ni<15, HGYM_OBS_FRAME, RO_E, RO_NT>();
constexpr int RO_NIP = hist_ni<3, HGYM_PRIV_FRAME, RO_E, RO_NT>();
// The rows after next (obs_ahead / priv_ahead: 13 + 1 older frames per env, ring -> rows, 78 KB per tile) are copied by the tile's CRITIC
// workgroup at its start, not by the idle wavefronts of the actor workgroup's per-env phase -- in a run of launches that phase waited
// 7 us for the copy's loads and acknowledged stores, against 4.5 us for its own arithmetic, and the critic workgroup ends ~3 us before
// the actor's.  The copy reads pre-reset history for an env that resets in THIS step; nobody reads those rows before the next launch,
// which zeroes them (prev_reset) -- the kernel boundary orders the two workgroups' stores to the same addresses.  (Measured and dropped:
// the copy on the actor workgroup's idle wavefronts, its stores as the launch's last instructions, LDS-only barriers around the per-env
// phase -- profiles/r04_rollout_env_part_findings.txt.)
// The staging phases (env_stage_in / env_stage_out) are instantiated for the one layout this launch accepts (rollout_env_args refuses
// every other): their general paths are compiled out -- 122 -> 108 KB of code, collection 2.99 -> 2.96 ms in a same-call A/B
// (profiles/r05a_bench_ab_base_rofast_dw32.txt).
constexpr int RO_NIA_C = hist_ni<15, HGYM_OBS_FRAME, RO_E, RO_NT, 2>();
constexpr int RO_NIAP_C = hist_ni<3, HGYM_PRIV_FRAME, RO_E, RO_NT, 2>();
constexpr int RO_CHAIN = 64 * kChainRoles;     // lanes of the per-env chain: four wavefronts by role (env_step_phase_a3)

// PRE (HgymEnvOut.obs_older_ready): the 14 older frames of this launch's stacked observation rows were written by the previous launch
// (as its obs_ahead), so the copy ring -> rows -- 11 HBM loads per lane issued after the first layer, in front of the second layer's
// weight ring in the in-order vmcnt queue, and their stores: 4.4 us of a 42 us launch -- is not in this kernel at all.  A launch
// that is given obs_ahead writes the 13 frames it already knows of the rows after next on the seven wavefronts that idle during
// the per-env phase, and this step's frame next to its own row's in the stack phase.
// NOCRITIC (hgym_rollout_step with values = NULL, header v7): no critic tiles -- grid rows = the actor + env workgroups and the
// finaliser; the critic runs once over the stored rows after the rollout (hgym_critic_values).  With PRE = PART = false the actor
// workgroup draws its own random numbers and copies its own history rows, as in the first launch of a rollout.
template <bool FIN, bool PRE, bool PART = false, bool NOCRITIC = false>
__global__ __launch_bounds__(RO_NT) void rollout_step_kernel(const FwdArgs f, const EnvArgs e, const FinArgs fin, const RolloutPP pp) {
    static_assert(!NOCRITIC || (!PRE && !PART), "rows ahead and the carried first layer are the critic workgroups' side jobs");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (FIN && blockIdx.y >= (NOCRITIC ? 1 : 2)) {
        // (phase clock: slot 6 of the env row = when this workgroup of the third grid row started, slot 7 of block 0 = the finaliser's end)
        long long* d2 = f.dbg ? f.dbg + ((int64_t)2 * gridDim.x + blockIdx.x) * 8 : nullptr;
        if (d2 && threadIdx.x == 0) d2[6] = (long long)__builtin_amdgcn_s_memrealtime();
        if (blockIdx.x == 0) {
            fin_block(fin, threadIdx.x, RO_NT);
            if (d2 && threadIdx.x == 0) d2[7] = (long long)__builtin_amdgcn_s_memrealtime();
        }
        return;
    }
    constexpr int U = 16 / 8;                       // n-blocks per wave per 256 first-layer columns (mlp_fwd_kernel)
    // actor and critic workgroups alternate in dispatch order (tile b: row 0 holds its actor when b is even, its critic when b
    // is odd), so that the long actor + env workgroups are spread evenly over neighbouring compute units
    const bool critic_wg = !NOCRITIC && ((blockIdx.x + blockIdx.y) & 1) != 0;
    if (critic_wg) {                                // critic tile
        // one instantiation only (first hidden layer 768 wide, rollout_fwd_args checks): with the three-way dispatch of
        // mlp_fwd_kernel next to the actor + env branch the compiler keeps a private-memory copy of the whole 3 KB argument
        fwd_body<32, 8, 4, 3 * U>(f, f.net[1], false, smem);
        // rows after next of this tile (HGYM_RO_AHEAD_CRITIC): loads issued here, behind the tile, where the launch's first rush on memory is
        // over; they travel while the draws are computed; the stores' acknowledgements are waited for under the first-layer weights below
        float ha[RO_NIA_C][4], hp[RO_NIAP_C][4];
        const int ring_s = (int)pp.in[1];
        // (unconditional: the ring always exists; a launch without rows after next drops them)
        hist_load<15, HGYM_OBS_FRAME, RO_NIA_C, 2>(e.st.obs_ring, (int)blockIdx.x * RO_E, RO_E, ring_s % 15, (int)threadIdx.x, RO_NT, ha);
        hist_load<3, HGYM_PRIV_FRAME, RO_NIAP_C, 2>(e.st.priv_ring, (int)blockIdx.x * RO_E, RO_E, ring_s % 3, (int)threadIdx.x, RO_NT, hp);
        if (pp.draws_out) {      // next step's draw tables of this tile (step counter + 1), written where the LDS tables would be
            float* base = pp.draws_out + (int64_t)blockIdx.x * pp.draws_len - lds_map(RO_E).u_delay;
            env_fill_draws<RO_E>(e, (int)blockIdx.x, (int)threadIdx.x, RO_NT, base, pp.in[0] + 1);
        }
        if (e.out.obs_ahead) {
            hist_store<15, HGYM_OBS_FRAME, RO_NIA_C, 2>(e.out.obs_ahead, (int)blockIdx.x * RO_E, RO_E, ring_s % 15, (int)threadIdx.x, RO_NT, nullptr,
                                                        e.cfg.clip_obs, ha);
            hist_store<3, HGYM_PRIV_FRAME, RO_NIAP_C, 2>(e.out.priv_ahead, (int)blockIdx.x * RO_E, RO_E, ring_s % 3, (int)threadIdx.x, RO_NT, nullptr,
                                                         e.cfg.clip_obs, hp);
        }
        if (pp.ah.acc_out) {     // k-steps [0, kb0) of the ACTOR's first layer for the next step's rows of this tile
            __syncthreads();     // (the head wavefronts of this tile may still read its LDS)
            l0_partial_ahead<2 * U>(f.net[0], pp.ah, f.M, smem);
        }
        phase_stamp(f.dbg, 7);
        return;
    }
    const int t = threadIdx.x, block = blockIdx.x;
    const int64_t csc0 = pp.in[0], ring_step = pp.in[1], sstep = pp.in[2];
    float* esm = reinterpret_cast<float*>(smem + pp.env_lds_off);
    float hist_o[RO_NIO][4], hist_p[RO_NIP][4];
    const int act_off = lds_map(RO_E).actions_in;
    const float* const draws_in = pp.draws_in ? pp.draws_in + (int64_t)block * pp.draws_len : nullptr;
    const int draws_len = pp.draws_len;
    auto early = [&](const EnvArgs& E) {
        env_reset_pose<RO_E>(E, t, RO_NT, esm);                       // one lane, under the tile's first loads
        // this step's draw tables, computed during the previous launch: a plain copy that travels with the tile's first loads
        // (plain float quads and unconditional clamped loads: a packed-struct array behind a condition is kept in private memory)
        float dq[2][4];
        const int dq4 = draws_len >> 2, dq_off = lds_map(RO_E).u_delay;
        const float* const dsrc = draws_in ? draws_in : E.st.commands;      // any readable address when there is nothing to copy
        const int dmax = draws_in ? dq4 - 1 : 0;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = t + u * RO_NT;
            stage_ld(dq[u], dsrc + 4 * (i < dmax ? i : dmax));
        }
        if (t < 256) env_stage_in<RO_E, true>(E, block, t, 256, esm);      // travels with the tile's own first loads
        if (draws_in) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = t + u * RO_NT;
                if (i < dq4) stage_st(esm + dq_off + 4 * i, dq[u]);
            }
        }
    };
    auto mid = [&](const EnvArgs& E) {
        if (!PRE) hist_load<15, HGYM_OBS_FRAME, RO_NIO>(E.st.obs_ring, block * RO_E, RO_E, (int)(ring_step % 15), t, RO_NT, hist_o);
    };
    auto put = [&](int row, int j, float v) { esm[act_off + row * 12 + j] = v; };
    // the env step's Philox draws: on the six wavefronts that have no head block, while the other two compute the head
    auto idle = [&](const EnvArgs& E) {
        if (!draws_in) env_fill_draws<RO_E>(E, block, t - 128, RO_NT - 128, esm, csc0);
        // these lanes' share of the 14 older frames (in registers since `mid`) -> the stacked rows of the next observation, while
        // the two head wavefronts finish the tile: three quarters of that store phase leave the chain behind the tile
        if (!PRE) hist_store<15, HGYM_OBS_FRAME, RO_NIO>(E.out.obs, block * RO_E, RO_E, (int)(ring_step % 15), t, RO_NT, nullptr, E.cfg.clip_obs, hist_o);
    };
    fwd_body<32, 8, 4, 2 * U, false, false, PART>(f, f.net[0], true, smem, early, mid, put, e, idle, FwdNoop(), nullptr, nullptr, FwdNoop(), &pp.l0);
    if (PART && f.net[0].xs) {
        // rows of this tile whose env was reset by the previous step: columns [0, 32 kb0) of their bf16 shadow were written ahead from
        // the un-reset history -- the row's older frames are zero now (the launch that reset them zeroed the fp32 row)
        const int pieces = 4 * pp.l0.kb0;           // 16-byte pieces per row
        for (int j = t; j < RO_E * pieces; j += RO_NT) {
            const int row = j / pieces, pc = j - row * pieces;
            const int m = block * RO_E + row;
            if (m < f.M && pp.l0.reset[m]) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(f.net[0].xs + (int64_t)m * f.net[0].ldxs) + pc * 16) = (u32x4){0u, 0u, 0u, 0u};
        }
    }
    __syncthreads();                                // the tile's actions are in the env image; the policy buffers are dead
    // phase clock of the env part (hgym_prof_phase_buffer): the slots of grid row 2, which stamps nothing itself
    long long* dbg = f.dbg ? f.dbg + (int64_t)2 * gridDim.x * 8 + (int64_t)block * 8 : nullptr;
    auto stamp = [&](int slot) {
        if (dbg && t == 0) dbg[slot] = (long long)__builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    // did the previous step reset this lane's env of the tile?  Loaded HERE, consumed behind phase B: read there it
    // is a memory round trip at the very end of the workgroup
    const bool prev_rs = PRE && pp.prev_reset && (t & 63) < RO_E && pp.prev_reset[block * RO_E + (t & (RO_E - 1))] != 0;
    const EnvArgs& A = e;
    // the two older privileged frames (12 registers the policy tile could not spare): loaded here, stored behind the joints phase
    if (!PRE) hist_load<3, HGYM_PRIV_FRAME, RO_NIP>(A.st.priv_ring, block * RO_E, RO_E, (int)(ring_step % 3), t, RO_NT, hist_p);
    if (!PRE) {
        if (t < 128)          // the head wavefronts' share; the others stored theirs under the head (idle hook)
            hist_store<15, HGYM_OBS_FRAME, RO_NIO>(A.out.obs, block * RO_E, RO_E, (int)(ring_step % 15), t, RO_NT, nullptr, A.cfg.clip_obs, hist_o);
    }
    stamp(1);
    env_step_phase_j<RO_E, true>(A, block, t, RO_NT, esm);       // joints + per-joint reward products; synthetic-physics remainder on waves 6, 7
    if (!PRE)
        hist_store<3, HGYM_PRIV_FRAME, RO_NIP>(A.out.priv_obs, block * RO_E, RO_E, (int)(ring_step % 3), t, RO_NT, nullptr, A.cfg.clip_obs,
                                               hist_p);
    __syncthreads();
    stamp(2);
    if (t < RO_CHAIN) env_step_phase_a3<RO_E>(A, block, t, RO_NT, esm, csc0);      // the per-env chain, four wavefronts by role
    __syncthreads();
    env_step_phase_f<RO_E>(A, block, t, RO_NT, esm);       // per-joint reset / reference pose / frame entries / last_* copies
    env_step_reward_sum<RO_E>(A, block, t, RO_NT, esm);   // (the last wavefront: phase F has the first six)
    __syncthreads();
    stamp(3);
    env_stage_out<RO_E, true>(A, block, t, RO_NT, esm);
    stamp(4);
    env_step_phase_b<15, 3, RO_E>(A, block, t, RO_NT, esm, csc0, ring_step, false, true);
    if (PRE && pp.prev_reset) {
        // this launch's next-observation rows were pre-written by the previous launch from the history as IT found it: an env the previous
        // step reset has zero older frames (13 of 15, 1 of 3) -- every wavefront reads the tile's 32 flags, loops over the set ones
        const unsigned long long mask = __ballot(prev_rs);
        for (unsigned long long mm = mask; mm; mm &= mm - 1) {
            const int le = __builtin_ctzll(mm);
            float* dobs = A.out.obs + (int64_t)(block * RO_E + le) * 15 * HGYM_OBS_FRAME;
            float* dpriv = A.out.priv_obs + (int64_t)(block * RO_E + le) * 3 * HGYM_PRIV_FRAME;
            for (int i = t; i < 13 * HGYM_OBS_FRAME; i += RO_NT) dobs[i] = 0.0f;
            for (int i = t; i < HGYM_PRIV_FRAME; i += RO_NT) dpriv[i] = 0.0f;
        }
    }
    stamp(5);
    if (block == 0 && t == 0) {
        pp.out[0] = csc0 + 1;
        pp.out[1] = ring_step + 1;
        pp.out[2] = sstep + 1;
        if (A.out.t_step) A.out.t_step[0] = sstep + 1;     // the caller's sampling-step counter stays current
    }
}

__global__ void rollout_begin_kernel(const int64_t* __restrict__ counters, const int64_t* __restrict__ step, RolloutScratch* scr, int parity) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    scr->pp[parity][0] = counters[0];
    scr->pp[parity][1] = counters[2];
    scr->pp[parity][2] = step[0];
}

__global__ __launch_bounds__(1024) void rollout_fin_kernel(const FinArgs f) { fin_block(f, threadIdx.x, blockDim.x); }

static FinArgs parity_fin(const HgymEnvConfig& cfg, const HgymEnvState& st, const HgymEnvOut& out, RolloutScratch* scr, int parity) {
    FinArgs f = make_fin_args(cfg, st, out, FIN_MODE_STEP);
    f.reset_count = &scr->reset_cnt[parity];
    f.episode_acc = scr->acc[parity];
    return f;
}

}  // namespace hgym

using namespace hgym;

extern "C" {

int32_t hgym_rollout_begin(const HgymEnvState* st, const int64_t* step_counter, void* scratch, int32_t parity, void* stream) {
    HG_REQUIRE(st && st->counters && step_counter && scratch, HGYM_E_BADARG, "null state / step counter / scratch");
    HG_REQUIRE(parity == 0 || parity == 1, HGYM_E_BADARG, "parity=%d", parity);
    hipLaunchKernelGGL(rollout_begin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, st->counters, step_counter, (RolloutScratch*)scratch,
                       parity);
    HG_CHECK_LAUNCH("rollout_begin_kernel");
    return HGYM_OK;
}

int32_t hgym_rollout_step(const HgymNetConfig* cfg, const HgymNet* net, const HgymEnvConfig* env_cfg, const HgymSimTensors* sim,
                          const HgymEnvState* st, const HgymEnvOut* out, const HgymEnvOut* prev_out, const float* obs, const float* priv,
                          uint64_t seed, float* actions, float* mu, float* sigma, float* logp, float* values, void* scratch,
                          int32_t parity, const HgymObsShadow* shadow, void* stream) {
    HG_REQUIRE(cfg && net && env_cfg && sim && st && out && scratch, HGYM_E_BADARG, "null argument");
    const bool nocritic = values == nullptr;      // deferred values (header v7): hgym_critic_values runs after the rollout
    HG_REQUIRE(obs && actions && mu && sigma && logp && (nocritic || priv), HGYM_E_BADARG, "null policy buffer");
    HG_REQUIRE(parity == 0 || parity == 1, HGYM_E_BADARG, "parity=%d", parity);
    // what the kernel compiles in (the header's "Supported:" list): the actor epilogue writes 12 actions per row into the env image
    // and the env part produces 15 x 47 / 3 x 73 wide rows, which the policy tiles read with these leading dimensions
    HG_REQUIRE(cfg->num_actions == HGYM_NUM_ACTIONS && cfg->num_obs == 15 * HGYM_OBS_FRAME && cfg->num_priv == 3 * HGYM_PRIV_FRAME,
               HGYM_E_UNSUPPORTED, "fused rollout step: net shape %d / %d -> %d is not XBot-L's 705 / 219 -> 12", cfg->num_obs, cfg->num_priv,
               cfg->num_actions);
    RolloutScratch* scr = (RolloutScratch*)scratch;
    const int M = env_cfg->num_envs;
    FwdArgs f;
    EnvArgs e;
    FinArgs fin;
    RolloutPP pp;
    memset(&fin, 0, sizeof(fin));
    pp.prev_reset = nullptr;
    size_t lds_pol = 0;
    int32_t rc = rollout_fwd_args(cfg, net, M, obs, priv, seed, &scr->pp[parity][2], actions, mu, sigma, logp, values, &f, &lds_pol, shadow);
    if (rc) return rc;
    rc = rollout_env_args(env_cfg, sim, st, out, actions, &e);
    if (rc) return rc;
    HG_REQUIRE(out->t_values == values, HGYM_E_BADARG, "the transition sink must take this launch's values");
    if (nocritic) {
        HG_REQUIRE(out->t_rewards && out->t_time_outs && (!prev_out || (prev_out->t_time_outs && !prev_out->t_values)), HGYM_E_BADARG,
                   "values = NULL: the transition sinks must be of the deferred kind (t_values NULL, t_time_outs set)");
        HG_REQUIRE(!out->obs_ahead && !out->priv_ahead && !out->obs_older_ready && !out->l0_ahead && !out->l0_ready && !out->obs_bf16_ahead,
                   HGYM_E_BADARG, "values = NULL: rows ahead / the carried first layer are the critic workgroups' side jobs");
    }
    e.reset_count = &scr->reset_cnt[parity];
    e.st.episode_acc = scr->acc[parity];
    if (prev_out) {
        HG_REQUIRE(prev_out->rew != out->rew && prev_out->reset != out->reset && prev_out->time_out != out->time_out, HGYM_E_BADARG,
                   "this step's and the previous step's rew / reset / time_out must be distinct buffers (the finaliser runs concurrently)");
        HG_REQUIRE(prev_out->time_out && prev_out->extras_time_outs && prev_out->extras_episode && prev_out->rew && prev_out->reset,
                   HGYM_E_BADARG, "null finaliser buffer");
        fin = parity_fin(*env_cfg, *st, *prev_out, scr, parity ^ 1);
        pp.prev_reset = prev_out->reset;
    }
    pp.in = scr->pp[parity];
    pp.out = scr->pp[parity ^ 1];
    pp.env_lds_off = (int)round_up((int64_t)lds_pol, 16);
    {   // draw tables handed from launch to launch: [parity][tile][draws_len] floats behind the scratch header
        const LdsMap m = lds_map(RO_E);
        pp.draws_len = m.frame - m.u_delay;
        HG_REQUIRE((pp.draws_len & 3) == 0 && (m.u_delay & 3) == 0 && pp.draws_len <= 2 * 4 * RO_NT &&
                       (size_t)pp.draws_len * 4 * 2 <= (size_t)HGYM_ROLLOUT_DRAW_BYTES_PER_ENV * RO_E,
                   HGYM_E_UNSUPPORTED, "draw tables of %d floats per tile do not fit the scratch layout", pp.draws_len);
        float* tables = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + HGYM_ROLLOUT_SCRATCH_HEADER_BYTES);
        const int64_t per_parity = (int64_t)(M / RO_E) * pp.draws_len;
        pp.draws_in = (prev_out && !nocritic) ? tables + parity * per_parity : nullptr;       // the first step of a rollout draws its own
        pp.draws_out = nocritic ? nullptr : tables + (parity ^ 1) * per_parity;
    }
    memset(&pp.l0, 0, sizeof(pp.l0));
    memset(&pp.ah, 0, sizeof(pp.ah));
    const bool part = out->l0_ready != nullptr;
    {   // first layer of the actor carried across launches (HgymEnvOut.l0_ahead / l0_ready)
        // k-steps formed ahead: whole 128-column chunks, at most 20 (640 of the 658 columns of the 14 older frames).  12 by default:
        // the critic workgroup pays for every k-step it takes over at the same L2 -> CU fill rate, and with all 20 it becomes the
        // launch's longest workgroup (profiles/r04_l0_ahead_ab.txt: collection 2.33 -> 2.27 ms with 12, 2.30 with 20, 2.34 with 8).
        // HGYM_L0_KB0 tunes the split (A/B runs).
        const char* kb0_env = getenv("HGYM_L0_KB0");          // (read per call: the tests run several splits in one process)
        const int kb0_v = kb0_env ? atoi(kb0_env) : 12;
        const int KB0_AHEAD = (kb0_v >= 4 && kb0_v <= 20 && kb0_v % 4 == 0) ? kb0_v : 12;
        HG_REQUIRE(!part || prev_out, HGYM_E_BADARG, "l0_ready on the first step of a rollout: no launch has left partial sums");
        HG_REQUIRE(!(part || out->l0_ahead) || (f.net[0].layer[0].KB == 24 && f.net[0].layer[0].N == 512 && HGYM_OBS_FRAME * 14 >= 32 * KB0_AHEAD),
                   HGYM_E_UNSUPPORTED, "the carried first layer is built for XBot-L's 15 x 47 -> 512 actor input");
        HG_REQUIRE(!out->l0_ahead || (((uintptr_t)out->l0_ahead & 15) == 0 && out->l0_ahead != out->l0_ready), HGYM_E_BADARG,
                   "l0_ahead must be 16-byte aligned and distinct from l0_ready");
        HG_REQUIRE(!out->obs_bf16_ahead || (out->l0_ahead && out->ld_obs_bf16_ahead >= 768 && out->ld_obs_bf16_ahead % 8 == 0 &&
                                           ((uintptr_t)out->obs_bf16_ahead & 15) == 0), HGYM_E_BADARG, "obs_bf16_ahead: needs l0_ahead, ld >= 768 (multiple of 8), 16-byte aligned");
        if (part) {
            pp.l0.acc = out->l0_ready;
            pp.l0.reset = prev_out->reset;
            pp.l0.kb0 = KB0_AHEAD;
        }
        if (out->l0_ahead) {
            pp.ah.acc_out = out->l0_ahead;
            pp.ah.xs_next = (__bf16*)out->obs_bf16_ahead;
            pp.ah.ldxs = out->ld_obs_bf16_ahead;
            pp.ah.shift = HGYM_OBS_FRAME;
            pp.ah.kb0 = KB0_AHEAD;
        }
    }
    f.dbg = phase_buffer((int64_t)(M / RO_E) * 3);
    const size_t lds = (size_t)pp.env_lds_off + step_smem_bytes(RO_E);
    const bool pre = out->obs_older_ready != 0;
    HG_REQUIRE(!pre || prev_out, HGYM_E_BADARG, "obs_older_ready on the first step of a rollout: no launch has written those frames");
    HG_REQUIRE(!part || pre, HGYM_E_UNSUPPORTED, "l0_ready is built together with obs_older_ready (the steady-state launch)");
    HG_REQUIRE((out->obs_ahead != nullptr) == (out->priv_ahead != nullptr), HGYM_E_BADARG, "obs_ahead and priv_ahead: both or neither");
    HG_REQUIRE(!out->obs_ahead || (out->obs_ahead != out->obs && out->priv_ahead != out->priv_obs), HGYM_E_BADARG,
               "obs_ahead / priv_ahead must be the rows AFTER obs / priv_obs");
    {
        const void* fn = nocritic ? (prev_out ? reinterpret_cast<const void*>(&rollout_step_kernel<true, false, false, true>)
                                              : reinterpret_cast<const void*>(&rollout_step_kernel<false, false, false, true>))
                       : part ? reinterpret_cast<const void*>(&rollout_step_kernel<true, true, true>)
                       : pre ? reinterpret_cast<const void*>(&rollout_step_kernel<true, true>)
                             : (prev_out ? reinterpret_cast<const void*>(&rollout_step_kernel<true, false>)
                                         : reinterpret_cast<const void*>(&rollout_step_kernel<false, false>));
        rc = ensure_dynamic_lds(fn, lds, "rollout_step_kernel");
        if (rc) return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    prof_begin(HGYM_PROF_ROLLOUT, s);
    if (nocritic && prev_out) hipLaunchKernelGGL((rollout_step_kernel<true, false, false, true>), dim3(M / RO_E, 2), dim3(RO_NT), lds, s, f, e, fin, pp);
    else if (nocritic) hipLaunchKernelGGL((rollout_step_kernel<false, false, false, true>), dim3(M / RO_E, 1), dim3(RO_NT), lds, s, f, e, fin, pp);
    else if (part) hipLaunchKernelGGL((rollout_step_kernel<true, true, true>), dim3(M / RO_E, 3), dim3(RO_NT), lds, s, f, e, fin, pp);
    else if (pre) hipLaunchKernelGGL((rollout_step_kernel<true, true>), dim3(M / RO_E, 3), dim3(RO_NT), lds, s, f, e, fin, pp);
    else if (prev_out) hipLaunchKernelGGL((rollout_step_kernel<true, false>), dim3(M / RO_E, 3), dim3(RO_NT), lds, s, f, e, fin, pp);
    else hipLaunchKernelGGL((rollout_step_kernel<false, false>), dim3(M / RO_E, 2), dim3(RO_NT), lds, s, f, e, fin, pp);
    {   // algorithmic HBM bytes of the fused step: the env step's + the policy's input rows and outputs
        // + the policy's weights, read once per launch (3 704 420 B / N per env-step in fp32 terms; here the bf16 forward
        // fragments the tiles actually stream, padded layout) -- until round 4 this term was left out (0.149 -> 0.155 at N = 4096)
        const double env_b = 4.0 * (245 + 14 * 47 + 2 * 73 + 15 * 47 + 3 * 73) + 6;
        const double pol_b = 4.0 * (cfg->num_obs + cfg->num_priv + 3 * cfg->num_actions + 2);
        double w_b = 0.0;
        for (int n = 0; n < (nocritic ? 1 : 2); ++n)
            for (int l = 0; l < 4; ++l) w_b += (double)f.net[n].layer[l].NB * f.net[n].layer[l].KB * 1024.0;
        // (deferred values: the critic's input rows, its value and its weights are hgym_critic_values' bytes, not this launch's)
        prof_end(HGYM_PROF_ROLLOUT, s, (double)M * (env_b + pol_b - (nocritic ? 4.0 * (cfg->num_priv + 1) : 0.0)) + w_b);
    }
    HG_CHECK_LAUNCH("rollout_step_kernel");
    return HGYM_OK;
}

int32_t hgym_rollout_end(const HgymEnvConfig* env_cfg, const HgymEnvState* st, const HgymEnvOut* last_out, void* scratch, int32_t parity,
                         void* stream) {
    HG_REQUIRE(env_cfg && st && last_out && scratch, HGYM_E_BADARG, "null argument");
    HG_REQUIRE(parity == 0 || parity == 1, HGYM_E_BADARG, "parity=%d", parity);
    HG_REQUIRE(st->counters && last_out->time_out && last_out->extras_time_outs && last_out->extras_episode && last_out->rew && last_out->reset,
               HGYM_E_BADARG, "null finaliser buffer");
    const FinArgs f = parity_fin(*env_cfg, *st, *last_out, (RolloutScratch*)scratch, parity);
    hipLaunchKernelGGL(rollout_fin_kernel, dim3(1), dim3(env_cfg->num_envs > 256 ? 1024 : 256), 0, (hipStream_t)stream, f);
    HG_CHECK_LAUNCH("rollout_fin_kernel");
    return HGYM_OK;
}

}  // extern "C"
