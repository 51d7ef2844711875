// hgym_fb3.hpp -- mlp_fb3_kernel: the update's forward + PPO loss + dZ chain of a 64-row tile on ROLE-SPECIALISED wavefronts (round 6).
//
// mlp_fb_kernel (hgym_fused.hpp) runs a tile on sixteen wavefronts that all do everything: gather input rows, stream weights, MFMA, ELU,
// store H / dZ.  Two things follow from that shape: every phase opens with `s_waitcnt vmcnt(0)` behind the previous
// phase's STORES (gfx950 has one counter for loads and stores and returns them out of order with respect to each other, so a wave that has a
// store in flight cannot wait for ONE load), and sixteen one-block-wide strips re-read the tile's activations from LDS sixteen times -- the
// second layer runs at the LDS port's rate, not the matrix pipe's.  Here the twelve wavefronts of a workgroup have roles:
//
//   * 8 COMPUTE wavefronts (2 per SIMD, 168 registers): strips twice as wide (first layer 4 x 4 accumulator blocks for the 512-wide actor,
//     two passes of 4 x 3 for the 768-wide critic; hidden layers 4 x 2), i.e. half the LDS fragment reads per MFMA.  Their only global
//     memory instructions are WEIGHT LOADS: activations and gradients are written to LDS and nowhere else, so the compiler's waitcnt pass
//     counts the weight ring exactly in every phase and nothing ever waits for a store.
//   * 4 SERVICE wavefronts (1 per SIMD): gather the tile's input rows from the bf16 observation shadow (every chunk of the tile requested at
//     tile start: 24 x 16 bytes in flight per lane), the loss inputs, the bias vectors; after each phase they copy the activation / gradient
//     block the compute wavefronts have just left in LDS to HBM -- a linear 16-bytes-per-lane copy, since a tile's blocks are contiguous in
//     the block layout -- while the next phase computes.
//   * Barriers wait for LDS traffic only (`s_waitcnt lgkmcnt(0); s_barrier`): weight loads primed a phase ahead and the service wavefronts'
//     stores stay in flight across them (`__syncthreads()` would drain both).
//
// Same arithmetic as mlp_fb_kernel element for element (same fragments in the same k order on one accumulator chain per output, the same
// epilogues, the same loss code): H / dZ and the loss partials are bit-identical (tests/test_fused_gpu.py).
// Instantiated for XBot-L's two shapes -- (first hidden width 512, six input chunks) and (768, two chunks) -- from the bf16 shadow only; every
// other case takes mlp_fb_kernel (hgym_net.hip: fused_grad).
#pragma once
#include "hgym_fused.hpp"
#pragma clang fp contract(fast)

namespace hgym {

constexpr int FB3_NC = 8;                                  // compute wavefronts
constexpr int FB3_NS = 4;                                  // service wavefronts
constexpr int FB3_THREADS = (FB3_NC + FB3_NS) * 64;
constexpr int FB3_SL = FB3_NS * 64;                        // service lanes
#ifndef FB3_D0
#define FB3_D0 2
#endif
#ifndef FB3_D1
#define FB3_D1 4
#endif
#ifndef FB3_D2
#define FB3_D2 8
#endif
#ifndef FB3_DB2
#define FB3_DB2 2
#endif
#ifndef FB3_DB1
#define FB3_DB1 2
#endif

// FB3_WAVE_CLOCK (instrumented variant, tools/probe_fb3_waves.py): lane 0 of EVERY wavefront stamps the wall clock into
// dbg[(workgroup * 16 + wave) * 8 + slot] -- the caller's phase buffer must hold 128 slots per workgroup
#ifdef FB3_WAVE_CLOCK
#ifndef FB3_WSLOTS
#define FB3_WSLOTS 8
#endif
__device__ __forceinline__ void fb3_wstamp(long long* dbg, int wave, int lane, int slot) {
    if (dbg && lane == 0) dbg[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + wave) * FB3_WSLOTS + slot] = (long long)__builtin_amdgcn_s_memrealtime();
}
#define FB3_WSTAMP(wave, lane, slot) fb3_wstamp(a.dbg, wave, lane, slot)
#define FB3_PSTAMP(slot)
#else
#define FB3_WSTAMP(wave, lane, slot)
#define FB3_PSTAMP(slot) phase_stamp(a.dbg, slot)
#endif

// workgroup barrier that orders LDS traffic only: global loads (weight rings) and stores (the service wavefronts' copies) stay in flight
__device__ __forceinline__ void fb3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS block -> HBM, `bytes` (a multiple of 4 KiB) by the 256 service lanes, 16 bytes per lane per step; all of a round's LDS reads are
// issued before its first store
template <bool NT>
__device__ __forceinline__ void fb3_copy_out(const char* __restrict__ lds, char* __restrict__ g, int bytes, int sl) {
    constexpr int STEP = FB3_SL * 16;
    int o = sl * 16;
#ifdef FB3_ABLATE_STORES      // timing ablation (results wrong by design): 1 = no copy at all, 4 = a quarter of the bytes
    if (FB3_ABLATE_STORES == 1) return;
    bytes /= FB3_ABLATE_STORES;
#endif
    for (; o + 3 * STEP < bytes; o += 4 * STEP) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u32x4*>(lds + o + u * STEP);
#pragma unroll
        for (int u = 0; u < 4; ++u) st_stream_u4<NT>(g + o + u * STEP, v[u]);
    }
    for (; o < bytes; o += STEP) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(lds + o);
        st_stream_u4<NT>(g + o, v);
    }
}

// dZ_out = (dZ_in * W) .* elu'(H) on the resident tile, IN PLACE over H (every lane reads a block entry and later writes that very entry);
// LDS only.  bwd_step (hgym_fused.hpp) is the form that also stores to HBM.
template <int G, int MB, int NW, int D, int MODE = 0, int GR, class Next>      // MODE 1 / 2: NBBc % D == 0 / != 0 known at compile time
__device__ __forceinline__ void fb3_bwd_step(WRing<GR, D>& R, const u32x4* __restrict__ WTf, int NBo, int NBBc, const char* in_lds, int CBin,
                                             char* h_lds, int wave, int lane, Next prime_next) {
    const int r = lane & 15, q = lane >> 4;
    const int loff = r * 32 + q * 8;
    bool primed = false;
    for (int nb0 = wave * G; nb0 < NBo; nb0 += NW * G) {
        f32x4 acc[MB][G];
        zero_acc<G, MB>(acc);
        if (MODE == 1 || (MODE == 0 && NBBc % D == 0)) mma_stream<G, MB, D, 1>(R, WTf + HG_WOFF((int64_t)nb0 * NBBc * 64) + lane, HG_WSTR(NBBc * 64), NBBc, in_lds, CBin, lane, acc);
        else mma_ring<G, MB, D, 1>(R, WTf + HG_WOFF((int64_t)nb0 * NBBc * 64) + lane, HG_WSTR(NBBc * 64), 0, NBBc, NBBc, in_lds, CBin, lane, acc);
        const int nxt = nb0 + NW * G;
        if (nxt < NBo) wring_prime<G, D>(R, WTf + HG_WOFF((int64_t)nxt * NBBc * 64) + lane, HG_WSTR(NBBc * 64), NBBc);
        else { prime_next(); primed = true; }
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                char* p = h_lds + (i * NBo + nb0 + g) * 512 + loff;
                const u32x2 y2 = *reinterpret_cast<const u32x2*>(p);
                const unsigned int w0 = y2[0], w1 = y2[1];
                const float y[4] = {bf16_bits_to_f32(w0 & 0xffffu), bf16_bits_to_f32(w0 >> 16), bf16_bits_to_f32(w1 & 0xffffu), bf16_bits_to_f32(w1 >> 16)};
                float d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = acc[i][g][e] * ((y[e] > 0.0f) ? 1.0f : (y[e] + 1.0f));   // elu'(z) from y = elu(z)
                *reinterpret_cast<u32x2*>(p) = pack_bf16x4(d[0], d[1], d[2], d[3]);
            }
    }
    if (!primed) prime_next();
}

// LDS map: mlp_fb_kernel's (fused_lds_p / _q / _bias, fb_lds_extra): P = H0 -> dZ0, Q = the two input chunk buffers -> H1 -> dZ1, the four
// bias vectors, R0 = the dZ3 tile, the head waves' partial sums, H2 -> dZ2, the tile rows' storage indices, the gathered loss inputs.
struct Fb3Lds {
    char *P, *Q, *R0, *H2;
    float *bl, *red, *lin;
    int* rowidx;
};
__device__ __forceinline__ Fb3Lds fb3_lds(const FusedNet& n, char* smem) {
    constexpr int BM = 64;
    Fb3Lds m;
    m.P = smem;
    m.Q = smem + fused_lds_p(n, BM);
    m.bl = reinterpret_cast<float*>(m.Q + fused_lds_q(n, BM));
    m.R0 = m.Q + fused_lds_q(n, BM) + fused_lds_bias(n);
    m.red = reinterpret_cast<float*>(m.R0 + BM * 64 * n.layer[3].NBB);
    m.H2 = m.R0 + BM * 64 * n.layer[3].NBB + 4 * 32 * 4;
    m.rowidx = reinterpret_cast<int*>(m.H2 + BM * n.layer[2].N * 2);
    m.lin = reinterpret_cast<float*>(m.rowidx + BM);
    return m;
}

// ---------------------------------------------------------------------------------------------------------------- service wavefronts
template <int NCT>          // input chunks of 128 columns (compile-time: every load of the tile is issued up front, exact vmcnt per chunk)
__device__ __forceinline__ void fb3_service(const FwdArgs& a, const FbLoss& L, const FusedNet& n, bool is_actor, char* smem, int sw, int lane) {
    constexpr int BM = 64, CH = BM * FUSED_CHUNK * 2;
    const Fb3Lds S = fb3_lds(n, smem);
    const int sl = sw * 64 + lane;
    const int m0 = blockIdx.x * BM;
    const FusedLayer &L0 = n.layer[0], &L1 = n.layer[1], &L2 = n.layer[2], &L3 = n.layer[3];
    // ---- the tile's input rows, gathered from the bf16 shadow: item j of this lane is what lane `lane` of wavefront sw * 4 + j stages in
    //      mlp_fb_kernel<XB16> (16 consecutive lanes = 8 rows x the two halves of one block row, lane groups = 4 consecutive column blocks)
    const char* srow[4];
    int loff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = ((sw & 1) * 4 + j) * 8 + ((lane >> 1) & 7), cb = (sw >> 1) * 4 + (lane >> 4), hf = lane & 1;
        int m = m0 + row;
        m = m < a.M ? m : a.M - 1;
        const int64_t src = a.idx ? a.idx[m] : (int64_t)m;
        srow[j] = reinterpret_cast<const char*>(n.xb + src * n.ldxb + cb * 16 + hf * 8);
        loff[j] = ((row >> 4) * 8 + cb) * 512 + (row & 15) * 32 + hf * 16;
        if (cb == 0 && hf == 0) S.rowidx[row] = (int)src;
    }
    u32x4 stg[NCT][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) stg[0][j] = ld_stream_u4<(kFusedNT & 1) != 0>(srow[j]);
    // the four bias vectors (<= 768 + 256 + 128 + 16 floats: five per lane), loaded behind chunk 0 and in front of the other chunks: loads
    // return in order, so B0 waits for chunk 0 and these only
    constexpr int BIT = (768 + 256 + 128 + 16 + FB3_SL - 1) / FB3_SL;
    float bv[BIT];
    const int bn0 = L0.N, bn1 = bn0 + L1.N, bn2 = bn1 + L2.N, bn3 = bn2 + 16;
#pragma unroll
    for (int u = 0; u < BIT; ++u) {
        int i = sl + u * FB3_SL;
        i = i < bn2 + L3.N ? i : bn2 + L3.N - 1;
        const float* src = i < bn0 ? L0.bias + i : (i < bn1 ? L1.bias + (i - bn0) : (i < bn2 ? L2.bias + (i - bn1) : L3.bias + (i - bn2)));
        bv[u] = *src;
    }
#pragma unroll
    for (int c = 1; c < NCT; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) stg[c][j] = ld_stream_u4<(kFusedNT & 1) != 0>(srow[j] + c * (FUSED_CHUNK * 2));
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(S.Q + loff[j]) = stg[0][j];
#pragma unroll
    for (int u = 0; u < BIT; ++u) {
        const int i = sl + u * FB3_SL;
        if (i < bn3) S.bl[i] = bv[u];
    }
    fb3_barrier();                                                       // B0: chunk 0, biases, row indices
    // ---- loss inputs of the tile's rows (scattered 48-byte rows and scalars of the storage) -> LDS; issued now, behind the input chunks,
    //      written once the last chunk is staged.  mlp_fb_kernel's l2idle on 512 lanes; here two rounds of 256.
    F4 v1[2], v2;
    float s2 = 0.0f;
    if (is_actor) {
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int j = sl + rd * FB3_SL, rw = j >> 3, k = j & 7;
            const int64_t ri = S.rowidx[rw];
            const float* base = k < 3 ? L.actions : (k < 6 ? L.old_mu : L.old_sigma);
            v1[rd] = *reinterpret_cast<const F4*>(base + ri * 12 + 4 * (k < 3 ? k : (k < 6 ? k - 3 : k - 6)));
        }
        const int64_t ri2 = S.rowidx[sl & 63];
        v2 = *reinterpret_cast<const F4*>(L.old_sigma + ri2 * 12 + 8);
        s2 = (sl < 128 ? L.advantages : L.logp)[ri2];
    } else {
        const int64_t ri = S.rowidx[sl & 63];
        s2 = (sl < 64 ? L.returns : L.values)[ri];
        v1[0] = v1[1] = v2 = F4{{0.f, 0.f, 0.f, 0.f}};
    }
#pragma unroll
    for (int c = 1; c < NCT; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(S.Q + (c & 1) * CH + loff[j]) = stg[c][j];
        fb3_barrier();                                                   // B(c): chunk c staged; the compute waves are done with chunk c - 1
    }
    if (is_actor) {
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int j = sl + rd * FB3_SL;
            *reinterpret_cast<F4*>(S.lin + (j >> 3) * FB_LIN_ACTOR + 4 * (j & 7)) = v1[rd];
        }
        if (sl < 64) *reinterpret_cast<F4*>(S.lin + (sl & 63) * FB_LIN_ACTOR + 32) = v2;
        else if (sl < 192) S.lin[(sl & 63) * FB_LIN_ACTOR + (sl < 128 ? 36 : 37)] = s2;
    } else if (sl < 128) {
        S.lin[(sl & 63) * 2 + (sl >> 6)] = s2;
    }
    constexpr bool NTH = (kFusedNT & 2) != 0, NTZ = (kFusedNT & 4) != 0;
    const int64_t row0 = (int64_t)m0;
    const int N0 = L0.N, N1 = L1.N, N2 = L2.N, NBB3 = L3.NBB;
    fb3_barrier();                                                       // B_L0: H0 in P
    FB3_WSTAMP(FB3_NC + sw, lane, 0);
    fb3_copy_out<NTH>(S.P, reinterpret_cast<char*>(n.H[0]) + row0 * N0 * 2, BM * N0 * 2, sl);
    FB3_WSTAMP(FB3_NC + sw, lane, 2);
    fb3_barrier();                                                       // B_L1: H1 in Q
    FB3_WSTAMP(FB3_NC + sw, lane, 3);
    fb3_copy_out<NTH>(S.Q, reinterpret_cast<char*>(n.H[1]) + row0 * N1 * 2, BM * N1 * 2, sl);
    fb3_barrier();                                                       // B_L2: H2
    fb3_copy_out<NTH>(S.H2, reinterpret_cast<char*>(n.H[2]) + row0 * N2 * 2, BM * N2 * 2, sl);
    fb3_barrier();                                                       // B_hd: dZ3 tile + the head waves' partial sums
    fb3_copy_out<false>(S.R0, reinterpret_cast<char*>(n.dZ[3]) + row0 * 64 * NBB3, BM * 64 * NBB3, sl);
    if (sl < 32) {
        const bool mine = is_actor ? (sl != 1 && sl < 28) : (sl == 1 || sl == 28);
        if (mine) L.partials[(int64_t)blockIdx.x * 32 + sl] = S.red[sl] + S.red[32 + sl] + S.red[64 + sl] + S.red[96 + sl];
    }
    fb3_barrier();                                                       // B_b3: dZ2 over H2
    fb3_copy_out<NTZ>(S.H2, reinterpret_cast<char*>(n.dZ[2]) + row0 * N2 * 2, BM * N2 * 2, sl);
    fb3_barrier();                                                       // B_b2: dZ1 over H1
    fb3_copy_out<NTZ>(S.Q, reinterpret_cast<char*>(n.dZ[1]) + row0 * N1 * 2, BM * N1 * 2, sl);
    fb3_barrier();                                                       // B_b1: dZ0 over H0
    fb3_copy_out<NTZ>(S.P, reinterpret_cast<char*>(n.dZ[0]) + row0 * N0 * 2, BM * N0 * 2, sl);
}

// ---------------------------------------------------------------------------------------------------------------- compute wavefronts
// PPO loss of one head wavefront (lane (r, q): row hw * 16 + r of the tile, head outputs 4q .. 4q + 3): mlp_fb_kernel's `head`, loss inputs
// from LDS (`lin`), the dZ3 block and the per-wave sums to LDS only.  ppo.py:128-168 forward scalars + the hand-written backward of the loss
// w.r.t. mu, std and V (oracle/ppo_oracle.py: ppo_loss_and_grads).
__device__ __forceinline__ void fb3_head(const FwdArgs& a, const FbLoss& L, const Fb3Lds& S, bool is_actor, int hw, int m, int lane, const float (&out)[4]) {
    const int r = lane & 15, q = lane >> 4;
    const int A = a.A;
    const float invB = 1.0f / (float)a.M;
    const bool valid = m < a.M;
    const int rl = hw * 16 + r;
    const float* lin = S.lin;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    float part[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) part[k] = 0.0f;
    if (is_actor) {
        float act[4] = {0.f, 0.f, 0.f, 0.f}, mo[4] = {0.f, 0.f, 0.f, 0.f}, so[4] = {1.f, 1.f, 1.f, 1.f}, sg[4] = {1.f, 1.f, 1.f, 1.f};
        if (q < 3) {
            const F4 qa = *reinterpret_cast<const F4*>(lin + rl * FB_LIN_ACTOR + 4 * q);
            const F4 qo = *reinterpret_cast<const F4*>(lin + rl * FB_LIN_ACTOR + 12 + 4 * q);
            const F4 qs = *reinterpret_cast<const F4*>(lin + rl * FB_LIN_ACTOR + 24 + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) { act[e] = qa.v[e]; mo[e] = qo.v[e]; so[e] = qs.v[e]; }
        }
        const float adv = lin[rl * FB_LIN_ACTOR + 36], lpold = lin[rl * FB_LIN_ACTOR + 37];
        float lp = 0.0f, ent = 0.0f, kl = 0.0f, diff[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            diff[e] = 0.0f;
            if (4 * q + e < A) {
                const float mm = out[e];
                const float s = mm * 0.0f + a.std_[4 * q + e];
                const float d = act[e] - mm;
                diff[e] = d;
                sg[e] = s;
                lp += -(d * d) / (2.0f * s * s) - logf(s) - 0.9189385332046727f;
                ent += 0.5f + 0.9189385332046727f + logf(s);
                kl += logf(s / so[e] + 1.e-5f) + (so[e] * so[e] + (mo[e] - mm) * (mo[e] - mm)) / (2.0f * (s * s)) - 0.5f;
            }
        }
        lp += __shfl_xor(lp, 16, 64);  lp += __shfl_xor(lp, 32, 64);
        ent += __shfl_xor(ent, 16, 64); ent += __shfl_xor(ent, 32, 64);
        kl += __shfl_xor(kl, 16, 64);  kl += __shfl_xor(kl, 32, 64);
        const float ratio = expf(lp - lpold);
        const float s1 = -adv * ratio;
        const float s2 = -adv * clampf(ratio, 1.0f - L.clip, 1.0f + L.clip);
        const float in_range = (ratio >= 1.0f - L.clip && ratio <= 1.0f + L.clip) ? 1.0f : 0.0f;
        const float w1 = s1 > s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f);
        const float d_lp = (-adv) * (w1 + (1.0f - w1) * in_range) * invB * ratio;
        if (valid) {
            if (q == 0) { part[0] = fmaxf(s1, s2); part[1] = ent; part[2] = kl; }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < A) {
                    const float s = sg[e], d = diff[e];
                    g[e] = d_lp * d / (s * s);
                    part[3 + e] = d_lp * (d * d / (s * s * s) - 1.0f / s) - (L.entropy_coef * invB) / s;
                    part[7 + e] = g[e];
                }
        }
    } else {
        const float ret = lin[rl * 2], vold = lin[rl * 2 + 1];
        const float v = out[0];
        const float vc = vold + clampf(v - vold, -L.clip, L.clip);
        const float l1 = (v - ret) * (v - ret), l2 = (vc - ret) * (vc - ret);
        const float v_in = ((v - vold) >= -L.clip && (v - vold) <= L.clip) ? 1.0f : 0.0f;
        const float u1 = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.0f);
        if (valid && q == 0) {
            g[0] = L.value_coef * invB * (u1 * 2.0f * (v - ret) + (1.0f - u1) * 2.0f * (vc - ret) * v_in);
            part[0] = fmaxf(l1, l2);
            part[1] = g[0];
        }
    }
    // dZ3 tile, block layout: row block hw, column block 0 holds this lane's 4 columns (block 1 is zero padding)
    const u32x2 pk = pack_bf16x4(g[0], g[1], g[2], g[3]);
    const u32x2 zero = {0u, 0u};
    *reinterpret_cast<u32x2*>(S.R0 + (hw * 2 + 0) * 512 + r * 32 + q * 8) = pk;
    *reinterpret_cast<u32x2*>(S.R0 + (hw * 2 + 1) * 512 + r * 32 + q * 8) = zero;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        float v = part[k];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        part[k] = v;
    }
    if (r == 0) {
        float* w = S.red + hw * 32;
        if (is_actor) {
            if (q == 0) { w[0] = part[0]; w[2] = part[1]; w[3] = part[2]; }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < 12) { w[4 + 4 * q + e] = part[3 + e]; w[16 + 4 * q + e] = part[7 + e]; }
        } else if (q == 0) {
            w[1] = part[0];
            w[28] = part[1];
        }
    }
}

template <int G1P, int NP, int NCT>      // first-layer strip width per pass, passes (NP * 8 * G1P = the first hidden width / 16), input chunks
__device__ __forceinline__ void fb3_compute(const FwdArgs& a, const FbLoss& L, const FusedNet& n, bool is_actor, char* smem, int wave, int lane) {
    static_assert(NP == 1 || NCT <= 2, "a second pass re-reads the input from the two chunk buffers: they must hold all of it");
    // ring depths (k-steps of weight fragments in flight per wave): what 168 registers allow per phase.  A hidden layer's k-step is 8 MFMAs
    // (~140 cycles) against an L2 round trip of several hundred: the two-deep rings of the 128-register kernel run them at the load latency.
    constexpr int BM = 64, NW = FB3_NC, D = FB3_D0, D1 = FB3_D1, D2 = FB3_D2, DB2 = FB3_DB2, DB1 = FB3_DB1, MB = BM / 16, CH = BM * FUSED_CHUNK * 2;
    const Fb3Lds S = fb3_lds(n, smem);
    const int r = lane & 15;
    const int m0 = blockIdx.x * BM;
    const FusedLayer &L0 = n.layer[0], &L1 = n.layer[1], &L2 = n.layer[2], &L3 = n.layer[3];
    const float* bl = S.bl;
    WRing<G1P, D> r0;
    WRing<2, D1> r1;
    WRing<1, D2> r2;
    WRing<1, 4> r3;
    WRing<1, 2> ra;
    WRing<2, DB2> rb;
    WRing<G1P, DB1> rc;
    FB3_PSTAMP(0);
    // ---------------------------------------------------------------- layer 0: input in 128-column chunks, staged by the service waves
    wring_prime<G1P, D>(r0, L0.Wf + HG_WOFF((int64_t)(wave * G1P) * L0.KB * 64) + lane, HG_WSTR(L0.KB * 64), L0.KB);
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        const int nb0 = (pass * NW + wave) * G1P;
        const u32x4* wl0 = L0.Wf + HG_WOFF((int64_t)nb0 * L0.KB * 64) + lane;
        f32x4 acc[MB][G1P];
        zero_acc<G1P, MB>(acc);
        if (pass == 0) {
            fb3_barrier();                                               // B0
            FB3_PSTAMP(1);
        }
        for (int c = 0; c + 1 < NCT; ++c) {
            mma_chunk<G1P, MB, D, false, 1>(r0, wl0, HG_WSTR(L0.KB * 64), c * 4, S.Q + (c & 1) * CH, 8, lane, acc);
            if (pass == 0) fb3_barrier();                                // B(c + 1)
        }
        mma_chunk<G1P, MB, D, true, 1>(r0, wl0, HG_WSTR(L0.KB * 64), (NCT - 1) * 4, S.Q + ((NCT - 1) & 1) * CH, 8, lane, acc);
        if (pass + 1 < NP) wring_prime<G1P, D>(r0, L0.Wf + HG_WOFF((int64_t)(((pass + 1) * NW + wave) * G1P) * L0.KB * 64) + lane, HG_WSTR(L0.KB * 64), L0.KB);
        else {
            FB3_PSTAMP(2);
            hidden_prime<2, D1>(r1, L1, wave, lane);
        }
        epilogue_elu_t<G1P, MB, false>(acc, bl, nb0, S.P, L0.NB, nullptr, 0, lane);
    }
    fb3_barrier();                                                       // B_L0
    FB3_PSTAMP(3);
    // ---------------------------------------------------------------- layers 1, 2: input resident in LDS
    auto prime2 = [&]() { hidden_prime<1, D2>(r2, L2, wave, lane); };
#ifdef FB3_WAVE_CLOCK
    {   // hidden_layer<2, ..> spelled out, with stamps
        FB3_WSTAMP(wave, lane, 0);
        const int nb0 = wave * 2;
        f32x4 acc[MB][2];
        zero_acc<2, MB>(acc);
        mma_stream<2, MB, D1, 1>(r1, L1.Wf + (int64_t)nb0 * L1.KB * 64 + lane, L1.KB * 64, L1.KB, S.P, L0.NB, lane, acc);
        FB3_WSTAMP(wave, lane, 1);
        prime2();
        epilogue_elu_t<2, MB, false>(acc, bl + L0.N, nb0, S.Q, L1.NB, nullptr, 0, lane);
        FB3_WSTAMP(wave, lane, 2);
    }
#else
    hidden_layer<2, MB, NW, D1, true>(r1, L1, bl + L0.N, S.P, L0.NB, S.Q, nullptr, 0, wave, lane, prime2);
#endif
    fb3_barrier();                                                       // B_L1
    FB3_WSTAMP(wave, lane, 3);
    FB3_PSTAMP(4);
    const int NBB3 = L3.NBB, N0 = L0.N, N1 = L1.N, N2 = L2.N;
    auto prime3 = [&]() {
        if (wave < MB) wring_prime<1, 4>(r3, L3.Wf + lane, 0, L3.KB);
        bwd_prime<1, 2>(ra, L3.WTf, N2 / 16, NBB3, wave, lane);
    };
    hidden_layer<1, MB, NW, D2, true>(r2, L2, bl + L0.N + L1.N, S.Q, L1.NB, S.H2, nullptr, 0, wave, lane, prime3);
    fb3_barrier();                                                       // B_L2
    FB3_PSTAMP(5);
    // ---------------------------------------------------------------- head + PPO loss: one wave per 16-row block
    if (wave < MB) {
        f32x4 hacc[1][1];
        hacc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int CB3 = L2.NB;
        if (L3.KB % 4 == 0) mma_stream<1, 1, 4>(r3, L3.Wf + lane, 0, L3.KB, S.H2 + wave * CB3 * 512, CB3, lane, hacc);
        else mma_ring<1, 1, 4>(r3, L3.Wf + lane, 0, 0, L3.KB, L3.KB, S.H2 + wave * CB3 * 512, CB3, lane, hacc);
        const int q = lane >> 4;
        float mu[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) mu[e] = hacc[0][0][e] + ((4 * q + e < L3.N) ? bl[L0.N + L1.N + L2.N + 4 * q + e] : 0.0f);
        fb3_head(a, L, S, is_actor, wave, m0 + wave * 16 + r, lane, mu);
    }
    FB3_PSTAMP(6);
    fb3_barrier();                                                       // B_hd
    // ---------------------------------------------------------------- dZ chain on the resident tile, in place
    auto primeb = [&]() { bwd_prime<2, DB2>(rb, L2.WTf, N1 / 16, L2.NBB, wave, lane); };
    fb3_bwd_step<1, MB, NW, 2>(ra, L3.WTf, N2 / 16, NBB3, S.R0, 2 * NBB3, S.H2, wave, lane, primeb);
    fb3_barrier();                                                       // B_b3
    auto primec = [&]() { bwd_prime<G1P, DB1>(rc, L1.WTf, N0 / 16, L1.NBB, wave, lane); };
    fb3_bwd_step<2, MB, NW, DB2>(rb, L2.WTf, N1 / 16, L2.NBB, S.H2, N2 / 16, S.Q, wave, lane, primec);
    fb3_barrier();                                                       // B_b2
    auto none = [&]() {};
    fb3_bwd_step<G1P, MB, NW, DB1>(rc, L1.WTf, N0 / 16, L1.NBB, S.Q, N1 / 16, S.P, wave, lane, none);
    FB3_PSTAMP(7);
    fb3_barrier();                                                       // B_b1
}

template <int G1P, int NP, int NCT>
__device__ __forceinline__ void fb3_body(const FwdArgs& a, const FbLoss& L, const FusedNet& n, bool is_actor, char* smem) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < FB3_NC) fb3_compute<G1P, NP, NCT>(a, L, n, is_actor, smem, wave, lane);
    else fb3_service<NCT>(a, L, n, is_actor, smem, wave - FB3_NC, lane);
}

// XBot-L's shape pair: (first hidden width, input chunks) = (512, 6) for the actor, (768, 2) for the critic (fb3_supported, hgym_update3.hip)
#ifndef FB3_NO_KERNEL      // (hgym_fb4.hpp's translation unit takes the helpers above without a second copy of this kernel)
__global__ __launch_bounds__(FB3_THREADS) void mlp_fb3_kernel(const FwdArgs a, const FbLoss L) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int which = a.net0 + blockIdx.y;
    const FusedNet& n = a.net[which];
    if (n.layer[0].NB == 32) fb3_body<4, 1, 6>(a, L, n, which == 0, smem);
    else fb3_body<3, 2, 2>(a, L, n, which == 0, smem);
}
#endif

}  // namespace hgym

#ifdef HGYM_TU_CONTRACT_OFF
#pragma clang fp contract(off)
#endif
