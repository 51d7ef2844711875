// hgym_fb4.hpp -- mlp_fb4_kernel: the update's forward + PPO loss + dZ chain on 128-ROW tiles, eight compute + four service wavefronts (round 6).
//
// Why 128 rows: profiles/r06_fb3_role_specialised_wavefronts.txt -- a 64-row tile's first layer is paced by the L2 -> CU weight stream (786 KB per
// tile at ~80 GB/s per CU; 10.0 us against 4.3 us with an L1-resident stream), whatever the wave count, strip width or ring depth, and its other
// phases by LDS fragment latency in strips of 8 MFMAs per k-step.  Twice the rows per weight fragment halve the stream and give every k-step 16
// MFMAs.  Round 4's 128-row kernel (csrc/experiments/hgym_fb2.hpp) had that and lost it to its H / dZ stores: eight waves that store cannot hide
// the `vmcnt(0)` drains.  Here the compute wavefronts do not store (hgym_fb3.hpp's roles) -- and 128 rows fit 160 KB of LDS because H0 is never
// resident:
//
//   LDS: X (64 KB) | Y (64 KB) | biases | head partial sums | row indices | loss inputs.
//   layer 0   in passes of 256 output columns (8 waves x 2 column blocks x 8 row blocks; two passes for the actor, three for the critic).  The
//             input streams through X in 128-column chunks (the critic's 256 columns stay resident for all passes; the actor's 768 are re-staged
//             per pass, from L2 the second time).  Each pass leaves its piece of H0 in Y; the service waves copy it to HBM (where the
//             weight-gradient kernel wants it anyway) under the next pass.
//   layer 1   contracts over H0 in ascending k: the pieces of the earlier passes come BACK from L2 through X in 128-column chunks (written one pass
//             or more ago by this CU's service waves, drained with vmcnt(0) in front of a barrier before anything reads them), the last piece is
//             read in place from Y.  H1 then overwrites Y.
//   layer 2, head + loss, dZ3 -> dZ2 (in place, X), dZ2 -> dZ1 (in place, Y): as hgym_fb3.hpp, eight row blocks per wave; all eight compute waves
//             are head waves (one 16-row block each).
//   dZ1 -> dZ0  in the same 256-column passes: the service waves bring the pass's piece of H0 back into X, the compute waves turn it into dZ0 in
//             place (elu' from y = elu(z)), the service waves copy it out and load the next piece.
//   Compute wavefronts: global LOADS of weight fragments only.  Every barrier waits for LDS traffic only.
//
// Arithmetic: mlp_fb_kernel's, element for element (same fragments, ascending k on one accumulator chain per output, same epilogues, same loss
// code; loss partials written per 64 rows in the 64-row kernel's association order): bit-identical gradients (tests/test_fused_gpu.py).
#pragma once
#include "hgym_fb3.hpp"
#pragma clang fp contract(fast)

namespace hgym {

constexpr int FB4_BM = 128, FB4_MB = 8;
constexpr int FB4_X = 64 * 1024, FB4_Y = 64 * 1024;          // LDS regions
constexpr int FB4_CH = FB4_BM * FUSED_CHUNK * 2;             // one 128-column chunk buffer (32 KB); X holds two
#ifndef FB4_CLOCK_PASS
#define FB4_CLOCK_PASS 0
#endif
#ifndef FB4_D0
#define FB4_D0 2
#endif
#ifndef FB4_D1
#define FB4_D1 2
#endif
#ifndef FB4_D2
#define FB4_D2 4
#endif
#ifndef FB4_DB2
#define FB4_DB2 2
#endif
#ifndef FB4_DB1
#define FB4_DB1 2
#endif

struct Fb4Lds {
    char *X, *Y;
    float *bl, *red, *lin;
    int* rowidx;
};
HG_HD int fb4_lds_bytes(const FusedNet& n) {
    const int lin = n.layer[3].N == 1 ? FB4_BM * 2 * 4 : FB4_BM * FB_LIN_ACTOR * 4;
    return FB4_X + FB4_Y + fused_lds_bias(n) + FB4_MB * 32 * 4 + FB4_BM * 4 + lin;
}
__device__ __forceinline__ Fb4Lds fb4_lds(const FusedNet& n, char* smem) {
    Fb4Lds m;
    m.X = smem;
    m.Y = smem + FB4_X;
    m.bl = reinterpret_cast<float*>(smem + FB4_X + FB4_Y);
    m.red = reinterpret_cast<float*>(smem + FB4_X + FB4_Y + fused_lds_bias(n));
    m.rowidx = reinterpret_cast<int*>(m.red + FB4_MB * 32);
    m.lin = reinterpret_cast<float*>(m.rowidx + FB4_BM);
    return m;
}
// H2 / dZ2 (128 x 128 bf16) and the dZ3 tile (128 rows x 32 columns) live in X once layer 1 is through with the chunk buffers
constexpr int FB4_H2_OFF = 0, FB4_DZ3_OFF = 32 * 1024;

__device__ __forceinline__ void fb4_drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---------------------------------------------------------------------------------------------------------------- service wavefronts
// a 256-column piece of H0 / dZ0: LDS [8 row blocks][16 column blocks][512 B]  <->  global block layout (row block stride NB0 * 512 B)
template <bool NT>
__device__ __forceinline__ void fb4_piece_out(const char* __restrict__ lds, char* __restrict__ gbase, int NB0, int p, int vrb, int sl) {
#ifdef FB3_ABLATE_STORES
    return;
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int uu = h * 8 + u, rb = uu >> 1, off = (uu & 1) * 4096 + sl * 16;
            v[u] = *reinterpret_cast<const u32x4*>(lds + rb * 8192 + off);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int uu = h * 8 + u, rb = uu >> 1, off = (uu & 1) * 4096 + sl * 16;
            if (rb < vrb) st_stream_u4<NT>(gbase + ((int64_t)rb * NB0 + 16 * p) * 512 + off, v[u]);
        }
    }
}
__device__ __forceinline__ void fb4_piece_load(u32x4 (&v)[16], const char* __restrict__ gbase, int NB0, int p, int vrb, int sl) {
#pragma unroll
    for (int uu = 0; uu < 16; ++uu) {
        int rb = uu >> 1;
        rb = rb < vrb ? rb : vrb - 1;
        v[uu] = ld_stream_u4<true>(gbase + ((int64_t)rb * NB0 + 16 * p) * 512 + (uu & 1) * 4096 + sl * 16);
    }
}
__device__ __forceinline__ void fb4_piece_write(char* lds, const u32x4 (&v)[16], int sl) {
#pragma unroll
    for (int uu = 0; uu < 16; ++uu) *reinterpret_cast<u32x4*>(lds + (uu >> 1) * 8192 + (uu & 1) * 4096 + sl * 16) = v[uu];
}
// a 128-column chunk of H0 (8 column blocks from cb0) -> one chunk buffer [8 row blocks][8 column blocks][512 B]
__device__ __forceinline__ void fb4_hchunk_load(u32x4 (&v)[8], const char* __restrict__ gbase, int NB0, int j, int vrb, int sl) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int rb = u < vrb ? u : vrb - 1;
        v[u] = ld_stream_u4<true>(gbase + ((int64_t)rb * NB0 + 8 * j) * 512 + sl * 16);
    }
}
__device__ __forceinline__ void fb4_chunk_write(char* buf, const u32x4 (&v)[8], int sl) {
#pragma unroll
    for (int u = 0; u < 8; ++u) *reinterpret_cast<u32x4*>(buf + u * 4096 + sl * 16) = v[u];
}

template <int NP0, int NCT>
__device__ __forceinline__ void fb4_service(const FwdArgs& a, const FbLoss& L, const FusedNet& n, bool is_actor, char* smem, int sw, int lane, int vrb) {
    constexpr int BM = FB4_BM;
    constexpr int NG = (NP0 - 1) * 2;                    // layer-1 chunks that come back from L2
    constexpr bool RESTAGE = NCT > 2;                    // the input does not fit X: every pass stages it again
    static_assert(NCT % 2 == 0 && NCT >= 2 && NG >= 2 && NG <= 4, "chunk buffer parity / layer-1 chunk schedule");
    const Fb4Lds S = fb4_lds(n, smem);
    const int sl = sw * 64 + lane;
    const int m0 = blockIdx.x * BM;
    const FusedLayer &L0 = n.layer[0], &L1 = n.layer[1], &L2 = n.layer[2], &L3 = n.layer[3];
    const int NB0 = L0.NB, N1 = L1.N, N2 = L2.N, NBB3 = L3.NBB;
    char* H0g = reinterpret_cast<char*>(n.H[0]) + (int64_t)m0 * L0.N * 2;
    char* Z0g = reinterpret_cast<char*>(n.dZ[0]) + (int64_t)m0 * L0.N * 2;
    // ---- input rows from the bf16 shadow: item j of this lane is what lane `lane` of (virtual) wavefront sw * 8 + j stages: 16 consecutive lanes =
    //      8 rows x the two halves of one block row, lane groups = 4 consecutive column blocks, 16 row groups x 2 chunk halves
    const char* srow[8];
    int loff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = ((sw & 1) * 8 + j) * 8 + ((lane >> 1) & 7), cb = (sw >> 1) * 4 + (lane >> 4), hf = lane & 1;
        int m = m0 + row;
        m = m < a.M ? m : a.M - 1;
        const int64_t src = a.idx ? a.idx[m] : (int64_t)m;
        srow[j] = reinterpret_cast<const char*>(n.xb + src * n.ldxb + cb * 16 + hf * 8);
        loff[j] = ((row >> 4) * 8 + cb) * 512 + (row & 15) * 32 + hf * 16;
        if (cb == 0 && hf == 0) S.rowidx[row] = (int)src;
    }
    u32x4 stg[2][8];
    auto load_chunk = [&](int c, u32x4 (&st)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) st[j] = ld_stream_u4<(kFusedNT & 1) != 0>(srow[j] + c * (FUSED_CHUNK * 2));
    };
    auto write_chunk = [&](int c, const u32x4 (&st)[8]) {
        char* buf = S.X + (c & 1) * FB4_CH;
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(buf + loff[j]) = st[j];
    };
    load_chunk(0, stg[0]);
    // the four bias vectors, behind chunk 0 and in front of the others (loads return in order)
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
    load_chunk(1, stg[1]);
    // ---- layer 0: K0 chunk stagings; at every pass boundary the two piece barriers and the copy of the finished piece
    bool lin_pending = true;
    F4 v1[4], v2[2];
    float s2a = 0.0f, s2b = 0.0f;
    constexpr int NPASS = RESTAGE ? NP0 : 1;
#pragma unroll 1
    for (int p = 0; p < NPASS; ++p) {
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            write_chunk(c, stg[c & 1]);                  // (NCT is even: the chunk's buffer and register stage follow c's parity in every pass)
            if (p == 0 && c == 0) {
#pragma unroll
                for (int u = 0; u < BIT; ++u) {
                    const int i = sl + u * FB3_SL;
                    if (i < bn3) S.bl[i] = bv[u];
                }
            }
            if (c + 2 < NCT || p + 1 < NPASS) load_chunk((c + 2) % NCT, stg[c & 1]);
            if (c == 0 && p > 0) {                       // first chunk of pass p > 0: the piece of pass p - 1
                fb3_barrier();                           // "Y free" (p - 1)
                fb4_drain_vm();
                fb3_barrier();                           // "piece p - 1 ready"
                fb4_piece_out<(kFusedNT & 2) != 0>(S.Y, H0g, NB0, p - 1, vrb, sl);
            }
            if (p == FB4_CLOCK_PASS) FB3_WSTAMP(FB3_NC + sw, lane, 2 * c);
            fb3_barrier();                               // "chunk ready"
            if (p == FB4_CLOCK_PASS) FB3_WSTAMP(FB3_NC + sw, lane, 2 * c + 1);
            if (p == 0 && c == 0) {
                // loss inputs of the tile's rows (mlp_fb_kernel's l2idle on 512 lanes per 64 rows; here four rounds of 256 over 128 rows): issued
                // behind the first chunks, written to LDS when layer 0's stagings are through
                if (is_actor) {
#pragma unroll
                    for (int rd = 0; rd < 4; ++rd) {
                        const int j = sl + rd * FB3_SL, rw = j >> 3, kk = j & 7;
                        const int64_t ri = S.rowidx[rw];
                        const float* base = kk < 3 ? L.actions : (kk < 6 ? L.old_mu : L.old_sigma);
                        v1[rd] = *reinterpret_cast<const F4*>(base + ri * 12 + 4 * (kk < 3 ? kk : (kk < 6 ? kk - 3 : kk - 6)));
                    }
                    const int64_t ria = S.rowidx[sl & 127];
                    v2[0] = *reinterpret_cast<const F4*>(L.old_sigma + ria * 12 + 8);           // lanes 0..127: row sl
                    s2a = (sl < 128 ? L.advantages : L.logp)[ria];                             // advantage (0..127) / old log-prob (128..255) of row sl & 127
                } else {
                    const int64_t ri = S.rowidx[sl & 127];
                    s2a = (sl < 128 ? L.returns : L.values)[ri];
                }
            }
        }
    }
    if (lin_pending) {
        if (is_actor) {
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) {
                const int j = sl + rd * FB3_SL;
                *reinterpret_cast<F4*>(S.lin + (j >> 3) * FB_LIN_ACTOR + 4 * (j & 7)) = v1[rd];
            }
            if (sl < 128) *reinterpret_cast<F4*>(S.lin + sl * FB_LIN_ACTOR + 32) = v2[0];
            S.lin[(sl & 127) * FB_LIN_ACTOR + (sl < 128 ? 36 : 37)] = s2a;
        } else {
            S.lin[(sl & 127) * 2 + (sl >> 7)] = s2a;
        }
        (void)s2b; (void)v2[1];
    }
    // remaining pieces (all of them when the input stayed resident; the last one otherwise)
#pragma unroll 1
    for (int p = RESTAGE ? NP0 - 1 : 0; p < NP0; ++p) {
        fb3_barrier();                                   // "Y free" (p)
        fb4_drain_vm();                                  // every earlier piece's stores are in L2 before the barrier that lets anyone read them back
        fb3_barrier();                                   // "piece p ready"
        if (p + 1 < NP0) fb4_piece_out<(kFusedNT & 2) != 0>(S.Y, H0g, NB0, p, vrb, sl);
    }
    // ---- layer 1: the earlier pieces of H0 come back from L2 in 128-column chunks; the last piece is read in place from Y and copied out when
    //      the chunk stagings are through (no load of this wave ever waits behind those stores)
    {
        u32x4 hs[2][8];
        fb4_hchunk_load(hs[0], H0g, NB0, 0, vrb, sl);
        fb4_hchunk_load(hs[1], H0g, NB0, 1, vrb, sl);
        fb4_chunk_write(S.X, hs[0], sl);
        fb4_chunk_write(S.X + FB4_CH, hs[1], sl);
        if (NG > 2) {
            fb4_hchunk_load(hs[0], H0g, NB0, 2, vrb, sl);
            fb4_hchunk_load(hs[1], H0g, NB0, 3, vrb, sl);
        }
        fb3_barrier();                                   // "layer-1 chunk 0 ready"
        fb3_barrier();                                   // "layer-1 chunk 1 ready"
        if (NG > 2) {
            fb4_chunk_write(S.X, hs[0], sl);
            fb3_barrier();                               // chunk 2
            fb4_chunk_write(S.X + FB4_CH, hs[1], sl);
            fb3_barrier();                               // chunk 3
        }
    }
    fb4_piece_out<(kFusedNT & 2) != 0>(S.Y, H0g, NB0, NP0 - 1, vrb, sl);
    constexpr bool NTH = (kFusedNT & 2) != 0, NTZ = (kFusedNT & 4) != 0;
    const int64_t row0 = (int64_t)m0;
    fb3_barrier();                                       // "layer 1 done reading"
    fb3_barrier();                                       // "H1 ready" (Y)
    fb3_copy_out<NTH>(S.Y, reinterpret_cast<char*>(n.H[1]) + row0 * N1 * 2, vrb * 16 * N1 * 2, sl);
    fb3_barrier();                                       // "H2 ready" (X)
    fb3_copy_out<NTH>(S.X + FB4_H2_OFF, reinterpret_cast<char*>(n.H[2]) + row0 * N2 * 2, vrb * 16 * N2 * 2, sl);
    fb3_barrier();                                       // "dZ3 ready"
    fb3_copy_out<false>(S.X + FB4_DZ3_OFF, reinterpret_cast<char*>(n.dZ[3]) + row0 * 64 * NBB3, vrb * 16 * 64 * NBB3, sl);
    if (sl < 64) {                                       // loss partials per 64 rows, the 64-row kernel's association order
        const int t = sl & 31, half = sl >> 5;
        const bool mine = is_actor ? (t != 1 && t < 28) : (t == 1 || t == 28);
        const float* rr = S.red + half * 128;
        if (mine && 2 * (int)blockIdx.x + half < (a.M + 63) / 64)
            L.partials[((int64_t)blockIdx.x * 2 + half) * 32 + t] = rr[t] + rr[32 + t] + rr[64 + t] + rr[96 + t];
    }
    fb3_barrier();                                       // "dZ2 ready" (X, in place)
    u32x4 hp[16];
    fb4_piece_load(hp, H0g, NB0, 0, vrb, sl);            // H0 piece 0 for dZ0's first pass: requested in front of the stores below
    fb3_copy_out<NTZ>(S.X + FB4_H2_OFF, reinterpret_cast<char*>(n.dZ[2]) + row0 * N2 * 2, vrb * 16 * N2 * 2, sl);
    fb3_barrier();                                       // "dZ1 ready" (Y, in place); X is dead
#pragma unroll 1
    for (int p = 0; p < NP0; ++p) {
        fb4_piece_write(S.X, hp, sl);
        fb3_barrier();                                   // "H0 piece p in X"
        if (p == 0) fb3_copy_out<NTZ>(S.Y, reinterpret_cast<char*>(n.dZ[1]) + row0 * N1 * 2, vrb * 16 * N1 * 2, sl);
        if (p + 1 < NP0) fb4_piece_load(hp, H0g, NB0, p + 1, vrb, sl);
        fb3_barrier();                                   // "dZ0 piece p ready"
        fb4_piece_out<NTZ>(S.X, Z0g, NB0, p, vrb, sl);
    }
}

// ---------------------------------------------------------------------------------------------------------------- compute wavefronts
// bias + ELU of a pass's strip -> LDS piece (16 column blocks per row block); epilogue_elu_t with the bias / LDS column origins apart
template <int G, int MB>
__device__ __forceinline__ void fb4_epilogue(f32x4 (&acc)[MB][G], const float* __restrict__ bias, int nb_abs, char* out_lds, int CBo, int nb_rel, int lane) {
    const int r = lane & 15, q = lane >> 4;
    const int loff = r * 32 + q * 8;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const F4 b = *reinterpret_cast<const F4*>(bias + (nb_abs + g) * 16 + 4 * q);
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = elu_bf(acc[i][g][e] + b.v[e]);
            *reinterpret_cast<u32x2*>(out_lds + (i * CBo + nb_rel + g) * 512 + loff) = pack_bf16x4(v[0], v[1], v[2], v[3]);
        }
    }
}

template <int NP0, int NCT>
__device__ __forceinline__ void fb4_compute(const FwdArgs& a, const FbLoss& L, const FusedNet& n, bool is_actor, char* smem, int wave, int lane) {
    // Wave -> (g, h): row half h (row blocks 4h .. 4h + 3) x column group g: every wave runs a 4 x 4 block register tile (hgym_fb3.hpp's, which
    // reaches the matrix pipe's rate when its weights are free), i.e. 4 KB of LDS fragments and 4 KB of weight fragments per 16 MFMAs.  The two
    // waves of a column group load the SAME weight fragments at the same time: one of the two requests is an L1 (or pending-miss) hit, the L2 ->
    // CU stream is that of one 128-row strip.  (8 x 2 tiles -- every fragment loaded once -- read 8 KB of LDS per 16 MFMAs: measured 2.0 us per
    // chunk against 0.72, the LDS port's rate.)
    constexpr int BM = FB4_BM, MB = 4, D = FB4_D0, D1 = FB4_D1, D2 = FB4_D2, DB2 = FB4_DB2, DB1 = FB4_DB1;
    constexpr int NG = (NP0 - 1) * 2;
    constexpr bool RESTAGE = NCT > 2;
    const Fb4Lds S = fb4_lds(n, smem);
    const int r = lane & 15;
    const int g = wave & 3, h = wave >> 2;
    const int m0 = blockIdx.x * BM;
    const FusedLayer &L0 = n.layer[0], &L1 = n.layer[1], &L2 = n.layer[2], &L3 = n.layer[3];
    const float* bl = S.bl;
    const int NBB3 = L3.NBB, N1 = L1.N, N2 = L2.N;
    const int rbo = 4 * h;                                   // first row block of this wave's half
    WRing<4, D> r0;
    WRing<4, D1> r1;
    WRing<2, D2> r2;
    WRing<1, 4> r3;
    WRing<2, 2> ra;
    WRing<4, DB2> rb;
    WRing<4, DB1> rc;
    FB3_PSTAMP(0);
    // ---------------------------------------------------------------- layer 0, NP0 passes of 256 output columns
    wring_prime<4, D>(r0, L0.Wf + HG_WOFF((int64_t)(g * 4) * L0.KB * 64) + lane, HG_WSTR(L0.KB * 64), L0.KB);
#pragma unroll 1
    for (int p = 0; p < NP0; ++p) {
        const int nbr = g * 4, nba = p * 16 + nbr;
        const u32x4* wl0 = L0.Wf + HG_WOFF((int64_t)nba * L0.KB * 64) + lane;
        const char* xh = S.X + rbo * 8 * 512;
        f32x4 acc[MB][4];
        zero_acc<4, MB>(acc);
        const bool bar = RESTAGE || p == 0;
        for (int c = 0; c + 1 < NCT; ++c) {
            if (bar) fb3_barrier();                                      // "chunk ready"
            if (p == 0 && c == 0) FB3_PSTAMP(1);
            if (p == FB4_CLOCK_PASS) FB3_WSTAMP(wave, lane, 2 * c);
            mma_chunk<4, MB, D, false, 1>(r0, wl0, HG_WSTR(L0.KB * 64), c * 4, xh + (c & 1) * FB4_CH, 8, lane, acc);
            if (p == FB4_CLOCK_PASS) FB3_WSTAMP(wave, lane, 2 * c + 1);
        }
        if (bar) fb3_barrier();
        if (p == FB4_CLOCK_PASS) FB3_WSTAMP(wave, lane, 2 * (NCT - 1));
        mma_chunk<4, MB, D, true, 1>(r0, wl0, HG_WSTR(L0.KB * 64), (NCT - 1) * 4, xh + ((NCT - 1) & 1) * FB4_CH, 8, lane, acc);
        if (p == FB4_CLOCK_PASS) FB3_WSTAMP(wave, lane, 2 * (NCT - 1) + 1);
        if (p + 1 < NP0) wring_prime<4, D>(r0, L0.Wf + HG_WOFF((int64_t)((p + 1) * 16 + nbr) * L0.KB * 64) + lane, HG_WSTR(L0.KB * 64), L0.KB);
        else {
            FB3_PSTAMP(2);
            wring_prime<4, D1>(r1, L1.Wf + HG_WOFF((int64_t)nbr * L1.KB * 64) + lane, HG_WSTR(L1.KB * 64), L1.KB);
        }
        fb3_barrier();                                                   // "Y free"
        if (p == FB4_CLOCK_PASS) FB3_WSTAMP(wave, lane, 12);
        fb4_epilogue<4, MB>(acc, bl, nba, S.Y + rbo * 16 * 512, 16, nbr, lane);
        if (p == FB4_CLOCK_PASS) FB3_WSTAMP(wave, lane, 13);
        fb3_barrier();                                                   // "piece p ready"
        if (p == FB4_CLOCK_PASS) FB3_WSTAMP(wave, lane, 14);
    }
    FB3_PSTAMP(3);
    // ---------------------------------------------------------------- layer 1: K ascending -- NG chunks through X, the last piece in place from Y
    {
        const int nbr = g * 4;
        const u32x4* wl1 = L1.Wf + HG_WOFF((int64_t)nbr * L1.KB * 64) + lane;
        const char* xh = S.X + rbo * 8 * 512;
        const char* yh = S.Y + rbo * 16 * 512;
        f32x4 acc[MB][4];
        zero_acc<4, MB>(acc);
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            fb3_barrier();                                               // "layer-1 chunk j ready"
            mma_chunk<4, MB, D1, false, 1>(r1, wl1, HG_WSTR(L1.KB * 64), j * 4, xh + (j & 1) * FB4_CH, 8, lane, acc);
        }
        mma_chunk<4, MB, D1, false, 1>(r1, wl1, HG_WSTR(L1.KB * 64), NG * 4, yh, 16, lane, acc);
        mma_chunk<4, MB, D1, true, 1>(r1, wl1, HG_WSTR(L1.KB * 64), NG * 4 + 4, yh + 8 * 512, 16, lane, acc);
        hidden_prime<2, D2>(r2, L2, g, lane);
        fb3_barrier();                                                   // "layer 1 done reading"
        fb4_epilogue<4, MB>(acc, bl + L0.N, nbr, S.Y + rbo * 16 * 512, 16, nbr, lane);
    }
    fb3_barrier();                                                       // "H1 ready"
    FB3_PSTAMP(4);
    auto prime3 = [&]() {
        wring_prime<1, 4>(r3, L3.Wf + lane, 0, L3.KB);
        bwd_prime<2, 2>(ra, L3.WTf, N2 / 16, NBB3, g, lane);
    };
    hidden_layer<2, MB, 4, D2, true>(r2, L2, bl + L0.N + L1.N, S.Y + rbo * 16 * 512, L1.NB, S.X + FB4_H2_OFF + rbo * (N2 / 16) * 512, nullptr, 0, g, lane, prime3);
    fb3_barrier();                                                       // "H2 ready"
    FB3_PSTAMP(5);
    // ---------------------------------------------------------------- head + PPO loss: every compute wave owns one 16-row block
    {
        f32x4 hacc[1][1];
        hacc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int CB3 = L2.NB;
        const char* h2 = S.X + FB4_H2_OFF + wave * CB3 * 512;
        mma_stream<1, 1, 4>(r3, L3.Wf + lane, 0, L3.KB, h2, CB3, lane, hacc);      // (third hidden width 128: 4 k-steps)
        const int q = lane >> 4;
        float mu[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) mu[e] = hacc[0][0][e] + ((4 * q + e < L3.N) ? bl[L0.N + L1.N + L2.N + 4 * q + e] : 0.0f);
        Fb3Lds S3;
        S3.P = S3.Q = S3.H2 = nullptr;
        S3.R0 = S.X + FB4_DZ3_OFF;
        S3.bl = S.bl; S3.red = S.red; S3.lin = S.lin; S3.rowidx = S.rowidx;
        fb3_head(a, L, S3, is_actor, wave, m0 + wave * 16 + r, lane, mu);
    }
    FB3_PSTAMP(6);
    fb3_barrier();                                                       // "dZ3 ready"
    // ---------------------------------------------------------------- dZ chain (row half h, column group g)
    auto primeb = [&]() { bwd_prime<4, DB2>(rb, L2.WTf, N1 / 16, L2.NBB, g, lane); };
    fb3_bwd_step<2, MB, 4, 2, 2>(ra, L3.WTf, N2 / 16, NBB3, S.X + FB4_DZ3_OFF + rbo * 2 * NBB3 * 512, 2 * NBB3,
                                 S.X + FB4_H2_OFF + rbo * (N2 / 16) * 512, g, lane, primeb);
    fb3_barrier();                                                       // "dZ2 ready"
    auto none = [&]() {};
    auto primec = [&]() { wring_prime<4, DB1>(rc, L1.WTf + HG_WOFF((int64_t)(g * 4) * L1.NBB * 64) + lane, HG_WSTR(L1.NBB * 64), L1.NBB); };
    fb3_bwd_step<4, MB, 4, DB2, 1>(rb, L2.WTf, N1 / 16, L2.NBB, S.X + FB4_H2_OFF + rbo * (N2 / 16) * 512, N2 / 16, S.Y + rbo * 16 * 512, g, lane, none);
    primec();
    fb3_barrier();                                                       // "dZ1 ready"
    // dZ0 in NP0 passes of 256 columns: (dZ1 * W1) .* elu'(H0 piece), the piece brought back into X by the service waves, in place
#pragma unroll 1
    for (int p = 0; p < NP0; ++p) {
        const int nbr = g * 4;
        [[maybe_unused]] const int nba = p * 16 + nbr;
        const int NBBc = L1.NBB;
        f32x4 acc[MB][4];
        zero_acc<4, MB>(acc);
        const u32x4* wl = L1.WTf + HG_WOFF((int64_t)nba * NBBc * 64) + lane;
        mma_stream<4, MB, DB1, 1>(rc, wl, HG_WSTR(NBBc * 64), NBBc, S.Y + rbo * 16 * 512, N1 / 16, lane, acc);      // (fb4_supported: layer 1 is 256 wide, 8 k-steps)
        if (p + 1 < NP0) wring_prime<4, DB1>(rc, L1.WTf + HG_WOFF((int64_t)((p + 1) * 16 + nbr) * NBBc * 64) + lane, HG_WSTR(NBBc * 64), NBBc);
        fb3_barrier();                                                   // "H0 piece p in X"
        const int q = lane >> 4, loff = r * 32 + q * 8;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                char* ptr = S.X + ((rbo + i) * 16 + nbr + gg) * 512 + loff;
                const u32x2 y2 = *reinterpret_cast<const u32x2*>(ptr);
                const unsigned int w0 = y2[0], w1 = y2[1];
                const float y[4] = {bf16_bits_to_f32(w0 & 0xffffu), bf16_bits_to_f32(w0 >> 16), bf16_bits_to_f32(w1 & 0xffffu), bf16_bits_to_f32(w1 >> 16)};
                float d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = acc[i][gg][e] * ((y[e] > 0.0f) ? 1.0f : (y[e] + 1.0f));
                *reinterpret_cast<u32x2*>(ptr) = pack_bf16x4(d[0], d[1], d[2], d[3]);
            }
        fb3_barrier();                                                   // "dZ0 piece p ready"
    }
    FB3_PSTAMP(7);
}

template <int NP0, int NCT>
__device__ __forceinline__ void fb4_body(const FwdArgs& a, const FbLoss& L, const FusedNet& n, bool is_actor, char* smem) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // valid 16-row blocks of this tile: the activation buffers are sized for the batch rounded up to 64 rows
    int vrb = (((a.M + 63) / 64) * 64 - (int)blockIdx.x * FB4_BM) / 16;
    vrb = vrb < FB4_MB ? vrb : FB4_MB;
#ifdef FB4_ONLY_COMPUTE      // (register census builds)
    if (wave < FB3_NC) fb4_compute<NP0, NCT>(a, L, n, is_actor, smem, wave, lane);
#elif defined(FB4_ONLY_SERVICE)
    if (wave >= FB3_NC) fb4_service<NP0, NCT>(a, L, n, is_actor, smem, wave - FB3_NC, lane, vrb);
#else
    if (wave < FB3_NC) fb4_compute<NP0, NCT>(a, L, n, is_actor, smem, wave, lane);
    else fb4_service<NP0, NCT>(a, L, n, is_actor, smem, wave - FB3_NC, lane, vrb);
#endif
}

// XBot-L's shape pair: (first hidden width, input chunks) = (512, 6) for the actor, (768, 2) for the critic (fb4_supported, hgym_update4.hip)
__global__ __launch_bounds__(FB3_THREADS) void mlp_fb4_kernel(const FwdArgs a, const FbLoss L) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int which = a.net0 + blockIdx.y;
    const FusedNet& n = a.net[which];
    if (n.layer[0].NB == 32) fb4_body<2, 6>(a, L, n, which == 0, smem);
    else fb4_body<3, 2>(a, L, n, which == 0, smem);
}

}  // namespace hgym

#ifdef HGYM_TU_CONTRACT_OFF
#pragma clang fp contract(off)
#endif
