// hgym_fb2.hpp -- the update's forward + PPO loss + dZ chain on 128-ROW tiles (round 4): `mlp_fb2_kernel`.
//
// Why another tile shape.  mlp_fb_kernel (hgym_fused.hpp) runs 64-row tiles on sixteen 128-register wavefronts.  Per 64 rows it
// streams the whole weight set (1.3-1.45 MB of fragments) from L2 into registers, and per k-step of the first layer the three
// consumers are balanced -- 512 cycles of MFMA per SIMD, 32 KiB of weights at the CU's ~36..56 B/clk fill rate, 64 KiB of LDS
// operand reads -- so none of them is ever hidden behind another: 0.23 of the MFMA peak, waves parked 62 % of the time
// (gpurun_out/r03h_sq_pmc_summary.txt).  A weight fragment that serves 8 row blocks instead of 4 halves the L2 -> CU stream per
// flop, eight 256-register wavefronts halve the LDS operand re-reads per flop (every wave reads the whole input tile for its
// column strip), and every barrier / epilogue / head phase is paid once per 128 rows.
//
// What made 128 rows fit 160 KB of LDS: the first hidden activation H0 is never resident together with anything else.
//   * net with a WIDE input (actor, 705 -> 768 columns): the input streams through LDS in 128-column chunks (2 x 32 KB) while
//     each wave accumulates 8 row blocks x (N0 / 128) column blocks of H0 in registers (128 for N0 = 512); H0 (128 KB) then
//     overwrites the chunk buffers, the second layer accumulates H1 IN REGISTERS (64) until every wave has finished reading H0,
//     and H1 is written over it.
//   * net with a NARROW input (critic, 219 -> 256 columns): the input tile (64 KB) stays resident, H0 is produced in 256-column
//     pieces (64 KB), each consumed at once by the second layer's partial sums (same k order as the unchunked sum: bit-identical).
//   Either way H0 is gone when the dZ chain reaches the first layer: elu'(H0) is re-read from the H0 this tile wrote to HBM for the
//   weight-gradient kernel ~30 us earlier (an L2 / Infinity-Cache hit), prefetched under the MFMAs of its strip.
//   LDS: 128 KB activations + biases + row indices + the gathered loss inputs = 153.5 KB.
//
// Tiles are 7 or 8 row blocks high, chosen on the host so that the tile count per net is a multiple of the CU count when the
// batch allows (61 440 rows = 512 tiles = two per CU and net, no tail round); a row block a tile does not own is computed
// (the MFMAs are unconditional) and never stored.
//
// Everything else is mlp_fb_kernel's: the weight ring, the block layout, the loss arithmetic (one lane per (row, 4 actions)),
// H / dZ in HBM exactly where dw_kernel_rs expects them.  Per element every sum runs over k in the same order as in the 64-row
// kernel, so H, dZ and the weight gradients are bit-identical to it; the loss partials are written per 16-row block and grouped by
// ppo_scalars_block exactly as the 64-row kernel groups its head waves, so the loss scalars and the std / head-bias gradients are
// bit-identical too (tests/test_fused_gpu.py: the shadow path, which takes this kernel, against the fp32-row path, which does not).
#pragma once
#include <algorithm>

#include "hgym_fb2_api.hpp"
#pragma clang fp contract(fast)

namespace hgym {

// Ring depths: the full-width first layer of a streamed input (4 fragments per k-step, 128 accumulator registers) affords two k-steps
// in flight; every narrower strip (1-2 fragments per k-step) needs FOUR -- with eight wavefronts per CU instead of sixteen, two
// k-steps of a 2-fragment strip are 32 KiB in flight per CU, a quarter of what the ~1 us L2 latency needs at the fill rate.
#ifndef FB2_D1
#define FB2_D1 2
#endif
#ifndef FB2_XB
#define FB2_XB 1          // x fragments of the narrow strips one k-step ahead (64 registers instead of 32)
#endif
#ifndef FB2_ABL
#define FB2_ABL 0         // timing ablations (WRONG results): 1 no H0 re-read in the last dZ step, 2 no H / dZ stores to HBM
#endif
constexpr int FB2_BM = 128, FB2_NW = 8, FB2_MB = 8, FB2_D = 2;
constexpr int FB2_P_BYTES = 128 * 1024;
constexpr int FB2_LIN_ACTOR = FB_LIN_ACTOR;

HG_HD int fb2_lds_bytes(const FusedNet& n) {
    const int lin = n.layer[3].N == 1 ? FB2_BM * 2 * 4 : FB2_BM * FB2_LIN_ACTOR * 4;
    return FB2_P_BYTES + fused_lds_bias(n) + FB2_BM * 4 + lin;
}

// bias + ELU -> bf16 -> LDS block layout (column block cbl0 + g of a matrix with CBl column blocks) and, for the row blocks this
// tile owns, the same blocks in HBM (column block cbg0 + g of CBg)
template <int G, int MB>
__device__ __forceinline__ void epi2(f32x4 (&acc)[MB][G], const float* __restrict__ bias, char* lds, int CBl, int cbl0,
                                     __bf16* __restrict__ Hg, int CBg, int cbg0, int64_t mbg0, int nblk, int lane) {
    const int r = lane & 15, q = lane >> 4;
    const int loff = r * 32 + q * 8;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const F4 b = *reinterpret_cast<const F4*>(bias + g * 16 + 4 * q);
#pragma unroll
        for (int i = 0; i < MB; i += 2) {
            u32x2 pk[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = elu_bf(acc[i + h][g][e] + b.v[e]);
                pk[h] = pack_bf16x4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<u32x2*>(lds + ((i + h) * CBl + cbl0 + g) * 512 + loff) = pk[h];
            }
            if (!(FB2_ABL & 2)) {
                char* pa = reinterpret_cast<char*>(Hg) + ((mbg0 + i) * CBg + cbg0 + g) * 512 + loff;
                st_pair<(HGYM_NT & 2) != 0>(pa, pa + (int64_t)CBg * 512, pk[0], pk[1], i < nblk, i + 1 < nblk, q);
            }
        }
    }
}

// dZ_out = (dZ_in * W) .* elu'(y) for the strip of G column blocks starting at nb0: W^T fragments as the MFMA A operand, dZ_in in LDS
// (CBin column blocks), y = elu(z) either in LDS (H_lds, NBo column blocks; out_lds may be the same buffer: every lane reads an
// entry and later writes that very entry) or in HBM (Hg: issued before the MFMAs, consumed after).  R: primed with this strip.
template <int G, int MB, int D, bool HGLOBAL, int GR>
__device__ __forceinline__ void bwd2_strip(WRing<GR, D>& R, const u32x4* __restrict__ WTf, int NBo, int NBBc, int nb0, const char* in_lds,
                                           int CBin, char* out_lds, const char* H_lds, const __bf16* __restrict__ Hg,
                                           __bf16* __restrict__ dZg, int64_t mbg0, int nblk, int lane) {
    const int r = lane & 15, q = lane >> 4;
    const int loff = r * 32 + q * 8;
    u32x2 aux[MB][G];
    if (HGLOBAL && (FB2_ABL & 1)) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int g = 0; g < G; ++g) aux[i][g] = (u32x2){0x3f803f80u, 0x3f803f80u};
    } else if (HGLOBAL) {
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int64_t ib = i < nblk ? i : 0;          // a row block of another tile: any valid address (the result is not stored)
#pragma unroll
            for (int g = 0; g < G; ++g)
                aux[i][g] = ld_stream_u2<(HGYM_NT & 4) != 0>(reinterpret_cast<const char*>(Hg) + ((mbg0 + ib) * NBo + nb0 + g) * 512 + loff);
        }
    }
    f32x4 acc[MB][G];
    zero_acc<G, MB>(acc);
    const u32x4* wl = WTf + (int64_t)nb0 * NBBc * 64 + lane;
    if (NBBc % D == 0) mma_stream<G, MB, D, FB2_XB>(R, wl, NBBc * 64, NBBc, in_lds, CBin, lane, acc);
    else mma_ring<G, MB, D, FB2_XB>(R, wl, NBBc * 64, 0, NBBc, NBBc, in_lds, CBin, lane, acc);
    if (!HGLOBAL) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int g = 0; g < G; ++g) aux[i][g] = *reinterpret_cast<const u32x2*>(H_lds + (i * NBo + nb0 + g) * 512 + loff);
    }
#pragma unroll
    for (int i = 0; i < MB; i += 2)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            u32x2 pk[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned int w0 = aux[i + h][g][0], w1 = aux[i + h][g][1];
                const float y[4] = {bf16_bits_to_f32(w0 & 0xffffu), bf16_bits_to_f32(w0 >> 16), bf16_bits_to_f32(w1 & 0xffffu),
                                    bf16_bits_to_f32(w1 >> 16)};
                float d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = acc[i + h][g][e] * ((y[e] > 0.0f) ? 1.0f : (y[e] + 1.0f));   // elu'(z) from y = elu(z)
                pk[h] = pack_bf16x4(d[0], d[1], d[2], d[3]);
                if (out_lds) *reinterpret_cast<u32x2*>(out_lds + ((i + h) * NBo + nb0 + g) * 512 + loff) = pk[h];
            }
            if (!(FB2_ABL & 2)) {
                char* pa = reinterpret_cast<char*>(dZg) + ((mbg0 + i) * NBo + nb0 + g) * 512 + loff;
                st_pair<(HGYM_NT & 4) != 0>(pa, pa + (int64_t)NBo * 512, pk[0], pk[1], i < nblk, i + 1 < nblk, q);
            }
        }
}

// mma_chunk (hgym_fused.hpp) for 8 row blocks with the x fragments fetched in two halves of four: 16 registers instead of 32 in
// the one loop that holds 128 accumulators (the SGPR-heavy kernel spills uniform values into whatever vector registers are left)
template <int G, int D, bool LAST, int GR>
__device__ __forceinline__ void mma_chunk8(WRing<GR, D>& R, const u32x4* __restrict__ wl, int wstride, int t0, const char* xl, int CBx,
                                           int lane, f32x4 (&acc)[8][G]) {
    static_assert(D == 2 || D == 4, "ring depth must divide the chunk");
    const int r = lane & 15, q = lane >> 4;
    const char* xb = xl + (q >> 1) * 512 + r * 32 + (q & 1) * 16;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int d = s4 % D;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 xa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xa[i] = *reinterpret_cast<const u32x4*>(xb + ((4 * h + i) * CBx + 2 * s4) * 512);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < G; ++g) mma_frag<__bf16>(R.w[d][g], xa[i], acc[4 * h + i][g]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!LAST || s4 + D < 4) {
#pragma unroll
            for (int g = 0; g < G; ++g) R.w[d][g] = wl[(int64_t)g * wstride + (t0 + s4 + D) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// NCH0: 256-column pieces of the first hidden width (2: 512, 3: 768).  STREAM: the input is wider than two 128-column chunks and
// streams through LDS (needs NCH0 <= 2: the full-width accumulators); otherwise it stays resident and H0 is produced piecewise.
template <int NCH0, bool STREAM>
__device__ __forceinline__ void fb2_body(const FwdArgs& a, const FbLoss& L, const Fb2Sched& sch, const FusedNet& n, bool is_actor, char* smem) {
    constexpr int BM = FB2_BM, NW = FB2_NW, MB = FB2_MB, D = FB2_D, D1 = FB2_D1, XB = FB2_XB;
    static_assert(!STREAM || NCH0 <= 2, "full-width first-layer accumulators: 8 x 4 blocks per wave at most");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x;
    const int mb0 = (int)(((int64_t)tile * sch.nb) / sch.T), mb1 = (int)(((int64_t)(tile + 1) * sch.nb) / sch.T);
    const int nblk = mb1 - mb0;                       // 1 .. 8 row blocks owned by this tile
    const int64_t mbg0 = mb0;
    const int m0 = mb0 * 16;
    const FusedLayer& L0 = n.layer[0];
    const FusedLayer& L1 = n.layer[1];
    const FusedLayer& L2 = n.layer[2];
    const FusedLayer& L3 = n.layer[3];
    char* P = smem;
    char* RA = smem;                                  // H1 / dZ1: 128 x 256 bf16
    char* RB = smem + 64 * 1024;                      // H2 / dZ2: 128 x 128 bf16
    char* R0 = smem + 96 * 1024;                      // dZ3: 128 x 32 bf16
    float* bl = reinterpret_cast<float*>(smem + FB2_P_BYTES);
    int* rowidx = reinterpret_cast<int*>(smem + FB2_P_BYTES + fused_lds_bias(n));
    float* lin = reinterpret_cast<float*>(rowidx + BM);
    const int A = a.A;
    const float invB = 1.0f / (float)a.M;

    // ---- bias vectors -> LDS (loads issued first, parked in registers, written next to the first input chunk)
    constexpr int BIT = (768 + 256 + 128 + 16 + NW * 64 - 1) / (NW * 64);
    float bv[BIT];
    const int bn0 = L0.N, bn1 = bn0 + L1.N, bn2 = bn1 + L2.N, bn3 = bn2 + 16;
#pragma unroll
    for (int u = 0; u < BIT; ++u) {
        int i = tid + u * NW * 64;
        i = i < bn2 + L3.N ? i : bn2 + L3.N - 1;
        const float* src = i < bn0 ? L0.bias + i : (i < bn1 ? L1.bias + (i - bn0) : (i < bn2 ? L2.bias + (i - bn1) : L3.bias + (i - bn2)));
        bv[u] = *src;
    }
    auto bias_to_lds = [&]() {
#pragma unroll
        for (int u = 0; u < BIT; ++u) {
            const int i = tid + u * NW * 64;
            if (i < bn3) bl[i] = bv[u];
        }
    };

    // ---- input staging from the bf16 shadow: a chunk is 128 rows x 256 B = 2048 16-byte items, four per lane.  Lane map as in
    // mlp_fb_kernel<XB16>: 16 consecutive lanes = 8 rows x the two halves of one block row (conflict-free 256-byte LDS write), the
    // four lane groups = four consecutive column blocks; item u of wave w = row group (4 w + u) >> 1 (8 rows), chunk half (4 w + u) & 1.
    const char* srow[4];
    int loffs[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int combo = wave * 4 + u;
        const int row = (combo >> 1) * 8 + ((lane >> 1) & 7), cb = (combo & 1) * 4 + (lane >> 4), hf = lane & 1;
        int m = m0 + row;
        m = m < a.M ? m : a.M - 1;
        const int64_t src = a.idx ? a.idx[m] : (int64_t)m;
        srow[u] = reinterpret_cast<const char*>(n.xb + src * n.ldxb + cb * 16 + hf * 8);
        if (cb == 0 && hf == 0) rowidx[row] = (int)src;
        loffs[u] = (row >> 4) * 512 * (STREAM ? 8 : 16) + cb * 512 + (row & 15) * 32 + hf * 16;
    }
    u32x4 stg[4];
    auto stage_load = [&](int c) {
#pragma unroll
        for (int u = 0; u < 4; ++u) stg[u] = ld_stream_u4<(HGYM_NT & 1) != 0>(srow[u] + c * (FUSED_CHUNK * 2));
    };
    auto stage_write = [&](char* base) {
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<u32x4*>(base + loffs[u]) = stg[u];
    };

    f32x4 acc1[MB][2];                                 // second-layer pre-activations of this wave's 32 columns, all 128 rows
    WRing<2, D1> r1;
    const u32x4* wl1 = L1.Wf + (int64_t)(wave * 2) * L1.KB * 64 + lane;
    phase_stamp(a.dbg, 0);
    if constexpr (STREAM) {
        // ------------------------------------------------------------ wide input: K-streamed first layer, full-width accumulators
        constexpr int G0 = 2 * NCH0;
        const int NC = L0.KB / 4;
        const int nb0 = wave * G0;
        f32x4 acc[MB][G0];
        zero_acc<G0, MB>(acc);
        WRing<G0, D> r0;
        const u32x4* wl0 = L0.Wf + (int64_t)nb0 * L0.KB * 64 + lane;
        wring_prime<G0, D>(r0, wl0, L0.KB * 64, L0.KB);
        stage_load(0);
        stage_write(P);
        bias_to_lds();
        __syncthreads();
        phase_stamp(a.dbg, 1);
        for (int c = 0; c + 1 < NC; ++c) {
            stage_load(c + 1);
            mma_chunk8<G0, D, false>(r0, wl0, L0.KB * 64, c * 4, P + (c & 1) * (BM * FUSED_CHUNK * 2), 8, lane, acc);
            stage_write(P + ((c + 1) & 1) * (BM * FUSED_CHUNK * 2));
            __syncthreads();
        }
        mma_chunk8<G0, D, true>(r0, wl0, L0.KB * 64, (NC - 1) * 4, P + ((NC - 1) & 1) * (BM * FUSED_CHUNK * 2), 8, lane, acc);
        phase_stamp(a.dbg, 2);
        wring_prime<2, D1>(r1, wl1, L1.KB * 64, L1.KB);
        __syncthreads();                               // every wave is done with the chunk buffers: H0 may overwrite them
        epi2<G0, MB>(acc, bl + nb0 * 16, P, L0.NB, nb0, n.H[0], L0.NB, nb0, mbg0, nblk, lane);
        __syncthreads();
        phase_stamp(a.dbg, 3);
        zero_acc<2, MB>(acc1);
        if (L1.KB % D1 == 0) mma_stream<2, MB, D1, XB>(r1, wl1, L1.KB * 64, L1.KB, P, L0.NB, lane, acc1);
        else mma_ring<2, MB, D1, XB>(r1, wl1, L1.KB * 64, 0, L1.KB, L1.KB, P, L0.NB, lane, acc1);
    } else {
        // ------------------------------------------------------------ narrow input: resident, H0 in 256-column pieces
        const int NC = L0.KB / 4;                      // 1 or 2 chunks of 128 columns: X = 128 x (NC * 128), CB = 16
        char* X = P;
        char* Cb = P + 64 * 1024;
        WRing<2, D1> r0;
        auto wl0 = [&](int nc) { return L0.Wf + (int64_t)(nc * 16 + wave * 2) * L0.KB * 64 + lane; };
        wring_prime<2, D1>(r0, wl0(0), L0.KB * 64, L0.KB);
        stage_load(0);
        u32x4 stg1[4];                                 // the second chunk's loads in flight together with the first's
        if (NC > 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) stg1[u] = ld_stream_u4<(HGYM_NT & 1) != 0>(srow[u] + FUSED_CHUNK * 2);
        }
        stage_write(X);
        if (NC > 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<u32x4*>(X + 8 * 512 + loffs[u]) = stg1[u];      // column blocks 8 .. 15
        }
        bias_to_lds();
        __syncthreads();
        phase_stamp(a.dbg, 1);
        zero_acc<2, MB>(acc1);
#pragma unroll
        for (int nc = 0; nc < NCH0; ++nc) {
            f32x4 acc[MB][2];
            zero_acc<2, MB>(acc);
            if (L0.KB % D1 == 0) mma_stream<2, MB, D1, 0>(r0, wl0(nc), L0.KB * 64, L0.KB, X, 16, lane, acc);
            else mma_ring<2, MB, D1, 0>(r0, wl0(nc), L0.KB * 64, 0, L0.KB, L0.KB, X, 16, lane, acc);
            if (nc == 0) phase_stamp(a.dbg, 2);
            // this piece's k-steps of the second layer: a sub-stream of 8 k-steps starting at k-block 8 nc
            wring_prime<2, D1>(r1, wl1 + nc * 8 * 64, L1.KB * 64, 8);
            if (nc > 0) __syncthreads();               // the previous piece's readers are done with Cb
            epi2<2, MB>(acc, bl + (nc * 16 + wave * 2) * 16, Cb, 16, wave * 2, n.H[0], L0.NB, nc * 16 + wave * 2, mbg0, nblk, lane);
            if (nc + 1 < NCH0) wring_prime<2, D1>(r0, wl0(nc + 1), L0.KB * 64, L0.KB);
            __syncthreads();
            mma_stream<2, MB, D1, 0>(r1, wl1 + nc * 8 * 64, L1.KB * 64, 8, Cb, 16, lane, acc1);
        }
        phase_stamp(a.dbg, 3);
    }
    // ---------------------------------------------------------------- common: H1 -> LDS, layer 2, head + loss, dZ chain
    WRing<1, D1> r2;
    const u32x4* wl2 = L2.Wf + (int64_t)wave * L2.KB * 64 + lane;
    wring_prime<1, D1>(r2, wl2, L2.KB * 64, L2.KB);
    // loss inputs of the tile's rows (scattered 48-byte rows + scalars), gathered into LDS under the second and third layer
    // (unconditional loads, conditional stores: a conditionally initialised vector ends up in private memory -- the critic's
    // lanes load the same three quads and drop them)
    float lq[3][4];
#pragma unroll
    for (int u = 0; u < 3; ++u) {                      // item j = tid + 512 u < 1152: row j / 9, quad j % 9 (actions 0-2, old mu 3-5, old sigma 6-8)
        int j = tid + u * 512;
        j = j < BM * 9 ? j : BM * 9 - 1;
        const int rw = j / 9, k = j - rw * 9;
        const int64_t ri = rowidx[rw];
        const float* base = k < 3 ? L.actions : (k < 6 ? L.old_mu : L.old_sigma);
        const F4 t = *reinterpret_cast<const F4*>(base + ri * 12 + 4 * (k < 3 ? k : (k < 6 ? k - 3 : k - 6)));
#pragma unroll
        for (int e = 0; e < 4; ++e) lq[u][e] = t.v[e];
    }
    const float* lsrc = is_actor ? (tid < BM ? L.advantages : L.logp) : (tid < BM ? L.returns : L.values);
    const float ls = lsrc[rowidx[tid & (BM - 1)]];
    __syncthreads();                                   // every wave is done reading H0 (P / Cb): H1 may overwrite it
    epi2<2, MB>(acc1, bl + L0.N + wave * 32, RA, 16, wave * 2, n.H[1], 16, wave * 2, mbg0, nblk, lane);
    if (is_actor) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int j = tid + u * 512;
            if (j < BM * 9) {
                const int rw = j / 9, k = j - rw * 9;
                const F4 t = {{lq[u][0], lq[u][1], lq[u][2], lq[u][3]}};
                *reinterpret_cast<F4*>(lin + rw * FB2_LIN_ACTOR + 4 * k) = t;
            }
        }
        if (tid < 2 * BM) lin[(tid & (BM - 1)) * FB2_LIN_ACTOR + (tid < BM ? 36 : 37)] = ls;
    } else if (tid < 2 * BM) {
        lin[(tid & (BM - 1)) * 2 + (tid < BM ? 0 : 1)] = ls;
    }
    __syncthreads();
    phase_stamp(a.dbg, 4);
    f32x4 acc2[MB][1];
    zero_acc<1, MB>(acc2);
    if (L2.KB % D1 == 0) mma_stream<1, MB, D1, XB>(r2, wl2, L2.KB * 64, L2.KB, RA, 16, lane, acc2);
    else mma_ring<1, MB, D1, XB>(r2, wl2, L2.KB * 64, 0, L2.KB, L2.KB, RA, 16, lane, acc2);
    WRing<1, 4> r3;                                    // head: row block `wave`, L3.KB k-steps of one fragment
    wring_prime<1, 4>(r3, L3.Wf + lane, 0, L3.KB);
    epi2<1, MB>(acc2, bl + L0.N + L1.N + wave * 16, RB, 8, wave, n.H[2], 8, wave, mbg0, nblk, lane);
    __syncthreads();
    phase_stamp(a.dbg, 5);
    // ---- head: one wave per row block; loss gradient on its lanes: lane (r, q) = row r of the block, outputs 4q .. 4q + 3
    {
        f32x4 hacc[1][1];
        hacc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int CB3 = L2.NB;
        if (L3.KB % 4 == 0) mma_stream<1, 1, 4>(r3, L3.Wf + lane, 0, L3.KB, RB + wave * CB3 * 512, CB3, lane, hacc);
        else mma_ring<1, 1, 4>(r3, L3.Wf + lane, 0, 0, L3.KB, L3.KB, RB + wave * CB3 * 512, CB3, lane, hacc);
        const int No = L3.N;
        const int m = m0 + wave * 16 + r;
        float out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = hacc[0][0][e] + ((4 * q + e < No) ? bl[L0.N + L1.N + L2.N + 4 * q + e] : 0.0f);
        const bool valid = m < a.M && wave < nblk;
        if (valid && n.out) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < No) n.out[(int64_t)m * n.ldo + 4 * q + e] = out[e];
        }
        // ppo.py:128-168 forward scalars + the hand-written backward of the loss w.r.t. mu, std and V (oracle/ppo_oracle.py:
        // ppo_loss_and_grads) -- mlp_fb_kernel's `head`, same arithmetic in the same order
        const int rl = wave * 16 + r;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        float part[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) part[k] = 0.0f;
        if (is_actor) {
            float act[4] = {0.f, 0.f, 0.f, 0.f}, mo[4] = {0.f, 0.f, 0.f, 0.f}, so[4] = {1.f, 1.f, 1.f, 1.f}, sg[4] = {1.f, 1.f, 1.f, 1.f};
            if (q < 3) {
                const F4 qa = *reinterpret_cast<const F4*>(lin + rl * FB2_LIN_ACTOR + 4 * q);
                const F4 qo = *reinterpret_cast<const F4*>(lin + rl * FB2_LIN_ACTOR + 12 + 4 * q);
                const F4 qs = *reinterpret_cast<const F4*>(lin + rl * FB2_LIN_ACTOR + 24 + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) { act[e] = qa.v[e]; mo[e] = qo.v[e]; so[e] = qs.v[e]; }
            }
            const float adv = lin[rl * FB2_LIN_ACTOR + 36], lpold = lin[rl * FB2_LIN_ACTOR + 37];
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
        // dZ3 tile, block layout: row block `wave`, column block 0 holds this lane's 4 columns (block 1 is zero padding)
        const u32x2 pk = pack_bf16x4(g[0], g[1], g[2], g[3]);
        const u32x2 zero = {0u, 0u};
        const int off0 = (wave * 2 + 0) * 512 + r * 32 + q * 8, off1 = (wave * 2 + 1) * 512 + r * 32 + q * 8;
        *reinterpret_cast<u32x2*>(R0 + off0) = pk;
        *reinterpret_cast<u32x2*>(R0 + off1) = zero;
        if (wave < nblk) {
            char* gz = reinterpret_cast<char*>(n.dZ[3]) + mbg0 * 2 * 512;
            *reinterpret_cast<u32x2*>(gz + off0) = pk;
            *reinterpret_cast<u32x2*>(gz + off1) = zero;
        }
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            float v = part[k];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            part[k] = v;
        }
        // per-ROW-BLOCK partial sums straight to HBM (row mbg0 + wave of `partials`): ppo_scalars_block adds four of them in the
        // order mlp_fb_kernel adds its four head waves (ScalArgs::group = 4), so the loss scalars and the std / head-bias gradients
        // are bit-identical to the 64-row kernel's whatever the tile boundaries are
        if (r == 0 && wave < nblk) {
            float* w = L.partials + (mbg0 + wave) * 32;
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
    // ---- dZ chain on the resident tile
    const int N0 = L0.N, N1 = L1.N, N2 = L2.N;
    const int NBB3 = L3.NBB;
    WRing<1, D1> ra;
    wring_prime<1, D1>(ra, L3.WTf + (int64_t)wave * NBB3 * 64 + lane, NBB3 * 64, NBB3);
    __syncthreads();          // the dZ3 tile is in LDS
    phase_stamp(a.dbg, 6);
    WRing<2, D1> rb;
    bwd2_strip<1, MB, D1, false>(ra, L3.WTf, N2 / 16, NBB3, wave, R0, 2 * NBB3, RB, RB, nullptr, n.dZ[2], mbg0, nblk, lane);
    wring_prime<2, D1>(rb, L2.WTf + (int64_t)(wave * 2) * L2.NBB * 64 + lane, L2.NBB * 64, L2.NBB);
    __syncthreads();
    bwd2_strip<2, MB, D1, false>(rb, L2.WTf, N1 / 16, L2.NBB, wave * 2, RB, N2 / 16, RA, RA, nullptr, n.dZ[1], mbg0, nblk, lane);
    wring_prime<2, D1>(rb, L1.WTf + (int64_t)(wave * 2) * L1.NBB * 64 + lane, L1.NBB * 64, L1.NBB);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NCH0; ++s) {
        const int nb0 = s * 16 + wave * 2;
        bwd2_strip<2, MB, D1, true>(rb, L1.WTf, N0 / 16, L1.NBB, nb0, RA, N1 / 16, nullptr, nullptr, n.H[0], n.dZ[0], mbg0, nblk, lane);
        if (s + 1 < NCH0) wring_prime<2, D1>(rb, L1.WTf + (int64_t)(nb0 + 16) * L1.NBB * 64 + lane, L1.NBB * 64, L1.NBB);
    }
    phase_stamp(a.dbg, 7);
}

// grid (T tiles, nets); 512 threads; one workgroup per CU (153.5 KB of LDS).  Nets: 0 actor, 1 critic.
// ONE instantiation per (actor shape, critic shape) pair, and only XBot-L's is built (hgym_fb2_api.hpp): a first version dispatched over
// all five bodies inside one kernel -- 281 KB of code inside hgym_net's code object -- and its mere PRESENCE (never launched) made
// multi-process runs on one GPU abort at random (hgym_fb2.hip has the story: the limit is the size of a device code object).
template <int NCH_A, bool STREAM_A, int NCH_C, bool STREAM_C>
__global__ __launch_bounds__(FB2_NW * 64) void mlp_fb2_kernel(const FwdArgs a, const FbLoss L, const Fb2Sched sch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int which = a.net0 + blockIdx.y;
    const FusedNet& n = a.net[which];
    if (which == 0) fb2_body<NCH_A, STREAM_A>(a, L, sch, n, true, smem);
    else fb2_body<NCH_C, STREAM_C>(a, L, sch, n, false, smem);
}

}  // namespace hgym
