// hgym_fused.hpp -- the bf16 fast path of the actor / critic (/ auxiliary head): three kernels that replace ~70 launches per minibatch.
//
//   mlp_fwd_kernel<BM>   gather + fp32->bf16 + Linear/ELU x3 + head (+ Gaussian sample / log-prob) for a tile of BM rows;
//                        activations never leave the CU between layers (LDS), weights stream from L2 straight into
//                        MFMA operand registers.  Used by the rollout (PPO.act, BM = 32; hgym_rollout.hip puts the env step of
//                        the tile's 32 envs behind it in the same launch) and for inference.
//   mlp_fb_kernel        the update: the same forward on 64-row tiles, the loss gradient on the head wavefronts (PPO surrogate /
//                        value loss / the auxiliary head's MSE) and dZ_l = (dZ_{l+1} * W_{l+1}) .* elu'(H_l) for l = 2, 1, 0 on the
//                        activations still resident in LDS; one grid row per net.
//   dw_kernel_rs         all weight-gradient products dW_l = dZ_l^T * X_l (+ bias gradients) in ONE launch:
//                        contraction over the batch, operands staged registers -> LDS and read with the gfx950
//                        transpose read (ds_read_b64_tr_b16), so no transposed copy of anything exists in HBM.
//
// Activation layout ("block layout", global and LDS alike): a (rows x cols) bf16 matrix is stored as 16x16 blocks,
// block (mb, cb) at ((mb * CB + cb) * 512) bytes, row-major inside the block (32 B per row).  With it
//   * an MFMA 16x16x32 operand fragment (lane l: row l&15, k-chunk l>>4) is one ds_read_b128 per lane, bank-conflict
//     free (rows 32 B apart, chunk pairs 512 B apart);
//   * an MFMA epilogue (lane l: row l&15, 4 consecutive columns 4*(l>>4)..) writes one 512-byte block per wave
//     instruction, to LDS and to HBM, fully coalesced;
//   * the transpose read of the weight-gradient kernel is lane-linear (address = block + 8 * lane).
// Weight operand layout: "fragment-major", fragment (nb, kb) = 1 KiB in exact lane order (lane l: row nb*16 + (l&15),
// k = kb*32 + 8*(l>>4) .. +7), so a wavefront's weight load is one contiguous 1 KiB global_load_dwordx4.
#pragma once
#include "hgym_finalize.hpp"
// This header's arithmetic is compiled with the same contraction setting in every translation unit (hgym_net.hip: the default;
// hgym_rollout.hip: built with -ffp-contract=off for the env arithmetic it also contains, and defines HGYM_TU_CONTRACT_OFF).
#pragma clang fp contract(fast)
#include "hgym_gemm.hpp"

namespace hgym {

constexpr int FUSED_CHUNK = 128;      // input columns staged per first-layer chunk (4 k-blocks of 32)
// Workgroup shapes: the update uses 64-row tiles with 16 wavefronts (4 per SIMD: the weight stream comes from L2 with
// ~1 us latency under load and only thread-level parallelism hides it); the rollout uses 32-row tiles with 8 wavefronts,
// two workgroups per CU.

struct __attribute__((packed, aligned(4))) F4 {
    float v[4];
};
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

struct FusedLayer {
    const u32x4* Wf;    // forward fragments  [NB][KB][64] x 16 B
    const u32x4* WTf;   // backward fragments [K/16][NBB][64] x 16 B (this layer's W, transposed role)
    const float* bias;  // [N] fp32, 4-byte aligned
    int K, N;           // logical in / out features
    int KB;             // forward contraction blocks of 32
    int NB;             // forward output blocks of 16
    int NBB;            // backward contraction blocks of 32 (N padded to 32)
};

struct FusedNet {
    FusedLayer layer[4];
    const float* x;     // fp32 input rows (observations), leading dimension ldx
    int64_t ldx;
    // bf16 shadow of the input rows: row-major, leading dimension a multiple of 8 and >= 32 * layer[0].KB, pad columns zero.
    //   xb != null (XB16 instantiations): the first layer gathers ITS INPUT from here -- 2 B per element, no conversion, and
    //              the weight-gradient kernel gathers the same rows again by index, so no X0 copy is written at all;
    //   xs != null (fp32-input instantiations, no row gather): the tile also stores the bf16 it has just formed for LDS as row
    //              m of this array -- how the rollout's policy launches leave the shadow of every storage slot behind.
    const __bf16* xb;
    int64_t ldxb;
    __bf16* xs;
    int64_t ldxs;
    __bf16* X0;         // bf16 copy of the gathered input, block layout, CB = 2 * layer[0].KB   (train only)
    __bf16* H[3];       // hidden activations, block layout, CB = layer[l].N / 16                 (train only)
    __bf16* dZ[4];      // pre-activation gradients; dZ[3] (32 * layer[3].NBB columns) is written by the loss kernel
    float* out;         // fp32 head output (M, layer[3].N) row-major
    int64_t ldo;
};

HG_HD int fused_lds_p(const FusedNet& n, int BM) { return BM * (n.layer[0].N > n.layer[2].N ? n.layer[0].N : n.layer[2].N) * 2; }
// fp32 copies of the four bias vectors (fetched at kernel entry so no epilogue waits on a cold global load) + 16 pad
HG_HD int fused_lds_bias(const FusedNet& n) { return (n.layer[0].N + n.layer[1].N + n.layer[2].N + 16 * n.layer[3].NB + 16) * 4; }
HG_HD int fused_lds_q(const FusedNet& n, int BM) {
    const int a = 2 * BM * FUSED_CHUNK * 2, b = BM * n.layer[1].N * 2;
    return a > b ? a : b;
}

__device__ __forceinline__ u32x2 pack_bf16x4(float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
    return __builtin_bit_cast(u32x2, v);
}
// Streaming accesses (data touched once per launch) carry the non-temporal hint so that they do not push the weight
// fragments, which every workgroup re-reads, out of the XCD's L2: kFusedNT bit 0 = input gather loads, bit 1 = X0 / H stores
// of the forward, bit 2 = H loads / dZ stores of the dZ chain.  Same-box A/B per minibatch (B = 61 440, storage 245 760 rows):
// forward 265.6 -> 257.6..262.8 us (bit 0 alone 255), dZ chain 134.4 -> 127.5 us, dW (reads what those wrote) 185 -> 181 us.
constexpr int kFusedNT = 7;
typedef f32x4 f32x4_u4 __attribute__((aligned(4)));
template <bool NT>
__device__ __forceinline__ F4 ld_stream_f4(const float* p) {
    if (NT) return __builtin_bit_cast(F4, __builtin_nontemporal_load(reinterpret_cast<const f32x4_u4*>(p)));
    return *reinterpret_cast<const F4*>(p);
}
template <bool NT>
__device__ __forceinline__ void st_stream_u2(char* p, u32x2 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p));
    else *reinterpret_cast<u32x2*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ u32x2 ld_stream_u2(const char* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
    return *reinterpret_cast<const u32x2*>(p);
}
template <bool NT>
__device__ __forceinline__ u32x4 ld_stream_u4(const char* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return *reinterpret_cast<const u32x4*>(p);
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned int lo16) { return __builtin_bit_cast(float, lo16 << 16); }

// Epilogue stores, 16 bytes per lane.  An MFMA epilogue lane (r = lane & 15, q = lane >> 4) holds 4 consecutive bf16 columns of row r of a
// 16 x 16 block: 8 bytes at r * 32 + q * 8, so the natural store is one dwordx2 per block -- and the update's tiles turned out to be
// paced by store ISSUE (round 4: without its H / dZ stores mlp_fb2_kernel ran 271 instead of 440 us on a slow-class box; MI355X_MICROARCH.md:
// a dwordx2 store tail moves ~7 B/clk/CU, dwordx4 twice that).  For TWO blocks A and B (same lane offsets), v_permlane16_swap_b32 exchanges
// the odd 16-lane rows of A's register with the even rows of B's: afterwards a lane with q even holds columns 8 (q / 2) .. + 7 of row r of
// block A (its own 8 bytes and its right neighbour's), a lane with q odd the same of block B -- one dwordx4 per lane for the pair, same
// bytes, same addresses, half the store instructions.  HGYM_WIDE_ST=0 restores the dwordx2 form (A/B builds).
template <bool NT>
__device__ __forceinline__ void st_stream_u4(char* p, u32x4 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
    else *reinterpret_cast<u32x4*>(p) = v;
}
// pa / pb: block A / B base + this lane's r * 32 + q * 8; oka / okb: whether block A / B is stored at all (wave-uniform)
template <bool NT, bool WIDE = true>
__device__ __forceinline__ void st_pair(char* pa, char* pb, u32x2 a, u32x2 b, bool oka, bool okb, int q) {
    if constexpr (!WIDE) {
        if (oka) st_stream_u2<NT>(pa, a);
        if (okb) st_stream_u2<NT>(pb, b);
        return;
    }
    const auto s0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
    const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
    const bool odd = q & 1;
    char* p = odd ? pb - 8 : pa;
    if (odd ? okb : oka) st_stream_u4<NT>(p, v);
}

// ---- weight stream ------------------------------------------------------------------------------------------------------
// The weight fragments of one output strip (G n-blocks) stream from L2 into a register ring of D k-steps per wavefront:
// slot t % D holds k-step t, and the slot is re-loaded with k-step t + D as soon as its MFMAs have been issued, so D - 1
// k-steps of loads (D * G KiB per wavefront) are in flight under the MFMAs of one.  L2 latency under load is ~1 us and the
// MFMAs of a k-step take ~0.1 us: a ring primed ONE layer (or one strip) ahead -- before the epilogue and the barrier of the
// previous one -- never starts cold.  D = 4 for the 8-wave rollout tiles (2 waves per SIMD, 256 VGPRs each), D = 2 for the
// 16-wave update tiles (4 waves per SIMD hide the rest).
template <int G, int D>
struct WRing {
    u32x4 w[D][G];
};

// issue the loads of k-steps [0, min(D, nk)) of the stream whose fragment (g, t) is wl[g * wstride + t * 64]  (wl = w0 + lane)
// (G <= GR: a ring declared for the widest strip also serves the narrower ones; the unused slots are never allocated)
template <int G, int D, int GR>
__device__ __forceinline__ void wring_prime(WRing<GR, D>& R, const u32x4* __restrict__ wl, int wstride, int nk) {
    static_assert(G <= GR, "ring too narrow");
    // unconditional (a stream shorter than the ring re-loads its last k-step): the number of loads in flight stays a
    // compile-time constant, which is what lets the compiler's s_waitcnt placement be exact instead of draining the ring
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int t = d < nk ? d : nk - 1;
#pragma unroll
        for (int g = 0; g < G; ++g) R.w[d][g] = wl[(int64_t)g * wstride + t * 64];
    }
}

// acc[i][g] += W(g, t0 + t) x X(i, t) for t < n, on a ring that holds k-steps t0 .. t0 + D - 1 (t0 % D == 0); refills the
// ring up to k-step nk_total.  xl -> LDS block layout (CBx column blocks per row block) of k-block t0.  The x fragments
// are double-buffered one k-step ahead (LDS latency only).
template <int G, int MB, int D, int XBP = -1, int GR>
__device__ __forceinline__ void mma_ring(WRing<GR, D>& R, const u32x4* __restrict__ wl, int wstride, int t0, int n, int nk_total,
                                         const char* xl, int CBx, int lane, f32x4 (&acc)[MB][G]) {
    static_assert(D % 2 == 0, "x double buffer follows the parity of the ring slot");
    // x fragments one k-step ahead only where registers allow (tiles of <= 2 row blocks; the 4-row-block update tiles run
    // 4 waves per SIMD, which covers the LDS latency) unless the caller says otherwise
    constexpr bool XB = XBP < 0 ? (MB <= 2) : (XBP != 0);
    const int r = lane & 15, q = lane >> 4;
    const char* xb = xl + (q >> 1) * 512 + r * 32 + (q & 1) * 16;
    u32x4 xa[MB], xc[XB ? MB : 1];
    if (XB) {
#pragma unroll
        for (int i = 0; i < MB; ++i) xa[i] = *reinterpret_cast<const u32x4*>(xb + (i * CBx) * 512);
    }
    // sched_barrier pins the order "x of the next k-step, MFMAs of this one, refill of this slot": left alone, the scheduler
    // sinks every load to just before its first use (fewer live registers) and the loop serialises on memory.
    for (int tt = 0; tt < n; tt += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int t = tt + d;
            if (t < n) {
                if (XB) {
                    if (t + 1 < n) {
#pragma unroll
                        for (int i = 0; i < MB; ++i) {
                            const u32x4 v = *reinterpret_cast<const u32x4*>(xb + (i * CBx + 2 * (t + 1)) * 512);
                            if (d & 1) xa[i] = v; else xc[i] = v;
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < MB; ++i) xa[i] = *reinterpret_cast<const u32x4*>(xb + (i * CBx + 2 * t) * 512);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int g = 0; g < G; ++g) mma_frag<__bf16>(R.w[d][g], (XB && (d & 1)) ? xc[i] : xa[i], acc[i][g]);
                __builtin_amdgcn_sched_barrier(0);
                if (t0 + t + D < nk_total) {
#pragma unroll
                    for (int g = 0; g < G; ++g) R.w[d][g] = wl[(int64_t)g * wstride + (t0 + t + D) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// The two branch-free forms of the above.  Every k-step of the steady state issues the same loads in the same order, so the
// vmcnt the compiler derives for "slot d is ready" is the exact (D - 1) * G (+ whatever else is in flight); conditions
// around the refills (as in mma_ring) make its path-merging analysis assume the worst and drain the ring every revolution.
//
// mma_stream: a whole stream of nk k-steps (nk % D == 0, nk >= D) whose input is resident in LDS; ring primed with 0..D-1.
template <int G, int MB, int D, int XBP = -1, int GR>
__device__ __forceinline__ void mma_stream(WRing<GR, D>& R, const u32x4* __restrict__ wl, int wstride, int nk, const char* xl, int CBx,
                                           int lane, f32x4 (&acc)[MB][G]) {
    static_assert(D % 2 == 0, "x double buffer follows the parity of the ring slot");
    constexpr bool XB = XBP < 0 ? (MB <= 2) : (XBP != 0);
    const int r = lane & 15, q = lane >> 4;
    const char* xb = xl + (q >> 1) * 512 + r * 32 + (q & 1) * 16;
    u32x4 xa[MB], xc[XB ? MB : 1];
    if (XB) {
#pragma unroll
        for (int i = 0; i < MB; ++i) xa[i] = *reinterpret_cast<const u32x4*>(xb + (i * CBx) * 512);
    }
    int tt = 0;
    for (; tt + D < nk; tt += D) {          // steady state: refill every slot
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int t = tt + d;
            if (XB) {
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(xb + (i * CBx + 2 * (t + 1)) * 512);
                    if (d & 1) xa[i] = v; else xc[i] = v;
                }
            } else {
#pragma unroll
                for (int i = 0; i < MB; ++i) xa[i] = *reinterpret_cast<const u32x4*>(xb + (i * CBx + 2 * t) * 512);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int g = 0; g < G; ++g) mma_frag<__bf16>(R.w[d][g], (XB && (d & 1)) ? xc[i] : xa[i], acc[i][g]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g) R.w[d][g] = wl[(int64_t)g * wstride + (t + D) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {           // last revolution: no refills
        const int t = tt + d;
        if (XB) {
            if (d + 1 < D) {
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(xb + (i * CBx + 2 * (t + 1)) * 512);
                    if (d & 1) xa[i] = v; else xc[i] = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < MB; ++i) xa[i] = *reinterpret_cast<const u32x4*>(xb + (i * CBx + 2 * t) * 512);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int g = 0; g < G; ++g) mma_frag<__bf16>(R.w[d][g], (XB && (d & 1)) ? xc[i] : xa[i], acc[i][g]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// mma_chunk: the 4 k-steps t0 .. t0 + 3 (t0 % 4 == 0, D divides 4) of a stream whose input arrives in 4-k-block LDS chunks
// (first layer).  LAST: this is the final chunk, nothing beyond k-step t0 + 3 exists.
template <int G, int MB, int D, bool LAST, int XBP = -1, int GR>
__device__ __forceinline__ void mma_chunk(WRing<GR, D>& R, const u32x4* __restrict__ wl, int wstride, int t0, const char* xl, int CBx,
                                          int lane, f32x4 (&acc)[MB][G]) {
    static_assert(D == 2 || D == 4, "ring depth must divide the chunk");
    constexpr bool XB = XBP < 0 ? (MB <= 2) : (XBP != 0);
    const int r = lane & 15, q = lane >> 4;
    const char* xb = xl + (q >> 1) * 512 + r * 32 + (q & 1) * 16;
    u32x4 xa[MB], xc[XB ? MB : 1];
    if (XB) {
#pragma unroll
        for (int i = 0; i < MB; ++i) xa[i] = *reinterpret_cast<const u32x4*>(xb + (i * CBx) * 512);
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int d = s4 % D;
        if (XB) {
            if (s4 + 1 < 4) {
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(xb + (i * CBx + 2 * (s4 + 1)) * 512);
                    if (s4 & 1) xa[i] = v; else xc[i] = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < MB; ++i) xa[i] = *reinterpret_cast<const u32x4*>(xb + (i * CBx + 2 * s4) * 512);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int g = 0; g < G; ++g) mma_frag<__bf16>(R.w[d][g], (XB && (s4 & 1)) ? xc[i] : xa[i], acc[i][g]);
        __builtin_amdgcn_sched_barrier(0);
        if (!LAST || s4 + D < 4) {
#pragma unroll
            for (int g = 0; g < G; ++g) R.w[d][g] = wl[(int64_t)g * wstride + (t0 + s4 + D) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ELU of the bf16 path: exp(z) as v_exp_f32(z * log2 e).  The libm expf the fp32 parity path keeps (elu_f) is ~15 VALU
// instructions per element (argument split, ldexp, two range selects) and the epilogues were VALU-bound on them; the result
// is rounded to bf16 (2^-8 relative) right after, against this form's <= 1e-6 relative error for |z| <= 18.
// HGYM_ELU_FAST=0 restores elu_f.
__device__ __forceinline__ float elu_bf(float z) {
    return z > 0.0f ? z : (__builtin_amdgcn_exp2f(z * 1.4426950408889634f) - 1.0f);
}

// bias + ELU, bf16, -> LDS block layout (next layer's input) and, when Hg != null, the same blocks in HBM
template <int G, int MB, bool STORE>
__device__ __forceinline__ void epilogue_elu_t(f32x4 (&acc)[MB][G], const float* __restrict__ bias, int nb0, char* out_lds, int CBo,
                                               __bf16* __restrict__ Hg, int64_t mbg0, int lane) {
    const int r = lane & 15, q = lane >> 4;
    const int loff = r * 32 + q * 8;
    static_assert(MB % 2 == 0, "row blocks are stored in pairs");
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const F4 b = *reinterpret_cast<const F4*>(bias + (nb0 + g) * 16 + 4 * q);
#pragma unroll
        for (int i = 0; i < MB; i += 2) {
            u32x2 pk[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = elu_bf(acc[i + h][g][e] + b.v[e]);
                pk[h] = pack_bf16x4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<u32x2*>(out_lds + ((i + h) * CBo + nb0 + g) * 512 + loff) = pk[h];
            }
            if (STORE) {
                char* pa = reinterpret_cast<char*>(Hg) + ((mbg0 + i) * CBo + nb0 + g) * 512 + loff;
                st_pair<(kFusedNT & 2) != 0, false>(pa, pa + (int64_t)CBo * 512, pk[0], pk[1], true, true, q);
            }
        }
    }
}
// (the store decision is wave-uniform and taken once here, not per block inside the unrolled epilogue)
template <int G, int MB>
__device__ __forceinline__ void epilogue_elu(f32x4 (&acc)[MB][G], const float* __restrict__ bias, int nb0, char* out_lds, int CBo,
                                             __bf16* __restrict__ Hg, int64_t mbg0, int lane) {
    if (Hg) epilogue_elu_t<G, MB, true>(acc, bias, nb0, out_lds, CBo, Hg, mbg0, lane);
    else epilogue_elu_t<G, MB, false>(acc, bias, nb0, out_lds, CBo, Hg, mbg0, lane);
}

template <int G, int MB>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[MB][G]) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[i][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// one hidden layer whose whole input is resident in LDS: waves split the n-blocks, G per strip.  R holds the primed stream
// of this wave's first strip (hidden_prime).  With AHEAD (8-wave tiles: registers to spare) `prime_next` is called before the
// epilogue of the last strip, so the following layer's stream is in flight across the epilogue and the barrier; without it
// the caller primes after the barrier.
template <int G, int D, int GR>
__device__ __forceinline__ void hidden_prime(WRing<GR, D>& R, const FusedLayer& L, int wave, int lane) {
    const int nb0 = wave * G;
    if (nb0 < L.NB) wring_prime<G, D>(R, L.Wf + (int64_t)nb0 * L.KB * 64 + lane, L.KB * 64, L.KB);
}

template <int G, int MB, int NW, int D, bool AHEAD, int GR, class Next>
__device__ __forceinline__ void hidden_layer(WRing<GR, D>& R, const FusedLayer& L, const float* bias, const char* in_lds, int CBin,
                                             char* out_lds, __bf16* Hg, int64_t mbg0, int wave, int lane, Next prime_next) {
    bool primed = false;
    for (int nb0 = wave * G; nb0 < L.NB; nb0 += NW * G) {
        f32x4 acc[MB][G];
        zero_acc<G, MB>(acc);
        const u32x4* wl = L.Wf + (int64_t)nb0 * L.KB * 64 + lane;
        constexpr int XBF = NW <= 8 ? 1 : -1;
        if (L.KB % D == 0) mma_stream<G, MB, D, XBF>(R, wl, L.KB * 64, L.KB, in_lds, CBin, lane, acc);
        else mma_ring<G, MB, D, XBF>(R, wl, L.KB * 64, 0, L.KB, L.KB, in_lds, CBin, lane, acc);
        const int nxt = nb0 + NW * G;
        if (nxt < L.NB) wring_prime<G, D>(R, L.Wf + (int64_t)nxt * L.KB * 64 + lane, L.KB * 64, L.KB);
        else if (AHEAD) { prime_next(); primed = true; }
        epilogue_elu<G, MB>(acc, bias, nb0, out_lds, L.NB, Hg, mbg0, lane);
    }
    if (AHEAD && !primed) prime_next();       // waves without a strip in this layer
}

// optional phase timestamps (hgym_prof_phase_buffer): thread 0 of every workgroup writes the 100 MHz wall clock at phase
// boundaries into dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + slot]
__device__ __forceinline__ void phase_stamp(long long* dbg, int slot) {
    if (dbg && threadIdx.x == 0) dbg[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + slot] = (long long)__builtin_amdgcn_s_memrealtime();
}

// log(sigma) of the sampling epilogue: hardware log2 times ln 2 as ONE non-fusable multiply.  libm's logf is inlined library
// code whose internal multiply-adds the backend contracts or not depending on the translation unit's -ffp-contract setting, so
// it returned different last bits in hgym_net.hip and hgym_rollout.hip; this form is the same instruction pair everywhere
// (<= 2 ulp of log for sigma in [1e-3, 1e3]; the log-probability is compared with the reference at 1e-5 relative).
__device__ __forceinline__ float log_sigma(float s) { return __fmul_rn(__builtin_amdgcn_logf(s), 0.6931471805599453f); }

// ---- the actor's first layer carried ACROSS launches of the rollout (round 4) ----------------------------------------------------
// An observation row is a stack of frames: the row of step t + 1 is the row of step t shifted by one frame (47 columns) plus the
// frame step t produces, so columns [0, 640) of the NEXT row -- 20 of the first layer's 24 k-steps -- are known when launch t starts.
// Launch t's critic workgroup of the tile (idle from ~16 us of the ~40 us launch on) forms those 20 k-steps of the actor's first
// layer for the next row (`l0_partial_ahead`) and leaves the fp32 accumulators in the caller's scratch; launch t + 1's actor tile
// (`fwd_body<.., PART>`) starts from them and runs the last 128-column chunk only: the same fragments, the same k order, the same
// accumulator chain -- the pre-activations are bit-identical -- and 8 of the first layer's 9.5 us leave the launch's critical path.
// Rows whose env was reset in between have zero older frames: their partial sums are dropped (0 + the chunk's products, as the
// reference computes on a zero history).
struct L0Part {
    const float* acc;        // [tile][wave][MB * G1][64 lanes][4]: pre-activations over k-steps [0, kb0) left by the previous launch
    const uint8_t* reset;    // (M,) the previous step's reset flags
    int kb0;                 // first k-step still to do here (a multiple of 4: whole 128-column chunks)
};
struct L0Ahead {
    float* acc_out;          // the same layout, for the NEXT launch (null: nothing to do)
    __bf16* xs_next;         // bf16 shadow rows of the next step's observation: columns [0, 32 kb0) are written here (null: no shadow)
    int64_t ldxs;
    int shift;               // columns a row shifts by per step (one frame)
    int kb0;                 // k-steps formed ahead
};

// by ALL 8 wavefronts of a 32-row workgroup whose own tile is finished (the caller has synchronised): rows m0 .. m0 + 31 of net n's fp32
// input, columns [shift, shift + 32 kb0) -> bf16 (the rounding fwd_body's staging applies) -> LDS block layout -> k-steps [0, kb0) of
// the first layer with fwd_body<32, 8, 4, G1>'s wave -> strip mapping -> acc_out.  kb0 % 4 == 0.
template <int G1>
__device__ __forceinline__ void l0_partial_ahead(const FusedNet& n, const L0Ahead& ah, int M, char* smem) {
    constexpr int BM = 32, NW = 8, MB = 2, D = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM;
    const FusedLayer& L0 = n.layer[0];
    const int CB = 2 * ah.kb0, groups = 8 * ah.kb0;          // column blocks of 16 / groups of 4 columns per row
    const int nb0 = wave * G1;
    WRing<G1, D> r0;
    const u32x4* wl0 = L0.Wf + (int64_t)nb0 * L0.KB * 64 + lane;
    wring_prime<G1, D>(r0, wl0, L0.KB * 64, ah.kb0);
    constexpr int IPR = 5;                                       // items per lane per round: (row, 4 consecutive new columns)
    for (int j0 = 0; j0 < BM * groups; j0 += NW * 64 * IPR) {
        F4 v[IPR];
        int row[IPR], c4[IPR];
#pragma unroll
        for (int u = 0; u < IPR; ++u) {
            int j = j0 + u * NW * 64 + tid;
            j = j < BM * groups ? j : BM * groups - 1;
            row[u] = j / groups;
            c4[u] = (j - row[u] * groups) * 4;
            int m = m0 + row[u];
            m = m < M ? m : M - 1;
            v[u] = *reinterpret_cast<const F4*>(n.x + (int64_t)m * n.ldx + ah.shift + c4[u]);
        }
#pragma unroll
        for (int u = 0; u < IPR; ++u) {
            const u32x2 pk = pack_bf16x4(v[u].v[0], v[u].v[1], v[u].v[2], v[u].v[3]);
            *reinterpret_cast<u32x2*>(smem + ((row[u] >> 4) * CB + (c4[u] >> 4)) * 512 + (row[u] & 15) * 32 + ((c4[u] >> 2) & 3) * 8) = pk;
            if (ah.xs_next && m0 + row[u] < M && j0 + u * NW * 64 + tid < BM * groups)
                st_stream_u2<(kFusedNT & 2) != 0>(reinterpret_cast<char*>(ah.xs_next + (int64_t)(m0 + row[u]) * ah.ldxs + c4[u]), pk);
        }
    }
    __syncthreads();
    f32x4 acc[MB][G1];
    zero_acc<G1, MB>(acc);
    mma_stream<G1, MB, D, 1>(r0, wl0, L0.KB * 64, ah.kb0, smem, CB, lane, acc);
    f32x4* dst = reinterpret_cast<f32x4*>(ah.acc_out) + ((int64_t)(blockIdx.x * NW + wave) * (MB * G1)) * 64 + lane;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int g = 0; g < G1; ++g) dst[(i * G1 + g) * 64] = acc[i][g];
}

struct FwdArgs {
    long long* dbg;
    FusedNet net[3];           // 0 actor, 1 critic, 2 auxiliary head
    int net0;                 // net index of blockIdx.y == 0
    int M;
    const int64_t* idx;       // optional row gather
    int train;                // write X0 / H[] for the backward pass
    int sample;               // actor head epilogue: Gaussian sample + log-prob (PPO.act)
    int A;                    // num_actions
    const float* std_;        // [A]
    const float* z;           // optional (M, A) normals; null -> Philox(k0, k1, *step)
    uint32_t k0, k1;
    const int64_t* step;
    float* actions;           // (M, A)
    float* sigma;             // (M, A)
    float* logp;              // (M,)
    int nets;                 // networks in this launch (blockIdx.y < nets)
    FinArgs fin;              // postponed env-step finaliser riding in this launch (blockIdx.y == nets, one workgroup); fin.N == 0: none
};

// Hooks of the fused rollout step (hgym_rollout.hip), three callables: `early()` runs once per workgroup right after the tile's
// first loads (bias, first weight k-steps, first input chunk) have been ISSUED and before anything waits for them -- the env
// step's own input loads and its Philox draws go there, under the same memory round trip; `mid()` runs between the first
// layer and the second, where the first layer's weight ring and accumulators are dead and ~60 registers are free for loads that
// may take the rest of the tile to arrive (the observation history); `put_action(row, j, a)` receives every sampled action of
// the tile (row within the tile, action index) so that the env phase reads them from LDS; `idle()` runs on the wavefronts that
// have no head block while the others compute the head; `head(wave, row, nb, out[4])` runs on the head wavefronts with each lane's
// row, the head's 16-column block and its four head outputs (the fused forward + backward kernel computes the PPO loss gradient there).  They are lambdas that capture the
// caller's locals by reference (a hook OBJECT carrying the prefetch arrays as members was kept in private memory by the
// compiler: 500 scratch instructions and a kernel four times slower).
// `extra` (whatever the hooks need from the kernel argument) reaches them as a call PARAMETER: a closure that captured a
// reference to the kernel argument would count as a capture of the whole argument block, of which the compiler then keeps a
// private-memory copy (3 KB of scratch per lane).
struct FwdNoop {
    template <class X> __device__ __forceinline__ void operator()(const X&) const {}
    __device__ __forceinline__ void operator()(int, int, float) const {}
    __device__ __forceinline__ void operator()(int, int, int, const float (&)[4]) const {}
};

template <int BM, int NW, int D, int G1, bool WIDE = false, bool XB16 = false, bool PART = false, class Early = FwdNoop, class Mid = FwdNoop,
          class Put = FwdNoop, class Extra = int, class Idle = FwdNoop, class Head = FwdNoop, class L2Idle = FwdNoop>
__device__ __forceinline__ void fwd_body(const FwdArgs& a, const FusedNet& n, bool is_actor, char* smem, Early&& hook_early = Early(),
                                         Mid&& hook_mid = Mid(), Put&& hook_put = Put(), const Extra& extra = Extra(),
                                         Idle&& hook_idle = Idle(), Head&& hook_head = Head(), char* h2_lds = nullptr,
                                         int* rowidx_lds = nullptr, L2Idle&& hook_l2idle = L2Idle(), const L0Part* part = nullptr) {
    // PART (the rollout's actor tile, hgym_rollout.hip): the first layer starts from the partial sums *part and runs chunks
    // [part->kb0 / 4, NC) only (struct L0Part above)
    static_assert(!PART || !XB16, "the carried first layer exists for fp32 input rows only");
    // rowidx_lds (BM ints of LDS): receives the storage row of every tile row (a.idx gathered once, by the lanes that stage the
    // input) for whoever needs it later in the tile; hook_l2idle(extra): runs on the wavefronts that have no strip in the third
    // layer while the others compute it (the fused forward + backward kernel gathers its loss inputs there).
    constexpr int MB = BM / 16;
    constexpr int IT = BM * 32 / (NW * 64);           // staging items per thread per chunk (BM rows x 32 float4)
    constexpr int RPP = NW * 2;                       // rows covered per staging pass (32 lanes per row)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int64_t mbg0 = m0 >> 4;
    char* P = smem;
    char* Q = smem + fused_lds_p(n, BM);
    // third-layer activations: over the first layer's (dead) in P, unless the caller wants H0 to survive (the fused forward +
    // backward kernel needs all three for its dZ chain) and provides a buffer of BM x layer[2].N bf16
    char* PH2 = h2_lds ? h2_lds : P;
    const FusedLayer& L0 = n.layer[0];
    const FusedLayer& L1 = n.layer[1];
    const FusedLayer& L2 = n.layer[2];
    const FusedLayer& L3 = n.layer[3];
    const int train = (a.train & 1) && n.X0, train_h = a.train & 2;     // bit 0: write X0 (a net without an X0 buffer shares another net's copy of the same rows), bit 1: write H[]
    // the four bias vectors -> LDS: the loads are issued here, ahead of everything else, and parked in registers; they are
    // written to LDS next to the first input chunk, so no epilogue ever waits on a cold global load
    float* bl = reinterpret_cast<float*>(smem + fused_lds_p(n, BM) + fused_lds_q(n, BM));
    constexpr int BIT = (768 + 768 + 768 + 96 + NW * 64 - 1) / (NW * 64);     // fused_supported: hidden widths <= 768, head <= 96
    float bv[BIT];
    const int bn0 = L0.N, bn1 = bn0 + L1.N, bn2 = bn1 + L2.N, bn3 = bn2 + (WIDE ? 16 * L3.NB : 16);
#pragma unroll
    for (int u = 0; u < BIT; ++u) {
        int i = tid + u * NW * 64;
        i = i < bn2 + L3.N ? i : bn2 + L3.N - 1;          // clamp: the pad entries re-read the last head bias (never used)
        const float* src = i < bn0 ? L0.bias + i : (i < bn1 ? L1.bias + (i - bn0) : (i < bn2 ? L2.bias + (i - bn1) : L3.bias + (i - bn2)));
        bv[u] = *src;
    }
    // sampling epilogue inputs (head waves of the actor): fetched now, used ~20 us later
    constexpr bool HOIST = NW <= 8;      // the 16-wave tiles have no registers to park them in (they spill)
    int64_t step_pre = 0;
    float std_pre[4] = {1.0f, 1.0f, 1.0f, 1.0f}, lsg_pre[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (HOIST && is_actor && a.sample && wave < BM / 16) {
        if (!a.z && a.step) step_pre = a.step[0];
#pragma unroll
        for (int e = 0; e < 4; ++e) std_pre[e] = a.std_[4 * q + e < a.A ? 4 * q + e : 0];
#pragma unroll
        for (int e = 0; e < 4; ++e) lsg_pre[e] = log_sigma(std_pre[e]);      // under the first input load instead of in the tail
    }
    auto bias_to_lds = [&]() {
#pragma unroll
        for (int u = 0; u < BIT; ++u) {
            const int i = tid + u * NW * 64;
            if (i < bn3) bl[i] = bv[u];
        }
    };
    constexpr bool AHEAD = NW <= 8;
    constexpr int XBF = NW <= 8 ? 1 : -1;  // 8-wave tiles have the registers to keep the x fragments one k-step ahead
    constexpr int GH = NW <= 8 ? 2 : 1;   // n-blocks per strip in the hidden layers (widths are multiples of 128 = 8 n-blocks)
    WRing<GH, D> r1, r2;             // weight streams of the two hidden layers
    auto prime1 = [&]() { hidden_prime<GH, D>(r1, L1, wave, lane); };

    // ---------------------------------------------------------------- layer 0: input streamed in 128-column chunks
    if constexpr (XB16) {
        // input rows from the bf16 shadow: a chunk is 64 (32) rows x 256 B = one 16-byte item per lane, loaded as it will lie in
        // LDS.  Lane map: 16 consecutive lanes = 8 rows x the two halves of one block row (a conflict-free 256-byte LDS
        // write), the four lane groups = four consecutive column blocks (128 contiguous bytes of each row per instruction);
        // waves = row groups of 8 x the two halves of the chunk.
        constexpr int RG = BM / 8;
        static_assert(NW == 2 * RG, "one 16-byte item per lane per chunk");
        const int NC = L0.KB / 4;
        const int row = (wave % RG) * 8 + ((lane >> 1) & 7), cb = (wave / RG) * 4 + (lane >> 4), hf = lane & 1;
        int m = m0 + row;
        m = m < a.M ? m : a.M - 1;
        const int64_t src = a.idx ? a.idx[m] : (int64_t)m;
        const char* srow = reinterpret_cast<const char*>(n.xb + src * n.ldxb + cb * 16 + hf * 8);
        if (rowidx_lds && cb == 0 && hf == 0) rowidx_lds[row] = (int)src;
        const int loff = ((row >> 4) * 8 + cb) * 512 + (row & 15) * 32 + hf * 16;
        u32x4 stg;
        auto stage_load = [&](int c) { stg = ld_stream_u4<(kFusedNT & 1) != 0>(srow + c * (FUSED_CHUNK * 2)); };
        auto stage_write = [&](int buf) { *reinterpret_cast<u32x4*>(Q + buf * (BM * FUSED_CHUNK * 2) + loff) = stg; };
        const int nb0 = wave * G1;
        f32x4 acc[MB][G1];
        zero_acc<G1, MB>(acc);
        WRing<G1, D> r0;
        const u32x4* wl0 = L0.Wf + (int64_t)nb0 * L0.KB * 64 + lane;
        phase_stamp(a.dbg, 0);
        wring_prime<G1, D>(r0, wl0, L0.KB * 64, L0.KB);
        stage_load(0);
        hook_early(extra);
        stage_write(0);
        bias_to_lds();
        __syncthreads();
        phase_stamp(a.dbg, 1);
        for (int c = 0; c + 1 < NC; ++c) {
            stage_load(c + 1);
            mma_chunk<G1, MB, D, false, XBF>(r0, wl0, L0.KB * 64, c * 4, Q + (c & 1) * (BM * FUSED_CHUNK * 2), 8, lane, acc);
            stage_write((c + 1) & 1);
            __syncthreads();
        }
        mma_chunk<G1, MB, D, true, XBF>(r0, wl0, L0.KB * 64, (NC - 1) * 4, Q + ((NC - 1) & 1) * (BM * FUSED_CHUNK * 2), 8, lane, acc);
        phase_stamp(a.dbg, 2);
        if (AHEAD) prime1();
        epilogue_elu<G1, MB>(acc, bl, nb0, P, L0.NB, train_h ? n.H[0] : nullptr, mbg0, lane);
    } else {
        const int NC = L0.KB / 4;
        const int CB0 = 2 * L0.KB;
        const int f4 = tid & 31;
        const float* srow[IT];
        int lrow[IT];
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const int row = u * RPP + (tid >> 5);
            int m = m0 + row;
            m = m < a.M ? m : a.M - 1;
            const int64_t src = a.idx ? a.idx[m] : (int64_t)m;
            srow[u] = n.x + src * n.ldx;
            lrow[u] = row;
            if (rowidx_lds && f4 == 0) rowidx_lds[row] = (int)src;
        }
        F4 stg[IT];
        // one unconditional 16-byte load per item: the address is clamped to the last full vector of the row.  stage_load
        // ONLY issues the loads (nothing here may consume them: the data must stay in flight under the MFMAs of the current
        // chunk); stage_write re-aligns / zero-fills the (at most one per row) straddling item with selects, converts and
        // stores to LDS.
        auto stage_load = [&](int c) {
            const int col = c * FUSED_CHUNK + f4 * 4;
            const int cc = col < L0.K - 4 ? col : L0.K - 4;
#pragma unroll
            for (int u = 0; u < IT; ++u) stg[u] = ld_stream_f4<(kFusedNT & 1) != 0>(srow[u] + cc);
        };
        auto stage_write = [&](int c, int buf) {
            char* dst = Q + buf * (BM * FUSED_CHUNK * 2);
            const int col = c * FUSED_CHUNK + f4 * 4;
            const int cc = col < L0.K - 4 ? col : L0.K - 4;
            const int sh = col - cc;                       // 0 for full items, 1..3 for the straddling one, >= 4: all padding
#pragma unroll
            for (int u = 0; u < IT; ++u) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = e + sh;
                    float r = 0.0f;
                    r = k == 0 ? stg[u].v[0] : r;
                    r = k == 1 ? stg[u].v[1] : r;
                    r = k == 2 ? stg[u].v[2] : r;
                    r = k == 3 ? stg[u].v[3] : r;
                    v[e] = r;
                }
                const u32x2 pk = pack_bf16x4(v[0], v[1], v[2], v[3]);
                const int inblk = (lrow[u] & 15) * 32 + (f4 & 3) * 8;
                *reinterpret_cast<u32x2*>(dst + ((lrow[u] >> 4) * 8 + (f4 >> 2)) * 512 + inblk) = pk;
                if (train)
                    st_stream_u2<(kFusedNT & 2) != 0>(reinterpret_cast<char*>(n.X0) +
                                                     ((mbg0 + (lrow[u] >> 4)) * CB0 + c * 8 + (f4 >> 2)) * 512 + inblk, pk);
                if (n.xs && m0 + lrow[u] < a.M)      // row m of the bf16 shadow (pad columns receive the zeros formed above)
                    st_stream_u2<(kFusedNT & 2) != 0>(reinterpret_cast<char*>(n.xs + (int64_t)(m0 + lrow[u]) * n.ldxs + col), pk);
            }
        };
        const int nb0 = wave * G1;
        f32x4 acc[MB][G1];
        const int c0 = PART ? part->kb0 / 4 : 0;                // first chunk this launch runs; k-steps and LDS buffers count from it
        bool drop[MB];
        if constexpr (PART) {
            const f32x4* src = reinterpret_cast<const f32x4*>(part->acc) + ((int64_t)(blockIdx.x * NW + wave) * (MB * G1)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int g = 0; g < G1; ++g) acc[i][g] = src[(i * G1 + g) * 64];
#pragma unroll
            for (int i = 0; i < MB; ++i) {                      // lane (r, q) holds row i * 16 + r of the tile
                int m = m0 + i * 16 + r;
                m = m < a.M ? m : a.M - 1;
                drop[i] = part->reset[m] != 0;
            }
        } else {
            zero_acc<G1, MB>(acc);
        }
        WRing<G1, D> r0;
        const u32x4* wl0 = L0.Wf + (int64_t)nb0 * L0.KB * 64 + lane + (int64_t)c0 * 4 * 64;
        phase_stamp(a.dbg, 0);
        wring_prime<G1, D>(r0, wl0, L0.KB * 64, L0.KB - c0 * 4);
        stage_load(c0);
        hook_early(extra);
        stage_write(c0, 0);
        bias_to_lds();
        if constexpr (PART) {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int g = 0; g < G1; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][g][e] = drop[i] ? 0.0f : acc[i][g][e];
        }
        __syncthreads();
        phase_stamp(a.dbg, 1);
        // steady state: no condition inside the body (see mma_stream); the last chunk is peeled
        for (int c = c0; c + 1 < NC; ++c) {
            stage_load(c + 1);
            mma_chunk<G1, MB, D, false, XBF>(r0, wl0, L0.KB * 64, (c - c0) * 4, Q + ((c - c0) & 1) * (BM * FUSED_CHUNK * 2), 8, lane, acc);
            stage_write(c + 1, (c + 1 - c0) & 1);
            __syncthreads();
        }
        mma_chunk<G1, MB, D, true, XBF>(r0, wl0, L0.KB * 64, (NC - 1 - c0) * 4, Q + ((NC - 1 - c0) & 1) * (BM * FUSED_CHUNK * 2), 8, lane, acc);
        phase_stamp(a.dbg, 2);
        if (AHEAD) prime1();
        epilogue_elu<G1, MB>(acc, bl, nb0, P, L0.NB, train_h ? n.H[0] : nullptr, mbg0, lane);
    }
    __syncthreads();
    phase_stamp(a.dbg, 3);
    hook_mid(extra);
    if (!AHEAD) prime1();
    // ---------------------------------------------------------------- layers 1, 2: input resident in LDS
    auto prime2 = [&]() { hidden_prime<GH, D>(r2, L2, wave, lane); };
    hidden_layer<GH, MB, NW, D, AHEAD>(r1, L1, bl + L0.N, P, L0.NB, Q, train_h ? n.H[1] : nullptr, mbg0, wave, lane, prime2);
    __syncthreads();
    phase_stamp(a.dbg, 4);
    if (!AHEAD) prime2();
    WRing<1, 4> r3;                  // head: one 16-row block per wave, L3.KB k-steps of one fragment
    auto prime3 = [&]() {
        if (wave < MB) wring_prime<1, 4>(r3, L3.Wf + lane, 0, L3.KB);
    };
    hidden_layer<GH, MB, NW, D, AHEAD>(r2, L2, bl + L0.N + L1.N, Q, L1.NB, PH2, train_h ? n.H[2] : nullptr, mbg0, wave, lane, prime3);
    if (wave * GH >= L2.NB) hook_l2idle(extra);       // wavefronts without a strip in this layer
    __syncthreads();
    phase_stamp(a.dbg, 5);
    if (!AHEAD) prime3();
    // ---------------------------------------------------------------- head: one wave per 16-row block
    if (wave < MB) {
        f32x4 hacc[1][1];
        hacc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int CB3 = L2.NB;
        if (L3.KB % 4 == 0) mma_stream<1, 1, 4>(r3, L3.Wf + lane, 0, L3.KB, PH2 + wave * CB3 * 512, CB3, lane, hacc);
        else mma_ring<1, 1, 4>(r3, L3.Wf + lane, 0, 0, L3.KB, L3.KB, PH2 + wave * CB3 * 512, CB3, lane, hacc);
        const f32x4 acc = hacc[0][0];
        const int m = m0 + wave * 16 + r;
        const int No = L3.N;
        float mu[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) mu[e] = acc[e] + ((4 * q + e < No) ? bl[L0.N + L1.N + L2.N + 4 * q + e] : 0.0f);
        if (m < a.M && n.out) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < No) n.out[(int64_t)m * n.ldo + 4 * q + e] = mu[e];
        }
        hook_head(wave, m, 0, mu);  // `idle()`'s counterpart: every lane of the head wavefronts, with its row, the head's column block and its 4 outputs
        if (is_actor && a.sample) {
#pragma clang fp contract(off)      // a = mu + sigma z and the log-probability as separate fp32 roundings, identically in every translation unit
            const int A = a.A;
            float zz[4];
            if (a.z) {
#pragma unroll
                for (int e = 0; e < 4; ++e) zz[e] = (m < a.M && 4 * q + e < A) ? a.z[(int64_t)m * A + 4 * q + e] : 0.0f;
            } else {
                const int64_t s = HOIST ? step_pre : (a.step ? a.step[0] : 0);
                const RngKey rk = {a.k0, a.k1, (uint32_t)s, (uint32_t)(s >> 32)};
                const U4 u = rng4(rk, (uint32_t)m, SLOT_POLICY + (uint32_t)q);
                box_muller(u.x, u.y, zz[0], zz[1]);
                box_muller(u.z, u.w, zz[2], zz[3]);
            }
            float lp = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * q + e;
                if (j < A) {
                    const float sg = mu[e] * 0.0f + (HOIST ? std_pre[e] : a.std_[j]);      // actor_critic.py:113
                    const float act = mu[e] + sg * zz[e];
                    const float d = act - mu[e];
                    lp += -(d * d) / (2.0f * sg * sg) - (HOIST ? lsg_pre[e] : log_sigma(sg)) - 0.9189385332046727f;
                    if (m < a.M) {
                        a.actions[(int64_t)m * A + j] = act;
                        a.sigma[(int64_t)m * A + j] = sg;
                    }
                    hook_put(wave * 16 + r, j, act);
                }
            }
            lp += __shfl_xor(lp, 16, 64);
            lp += __shfl_xor(lp, 32, 64);
            if (q == 0 && m < a.M) a.logp[m] = lp;
        }
        // heads wider than one n-block (the auxiliary net, its own instantiation): the remaining blocks, same wave, one after
        // the other
        for (int nb = 1; WIDE && nb < L3.NB; ++nb) {
            const u32x4* wl = L3.Wf + (int64_t)nb * L3.KB * 64 + lane;
            wring_prime<1, 4>(r3, wl, 0, L3.KB);
            hacc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (L3.KB % 4 == 0) mma_stream<1, 1, 4>(r3, wl, 0, L3.KB, PH2 + wave * CB3 * 512, CB3, lane, hacc);
            else mma_ring<1, 1, 4>(r3, wl, 0, 0, L3.KB, L3.KB, PH2 + wave * CB3 * 512, CB3, lane, hacc);
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = nb * 16 + 4 * q + e;
                y[e] = hacc[0][0][e] + (col < No ? bl[L0.N + L1.N + L2.N + col] : 0.0f);
            }
            if (m < a.M && n.out) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (nb * 16 + 4 * q + e < No) n.out[(int64_t)m * n.ldo + nb * 16 + 4 * q + e] = y[e];
            }
            hook_head(wave, m, nb, y);
        }
    } else {
        hook_idle(extra);      // the wavefronts without a head block (NW - BM / 16 of them): free for the duration of the head
    }
    phase_stamp(a.dbg, 6);
}

template <int BM, int NW, int D, bool FIN = false>
__global__ __launch_bounds__(NW * 64) void mlp_fwd_kernel(const FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (FIN && (int)blockIdx.y >= a.nets) {       // the extra grid row: the previous env step's finaliser, one workgroup
        if (blockIdx.x == 0) fin_block(a.fin, threadIdx.x, NW * 64);
        return;
    }
    const int which = a.net0 + blockIdx.y;
    const FusedNet& n = a.net[which];
    const int g1 = n.layer[0].NB / NW;     // first hidden width 256 / 512 / 768
    constexpr int U = 16 / NW;             // n-blocks per wave per 256 columns
    if (n.layer[3].NB > 1) {               // wide head: the auxiliary net (first hidden width 512 only)
        if (g1 == 2 * U) fwd_body<BM, NW, D, 2 * U, true>(a, n, false, smem);
        return;
    }
    if (g1 == 2 * U) fwd_body<BM, NW, D, 2 * U>(a, n, which == 0, smem);
    else if (g1 == 3 * U) fwd_body<BM, NW, D, 3 * U>(a, n, which == 0, smem);
    else if (g1 == U) fwd_body<BM, NW, D, U>(a, n, which == 0, smem);
}

// ================================================================================================ backward (dX chain)
// dZ_out[m][k'] = (sum_n dZ_in[m][n] * W[n][k']) * elu'(H[m][k']): W^T fragments as the MFMA A operand.
// R: ring primed with this wave's first strip (bwd_prime); prime_next: called before the epilogue of the last strip.
template <int G, int D, int GR>
__device__ __forceinline__ void bwd_prime(WRing<GR, D>& R, const u32x4* __restrict__ WTf, int NBo, int NBBc, int wave, int lane) {
    const int nb0 = wave * G;
    if (nb0 < NBo) wring_prime<G, D>(R, WTf + (int64_t)nb0 * NBBc * 64 + lane, NBBc * 64, NBBc);
}

// HLDS / H_lds: the tile's y = elu(z) in LDS (block layout, NBo column blocks) instead of the global Hg -- the fused forward +
// backward kernel still has it there; out_lds may then be the SAME buffer (every lane reads a block entry and later writes that
// very entry).
template <int G, int MB, int NW, int D, bool AHEAD, bool HLDS = false, int GR, class Next>
__device__ __forceinline__ void bwd_step(WRing<GR, D>& R, const u32x4* __restrict__ WTf, int NBo, int NBBc, const char* in_lds, int CBin,
                                         char* out_lds, __bf16* __restrict__ dZg, const __bf16* __restrict__ Hg, int64_t mbg0, int wave,
                                         int lane, Next prime_next, const char* H_lds = nullptr) {
    const int r = lane & 15, q = lane >> 4;
    const int loff = r * 32 + q * 8;
    bool primed = false;
    for (int nb0 = wave * G; nb0 < NBo; nb0 += NW * G) {
        f32x4 acc[MB][G];
        zero_acc<G, MB>(acc);
        if (NBBc % D == 0) mma_stream<G, MB, D>(R, WTf + (int64_t)nb0 * NBBc * 64 + lane, NBBc * 64, NBBc, in_lds, CBin, lane, acc);
        else mma_ring<G, MB, D>(R, WTf + (int64_t)nb0 * NBBc * 64 + lane, NBBc * 64, 0, NBBc, NBBc, in_lds, CBin, lane, acc);
        const int nxt = nb0 + NW * G;
        if (nxt < NBo) wring_prime<G, D>(R, WTf + (int64_t)nxt * NBBc * 64 + lane, NBBc * 64, NBBc);
        else if (AHEAD) { prime_next(); primed = true; }
        u32x2 aux[MB][G];     // y = elu(z) of this tile (every load is issued before the first store below)
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (HLDS) aux[i][g] = *reinterpret_cast<const u32x2*>(H_lds + (i * NBo + nb0 + g) * 512 + loff);
                else aux[i][g] = ld_stream_u2<(kFusedNT & 4) != 0>(reinterpret_cast<const char*>(Hg) + ((mbg0 + i) * NBo + nb0 + g) * 512 + loff);
            }
        static_assert(MB % 2 == 0, "row blocks are stored in pairs");
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
                char* pa = reinterpret_cast<char*>(dZg) + ((mbg0 + i) * NBo + nb0 + g) * 512 + loff;
                st_pair<(kFusedNT & 4) != 0, false>(pa, pa + (int64_t)NBo * 512, pk[0], pk[1], true, true, q);
            }
    }
    if (AHEAD && !primed) prime_next();
}

// ================================================================================================ forward + loss + dZ chain
// One kernel per minibatch instead of forward -> loss -> backward launches: a tile's
// forward leaves H1 and H2 in LDS (the two ping-pong buffers), the head wavefronts turn their outputs into the PPO loss gradient
// (the per-sample arithmetic of ppo_loss_kernel, one lane per (row, 4 actions)) -- dZ3 goes to LDS and to HBM -- and the dZ chain
// runs on the resident tile: through W3 against H2 (LDS, in place), through W2 against H1 (LDS, in place), through W1 against
// H0 (LDS: the forward is told to put H2 into a buffer of its own instead of over H0).
// HBM traffic per minibatch drops by the H re-reads and the loss kernel's own gathers; three launch boundaries go away.
struct FbLoss {
    const float* actions;      // (T*N, A) storage columns, gathered through FwdArgs::idx
    const float* old_mu;
    const float* old_sigma;
    const float* values;       // (T*N,)
    const float* advantages;
    const float* returns;
    const float* logp;
    float clip, value_coef, entropy_coef;
    float* partials;           // [tiles][32] per-tile sums, ppo_loss_kernel's layout: 0 surrogate, 1 value loss, 2 entropy, 3 kl,
                               // 4..15 d std, 16..27 sum d mu (head bias gradient), 28 sum d V, 29 the auxiliary head's squared error;
                               // the actor tile writes its entries, the critic tile its two, the auxiliary tile its one
    // auxiliary (denoising) head, a third grid row: MSE against columns [aux_off, aux_off + layer[3].N) of the gathered target rows
    const float* aux_target;
    int64_t aux_ldt;
    int aux_off;
    float aux_coef;
};

// LDS of the fused kernel behind the forward's P / Q / bias regions: the dZ3 tile (64 rows x 32 * layer[3].NBB bf16 columns), the
// head waves' partial sums, and H2 (64 x layer[2].N bf16) -- the forward writes it there instead of over H0, so that all three
// activations are resident for the dZ chain
// ... and, behind H2, the storage row of each of the tile's 64 rows (256 B) and the loss inputs gathered through them: actor 64 x
// [actions 12 | old mu 12 | old sigma 12 | advantage | old log-prob | pad 2] floats, critic 64 x [return | old value]
constexpr int FB_LIN_ACTOR = 40;      // floats per row
HG_HD int fb_lds_lin(const FusedNet& n) { return n.layer[3].N == 1 ? 64 * 2 * 4 : (n.layer[3].N <= 12 ? 64 * FB_LIN_ACTOR * 4 : 0); }
HG_HD int fb_lds_extra(const FusedNet& n) { return 64 * 64 * n.layer[3].NBB + 4 * 32 * 4 + 64 * n.layer[2].N * 2 + 256 + fb_lds_lin(n); }

template <int G1, bool AUX = false, bool XB16 = false>
__device__ __forceinline__ void fb_body(const FwdArgs& a, const FbLoss& L, const FusedNet& n, bool is_actor, char* smem) {
    constexpr int BM = 64, NW = 16, D = 2, MB = BM / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int64_t mbg0 = m0 >> 4;
    char* P = smem;
    char* Q = smem + fused_lds_p(n, BM);
    char* R0 = smem + fused_lds_p(n, BM) + fused_lds_q(n, BM) + fused_lds_bias(n);
    const int NBB3 = n.layer[3].NBB, CB3 = 2 * NBB3;      // the head gradient's column blocks of 16 (padded to the backward's k-steps of 32)
    float* red = reinterpret_cast<float*>(R0 + BM * 64 * NBB3);
    char* H2 = R0 + BM * 64 * NBB3 + 4 * 32 * 4;
    int* rowidx = reinterpret_cast<int*>(H2 + BM * n.layer[2].N * 2);
    float* lin = reinterpret_cast<float*>(rowidx + BM);
    const int A = a.A;
    // The loss inputs of the tile's rows (scattered 48-byte rows and scalars of the storage) are gathered into LDS by the eight
    // wavefronts that have no strip in the 128-wide third layer, while the other eight compute it: the head wavefronts used to
    // start with two dependent round trips (a.idx[m], then the rows) -- 5 of an actor tile's 47 us.
    const bool pre = !AUX && n.layer[2].NB <= 8 && (is_actor ? A == 12 : true);
    const float invB = 1.0f / (float)a.M;
    float aux_se = 0.0f;
    // auxiliary head: lane (r, q) of head wave hw, row m, outputs nb * 16 + 4q .. + 3 (called once per column block).
    // loss = coef * mean_b sum_j (y - t)^2 / No, dL/dy = 2 coef (y - t) / (B No)    (aux_mse_kernel's arithmetic)
    auto head_aux = [&](int hw, int m, int nb, const float (&out)[4]) {
        const int No = n.layer[3].N;
        const bool valid = m < a.M;
        const int64_t row = rowidx[hw * 16 + r];
        const float* t = L.aux_target + row * L.aux_ldt + L.aux_off;
        const float gs = 2.0f * L.aux_coef / ((float)a.M * (float)No);
        float g[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = nb * 16 + 4 * q + e;
            float d = 0.0f;
            if (valid && j < No) {
                d = out[e] - t[j];
                aux_se += d * d;
            }
            g[e] = gs * d;
        }
        const u32x2 pk = pack_bf16x4(g[0], g[1], g[2], g[3]);
        const int off = (hw * CB3 + nb) * 512 + r * 32 + q * 8;
        char* gz = reinterpret_cast<char*>(n.dZ[3]) + mbg0 * CB3 * 512;
        *reinterpret_cast<u32x2*>(R0 + off) = pk;
        *reinterpret_cast<u32x2*>(gz + off) = pk;
        if (nb == 0) {      // the padding blocks behind the head's last one
            const u32x2 zero = {0u, 0u};
            for (int cb = n.layer[3].NB; cb < CB3; ++cb) {
                const int offz = (hw * CB3 + cb) * 512 + r * 32 + q * 8;
                *reinterpret_cast<u32x2*>(R0 + offz) = zero;
                *reinterpret_cast<u32x2*>(gz + offz) = zero;
            }
        }
    };
    auto l2idle = [&](int) {
        if (!pre) return;
        const int j = tid - 8 * 64;                 // 0 .. 511 (wavefronts 8 .. 15)
        if (j < 0) return;
        if (is_actor) {
            // round 1: (row, quad k): actions 0..2, old mu 3..5, old sigma 6..7; round 2: old sigma quad 2, advantage, old log-prob.
            // Every load is issued before the first LDS write.
            const int rw = j >> 3, k = j & 7;
            const int64_t ri = rowidx[rw];
            const float* base = k < 3 ? L.actions : (k < 6 ? L.old_mu : L.old_sigma);
            const F4 v1 = *reinterpret_cast<const F4*>(base + ri * 12 + 4 * (k < 3 ? k : (k < 6 ? k - 3 : k - 6)));
            // (unconditional loads, conditional stores: a conditionally initialised vector ends up in private memory)
            const int rw2 = j & 63;
            const int64_t ri2 = rowidx[rw2];
            const F4 v2 = *reinterpret_cast<const F4*>(L.old_sigma + ri2 * 12 + 8);
            const float s2 = (j < 128 ? L.advantages : L.logp)[ri2];
            *reinterpret_cast<F4*>(lin + rw * FB_LIN_ACTOR + 4 * k) = v1;
            if (j < 64) *reinterpret_cast<F4*>(lin + rw2 * FB_LIN_ACTOR + 32) = v2;
            else if (j < 192) lin[rw2 * FB_LIN_ACTOR + (j < 128 ? 36 : 37)] = s2;
        } else if (j < 128) {
            const int rw = j & 63;
            const int64_t ri = rowidx[rw];
            lin[rw * 2 + (j >> 6)] = j < 64 ? L.returns[ri] : L.values[ri];
        }
    };
    auto head = [&](int hw, int m, int, const float (&out)[4]) {
        // lane (r, q) of head wave hw: row m, head outputs 4q .. 4q + 3.  ppo.py:128-168 forward scalars + the hand-written
        // backward of the loss w.r.t. mu, std and V (oracle/ppo_oracle.py: ppo_loss_and_grads), as in ppo_loss_kernel
        const bool valid = m < a.M;
        const int rl = hw * 16 + r;                 // row of the tile
        const int64_t row = rowidx[rl];
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        float part[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) part[k] = 0.0f;
        if (is_actor) {
            float act[4] = {0.f, 0.f, 0.f, 0.f}, mo[4] = {0.f, 0.f, 0.f, 0.f}, so[4] = {1.f, 1.f, 1.f, 1.f}, sg[4] = {1.f, 1.f, 1.f, 1.f};
            if (pre) {
                if (q < 3) {
                    const F4 qa = *reinterpret_cast<const F4*>(lin + rl * FB_LIN_ACTOR + 4 * q);
                    const F4 qo = *reinterpret_cast<const F4*>(lin + rl * FB_LIN_ACTOR + 12 + 4 * q);
                    const F4 qs = *reinterpret_cast<const F4*>(lin + rl * FB_LIN_ACTOR + 24 + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { act[e] = qa.v[e]; mo[e] = qo.v[e]; so[e] = qs.v[e]; }
                }
            } else if (4 * q + 3 < A) {
                const F4 qa = *reinterpret_cast<const F4*>(L.actions + row * A + 4 * q);
                const F4 qo = *reinterpret_cast<const F4*>(L.old_mu + row * A + 4 * q);
                const F4 qs = *reinterpret_cast<const F4*>(L.old_sigma + row * A + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) { act[e] = qa.v[e]; mo[e] = qo.v[e]; so[e] = qs.v[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * q + e < A) { act[e] = L.actions[row * A + 4 * q + e]; mo[e] = L.old_mu[row * A + 4 * q + e]; so[e] = L.old_sigma[row * A + 4 * q + e]; }
            }
            const float adv = pre ? lin[rl * FB_LIN_ACTOR + 36] : L.advantages[row], lpold = pre ? lin[rl * FB_LIN_ACTOR + 37] : L.logp[row];
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
            const float ret = pre ? lin[rl * 2] : L.returns[row], vold = pre ? lin[rl * 2 + 1] : L.values[row];
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
        const int off0 = (hw * 2 + 0) * 512 + r * 32 + q * 8, off1 = (hw * 2 + 1) * 512 + r * 32 + q * 8;
        *reinterpret_cast<u32x2*>(R0 + off0) = pk;
        *reinterpret_cast<u32x2*>(R0 + off1) = zero;
        char* gz = reinterpret_cast<char*>(n.dZ[3]) + mbg0 * 2 * 512;
        *reinterpret_cast<u32x2*>(gz + off0) = pk;
        *reinterpret_cast<u32x2*>(gz + off1) = zero;
        // sums over the 16 rows of this wave (lanes sharing q), then across the head waves through `red`
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            float v = part[k];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            part[k] = v;
        }
        if (r == 0) {
            float* w = red + hw * 32;
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
    };
    if constexpr (AUX) {
        fwd_body<BM, NW, D, G1, true, XB16>(a, n, false, smem, FwdNoop(), FwdNoop(), FwdNoop(), 0, FwdNoop(), head_aux, H2, rowidx);
        if (wave < MB) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) aux_se += __shfl_xor(aux_se, o, 64);
            if (lane == 0) red[wave * 32 + 29] = aux_se;
        }
    } else {
        fwd_body<BM, NW, D, G1, false, XB16>(a, n, is_actor, smem, FwdNoop(), FwdNoop(), FwdNoop(), 0, FwdNoop(), head, H2, rowidx, l2idle);
    }
    __syncthreads();          // dZ3 tile and the per-wave sums are in LDS; H0 sits in P, H1 in Q, H2 in its own buffer
    if (tid < 32) {
        const bool mine = AUX ? tid == 29 : (is_actor ? (tid != 1 && tid < 28) : (tid == 1 || tid == 28));
        if (mine) L.partials[(int64_t)blockIdx.x * 32 + tid] = red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid];
    }
    // ---- dZ chain on the resident tile
    const int N0 = n.layer[0].N, N1 = n.layer[1].N, N2 = n.layer[2].N;
    WRing<1, D> ra;
    WRing<1, D> rb;
    WRing<G1, D> rc;
    auto none = [&]() {};
    bwd_prime<1, D>(ra, n.layer[3].WTf, N2 / 16, NBB3, wave, lane);
    bwd_step<1, MB, NW, D, false, true>(ra, n.layer[3].WTf, N2 / 16, NBB3, R0, 2 * NBB3, H2, n.dZ[2], nullptr, mbg0, wave, lane, none, H2);
    __syncthreads();
    bwd_prime<1, D>(rb, n.layer[2].WTf, N1 / 16, n.layer[2].NBB, wave, lane);
    bwd_step<1, MB, NW, D, false, true>(rb, n.layer[2].WTf, N1 / 16, n.layer[2].NBB, H2, N2 / 16, Q, n.dZ[1], nullptr, mbg0, wave, lane, none, Q);
    __syncthreads();
    bwd_prime<G1, D>(rc, n.layer[1].WTf, N0 / 16, n.layer[1].NBB, wave, lane);
    bwd_step<G1, MB, NW, D, false, true>(rc, n.layer[1].WTf, N0 / 16, n.layer[1].NBB, Q, N1 / 16, nullptr, n.dZ[0], nullptr, mbg0, wave, lane, none, P);
    phase_stamp(a.dbg, 7);
}

// XB16: every net of the launch gathers its input rows from the bf16 shadow (FusedNet::xb) instead of the fp32 storage rows
template <bool XB16 = false>
__global__ __launch_bounds__(1024) void mlp_fb_kernel(const FwdArgs a, const FbLoss L) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int which = a.net0 + blockIdx.y;
    const FusedNet& n = a.net[which];
    const int g1 = n.layer[0].NB / 16;     // first hidden width 256 / 512 / 768
    if (which == 2) {                      // the auxiliary head (wide head, first hidden width 512 only): MSE instead of the PPO loss
        if (g1 == 2) fb_body<2, true, XB16>(a, L, n, false, smem);
        return;
    }
    if (g1 == 2) fb_body<2, false, XB16>(a, L, n, which == 0, smem);
    else if (g1 == 3) fb_body<3, false, XB16>(a, L, n, which == 0, smem);
    else if (g1 == 1) fb_body<1, false, XB16>(a, L, n, which == 0, smem);
}

// ================================================================================================ per-minibatch loss scalars
constexpr int LOSS_PARTIALS = 32;   // floats per loss workgroup: surrogate, value loss, entropy, kl, dstd[12], dbias_mu[12], dbias_v, aux, pad
// Sums the per-tile loss partials of a minibatch ([nblocks][32], mlp_fb_kernel / ppo_loss_kernel) into opt_state and writes the
// gradients that are plain column sums: std, the two head biases (fused path), the KL slot behind the flat gradient.
// One workgroup of 256 or 512 threads -- its own launch (ppo_scalars_kernel) or, on the fused path, one extra workgroup of the
// weight-gradient launch that follows the loss anyway (dw_kernel_rs: DwArgs::scal_bid).  The summation order does not depend on the
// thread count (16 interleaved partial sums per quantity, combined in a fixed order).
struct ScalArgs {
    int nblocks, B, A, aux_No;
    const float* partials;
    float* grads_std;
    float* grads_bmu;      // may be null (generic path: the head bias gradients come from the GEMMs)
    float* grads_bv;
    float* kl_slot;
    double* opt;
    double beta1, beta2;   // Adam's: beta^t of the step this gradient will be applied in is left in opt[13..15] (below)
    // hgym_ppo_apply's single-thread prologue (adaptive-KL learning rate, Adam step count, bias corrections) done HERE, beside the
    // weight-gradient launch, when the caller has promised that exactly this gradient is applied next on one rank
    // (HgymPPOConfig.grad_norm_ready): it needs nothing but the minibatch KL this block has just formed, and as a launch of its own it
    // sat alone on the critical path between the slab sum and Adam (4.7 us + a launch boundary, eight times per iteration)
    int do_prologue, adaptive_lr;
    float desired_kl;
    double lr_min, lr_max;
    int group;             // 0 / 1: `partials` holds one row per loss workgroup; 4: one row per 16-row block (mlp_fb2_kernel), four
                           // consecutive rows are added first, ((p0 + p1) + p2) + p3 in fp32 -- the sum mlp_fb_kernel's 64-row tile forms
                           // over its four head waves -- and nblocks counts those groups
};
__device__ __forceinline__ void ppo_scalars_block(const ScalArgs& a, int tid, int nthreads) {
    __shared__ double red[16][LOSS_PARTIALS + 1];
    // beta1^t, beta2^t for hgym_ppo_apply's prologue: two double-precision pow() are ~6 us in that single-thread kernel, on the
    // minibatch's critical path; here they run on two lanes of the last wavefront beside a launch that takes 170 us anyway.  Keyed
    // by t (opt[13]): the prologue recomputes them if the step count is not the one assumed here.
    // opt[13] is also the "prologue done, not applied yet" marker: == opt[1] (> 0) after a prologue taken here or in apply_prologue_kernel,
    // -1 once adam_kernel has applied the step.  A second gradient call before the apply finds the marker set and leaves the step's
    // learning-rate decision, step count and powers alone (the update then is the reference's: one optimiser step per apply).
    if (tid >= nthreads - 2 && !(a.opt[13] == a.opt[1] && a.opt[1] > 0.0)) {
        const double t = a.opt[1] + 1.0;
        const double pw = pow(tid == nthreads - 2 ? a.beta1 : a.beta2, t);
        a.opt[tid == nthreads - 2 ? 14 : 15] = pw;
        if (tid == nthreads - 2) a.opt[13] = t;
    }
    const float* __restrict__ partials = a.partials;
    const int nblocks = a.nblocks, B = a.B, A = a.A;
    const int k = tid & (LOSS_PARTIALS - 1);
    for (int part = tid / LOSS_PARTIALS; part < 16; part += nthreads / LOSS_PARTIALS) {     // 16 partial sums per quantity
        // four independent partial sums (combined in a fixed order): a single dependent chain of nblocks / 16 loads was
        // latency-bound once the fused forward + backward kernel started handing in one partial row per 64-row tile (960 rows)
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int b = part;
        if (a.group == 4) {
            auto ld = [&](int g) -> float {
                const float* p = partials + (int64_t)g * 4 * LOSS_PARTIALS + k;
                const float p0 = p[0], p1 = p[LOSS_PARTIALS], p2 = p[2 * LOSS_PARTIALS], p3 = p[3 * LOSS_PARTIALS];
                return ((p0 + p1) + p2) + p3;
            };
            for (; b + 48 < nblocks; b += 64) {
                const float v0 = ld(b), v1 = ld(b + 16), v2 = ld(b + 32), v3 = ld(b + 48);
                s0 += (double)v0; s1 += (double)v1; s2 += (double)v2; s3 += (double)v3;
            }
            for (; b < nblocks; b += 16) s0 += (double)ld(b);
        } else {
            for (; b + 48 < nblocks; b += 64) {
                const float v0 = partials[(int64_t)b * LOSS_PARTIALS + k], v1 = partials[(int64_t)(b + 16) * LOSS_PARTIALS + k];
                const float v2 = partials[(int64_t)(b + 32) * LOSS_PARTIALS + k], v3 = partials[(int64_t)(b + 48) * LOSS_PARTIALS + k];
                s0 += (double)v0; s1 += (double)v1; s2 += (double)v2; s3 += (double)v3;
            }
            for (; b < nblocks; b += 16) s0 += (double)partials[(int64_t)b * LOSS_PARTIALS + k];
        }
        red[part][k] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (tid < LOSS_PARTIALS) {
        double t = 0.0;
        for (int p = 0; p < 16; ++p) t += red[p][tid];
        const int q = tid;
        double* __restrict__ opt = a.opt;
        if (q == 0) opt[3] += t / B;
        if (q == 1) opt[4] += t / B;
        if (q == 2) opt[5] += t / B;
        if (q == 3) {
            opt[2] += t / B;
            opt[8] = t / B;
            opt[7] += 1.0;
            opt[9] = 0.0;                     // squared gradient norm: accumulated by reduce_slabs_kernel later in this call
            a.kl_slot[0] = (float)(t / B);    // grads[P]: travels with the gradient in the ranks' one all-reduce
            if (a.do_prologue && !(opt[13] == opt[1] && opt[1] > 0.0)) {      // apply_prologue_kernel's arithmetic (hgym_net.hip), one rank: ppo.py:140-148 in python doubles
                double lr = opt[0];
                if (a.adaptive_lr) {
                    const double kl = t / B;
                    if (kl > (double)a.desired_kl * 2.0) lr = fmax(a.lr_min, lr / 1.5);
                    else if (kl < (double)a.desired_kl / 2.0 && kl > 0.0) lr = fmin(a.lr_max, lr * 1.5);
                    opt[0] = lr;
                }
                const double ts = opt[1] + 1.0;           // (= opt[13]: the two lanes above prepared beta^ts before the barrier)
                opt[1] = ts;
                const double bc1 = 1.0 - opt[14], bc2 = 1.0 - opt[15];
                opt[11] = (double)(float)(lr / bc1);      // step size
                opt[12] = (double)(float)sqrt(bc2);
            }
        }
        if (q >= 4 && q < 16 && q - 4 < A) a.grads_std[q - 4] = (float)t;
        if (q >= 16 && q < 28 && q - 16 < A && a.grads_bmu) a.grads_bmu[q - 16] = (float)t;
        if (q == 28 && a.grads_bv) a.grads_bv[0] = (float)t;
        if (q == 29 && a.aux_No > 0) opt[10] += t / ((double)B * (double)a.aux_No);    // auxiliary head's MSE (the fused kernel's third grid row)
    }
}

// ================================================================================================ weight gradients
// dW[n][k] = sum_m Z[m][n] * X[m][k]  (Z = dZ_l, X = layer input): tiles of 128 dZ columns x 256 input columns, eight wavefronts
// (64 x 64 each), contraction split over blockIdx.y.
//
// How the kernel got its shape (all on B = 61 440, one box, profiles/r03_dw_ablation.txt):
//  * Operand transport: every wave keeps DW_RS stages of its three 1-KiB pieces in registers (global_load_dwordx4) and passes one
//    stage per step to the LDS ring with ds_write_b128; fragments are read with the transpose read.  (A first version DMA'd the
//    pieces straight into LDS with global_load_lds: a piece costs the issuing wave 60-185 cycles (MI355X_MICROARCH.md, LDS-DMA
//    issue cost) next to 20 MFMAs = 320 cycles, so the DMA issue set the pace -- 272 us per launch against 183 us.)
//  * Round 2's kernel (128 x 128 tiles, four waves, two workgroups per CU, 189 us) took the same time with its LDS traffic
//    (177.9 us) or three quarters of its MFMAs (186.1 us) removed, and 135 us when every load hit a page of zeros: the pace is set
//    by the CU's fill path -- 16 KiB per 32-row step per tile, two tiles per CU -- and neither by MFMA nor LDS; not by HBM either:
//    a quarter of the rows, written microseconds earlier and still in the Infinity Cache, is no faster per row, and register rings
//    of 2..5 stages all measure the same.  A 128 x 256 tile moves 24 KiB per step for twice the products -- three quarters of the
//    bytes per MFMA through L1 -- and 32 tiles x 8 splits is exactly one workgroup per CU: 189 -> 180 us.
//  * On L1-resident operands, removing the LDS reads, the LDS writes or three quarters of the MFMAs each saved its own 20-30 us of
//    the 135: a sum, not a maximum -- the waves leave every barrier in the same phase, two MFMA streams queue at each matrix
//    pipe and then nobody uses it.  Hence the two wave groups half a step apart (below): 180 -> 174 us (zero-page floor 133 -> 120).
//  * What remains above the floor is the fill path itself: +29 us when the operands come from L2 (the same two row blocks over and
//    over), +24 us more from HBM.
constexpr int DW_THREADS = 512;
constexpr int DW_Z_BYTES = 8192;           // 32 rows x 8 Z blocks x 512 B
constexpr int DW_STAGE_BYTES = 24576;      // + 32 rows x 16 X blocks
constexpr int DW_LDS_STAGES = 3;
constexpr int DW_TILE_N = 128, DW_TILE_K = 256;
constexpr int DW_MAX_PRODUCTS = 12;        // actor + critic + auxiliary head, four layers each

struct DwProduct {
    const __bf16* Z;      // block layout, CBz column blocks per row block
    const __bf16* X;      // block layout, CBx -- or, with gidx, the row-major bf16 shadow the rows are gathered from
    const int64_t* gidx;  // non-null: batch row m of X is row gidx[m] of the shadow (leading dimension ldg elements, CBx = ldg / 16):
                          // a first-layer product whose operand was never copied (mlp_fb_kernel<XB16>)
    int64_t ldg;
    int CBz, CBx;
    int N, K;             // valid rows / cols of dW (row-major, leading dimension K)
    int64_t w_off;        // offset of dW in a slab (floats)
    int64_t b_off;        // offset of the bias gradient (column sums of Z), -1: none
    int tiles_n, tiles_k, tile0;
};

struct DwArgs {
    DwProduct p[DW_MAX_PRODUCTS];
    int np;
    int total_tiles;
    int splits;
    int steps_total;       // row blocks of 32 over the (padded) batch
    int steps_per_split;
    float* slabs;
    int64_t slab_stride;   // floats
    const char* zeros;     // 4 KiB of zeros (workspace): source of the stages past the end of a split
    int B;                 // valid batch rows (gathered products clamp their row indices to it)
    int scal_bid;          // >= 0: this workgroup sums the minibatch's loss partials instead (ppo_scalars_block); -1: none
    ScalArgs sc;
};

__device__ __forceinline__ u32x4 tr_frag(const char* p0, const char* p1) {
    typedef s16x4 __attribute__((address_space(3)))* lds_p;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(const_cast<char*>(p0)));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(const_cast<char*>(p1)));
    struct { s16x4 a, b; } pr = {a, b};
    return __builtin_bit_cast(u32x4, pr);
}

template <int DW_RS>      // register stages in flight per wave
__global__ __launch_bounds__(DW_THREADS, 1) void dw_kernel_rs(const DwArgs a) {
    static_assert(DW_RS == 3, "ring indices below: three register stages over three LDS stages, unrolled by six");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;                  // 64-column strip of the X side (4), of the Z side (2)
    const int r = lane & 15, q = lane >> 4;
    const int bid = blockIdx.x;
    if (bid == a.scal_bid) {      // the loss scalars ride in this launch: one workgroup, ~10 us beside 170 us of tiles
        ppo_scalars_block(a.sc, tid, DW_THREADS);
        return;
    }
    // Block -> (tile, split).  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), each with a private
    // 4 MiB L2.  All tiles of one split stream the SAME batch rows, so a split is pinned to one XCD: its 32 tiles run there
    // concurrently and every operand block is fetched from HBM once and re-read from that L2 by the other tiles.
    const int xcd = bid & 7, slot = bid >> 3;
    const int split = xcd + 8 * (slot / a.total_tiles);
    const int tile = slot % a.total_tiles;
    if (split >= a.splits) return;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < DW_MAX_PRODUCTS; ++i)
        if (i < a.np && tile >= a.p[i].tile0) pi = i;
    const DwProduct& P = a.p[pi];
    const int tl = tile - P.tile0;
    const int tn = tl / P.tiles_k, tk = tl - tn * P.tiles_k;
    const int cbz0 = tn * 8, cbx0 = tk * 16;
    const int step0 = split * a.steps_per_split;
    int nsteps = a.steps_total - step0;
    nsteps = nsteps < a.steps_per_split ? nsteps : a.steps_per_split;

    // Wave w moves, of every 32-row stage, row block w & 1 of Z block pair w >> 1 and of X block pairs 2 (w >> 1), 2 (w >> 1) + 1
    // (piece = two adjacent 16 x 16 blocks of one row block, 1 KiB, lane-linear).
    //
    // Addressing: every load is (wave-uniform 64-bit base in SGPRs, advanced by scalar arithmetic) + (32-bit per-lane byte offset),
    // the saddr form of global_load; per-lane 64-bit addresses cost ~50 of a step's 126 instructions.  "Past the end of the split ->
    // the page of zeros" is a scalar select on that base, never a branch around a load (a conditional load makes the compiler's
    // vmcnt conservative), and an OFFSET from the operand's own pointer (a pointer rebuilt from an integer becomes a FLAT access).
    //
    // Gathered X operand (first-layer products on the bf16 shadow, < 4 GiB: fused_grad checks): the lane that holds bytes
    // [16 l, 16 l + 16) of a piece holds row (l & 31) >> 1 of the row block, block l >> 5 of the pair, half l & 1 -- in the row-major
    // shadow that is 16 bytes of row gidx[m].  The 16 row indices of a stage are one 4-byte load per lane (row l & 15, replicated
    // over the wave) issued TWO stages ahead of the stage's data and in front of the data loads issued with it: a wave's loads
    // return in order, so when the index is needed only the younger stages are behind it and the data ring keeps its depth.  Every
    // tile issues the index load (from the zero page when it has nothing to gather): one instruction stream, exact vmcnt.
    const int mbl = __builtin_amdgcn_readfirstlane(wave & 1), pp = __builtin_amdgcn_readfirstlane(wave >> 1);
    const char* zbase = reinterpret_cast<const char*>(P.Z);
    const char* xbase = reinterpret_cast<const char*>(P.X);
    const bool gx = P.gidx != nullptr;
    const unsigned int lofs = lane * 16;
    int cbz = cbz0 + 2 * pp;
    cbz = cbz + 1 < P.CBz ? cbz : P.CBz - 2;                          // surplus pairs: clamped onto valid blocks (results unused)
    const int64_t zofs = ((int64_t)(2 * step0 + mbl) * P.CBz + cbz) * 512;
    const int64_t zstep = (int64_t)2 * P.CBz * 512;                   // bytes per 32-row step
    const unsigned int ldgb = (unsigned int)(P.ldg * 2);
    const int irow = mbl * 16 + (lane & 15);                          // row of the 32-row step whose index this lane fetches
    const int bsrc = ((lane & 31) >> 1) * 4;                          // ds_bpermute address: the lane that holds this lane's row
    unsigned int vb[2];                                               // per-lane constant part of the byte offset
    int64_t xofs[2];                                                  // wave-uniform part: the piece's offset in a linear operand
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int cb = cbx0 + 2 * (2 * pp + j);
        cb = cb + 1 < P.CBx ? cb : P.CBx - 2;
        const unsigned int gcol = (unsigned int)(((cb + (lane >> 5)) * 16 + (lane & 1) * 8) * 2);
        vb[j] = gx ? gcol : lofs;
        xofs[j] = gx ? 0 : ((int64_t)(2 * step0 + mbl) * P.CBx + cb) * 512;
    }
    const int64_t xstep = gx ? 0 : (int64_t)2 * P.CBx * 512;
    const int64_t zdz = a.zeros - zbase, zdx = a.zeros - xbase;
    const char* ibase = gx ? reinterpret_cast<const char*>(P.gidx) : a.zeros;
    const int mg0 = gx ? -1 : 0;
    auto load_idx = [&](int t) -> int {
        int mrow = (step0 + (t < nsteps ? t : nsteps - 1)) * 32 + irow;
        mrow = mrow < a.B ? mrow : a.B - 1;
        // the low word only (indices are < 2^31): the dead upper half of an 8-byte load is a register the allocator hands out
        // again at once, and overwriting it has to wait for the load -- i.e. for every older load of the ring
        return *reinterpret_cast<const int*>(ibase + ((unsigned int)((mrow & mg0) | ((lane & 15) & ~mg0)) << 3));
    };
    // index ring: slot t % 3 holds the indices of stage t (a compile-time slot in the unrolled loop: rotating two registers
    // instead costs a move of a just-loaded value, i.e. a wait that drains the data ring once per revolution)
    int GI[3];
    GI[0] = load_idx(0);
    GI[1] = load_idx(1);
    u32x4 R[DW_RS][3];
    auto load = [&](int t, int slot3, u32x4 (&rr)[3]) {               // slot3 == t % 3
        const bool in = t < nsteps;                                   // wave-uniform: scalar selects below
        GI[(slot3 + 2) % 3] = load_idx(t + 2);                        // issued BEFORE this stage's data
        const unsigned int rmask = (gx && in) ? 0xffffffffu : 0u;
        const unsigned int rterm = ((unsigned int)__builtin_amdgcn_ds_bpermute(bsrc, GI[slot3]) * ldgb) & rmask;
        const int64_t m = in ? -1 : 0;                                // (integer masks: a `?:` on the pointer becomes a branch)
        rr[0] = *reinterpret_cast<const u32x4*>(zbase + (((zofs + (int64_t)t * zstep) & m) | (zdz & ~m)) + lofs);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            rr[1 + j] = *reinterpret_cast<const u32x4*>(xbase + (((xofs[j] + (int64_t)t * xstep) & m) | (zdx & ~m)) + (vb[j] + rterm));
    };
    // LDS stage: [row block][8 Z blocks][512 B], then [row block][16 X blocks][512 B]
    auto store = [&](int buf, const u32x4 (&rr)[3]) {
        char* dz = smem + buf * DW_STAGE_BYTES + mbl * 4096 + pp * 1024 + lane * 16;
        char* dx = smem + buf * DW_STAGE_BYTES + DW_Z_BYTES + mbl * 8192 + pp * 2048 + lane * 16;
        *reinterpret_cast<u32x4*>(dz) = rr[0];
        *reinterpret_cast<u32x4*>(dx) = rr[1];
        *reinterpret_cast<u32x4*>(dx + 1024) = rr[2];
    };

    f32x4 acc[4][4];
    f32x4 accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = (P.b_off >= 0) && tk == 0 && wi == 0;
    const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    auto frags = [&](int stage, u32x4 (&xa)[4], u32x4 (&zb)[4]) {
        const char* st = smem + stage * DW_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* px = st + DW_Z_BYTES + (wi * 4 + i) * 512 + lane * 8;
            xa[i] = tr_frag(px, px + 8192);
            const char* pz = st + (wj * 4 + i) * 512 + lane * 8;
            zb[i] = tr_frag(pz, pz + 4096);
        }
    };
    auto mfmas = [&](const u32x4 (&xa)[4], const u32x4 (&zb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) mma_frag<__bf16>(xa[i], zb[jj], acc[i][jj]);
        if (do_bias) {      // column sums of Z (measured: making every tile issue them, to keep a split's tiles in step, changes nothing)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) mma_frag<__bf16>(ones, zb[jj], accb[jj]);
        }
    };

    // Pipeline.  The LDS ring holds stages t, t + 1 (read) and t + 2 (being written), the register ring stages t + 2 .. t + 4; the
    // MFMAs of step t run on fragments read during step t - 1.  A step is two half-steps with a barrier after each, and the
    // wavefronts work in two groups half a step apart: while waves 0-3 (one per SIMD) issue the MFMAs of a step, waves 4-7 --
    // their SIMD partners -- move data (global loads of stage t + 4, LDS writes of stage t + 2, transpose reads of step t + 1),
    // and vice versa.  The lagging group enters one barrier late and the leading group leaves one barrier late: every wave
    // executes the same number of barriers.  Stage t + 2 is written in half-steps 2t + 1 (leading) and 2t + 2 (lagging) and first
    // read in half-step 2t + 3; its buffer was last read (stage t - 1) in half-step 2t - 2.  The steady-state loop is branch-free
    // -- stages past the end of the split come from the page of zeros and add nothing.
    u32x4 XA[2][4], ZB[2][4];
#pragma unroll
    for (int j = 0; j < DW_RS; ++j) load(j, j % 3, R[j]);
    store(0, R[0]);
    load(DW_RS, DW_RS % 3, R[0]);
    store(1, R[1]);
    __syncthreads();
    frags(0, XA[0], ZB[0]);
    const int np = (nsteps + 5) / 6 * 6;
    const int lagw = __builtin_amdgcn_readfirstlane(wave >> 2);
    if (lagw) __builtin_amdgcn_s_barrier();
    for (int t0 = 0; t0 < np; t0 += 6) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int t = t0 + j;
            __builtin_amdgcn_s_setprio(1);
            mfmas(XA[j & 1], ZB[j & 1]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            load(t + 1 + DW_RS, (j + 1 + DW_RS) % 3, R[(j + 1) % DW_RS]);      // that slot held stage t + 1: in LDS since step t - 1
            __builtin_amdgcn_sched_barrier(0);
            store((j + 2) % 3, R[(j + 2) % DW_RS]);                            // stage t + 2, loaded two steps ago
            frags((j + 1) % 3, XA[(j + 1) & 1], ZB[(j + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    }
    if (!lagw) __builtin_amdgcn_s_barrier();

    // lane holds dW[n = .. + r][k = .. + 4q + e]
    float* __restrict__ slab = a.slabs + (int64_t)split * a.slab_stride;
    float* __restrict__ out = slab + P.w_off;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nn = (cbz0 + wj * 4 + j) * 16 + r;
        if (nn >= P.N) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = (cbx0 + wi * 4 + i) * 16 + 4 * q;
            float* p = out + (int64_t)nn * P.K + kk;
            if (kk + 3 < P.K) {
                F4 v = {{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]}};
                *reinterpret_cast<F4*>(p) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (kk + e < P.K) p[e] = acc[i][j][e];
            }
        }
        if (do_bias && q == 0) slab[P.b_off + nn] = accb[j][0];
    }
}

}  // namespace hgym

#ifdef HGYM_TU_CONTRACT_OFF
#pragma clang fp contract(off)
#endif
