// hgym_gemm.hpp -- the one dense kernel of the hot path: C[M,N] = epilogue(A[M,K] * B[N,K]^T) on MFMA.
//
// Both operands are K-contiguous ("NT").  The reference's nn.Linear weights are stored (out,in) = [N][K], so the
// forward pass is NT as-is; the two backward products are made NT by keeping transposed operand copies
// (dX = dY * (W^T)^T with a W^T shadow kept by the Adam kernel; dW = dY^T * (X^T)^T with activation
// transposes), see hgym_net.hip.
//
// Data path, designed around the 64-lane wavefront and the 16x16 MFMA operand shape:
//   * an operand "fragment" is 16 rows x 64 bytes of K (32 bf16 or 16 f32): exactly what one wavefront feeds
//     to the matrix core -- lane l supplies row (l & 15), 16-byte chunk (l >> 4).
//   * global -> LDS staging moves whole fragments: lane l loads its own 16 bytes (global_load_dwordx4) and
//     stores them at LDS offset frag*1024 + l*16.  The LDS image is therefore already in MFMA operand order:
//     fragment reads are lane-linear ds_read_b128 (conflict-free by construction, no swizzle, no padding),
//     and staging writes are lane-linear ds_write_b128.
//   * bf16: one v_mfma_f32_16x16x32_bf16 consumes a fragment pair.  f32 (parity mode): the 16-byte chunk is
//     4 consecutive k's; k is a summation index, so the four v_mfma_f32_16x16x4_f32 sub-steps take element i of
//     every lane's chunk (a fixed permutation of k applied to A and B alike) -- exact fp32 FMA chains.
//   * operands are swapped (weights as the MFMA "A", activations as "B") so each lane ends up with 4
//     CONSECUTIVE output columns of one output row: the epilogue (bias, ELU, ELU' mask) runs in registers and
//     stores 8/16-byte row-major pieces.
//   * LDS double buffer, register-staged prefetch of the next K stage, one barrier per stage (128 B of K).
//   * split-K over blockIdx.z for the weight-gradient products (K = batch = 61 440): each split writes its own
//     fp32 slab (deterministic; the slabs are summed by the gradient-finalise kernel).
#pragma once
#include "hgym_common.hpp"

namespace hgym {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // one 16-byte operand chunk

constexpr int KSTAGE = 2;  // fragments of K per pipeline stage (2 x 64 B per row)

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int kFragK = 16;   // elements of K per 64-byte fragment row
};
template <> struct ElemTraits<__bf16> {
    static constexpr int kFragK = 32;
};
template <typename T> constexpr int stage_elems() { return ElemTraits<T>::kFragK * KSTAGE; }

enum { ACT_NONE = 0, ACT_ELU = 1 };

struct GemmArgs {
    const void* A;     // [rowsA][lda]  K-contiguous
    const void* B;     // [rowsB][ldb]  K-contiguous
    int64_t lda, ldb;
    int rowsA, rowsB;  // rows that may be read (row indices are clamped to these)
    int M, N, K;       // output rows, output cols (logical), contraction length (multiple of stage_elems)
    void* Ct;          // optional output in the operand type, row-major, ldct
    int64_t ldct;
    float* Cf;         // optional fp32 output, row-major, ldcf (split-K: slab z at Cf + z*slab_stride)
    int64_t ldcf;
    int64_t slab_stride;
    const float* bias;  // optional [N]
    int act;            // ACT_*
    const void* aux;    // optional [M][ldaux] operand-type matrix y = elu(z): output is multiplied by elu'(z) = y>0 ? 1 : y+1
    int64_t ldaux;
    int k_chunk;        // K range per split (multiple of stage_elems); K itself when not split
};

HG_HD float elu_f(float z) { return z > 0.0f ? z : (expf(z) - 1.0f); }

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__bf16>(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

template <typename T> __device__ __forceinline__ void mma_frag(u32x4 w, u32x4 x, f32x4& acc);
template <> __device__ __forceinline__ void mma_frag<__bf16>(u32x4 w, u32x4 x, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma_frag<float>(u32x4 w, u32x4 x, f32x4& acc) {
    const f32x4 wf = __builtin_bit_cast(f32x4, w), xf = __builtin_bit_cast(f32x4, x);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0], xf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[1], xf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[2], xf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[3], xf[3], acc, 0, 0, 0);
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_nt_kernel(const GemmArgs g) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int FA = (BM / 16) * KSTAGE;          // A fragments per stage
    constexpr int FB = (BN / 16) * KSTAGE;
    constexpr int F = FA + FB;
    constexpr int FPW = (F + NW - 1) / NW;          // fragments staged per wave
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int FRAGK = ElemTraits<T>::kFragK;
    constexpr int SE = FRAGK * KSTAGE;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of the 16x16 MFMA tile");

    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];   // 2 stages x F fragments x 1 KiB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // Block order: N-tiles fastest.  (An XCD-affine remap that put all N-tiles of one M-tile on the same XCD was
    // measured on MI355X: neutral at M = 61 440 and 2.5x slower at M = 4096 -- same-line L2 contention -- so the plain
    // order stays; the operand panels of these skinny layers are served by L2 / Infinity Cache either way.)
    const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = bz * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);
    const int nst = (kend - kbeg) / SE;

    const char* Ab = (const char*)g.A;
    const char* Bb = (const char*)g.B;
    const int lrow = lane & 15, lchunk = lane >> 4;

    // per-wave staging assignments: fragment f -> (operand, 16-row block, k sub-fragment).  When F is not a
    // multiple of the wave count the surplus slots re-stage fragment F-1 (identical bytes, same LDS slot).
    const char* src[FPW];
    int fidx[FPW];
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
        const int f = min(wave + i * NW, F - 1);
        fidx[i] = f;
        const bool isB = f >= FA;
        const int ff = isB ? f - FA : f;
        const int rb = ff / KSTAGE, ks = ff % KSTAGE;
        int row = (isB ? n0 : m0) + rb * 16 + lrow;
        row = min(row, (isB ? g.rowsB : g.rowsA) - 1);
        const int64_t ld = isB ? g.ldb : g.lda;
        src[i] = (isB ? Bb : Ab) + ((int64_t)row * ld + kbeg + ks * FRAGK) * (int64_t)sizeof(T) + lchunk * 16;
    }
    u32x4 stg[FPW];
#define HG_LOAD_STAGE(s_)                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < FPW; ++i_)                                                          \
        stg[i_] = *reinterpret_cast<const u32x4*>(src[i_] + (int64_t)(s_) * SE * (int64_t)sizeof(T));
#define HG_WRITE_STAGE(buf_)                                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < FPW; ++i_) lds[((buf_) * F + fidx[i_]) * 64 + lane] = stg[i_];

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nst > 0) {
        HG_LOAD_STAGE(0)
        HG_WRITE_STAGE(0)
    }
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        if (s + 1 < nst) { HG_LOAD_STAGE(s + 1) }
#pragma unroll
        for (int ks = 0; ks < KSTAGE; ++ks) {
            u32x4 xa[TM], wb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) xa[i] = lds[(buf * F + (wm * TM + i) * KSTAGE + ks) * 64 + lane];
#pragma unroll
            for (int j = 0; j < TN; ++j) wb[j] = lds[(buf * F + FA + (wn * TN + j) * KSTAGE + ks) * 64 + lane];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mma_frag<T>(wb[j], xa[i], acc[i][j]);
        }
        if (s + 1 < nst) { HG_WRITE_STAGE(buf ^ 1) }
        __syncthreads();
    }
#undef HG_LOAD_STAGE
#undef HG_WRITE_STAGE

    // Epilogue.  Lane holds, for output row m = ... + (lane & 15), the 4 consecutive columns n = ... + 4*(lane>>4) + r.
    // Every load the epilogue needs (bias, the ELU' operand) is issued BEFORE the first store: the outputs may alias
    // the inputs as far as the compiler can tell, so a load placed after a store would wait for a full memory round
    // trip per fragment (measured: that serialisation was half of the kernel's time).
    float* __restrict__ Cf = g.Cf ? g.Cf + (int64_t)bz * g.slab_stride : nullptr;
    T* __restrict__ Ct = (T*)g.Ct;
    const T* __restrict__ aux = (const T*)g.aux;
    struct alignas(4 * sizeof(T)) Pack { T e[4]; };
    float bias_r[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nb = n0 + wn * WTN + j * 16 + 4 * lchunk;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[j][r] = (g.bias && nb + r < g.N) ? g.bias[nb + r] : 0.0f;
    }
    const bool vec_t = (g.ldct & 3) == 0, vec_f = (g.ldcf & 3) == 0, vec_a = (g.ldaux & 3) == 0;
    if (aux) {
        Pack ax[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = min(m0 + wm * WTM + i * 16 + lrow, g.M - 1);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nb = n0 + wn * WTN + j * 16 + 4 * lchunk;
                const T* p = aux + (int64_t)m * g.ldaux + nb;
                if (nb + 3 < g.N && vec_a) {
                    ax[i][j] = *reinterpret_cast<const Pack*>(p);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) ax[i][j].e[r] = (nb + r < g.N) ? p[r] : from_f32<T>(0.0f);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y = to_f32<T>(ax[i][j].e[r]);
                    acc[i][j][r] *= (y > 0.0f) ? 1.0f : (y + 1.0f);   // elu'(z) from y = elu(z)
                }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WTM + i * 16 + lrow;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = n0 + wn * WTN + j * 16 + 4 * lchunk;
            if (nb >= g.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r] + bias_r[j][r];
                if (g.act == ACT_ELU) v[r] = elu_f(v[r]);
            }
            const bool full = nb + 3 < g.N;
            if (Cf) {
                float* p = Cf + (int64_t)m * g.ldcf + nb;
                if (full && vec_f) {
                    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb + r < g.N) p[r] = v[r];
                }
            }
            if (Ct) {
                T* p = Ct + (int64_t)m * g.ldct + nb;
                if (full && vec_t) {
                    Pack pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pk.e[r] = from_f32<T>(v[r]);
                    *reinterpret_cast<Pack*>(p) = pk;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb + r < g.N) p[r] = from_f32<T>(v[r]);
                }
            }
        }
    }
}

// Host-side dispatch over the tile configurations.
template <typename T> int32_t launch_gemm(const GemmArgs& g, int splits, hipStream_t s);

}  // namespace hgym
