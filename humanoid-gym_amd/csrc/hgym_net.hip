// hgym_net.hip -- actor/critic forward, PPO loss + hand-written backward, grad-norm clip + Adam
// (SURVEY.md §8a rows A1, A2, A8, A10, A11).  All dense products go through gemm_nt_kernel (hgym_gemm.hpp);
// everything else here is the HBM-bound glue: operand packing / transposes, the per-sample Gaussian / PPO
// arithmetic, reductions, the optimiser.
//
// Memory plan (caller-allocated workspace, zero-filled once; layout from ws_layout()):
//   per layer l of each net:  Wp  [N16][Kp]   operand-precision copy of W (rows padded to 16, K to a stage)
//                             WTp [K16][Ncp]  operand-precision copy of W^T (for dX = dY * W)
//   per layer input:          X   [maxM][Kp]  row-major activations (X0 = packed observations)
//                             XT  [K16][Mp]   their transposes (for dW = dY^T * X), update path only
//   per layer output grad:    dY  [maxM][Ncp] and dYT [N16][Mp]
//   split-K slabs             [splits][P] fp32, loss partial sums, scalar outputs
// Pad rows/columns are never written with non-zero data, so zero-fill at allocation keeps every padded
// contraction exact.
#include <stdlib.h>

#include "hgym_gemm.hpp"
#include "hgym_fused.hpp"

namespace hgym {

// hgym_update.hip
int32_t launch_mlp_fb(const FwdArgs& fb, const FbLoss& fl, bool shadow, int tiles, int nets, size_t lds, hipStream_t s);

// ------------------------------------------------------------------------------------------------ layouts
struct LayerLayout {
    int K, N;                // in, out features
    int Kp, Ncp, N16, K16;   // padded sizes (see above)
    int64_t w_off, b_off;    // offsets (floats) into the flat fp32 parameter vector
    int64_t Wp, WTp;         // byte offsets into the workspace
    int64_t X, XT, dY, dYT;  // byte offsets: layer input, its transpose, output grad, its transpose
    // fused bf16 path (hgym_fused.hpp): fragment-major operand copies
    int KBf, NBf, NBBf;      // forward k-blocks of 32, output blocks of 16, backward contraction blocks of 32
    int64_t Wf, WTf;         // byte offsets
};

struct NetLayout {
    int L;
    LayerLayout layer[HGYM_MAX_LAYERS];
    int64_t out_f32;         // [maxM][N_last] fp32 output of the last layer (update path)
    int64_t X0b, Hb[3], dZb[4];   // fused path: block-layout activations / pre-activation gradients
};

struct WsLayout {
    int es;                  // operand element size
    int SE;                  // stage elements (contraction padding unit)
    int64_t maxM, Mp;        // max batch and its contraction padding (row stride of every transposed buffer)
    int64_t P;               // total parameters
    int64_t Ps;              // slab stride (P rounded up so every slab starts 256-byte aligned)
    NetLayout net[3];        // 0 actor, 1 critic, 2 auxiliary head (optional)
    int fused_aux;           // 1: the auxiliary head also has the fused layout (its own launches of the three fused kernels)
    int nnets;               // 2 or 3
    int64_t aux_p0;          // first parameter of the auxiliary head in the flat vector (= P when absent)
    int splits;
    int64_t slabs;           // [splits][P] fp32
    int64_t partials;        // [MAX_LOSS_BLOCKS][16] fp32
    int64_t zeros;           // 4 KiB that nothing ever writes (the workspace arrives zero-filled)
    int64_t sqn;             // [SQN_BLOCKS] fp64 per-workgroup sums of sqnorm_prologue_kernel + its arrival counter (zero between launches)
    int64_t total_bytes;
    int fused;               // 1: bf16 fast path (three fused kernels) is usable for this configuration
    int64_t Mpad;            // fused path: batch padded to the 64-row tile
    int dw_splits;
};

constexpr int MAX_LOSS_BLOCKS = 8192;   // 256 samples each: minibatches up to 2 M samples
constexpr int SQN_BLOCKS = 256;         // workgroups of sqnorm_prologue_kernel
constexpr int RSN_X = 96;               // workgroups per segment of reduce_slabs_kernel

static bool fused_supported(const HgymNetConfig* c) {
    if (c->precision != HGYM_BF16 || c->actor_layers != 4 || c->critic_layers != 4) return false;
    if (getenv("HGYM_NO_FUSED")) return false;
    for (int which = 0; which < 2; ++which) {
        const int32_t* d = which == 0 ? c->actor_dims : c->critic_dims;
        const int g1 = d[1] / 128;
        if (d[1] % 128 || !(g1 == 2 || g1 == 4 || g1 == 6)) return false;
        if (d[2] % 128 || d[2] > 768 || d[3] % 128 || d[3] > 768) return false;
        if (d[4] > 16) return false;
    }
    return true;
}
// the auxiliary head through the fused kernels: same trunk constraints, head up to 96 columns (three 32-wide contraction blocks)
static bool fused_aux_supported(const HgymNetConfig* c) {
    if (c->aux_layers != 4 || getenv("HGYM_NO_FUSED_AUX")) return false;
    const int32_t* d = c->aux_dims;
    if (d[1] != 512) return false;         // the wide-head instantiation of the forward exists for this first width only
    if (d[2] % 128 || d[2] > 768 || d[3] % 128 || d[3] > 768) return false;
    return d[4] > 16 && d[4] <= 96;
}
constexpr int MAX_SPLITS = 32;

static int32_t ws_layout(const HgymNetConfig* c, WsLayout* w) {
    HG_REQUIRE(c, HGYM_E_BADARG, "null net config");
    HG_REQUIRE(c->precision == HGYM_F32 || c->precision == HGYM_BF16, HGYM_E_BADARG, "precision=%d", c->precision);
    HG_REQUIRE(c->actor_layers >= 1 && c->actor_layers <= HGYM_MAX_LAYERS && c->critic_layers >= 1 &&
                   c->critic_layers <= HGYM_MAX_LAYERS, HGYM_E_SHAPE, "layer counts %d/%d", c->actor_layers, c->critic_layers);
    HG_REQUIRE(c->max_batch > 0, HGYM_E_SHAPE, "max_batch=%d", c->max_batch);
    HG_REQUIRE(c->actor_dims[0] == c->num_obs && c->critic_dims[0] == c->num_priv &&
                   c->actor_dims[c->actor_layers] == c->num_actions && c->critic_dims[c->critic_layers] == 1,
               HGYM_E_SHAPE, "layer dims inconsistent with num_obs/num_priv/num_actions");
    HG_REQUIRE(c->num_actions >= 1 && c->num_actions <= 12, HGYM_E_UNSUPPORTED, "num_actions=%d (loss kernel reduces <=12 per-action sums)",
               c->num_actions);
    memset(w, 0, sizeof(*w));
    w->es = c->precision == HGYM_F32 ? 4 : 2;
    w->SE = c->precision == HGYM_F32 ? stage_elems<float>() : stage_elems<__bf16>();
    w->maxM = c->max_batch;
    w->Mp = round_up(c->max_batch, w->SE);
    w->fused = fused_supported(c) ? 1 : 0;
    w->Mpad = round_up(c->max_batch, 64);
    int64_t off = 0, poff = c->num_actions;  // std first (state_dict order)
    auto take = [&](int64_t bytes) {
        const int64_t o = off;
        off += round_up(bytes, 256);
        return o;
    };
    HG_REQUIRE(c->aux_layers >= 0 && c->aux_layers <= HGYM_MAX_LAYERS, HGYM_E_SHAPE, "aux_layers=%d", c->aux_layers);
    if (c->aux_layers > 0)
        HG_REQUIRE(c->aux_dims[0] == c->num_obs && c->aux_target_offset >= 0 &&
                       c->aux_target_offset + c->aux_dims[c->aux_layers] <= c->num_priv && c->actor_layers + c->critic_layers + c->aux_layers <= 16,
                   HGYM_E_SHAPE, "auxiliary head: input must be num_obs, targets must lie inside the privileged row");
    w->nnets = c->aux_layers > 0 ? 3 : 2;
    w->fused_aux = (w->fused && c->aux_layers > 0 && fused_aux_supported(c)) ? 1 : 0;
    w->aux_p0 = -1;
    for (int which = 0; which < w->nnets; ++which) {
        NetLayout& n = w->net[which];
        n.L = which == 0 ? c->actor_layers : (which == 1 ? c->critic_layers : c->aux_layers);
        const int32_t* dims = which == 0 ? c->actor_dims : (which == 1 ? c->critic_dims : c->aux_dims);
        const bool fused_net = w->fused && (which < 2 || w->fused_aux);
        if (which == 2) w->aux_p0 = poff;
        for (int l = 0; l < n.L; ++l) {
            LayerLayout& y = n.layer[l];
            y.K = dims[l];
            y.N = dims[l + 1];
            HG_REQUIRE(y.K > 0 && y.N > 0, HGYM_E_SHAPE, "non-positive layer dim");
            y.Kp = (int)round_up(y.K, w->SE);
            y.Ncp = (int)round_up(y.N, w->SE);
            y.N16 = (int)round_up(y.N, 16);
            y.K16 = (int)round_up(y.K, 16);
            y.w_off = poff;
            poff += (int64_t)y.N * y.K;
            y.b_off = poff;
            poff += y.N;
            if (fused_net) {
                y.KBf = (int)round_up(y.K, l == 0 ? FUSED_CHUNK : 32) / 32;
                y.NBf = y.N16 / 16;
                y.NBBf = (int)round_up(y.N, 32) / 32;
                y.Wf = take((int64_t)y.NBf * y.KBf * 1024);
                y.WTf = take((int64_t)y.K16 / 16 * y.NBBf * 1024);
            } else {
                y.Wp = take((int64_t)y.N16 * y.Kp * w->es);
                y.WTp = take((int64_t)y.K16 * y.Ncp * w->es);
                y.X = take(w->maxM * y.Kp * w->es);
                y.XT = take((int64_t)y.K16 * w->Mp * w->es);
                y.dY = take(w->maxM * y.Ncp * w->es);
                y.dYT = take((int64_t)y.N16 * w->Mp * w->es);
            }
        }
        n.out_f32 = take(w->maxM * (int64_t)dims[n.L] * 4);
        if (fused_net) {
            n.X0b = take(w->Mpad * (int64_t)n.layer[0].KBf * 32 * 2);
            for (int l = 0; l < 3; ++l) {
                n.Hb[l] = take(w->Mpad * (int64_t)n.layer[l].N * 2);
                n.dZb[l] = take(w->Mpad * (int64_t)n.layer[l].N * 2);
            }
            n.dZb[3] = take(w->Mpad * 32 * n.layer[3].NBBf * 2);
        }
    }
    w->P = poff;
    if (w->aux_p0 < 0) w->aux_p0 = poff;
    w->Ps = round_up(poff, 64);
    w->splits = MAX_SPLITS;
    w->dw_splits = 8;        // batch splits of the dW contraction, one per XCD (16 measured 3 % slower: twice the slab traffic)
    w->slabs = take((int64_t)w->splits * w->Ps * 4);
    w->partials = take((int64_t)MAX_LOSS_BLOCKS * LOSS_PARTIALS * 4);
    w->zeros = take(4096);
    w->sqn = take((SQN_BLOCKS + 2) * 8);
    w->total_bytes = off;
    return HGYM_OK;
}

// ------------------------------------------------------------------------------------------------ small kernels
__global__ __launch_bounds__(1024) void fin_only_kernel(const FinArgs f) { fin_block(f, threadIdx.x, blockDim.x); }

// dst[m][k] = T(src[row(m)][k]) for k < K; row(m) = idx ? idx[m] : m.  Pad columns are left untouched (zero).
template <typename T>
__global__ __launch_bounds__(256) void pack_rows_kernel(int M, int K, const float* __restrict__ src, int64_t ld_src,
                                                        const int64_t* __restrict__ idx, T* __restrict__ dst, int64_t ld_dst) {
    const int64_t total = (int64_t)M * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / K), k = (int)(i - (int64_t)m * K);
        const int64_t r = idx ? idx[m] : m;
        dst[(int64_t)m * ld_dst + k] = from_f32<T>(src[r * ld_src + k]);
    }
}

// out[c][m] = in[m][c] for c < C, m < M; out[c][m] = 0 for M <= m < Mp  (Mp = contraction padding of this call)
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(int M, int Mp, int C, const T* __restrict__ in, int64_t ld_in,
                                                        T* __restrict__ out, int64_t ld_out) {
    __shared__ T tile[64][66];
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 4 row groups
    for (int r = ty; r < 64; r += 4) {
        const int m = m0 + r, c = c0 + tx;
        tile[r][tx] = (m < M && c < C) ? in[(int64_t)m * ld_in + c] : from_f32<T>(0.0f);
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, m = m0 + tx;
        if (c < C && m < Mp) out[(int64_t)c * ld_out + m] = tile[tx][r];
    }
}

// out[r] += sum_m in[r][m] (bias gradients from dY^T).  grid = (column chunks, rows); every lane issues four
// independent 16-byte loads, one fp32 atomic per workgroup.  `out` must be zero on entry.
template <typename T>
__global__ __launch_bounds__(256) void rowsum_kernel(int Mp, const T* __restrict__ in, int64_t ld_in, float* __restrict__ out) {
    constexpr int V = 16 / sizeof(T);
    __shared__ float red[4];
    const T* row = in + (int64_t)blockIdx.y * ld_in;
    const int base = blockIdx.x * (256 * 4 * V);
    float s = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int m = base + (u * 256 + threadIdx.x) * V;
        if (m < Mp) {   // Mp is a multiple of V and the pad columns are zero
            const u32x4 raw = *reinterpret_cast<const u32x4*>(row + m);
            const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
            for (int k = 0; k < V; ++k) s += to_f32<T>(e[k]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&out[blockIdx.y], red[0] + red[1] + red[2] + red[3]);
}

// Auxiliary head loss: L = coef * mean_b mean_j (y[b][j] - t[b][j])^2 with t = priv[idx[b]][off + j].  Writes
// dL/dy in operand precision, row-major (dY, leading dimension ld) and transposed (dYT [No16][Mp]) for the generic backward,
// zero in the contraction padding rows B..Bp, and adds the (unweighted) minibatch MSE to opt[10].
template <typename T>
__global__ __launch_bounds__(256) void aux_mse_kernel(int B, int Bp, int No, const float* __restrict__ y, const float* __restrict__ priv,
                                                      int64_t ldp, int off, const int64_t* __restrict__ idx, float coef,
                                                      T* __restrict__ dY, int64_t ld, T* __restrict__ dYT, int64_t ldt,
                                                      double* __restrict__ opt) {
    __shared__ float red[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float se = 0.0f;
    if (i < B) {
        const float* t = priv + idx[i] * ldp + off;
        const float g = 2.0f * coef / ((float)B * (float)No);
        for (int j = 0; j < No; ++j) {
            const float d = y[(int64_t)i * No + j] - t[j];
            se += d * d;
            const T gv = from_f32<T>(g * d);
            dY[(int64_t)i * ld + j] = gv;
            dYT[(int64_t)j * ldt + i] = gv;
        }
    } else if (i < Bp) {
        for (int j = 0; j < No; ++j) dYT[(int64_t)j * ldt + i] = from_f32<T>(0.0f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) se += __shfl_down(se, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = se;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&opt[10], (double)(red[0] + red[1] + red[2] + red[3]) / ((double)B * (double)No));
}

// PPO.act epilogue (actor_critic.py:111-120): a = mu + sigma*z, logp = sum log N(a; mu, sigma); sigma = std.
__global__ __launch_bounds__(256) void act_sample_kernel(int M, int A, const float* __restrict__ mu, const float* __restrict__ std_,
                                                         const float* __restrict__ z, uint64_t seed, const int64_t* __restrict__ step,
                                                         float* __restrict__ actions, float* __restrict__ sigma, float* __restrict__ logp) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    RngKey rk = {(uint32_t)seed, (uint32_t)(seed >> 32), 0u, 0u};
    if (!z) {
        const int64_t s = step ? step[0] : 0;
        rk.s0 = (uint32_t)s;
        rk.s1 = (uint32_t)(s >> 32);
    }
    float lp = 0.0f;
    float zn[16];
    if (!z) normals_block<4>(rk, (uint32_t)m, SLOT_POLICY, zn);
#pragma unroll
    for (int j = 0; j < 16; ++j) {      // fully unrolled so zn[] stays in registers
        if (j >= A) break;
        const float mj = mu[(int64_t)m * A + j];
        const float sg = mj * 0.0f + std_[j];            // actor_critic.py:113 (propagates NaN like the reference)
        const float zz = z ? z[(int64_t)m * A + j] : zn[j];
        const float a = mj + sg * zz;
        actions[(int64_t)m * A + j] = a;
        sigma[(int64_t)m * A + j] = sg;
        const float d = a - mj;
        lp += -(d * d) / (2.0f * sg * sg) - logf(sg) - 0.9189385332046727f;
    }
    logp[m] = lp;
}

// One thread per sample of the minibatch: ppo.py:128-168 forward scalars + the hand-written backward of `loss`
// w.r.t. mu, std and V (oracle/ppo_oracle.py:ppo_loss_and_grads is the executable specification).
struct LossArgs {
    HgymBatch b;
    int A;                       // num_actions
    const float* mu;             // [B][A] current policy mean (fp32 output of the actor)
    const float* val;            // [B]    current value
    const float* std_;           // [A]
    float clip, value_coef, entropy_coef;
    int fused;                   // 1: dmu / dval are block-layout tiles of 32 columns (hgym_fused.hpp); 0: row-major + transposes
    void* dmu;  int64_t ld_dmu;  // [B][Ncp] operand type
    void* dmuT; int64_t ld_t;    // [A16][Mp]
    void* dval; int64_t ld_dval; // [B][Ncp]
    void* dvalT;                 // [16][Mp]
    int Bp;                      // contraction padding of B for this call
    float* partials;             // [gridDim.x][32]: surrogate, value loss, entropy, kl, dstd[12], sum dmu[12], sum dval, pad
};

template <typename T>
__global__ __launch_bounds__(256) void ppo_loss_kernel(const LossArgs a) {
    __shared__ float red[4][LOSS_PARTIALS];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int B = a.b.B, A = a.A;
    const float invB = 1.0f / (float)B;
    float acc[LOSS_PARTIALS];
#pragma unroll
    for (int k = 0; k < LOSS_PARTIALS; ++k) acc[k] = 0.0f;
    T* dmu = (T*)a.dmu;
    T* dmuT = (T*)a.dmuT;
    T* dval = (T*)a.dval;
    T* dvalT = (T*)a.dvalT;
    // fused layout: sample i lives in row block i>>4, row i&15 of a 2-block (32-column) tile; only block 0 is ever non-zero
    char* fmu = (char*)a.dmu + ((int64_t)(i >> 4) * 2) * 512 + (i & 15) * 32;
    char* fval = (char*)a.dval + ((int64_t)(i >> 4) * 2) * 512 + (i & 15) * 32;
    if (i < B) {
        const int64_t r = a.b.idx[i];
        const float adv = a.b.advantages[r], ret = a.b.returns[r], vold = a.b.values[r], lpold = a.b.logp[r];
        const float v = a.val[i];
        float lp = 0.0f, ent = 0.0f, kl = 0.0f;
        float diff[16], sg[16];
        // the gathered rows (actions, old mu, old sigma: A floats each, at a random row r) and this sample's mu row.  For the
        // XBot-L width (A = 12: 48-byte rows, 16-byte aligned) they are fetched as three 16-byte loads each: a 4-byte load
        // per element costs the texture path one cache line per lane PER ELEMENT (the rows are scattered), a vector load one
        // per lane per 4 elements -- the kernel was bound by exactly that (31 us for 61 440 samples).
        float rowm[16], rowa[16], rowmo[16], rowso[16];
        if (A == 12) {
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const float4 qm = reinterpret_cast<const float4*>(a.mu + (int64_t)i * 12)[v];
                const float4 qa = reinterpret_cast<const float4*>(a.b.actions + r * 12)[v];
                const float4 qo = reinterpret_cast<const float4*>(a.b.mu + r * 12)[v];
                const float4 qs = reinterpret_cast<const float4*>(a.b.sigma + r * 12)[v];
                rowm[4 * v] = qm.x; rowm[4 * v + 1] = qm.y; rowm[4 * v + 2] = qm.z; rowm[4 * v + 3] = qm.w;
                rowa[4 * v] = qa.x; rowa[4 * v + 1] = qa.y; rowa[4 * v + 2] = qa.z; rowa[4 * v + 3] = qa.w;
                rowmo[4 * v] = qo.x; rowmo[4 * v + 1] = qo.y; rowmo[4 * v + 2] = qo.z; rowmo[4 * v + 3] = qo.w;
                rowso[4 * v] = qs.x; rowso[4 * v + 1] = qs.y; rowso[4 * v + 2] = qs.z; rowso[4 * v + 3] = qs.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j >= A) continue;
                rowm[j] = a.mu[(int64_t)i * A + j];
                rowa[j] = a.b.actions[r * A + j];
                rowmo[j] = a.b.mu[r * A + j];
                rowso[j] = a.b.sigma[r * A + j];
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            diff[j] = 0.0f;
            sg[j] = 1.0f;
            if (j >= A) continue;
            const float m = rowm[j];
            const float s = m * 0.0f + a.std_[j];
            const float act = rowa[j];
            const float mo = rowmo[j], so = rowso[j];
            const float d = act - m;
            diff[j] = d;
            sg[j] = s;
            lp += -(d * d) / (2.0f * s * s) - logf(s) - 0.9189385332046727f;
            ent += 0.5f + 0.9189385332046727f + logf(s);
            kl += logf(s / so + 1.e-5f) + (so * so + (mo - m) * (mo - m)) / (2.0f * (s * s)) - 0.5f;
        }
        const float ratio = expf(lp - lpold);
        const float s1 = -adv * ratio;
        const float s2 = -adv * clampf(ratio, 1.0f - a.clip, 1.0f + a.clip);
        const float surr = fmaxf(s1, s2);
        const float vc = vold + clampf(v - vold, -a.clip, a.clip);
        const float l1 = (v - ret) * (v - ret), l2 = (vc - ret) * (vc - ret);
        const float vl = fmaxf(l1, l2);
        // backward (torch.max splits ties evenly between its operands; clamp passes gradient on the closed range)
        const float in_range = (ratio >= 1.0f - a.clip && ratio <= 1.0f + a.clip) ? 1.0f : 0.0f;
        const float w1 = s1 > s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f);
        const float d_lp = (-adv) * (w1 + (1.0f - w1) * in_range) * invB * ratio;
        const float v_in = ((v - vold) >= -a.clip && (v - vold) <= a.clip) ? 1.0f : 0.0f;
        const float u1 = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.0f);
        const float d_v = a.value_coef * invB * (u1 * 2.0f * (v - ret) + (1.0f - u1) * 2.0f * (vc - ret) * v_in);
        float gm[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            gm[j] = 0.0f;
            if (j >= A) continue;
            const float s = sg[j], d = diff[j];
            const float g_mu = d_lp * d / (s * s);
            const float g_sg = d_lp * (d * d / (s * s * s) - 1.0f / s) - (a.entropy_coef * invB) / s;
            gm[j] = g_mu;
            if (j < 12) {
                acc[4 + j] = g_sg;
                acc[16 + j] = g_mu;
            }
            if (!a.fused) {
                dmu[(int64_t)i * a.ld_dmu + j] = from_f32<T>(g_mu);
                dmuT[(int64_t)j * a.ld_t + i] = from_f32<T>(g_mu);
            }
        }
        if (a.fused) {
            if constexpr (sizeof(T) == 2) {
                u32x4 lo, hi;
                const u32x2 p0 = pack_bf16x4(gm[0], gm[1], gm[2], gm[3]), p1 = pack_bf16x4(gm[4], gm[5], gm[6], gm[7]);
                const u32x2 p2 = pack_bf16x4(gm[8], gm[9], gm[10], gm[11]), p3 = pack_bf16x4(gm[12], gm[13], gm[14], gm[15]);
                lo = (u32x4){p0[0], p0[1], p1[0], p1[1]};
                hi = (u32x4){p2[0], p2[1], p3[0], p3[1]};
                reinterpret_cast<u32x4*>(fmu)[0] = lo;
                reinterpret_cast<u32x4*>(fmu)[1] = hi;
                const u32x2 pv = pack_bf16x4(d_v, 0.0f, 0.0f, 0.0f);
                reinterpret_cast<u32x4*>(fval)[0] = (u32x4){pv[0], 0u, 0u, 0u};
                reinterpret_cast<u32x4*>(fval)[1] = (u32x4){0u, 0u, 0u, 0u};
            }
        } else {
            dval[(int64_t)i * a.ld_dval] = from_f32<T>(d_v);
            dvalT[i] = from_f32<T>(d_v);
        }
        acc[0] = surr;
        acc[1] = vl;
        acc[2] = ent;
        acc[3] = kl;
        acc[28] = d_v;
    } else if (i < a.Bp) {   // zero the contraction padding of the gradients
        if (a.fused) {
            reinterpret_cast<u32x4*>(fmu)[0] = (u32x4){0u, 0u, 0u, 0u};
            reinterpret_cast<u32x4*>(fmu)[1] = (u32x4){0u, 0u, 0u, 0u};
            reinterpret_cast<u32x4*>(fval)[0] = (u32x4){0u, 0u, 0u, 0u};
            reinterpret_cast<u32x4*>(fval)[1] = (u32x4){0u, 0u, 0u, 0u};
        } else {
            for (int j = 0; j < A; ++j) dmuT[(int64_t)j * a.ld_t + i] = from_f32<T>(0.0f);
            dvalT[i] = from_f32<T>(0.0f);
        }
    }
#pragma unroll
    for (int k = 0; k < LOSS_PARTIALS; ++k) {
        float s = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < LOSS_PARTIALS) a.partials[(int64_t)blockIdx.x * LOSS_PARTIALS + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// opt_state: [0] lr [1] adam step [2] kl sum [3] surrogate sum [4] value-loss sum [5] entropy sum [6] grad norm
//            [7] minibatches accumulated [8] last minibatch mean KL [9] grad sq-norm accumulator
// grads_bmu / grads_bv (fused path only): gradients of the two head biases = column sums of the head gradients.
__global__ __launch_bounds__(512) void ppo_scalars_kernel(const ScalArgs a) { ppo_scalars_block(a, threadIdx.x, blockDim.x); }

// gradient finalise: sum the split-K slabs of every weight matrix into the flat gradient vector
struct Segment {
    int64_t off;     // offset in the flat parameter vector
    int rows, cols;  // N, K  (bias / std: rows = count, cols = 1, no shadow)
    int splits;      // slabs to sum (0: gradient already final in `grads`)
    void* Wp;        // operand-precision shadow [N16][ldw] or null
    int64_t ldw;
    void* WTp;       // operand-precision transposed shadow [K16][ldwt] or null
    int64_t ldwt;
    void* Wf;        // fused path: forward fragments [NB][KB][64][8] or null
    void* WTf;       // fused path: backward fragments [K/16][NBB][64][8]
    int KB, NBB;
};
struct SegTable {
    int n;
    Segment s[2 * HGYM_MAX_LAYERS * 2 + 1];
};

// Also accumulates the squared norm of the finished gradient into opt[9] (zeroed by ppo_scalars_kernel earlier in the same
// hgym_ppo_grad): with one rank that IS the norm clip_grad_norm_ needs, and hgym_ppo_apply skips its own pass over the
// gradient.  (Segments whose gradient is already final -- splits == 0 -- are only read.)
// The norm stays on fp64 atomics in arrival order (1 632 workgroups, one non-returning atomicAdd each) -- made order-INDEPENDENT by rounding every
// partial to a common quantum first (below): exact additions commute.  Round 6 first built the order-fixed forms -- per-workgroup partials + last
// arriver with a release fence per workgroup: 65 us instead of 11 (every fence writes the XCD's L2 back while the others fill it with gradient
// lines); write-through partials + acknowledged store + returning counter atomic, on one and on two levels: 26-29 us (three to six dependent
// memory-side round trips at the tail of an 11 us kernel) -- and dropped them (profiles/r06_update_graph_and_norm_order.txt).  With several ranks
// the norm is sqnorm_prologue_kernel's, which is order-fixed (256 workgroups, one launch).
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const SegTable tab, int64_t P, const float* __restrict__ slabs,
                                                           float* __restrict__ grads, double* __restrict__ opt) {
    __shared__ double red[4];
    const Segment& sg = tab.s[blockIdx.y];
    const int64_t n = (int64_t)sg.rows * sg.cols;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double sq = 0.0;
    if (sg.splits == 0) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            const float g = grads[sg.off + i];
            sq += (double)g * (double)g;
        }
    } else {
        // 16-byte body between a scalar head and tail: a segment need not start on a 16-byte boundary of the flat vector (the
        // critic's one-element head bias puts every segment of the auxiliary net at offset = 1 mod 4)
        const bool vec = (P & 3) == 0 && (((uintptr_t)grads | (uintptr_t)slabs) & 15) == 0;      // P = slab stride
        const int64_t head = vec ? ((4 - (sg.off & 3)) & 3) < n ? ((4 - (sg.off & 3)) & 3) : n : n;
        const int64_t n4 = (n - head) >> 2, tail0 = head + 4 * n4;
        const int64_t base = sg.off + head;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sg.splits == 8) {      // the usual split count: all eight loads in flight, then the same ascending sum
                float4 v[8];
#pragma unroll
                for (int z = 0; z < 8; ++z) v[z] = *reinterpret_cast<const float4*>(slabs + (int64_t)z * P + base + 4 * i);
#pragma unroll
                for (int z = 0; z < 8; ++z) { acc.x += v[z].x; acc.y += v[z].y; acc.z += v[z].z; acc.w += v[z].w; }
            } else {
                for (int z = 0; z < sg.splits; ++z) {
                    const float4 v = *reinterpret_cast<const float4*>(slabs + (int64_t)z * P + base + 4 * i);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            }
            *reinterpret_cast<float4*>(grads + base + 4 * i) = acc;
            sq += (double)acc.x * (double)acc.x + (double)acc.y * (double)acc.y + (double)acc.z * (double)acc.z + (double)acc.w * (double)acc.w;
        }
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < head + (n - tail0); i += stride) {
            const int64_t e = i < head ? i : tail0 + (i - head);
            float s = 0.0f;
            for (int z = 0; z < sg.splits; ++z) s += slabs[(int64_t)z * P + sg.off + e];
            grads[sg.off + e] = s;
            sq += (double)s * (double)s;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        // Pre-rounded to a common quantum (2^-46): every partial, and therefore every intermediate value of opt[9], is an integer multiple of
        // it, so while the total stays below 2^53 quanta = 128 (a gradient norm of 11.3, eleven times the clip threshold) the fp64 additions
        // are EXACT -- and exact additions commute: the atomics may arrive in any order, the sum has the same bits.  (Above 128 the additions
        // round again and the last bits depend on the order, as they always did; the step is then clipped by more than 11x and sees `total`
        // as a float.)  Cost of the rounding: at most 1 632 x 2^-47 = 1.2e-11 absolute on the squared norm.
        double t = red[0] + red[1] + red[2] + red[3];
        t = __builtin_rint(t * 0x1p46) * 0x1p-46;
        if (t != 0.0) atomicAdd(&opt[9], t);
    }
}

// (inv_w: 1 / world_size -- with several ranks the gradient vector holds the all-reduced SUM and the mean is formed in sqnorm_prologue_kernel and in
// adam_kernel: an fp32 product, what `grad.mul_(1 / world)` would have stored; 1.0f for one rank, which is exact.)

// ppo.py:140-148 (adaptive-KL learning rate, python-double arithmetic) + Adam step counter + the squared norm Adam will clip with: `sq` when
// this call determined it itself (sqnorm_prologue_kernel), 0 when a norm pass follows, untouched when the gradient call left it (have_sq < 0)
__device__ __forceinline__ void apply_prologue(const HgymPPOConfig& p, const float* __restrict__ kl_slot, float inv_w, double* __restrict__ opt,
                                               int have_sq, double sq) {
    if (p.world_size > 1) opt[8] = (double)(kl_slot[0] * inv_w);     // mean over ranks of the minibatch KL: same LR branch everywhere
    if (opt[13] == opt[1] && opt[1] > 0.0) {     // the gradient call already took this step's prologue (marker: ppo_scalars_block) and the
        if (have_sq >= 0) opt[9] = have_sq ? sq : 0.0;                // caller applies with another configuration: not a second time
        return;
    }
    if (p.adaptive_lr) {
        const double kl = opt[8];
        double lr = opt[0];
        if (kl > (double)p.desired_kl * 2.0) lr = fmax(p.lr_min, lr / 1.5);
        else if (kl < (double)p.desired_kl / 2.0 && kl > 0.0) lr = fmin(p.lr_max, lr * 1.5);
        opt[0] = lr;
    }
    const double t = opt[1] + 1.0;
    opt[1] = t;
    // Adam's bias corrections, once per step instead of two double-precision pow() per thread of adam_kernel (1.1 M threads): the
    // same double arithmetic, rounded to the floats the update uses
    const bool cached = opt[13] == t;             // ppo_scalars_block left beta^t of this step (same pow(), same arguments): kernel 8.1 -> 4.8 us
    const double bc1 = 1.0 - (cached ? opt[14] : pow((double)p.beta1, t)), bc2 = 1.0 - (cached ? opt[15] : pow((double)p.beta2, t));
    opt[11] = (double)(float)(opt[0] / bc1);      // step size
    opt[12] = (double)(float)sqrt(bc2);
    opt[13] = t;                                  // marker: this step's prologue is done (adam_kernel clears it)
    if (have_sq >= 0) opt[9] = have_sq ? sq : 0.0;
}
__global__ void apply_prologue_kernel(const HgymPPOConfig p, const float* __restrict__ kl_slot, float inv_w, double* __restrict__ opt) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // (a norm pass follows this launch only when the caller could not reuse the gradient call's: several ranks / foreign gradients)
    apply_prologue(p, kl_slot, inv_w, opt, (p.world_size > 1 || !p.grad_norm_ready) ? 0 : -1, 0.0);
}

// The two launches hgym_ppo_apply used to start with when the norm of the gradient call cannot be reused (several ranks: the norm is of the
// rank MEAN; foreign gradients) -- the prologue and the squared-norm pass -- as one: every workgroup leaves its fp64 partial sum, the last one
// to arrive adds the SQN_BLOCKS partials in a fixed order (no atomics on the sum: the same bits on every rank and in every run) and takes the
// prologue with the total.  part[SQN_BLOCKS] is the arrival counter (an unsigned int, zero between launches: the last workgroup resets it).
__global__ __launch_bounds__(256) void sqnorm_prologue_kernel(int64_t P, const float* __restrict__ g, float inv_w, const HgymPPOConfig p,
                                                              const float* __restrict__ kl_slot, double* __restrict__ opt,
                                                              double* __restrict__ part) {
    __shared__ double red[4];
    __shared__ unsigned int s_last;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * inv_w;
        s += (double)gi * (double)gi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    unsigned int* counter = reinterpret_cast<unsigned int*>(part + SQN_BLOCKS);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        __threadfence();
        s_last = atomicAdd(counter, 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    double v = threadIdx.x < gridDim.x ? reinterpret_cast<const volatile double*>(part)[threadIdx.x] : 0.0;      // (behind the fence: the peers' partials)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        *counter = 0u;
        apply_prologue(p, kl_slot, inv_w, opt, 1, red[0] + red[1] + red[2] + red[3]);
    }
}

// writes one master weight into every compute-precision operand copy of its layer
template <typename T>
__device__ __forceinline__ void write_shadows(const Segment& sg, int r, int c, float w) {
    if (sg.Wp) {
        ((T*)sg.Wp)[(int64_t)r * sg.ldw + c] = from_f32<T>(w);
        if (sg.WTp) ((T*)sg.WTp)[(int64_t)c * sg.ldwt + r] = from_f32<T>(w);
    }
    if (sg.Wf) {   // r = n (output feature), c = k (input feature); see hgym_fused.hpp for the fragment order
        ((T*)sg.Wf)[((((int64_t)(r >> 4) * sg.KB + (c >> 5)) * 64) + ((c & 31) >> 3) * 16 + (r & 15)) * 8 + (c & 7)] = from_f32<T>(w);
        ((T*)sg.WTf)[((((int64_t)(c >> 4) * sg.NBB + (r >> 5)) * 64) + ((r & 31) >> 3) * 16 + (c & 15)) * 8 + (r & 7)] = from_f32<T>(w);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void adam_kernel(const SegTable tab, const HgymPPOConfig p, float* __restrict__ params,
                                                   float* __restrict__ grads, float* __restrict__ m_, float* __restrict__ v_,
                                                   float inv_w, double* __restrict__ opt) {
    const Segment& sg = tab.s[blockIdx.y];
    const int64_t n = (int64_t)sg.rows * sg.cols;
    // nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1 (fp32 tensor arithmetic)
    const float total = (float)sqrt(opt[9]);
    float coef = p.max_grad_norm / (total + 1e-6f);
    coef = fminf(coef, 1.0f);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        opt[6] = (double)total;
        opt[13] = -1.0;                           // the step whose prologue was pending is being applied (nobody reads opt[13] in this launch)
    }
    const float step_size = (float)opt[11], sqrt_bc2 = (float)opt[12];      // apply_prologue_kernel: lr / (1 - beta1^t), sqrt(1 - beta2^t)
    auto adam1 = [&](int64_t q) -> float {            // one parameter: clip, moments, step; returns the new weight
        const float g = (grads[q] * inv_w) * coef;
        grads[q] = g;
        const float m = m_[q] * p.beta1 + (1.0f - p.beta1) * g;
        const float v = v_[q] * p.beta2 + (1.0f - p.beta2) * (g * g);
        m_[q] = m;
        v_[q] = v;
        const float denom = sqrtf(v) / sqrt_bc2 + p.adam_eps;
        const float w = params[q] + (-step_size * m) / denom;
        params[q] = w;
        return w;
    };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float w = adam1(sg.off + i);
        if (sg.Wp || sg.Wf) {
            const int r = (int)(i / sg.cols), c = (int)(i - (int64_t)r * sg.cols);
            write_shadows<T>(sg, r, c, w);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sync_shadow_kernel(const SegTable tab, const float* __restrict__ params) {
    const Segment& sg = tab.s[blockIdx.y];
    if (!sg.Wp && !sg.Wf) return;
    const int64_t n = (int64_t)sg.rows * sg.cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / sg.cols), c = (int)(i - (int64_t)r * sg.cols);
        write_shadows<T>(sg, r, c, params[sg.off + i]);
    }
}

// ------------------------------------------------------------------------------------------------ GEMM dispatch
template <typename T, int BM, int BN, int WMs, int WNs>
static void launch_cfg(const GemmArgs& g, int splits, hipStream_t s) {
    constexpr int F = ((BM / 16) + (BN / 16)) * KSTAGE;
    constexpr size_t lds = (size_t)2 * F * 1024;
    // > 64 KiB of LDS needs the opt-in, per device (a failure surfaces as the launch error the caller checks)
    (void)ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_nt_kernel<T, BM, BN, WMs, WNs>), lds, "gemm_nt_kernel");
    dim3 grid(ceil_div(g.N, BN), ceil_div(g.M, BM), splits);
    hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN, WMs, WNs>), grid, dim3(WMs * WNs * 64), lds, s, g);
}

template <typename T>
int32_t launch_gemm(const GemmArgs& g0, int splits, hipStream_t s) {
    GemmArgs g = g0;
    constexpr int SE = stage_elems<T>();
    HG_REQUIRE(g.K % SE == 0 && g.K > 0, HGYM_E_SHAPE, "gemm K=%d not a multiple of %d", g.K, SE);
    HG_REQUIRE(g.lda * (int64_t)sizeof(T) % 16 == 0 && g.ldb * (int64_t)sizeof(T) % 16 == 0, HGYM_E_SHAPE, "gemm operand rows must be 16-byte aligned");
    if (splits < 1) splits = 1;
    int stages = g.K / SE;
    if (splits > stages) splits = stages;
    const int per = ceil_div(stages, splits);
    splits = ceil_div(stages, per);
    g.k_chunk = per * SE;
    prof_begin(HGYM_PROF_GEMM, s);
    const int64_t t128 = (int64_t)ceil_div(g.M, 128) * ceil_div(g.N, 128) * splits;
    if (g.N <= 16) launch_cfg<T, 128, 16, 4, 1>(g, splits, s);
    else if (g.M <= 16) launch_cfg<T, 16, 128, 1, 4>(g, splits, s);
    else if (t128 >= 192) launch_cfg<T, 128, 128, 2, 2>(g, splits, s);
    else launch_cfg<T, 64, 64, 2, 2>(g, splits, s);
    prof_end(HGYM_PROF_GEMM, s, 2.0 * (double)g.M * (double)g.N * (double)g.K);   // padded K: the flops the MFMAs execute
    HG_CHECK_LAUNCH("gemm_nt_kernel");
    return splits;
}

// ------------------------------------------------------------------------------------------------ orchestration
template <typename T>
struct NetRunner {
    const HgymNetConfig& cfg;
    const HgymNet& net;
    WsLayout w;
    hipStream_t s;
    char* ws;

    template <typename U> U* at(int64_t off) const { return reinterpret_cast<U*>(ws + off); }

    SegTable segments(bool with_slabs) const {
        SegTable t;
        memset(&t, 0, sizeof(t));
        Segment& sd = t.s[t.n++];
        sd.off = 0;
        sd.rows = cfg.num_actions;
        sd.cols = 1;
        for (int which = 0; which < w.nnets; ++which)
            for (int l = 0; l < w.net[which].L; ++l) {
                const LayerLayout& y = w.net[which].layer[l];
                const bool fused_net = w.fused && (which < 2 || w.fused_aux);
                Segment& a = t.s[t.n++];
                a.off = y.w_off;
                a.rows = y.N;
                a.cols = y.K;
                if (fused_net) {
                    a.splits = with_slabs ? w.dw_splits : 0;
                    a.Wf = ws + y.Wf;
                    a.WTf = ws + y.WTf;
                    a.KB = y.KBf;
                    a.NBB = y.NBBf;
                } else {
                    a.splits = with_slabs ? split_count(y) : 0;
                    a.Wp = ws + y.Wp;
                    a.ldw = y.Kp;
                    a.WTp = ws + y.WTp;
                    a.ldwt = y.Ncp;
                }
                Segment& b = t.s[t.n++];
                b.off = y.b_off;
                b.rows = y.N;
                b.cols = 1;
                // bias gradients that come from the slabs: hidden layers (the dW kernel's column sums of dZ); the actor / critic
                // head biases are written by ppo_scalars_kernel, the auxiliary head's is a column sum like the others
                if (fused_net && with_slabs && (l < w.net[which].L - 1 || which == 2)) b.splits = w.dw_splits;
            }
        return t;
    }

    int cur_Mp = 0;   // contraction padding of the current batch
    int split_count(const LayerLayout& y) const {
        // enough workgroups to fill 256 CUs: tiles(N x K) x splits ~ 512
        const int tiles = ceil_div(y.N, y.N <= 16 ? 16 : 128) * ceil_div(y.K, 128);
        int sp = ceil_div(512, tiles);
        if (sp > w.splits) sp = w.splits;
        const int stages = cur_Mp / w.SE;
        if (sp > stages) sp = stages;
        if (sp < 1) sp = 1;
        const int per = ceil_div(stages, sp);
        return ceil_div(stages, per);
    }

    // ------------------------------------------------------------------ fused bf16 path (hgym_fused.hpp)
    struct SampleOut {
        const float* z; uint64_t seed; const int64_t* step; float* actions; float* sigma; float* logp;
    };

    FusedNet fused_net(int which, const float* x, int64_t ldx, float* out, int64_t ldo) const {
        FusedNet f;
        memset(&f, 0, sizeof(f));
        const NetLayout& n = w.net[which];
        for (int l = 0; l < 4; ++l) {
            const LayerLayout& y = n.layer[l];
            FusedLayer& d = f.layer[l];
            d.Wf = at<u32x4>(y.Wf);
            d.WTf = at<u32x4>(y.WTf);
            d.bias = net.params + y.b_off;
            d.K = y.K;
            d.N = y.N;
            d.KB = y.KBf;
            d.NB = y.NBf;
            d.NBB = y.NBBf;
        }
        f.x = x;
        f.ldx = ldx;
        f.X0 = at<__bf16>(n.X0b);
        for (int l = 0; l < 3; ++l) f.H[l] = at<__bf16>(n.Hb[l]);
        for (int l = 0; l < 4; ++l) f.dZ[l] = at<__bf16>(n.dZb[l]);
        f.out = out;
        f.ldo = ldo;
        return f;
    }

    template <int BM, int NW, int D, bool FIN>
    int32_t launch_fwd_k(const FwdArgs& a, int nets) {
        size_t lds = 0;
        for (int i = 0; i < nets; ++i) {
            const FusedNet& n = a.net[a.net0 + i];
            lds = std::max(lds, (size_t)fused_lds_p(n, BM) + (size_t)fused_lds_q(n, BM) + (size_t)fused_lds_bias(n));
        }
        const int32_t rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&mlp_fwd_kernel<BM, NW, D, FIN>), lds, "mlp_fwd_kernel");
        if (rc_lds) return rc_lds;
        FwdArgs b = a;
        b.nets = nets;
        const int rows = nets + (FIN ? 1 : 0);          // + the finaliser's grid row (its workgroup 0 works, the rest exit)
        b.dbg = phase_buffer((int64_t)ceil_div(a.M, BM) * rows);
        hipLaunchKernelGGL((mlp_fwd_kernel<BM, NW, D, FIN>), dim3(ceil_div(a.M, BM), rows), dim3(NW * 64), lds, s, b);
        HG_CHECK_LAUNCH("mlp_fwd_kernel");
        return HGYM_OK;
    }
    template <int BM, int NW, int D>
    int32_t launch_fwd(const FwdArgs& a, int nets) {
        return a.fin.N > 0 ? launch_fwd_k<BM, NW, D, true>(a, nets) : launch_fwd_k<BM, NW, D, false>(a, nets);
    }

    // bf16 shadow rows of net `which`'s input (hgym_net_shadow_ld): the first layer's padded width
    int64_t shadow_ld(int which) const { return (int64_t)w.net[which].layer[0].KBf * 32; }

    // sh (policy launches, no row gather): net 0 / 1 also store the bf16 of their input rows as rows of sh->obs / sh->priv
    FwdArgs make_fwd_args(int first, int nets, int M, const float* const xs[3], const int64_t ldxs[3], const int64_t* idx, float* const outs[3],
                          const int64_t ldos[3], bool train, const SampleOut* smp, const FinArgs* fin, const HgymObsShadow* sh = nullptr) const {
        FwdArgs a;
        memset(&a, 0, sizeof(a));
        for (int i = first; i < first + nets; ++i) a.net[i] = fused_net(i, xs[i], ldxs[i], outs[i], ldos[i]);
        if (sh && !idx && !train) {
            if (first == 0 && sh->obs) { a.net[0].xs = (__bf16*)sh->obs; a.net[0].ldxs = sh->ld_obs; }
            if (first <= 1 && first + nets > 1 && sh->priv) { a.net[1].xs = (__bf16*)sh->priv; a.net[1].ldxs = sh->ld_priv; }
        }
        a.net0 = first;
        a.M = M;
        a.idx = idx;
        if (fin) a.fin = *fin;
        a.train = train ? 3 : 0;
        a.A = cfg.num_actions;
        a.std_ = net.params;
        if (smp) {
            a.sample = 1;
            a.z = smp->z;
            a.k0 = (uint32_t)smp->seed;
            a.k1 = (uint32_t)(smp->seed >> 32);
            a.step = smp->step;
            a.actions = smp->actions;
            a.sigma = smp->sigma;
            a.logp = smp->logp;
        }
        return a;
    }

    // forward of `nets` networks starting at `first` in ONE launch; xs/outs indexed by net id
    int32_t fused_forward(int first, int nets, int M, const float* const xs[3], const int64_t ldxs[3], const int64_t* idx, float* const outs[3],
                          const int64_t ldos[3], bool train, const SampleOut* smp, const FinArgs* fin = nullptr, const HgymObsShadow* sh = nullptr) {
        HG_REQUIRE(M > 0 && M <= w.maxM, HGYM_E_SHAPE, "batch %d exceeds max_batch %lld", M, (long long)w.maxM);
        if (sh) {
            HG_REQUIRE((!sh->obs || (sh->ld_obs >= shadow_ld(0) && sh->ld_obs % 8 == 0 && ((uintptr_t)sh->obs & 15) == 0)) &&
                           (!sh->priv || (sh->ld_priv >= shadow_ld(1) && sh->ld_priv % 8 == 0 && ((uintptr_t)sh->priv & 15) == 0)),
                       HGYM_E_SHAPE, "observation shadow: leading dimensions %lld / %lld (need >= %lld / %lld, multiples of 8, 16-byte aligned rows)",
                       (long long)sh->ld_obs, (long long)sh->ld_priv, (long long)shadow_ld(0), (long long)shadow_ld(1));
        }
        const FwdArgs a = make_fwd_args(first, nets, M, xs, ldxs, idx, outs, ldos, train, smp, fin, sh);
        const int pcls = train ? HGYM_PROF_MLP_FWD : HGYM_PROF_POLICY;
        prof_begin(pcls, s);
        // 32-row tiles x 8 waves (weight ring depth 4) while they fit the chip in one round (one workgroup per CU: 2 * M / 32
        // <= CUs, i.e. 4096 envs), 64-row tiles x 16 waves (depth 2) for the update and for larger rollouts; measured
        // alternatives (64 rows x 8 waves, 32-row tiles for the update, other depths) were equal or slower
        const int cus = device_cus() > 0 ? device_cus() : 256;
        // (32-row tiles two per CU at 8192 rows: collection 5.32 vs 3.82 ms, profiles/r03_envs8192_policy_tiles_ab.txt)
        const int32_t rc = (train || nets * ceil_div(M, 32) > cus) ? launch_fwd<64, 16, 2>(a, nets) : launch_fwd<32, 8, 4>(a, nets);
        double fl = 0.0;
        for (int i = first; i < first + nets; ++i)
            for (int l = 0; l < 4; ++l) fl += 2.0 * (double)M * w.net[i].layer[l].N * w.net[i].layer[l].K;   // algorithmic (unpadded) flops
        prof_end(pcls, s, fl);
        return rc;
    }

    // all weight (and hidden bias) gradients of nets [first, first + count): one launch, split-K slabs
    // (the auxiliary head, net 2, reads the ACTOR's bf16 copy of the gathered observation rows as its first-layer operand -- same
    // rows, same columns -- and every bias gradient of it is a column sum of dZ: its loss has no per-tile partial sums for them)
    // gb (mlp_fb_kernel<XB16> ran): the first-layer operands were never copied -- those products gather their rows by gb->idx from the
    // bf16 shadows
    int32_t fused_dw(int first, int count, int B, const ScalArgs* sc = nullptr, const HgymBatch* gb = nullptr) {
        const int Bp = (int)round_up(B, 64);
        DwArgs d;
        memset(&d, 0, sizeof(d));
        d.B = B;
        int tile = 0;
        double fl = 0.0;
        for (int i = first; i < first + count; ++i)
            for (int l = 0; l < 4; ++l) {
                const NetLayout& n = w.net[i];
                const LayerLayout& y = n.layer[l];
                HG_REQUIRE(d.np < DW_MAX_PRODUCTS, HGYM_E_UNSUPPORTED, "too many weight-gradient products in one launch");
                DwProduct& p = d.p[d.np++];
                p.Z = at<__bf16>(n.dZb[l]);
                p.CBz = l < 3 ? y.N / 16 : 2 * y.NBBf;
                p.X = l == 0 ? at<__bf16>(w.net[i == 2 ? 0 : i].X0b) : at<__bf16>(n.Hb[l - 1]);
                p.CBx = l == 0 ? 2 * y.KBf : y.K / 16;
                if (l == 0 && gb) {
                    p.X = (const __bf16*)(i == 1 ? gb->priv_bf16 : gb->obs_bf16);
                    p.gidx = gb->idx;
                    p.ldg = shadow_ld(i == 1 ? 1 : 0);
                    p.CBx = (int)(p.ldg / 16);
                }
                p.N = y.N;
                p.K = y.K;
                p.w_off = y.w_off;
                p.b_off = (l < 3 || i == 2) ? y.b_off : -1;
                p.tiles_n = ceil_div(y.N, DW_TILE_N);
                p.tiles_k = ceil_div(y.K, DW_TILE_K);
                p.tile0 = tile;
                tile += p.tiles_n * p.tiles_k;
                fl += 2.0 * (double)B * y.N * y.K;
            }
        d.total_tiles = tile;
        d.splits = w.dw_splits;
        d.steps_total = Bp / 32;
        d.steps_per_split = ceil_div(d.steps_total, w.dw_splits);
        d.slabs = at<float>(w.slabs);
        d.slab_stride = w.Ps;
        d.zeros = at<char>(w.zeros);
        int blocks = tile * (int)round_up(w.dw_splits, 8);
        d.scal_bid = -1;
        if (sc) {
            d.sc = *sc;
            d.scal_bid = blocks++;
        }
        const int32_t rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&dw_kernel_rs<3>), DW_LDS_STAGES * DW_STAGE_BYTES, "dw_kernel_rs");
        if (rc_lds != HGYM_OK) return rc_lds;
        prof_begin(HGYM_PROF_DW, s);
        hipLaunchKernelGGL(dw_kernel_rs<3>, dim3(blocks), dim3(DW_THREADS), DW_LDS_STAGES * DW_STAGE_BYTES, s, d);
        prof_end(HGYM_PROF_DW, s, fl);
        HG_CHECK_LAUNCH("dw_kernel_rs");
        return HGYM_OK;
    }

    // slab reduction of the segments whose parameters lie in [lo, hi) of the flat vector
    int32_t reduce_range(int64_t lo, int64_t hi) {
        const SegTable all = segments(true);
        SegTable tab;
        memset(&tab, 0, sizeof(tab));
        int64_t elems = 0;
        for (int i = 0; i < all.n; ++i)
            if (all.s[i].off >= lo && all.s[i].off < hi) {
                tab.s[tab.n++] = all.s[i];
                elems += (int64_t)all.s[i].rows * all.s[i].cols;
            }
        if (tab.n == 0) return HGYM_OK;
        prof_begin(HGYM_PROF_REDUCE, s);
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(RSN_X, tab.n), dim3(256), 0, s, tab, w.Ps, at<float>(w.slabs), net.grads, net.opt_state);
        prof_end(HGYM_PROF_REDUCE, s, (double)elems * 4.0 * (w.dw_splits + 1));
        HG_CHECK_LAUNCH("reduce_slabs_kernel");
        return HGYM_OK;
    }

    // part < 0: the whole minibatch gradient.  part 0 / 1: the two halves of hgym_ppo_grad_part -- 0 leaves std's and the actor's
    // gradient final, 1 the critic's, the auxiliary head's and the KL slot.
    int32_t fused_grad(const HgymPPOConfig& ppo, const HgymBatch& b, int part = -1) {
        const int B = b.B, A = cfg.num_actions;
        const int64_t critic_off = w.net[1].layer[0].w_off;
        const bool aux_fb = w.nnets > 2 && w.fused_aux;      // the auxiliary head as a third grid row of the same launches
        // bf16 shadows of the storage rows (HgymBatch.obs_bf16 / priv_bf16): gather 2 B per element, keep no operand copy
        static const bool no_shadow = getenv("HGYM_NO_SHADOW") != nullptr;       // A/B experiments only
        const bool shadow = b.obs_bf16 && b.priv_bf16 && !no_shadow && b.num_rows > 0 &&
                            b.num_rows * shadow_ld(0) * 2 < ((int64_t)1 << 32) && b.num_rows * shadow_ld(1) * 2 < ((int64_t)1 << 32);
        const HgymBatch* gb = shadow ? &b : nullptr;
        // In two parts (hgym_ppo_grad_part): part 0 runs everything up to and including ALL weight-gradient products and sums the
        // slabs of [std | actor]; part 1 only sums the rest.  (Until round 3 part 1 also launched the critic's products on their own:
        // two launches of 144 and 112 workgroups on 256 CUs, each as long as the full one -- 2 x 142 us against 178 us.)
        if (part == 1) return reduce_range(critic_off, w.P);
        float* mu = at<float>(w.net[0].out_f32);
        float* val = at<float>(w.net[1].out_f32);
        const float* xs[3] = {b.obs, b.priv, b.obs};
        const int64_t ldxs[3] = {cfg.num_obs, cfg.num_priv, cfg.num_obs};
        float* outs[3] = {mu, val, nullptr};
        const int64_t ldos[3] = {A, 1, 0};
        const int Bp = (int)round_up(B, 64);
        const int nets = aux_fb ? 3 : 2;
        // (128-row tiles on eight 256-register wavefronts -- round 4's mlp_fb2_kernel -- measured equal to slightly slower:
        // profiles/r04_fb2_128row_tiles_negative_result.txt; its source is profiles/r04_fb2_128row_kernel.patch; round 6 rebuilt the idea on role-specialised wavefronts at 64 and 128 rows: profiles/r06_fb3_role_specialised_wavefronts.txt.)
        const int tiles = Bp / 64;
        HG_REQUIRE(tiles <= MAX_LOSS_BLOCKS, HGYM_E_UNSUPPORTED, "minibatch %d too large for the loss partial buffer", B);
        HG_REQUIRE(B <= w.maxM, HGYM_E_SHAPE, "minibatch %d exceeds max_batch %lld", B, (long long)w.maxM);
        int32_t rc = HGYM_OK;
        {   // forward + PPO loss + dZ chain of both nets: ONE launch (hgym_fused.hpp: mlp_fb_kernel)
            FwdArgs fa = make_fwd_args(0, nets, B, xs, ldxs, b.idx, outs, ldos, true, nullptr, nullptr);
            fa.net[2].X0 = nullptr;       // the head's first-layer operand for the weight gradient is the actor's copy (fused_dw)
            if (shadow) {
                HG_REQUIRE((((uintptr_t)b.obs_bf16 | (uintptr_t)b.priv_bf16) & 15) == 0, HGYM_E_BADARG, "observation shadows must be 16-byte aligned");
                for (int i = 0; i < nets; ++i) {
                    fa.net[i].xb = (const __bf16*)(i == 1 ? b.priv_bf16 : b.obs_bf16);
                    fa.net[i].ldxb = shadow_ld(i == 1 ? 1 : 0);
                    fa.net[i].X0 = nullptr;
                }
            }
            FbLoss fl;
            memset(&fl, 0, sizeof(fl));
            fl.actions = b.actions;
            fl.old_mu = b.mu;
            fl.old_sigma = b.sigma;
            fl.values = b.values;
            fl.advantages = b.advantages;
            fl.returns = b.returns;
            fl.logp = b.logp;
            fl.clip = ppo.clip_param;
            fl.value_coef = ppo.value_loss_coef;
            fl.entropy_coef = ppo.entropy_coef;
            fl.partials = at<float>(w.partials);
            fl.aux_target = b.priv;
            fl.aux_ldt = cfg.num_priv;
            fl.aux_off = cfg.aux_target_offset;
            fl.aux_coef = ppo.aux_coef;
            size_t lds = 0;
            for (int i = 0; i < nets; ++i)
                lds = std::max(lds, (size_t)fused_lds_p(fa.net[i], 64) + (size_t)fused_lds_q(fa.net[i], 64) + (size_t)fused_lds_bias(fa.net[i]) +
                                        (size_t)fb_lds_extra(fa.net[i]));
            HG_REQUIRE(lds <= 160 * 1024, HGYM_E_UNSUPPORTED, "mlp_fb_kernel needs %zu bytes of LDS", lds);
            FwdArgs fb = fa;
            fb.nets = nets;
            fb.dbg = phase_buffer((int64_t)tiles * nets);
            prof_begin(HGYM_PROF_MLP_FWD, s);
            const int32_t rc_fb = launch_mlp_fb(fb, fl, shadow, tiles, nets, lds, s);      // (hgym_update.hip: the kernel's own code object)
            if (rc_fb) return rc_fb;
            double flops = 0.0;
            for (int i = 0; i < nets; ++i) {
                for (int l = 0; l < 4; ++l) flops += 2.0 * (double)B * w.net[i].layer[l].N * w.net[i].layer[l].K;
                for (int l = 1; l < 4; ++l) flops += 2.0 * (double)B * w.net[i].layer[l].K * w.net[i].layer[l].N;
            }
            prof_end(HGYM_PROF_MLP_FWD, s, flops);
            HG_CHECK_LAUNCH("mlp_fb_kernel");
        }
        // the minibatch's loss scalars (per-tile partials -> opt_state, std / head-bias gradients, KL slot): one extra workgroup of
        // the weight-gradient launch that follows anyway (it needs nothing but the partials mlp_fb_kernel has just written)
        const ScalArgs sc = {tiles, B, A, aux_fb ? w.net[2].layer[3].N : 0, at<float>(w.partials), net.grads,
                             net.grads + w.net[0].layer[3].b_off, net.grads + w.net[1].layer[3].b_off, net.grads + w.P, net.opt_state,
                             (double)ppo.beta1, (double)ppo.beta2, prologue_in_grad(ppo) ? 1 : 0, ppo.adaptive_lr, ppo.desired_kl, ppo.lr_min,
                             ppo.lr_max, 1};
        rc = fused_dw(0, nets, B, &sc, gb);
        if (rc) return rc;
        if (w.nnets > 2 && !aux_fb) {
            const int32_t rca = aux_grad(ppo, b);
            if (rca) return rca;
        }
        return part == 0 ? reduce_range(0, critic_off) : reduce_range(0, w.P);
    }

    int32_t act(int M, const float* obs, const float* priv, const float* z, uint64_t seed, const int64_t* step, float* actions, float* mu,
                float* sigma, float* logp, float* values, const FinArgs* fin = nullptr, const HgymObsShadow* sh = nullptr) {
        if (w.fused) {
            const float* xs[3] = {obs, priv, nullptr};
            const int64_t ldxs[3] = {cfg.num_obs, cfg.num_priv, 0};
            float* outs[3] = {mu, values, nullptr};
            const int64_t ldos[3] = {cfg.num_actions, 1, 0};
            const SampleOut smp = {z, seed, step, actions, sigma, logp};
            return fused_forward(0, 2, M, xs, ldxs, nullptr, outs, ldos, false, &smp, fin, sh);
        }
        HG_REQUIRE(!sh || (!sh->obs && !sh->priv), HGYM_E_UNSUPPORTED, "the observation shadow exists on the fused bf16 path only (hgym_net_shadow_ld = 0 here)");
        if (fin) {     // generic path: the postponed finaliser as its own (tiny) launch
            hipLaunchKernelGGL(fin_only_kernel, dim3(1), dim3(1024), 0, s, *fin);
            HG_CHECK_LAUNCH("fin_only_kernel");
        }
        int32_t rc = forward(0, M, obs, cfg.num_obs, nullptr, mu, cfg.num_actions, false);
        if (rc) return rc;
        rc = forward(1, M, priv, cfg.num_priv, nullptr, values, 1, false);
        if (rc) return rc;
        hipLaunchKernelGGL(act_sample_kernel, dim3(ceil_div(M, 256)), dim3(256), 0, s, M, cfg.num_actions, mu, net.params, z, seed, step,
                           actions, sigma, logp);
        HG_CHECK_LAUNCH("act_sample_kernel");
        return HGYM_OK;
    }

    // the critic over M rows in pieces of at most max_batch (a multiple of 64 rows each: whole tiles); fused path: 64-row tiles, the
    // shadow's priv rows written by the tiles that read them
    int32_t critic_values(int64_t M, const float* priv, float* values, const HgymObsShadow* sh) {
        const int64_t piece = std::max<int64_t>(64, w.maxM / 64 * 64);
        for (int64_t m0 = 0; m0 < M; m0 += piece) {
            const int m = (int)std::min<int64_t>(piece, M - m0);
            const float* x = priv + m0 * cfg.num_priv;
            int32_t rc;
            if (w.fused) {
                const float* xs[3] = {nullptr, x, nullptr};
                const int64_t ldxs[3] = {0, cfg.num_priv, 0};
                float* outs[3] = {nullptr, values + m0, nullptr};
                const int64_t ldos[3] = {0, 1, 0};
                HgymObsShadow s1 = {nullptr, 0, nullptr, 0};
                if (sh && sh->priv) {
                    s1.priv = (char*)sh->priv + m0 * sh->ld_priv * 2;
                    s1.ld_priv = sh->ld_priv;
                }
                rc = fused_forward(1, 1, m, xs, ldxs, nullptr, outs, ldos, false, nullptr, nullptr, (sh && sh->priv) ? &s1 : nullptr);
            } else {
                rc = forward(1, m, x, cfg.num_priv, nullptr, values + m0, 1, false);
            }
            if (rc) return rc;
        }
        return HGYM_OK;
    }

    int32_t forward(int which, int M, const float* x, int64_t ldx, const int64_t* idx, float* y_out, int64_t ld_out, bool train) {
        if (w.fused && (which < 2 || w.fused_aux)) {
            const float* xs[3] = {x, x, x};
            const int64_t ldxs[3] = {ldx, ldx, ldx};
            float* outs[3] = {y_out, y_out, y_out};
            const int64_t ldos[3] = {ld_out, ld_out, ld_out};
            return fused_forward(which, 1, M, xs, ldxs, idx, outs, ldos, train, nullptr);
        }
        const NetLayout& n = w.net[which];
        HG_REQUIRE(M > 0 && M <= w.maxM, HGYM_E_SHAPE, "batch %d exceeds max_batch %lld", M, (long long)w.maxM);
        const LayerLayout& l0 = n.layer[0];
        const int64_t total = (int64_t)M * l0.K;
        hipLaunchKernelGGL((pack_rows_kernel<T>), dim3((int)std::min<int64_t>(ceil_div(total, 256), 4096)), dim3(256), 0, s, M, l0.K, x,
                           ldx, idx, at<T>(l0.X), (int64_t)l0.Kp);
        HG_CHECK_LAUNCH("pack_rows_kernel");
        const int Mp = (int)round_up(M, w.SE);
        for (int l = 0; l < n.L; ++l) {
            const LayerLayout& y = n.layer[l];
            const bool last = l == n.L - 1;
            if (train) {
                hipLaunchKernelGGL((transpose_kernel<T>), dim3(ceil_div(Mp, 64), ceil_div(y.K, 64)), dim3(256), 0, s, M, Mp, y.K,
                                   at<T>(y.X), (int64_t)y.Kp, at<T>(y.XT), w.Mp);
                HG_CHECK_LAUNCH("transpose_kernel");
            }
            GemmArgs g;
            memset(&g, 0, sizeof(g));
            g.A = at<T>(y.X);
            g.lda = y.Kp;
            g.rowsA = M;
            g.B = at<T>(y.Wp);
            g.ldb = y.Kp;
            g.rowsB = y.N16;
            g.M = M;
            g.N = y.N;
            g.K = y.Kp;
            g.bias = net.params + y.b_off;
            if (last) {
                g.Cf = y_out;
                g.ldcf = ld_out;
            } else {
                g.act = ACT_ELU;
                g.Ct = at<T>(n.layer[l + 1].X);
                g.ldct = n.layer[l + 1].Kp;
            }
            const int32_t rc = launch_gemm<T>(g, 1, s);
            if (rc < 0) return rc;
        }
        return HGYM_OK;
    }

    // backward of one net given dY / dYT of its last layer already in the workspace
    int32_t backward(int which, int M) {
        const NetLayout& n = w.net[which];
        const int Mp = (int)round_up(M, w.SE);
        cur_Mp = Mp;
        for (int l = n.L - 1; l >= 0; --l) {
            const LayerLayout& y = n.layer[l];
            {   // dW_l[N][K] = sum_m dY[m][n] * X[m][k]  (contraction over the batch, split-K slabs)
                GemmArgs g;
                memset(&g, 0, sizeof(g));
                g.A = at<T>(y.dYT);
                g.lda = w.Mp;
                g.rowsA = y.N16;
                g.B = at<T>(y.XT);
                g.ldb = w.Mp;
                g.rowsB = y.K16;
                g.M = y.N;
                g.N = y.K;
                g.K = Mp;
                g.Cf = at<float>(w.slabs) + y.w_off;
                g.ldcf = y.K;
                g.slab_stride = w.Ps;
                const int want = split_count(y);
                const int32_t got = launch_gemm<T>(g, want, s);
                if (got < 0) return got;
                HG_REQUIRE(got == want, HGYM_E_LAUNCH, "split-K mismatch %d vs %d", got, want);
                hipLaunchKernelGGL((rowsum_kernel<T>), dim3(ceil_div(Mp, 256 * 4 * (16 / (int)sizeof(T))), y.N), dim3(256), 0, s, Mp,
                                   at<T>(y.dYT), w.Mp, net.grads + y.b_off);
                HG_CHECK_LAUNCH("rowsum_kernel");
            }
            if (l > 0) {   // dX = (dY * W) .* elu'(X_l)  -> dY of layer l-1
                const LayerLayout& p = n.layer[l - 1];
                GemmArgs g;
                memset(&g, 0, sizeof(g));
                g.A = at<T>(y.dY);
                g.lda = y.Ncp;
                g.rowsA = M;
                g.B = at<T>(y.WTp);
                g.ldb = y.Ncp;
                g.rowsB = y.K16;
                g.M = M;
                g.N = y.K;
                g.K = y.Ncp;
                g.aux = at<T>(y.X);
                g.ldaux = y.Kp;
                g.Ct = at<T>(p.dY);
                g.ldct = p.Ncp;
                const int32_t rc = launch_gemm<T>(g, 1, s);
                if (rc < 0) return rc;
                hipLaunchKernelGGL((transpose_kernel<T>), dim3(ceil_div(Mp, 64), ceil_div(p.N, 64)), dim3(256), 0, s, M, Mp, p.N,
                                   at<T>(p.dY), (int64_t)p.Ncp, at<T>(p.dYT), w.Mp);
                HG_CHECK_LAUNCH("transpose_kernel");
            }
        }
        return HGYM_OK;
    }

    // Auxiliary (denoising) head, HgymNetConfig::aux_*: forward on the gathered observation rows, MSE against the target
    // columns of the gathered privileged rows, backward.  Leaves its weight gradients in the split-K slabs and its bias
    // gradients in net.grads; the caller's slab reduction finishes them together with everything else.
    int32_t aux_grad(const HgymPPOConfig& ppo, const HgymBatch& b) {
        const NetLayout& n = w.net[2];
        const LayerLayout& last = n.layer[n.L - 1];
        const int B = b.B, No = last.N;
        float* y = at<float>(n.out_f32);
        if (hipMemsetAsync(net.grads + w.aux_p0, 0, (size_t)(w.P - w.aux_p0) * sizeof(float), s) != hipSuccess)
            HG_FAIL(HGYM_E_LAUNCH, "memset of the auxiliary gradients failed");
        int32_t rc = forward(2, B, b.obs, cfg.num_obs, b.idx, y, No, true);
        if (rc) return rc;
        const int Bp = (int)round_up(B, w.SE);
        hipLaunchKernelGGL((aux_mse_kernel<T>), dim3(ceil_div(Bp, 256)), dim3(256), 0, s, B, Bp, No, y, b.priv, (int64_t)cfg.num_priv,
                           cfg.aux_target_offset, b.idx, ppo.aux_coef, at<T>(last.dY), (int64_t)last.Ncp, at<T>(last.dYT), w.Mp,
                           net.opt_state);
        HG_CHECK_LAUNCH("aux_mse_kernel");
        cur_Mp = Bp;
        return backward(2, B);
    }

    int32_t grad(const HgymPPOConfig& ppo, const HgymBatch& b, int part = -1) {
        const int B = b.B, A = cfg.num_actions;
        HG_REQUIRE(B > 0 && B <= w.maxM, HGYM_E_SHAPE, "minibatch %d exceeds max_batch %lld", B, (long long)w.maxM);
        if (w.fused) return fused_grad(ppo, b, part);
        if (part == 1) return HGYM_OK;      // layer-by-layer path: part 0 does everything (the caller's first bucket goes out complete, just later)
        float* mu = at<float>(w.net[0].out_f32);
        float* val = at<float>(w.net[1].out_f32);
        if (hipMemsetAsync(net.grads, 0, (size_t)w.P * sizeof(float), s) != hipSuccess) HG_FAIL(HGYM_E_LAUNCH, "memset of grads failed");
        int32_t rc = forward(0, B, b.obs, cfg.num_obs, b.idx, mu, A, true);
        if (rc) return rc;
        rc = forward(1, B, b.priv, cfg.num_priv, b.idx, val, 1, true);
        if (rc) return rc;
        const int Bp = (int)round_up(B, w.SE);
        const int nblocks = ceil_div(Bp, 256);
        HG_REQUIRE(nblocks <= MAX_LOSS_BLOCKS, HGYM_E_UNSUPPORTED, "minibatch %d too large for the loss partial buffer", B);
        const LayerLayout& la = w.net[0].layer[w.net[0].L - 1];
        const LayerLayout& lc = w.net[1].layer[w.net[1].L - 1];
        LossArgs a;
        memset(&a, 0, sizeof(a));
        a.b = b;
        a.A = A;
        a.mu = mu;
        a.val = val;
        a.std_ = net.params;
        a.clip = ppo.clip_param;
        a.value_coef = ppo.value_loss_coef;
        a.entropy_coef = ppo.entropy_coef;
        a.dmu = at<T>(la.dY);
        a.ld_dmu = la.Ncp;
        a.dmuT = at<T>(la.dYT);
        a.ld_t = w.Mp;
        a.dval = at<T>(lc.dY);
        a.ld_dval = lc.Ncp;
        a.dvalT = at<T>(lc.dYT);
        a.Bp = Bp;
        a.partials = at<float>(w.partials);
        prof_begin(HGYM_PROF_LOSS, s);
        hipLaunchKernelGGL((ppo_loss_kernel<T>), dim3(nblocks), dim3(256), 0, s, a);
        prof_end(HGYM_PROF_LOSS, s, (double)B * (4.0 * (5 * A + 6) + (double)sizeof(T) * (2 * A + 2)));
        HG_CHECK_LAUNCH("ppo_loss_kernel");
        const ScalArgs sc = {nblocks, B, A, 0, at<float>(w.partials), net.grads, nullptr, nullptr, net.grads + w.P, net.opt_state,
                             (double)ppo.beta1, (double)ppo.beta2, 0, 0, 0.0f, 0.0, 0.0, 1};
        hipLaunchKernelGGL(ppo_scalars_kernel, dim3(1), dim3(512), 0, s, sc);
        HG_CHECK_LAUNCH("ppo_scalars_kernel");
        cur_Mp = Bp;
        rc = backward(0, B);
        if (rc) return rc;
        rc = backward(1, B);
        if (rc) return rc;
        if (w.nnets > 2) {
            rc = aux_grad(ppo, b);
            if (rc) return rc;
        }
        const SegTable tab = segments(true);
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(RSN_X, tab.n), dim3(256), 0, s, tab, w.Ps, at<float>(w.slabs), net.grads, net.opt_state);
        HG_CHECK_LAUNCH("reduce_slabs_kernel");
        return HGYM_OK;
    }

    // The fused path on one rank with grad_norm_ready: the loss-scalar workgroup that rides in the weight-gradient launch has already taken
    // the learning-rate decision and prepared Adam's step scalars (ScalArgs::do_prologue); hgym_ppo_apply then starts with Adam.
    bool prologue_in_grad(const HgymPPOConfig& ppo) const { return w.fused && ppo.world_size <= 1 && ppo.grad_norm_ready != 0; }

    int32_t apply(const HgymPPOConfig& ppo) {
        prof_begin(HGYM_PROF_APPLY, s);
        const float inv_w = ppo.world_size > 1 ? (float)(1.0 / (double)ppo.world_size) : 1.0f;
        if (ppo.world_size > 1 || !ppo.grad_norm_ready)        // else: reduce_slabs_kernel left the squared norm in opt[9]
            hipLaunchKernelGGL(sqnorm_prologue_kernel, dim3(SQN_BLOCKS), dim3(256), 0, s, w.P, net.grads, inv_w, ppo, net.grads + w.P, net.opt_state,
                               at<double>(w.sqn));
        else if (!prologue_in_grad(ppo))
            hipLaunchKernelGGL(apply_prologue_kernel, dim3(1), dim3(64), 0, s, ppo, net.grads + w.P, inv_w, net.opt_state);
        const SegTable tab = segments(false);
        // 256 workgroups per segment: the two first-layer matrices hold 57 % of the parameters, and 64 workgroups (a quarter of
        // the CUs) walked them in 22 dependent load -> store rounds per lane (30.7 us; 18.1 us with 256, 20.8 us with 512)
        hipLaunchKernelGGL((adam_kernel<T>), dim3(256, tab.n), dim3(256), 0, s, tab, ppo, net.params, net.grads, net.adam_m, net.adam_v,
                           inv_w, net.opt_state);
        prof_end(HGYM_PROF_APPLY, s, (double)w.P * 36.0);
        HG_CHECK_LAUNCH("adam_kernel");
        return HGYM_OK;
    }

    int32_t sync_shadow() {
        const SegTable tab = segments(false);
        hipLaunchKernelGGL((sync_shadow_kernel<T>), dim3(64, tab.n), dim3(256), 0, s, tab, net.params);
        HG_CHECK_LAUNCH("sync_shadow_kernel");
        return HGYM_OK;
    }
};

static int32_t check_net(const HgymNetConfig* cfg, const HgymNet* net, WsLayout* w) {
    HG_REQUIRE(cfg && net, HGYM_E_BADARG, "null net config / net");
    const int32_t rc = ws_layout(cfg, w);
    if (rc) return rc;
    HG_REQUIRE(net->params && net->workspace, HGYM_E_BADARG, "null params / workspace");
    HG_REQUIRE(((uintptr_t)net->workspace & 255) == 0, HGYM_E_BADARG, "workspace must be 256-byte aligned");
    return HGYM_OK;
}

// For hgym_rollout.hip (the fused policy + env step lives in a translation unit of its own: it also contains the env
// arithmetic, which is built with -ffp-contract=off): the FwdArgs record of one PPO.act launch over the actor and the critic
// with 32-row tiles, and the dynamic LDS those tiles need.  Fails unless the fused bf16 path serves this configuration with
// XBot-L's first hidden widths (actor 512, critic 768: the instantiations the rollout kernel carries).
int32_t rollout_fwd_args(const HgymNetConfig* cfg, const HgymNet* net, int M, const float* obs, const float* priv, uint64_t seed,
                         const int64_t* step, float* actions, float* mu, float* sigma, float* logp, float* values, FwdArgs* out,
                         size_t* lds_bytes, const HgymObsShadow* sh) {
    WsLayout w;
    const int32_t rc = check_net(cfg, net, &w);
    if (rc) return rc;
    HG_REQUIRE(cfg->precision == HGYM_BF16 && w.fused, HGYM_E_UNSUPPORTED, "the fused rollout step needs the bf16 fused path");
    HG_REQUIRE(M > 0 && M <= w.maxM, HGYM_E_SHAPE, "batch %d exceeds max_batch %lld", M, (long long)w.maxM);
    HG_REQUIRE(cfg->actor_dims[1] == 512 && cfg->critic_dims[1] == 768, HGYM_E_UNSUPPORTED,
               "fused rollout step: first hidden widths 512 / 768 (XBot-L) only, not %d / %d", cfg->actor_dims[1], cfg->critic_dims[1]);
    NetRunner<__bf16> R{*cfg, *net, w, nullptr, (char*)net->workspace};
    const float* xs[3] = {obs, priv, nullptr};
    const int64_t ldxs[3] = {cfg->num_obs, cfg->num_priv, 0};
    float* outs[3] = {mu, values, nullptr};
    const int64_t ldos[3] = {cfg->num_actions, 1, 0};
    const NetRunner<__bf16>::SampleOut smp = {nullptr, seed, step, actions, sigma, logp};
    if (sh)
        HG_REQUIRE((!sh->obs || (sh->ld_obs >= R.shadow_ld(0) && sh->ld_obs % 8 == 0 && ((uintptr_t)sh->obs & 15) == 0)) &&
                       (!sh->priv || (sh->ld_priv >= R.shadow_ld(1) && sh->ld_priv % 8 == 0 && ((uintptr_t)sh->priv & 15) == 0)),
                   HGYM_E_SHAPE, "observation shadow: bad leading dimension / alignment");
    *out = R.make_fwd_args(0, 2, M, xs, ldxs, nullptr, outs, ldos, false, &smp, nullptr, sh);
    out->nets = 2;
    size_t lds = 0;
    for (int i = 0; i < 2; ++i)
        lds = std::max(lds, (size_t)fused_lds_p(out->net[i], 32) + (size_t)fused_lds_q(out->net[i], 32) + (size_t)fused_lds_bias(out->net[i]));
    *lds_bytes = lds;
    return HGYM_OK;
}

#define HG_DISPATCH(cfg, net, w, stream, expr)                                                     \
    do {                                                                                           \
        if ((cfg)->precision == HGYM_F32) {                                                        \
            NetRunner<float> R{*(cfg), *(net), (w), (hipStream_t)(stream), (char*)(net)->workspace}; \
            return R.expr;                                                                         \
        } else {                                                                                   \
            NetRunner<__bf16> R{*(cfg), *(net), (w), (hipStream_t)(stream), (char*)(net)->workspace}; \
            return R.expr;                                                                         \
        }                                                                                          \
    } while (0)

}  // namespace hgym

using namespace hgym;

extern "C" {

int64_t hgym_net_param_count(const HgymNetConfig* cfg) {
    WsLayout w;
    if (ws_layout(cfg, &w)) return -1;
    return w.P;
}

int64_t hgym_net_workspace_bytes(const HgymNetConfig* cfg) {
    WsLayout w;
    if (ws_layout(cfg, &w)) return -1;
    return w.total_bytes;
}

int32_t hgym_net_sync_shadow(const HgymNetConfig* cfg, const HgymNet* net, void* stream) {
    WsLayout w;
    const int32_t rc = check_net(cfg, net, &w);
    if (rc) return rc;
    HG_DISPATCH(cfg, net, w, stream, sync_shadow());
}

int32_t hgym_mlp_forward(const HgymNetConfig* cfg, const HgymNet* net, int32_t which, int32_t M, const float* x, int64_t ldx, float* y,
                         void* stream) {
    WsLayout w;
    const int32_t rc = check_net(cfg, net, &w);
    if (rc) return rc;
    HG_REQUIRE(which == 0 || which == 1 || (which == 2 && cfg->aux_layers > 0), HGYM_E_BADARG, "which=%d", which);
    HG_REQUIRE(x && y, HGYM_E_BADARG, "null x / y");
    const int nout = which == 0 ? cfg->num_actions : (which == 1 ? 1 : cfg->aux_dims[cfg->aux_layers]);
    HG_DISPATCH(cfg, net, w, stream, forward(which, M, x, ldx, nullptr, y, nout, false));
}

int32_t hgym_critic_values(const HgymNetConfig* cfg, const HgymNet* net, int64_t M, const float* priv, float* values,
                           const HgymObsShadow* shadow, void* stream) {
    WsLayout w;
    const int32_t rc = check_net(cfg, net, &w);
    if (rc) return rc;
    HG_REQUIRE(M > 0 && priv && values, HGYM_E_BADARG, "M=%lld, null priv / values", (long long)M);
    HG_REQUIRE(!shadow || !shadow->priv || w.fused, HGYM_E_UNSUPPORTED, "the observation shadow exists on the fused bf16 path only");
    HG_DISPATCH(cfg, net, w, stream, critic_values(M, priv, values, shadow));
}

int64_t hgym_net_shadow_ld(const HgymNetConfig* cfg, int32_t which) {
    WsLayout w;
    if (ws_layout(cfg, &w) != HGYM_OK || !w.fused || which < 0 || which > 1) return 0;
    return (int64_t)w.net[which].layer[0].KBf * 32;
}

int32_t hgym_policy_act(const HgymNetConfig* cfg, const HgymNet* net, int32_t M, const float* obs, const float* priv, const float* z,
                        uint64_t seed, const int64_t* step_counter, float* actions, float* mu, float* sigma, float* logp, float* values,
                        const HgymObsShadow* shadow, void* stream) {
    WsLayout w;
    int32_t rc = check_net(cfg, net, &w);
    if (rc) return rc;
    HG_REQUIRE(obs && priv && actions && mu && sigma && logp && values, HGYM_E_BADARG, "null pointer");
    HG_DISPATCH(cfg, net, w, stream, act(M, obs, priv, z, seed, step_counter, actions, mu, sigma, logp, values, nullptr, shadow));
}

int32_t hgym_policy_act_fin(const HgymNetConfig* cfg, const HgymNet* net, int32_t M, const float* obs, const float* priv, const float* z,
                            uint64_t seed, const int64_t* step_counter, float* actions, float* mu, float* sigma, float* logp,
                            float* values, const HgymEnvConfig* env_cfg, const HgymEnvState* env_st, const HgymEnvOut* env_out,
                            const HgymObsShadow* shadow, void* stream) {
    WsLayout w;
    int32_t rc = check_net(cfg, net, &w);
    if (rc) return rc;
    HG_REQUIRE(obs && priv && actions && mu && sigma && logp && values, HGYM_E_BADARG, "null pointer");
    HG_REQUIRE(env_cfg && env_st && env_out, HGYM_E_BADARG, "null env cfg/state/out");
    HG_REQUIRE(env_st->counters && env_st->episode_acc && env_out->time_out && env_out->extras_time_outs && env_out->extras_episode &&
                   env_out->rew && env_out->reset, HGYM_E_BADARG, "null finaliser buffer");
    HG_REQUIRE(env_cfg->num_envs > 0, HGYM_E_SHAPE, "num_envs=%d", env_cfg->num_envs);
    const FinArgs fin = make_fin_args(*env_cfg, *env_st, *env_out, FIN_MODE_STEP);
    HG_DISPATCH(cfg, net, w, stream, act(M, obs, priv, z, seed, step_counter, actions, mu, sigma, logp, values, &fin, shadow));
}

int32_t hgym_ppo_grad(const HgymNetConfig* cfg, const HgymPPOConfig* ppo, const HgymNet* net, const HgymBatch* batch, void* stream) {
    WsLayout w;
    const int32_t rc = check_net(cfg, net, &w);
    if (rc) return rc;
    HG_REQUIRE(ppo && batch, HGYM_E_BADARG, "null ppo / batch");
    HG_REQUIRE(net->grads && net->opt_state, HGYM_E_BADARG, "null grads / opt_state");
    HG_REQUIRE(batch->obs && batch->priv && batch->actions && batch->values && batch->advantages && batch->returns && batch->logp &&
                   batch->mu && batch->sigma && batch->idx, HGYM_E_BADARG, "null batch tensor");
    HG_DISPATCH(cfg, net, w, stream, grad(*ppo, *batch));
}

int32_t hgym_ppo_grad_part(const HgymNetConfig* cfg, const HgymPPOConfig* ppo, const HgymNet* net, const HgymBatch* batch, int32_t part,
                           void* stream) {
    WsLayout w;
    const int32_t rc = check_net(cfg, net, &w);
    if (rc) return rc;
    HG_REQUIRE(ppo && batch, HGYM_E_BADARG, "null ppo / batch");
    HG_REQUIRE(part == 0 || part == 1, HGYM_E_BADARG, "part=%d (0 or 1)", part);
    HG_REQUIRE(net->grads && net->opt_state, HGYM_E_BADARG, "null grads / opt_state");
    HG_REQUIRE(batch->obs && batch->priv && batch->actions && batch->values && batch->advantages && batch->returns && batch->logp &&
                   batch->mu && batch->sigma && batch->idx, HGYM_E_BADARG, "null batch tensor");
    HG_DISPATCH(cfg, net, w, stream, grad(*ppo, *batch, part));
}

int64_t hgym_net_param_offset(const HgymNetConfig* cfg, int32_t which) {
    WsLayout w;
    if (ws_layout(cfg, &w) != HGYM_OK) return -1;
    if (which < 0 || which >= w.nnets) return which == w.nnets ? w.P : -1;
    return w.net[which].layer[0].w_off;
}

int32_t hgym_ppo_apply(const HgymNetConfig* cfg, const HgymPPOConfig* ppo, const HgymNet* net, void* stream) {
    WsLayout w;
    const int32_t rc = check_net(cfg, net, &w);
    if (rc) return rc;
    HG_REQUIRE(ppo, HGYM_E_BADARG, "null ppo");
    HG_REQUIRE(net->grads && net->adam_m && net->adam_v && net->opt_state, HGYM_E_BADARG, "null optimiser buffers");
    HG_DISPATCH(cfg, net, w, stream, apply(*ppo));
}

}  // extern "C"
