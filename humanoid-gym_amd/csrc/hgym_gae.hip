// hgym_gae.hip -- rollout-storage side: time-out bootstrap store, GAE(lambda), advantage normalisation.
//
// GAE (algo/ppo/rollout_storage.py:122-133) is the backward linear recurrence
//     A_t = delta_t + c_t * A_{t+1},   delta_t = r_t + nt_t*gamma*V_{t+1} - V_t,   c_t = nt_t*gamma*lambda
// i.e. a suffix composition of affine maps x -> b + a*x.  One 64-lane wavefront owns one env and one lane one
// timestep, so the whole T=60 rollout is a single 6-step Kogge-Stone scan done with wavefront shuffles
// (no LDS traffic in the scan, no 60-step serial dependency); longer rollouts are walked in 64-step tiles
// from the end with a scalar carry.  Loads/stores of the time-major (T,N) arrays go through an LDS tile
// [64 steps][16 envs] so that HBM sees 64-byte runs instead of 4-byte gathers.  HBM-bound: 25 B/env-step.
#include "hgym_common.hpp"

namespace hgym {

constexpr int GAE_ENVS = 16;   // envs per workgroup (4 per wavefront)
constexpr int GAE_PAD = 17;    // LDS row stride (odd: conflict-free column reads)

// BOOT (hgym_gae_bootstrap): `rewards` holds raw rewards; r_t + gamma * (V_t * time_outs_t) -- PPO.process_env_step's bootstrap
// (ppo.py:107-108) in store_step_kernel's three fp32 roundings (this file is built with -ffp-contract=off) -- is formed while the
// tile is staged, used by the scan and written back, so the storage column ends up as the per-step path leaves it.
template <bool BOOT>
__global__ __launch_bounds__(256) void gae_kernel(int T, int N, float* __restrict__ rewards,
                                                  const float* __restrict__ values, const uint8_t* __restrict__ dones,
                                                  const uint8_t* __restrict__ time_outs,
                                                  const float* __restrict__ last_values, float gamma, float lam,
                                                  float* __restrict__ returns, float* __restrict__ advantages,
                                                  double* __restrict__ stats) {
    __shared__ float s_r[64][GAE_PAD];
    __shared__ float s_v[65][GAE_PAD];
    __shared__ float s_d[64][GAE_PAD];
    __shared__ float s_carry[GAE_ENVS];
    __shared__ double s_sum[4][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e0 = blockIdx.x * GAE_ENVS;
    const int nE = min(GAE_ENVS, N - e0);
    if (tid < GAE_ENVS) s_carry[tid] = 0.0f;
    double sum1 = 0.0, sum2 = 0.0;
    const int ntiles = (T + 63) / 64;
    for (int tile = ntiles - 1; tile >= 0; --tile) {
        const int t0 = tile * 64;
        const int nT = min(64, T - t0);
        __syncthreads();
        // stage [nT (+1 row of next values)][nE]
        for (int i = tid; i < 65 * GAE_ENVS; i += 256) {
            const int t = i / GAE_ENVS, el = i % GAE_ENVS;
            if (el < nE) {
                const int tg = t0 + t;
                if (t < nT) {
                    const int64_t gi = (int64_t)tg * N + e0 + el;
                    const float v = values[gi];
                    float r = rewards[gi];
                    if (BOOT) {
                        const float to = (float)(time_outs[gi] != 0);
                        r = r + gamma * (v * to);
                        rewards[gi] = r;
                    }
                    s_r[t][el] = r;
                    s_d[t][el] = (float)dones[gi];
                    s_v[t][el] = v;
                } else if (t == nT) {
                    s_v[t][el] = (tg >= T) ? last_values[e0 + el] : values[(int64_t)tg * N + e0 + el];
                }
            }
        }
        __syncthreads();
        for (int q = 0; q < GAE_ENVS / 4; ++q) {
            const int el = wave * (GAE_ENVS / 4) + q;
            if (el >= nE) break;                       // wave-uniform
            const bool valid = lane < nT;
            float a = 1.0f, b = 0.0f, v = 0.0f;
            if (valid) {
                v = s_v[lane][el];
                const float nt = 1.0f - s_d[lane][el];
                b = s_r[lane][el] + nt * gamma * s_v[lane + 1][el] - v;
                a = nt * gamma * lam;
            }
            // inclusive suffix scan of affine maps: after the loop lane t holds F_t = f_t o f_{t+1} o ... o f_63
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float a2 = __shfl_down(a, off, 64);
                const float b2 = __shfl_down(b, off, 64);
                if (lane + off < 64) {
                    b = b + a * b2;
                    a = a * a2;
                }
            }
            const float carry = s_carry[el];
            const float adv_t = b + a * carry;        // A_t
            const float ret = adv_t + v;
            const float adv = ret - v;                // rollout_storage.py:135 forms returns - values
            const float head = __shfl(adv_t, 0, 64);
            if (valid) {
                s_r[lane][el] = ret;                  // reuse the tile as the output staging area
                s_d[lane][el] = adv;
                sum1 += (double)adv;
                sum2 += (double)adv * (double)adv;
            }
            if (lane == 0) s_carry[el] = head;
        }
        __syncthreads();
        for (int i = tid; i < 64 * GAE_ENVS; i += 256) {
            const int t = i / GAE_ENVS, el = i % GAE_ENVS;
            if (t < nT && el < nE) {
                returns[(int64_t)(t0 + t) * N + e0 + el] = s_r[t][el];
                advantages[(int64_t)(t0 + t) * N + e0 + el] = s_d[t][el];
            }
        }
    }
    // advantage statistics for the normalisation: wave reduce -> block reduce -> per-workgroup partial sums in the caller's scratch
    // (stats + 4), added up IN A FIXED ORDER by the last workgroup to arrive (stats[3] holds the arrival counter): the same bits in every
    // run and on every rank, like the gradient norm (sqnorm_prologue_kernel) -- fp64 atomics on two words summed in arrival order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum1 += __shfl_down(sum1, off, 64);
        sum2 += __shfl_down(sum2, off, 64);
    }
    if (lane == 0) {
        s_sum[wave][0] = sum1;
        s_sum[wave][1] = sum2;
    }
    __syncthreads();
    __shared__ int s_last;
    double* __restrict__ part = stats + 4;
    unsigned int* cnt = reinterpret_cast<unsigned int*>(stats + 3);
    if (tid == 0) {
        part[2 * blockIdx.x] = s_sum[0][0] + s_sum[1][0] + s_sum[2][0] + s_sum[3][0];
        part[2 * blockIdx.x + 1] = s_sum[0][1] + s_sum[1][1] + s_sum[2][1] + s_sum[3][1];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned int done = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (done == gridDim.x - 1u) ? 1 : 0;
    }
    __syncthreads();
    if (s_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        double a = 0.0, b = 0.0;
        for (int i = tid; i < (int)gridDim.x; i += 256) {      // thread t: workgroups t, t + 256, ... in that order
            a += part[2 * i];
            b += part[2 * i + 1];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_down(a, off, 64);
            b += __shfl_down(b, off, 64);
        }
        __syncthreads();
        if (lane == 0) {
            s_sum[wave][0] = a;
            s_sum[wave][1] = b;
        }
        __syncthreads();
        if (tid == 0) {
            stats[0] = s_sum[0][0] + s_sum[1][0] + s_sum[2][0] + s_sum[3][0];
            stats[1] = s_sum[0][1] + s_sum[1][1] + s_sum[2][1] + s_sum[3][1];
            stats[2] = (double)T * (double)N;
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the next call counts from zero
        }
    }
}

// rollout_storage.py:136: (adv - mean) / (std_unbiased + 1e-8), mean/std from fp64 sums, arithmetic in fp32
__global__ __launch_bounds__(256) void adv_normalize_kernel(int64_t count, float* __restrict__ adv, const double* __restrict__ stats) {
    const double n = stats[2];
    const double mean_d = stats[0] / n;
    double var = (stats[1] - stats[0] * stats[0] / n) / (n - 1.0);
    if (var < 0.0) var = 0.0;
    const float mean = (float)mean_d;
    const float denom = (float)sqrt(var) + 1e-8f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        adv[i] = (adv[i] - mean) / denom;
}

// ppo.py:103-113 + rollout_storage.py:91-94 for the scalar columns of one step
__global__ __launch_bounds__(256) void store_step_kernel(int n, const float* __restrict__ rew, const float* __restrict__ values,
                                                         const uint8_t* __restrict__ time_outs, const uint8_t* __restrict__ dones,
                                                         float gamma, float* __restrict__ rewards_slot, uint8_t* __restrict__ dones_slot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float to = time_outs ? (float)(time_outs[i] != 0) : 0.0f;
    rewards_slot[i] = rew[i] + gamma * (values[i] * to);
    dones_slot[i] = dones[i] != 0;
}

// The minibatch permutation of RolloutStorage.mini_batch_generator (rollout_storage.py:149: torch.randperm(T*N)) as a keyed
// bijection evaluated per index: a 6-round balanced Feistel network over the next even power of two >= n, cycle-walked back
// into [0, n) (the walk visits < 2 candidates per index on average, < 4/3 for the XBot-L batch of 245 760 in 2^18).
// torch.randperm on the device is a radix sort of random keys + 8 merge passes (125 us per iteration at this size, measured);
// this is one launch of a few microseconds and needs no scratch.  Any uniform shuffle serves the PPO update equally; the
// draw is identified by (seed, draw) so that runs are reproducible and ranks can use different or equal streams.
HG_HD uint32_t perm_mix(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
HG_HD uint64_t perm_key(uint64_t seed, uint64_t draw) { return (seed ^ (draw * 0x9e3779b97f4a7c15ull)) * 0xd1342543de82ef95ull + draw; }
HG_HD int64_t perm_index(int64_t i, int64_t n, int half_bits, uint32_t k0, uint32_t k1) {
    const uint32_t mask = (1u << half_bits) - 1u;
    uint64_t x = (uint64_t)i;
    do {
        uint32_t l = (uint32_t)(x >> half_bits) & mask, r = (uint32_t)x & mask;
#pragma unroll
        for (uint32_t round = 0; round < 6; ++round) {
            const uint32_t f = perm_mix(r ^ perm_mix(k0 + round * 0x9e3779b9u) ^ k1) & mask;
            const uint32_t t = l ^ f;
            l = r;
            r = t;
        }
        x = ((uint64_t)l << half_bits) | r;
    } while (x >= (uint64_t)n);
    return (int64_t)x;
}
__global__ __launch_bounds__(256) void randperm_kernel(int64_t n, int half_bits, uint32_t k0, uint32_t k1, int64_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = perm_index(i, n, half_bits, k0, k1);
}

// The same permutation with the draw number read from device memory (header v9): nothing in the launch's arguments changes from one
// learning iteration to the next, so a captured update (HIP graph) replays it -- the caller advances *draw on the device.
__global__ __launch_bounds__(256) void randperm_dev_kernel(int64_t n, int half_bits, uint64_t seed, const int64_t* __restrict__ draw_p,
                                                           int64_t* __restrict__ out) {
    const uint64_t draw = (uint64_t)draw_p[0];
    const uint64_t key = perm_key(seed, draw);
    const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = perm_index(i, n, half_bits, k0, k1);
}

}  // namespace hgym

using namespace hgym;

extern "C" {

int32_t hgym_randperm_dev(int64_t n, uint64_t seed, const int64_t* draw, int64_t* out, void* stream) {
    HG_REQUIRE(n > 0 && n <= ((int64_t)1 << 40), HGYM_E_SHAPE, "n=%lld", (long long)n);
    HG_REQUIRE(out && draw, HGYM_E_BADARG, "null pointer");
    int half_bits = 1;
    while (((int64_t)1 << (2 * half_bits)) < n) ++half_bits;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(randperm_dev_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, half_bits, seed, draw, out);
    HG_CHECK_LAUNCH("randperm_dev_kernel");
    return HGYM_OK;
}

int32_t hgym_randperm(int64_t n, uint64_t seed, uint64_t draw, int64_t* out, void* stream) {
    HG_REQUIRE(n > 0 && n <= ((int64_t)1 << 40), HGYM_E_SHAPE, "n=%lld", (long long)n);
    HG_REQUIRE(out, HGYM_E_BADARG, "null output");
    int half_bits = 1;
    while (((int64_t)1 << (2 * half_bits)) < n) ++half_bits;
    const uint64_t key = perm_key(seed, draw);
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(randperm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, half_bits, (uint32_t)key, (uint32_t)(key >> 32), out);
    HG_CHECK_LAUNCH("randperm_kernel");
    return HGYM_OK;
}

int32_t hgym_store_step(int32_t n, const float* rew, const float* values, const uint8_t* time_outs, const uint8_t* dones,
                        float gamma, float* rewards_slot, uint8_t* dones_slot, void* stream) {
    HG_REQUIRE(n > 0, HGYM_E_SHAPE, "n=%d", n);
    HG_REQUIRE(rew && values && dones && rewards_slot && dones_slot, HGYM_E_BADARG, "null pointer");
    hipLaunchKernelGGL(store_step_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, n, rew, values, time_outs, dones,
                       gamma, rewards_slot, dones_slot);
    HG_CHECK_LAUNCH("store_step_kernel");
    return HGYM_OK;
}

int32_t hgym_gae(int32_t T, int32_t n, const float* rewards, const float* values, const uint8_t* dones, const float* last_values,
                 float gamma, float lam, float* returns, float* advantages, double* stats, void* stream) {
    HG_REQUIRE(T > 0 && n > 0, HGYM_E_SHAPE, "T=%d n=%d", T, n);
    HG_REQUIRE(rewards && values && dones && last_values && returns && advantages && stats, HGYM_E_BADARG, "null pointer");
    prof_begin(HGYM_PROF_GAE, (hipStream_t)stream);
    hipLaunchKernelGGL(gae_kernel<false>, dim3(ceil_div(n, GAE_ENVS)), dim3(256), 0, (hipStream_t)stream, T, n, const_cast<float*>(rewards), values,
                       dones, nullptr, last_values, gamma, lam, returns, advantages, stats);
    prof_end(HGYM_PROF_GAE, (hipStream_t)stream, (double)T * n * 17.0);   // r,V f32 + done u8 read, ret, adv f32 written
    HG_CHECK_LAUNCH("gae_kernel");
    return HGYM_OK;
}

int32_t hgym_gae_bootstrap(int32_t T, int32_t n, float* rewards, const float* values, const uint8_t* dones, const uint8_t* time_outs,
                           const float* last_values, float gamma, float lam, float* returns, float* advantages, double* stats, void* stream) {
    HG_REQUIRE(T > 0 && n > 0, HGYM_E_SHAPE, "T=%d n=%d", T, n);
    HG_REQUIRE(rewards && values && dones && time_outs && last_values && returns && advantages && stats, HGYM_E_BADARG, "null pointer");
    prof_begin(HGYM_PROF_GAE, (hipStream_t)stream);
    hipLaunchKernelGGL(gae_kernel<true>, dim3(ceil_div(n, GAE_ENVS)), dim3(256), 0, (hipStream_t)stream, T, n, rewards, values, dones, time_outs,
                       last_values, gamma, lam, returns, advantages, stats);
    prof_end(HGYM_PROF_GAE, (hipStream_t)stream, (double)T * n * 22.0);   // + the time-out byte read, the bootstrapped reward written back
    HG_CHECK_LAUNCH("gae_kernel<bootstrap>");
    return HGYM_OK;
}

int32_t hgym_adv_normalize(int64_t count, float* advantages, const double* stats, void* stream) {
    HG_REQUIRE(count > 1, HGYM_E_SHAPE, "count=%lld", (long long)count);
    HG_REQUIRE(advantages && stats, HGYM_E_BADARG, "null pointer");
    const int blocks = (int)((count + 255) / 256 > 2048 ? 2048 : (count + 255) / 256);
    hipLaunchKernelGGL(adv_normalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, count, advantages, stats);
    HG_CHECK_LAUNCH("adv_normalize_kernel");
    return HGYM_OK;
}

}  // extern "C"
