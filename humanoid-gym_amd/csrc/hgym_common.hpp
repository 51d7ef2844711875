// hgym_common.hpp -- error plumbing, launch helpers, Philox4x32-10, small math shared by all kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/hgym.h"

#define HG_HD __host__ __device__ __forceinline__

namespace hgym {

// ---------------------------------------------------------------------------------------------- errors
char* last_error_buf();  // thread-local, defined in hgym_capi.hip

#define HG_FAIL(code, ...)                                            \
    do {                                                              \
        snprintf(::hgym::last_error_buf(), 512, __VA_ARGS__);         \
        return (code);                                                \
    } while (0)

#define HG_REQUIRE(cond, code, ...) \
    do {                            \
        if (!(cond)) HG_FAIL(code, __VA_ARGS__); \
    } while (0)

#define HG_CHECK_LAUNCH(what)                                                                   \
    do {                                                                                        \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ != hipSuccess) HG_FAIL(HGYM_E_LAUNCH, "%s: %s", what, hipGetErrorString(e_));     \
    } while (0)

// ---------------------------------------------------------------------------------------------- profiling hooks
// (hgym_capi.hip) begin/end bracket one launch of a profiled kernel class with HIP events when enabled.
bool prof_on();
void prof_begin(int cls, hipStream_t s);
void prof_end(int cls, hipStream_t s, double work);
long long* phase_buffer(int64_t blocks);     // hgym_prof_phase_buffer: null unless set and large enough
int device_cus();                           // compute units of the CURRENT device (cached per device id)
// Opt-in for more than 64 KiB of dynamic LDS: hipFuncSetAttribute(MaxDynamicSharedMemorySize) for kernel `fn` on the CURRENT device.
// The attribute is per (function, device): the reservation is cached under that pair (mutex-guarded), so a process that drives
// several devices, or several host threads, never launches with more dynamic LDS than was reserved on that device.
// Returns HGYM_OK or HGYM_E_LAUNCH (message in hgym_last_error).
int32_t ensure_dynamic_lds(const void* fn, size_t bytes, const char* what);

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------------------------- Philox4x32-10
// Counter-based RNG (Salmon et al. 2011), keyed by (seed); counter = (env, step_lo, step_hi, slot).
// Pure 32-bit integer arithmetic: identical on host and device; oracle/philox.py restates it and
// tests/test_philox.py pins all three (oracle, this source on the host, the device) to the Random123 known-answer vectors.
struct U4 {
    uint32_t x, y, z, w;
};

HG_HD U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // each round's two 32 x 32 -> 64 bit products as ONE 64-bit multiply each: the compiler then emits one v_mad_u64_u32 per
        // product instead of a v_mul_hi_u32 + v_mul_lo_u32 pair (all quarter-rate: 20 instead of 40 per call; the env step draws
        // ~1000 calls per 32-env block per step on its critical path)
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c.x, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        U4 n;
        n.x = hi1 ^ c.y ^ k0;
        n.y = lo1;
        n.z = hi0 ^ c.w ^ k1;
        n.w = lo0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// 24-bit mantissa uniform in [0,1): exact in fp32, never 1.0
HG_HD float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

struct RngKey {
    uint32_t k0, k1;     // seed
    uint32_t s0, s1;     // step
};

HG_HD U4 rng4(const RngKey& k, uint32_t env, uint32_t slot) {
    U4 c;
    c.x = env;
    c.y = k.s0;
    c.z = k.s1;
    c.w = slot;
    return philox4x32_10(c, k.k0, k.k1);
}

// Box-Muller on two uniforms; u1 is mapped to (0,1] so that log() is finite.  On the device the four transcendentals are
// the hardware ones (v_log_f32 = log2, v_sqrt_f32, v_sin_f32 / v_cos_f32 take the angle in revolutions, i.e. u2 itself):
// ~10 instructions instead of ~250 for libm's logf / sinf / cosf with their range reductions -- the sampling epilogue of
// the policy step and the noise-table fill of the env step sit on the rollout's latency chain.  These are draws of the
// built-in generator (nothing in the reference to match bit for bit); the host emulation keeps libm and agrees to ~1e-6.
// HGYM_FAST_NORMALS=0 restores libm on the device.
HG_HD void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = 1.0f - u01(a);
    const float u2 = u01(b);
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));     // -2 ln 2 * log2(u1)
    z0 = r * __builtin_amdgcn_cosf(u2);
    z1 = r * __builtin_amdgcn_sinf(u2);
#else
    const float r = sqrtf(-2.0f * logf(u1));
    const float t = 6.2831855f * u2;
    z0 = r * cosf(t);
    z1 = r * sinf(t);
#endif
}

// Philox slot map (one counter word); every consumer owns a disjoint range.
enum : uint32_t {
    SLOT_DELAY_CMD = 0,   // x: action delay, y,z,w: callback command resample
    SLOT_CMD_RESET = 1,   // x,y,z: reset command resample
    SLOT_ACT = 2,         // 2..4  : 12 action-noise normals
    SLOT_DOF = 5,         // 5..7  : 12 reset joint offsets
    SLOT_PUSH = 8,        // 8..9  : 5 push draws
    SLOT_TERRAIN = 10,    // x,y: spawn jitter of custom origins, z: terrain-level redraw
    SLOT_OBS = 16,        // 16..27: 47 observation-noise normals (pairs)
    SLOT_PHYS = 32,       // 32..47: synthetic physics
    SLOT_POLICY = 64      // 64..66: 12 policy-sampling normals
};

// standard normal number `i` of a block of normals starting at slot `base` (2 normals per uniform pair)
HG_HD float normal_at(const RngKey& k, uint32_t env, uint32_t base, int i) {
    const int pair = i >> 1;                 // pair index
    const U4 r = rng4(k, env, base + (uint32_t)(pair >> 1));
    float z0, z1;
    if (pair & 1) box_muller(r.z, r.w, z0, z1);
    else box_muller(r.x, r.y, z0, z1);
    return (i & 1) ? z1 : z0;
}

// 4*CALLS standard normals from CALLS Philox evaluations starting at slot `base`; element i equals normal_at(.., i)
template <int CALLS>
HG_HD void normals_block(const RngKey& k, uint32_t env, uint32_t base, float* out) {
#pragma unroll
    for (int c = 0; c < CALLS; ++c) {
        const U4 r = rng4(k, env, base + (uint32_t)c);
        box_muller(r.x, r.y, out[4 * c + 0], out[4 * c + 1]);
        box_muller(r.z, r.w, out[4 * c + 2], out[4 * c + 3]);
    }
}

HG_HD float uniform_at(const RngKey& k, uint32_t env, uint32_t base, int i) {
    const U4 r = rng4(k, env, base + (uint32_t)(i >> 2));
    const int j = i & 3;
    return u01(j == 0 ? r.x : j == 1 ? r.y : j == 2 ? r.z : r.w);
}

HG_HD float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

}  // namespace hgym
