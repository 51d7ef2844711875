"""Philox4x32-10 and the draw layout of the HIP path's internal generator -- TEST INFRASTRUCTURE (checker only).

The reference draws its noise from torch's global generator (humanoid_env.py:194,196,251; legged_robot.py:328-331,367;
actor_critic.py:118 `Normal.sample`); a counter-based generator replaces it on the device so that every lane can draw
without state.  Nothing in the reference pins the *values*; what is pinned here is that the generator IS Philox4x32-10
(Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) -- against the known-answer vectors of
the authors' Random123 distribution (`kat_vectors`, philox4x32 10 rows) -- and how counters / keys / slots map to draws
(`humanoid-gym_amd/csrc/hgym_common.hpp`: `philox4x32_10`, `rng4`, `u01`, `box_muller`, `uniform_at`, `normal_at`).

numpy, vectorised over arbitrary counter arrays; pure 32-bit integer arithmetic, so it is bit-exact with the device.
"""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57          # Philox4x32 round multipliers
W0, W1 = 0x9E3779B9, 0xBB67AE85          # Weyl key increments (golden ratio, sqrt(3) - 1)

# Random123 kat_vectors, "philox4x32 10": (counter[4], key[2]) -> output[4]
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Ten Philox rounds on counter words (c0..c3) under key (k0, k1); broadcasts over numpy arrays; returns 4 uint32 arrays."""
    c = [np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF for x in (c0, c1, c2, c3)]
    k0 = np.asarray(k0, dtype=np.uint64) & 0xFFFFFFFF
    k1 = np.asarray(k1, dtype=np.uint64) & 0xFFFFFFFF
    x, y, z, w = np.broadcast_arrays(*c)
    for _ in range(10):
        p0 = M0 * x                       # 64-bit products of 32-bit operands: no overflow in uint64
        p1 = M1 * z
        hi0, lo0 = p0 >> 32, p0 & 0xFFFFFFFF
        hi1, lo1 = p1 >> 32, p1 & 0xFFFFFFFF
        x, y, z, w = hi1 ^ y ^ k0, lo1, hi0 ^ w ^ k1, lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(v.astype(np.uint32) for v in (x, y, z, w))


def rng4(seed, step, env, slot):
    """hgym_common.hpp `rng4`: key = the two halves of the 64-bit seed, counter = (env, step_lo, step_hi, slot)."""
    seed, step = int(seed) & 0xFFFFFFFFFFFFFFFF, int(step) & 0xFFFFFFFFFFFFFFFF
    return philox4x32_10(env, step & 0xFFFFFFFF, step >> 32, slot, seed & 0xFFFFFFFF, seed >> 32)


def u01(x):
    """24-bit uniform in [0, 1): exact in fp32."""
    return (np.asarray(x, dtype=np.uint32) >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)


def box_muller(a, b):
    """Two standard normals from two 32-bit words, the libm form of `box_muller` (the device evaluates the same expression
    with the hardware log2 / sqrt / sin / cos: agreement ~1e-6 absolute, not bit-exact)."""
    u1 = np.float32(1.0) - u01(a)
    u2 = u01(b)
    r = np.sqrt(np.float32(-2.0) * np.log(u1, dtype=np.float32), dtype=np.float32)
    t = np.float32(6.2831855) * u2
    return r * np.cos(t, dtype=np.float32), r * np.sin(t, dtype=np.float32)


def uniforms(seed, step, env, base, n):
    """`uniform_at(k, env, base, i)` for i in [0, n): four uniforms per Philox call, slots base, base+1, ..."""
    env = np.asarray(env, dtype=np.uint32)
    out = np.empty(env.shape + (n,), dtype=np.float32)
    for c in range((n + 3) // 4):
        r = rng4(seed, step, env, base + c)
        for j in range(4):
            if 4 * c + j < n:
                out[..., 4 * c + j] = u01(r[j])
    return out


def normals(seed, step, env, base, n):
    """`normal_at(k, env, base, i)` for i in [0, n): per Philox call (x, y) -> normals 4c, 4c+1 and (z, w) -> 4c+2, 4c+3."""
    env = np.asarray(env, dtype=np.uint32)
    out = np.empty(env.shape + (n,), dtype=np.float32)
    for c in range((n + 3) // 4):
        r = rng4(seed, step, env, base + c)
        z = box_muller(r[0], r[1]) + box_muller(r[2], r[3])
        for j in range(4):
            if 4 * c + j < n:
                out[..., 4 * c + j] = z[j]
    return out


SLOT_POLICY = 64      # hgym_common.hpp slot map: 64..66 = the 12 policy-sampling normals of one row


# ---- reproducible test tensors ------------------------------------------------------------------------------------------
# Large fixture INPUTS (initial parameters, observation batches) are not stored: they are regenerated from (seed, tag) with the
# functions below wherever a fixture is recorded (tests/golden/gen_fixtures.py, on the reference) and replayed (tests/, on the
# GPU box).  Integer Philox words -> float64 arithmetic -> one rounding to fp32, so the values do not depend on the platform's
# fp32 libm.
def _words(seed, tag, n):
    calls = (n + 3) // 4
    r = rng4(seed, tag, np.arange(calls, dtype=np.uint32), 0)
    return np.stack(r, axis=1).reshape(-1)[:n]


def fill_uniform(seed, tag, shape, lo=0.0, hi=1.0):
    n = int(np.prod(shape))
    u = (_words(seed, tag, n) >> 8).astype(np.float64) * (1.0 / 16777216.0)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def fill_normal(seed, tag, shape, scale=1.0):
    n = int(np.prod(shape))
    m = (n + 1) // 2
    w = _words(seed, tag, 2 * m).astype(np.uint64)
    u1 = 1.0 - (w[0::2] >> 8).astype(np.float64) * (1.0 / 16777216.0)
    u2 = (w[1::2] >> 8).astype(np.float64) * (1.0 / 16777216.0)
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.stack([r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)], axis=1).reshape(-1)[:n]
    return (scale * z).astype(np.float32).reshape(shape)
