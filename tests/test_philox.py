"""The internal generator of the HIP path (csrc/hgym_common.hpp: philox4x32_10 / rng4 / u01 / box_muller / uniform_at / normal_at)
is Philox4x32-10: (1) oracle/philox.py against the Random123 known-answer vectors, (2) the product's SOURCE, compiled for the
host by tests/hostcheck, against the same vectors and against the oracle's counter / slot layout, (3) `-m gpu`: the policy
kernel's in-kernel draws (hgym_policy_act with z = NULL) against the oracle's stream for the same (seed, step, row)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import philox as X


def test_oracle_philox_known_answers():
    for ctr, key, want in X.KAT:
        got = X.philox4x32_10(*ctr, *key)
        assert tuple(int(v) for v in got) == want, (ctr, key)
    # vectorised evaluation = element-wise evaluation
    envs = np.arange(7, dtype=np.uint32)
    r = X.rng4(0x123456789ABCDEF0, (5 << 32) | 17, envs, 3)
    for e in range(7):
        one = X.philox4x32_10(e, 17, 5, 3, 0x9ABCDEF0, 0x12345678)
        assert [int(v[e]) for v in r] == [int(v) for v in one]


@pytest.fixture(scope="module")
def hostlib():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck"))
    import build_hostcheck
    lib = C.CDLL(build_hostcheck.build())
    lib.hc_philox.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
    lib.hc_philox.restype = None
    lib.hc_streams.argtypes = [C.c_uint64, C.c_int64, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.hc_streams.restype = None
    return lib


def test_kernel_source_philox_known_answers(hostlib):
    """hgym_common.hpp's philox4x32_10 (hc_philox takes key first, then the counter) on the Random123 vectors."""
    out = (C.c_uint32 * 4)()
    for ctr, key, want in X.KAT:
        hostlib.hc_philox(key[0], key[1], ctr[0], ctr[1], ctr[2], ctr[3], out)
        assert tuple(out) == want, (ctr, key, [hex(v) for v in out])


@pytest.mark.parametrize("seed,step,env,slot", [(5, 0, 0, 0), (0xDEADBEEFCAFEF00D, (3 << 32) + 99, 4095, 16), (7, 2399, 123456, 64)])
def test_kernel_source_stream_layout(hostlib, seed, step, env, slot):
    """uniform_at / normal_at of the kernel source == the oracle's (seed -> key halves, (env, step_lo, step_hi, slot) counter,
    four draws per call, Box-Muller pairing (x, y) and (z, w)); uniforms bit-exact, normals to libm rounding."""
    n = 13
    u, z = (C.c_float * n)(), (C.c_float * n)()
    hostlib.hc_streams(seed, step, env, slot, n, u, z)
    want_u = X.uniforms(seed, step, np.uint32(env), slot, n)
    want_z = X.normals(seed, step, np.uint32(env), slot, n)
    assert np.array_equal(np.array(u[:], dtype=np.float32), want_u)
    np.testing.assert_allclose(np.array(z[:], dtype=np.float32), want_z, rtol=2e-6, atol=2e-6)
    assert (want_u >= 0).all() and (want_u < 1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_device_policy_draws_are_the_oracle_stream(precision):
    """hgym_policy_act with z = NULL: (action - mu) / sigma of row m, column j must be normal j of slots 64.. of counter
    (m, step) under key `seed` -- the device evaluates Box-Muller with the hardware log2 / sqrt / sin / cos (~1e-6)."""
    from hgym import NetBuffers, make_net_config
    M, seed, step0 = 333, 0x0123456789ABCDEF, (1 << 33) + 41
    cfg = make_net_config(705, 219, 12, [512, 256, 128], [768, 256, 128], precision, 512)
    torch.manual_seed(2)
    net = NetBuffers(cfg, "cuda", learning_rate=1e-3)
    for k, v in net.views.items():
        v.copy_(torch.randn(v.shape, device="cuda") * 0.03)
    net.views["std"].copy_(torch.rand(12, device="cuda") + 0.5)
    net.sync_shadow()
    obs, priv = torch.randn(M, 705, device="cuda"), torch.randn(M, 219, device="cuda")
    step = torch.full((1,), step0, dtype=torch.int64, device="cuda")
    out = net.act(obs, priv, seed=seed, step_counter=step)
    torch.cuda.synchronize()
    z = ((out["actions"] - out["mu"]) / out["sigma"]).cpu().numpy()
    want = X.normals(seed, step0, np.arange(M, dtype=np.uint32), X.SLOT_POLICY, 12)
    # (a - mu) / sigma re-derives z through two fp32 roundings of values up to |mu| + 4 sigma
    np.testing.assert_allclose(z, want, rtol=0, atol=2e-5 * max(1.0, float(out["mu"].abs().max())))
