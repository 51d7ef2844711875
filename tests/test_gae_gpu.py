"""-m gpu: GAE wavefront scan + advantage normalisation + time-out bootstrap store vs the oracle.
Bar: returns / advantages within 1e-5 relative of the reference computation (north_star)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import ppo_oracle as P

pytestmark = pytest.mark.gpu


def _run(T, N, rewards, values, dones, last, gamma=0.994, lam=0.9):
    from hgym import _lib as L
    dev = "cuda"
    r, v, d, lv = rewards.to(dev).contiguous(), values.to(dev).contiguous(), dones.to(dev).to(torch.uint8).contiguous(), last.to(dev).contiguous()
    ret, adv = torch.zeros(T, N, device=dev), torch.zeros(T, N, device=dev)
    stats = L.gae_stats(N, dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(L.lib.hgym_gae(T, N, L.fptr(r), L.fptr(v), L.u8ptr(d), L.fptr(lv), gamma, lam, L.fptr(ret), L.fptr(adv), L.f64ptr(stats), s))
    raw = adv.clone()
    L.check(L.lib.hgym_adv_normalize(T * N, L.fptr(adv), L.f64ptr(stats), s))
    torch.cuda.synchronize()
    return ret.cpu(), raw.cpu(), adv.cpu(), stats.cpu()


def test_gae_known_answer_gpu(golden_dir):
    G = np.load(os.path.join(golden_dir, "gae.npz"))
    t = lambda k: torch.from_numpy(G[k])
    ret, raw, adv, _ = _run(4, 2, t("kat_rewards"), t("kat_values"), t("kat_dones"), t("kat_last"))
    np.testing.assert_allclose(ret.numpy(), G["kat_returns"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(adv.numpy(), G["kat_adv"], rtol=1e-5, atol=1e-6)
    ret, raw, adv, _ = _run(60, 24, t("rnd_rewards"), t("rnd_values"), t("rnd_dones"), t("rnd_last"))
    np.testing.assert_allclose(ret.numpy(), G["rnd_returns"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(adv.numpy(), G["rnd_adv"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("T,N", [(60, 4096), (24, 100), (64, 17), (150, 33), (1, 5)])
def test_gae_vs_oracle(T, N):
    g = torch.Generator().manual_seed(T * 1000 + N)
    r = torch.rand(T, N, generator=g) * 0.3
    v = torch.randn(T, N, generator=g) * 2 + 3
    d = torch.rand(T, N, generator=g) < 0.03
    lv = torch.randn(N, generator=g) * 2 + 3
    ret, raw, adv, stats = _run(T, N, r, v, d, lv)
    oret, oraw = P.gae_returns(r, v, d, lv, 0.994, 0.9)
    np.testing.assert_allclose(ret.numpy(), oret.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(raw.numpy(), oraw.numpy(), rtol=1e-5, atol=2e-5)
    assert float(stats[2]) == T * N and float(stats[3]) == 0.0
    np.testing.assert_allclose(float(stats[0]), float(oraw.double().sum()), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(float(stats[1]), float((oraw.double() ** 2).sum()), rtol=1e-6, atol=1e-6)
    ret2, raw2, adv2, stats2 = _run(T, N, r, v, d, lv)
    assert torch.equal(stats2[:3], stats[:3])              # order-fixed sums: reproducible bits
    if T * N > 1:
        np.testing.assert_allclose(adv.numpy(), P.normalize_advantages(oraw).numpy(), rtol=1e-4, atol=2e-5)


def test_gae_properties_full_size():
    """Size-independent properties at the BASELINE size: linearity in rewards, dones cut the recursion."""
    T, N = 60, 8192
    g = torch.Generator().manual_seed(0)
    r1, r2 = torch.rand(T, N, generator=g), torch.rand(T, N, generator=g)
    v = torch.zeros(T, N)
    d = torch.rand(T, N, generator=g) < 0.05
    lv = torch.zeros(N)
    a, _, _, _ = _run(T, N, r1, v, d, lv)
    b, _, _, _ = _run(T, N, r2, v, d, lv)
    c, _, _, _ = _run(T, N, r1 + r2, v, d, lv)
    np.testing.assert_allclose((a + b).numpy(), c.numpy(), rtol=1e-5, atol=1e-5)
    # at a done step the return is exactly the reward (V=0): the recursion is cut
    assert torch.equal(c[d], (r1 + r2)[d])


def test_store_step_gpu():
    from hgym import _lib as L
    n = 1000
    g = torch.Generator().manual_seed(3)
    rew, val = torch.rand(n, generator=g), torch.randn(n, 1, generator=g)
    to = torch.rand(n, generator=g) < 0.3
    dn = to | (torch.rand(n, generator=g) < 0.2)
    want = P.bootstrap_rewards(rew, val, to, 0.994)
    out_r, out_d = torch.zeros(n, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    keep = [rew.cuda(), val.cuda().contiguous(), to.cuda().to(torch.uint8), dn.cuda().to(torch.uint8)]
    L.check(L.lib.hgym_store_step(n, L.fptr(keep[0]), L.fptr(keep[1]), L.u8ptr(keep[2]), L.u8ptr(keep[3]), 0.994,
                                  L.fptr(out_r), L.u8ptr(out_d), s))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out_r.cpu().numpy(), want.numpy(), rtol=1e-6, atol=1e-7)
    assert torch.equal(out_d.cpu().bool(), dn)


@pytest.mark.parametrize("T,N", [(60, 4096), (24, 100), (130, 37)])
def test_gae_bootstrap_equals_store_step_then_gae(T, N):
    """hgym_gae_bootstrap (deferred values, header v7) on RAW rewards + the bootstrap's time-out flags == hgym_store_step's bootstrap
    (ppo.py:107-108) followed by hgym_gae: returns, advantages, statistics AND the rewards column it writes back, bit for bit."""
    import ctypes as C
    from hgym import _lib as L
    g = torch.Generator(device="cuda").manual_seed(T * 1000 + N)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    raw, val, lv = r(T, N), r(T, N), r(N)
    dones = (torch.rand(T, N, device="cuda", generator=g) < 0.03).to(torch.uint8)
    tos = (torch.rand(T, N, device="cuda", generator=g) < 0.1).to(torch.uint8)
    gamma, lam = 0.994, 0.9
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    boot = torch.empty_like(raw)
    dslot = torch.empty_like(dones)
    for t in range(T):
        L.check(L.lib.hgym_store_step(N, L.fptr(raw[t]), L.fptr(val[t]), L.u8ptr(tos[t]), L.u8ptr(dones[t]), gamma, L.fptr(boot[t]),
                                      L.u8ptr(dslot[t]), s), "hgym_store_step")
    out = {}
    for kind in ("two steps", "bootstrap"):
        ret, adv, stats = torch.empty(T, N, device="cuda"), torch.empty(T, N, device="cuda"), L.gae_stats(N, "cuda")
        if kind == "two steps":
            rew = boot.clone()
            L.check(L.lib.hgym_gae(T, N, L.fptr(rew), L.fptr(val), L.u8ptr(dones), L.fptr(lv), gamma, lam, L.fptr(ret), L.fptr(adv),
                                   L.f64ptr(stats), s), "hgym_gae")
        else:
            rew = raw.clone()
            L.check(L.lib.hgym_gae_bootstrap(T, N, L.fptr(rew), L.fptr(val), L.u8ptr(dones), L.u8ptr(tos), L.fptr(lv), gamma, lam, L.fptr(ret),
                                             L.fptr(adv), L.f64ptr(stats), s), "hgym_gae_bootstrap")
        torch.cuda.synchronize()
        out[kind] = (rew, ret, adv, stats)
    assert int((tos != 0).sum()) > 0 and not torch.equal(out["two steps"][0], raw)
    assert torch.equal(out["bootstrap"][0], out["two steps"][0])                    # the column written back = the per-step path's
    assert torch.equal(out["bootstrap"][1], out["two steps"][1]) and torch.equal(out["bootstrap"][2], out["two steps"][2])
    assert float(out["bootstrap"][3][2]) == float(T * N)
    # the statistics: per-workgroup partial sums added in workgroup order by the last workgroup (header v8) -- the same BITS from both
    # kernels, and from a second run of either; the arrival counter is left at zero
    assert torch.equal(out["bootstrap"][3][:3], out["two steps"][3][:3]) and float(out["bootstrap"][3][3]) == 0.0
