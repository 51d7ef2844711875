"""Pins oracle/ppo_oracle.py against vectors recorded from the reference's own algo package
(tests/golden/gen_fixtures.py): GAE known answer, the trained policy_example.pt
actor (item 2), and one full PPO iteration (act -> bootstrap -> GAE -> 8 minibatch Adam steps).  CPU only."""
import os

import numpy as np
import torch

from oracle import ppo_oracle as P

T = lambda a: torch.from_numpy(np.asarray(a))


def test_gae_known_answer(golden_dir):
    G = np.load(os.path.join(golden_dir, "gae.npz"))
    ret, adv = P.gae_returns(T(G["kat_rewards"]), T(G["kat_values"]), T(G["kat_dones"]), T(G["kat_last"]), 0.994, 0.9)
    # the literal numbers recorded in SURVEY.md §8c
    want = np.array([[1.02981997, 0.67818856], [0.0, 0.15473786], [3.31718922, 1.22412014], [1.39459991, 1.0]], np.float32)
    np.testing.assert_allclose(ret.numpy(), want, rtol=1e-6, atol=1e-7)
    want_adv = np.array([[0.28589600, -0.17526522], [-0.96987432, -0.91398144], [2.21309066, -0.02625438],
                         [0.04571262, -0.45932385]], np.float32)
    np.testing.assert_allclose(P.normalize_advantages(adv).numpy(), want_adv, rtol=2e-6, atol=1e-7)
    assert torch.equal(ret, T(G["kat_returns"]))
    assert torch.equal(P.normalize_advantages(adv), T(G["kat_adv"]))


def test_gae_seeded(golden_dir):
    G = np.load(os.path.join(golden_dir, "gae.npz"))
    ret, adv = P.gae_returns(T(G["rnd_rewards"]), T(G["rnd_values"]), T(G["rnd_dones"]), T(G["rnd_last"]),
                             float(G["gamma"]), float(G["lam"]))
    assert torch.equal(ret, T(G["rnd_returns"]))
    assert torch.equal(P.normalize_advantages(adv), T(G["rnd_adv"]))


def _policy_layers(G):
    return [(T(G["w_%d_weight" % i]), T(G["w_%d_bias" % i])) for i in (0, 2, 4, 6)]


def test_policy_example_known_answers(golden_dir):
    G = np.load(os.path.join(golden_dir, "policy_example.npz"))
    assert bytes(G["sha256"]).decode().startswith("b0c1fc24")
    layers = _policy_layers(G)
    assert [tuple(W.shape) for W, _ in layers] == [(512, 705), (256, 512), (128, 256), (12, 128)]
    y0 = P.mlp_forward(torch.zeros(1, 705), layers)
    want0 = [0.08470258, -0.02338534, 0.00571555, 0.23484124, 0.63823998, -0.22751239, -0.11294249, -0.15007278,
             0.20418212, 0.35353008, 0.00772867, -0.45297036]
    np.testing.assert_allclose(y0.numpy().ravel(), want0, rtol=1e-5, atol=1e-6)
    y1 = P.mlp_forward(torch.linspace(-1, 1, 705)[None], layers)
    want1 = [-3.45668268, -1.15925932, 0.67663300, 1.16031253, 3.21217203, 0.76701498, -4.01804018, -0.67867357,
             0.63462263, -0.34974584, 3.56635022, 1.24108696]
    np.testing.assert_allclose(y1.numpy().ravel(), want1, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(P.mlp_forward(T(G["x_rand"]), layers).numpy(), G["y_rand"], rtol=1e-5, atol=2e-5)


def _storage(G, p):
    Tn, N = G["obs"].shape[:2]
    st = dict(obs=T(G["obs"]), priv=T(G["priv"]), actions=torch.zeros(Tn, N, 12), values=torch.zeros(Tn, N, 1),
              logp=torch.zeros(Tn, N, 1), mu=torch.zeros(Tn, N, 12), sigma=torch.zeros(Tn, N, 12),
              rewards=torch.zeros(Tn, N, 1))
    for t in range(Tn):
        a, v, lp, mu, sg = P.policy_act(p, st["obs"][t], st["priv"][t], T(G["z"][t]))
        np.testing.assert_allclose(a.numpy(), G["actions"][t], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(v.numpy(), G["values"][t], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(lp.numpy(), G["logp"][t], rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(mu.numpy(), G["mu"][t], rtol=1e-6, atol=1e-6)
        assert torch.equal(sg, T(G["sigma"][t]))
        st["actions"][t], st["values"][t], st["logp"][t, :, 0], st["mu"][t], st["sigma"][t] = a, v, lp, mu, sg
        st["rewards"][t, :, 0] = P.bootstrap_rewards(T(G["rew_in"][t]), v, T(G["time_outs"][t]), 0.994)
    np.testing.assert_allclose(st["rewards"].numpy(), G["st_rewards"], rtol=1e-6, atol=1e-6)
    return st


def test_ppo_iteration_matches_reference(golden_dir):
    G = np.load(os.path.join(golden_dir, "ppo_update.npz"))
    p = P.Params.from_npz(G, "p0_")
    st = _storage(G, p)
    last_v = P.mlp_forward(T(G["last_priv"]), p.critic).squeeze(-1)
    ret, adv = P.gae_returns(st["rewards"].squeeze(-1), st["values"].squeeze(-1), T(G["done"]), last_v, 0.994, 0.9)
    np.testing.assert_allclose(ret.numpy(), G["st_returns"].squeeze(-1), rtol=1e-5, atol=1e-5)   # north_star: 1e-5 rel
    st["returns"] = ret.unsqueeze(-1)
    st["advantages"] = P.normalize_advantages(adv).unsqueeze(-1)
    np.testing.assert_allclose(st["advantages"].numpy(), G["st_advantages"], rtol=1e-4, atol=1e-5)
    opt = P.Adam(p)
    trace = []
    lr, mvl, msl = P.ppo_update(p, opt, st, T(G["perm"]), lr=1e-3, trace=trace)
    # learning-rate schedule: every one of the 8 adaptive decisions identical
    np.testing.assert_allclose([t["lr"] for t in trace], G["lrs"], rtol=1e-12)
    assert abs(lr - float(G["final_lr"])) < 1e-15
    # clipped gradients of the first minibatch vs autograd
    g0 = trace[0]["grads"]
    names = ["std"] + ["actor_%d_%s" % (i, k) for i in (0, 2, 4, 6) for k in ("weight", "bias")] + \
            ["critic_%d_%s" % (i, k) for i in (0, 2, 4, 6) for k in ("weight", "bias")]
    for nme, g in zip(names, g0.tensors()):
        ref = G["g0_" + nme]
        scale = max(np.abs(ref).max(), 1e-8)
        assert np.abs(g.numpy() - ref).max() <= 2e-5 * scale + 1e-9, nme
    np.testing.assert_allclose(mvl, float(G["mean_value_loss"]), rtol=1e-5)
    np.testing.assert_allclose(msl, float(G["mean_surrogate_loss"]), rtol=1e-4, atol=1e-7)
    for nme, t in zip(names, p.tensors()):
        np.testing.assert_allclose(t.numpy(), G["pF_" + nme], rtol=1e-4, atol=2e-6, err_msg=nme)


def test_ppo_iteration_full_width_matches_reference():
    """The same iteration at the FULL XBot-L layer widths (tests/golden/ppo_update_full.npz: the reference's PPO on inputs
    regenerated from a seed, ppo_full_case.py): oracle vs reference, fp32."""
    import ppo_full_common as F
    G, p0, I = F.load()
    p = P.Params.from_npz({"p0_" + k.replace(".", "_"): v for k, v in p0.items()}, "p0_")
    Gin = dict(obs=I["obs"], priv=I["priv"], z=I["z"], rew_in=I["rew_in"], time_outs=I["time_outs"], actions=G["actions"],
               values=G["values"], logp=G["logp"], mu=G["mu"], sigma=np.ones_like(G["mu"]), st_rewards=G["st_rewards"])
    st = _storage(Gin, p)
    last_v = P.mlp_forward(T(I["last_priv"]), p.critic).squeeze(-1)
    ret, adv = P.gae_returns(st["rewards"].squeeze(-1), st["values"].squeeze(-1), T(I["done"]), last_v, 0.994, 0.9)
    np.testing.assert_allclose(ret.numpy(), G["st_returns"].squeeze(-1), rtol=1e-5, atol=1e-5)
    st["returns"] = ret.unsqueeze(-1)
    st["advantages"] = P.normalize_advantages(adv).unsqueeze(-1)
    np.testing.assert_allclose(st["advantages"].numpy(), G["st_advantages"], rtol=1e-4, atol=1e-5)
    opt = P.Adam(p)
    trace = []
    lr, mvl, msl = P.ppo_update(p, opt, st, T(I["perm"]), lr=1e-3, trace=trace)
    np.testing.assert_allclose([t["lr"] for t in trace], G["lrs"], rtol=1e-12)
    g0 = dict(zip(F.CASE.NAMES, (g.numpy() for g in trace[0]["grads"].tensors())))
    for name, d in F.compare(G, "g0", g0).items():
        assert d["sample_max_err"] <= 2e-5 and abs(d["norm_ratio"] - 1) <= 1e-4, (name, d)
    dP = {k: t.numpy() - p0[k] for k, t in zip(F.CASE.NAMES, p.tensors())}
    for name, d in F.compare(G, "dP", dP).items():
        assert d["sample_max_err"] <= 2e-3 and abs(d["norm_ratio"] - 1) <= 1e-3, (name, d)
    np.testing.assert_allclose(mvl, float(G["mean_value_loss"]), rtol=1e-5)
    np.testing.assert_allclose(msl, float(G["mean_surrogate_loss"]), rtol=1e-4, atol=1e-7)
