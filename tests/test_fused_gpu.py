"""-m gpu: the fast paths specifically -- the fused bf16 MLP kernels (hgym_fused.hpp), the HIP-graph captured rollout
and the single-launch synthetic env step -- against the oracle, against the generic fp32 path and against their own
component-wise forms.  Tolerances are written next to each check."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import bf16_report as BR
from oracle import ppo_oracle as P
from oracle import xbot_constants as K

pytestmark = pytest.mark.gpu
NAMES = ["std"] + ["actor.%d.%s" % (i, k) for i in (0, 2, 4, 6) for k in ("weight", "bias")] + \
        ["critic.%d.%s" % (i, k) for i in (0, 2, 4, 6) for k in ("weight", "bias")]


def _net(precision, max_batch, ah=K.ACTOR_HIDDEN, ch=K.CRITIC_HIDDEN, n_obs=705, n_priv=219):
    from hgym import NetBuffers, make_net_config
    return NetBuffers(make_net_config(n_obs, n_priv, 12, ah, ch, precision, max_batch), "cuda", learning_rate=1e-3)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


@pytest.mark.parametrize("M", [1, 31, 64, 100, 4096, 5000, 20000])
def test_fused_forward_ragged_sizes_vs_oracle(M):
    """Both tile shapes of mlp_fwd_kernel (32-row rollout tiles below 16 384 rows, 64-row tiles above), batch sizes that
    are not tile multiples, with and without a row gather.  bf16 operands, fp32 accumulate: <= 1e-2 of the output scale (measured errors in the terminal summary)."""
    g = torch.Generator().manual_seed(M)
    p = P.Params.random(705, 219, 12, K.ACTOR_HIDDEN, K.CRITIC_HIDDEN, g)
    net = _net("bf16", max(M, 64))
    net.load_state_dict(dict(zip(NAMES, p.tensors())))
    obs = (torch.randn(M, 705, generator=g) * 2).clamp(-18, 18)
    priv = (torch.randn(M, 219, generator=g) * 2).clamp(-18, 18)
    mu = net.forward(0, obs.cuda())
    v = net.forward(1, priv.cuda())
    torch.cuda.synchronize()
    BR.check("fused forward vs fp32 oracle, actor mu", _rel(mu.cpu(), P.mlp_forward(obs, p.actor)))
    BR.check("fused forward vs fp32 oracle, critic value", _rel(v.cpu(), P.mlp_forward(priv, p.critic)))


def test_fused_act_matches_unfused_sampling_arithmetic():
    """PPO.act through the fused kernel (head epilogue samples and evaluates the log-prob) against the oracle's
    Gaussian arithmetic applied to the kernel's own mu: actions/log-prob/sigma exact to fp32 round-off (1e-6)."""
    g = torch.Generator().manual_seed(5)
    M = 777
    p = P.Params.random(705, 219, 12, K.ACTOR_HIDDEN, K.CRITIC_HIDDEN, g)
    p.std = torch.rand(12, generator=g) + 0.5
    net = _net("bf16", 1024)
    net.load_state_dict(dict(zip(NAMES, p.tensors())))
    obs, priv, z = torch.randn(M, 705, generator=g), torch.randn(M, 219, generator=g), torch.randn(M, 12, generator=g)
    out = net.act(obs.cuda(), priv.cuda(), z=z.cuda())
    torch.cuda.synchronize()
    mu, sg = out["mu"].cpu(), out["sigma"].cpu()
    assert torch.equal(sg, (mu * 0 + p.std))
    np.testing.assert_allclose(out["actions"].cpu().numpy(), (mu + sg * z).numpy(), rtol=1e-6, atol=1e-6)
    lp = P.gaussian_log_prob(out["actions"].cpu(), mu, sg)
    np.testing.assert_allclose(out["logp"].cpu().numpy(), lp.numpy(), rtol=1e-5, atol=1e-4)
    BR.check("fused policy step vs fp32 oracle, actor mu", _rel(mu, P.mlp_forward(obs, p.actor)))
    BR.check("fused policy step vs fp32 oracle, critic value", _rel(out["values"].cpu(), P.mlp_forward(priv, p.critic)))


BF16_OPERAND_TOL = 5e-3     # fused kernels vs the bf16-operand oracle, per tensor, rel-L2 (measured 1.1e-3 at B=333, 6.6e-4 at B=4096)


@pytest.mark.parametrize("S,B,n_obs,n_priv", [(700, 333, 705, 219), (5000, 4096, 705, 219), (900, 517, 4 * 47, 2 * 73), (300, 64, 47, 73)])
def test_fused_grad_vs_oracle_per_tensor(S, B, n_obs, n_priv):
    """Un-clipped gradient of one minibatch, ragged batch, fused bf16 kernels (mlp_fwd / mlp_bwd / dw with the transpose
    read) vs the oracle's hand-written fp32 backward, per parameter tensor in relative L2 norm (tolerances below), whole
    gradient cosine > 0.995, scalar losses within 1e-2.  Input widths: XBot-L's 15 x 47 / 3 x 73 and two other frame stacks
    (humanoid_config.py:40-45: 4 / 2 frames, 1 / 1 frame) -- widths that are not multiples of the first layer's k-step."""
    from hgym import make_ppo_config, make_batch
    g = torch.Generator().manual_seed(S)
    p = P.Params.random(n_obs, n_priv, 12, K.ACTOR_HIDDEN, K.CRITIC_HIDDEN, g)
    p.std = torch.rand(12, generator=g) * 0.5 + 0.75
    net = _net("bf16", max(B, 512), n_obs=n_obs, n_priv=n_priv)
    net.load_state_dict(dict(zip(NAMES, p.tensors())))
    obs, priv = torch.randn(S, n_obs, generator=g), torch.randn(S, n_priv, generator=g)
    act, mu_o = torch.randn(S, 12, generator=g), torch.randn(S, 12, generator=g) * 0.3
    sg_o = torch.rand(S, 12, generator=g) * 0.5 + 0.75
    val, adv, ret = torch.randn(S, generator=g), torch.randn(S, generator=g), torch.randn(S, generator=g)
    with torch.no_grad():
        mu_now = P.mlp_forward(obs, p.actor)
    lp_o = P.gaussian_log_prob(act, mu_now, mu_now * 0 + p.std) + torch.randn(S, generator=g) * 0.3
    idx = torch.randperm(S, generator=g)[:B].contiguous()
    out = P.ppo_loss_and_grads(p, obs[idx], priv[idx], act[idx], val[idx], adv[idx], ret[idx], lp_o[idx], mu_o[idx], sg_o[idx])
    c = lambda t: t.cuda().contiguous()
    keep = [c(obs), c(priv), c(act), c(val), c(adv), c(ret), c(lp_o), c(mu_o), c(sg_o), c(idx)]
    for rep in range(2):            # twice: the second call must not depend on anything the first left behind
        net.ppo_grad(make_ppo_config(), make_batch(*keep))
    torch.cuda.synchronize()
    gv = net.grad_views()
    num = da = db = 0.0
    for k, ref in zip(NAMES, out["grads"].tensors()):
        a, b = gv[k].cpu().double().flatten(), ref.double().flatten()
        l2 = float((a - b).norm() / b.norm().clamp_min(1e-30))
        # critic: pure bf16 operand rounding (measured 2e-3..1.1e-2 on the weight matrices; the 1-element head bias is a
        # single sum of bf16-rounded dZ over B samples and lands at up to 2.1e-2 for B=333; 3.2e-2 on the 128-element critic.4.bias at
        # B=517 with 146 inputs -- hence the 4e-2 bound; the statement about the KERNELS is the 5e-3 check further down).  actor / std: the surrogate gradient carries the factor
        # ratio = exp(logp - logp_old) and the clip indicator, both functions of mu, so the ~1e-2 bf16 error of mu moves
        # samples across the clip boundary (measured 6e-2..8e-2, identical for the fused and the generic bf16 path;
        # the fp32 path is exact to 1e-5, tests/test_net_gpu.py)
        tol = 4e-2 if k.startswith("critic") else 0.12
        # (a 64-row batch is one tile: a handful of samples crossing the clip boundary is most of its std gradient -- the fp32
        # comparison says nothing there, the bf16-operand comparison below does)
        assert B < 300 or (l2 <= tol and _rel(a.numpy(), b.numpy()) <= 0.25), (k, l2, _rel(a.numpy(), b.numpy()))
        num += float(a @ b); da += float(a @ a); db += float(b @ b)
    assert num / (da ** 0.5 * db ** 0.5) > (0.995 if B >= 300 else 0.98)
    # ... and against the oracle evaluated on bf16 OPERANDS (quant=bf16_round: inputs, weights, activations and stored dZ rounded
    # where the kernels round them, fp32 accumulation): the clip indicators now agree, what is left is accumulation order and the
    # hardware exp -- an order of magnitude tighter, which is the statement about the KERNELS
    outq = P.ppo_loss_and_grads(p, obs[idx], priv[idx], act[idx], val[idx], adv[idx], ret[idx], lp_o[idx], mu_o[idx], sg_o[idx],
                                quant=P.bf16_round)
    worst = 0.0
    for k, ref in zip(NAMES, outq["grads"].tensors()):
        a, b = gv[k].cpu().double().flatten(), ref.double().flatten()
        l2 = float((a - b).norm() / b.norm().clamp_min(1e-30))
        worst = max(worst, l2)
        assert l2 <= BF16_OPERAND_TOL, (k, l2)
    print("fused bf16 gradient vs bf16-operand oracle: worst per-tensor rel-L2 %.3e (S=%d, B=%d)" % (worst, S, B))
    assert _rel(net.forward(0, c(obs[idx][:64])).cpu(), P.mlp_forward(obs[idx][:64], p.actor, quant=P.bf16_round)) <= 2e-3
    opt = net.opt_state.cpu()
    np.testing.assert_allclose(float(opt[8]), float(out["kl"]), rtol=2e-2, atol=1e-4)
    np.testing.assert_allclose(float(opt[4]) / 2, float(out["value_loss"]), rtol=1e-2)


def test_gradient_norm_of_the_gradient_call_has_the_same_bits_every_time():
    """hgym_ppo_grad's squared gradient norm (opt_state[9]; one rank: what hgym_ppo_apply clips with) reaches its word through 1 632 fp64
    atomics in arrival order.  reduce_slabs_kernel rounds every partial to a common quantum (2^-46) first, so the additions are exact and the
    total has the same bits whatever the order (while it stays below 128): six gradient calls on the same minibatch, on fresh nets, agree bit
    for bit in [9], and [9] is the fp64 squared norm of the gradient vector to the rounding's bound (1.2e-11 absolute)."""
    from hgym import make_ppo_config, make_batch
    g = torch.Generator().manual_seed(11)
    S = B = 4096
    p = P.Params.random(705, 219, 12, K.ACTOR_HIDDEN, K.CRITIC_HIDDEN, g)
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(12)
    r = lambda *sh: torch.randn(*sh, device=dev, generator=gen)
    obs, priv = r(S, 705), r(S, 219)
    act, mu_o, sg_o = r(S, 12), r(S, 12) * 0.3, torch.ones(S, 12, device=dev)
    val, adv, ret = r(S), r(S), r(S)
    lp_o = -12.0 + r(S)
    idx = torch.randperm(S, device=dev, generator=gen).contiguous()
    seen = []
    for rep in range(6):
        net = _net("bf16", B)
        net.load_state_dict(dict(zip(NAMES, p.tensors())))
        net.ppo_grad(make_ppo_config(), make_batch(obs, priv, act, val, adv, ret, lp_o, mu_o, sg_o, idx))
        torch.cuda.synchronize()
        sq = float(net.opt_state[9])
        want = float(net.grads[:net.P].double().pow(2).sum())
        assert 0.0 < sq < 128.0 and abs(sq - want) <= 2e-11 + 1e-13 * want, (sq, want)
        seen.append((net.opt_state[9].clone(), net.grads.clone()))
        del net
    for a, b in seen[1:]:
        assert torch.equal(b, seen[0][1])                                      # (the gradient itself: fixed-order slab sums)
        assert torch.equal(a, seen[0][0]), (float(a), float(seen[0][0]))       # ... and its squared norm


def test_fused_and_generic_bf16_paths_agree(monkeypatch):
    """The same bf16 configuration through the fused kernels and through the generic MFMA GEMM path (HGYM_NO_FUSED=1):
    forward within 1e-2, gradient cosine > 0.999 -- the two share no kernel except the loss."""
    from hgym import make_ppo_config, make_batch
    g = torch.Generator().manual_seed(2)
    S = B = 2048
    p = P.Params.random(705, 219, 12, K.ACTOR_HIDDEN, K.CRITIC_HIDDEN, g)
    dev = "cuda"
    obs, priv = torch.randn(S, 705, device=dev), torch.randn(S, 219, device=dev)
    act, mu_o, sg_o = torch.randn(S, 12, device=dev), torch.randn(S, 12, device=dev) * 0.3, torch.ones(S, 12, device=dev)
    val, adv, ret = torch.randn(S, device=dev), torch.randn(S, device=dev), torch.randn(S, device=dev)
    lp_o = -12.0 + torch.randn(S, device=dev)
    idx = torch.randperm(S, device=dev).contiguous()
    res = {}
    for tag in ("fused", "generic"):
        if tag == "generic":
            monkeypatch.setenv("HGYM_NO_FUSED", "1")
        net = _net("bf16", B)
        net.load_state_dict(dict(zip(NAMES, p.tensors())))
        mu = net.forward(0, obs)
        net.ppo_grad(make_ppo_config(), make_batch(obs, priv, act, val, adv, ret, lp_o, mu_o, sg_o, idx))
        torch.cuda.synchronize()
        res[tag] = (mu.clone(), net.grads.clone())
        del net
    monkeypatch.delenv("HGYM_NO_FUSED")
    assert _rel(res["fused"][0].cpu(), res["generic"][0].cpu()) <= 1e-2
    a, b = res["fused"][1].double(), res["generic"][1].double()
    assert float(a @ b) / float(a.norm() * b.norm()) > 0.999


def _runner(num_envs, seed):
    from humanoid.envs import task_registry
    from humanoid.utils import get_args
    args = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", str(num_envs), "--seed", str(seed)])
    # like the reference, make_env seeds from the REGISTERED train cfg and --seed only reaches it in make_alg_runner
    # (utils/task_registry.py:88-112): pin it so both constructions see the same seed
    task_registry.train_cfgs[args.task].seed = seed
    env, _ = task_registry.make_env(name=args.task, args=args)
    runner, _ = task_registry.make_alg_runner(env=env, name=args.task, args=args, log_root=None)
    return runner


def test_graph_replay_equals_eager_rollout(monkeypatch):
    """Three learning iterations with the rollout captured in a HIP graph (iteration 1 eager warm-up, 2 capture + replay,
    3 replay) and three fully eager iterations from the same seeds: rollout storage and parameters bit-identical
    (same kernels, same device-resident counters; only the launch mechanism differs).  The eager leg also turns the
    env-side transition sink off (HGYM_ENV_SINK=0: PPO.process_env_step launches hgym_store_step itself and PPO.act bumps
    the sampling step), so the two ways of storing the scalar columns are compared bit for bit as well."""
    from humanoid.algo import PPO
    PPO.precision = "bf16"
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("HGYM_GRAPH", mode)
        monkeypatch.setenv("HGYM_ENV_SINK", mode)
        torch.manual_seed(1234)
        np.random.seed(1234)
        r = _runner(256, 77)
        perms = []
        real = torch.randperm

        def fixed(n, *a, **k):      # the minibatch permutation comes from torch's global generator: pin it
            gen = torch.Generator().manual_seed(len(perms))
            perms.append(n)
            return real(n, generator=gen).to(k.get("device", "cpu"))
        monkeypatch.setattr(torch, "randperm", fixed)
        r.env.episode_length_buf = torch.arange(256, device="cuda") * 9
        r.learn(num_learning_iterations=3, init_at_random_ep_len=False)
        torch.cuda.synchronize()
        monkeypatch.setattr(torch, "randperm", real)
        assert (r._graph is not None) == (mode == "1")
        st = r.alg.storage
        outs[mode] = (r.alg.net.params.clone(), st._obs_all.clone(), st.rewards.clone(), st.actions.clone(), st.values.clone(),
                      st.dones.clone(), st.returns.clone(), r.alg._sample_step.clone())
        del r
    for a, b in zip(outs["1"], outs["0"]):
        assert torch.equal(a, b)


def test_captured_update_equals_eager_update(monkeypatch):
    """Header v9 / round 6: compute_returns() + update() replayed from a second HIP graph (the whole iteration = two graph launches) against the
    same five iterations with the update's ~45 launches issued from Python (HGYM_GRAPH_UPDATE=0): parameters, Adam moments, optimiser
    scalars (learning-rate decisions, step count, loss sums), the last permutation and the storage bit-identical; the permutation's draw
    number on the device equals the host's count, and a sixth and seventh iteration through a SECOND learn() call (same graphs, still
    valid) keep it so."""
    from humanoid.algo import PPO
    PPO.precision = "bf16"
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("HGYM_GRAPH_UPDATE", mode)
        torch.manual_seed(4321)
        np.random.seed(4321)
        r = _runner(256, 78)
        r.env.episode_length_buf = torch.arange(256, device="cuda") * 7
        r.learn(num_learning_iterations=5, init_at_random_ep_len=False)
        r.learn(num_learning_iterations=2, init_at_random_ep_len=False)
        torch.cuda.synchronize()
        alg, st = r.alg, r.alg.storage
        assert (r._update_graph is not None) == (mode == "1") and r._graph is not None
        assert alg._perm_draws == 7 and int(alg._perm_draws_dev) == 7
        assert int(alg.net.opt_state[1]) == 7 * alg.num_learning_epochs * alg.num_mini_batches
        opt = alg.net.opt_state.clone()
        # [9], the squared gradient norm, arrives through fp64 atomics in any order: reduce_slabs_kernel rounds the partials to a common
        # quantum, which makes the sum exact -- the same bits in every run -- while it is below 128 (the kernel's comment); beyond, its last
        # bits may differ under EITHER launch mechanism and only the norm it rounds to ([6]) is compared
        if float(opt[9]) >= 128.0:
            opt[9] = 0.0
        outs[mode] = (alg.net.params.clone(), alg.net.adam_m.clone(), alg.net.adam_v.clone(), opt, st._perm.clone(),
                      st._obs_all.clone(), st.returns.clone(), st.advantages.clone(), alg._sample_step.clone())
        del r
    names = ("params", "adam_m", "adam_v", "opt_state", "perm", "obs_all", "returns", "advantages", "sample_step")
    for nm, a, b in zip(names, outs["1"], outs["0"]):
        assert torch.equal(a, b), (nm, (a != b).nonzero()[:8].tolist(), a.flatten()[:16].tolist(), b.flatten()[:16].tolist())


@pytest.mark.parametrize("graph", ["0", "1"])
def test_fused_rollout_step_equals_act_then_step(monkeypatch, graph):
    """hgym_rollout_step (ONE launch per vec-step: actor tile + env step of the same 32 envs, critic tile, the previous step's
    finaliser) against hgym_policy_act_fin followed by hgym_env_step_synth (two launches): after two learning iterations from
    the same seeds the rollout storage, the env state (state fields, sim tensors, history rings, episode lengths, counters,
    extras) and the parameters are bit-identical -- the fused kernel runs the same source for every phase, and its ping-pong
    step counters / per-parity accumulators reproduce the sequence the device-resident counters go through."""
    from humanoid.algo import PPO
    PPO.precision = "bf16"
    monkeypatch.setenv("HGYM_GRAPH", graph)
    outs = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("HGYM_FUSE_ROLLOUT", fuse)
        torch.manual_seed(4321)
        np.random.seed(4321)
        r = _runner(512, 31)
        r.env.episode_length_buf = (torch.arange(512, device="cuda") * 37) % 2400       # time-outs, command resampling inside the window
        r.env._buf.counters[0] = 390                                                   # a push (every 400 steps) too
        n_iter = 3 if graph == "1" else 2
        r.learn(num_learning_iterations=n_iter, init_at_random_ep_len=False)
        torch.cuda.synchronize()
        st, b = r.alg.storage, r.env._buf
        assert int(st.dones.sum()) > 0
        outs[fuse] = dict(params=r.alg.net.params.clone(), obs=st._obs_all.clone(), priv=st._priv_all.clone(), rewards=st.rewards.clone(),
                          actions=st.actions.clone(), values=st.values.clone(), logp=st.actions_log_prob.clone(), mu=st.mu.clone(),
                          dones=st.dones.clone(), returns=st.returns.clone(), sample_step=r.alg._sample_step.clone(),
                          state=b._state.clone(), root=b.root.clone(), dof_pos=b.dof_pos.clone(), dof_vel=b.dof_vel.clone(),
                          contact=b.contact.clone(), rigid=b.rigid.clone(), obs_ring=b.obs_ring.clone(), priv_ring=b.priv_ring.clone(),
                          ep_len=b.episode_length.clone(), counters=b.counters.clone(), rew=b.rew.clone(), reset=b.reset.clone(),
                          time_out=b.time_out.clone(), extras_time_outs=b.extras_time_outs.clone(), episode_acc=b.episode_acc.clone(),
                          env_obs=r.env.obs_buf.clone())
        fused_ran = os.environ.get("HGYM_FUSE_ROLLOUT") == "1" and r.env.rollout_fused_supported(r.alg.net)
        assert fused_ran == (fuse == "1")
        extras = b.extras_episode.clone()
        outs[fuse]["extras_episode"] = extras
        del r
    for k in outs["1"]:
        if k == "extras_episode":        # means over resetting envs: fp32 atomics, order-dependent in the last bit
            np.testing.assert_allclose(outs["1"][k].cpu().numpy(), outs["0"][k].cpu().numpy(), rtol=1e-5, atol=1e-9)
        else:
            assert torch.equal(outs["1"][k], outs["0"][k]), k


@pytest.mark.parametrize("use_ref_actions", [0, 1])
def test_single_launch_env_step_equals_componentwise_calls(use_ref_actions):
    """(use_ref_actions = 1: also the in-place `actions += ref_action` on the caller's tensor)
    hgym_env_step_synth (one launch: draws, joints, per-env phase, history) against the same step issued as
    hgym_pre_physics -> hgym_synth_physics -> hgym_post_physics (three independent kernels, internal Philox): every
    output and every piece of state bit-identical over 25 steps including resets, pushes and command resampling."""
    from hgym import EnvBuffers, default_env_config, _lib as L
    N = 1000                    # not a multiple of the 16-env workgroup: exercises the partial last block
    bufs = []
    for k in range(2):
        cfg = default_env_config(N, seed=99)
        cfg.use_ref_actions = use_ref_actions
        b = EnvBuffers(cfg, "cuda")
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(L.lib.hgym_env_prime(C.byref(cfg), C.byref(b.sim_struct()), C.byref(b.state_struct()), C.byref(b.out_struct()),
                                     C.byref(b.noise_struct()), s))
        b.episode_length.copy_((torch.arange(N) * 7 % 2400).cuda())
        b.counters[0] = 395                          # a push (every 400 steps) falls inside the window
        bufs.append((cfg, b))
    g = torch.Generator().manual_seed(3)
    for t in range(25):
        a = (torch.randn(N, 12, generator=g) * 3).cuda()
        a2 = a.clone()
        cfg, b = bufs[0]
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(L.lib.hgym_env_step_synth(C.byref(cfg), C.byref(b.sim_struct()), C.byref(b.state_struct()), C.byref(b.out_struct()),
                                          L.fptr(a), s))
        cfg, b = bufs[1]
        nz = b.noise_struct()
        L.check(L.lib.hgym_pre_physics(C.byref(cfg), C.byref(b.state_struct()), L.fptr(a2), C.byref(nz), s))
        L.check(L.lib.hgym_synth_physics(C.byref(cfg), C.byref(b.sim_struct()), C.byref(b.state_struct()), s))
        L.check(L.lib.hgym_post_physics(C.byref(cfg), C.byref(b.sim_struct()), C.byref(b.state_struct()), C.byref(b.out_struct()),
                                        C.byref(nz), s))
        torch.cuda.synchronize()
        x, y = bufs[0][1], bufs[1][1]
        assert torch.equal(a, a2), t                       # the caller's tensor: untouched, or the same in-place update
        for name in ("obs", "priv_obs", "rew", "reset", "time_out", "episode_length", "root", "dof_pos", "dof_vel", "contact", "rigid",
                     "_state", "obs_ring", "priv_ring", "extras_time_outs", "counters"):
            assert torch.equal(getattr(x, name), getattr(y, name)), (t, name)
        # means over the resetting envs are accumulated with fp32 atomics (order-dependent in the last bit)
        np.testing.assert_allclose(x.extras_episode.cpu().numpy(), y.extras_episode.cpu().numpy(), rtol=1e-5, atol=1e-9)
    assert int(bufs[0][1].reset.sum()) >= 0


def test_adam_writes_the_same_operand_copies_as_the_refresh_from_master():
    """adam_kernel keeps the fragment-major bf16 operand copies (W and W^T images) of every weight matrix current;
    hgym_net_sync_shadow re-derives them from the fp32 master parameters.  After clipped Adam steps on random gradients
    the two must be byte-identical (every edge: 705, 219, 12 and 1 are not multiples of the 16/32-wide fragments)."""
    from hgym import make_ppo_config
    net = _net("bf16", 512)
    g = torch.Generator(device="cuda").manual_seed(3)
    for k, v in net.views.items():
        v.copy_(torch.randn(v.shape, device="cuda", generator=g) * 0.05)
    net.sync_shadow()
    net.grads.copy_(torch.randn(net.P, device="cuda", generator=g))
    for _ in range(2):
        net.ppo_apply(make_ppo_config())
    torch.cuda.synchronize()
    after_adam = net.workspace.clone()
    net.sync_shadow()
    torch.cuda.synchronize()
    assert torch.equal(after_adam, net.workspace)


def test_rollout_step_refuses_foreign_net_shapes_on_the_c_side():
    """include/hgym.h's "Supported:" list is enforced by hgym_rollout_step itself, not only by the Python wrapper: a net whose
    observation width is not XBot-L's 705 would make the env part write 705-wide rows the policy tiles then read with another
    leading dimension.  HGYM_E_UNSUPPORTED before anything is launched."""
    from hgym import EnvBuffers, NetBuffers, default_env_config, make_net_config, _lib as L
    N = 64
    cfg = default_env_config(N)
    buf = EnvBuffers(cfg, "cuda")
    net = NetBuffers(make_net_config(700, 219, 12, K.ACTOR_HIDDEN, K.CRITIC_HIDDEN, "bf16", 64), "cuda")
    z = lambda *s: torch.zeros(*s, device="cuda")
    val = z(N)
    sink = dict(values=val, rewards=z(N), dones=torch.zeros(N, dtype=torch.uint8, device="cuda"),
                step=torch.zeros(1, dtype=torch.int64, device="cuda"), gamma=0.99)
    out = buf.out_struct(z(N, 705), z(N, 219), sink, True)
    before = buf._state.clone()
    rc = L.lib.hgym_rollout_step(C.byref(net.cfg), C.byref(net.struct), C.byref(cfg), C.byref(buf.sim_struct()), C.byref(buf.state_struct()),
                                 C.byref(out), None, L.fptr(z(N, 700)), L.fptr(z(N, 219)), 1, L.fptr(z(N, 12)), L.fptr(z(N, 12)),
                                 L.fptr(z(N, 12)), L.fptr(z(N)), L.fptr(val), C.c_void_p(buf.rollout_scratch.data_ptr()), 0, None,
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == -4 and b"705" in L.lib.hgym_last_error()
    assert torch.equal(before, buf._state)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two devices in one process")
def test_two_devices_in_one_process_share_nothing():
    """The library keeps one piece of state between calls -- how much dynamic LDS was reserved per (kernel, device).  One process
    driving two devices (one runner each, alternating) must get, on each device, exactly what that runner produces alone."""
    from humanoid.algo import PPO
    PPO.precision = "bf16"
    outs = {}
    for mode in ("alone", "interleaved"):
        rs = []
        for d in (0, 1):
            torch.cuda.set_device(d)
            torch.manual_seed(7)
            np.random.seed(7)
            from humanoid.envs import task_registry
            from humanoid.utils import get_args
            dev = "cuda:%d" % d
            args = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", "256", "--seed", "3", "--sim_device", dev, "--rl_device", dev])
            task_registry.train_cfgs[args.task].seed = 3
            env, _ = task_registry.make_env(name=args.task, args=args)
            r, _ = task_registry.make_alg_runner(env=env, name=args.task, args=args, log_root=None)
            rs.append(r)
            if mode == "alone":
                r.learn(num_learning_iterations=2, init_at_random_ep_len=False)
        if mode == "interleaved":
            for _ in range(2):
                for d, r in enumerate(rs):
                    torch.cuda.set_device(d)
                    r.learn(num_learning_iterations=1, init_at_random_ep_len=False)
        for d, r in enumerate(rs):
            torch.cuda.set_device(d)
            torch.cuda.synchronize()
            outs[(mode, d)] = r.alg.net.params.cpu().clone()
        del rs
    torch.cuda.set_device(0)
    for d in (0, 1):
        assert torch.equal(outs[("alone", d)], outs[("interleaved", d)]), d
    assert torch.equal(outs[("alone", 0)], outs[("alone", 1)])


# ------------------------------------------------------------------------------------------------ bf16 observation shadow
@pytest.mark.parametrize("M", [64, 100, 4096, 5000, 20000])
def test_policy_launch_leaves_the_bf16_shadow_of_its_input_rows(M):
    """HgymObsShadow: hgym_policy_act with a shadow argument stores, row m, the bf16 (round-to-nearest-even) of obs[m] / priv[m],
    zero in the pad columns -- through both tile shapes (32-row tiles up to 4096 rows, 64-row tiles beyond) and ragged row counts;
    the launch's other outputs do not change."""
    g = torch.Generator().manual_seed(M)
    p = P.Params.random(705, 219, 12, K.ACTOR_HIDDEN, K.CRITIC_HIDDEN, g)
    net = _net("bf16", max(M, 64))
    net.load_state_dict(dict(zip(NAMES, p.tensors())))
    assert (net.shadow_ld(0), net.shadow_ld(1)) == (768, 256)
    obs = (torch.randn(M, 705, generator=g) * 3).clamp(-18, 18).cuda()
    priv = (torch.randn(M, 219, generator=g) * 3).clamp(-18, 18).cuda()
    z = torch.randn(M, 12, generator=g).cuda()
    so = torch.full((M, 768), 7.0, dtype=torch.bfloat16, device="cuda")
    sp = torch.full((M, 256), 7.0, dtype=torch.bfloat16, device="cuda")
    a = net.act(obs, priv, z=z)
    b = net.act(obs, priv, z=z, shadow=(so, sp))
    torch.cuda.synchronize()
    for k in ("actions", "mu", "sigma", "logp", "values"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(so[:, :705], obs.to(torch.bfloat16)) and torch.equal(sp[:, :219], priv.to(torch.bfloat16))
    assert float(so[:, 705:].float().abs().max()) == 0.0 and float(sp[:, 219:].float().abs().max()) == 0.0


@pytest.mark.parametrize("S,B", [(700, 333), (5000, 4096), (61440, 61440)])
def test_update_from_the_bf16_shadow_equals_update_from_fp32_rows(S, B):
    """hgym_ppo_grad with HgymBatch.obs_bf16 / priv_bf16 (first layer gathers 2-byte elements, the weight-gradient kernel
    gathers its first-layer operand from the same shadow by index, no operand copy is written) against the same call on the
    fp32 rows (gather + convert + X0 copy): the shadow holds exactly the bf16 the other path forms on the fly, every product
    sees the same operands in the same order -- the gradients are bit-identical.  Ragged batches and the BASELINE minibatch."""
    from hgym import make_ppo_config, make_batch
    g = torch.Generator().manual_seed(S + B)
    p = P.Params.random(705, 219, 12, K.ACTOR_HIDDEN, K.CRITIC_HIDDEN, g)
    net = _net("bf16", max(B, 512))
    net.load_state_dict(dict(zip(NAMES, p.tensors())))
    dev = "cuda"
    gd = torch.Generator(device=dev).manual_seed(S)
    obs, priv = torch.randn(S, 705, device=dev, generator=gd).clamp_(-18, 18), torch.randn(S, 219, device=dev, generator=gd).clamp_(-18, 18)
    act, mu_o = torch.randn(S, 12, device=dev, generator=gd), torch.randn(S, 12, device=dev, generator=gd) * 0.3
    sg_o = torch.rand(S, 12, device=dev, generator=gd) * 0.5 + 0.75
    val, adv, ret = (torch.randn(S, device=dev, generator=gd) for _ in range(3))
    lp_o = -12.0 + torch.randn(S, device=dev, generator=gd)
    idx = torch.randperm(S, device=dev, generator=gd)[:B].contiguous()
    so = torch.zeros(S, 768, dtype=torch.bfloat16, device=dev)
    sp = torch.zeros(S, 256, dtype=torch.bfloat16, device=dev)
    so[:, :705] = obs.to(torch.bfloat16)
    sp[:, :219] = priv.to(torch.bfloat16)
    cols = (obs, priv, act, val, adv, ret, lp_o, mu_o, sg_o, idx)
    net.ppo_grad(make_ppo_config(), make_batch(*cols))
    torch.cuda.synchronize()
    want, want_opt = net.grads_ext.clone(), net.opt_state.clone()
    net.grads_ext.zero_()
    net.opt_state[2:10] = 0.0
    net.ppo_grad(make_ppo_config(), make_batch(*cols, obs_bf16=so, priv_bf16=sp))
    torch.cuda.synchronize()
    assert torch.equal(net.grads_ext, want)
    assert torch.equal(net.opt_state[2:9], want_opt[2:9])          # loss sums, KL, counters
    # the squared gradient norm is accumulated with fp64 atomics across workgroups: order-dependent in the last bits
    np.testing.assert_allclose(float(net.opt_state[9]), float(want_opt[9]), rtol=1e-12)
    # ... and in two halves (the data-parallel update's buckets)
    net.grads_ext.zero_()
    for part in (0, 1):
        net.ppo_grad_part(make_ppo_config(), make_batch(*cols, obs_bf16=so, priv_bf16=sp), part)
    torch.cuda.synchronize()
    assert torch.equal(net.grads_ext, want)


@pytest.mark.parametrize("kb0", ["12", "20", "4"])
def test_first_layer_carried_across_launches_changes_nothing(monkeypatch, kb0):
    """HgymEnvOut.l0_ahead / l0_ready (csrc/hgym_fused.hpp: L0Part / l0_partial_ahead): the critic workgroups of launch t form kb0 of the 24
    k-steps of the ACTOR's first layer for the rows of step t + 1 (step t's rows shifted by one frame), launch t + 1's actor tile starts
    from those fp32 partial sums, reads only the remaining columns, and drops the sums of rows whose env was reset in between.  Three
    learning iterations (eager, capture + replay, replay; 1024 envs x 60 steps) with the hand-over and
    without it (HGYM_L0_AHEAD=0) from the same seeds: rollout storage (observations, actions, log-probs, values, rewards, dones), the
    bf16 shadows and the parameters must be BIT-identical -- same fragments, same k order, same accumulator chain.
    Reference: /root/reference/humanoid/algo/ppo/actor_critic.py:53-80 (the actor's first Linear), on_policy_runner.py:129-141."""
    from humanoid.algo import PPO
    PPO.precision = "bf16"
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("HGYM_L0_AHEAD", mode)
        monkeypatch.setenv("HGYM_L0_KB0", kb0)
        torch.manual_seed(11)
        np.random.seed(11)
        r = _runner(1024, 31)
        assert r.env._l0_ahead == (mode == "1")
        r.learn(num_learning_iterations=3, init_at_random_ep_len=True)
        torch.cuda.synchronize()
        st = r.alg.storage
        used = r.env._buf._l0_partial is not None and bool((r.env._buf._l0_partial != 0).any())
        assert used == (mode == "1")                  # the hand-over buffers were really written (or never allocated)
        outs[mode] = dict(params=r.alg.net.params.clone(), obs=st._obs_all.clone(), act=st.actions.clone(), logp=st.actions_log_prob.clone(),
                          val=st.values.clone(), rew=st.rewards.clone(), dones=st.dones.clone(), sh=st._obs_bf16.clone(),
                          resets=int(st.dones.sum()))
        del r
    a, b = outs["1"], outs["0"]
    assert a["resets"] == b["resets"] and a["resets"] >= 10
    for k in ("obs", "act", "logp", "val", "rew", "dones", "sh", "params"):
        assert torch.equal(a[k], b[k]), k
    assert torch.isfinite(a["params"]).all()


def test_deferred_values_rollout_equals_inline_rollout(monkeypatch):
    """OnPolicyRunner.learn with the critic deferred (HGYM_ROLLOUT_CRITIC=deferred: launches without critic tiles + ONE critic pass behind
    the last step, the path of more than 4096 envs per GPU) against the inline form from the same seeds, 2048 envs, three iterations
    (eager, capture + replay, replay).  The first rollout must agree bit for bit in everything the actor and the env produce --
    observations, actions, log-probs, dones -- and to the bf16 path's tolerance in what depends on V(s_t): values, bootstrapped
    rewards, returns; both runs train (finite, moving parameters); the deferred run's graph replays end with the critic pass."""
    from humanoid.algo import PPO
    PPO.precision = "bf16"
    outs = {}
    for mode in ("inline", "deferred"):
        monkeypatch.setenv("HGYM_ROLLOUT_CRITIC", mode)
        torch.manual_seed(5)
        np.random.seed(5)
        r = _runner(2048, 31)
        assert r.env.rollout_fused_mode(r.alg.net) == mode
        p0 = r.alg.net.params.clone()
        snap = {}
        orig = r.alg.compute_returns

        def spy(last, orig=orig, snap=snap, r=r):
            orig(last)
            if not snap:
                st = r.alg.storage
                torch.cuda.synchronize()
                snap.update(obs=st.observations.clone(), act=st.actions.clone(), logp=st.actions_log_prob.clone(), dones=st.dones.clone(),
                            val=st.values.clone(), rew=st.rewards.clone(), ret=st.returns.clone(), priv_sh=st._priv_bf16.clone(),
                            valid=list(st.shadow_valid))
        r.alg.compute_returns = spy
        r.learn(num_learning_iterations=3, init_at_random_ep_len=True)
        torch.cuda.synchronize()
        assert r._graph is not None and torch.isfinite(r.alg.net.params).all() and not torch.equal(r.alg.net.params, p0)
        assert int(r.alg.net.opt_state[1]) == 24 and float(r.alg.storage.values.abs().max()) > 0
        outs[mode] = snap
        del r
    a, b = outs["inline"], outs["deferred"]
    for k in ("obs", "act", "logp", "dones", "priv_sh"):
        assert torch.equal(a[k], b[k]), k
    assert all(a["valid"]) and all(b["valid"]) and int(a["dones"].sum()) >= 10
    for k in ("val", "rew", "ret"):       # the two tile shapes of the critic (32 rows inline, 64 rows one pass) round differently
        BR.check("deferred-values rollout vs inline rollout, %s" % k, float((a[k] - b[k]).abs().max() / a[k].abs().max()))


def test_runner_update_reads_the_shadow_the_rollout_wrote(monkeypatch):
    """End to end: two learning iterations with the shadow (default) and without (HGYM_SHADOW=0) from the same seeds end in
    bit-identical parameters, and the shadow slots hold the bf16 of the stored observation rows."""
    from humanoid.algo import PPO
    PPO.precision = "bf16"
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("HGYM_SHADOW", mode)
        torch.manual_seed(5)
        np.random.seed(5)
        r = _runner(512, 23)
        r.learn(num_learning_iterations=3, init_at_random_ep_len=True)
        torch.cuda.synchronize()
        st = r.alg.storage
        assert (st._obs_bf16 is not None) == (mode == "1")
        if mode == "1":
            # slot 0 was rewritten by storage.clear() after its launch; the others still hold what their launches read
            assert torch.equal(st._obs_bf16[1:, :, :705], st.observations[1:].to(torch.bfloat16))
            assert torch.equal(st._priv_bf16[1:, :, :219], st.privileged_observations[1:].to(torch.bfloat16))
        outs[mode] = r.alg.net.params.clone()
        del r
    assert torch.equal(outs["1"], outs["0"])
