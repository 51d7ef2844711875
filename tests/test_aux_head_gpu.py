"""-m gpu: the auxiliary (denoising) head of BASELINE configs[4] / SURVEY.md 8f item 4.  The reference ships NO code for it
(README.md:113), so parity is unpinned by construction; the checks are against a plain PyTorch fp32 autograd model of the design
in DESIGN.md section 9: an MLP obs -> hidden -> 73 that regresses the newest privileged frame (columns 146..218 of the
privileged row) under coef * MSE, trained jointly with PPO (one flat gradient, one clip, one Adam)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
AUX_OUT, AUX_OFF = 73, 146
# two head shapes: [256, 128, 64] falls outside the fused kernels' layer shapes and runs layer by layer (generic GEMMs);
# [512, 256, 128] (the humanoid_dwl_ppo task's) runs through the fused forward / backward / weight-gradient kernels in bf16
AUX_SHAPES = {"generic": [256, 128, 64], "fused": [512, 256, 128]}


def _nets(precision, B, AUX_H=AUX_SHAPES["generic"]):
    from hgym import NetBuffers, make_net_config
    torch.manual_seed(3)
    out = []
    for aux in (False, True):
        cfg = make_net_config(705, 219, 12, [512, 256, 128], [768, 256, 128], precision, B,
                              aux_hidden=AUX_H if aux else None, aux_out=AUX_OUT if aux else 0, aux_target_offset=AUX_OFF)
        out.append(NetBuffers(cfg, "cuda", learning_rate=1e-3))
    plain, full = out
    g = torch.Generator(device="cuda").manual_seed(1)
    for k, v in full.views.items():
        v.copy_(torch.randn(v.shape, device="cuda", generator=g) * (0.05 if v.dim() > 1 else 0.01))
    full.views["std"].fill_(1.0)
    for k, v in plain.views.items():
        v.copy_(full.views[k])
    plain.sync_shadow()
    full.sync_shadow()
    return plain, full


def _batch(S, B):
    from hgym import make_batch
    g = torch.Generator(device="cuda").manual_seed(2)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    cols = (r(S, 705), r(S, 219), r(S, 12), r(S), r(S), r(S), r(S) - 12.0, r(S, 12) * 0.3, torch.ones(S, 12, device="cuda"))
    idx = torch.randperm(S, device="cuda", generator=g)[:B].contiguous()
    return cols, idx, make_batch(*cols, idx)


def _torch_denoiser(net, AUX_H):
    dims = [705] + AUX_H + [AUX_OUT]
    layers = []
    for l in range(len(dims) - 1):
        lin = torch.nn.Linear(dims[l], dims[l + 1]).cuda()
        with torch.no_grad():
            lin.weight.copy_(net.views["denoiser.%d.weight" % (2 * l)])
            lin.bias.copy_(net.views["denoiser.%d.bias" % (2 * l)])
        layers += [lin] + ([torch.nn.ELU()] if l < len(dims) - 2 else [])
    return torch.nn.Sequential(*layers)


@pytest.mark.parametrize("precision,tol,shape", [("f32", 2e-4, "generic"), ("bf16", 4e-2, "generic"), ("bf16", 4e-2, "fused"),
                                                 ("f32", 2e-4, "fused")])
def test_aux_head_forward_loss_and_gradient_vs_autograd(precision, tol, shape):
    from hgym import make_ppo_config
    S, B, coef = 900, 700, 0.5
    AUX_H = AUX_SHAPES[shape]
    plain, full = _nets(precision, B, AUX_H)
    assert full.P == plain.P + sum(v.numel() for k, v in full.views.items() if k.startswith("denoiser"))
    cols, idx, batch = _batch(S, B)
    ppo_plain, ppo_full = make_ppo_config(), make_ppo_config(aux_coef=coef)
    plain.ppo_grad(ppo_plain, batch)
    full.opt_state[10] = 0.0
    for _ in range(2):                                      # twice: nothing may accumulate across calls except opt[10]
        full.ppo_grad(ppo_full, batch)
    torch.cuda.synchronize()
    # the PPO part is untouched by the extra head: same kernels, same inputs -> identical bits
    gp, gf = plain.grad_views(), full.grad_views()
    for k in gp:
        assert torch.equal(gp[k], gf[k]), k
    # the head against autograd
    model = _torch_denoiser(full, AUX_H)
    obs, priv = cols[0][idx], cols[1][idx]
    y = model(obs)
    mse = ((y - priv[:, AUX_OFF:AUX_OFF + AUX_OUT]) ** 2).mean()
    (coef * mse).backward()
    np.testing.assert_allclose(float(full.opt_state[10]) / 2, float(mse.detach()), rtol=tol)
    yk = full.forward(2, obs.contiguous())
    torch.cuda.synchronize()
    scale = float(y.abs().max())
    assert float((yk - y).abs().max()) <= tol * scale
    for l, mod in enumerate(m for m in model if isinstance(m, torch.nn.Linear)):
        for nm, ref in (("weight", mod.weight.grad), ("bias", mod.bias.grad)):
            got = gf["denoiser.%d.%s" % (2 * l, nm)]
            err = float((got - ref).norm() / ref.norm().clamp_min(1e-30))
            assert err <= tol, (l, nm, err)


@pytest.mark.parametrize("shape", ["generic", "fused"])
def test_joint_update_moves_the_head_and_lowers_its_loss(shape):
    """A few joint Adam steps (PPO gradient + head gradient, one clip) at the BASELINE minibatch size: the head's MSE on the
    batch falls, everything stays finite, the head's operand copies follow its master weights."""
    from hgym import make_ppo_config
    S = B = 61440
    _, full = _nets("bf16", B, AUX_SHAPES[shape])
    cols, idx, batch = _batch(S, B)
    ppo = make_ppo_config(aux_coef=1.0, grad_norm_ready=True)
    losses = []
    for it in range(6):
        full.opt_state[10] = 0.0
        full.ppo_grad(ppo, batch)
        full.ppo_apply(ppo)
        losses.append(float(full.opt_state[10]))
    torch.cuda.synchronize()
    assert torch.isfinite(full.params).all()
    assert losses[-1] < losses[0], losses
    before = full.workspace.clone()
    full.sync_shadow()
    torch.cuda.synchronize()
    assert torch.equal(before, full.workspace)


def test_dwl_task_trains_end_to_end(tmp_path, monkeypatch):
    """The registered humanoid_dwl_ppo task through the reference-shaped surface: the runner builds the head from the policy
    cfg, the graph-captured rollout is unaffected by it, the head's MSE (Loss/denoise_mse) falls over a few iterations, its
    estimate is served by ActorCritic.denoise, and a checkpoint round-trips the head's parameters and Adam moments."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "humanoid-gym_amd"))
    from humanoid.envs import task_registry   # noqa: F401  (registers the tasks)
    from humanoid.utils import get_args
    from humanoid.algo import PPO
    PPO.precision = "bf16"
    monkeypatch.setenv("HGYM_ASYNC", "0")          # per-iteration loss read-back (what a logging run does)
    torch.manual_seed(5)
    np.random.seed(5)
    args = get_args(["--task=humanoid_dwl_ppo", "--headless", "--num_envs", "512", "--seed", "5"])
    env, _ = task_registry.make_env(name=args.task, args=args)
    runner, _ = task_registry.make_alg_runner(env=env, name=args.task, args=args, log_root=None)
    ac = runner.alg.actor_critic
    assert [k for k in ac.state_dict() if k.startswith("denoiser.")] == [
        "denoiser.%d.%s" % (i, w) for i in (0, 2, 4, 6) for w in ("weight", "bias")]
    losses = []
    for _ in range(6):
        runner.learn(num_learning_iterations=1, init_at_random_ep_len=False)
        losses.append(runner.alg.last_denoise_loss)
    assert all(np.isfinite(losses)) and losses[-1] < 0.7 * losses[0], losses
    obs, priv = env.get_observations(), env.get_privileged_observations()
    est = ac.denoise(obs)
    assert est.shape == (512, 73) and bool(torch.isfinite(est).all())
    # the served estimate is the module's own forward on the same (fp32 master) parameters, bf16-rounded operands
    want = ac.denoiser(obs)
    assert float((est - want).abs().max()) <= 4e-2 * max(1.0, float(want.abs().max()))
    err = float(((est - priv[:, -73:]) ** 2).mean())
    base = float((priv[:, -73:] ** 2).mean())
    assert err < base, (err, base)                 # better than predicting zero after six iterations
    path = str(tmp_path / "model.pt")
    runner.save(path)
    sd = torch.load(path)
    n_views = len(runner.alg.net.views)
    assert len(sd["optimizer_state_dict"]["state"]) == n_views
    before = {k: v.clone() for k, v in ac.state_dict().items()}
    with torch.no_grad():
        for k, v in ac.state_dict().items():
            if k.startswith("denoiser."):
                v.zero_()
    runner.load(path)
    for k, v in ac.state_dict().items():
        assert torch.equal(v, before[k]), k
    assert torch.equal(ac.denoise(obs), est)       # operand copies refreshed by load_state_dict
