"""-m gpu: the drop-in stack end to end -- task_registry.make_env / make_alg_runner / OnPolicyRunner.learn exactly as
the reference's scripts/train.py drives them -- plus checkpoint save / load and policy export."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(num_envs, extra=()):
    from humanoid.envs import task_registry
    from humanoid.utils import get_args
    args = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", str(num_envs)] + list(extra))
    env, env_cfg = task_registry.make_env(name=args.task, args=args)
    return env, args, task_registry


def test_env_api_shapes_and_semantics():
    env, args, _ = _make(128)
    assert (env.num_envs, env.num_obs, env.num_privileged_obs, env.num_actions) == (128, 705, 219, 12)
    assert env.dt == pytest.approx(0.01) and env.max_episode_length == 2400
    obs, priv = env.reset()
    assert obs.shape == (128, 705) and priv.shape == (128, 219)
    o2, p2, rew, dones, infos = env.step(torch.zeros(128, 12, device=env.device))
    assert rew.shape == (128,) and dones.dtype == torch.bool and set(infos) >= {"episode", "time_outs"}
    assert len(infos["episode"]) == 22 and all(k.startswith("rew_") for k in infos["episode"])
    assert infos["time_outs"].dtype == torch.bool and infos["time_outs"].shape == (128,)
    # tensors handed out by one step survive the next step (an algorithm may keep references for one step)
    keep = o2.clone()
    o3, *_ = env.step(torch.zeros(128, 12, device=env.device))
    assert torch.equal(keep, o2) and o3.data_ptr() != o2.data_ptr()
    # the runner rebinds episode_length_buf; the env must pick the new values up
    env.episode_length_buf = torch.full_like(env.episode_length_buf, 2400)
    _, _, _, dones, infos = env.step(torch.zeros(128, 12, device=env.device))
    torch.cuda.synchronize()
    assert bool(dones.all()) and bool(infos["time_outs"].all()) and int(env.episode_length_buf.sum()) == 0
    for name, shape in (("commands", (128, 4)), ("dof_pos", (128, 12)), ("dof_vel", (128, 12)), ("torques", (128, 12)),
                        ("base_lin_vel", (128, 3)), ("base_ang_vel", (128, 3)), ("contact_forces", (128, 13, 3)),
                        ("root_states", (128, 13)), ("rigid_state", (128, 13, 13))):
        assert tuple(getattr(env, name).shape) == shape, name
    env.commands[:, 0] = 0.5            # play.py writes commands in place
    torch.cuda.synchronize()
    assert float(env._buf.f["commands"][0].min()) == 0.5


def test_training_iterations_and_checkpoint(tmp_path):
    from humanoid.algo import PPO
    from humanoid.utils import export_policy_as_jit
    PPO.precision = "bf16"
    env, args, reg = _make(512, ["--max_iterations", "3"])
    runner, train_cfg = reg.make_alg_runner(env=env, name=args.task, args=args, log_root=str(tmp_path))
    p0 = runner.alg.net.params.clone()
    runner.learn(num_learning_iterations=3, init_at_random_ep_len=True)
    torch.cuda.synchronize()
    net = runner.alg.net
    assert torch.isfinite(net.params).all() and not torch.equal(p0, net.params)
    assert int(net.opt_state[1]) == 3 * 8                       # 2 epochs x 4 minibatches per iteration
    assert 1e-5 <= runner.alg.learning_rate <= 1e-2
    st = runner.alg.storage
    assert torch.isfinite(st.returns).all() and torch.isfinite(st.advantages).all()
    assert abs(float(st.advantages.mean())) < 1e-3 and abs(float(st.advantages.std()) - 1.0) < 1e-2
    # the storage rows really are what the env produced: slot t+1 history = slot t shifted (non-reset envs)
    keep = ~st.dones[3].view(-1).bool()
    assert torch.equal(st.observations[4][keep][:, :14 * 47], st.observations[3][keep][:, 47:])
    runner.wait_for_saves()            # learn() hands its checkpoints to the background writer (HGYM_ASYNC_SAVE, default on) and returns
    ckpts = [f for f in os.listdir(runner.log_dir) if f.startswith("model_")]
    assert "model_3.pt" in ckpts and "model_0.pt" in ckpts
    ck = torch.load(os.path.join(runner.log_dir, "model_3.pt"), map_location="cpu")
    assert set(ck) == {"model_state_dict", "optimizer_state_dict", "iter", "infos"}
    assert list(ck["model_state_dict"])[:3] == ["std", "actor.0.weight", "actor.0.bias"]
    # load into a fresh runner: parameters (and the MFMA operand shadows) follow
    env2, args2, reg2 = _make(64)
    r2, _ = reg2.make_alg_runner(env=env2, name=args2.task, args=args2, log_root=None)
    r2.load(os.path.join(runner.log_dir, "model_3.pt"))
    # ... and every device generator continues where a run of 3 iterations has it: the policy's sampling step, the minibatch
    # permutation draw and the env's common step counter (the Philox counter word of commands / pushes / noise / reset draws)
    T = r2.num_steps_per_env
    assert int(r2.alg._sample_step) == 3 * T and r2.alg._perm_draws == 3
    assert int(env2._buf.counters[0]) == int(runner.env._buf.counters[0]) == 1 + 3 * T
    r2.load(os.path.join(runner.log_dir, "model_3.pt"))        # idempotent
    assert int(env2._buf.counters[0]) == 1 + 3 * T
    x = torch.randn(64, 705, device="cuda")
    a1 = runner.alg.actor_critic.act_inference(x)
    a2 = r2.alg.actor_critic.act_inference(x)
    torch.cuda.synchronize()
    assert torch.equal(a1, a2)
    # exported TorchScript actor (what sim2sim.py loads) agrees with the MFMA forward to bf16 accuracy
    export_policy_as_jit(runner.alg.actor_critic, str(tmp_path / "exp"))
    pol = torch.jit.load(str(tmp_path / "exp" / "policy_1.pt"))
    ref = pol(x.cpu())
    err = float((a1.cpu() - ref).abs().max() / ref.abs().max())
    assert err < 3e-2, err


def test_background_checkpoint_equals_the_synchronous_one(tmp_path, monkeypatch):
    """HGYM_ASYNC_SAVE=1 (pinned-host snapshot + writer thread, OnPolicyRunner.save) must write what the default torch.save path
    writes: same keys, same parameter tensors, same Adam moments / step / param-group hyper-parameters, same learning rate (ADVICE r04:
    the writer rebuilds the dict from flat offsets -- a drift between the two would be silent)."""
    from humanoid.algo import PPO
    PPO.precision = "bf16"
    env, args, reg = _make(256)
    runner, _ = reg.make_alg_runner(env=env, name=args.task, args=args, log_root=None)
    runner.learn(num_learning_iterations=2, init_at_random_ep_len=True)
    torch.cuda.synchronize()
    monkeypatch.setenv("HGYM_ASYNC_SAVE", "0")
    runner.save(str(tmp_path / "sync.pt"))
    monkeypatch.setenv("HGYM_ASYNC_SAVE", "1")
    runner.save(str(tmp_path / "async.pt"))          # wait=True: on disk when the call returns
    a = torch.load(str(tmp_path / "sync.pt"), map_location="cpu")
    b = torch.load(str(tmp_path / "async.pt"), map_location="cpu")
    assert set(a) == set(b) and a["iter"] == b["iter"] == 2
    assert list(a["model_state_dict"]) == list(b["model_state_dict"])
    for k in a["model_state_dict"]:
        assert torch.equal(a["model_state_dict"][k], b["model_state_dict"][k]), k
    oa, ob = a["optimizer_state_dict"], b["optimizer_state_dict"]
    assert oa["param_groups"] == ob["param_groups"], (oa["param_groups"], ob["param_groups"])
    assert sorted(oa["state"]) == sorted(ob["state"])
    for i in oa["state"]:
        for f in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.equal(oa["state"][i][f], ob["state"][i][f]), (i, f)


def test_f32_runner_matches_torch_module():
    """fp32 parity mode: the bound nn.Module (plain torch on the same parameters) and the HIP forward agree to 1e-5."""
    from humanoid.algo import PPO
    PPO.precision = "f32"
    try:
        env, args, reg = _make(64)
        runner, _ = reg.make_alg_runner(env=env, name=args.task, args=args, log_root=None)
        runner.learn(num_learning_iterations=1, init_at_random_ep_len=True)
        x = torch.randn(64, 705, device="cuda")
        ac = runner.alg.actor_critic
        a = ac.act_inference(x)
        b = ac.actor(x)
        torch.cuda.synchronize()
        assert float((a - b).abs().max() / b.abs().max()) < 1e-5
    finally:
        PPO.precision = "bf16"


def test_log_sink_is_the_reference_bookkeeping():
    """HgymEnvOut.log_*: the step finaliser's device-side logging book-keeping against the reference's host loop
    (on_policy_runner.py:143-156) replayed on the same per-step rewards / dones / infos: running returns and lengths per env,
    rewbuffer / lenbuffer = the last 100 finished episodes in env order, ep_infos = extras["episode"] appended every step."""
    from collections import deque
    from humanoid.envs import task_registry
    from humanoid.utils import get_args
    N, steps = 384, 70
    args = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", str(N)])
    env, _ = task_registry.make_env(name=args.task, args=args)
    env.episode_length_buf = torch.randint(2300, 2400, (N,), device="cuda")      # many time-outs inside the window (> 100 episodes)
    assert env.bind_log_sink(True)
    values = torch.zeros(N, 1, device="cuda")
    sink = dict(values=values, rewards=torch.zeros(N, 1, device="cuda"), dones=torch.zeros(N, 1, dtype=torch.uint8, device="cuda"),
                step=torch.zeros(1, dtype=torch.int64, device="cuda"), gamma=0.99)
    env.bind_transition(sink)
    g = torch.Generator().manual_seed(0)
    rewbuffer, lenbuffer, ep_infos = deque(maxlen=100), deque(maxlen=100), []
    cur_r, cur_l = torch.zeros(N), torch.zeros(N)
    for t in range(steps):
        a = (torch.randn(N, 12, generator=g) * 0.5).cuda()
        obs, priv, rew, dones, infos = env.step(a)
        torch.cuda.synchronize()
        rew, dones = rew.cpu().clone(), dones.cpu().clone()
        ep_infos.append({k: float(v) for k, v in infos["episode"].items()})
        cur_r += rew
        cur_l += 1
        ids = (dones > 0).nonzero(as_tuple=False)
        rewbuffer.extend(cur_r[ids][:, 0].numpy().tolist())
        lenbuffer.extend(cur_l[ids][:, 0].numpy().tolist())
        cur_r[ids] = 0
        cur_l[ids] = 0
    ep_mean, ring_r, ring_l = env.log_sink_read()
    assert len(rewbuffer) == 100 == len(ring_r)
    assert sorted(ring_l) == sorted(lenbuffer) and sorted(ring_r) == sorted(rewbuffer)     # the same 100 episodes (ring order is a rotation)
    for k in ep_infos[0]:
        want = float(np.mean([e[k] for e in ep_infos]))
        assert abs(ep_mean[k] - want) <= 1e-6 + 1e-5 * abs(want), (k, ep_mean[k], want)
    torch.testing.assert_close(env._buf.log_cur[0].cpu(), cur_r, rtol=0, atol=0)
    torch.testing.assert_close(env._buf.log_cur[1].cpu(), cur_l, rtol=0, atol=0)
    env.bind_transition(None)
    env.bind_log_sink(False)


def test_asynchronous_log_prints_what_the_synchronous_one_does(tmp_path, capsys, monkeypatch):
    """Logging runs format iteration k's block from a pinned-host snapshot while iteration k + 1 runs (OnPolicyRunner._log_flush).
    Same seed, HGYM_ASYNC=0 (the per-iteration synchronising loop) against the default: every block present, in order, and every
    line that is not a wall-clock figure identical (losses, noise std, mean reward / episode length, all 22 episode terms)."""
    from humanoid.envs import task_registry
    from humanoid.utils import get_args

    def run(tag, async_on):
        monkeypatch.setenv("HGYM_ASYNC", "1" if async_on else "0")
        args = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", "256", "--seed", "3"])
        env, _ = task_registry.make_env(name=args.task, args=args)
        env.episode_length_buf = torch.randint(2340, 2400, (256,), device="cuda")     # time-outs inside the window: episode statistics exist
        runner, _ = task_registry.make_alg_runner(env=env, name=args.task, args=args, log_root=str(tmp_path / tag))
        capsys.readouterr()
        runner.learn(num_learning_iterations=4, init_at_random_ep_len=False)
        torch.cuda.synchronize()
        out = capsys.readouterr().out
        keep = [ln for ln in out.splitlines() if ":" in ln and not any(w in ln for w in ("steps/s", "time:", "ETA", "Iteration time"))]
        return keep, out.count("Learning iteration")

    # (as in the reference, `--seed` reaches the ENV through the task's registered train cfg, i.e. from the second make_env of a
    # process on: helpers.update_cfg_from_args / task_registry.get_cfgs -- one throw-away run puts both compared runs behind that)
    run("warm", True)
    sync_lines, sync_blocks = run("sync", False)
    async_lines, async_blocks = run("async", True)
    assert sync_blocks == async_blocks == 4
    assert any("Mean reward" in ln for ln in sync_lines) and any("rew_tracking_lin_vel" in ln for ln in sync_lines)
    assert sync_lines == async_lines


def test_full_size_rollout_storage_is_self_consistent():
    """BASELINE size (4096 envs x 60 steps through the one-launch-per-step rollout, replayed from its HIP graph), size-independent
    properties of what the rollout leaves in the storage: the stored log-probabilities are those of the stored actions under the
    stored (mu, sigma); sigma is the std parameter; mu and the values are what the actor / critic return for the stored rows
    when asked again in one batch (the 64-row-tile forward instead of the rollout's 32-row tiles); every row's history is the
    previous row's shifted by one frame, or zero where the env had just been reset; rewards finite, dones boolean."""
    import math
    from humanoid.algo import PPO
    PPO.precision = "bf16"
    env, args, reg = _make(4096)
    runner, _ = reg.make_alg_runner(env=env, name=args.task, args=args, log_root=None)
    alg, ac = runner.alg, runner.alg.actor_critic
    alg.learning_rate = 0.0                     # the updates of this run must not move the parameters the rollout used
    p0 = alg.net.params.clone()
    runner.learn(num_learning_iterations=3, init_at_random_ep_len=True)     # eager, captured, replayed
    torch.cuda.synchronize()
    assert torch.equal(alg.net.params, p0) and runner._graph is not None
    st = alg.storage
    T, N = st.num_transitions_per_env, st.num_envs
    assert (T, N) == (60, 4096)
    obs, priv = st.observations, st.privileged_observations
    act, mu, sg, lp, val = st.actions, st.mu, st.sigma, st.actions_log_prob.view(T, N), st.values.view(T, N)
    # sampling epilogue
    assert torch.equal(sg, ac.std.detach().expand_as(sg))
    want = (-((act - mu) ** 2) / (2 * sg ** 2) - sg.log() - 0.5 * math.log(2 * math.pi)).sum(-1)
    assert float((lp - want).abs().max()) <= 2e-5 * float(want.abs().max())
    z = (act - mu) / sg
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1.0) < 5e-3           # 2.9 M standard normals
    # the networks, asked again over whole slabs of rows (another tile shape, same arithmetic per row)
    # (slot 0 is skipped throughout: storage.clear() has already rotated the rollout's last observation into it for the next iteration)
    for t0 in (1, 29, 58):
        rows_o, rows_p = obs[t0:t0 + 2].reshape(2 * N, -1), priv[t0:t0 + 2].reshape(2 * N, -1)
        mu2 = ac.act_inference(rows_o).view(2, N, -1)
        v2 = ac.evaluate(rows_p).view(2, N)
        assert float((mu2 - mu[t0:t0 + 2]).abs().max()) <= 1e-6 * max(1.0, float(mu.abs().max()))
        assert float((v2 - val[t0:t0 + 2]).abs().max()) <= 1e-6 * max(1.0, float(val.abs().max()))
    # history: 15 x 47 actor frames, 3 x 73 critic frames
    dones = st.dones.view(T, N).bool()
    assert st.dones.dtype in (torch.uint8, torch.bool) or set(st.dones.unique().tolist()) <= {0, 1}
    for t in range(1, T - 1):
        keep, gone = ~dones[t], dones[t]
        assert torch.equal(obs[t + 1][keep][:, :14 * 47], obs[t][keep][:, 47:])
        assert torch.equal(priv[t + 1][keep][:, :2 * 73], priv[t][keep][:, 73:])
        if bool(gone.any()):
            assert float(obs[t + 1][gone][:, :14 * 47].abs().max()) == 0.0
            assert float(priv[t + 1][gone][:, :2 * 73].abs().max()) == 0.0
    assert int(dones.sum()) > 0                                  # resets did occur in the window
    assert torch.isfinite(st.rewards).all() and torch.isfinite(obs).all() and torch.isfinite(priv).all()


@pytest.mark.parametrize("precision", ["bf16", "f32"])
def test_other_frame_stacks_and_a_ragged_env_count_train(precision, tmp_path):
    """What the reference's config CAN resize (humanoid_config.py:40-45: frame_stack, c_frame_stack -> num_observations,
    num_privileged_obs; any num_envs; train_cfg's rollout length / minibatch count): 100 envs, 4 x 47 actor inputs, 2 x 73 critic
    inputs, 24 steps per rollout in 2 minibatches, through make_env / make_alg_runner / learn with logging on.  The env runs the
    runtime-sized env_step_kernel (tests/test_env_gpu.py::test_generic_frame_stack_gpu pins it against the oracle), the storage and the
    update take their widths from the env; here: shapes, finiteness, the optimiser's step count, a checkpoint that loads back."""
    from humanoid.algo import PPO
    from humanoid.envs import task_registry
    from humanoid.utils import get_args
    PPO.precision = precision
    try:
        args = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", "100", "--seed", "11"])
        import copy
        env_cfg, train_cfg = (copy.deepcopy(c) for c in task_registry.get_cfgs(name=args.task))     # (the registry hands out its singletons)
        env_cfg.env.frame_stack, env_cfg.env.c_frame_stack = 4, 2
        env_cfg.env.num_observations = 4 * env_cfg.env.num_single_obs
        env_cfg.env.num_privileged_obs = 2 * env_cfg.env.single_num_privileged_obs
        train_cfg.runner.num_steps_per_env = 24
        train_cfg.algorithm.num_mini_batches = 2
        env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
        assert (env.num_envs, env.num_obs, env.num_privileged_obs) == (100, 188, 146)
        runner, _ = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg, log_root=str(tmp_path))
        ac = runner.alg.actor_critic
        assert ac.actor[0].in_features == 188 and ac.critic[0].in_features == 146
        runner.learn(num_learning_iterations=3, init_at_random_ep_len=True)
        torch.cuda.synchronize()
        st = runner.alg.storage
        assert tuple(st.observations.shape) == (24, 100, 188) and tuple(st.privileged_observations.shape) == (24, 100, 146)
        net = runner.alg.net
        assert torch.isfinite(net.params).all() and int(net.opt_state[1]) == 3 * 2 * 2      # 2 epochs x 2 minibatches x 3 iterations
        assert torch.isfinite(st.returns).all() and torch.isfinite(st.advantages).all()
        runner.wait_for_saves()
        path = os.path.join(runner.log_dir, "model_3.pt")
        assert os.path.exists(path)
        before = net.params.clone()
        net.params.zero_()
        runner.load(path)
        assert torch.equal(net.params, before)
        obs = env.get_observations()
        act = runner.get_inference_policy(device=env.device)(obs)
        assert tuple(act.shape) == (100, 12) and torch.isfinite(act).all()
    finally:
        PPO.precision = "bf16"
