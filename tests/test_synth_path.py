"""The TIMED env configuration -- internal Philox draws + synthetic physics, what bench.py runs -- against the oracle consuming the
same counter-based stream (oracle/synth_env_oracle.py), end to end.  Reference semantics at stake: the consumers of the draws
(/root/reference/humanoid/envs/base/legged_robot.py:328-333,367; envs/custom/humanoid_env.py:88-93,194-196,251).

  not gpu : the kernel SOURCE compiled for the host (tests/hostcheck), both per-env chain forms
  gpu     : `hgym_env_prime` + `hgym_env_step_synth` at N = 4096, and the env part of `hgym_rollout_step` (the one-launch-per-vec-step
            path of the rollout) fed with the policy's own sampled actions, >= 20 steps with a push, command resamples,
            time-outs and base-link resets inside the window.  Masks bit-exact, floats 1e-5 (tests/synth_common.py)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

import bf16_report as BR

import env_common as EC
import synth_common as SC
from oracle import synth_env_oracle as S
from oracle import xbot_constants as K
from oracle.xbot_env_oracle import XBotEnvOracle


def test_oracle_draw_layout_matches_kernel_source_streams():
    """env_draws' tables are uniform_at / normal_at of the kernel source (tests/hostcheck build) at the slot map's offsets."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck"))
    import build_hostcheck
    lib = C.CDLL(build_hostcheck.build())
    lib.hc_streams.argtypes = [C.c_uint64, C.c_int64, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.hc_streams.restype = None
    seed, step, env = 0x1234ABCD5678EF01, (2 << 32) + 777, 4001
    d = S.env_draws(seed, step, np.array([env], dtype=np.uint32))

    def stream(slot, n):
        u, z = (C.c_float * n)(), (C.c_float * n)()
        lib.hc_streams(seed, step, env, slot, n, u, z)
        return np.array(u[:], dtype=np.float32), np.array(z[:], dtype=np.float32)
    u0, _ = stream(S.SLOT_DELAY_CMD, 4)
    assert float(d["u_delay"][0]) == u0[0] and np.array_equal(d["u_cmd"][0, :3].numpy(), u0[1:4])
    assert np.array_equal(d["u_cmd"][0, 3:].numpy(), stream(S.SLOT_CMD_RESET, 3)[0])
    np.testing.assert_allclose(d["z_act"][0].numpy(), stream(S.SLOT_ACT, 12)[1], rtol=2e-6, atol=2e-6)
    assert np.array_equal(d["u_dof"][0].numpy(), stream(S.SLOT_DOF, 12)[0])
    assert np.array_equal(d["u_push"][0].numpy(), stream(S.SLOT_PUSH, 5)[0])
    np.testing.assert_allclose(d["z_obs"][0].numpy(), stream(S.SLOT_OBS, 47)[1], rtol=2e-6, atol=2e-6)
    pu, pz = stream(S.SLOT_PHYS, 36)
    for c in range(S.PHYS_CALLS):
        got = d["phys"][0, 4 * c:4 * c + 4].numpy()
        if c in S.PHYS_NORMAL_CALLS:
            np.testing.assert_allclose(got, pz[4 * c:4 * c + 4], rtol=2e-6, atol=2e-6)
        else:
            assert np.array_equal(got, pu[4 * c:4 * c + 4])
    # prime / reset_all draw from a disjoint counter range
    assert not np.array_equal(S.env_draws(seed, step, np.array([env]), mode_step=False)["u_dof"].numpy(), d["u_dof"].numpy())


@pytest.mark.parametrize("split", [0, 1, 2, 3])
@pytest.mark.parametrize("N,epb,nthreads", [(96, 8, 64), (37, 16, 256)])
def test_fused_synthetic_step_host_vs_oracle(N, epb, nthreads, split):
    """hc_env_step_ex(fused = 1, no noise tables): stage-in, env_fill_draws, joints + synthetic physics, per-env chain, history
    -- the kernel source on the host -- against the oracle on the same Philox stream, 30 steps."""
    from hgym import EnvBuffers, default_env_config
    be = EC.HostBackend(envs_per_block=epb, nthreads=nthreads, split=split)
    seed = 0xC0FFEE1234
    g = torch.Generator().manual_seed(N)
    cfg = default_env_config(N, seed=seed)
    buf = EnvBuffers(cfg, "cpu")
    buf.f["friction"].copy_((0.1 + 1.9 * torch.rand(N, generator=g)).view(1, N))
    buf.f["body_mass"].copy_((10.0 + 10.0 * torch.rand(N, generator=g)).view(1, N))
    sim, st, out = buf.sim_struct(), buf.state_struct(), buf.out_struct()
    o = XBotEnvOracle(N, frictions=buf.view("friction").clone(), body_mass=buf.view("body_mass").clone())
    be.lib.hc_env_step_ex(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), None, None, 1, 0, epb, nthreads, 0)      # prime
    S.synth_prime(o, seed)
    flips = [0]
    SC.compare(buf, o, "prime", flips)
    SC.plant(buf, o, g)
    counts = dict(reset=0, timeout=0, push=0)
    for t in range(30):
        a = torch.randn(N, 12, generator=g) * 1.5
        if t % 5 == 2:
            a[t % N] *= 40.0
        ad = a.clone().contiguous()
        be.lib.hc_env_step_ex(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), None, C.cast(ad.data_ptr(), C.POINTER(C.c_float)),
                              0, 1, epb, nthreads, split)
        _, _, _, _, info = S.synth_step(o, seed, a)
        SC.compare(buf, o, "step %d" % t, flips)
        assert torch.equal(ad, a)                       # the caller's action tensor is read-only without use_ref_actions
        counts["reset"] += int(o.reset.sum())
        counts["timeout"] += int(o.time_out.sum())
        counts["push"] += int(info["pushed"])
    SC.report("kernel source on the host (hc_env_step_ex, fused, split=%d) vs oracle, N=%d, 30 steps: %s" % (split, N, counts), flips[0])
    assert counts["push"] == 1 and counts["timeout"] >= 3 and counts["reset"] > counts["timeout"] - 1
    assert int(buf.counters[0]) == 397 + 30


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("N,steps", [(4096, 24), (1000, 12)])
def test_env_step_synth_internal_philox_vs_oracle_gpu(N, steps):
    """hgym_env_prime + hgym_env_step_synth (internal Philox, synthetic physics: one launch per step) at the BASELINE env count and
    at a count that is not a multiple of the workgroup's 16 envs."""
    from hgym import EnvBuffers, default_env_config, _lib as L
    seed = 0x5EED0000 + N
    g = torch.Generator().manual_seed(N)
    cfg = default_env_config(N, seed=seed)
    buf = EnvBuffers(cfg, "cuda")
    buf.f["friction"].copy_((0.1 + 1.9 * torch.rand(N, generator=g)).view(1, N))
    buf.f["body_mass"].copy_((10.0 + 10.0 * torch.rand(N, generator=g)).view(1, N))
    sim, st, out, nz = buf.sim_struct(), buf.state_struct(), buf.out_struct(), buf.noise_struct()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    o = XBotEnvOracle(N, frictions=buf.view("friction").cpu().clone(), body_mass=buf.view("body_mass").cpu().clone())
    L.check(L.lib.hgym_env_prime(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), C.byref(nz), s), "prime")
    torch.cuda.synchronize()
    S.synth_prime(o, seed)
    flips = [0]
    SC.compare(buf, o, "prime", flips)
    SC.plant(buf, o, g)
    counts = dict(reset=0, timeout=0, push=0, base_hit=0)
    for t in range(steps):
        a = torch.randn(N, 12, generator=g) * 1.5
        ad = a.cuda()
        L.check(L.lib.hgym_env_step_synth(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), L.fptr(ad), s), "step")
        torch.cuda.synchronize()
        _, _, _, _, info = S.synth_step(o, seed, a)
        SC.compare(buf, o, "step %d" % t, flips, check_obs=(t % 4 == 0 or t == steps - 1))
        counts["reset"] += int(o.reset.sum())
        counts["timeout"] += int(o.time_out.sum())
        counts["base_hit"] += int((o.reset & ~o.time_out).sum())
        counts["push"] += int(info["pushed"])
    SC.report("internal-Philox env step (hgym_env_step_synth) vs oracle, N=%d, %d steps: %s" % (N, steps, counts), flips[0])
    assert counts["push"] == 1 and counts["timeout"] >= 3 and counts["base_hit"] >= 3


@pytest.mark.gpu
@pytest.mark.parametrize("ahead", [True, False])
def test_rollout_step_env_part_vs_oracle_gpu(monkeypatch, ahead):
    """hgym_rollout_step (actor tile + env step of the same 32 envs + critic tile + previous finaliser, ONE launch per vec-step),
    4096 envs, 24 steps: the policy's sampled actions (storage) are the oracle's inputs; the next observations the launch wrote
    into the storage slots, the rewards / dones its finaliser stored (time-out bootstrap included: ppo.py:107-108 on the
    stale-by-design extras["time_outs"]) and the env state at the end are compared with the oracle on the same Philox stream.
    Reference: /root/reference/humanoid/algo/ppo/on_policy_runner.py:129-141 (the loop body this launch stands for)."""
    from humanoid.algo import PPO
    from humanoid.envs import task_registry
    from humanoid.utils import get_args
    PPO.precision = "bf16"
    monkeypatch.setenv("HGYM_GRAPH", "0")
    torch.manual_seed(99)
    np.random.seed(99)
    N, T = 4096, 24
    args = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", str(N), "--seed", "17"])
    task_registry.train_cfgs[args.task].seed = 17
    env, _ = task_registry.make_env(name=args.task, args=args)
    runner, _ = task_registry.make_alg_runner(env=env, name=args.task, args=args, log_root=None)
    alg, buf = runner.alg, env._buf
    assert env.rollout_fused_supported(alg.net)
    seed = int(env._ncfg.seed)
    g = torch.Generator().manual_seed(3)
    SC.plant(buf, None, g, csc=390)
    torch.cuda.synchronize()
    o = SC.oracle_from_buffers(buf)
    st = alg.storage
    obs_all, priv_all = st._obs_all, st._priv_all
    obs_all[0].copy_(env.get_observations())
    priv_all[0].copy_(env.get_privileged_observations())
    alg.env_stores_transitions = True
    with torch.inference_mode():
        env.rollout_begin(alg._sample_step, T)
        obs, pobs = obs_all[0], priv_all[0]
        for i in range(T):
            # ahead: the launch also writes the older frames of the slot after next (HgymEnvOut.obs_ahead) and the next one skips
            # its own copy -- the rows compared below are produced by a different launch, and must not differ
            alg.fused_rollout_step(env, i, obs, pobs, obs_all[i + 1], priv_all[i + 1],
                                   (obs_all[i + 2], priv_all[i + 2]) if (ahead and i + 2 <= T) else None)
            obs, pobs = obs_all[i + 1], priv_all[i + 1]
        env.rollout_end()
    torch.cuda.synchronize()
    flips = [0]
    counts = dict(reset=0, timeout=0, push=0, boot=0)
    for i in range(T):
        a = st.actions[i].cpu()
        obs_o, priv_o, rew_o, reset_o, info = S.synth_step(o, seed, a)
        rew_dev_boot = st.rewards[i].view(-1).cpu()
        boot = alg.gamma * (st.values[i].view(-1).cpu() * o.extras_time_outs.float())        # ppo.py:107-108
        # device: rew + gamma * (V * to) in fp32; undo the bootstrap with the same fp32 arithmetic to reach the env's reward
        want = o.rew + boot
        d = (rew_dev_boot - want).abs()
        bad = (d > (EC.ATOL + EC.RTOL * want.abs())).nonzero().flatten().tolist()
        for e in bad:                               # low_speed threshold flips (tests/synth_common.py): counted, re-synchronised
            assert float(d[e]) <= SC.LOW_SPEED_QUANTUM, (i, e, float(d[e]))
            o.rew[e] = rew_dev_boot[e] - boot[e]
            o.episode_sums[e, K.REWARD_NAMES.index("low_speed")] += (rew_dev_boot[e] - want[e])
        flips[0] += len(bad)
        EC.exact(st.dones[i].view(-1), reset_o, "dones %d" % i)
        EC.close(obs_all[i + 1], obs_o, "next obs %d" % i)
        EC.close(priv_all[i + 1], priv_o, "next privileged obs %d" % i)
        counts["reset"] += int(reset_o.sum())
        counts["timeout"] += int(o.time_out.sum())
        counts["push"] += int(info["pushed"])
        counts["boot"] += int((boot != 0).sum())
    assert flips[0] <= 2, flips
    # the env state the rollout leaves behind (the last step used the primary rew / reset / time_out set)
    o.rew = buf.rew.cpu().clone() if flips[0] else o.rew
    EC.compare_state(SC.Holder(buf), o, "after the rollout", check_obs=False)
    SC.report("fused rollout step (hgym_rollout_step, rows ahead: %s), env part vs oracle, N=%d, %d steps: %s" % (ahead, N, T, counts), flips[0])
    assert counts["push"] == 1 and counts["timeout"] >= 3 and counts["reset"] > counts["timeout"] and counts["boot"] >= 3
    assert int(buf.counters[0]) == 390 + T and int(alg._sample_step) == T


@pytest.mark.gpu
def test_two_launch_rollout_8192_envs_vs_oracle_gpu(monkeypatch):
    """BASELINE configs[3]'s TIMED rollout at its own size: 8192 envs take the two-launch path -- `hgym_policy_act_fin`
    (64-row policy tiles, `mlp_fwd_kernel<64,16,2>`, the previous step's finaliser riding as one extra workgroup) then
    `hgym_env_step_synth` with `defer_finalize`, internal Philox and the transition sink -- driven exactly as
    OnPolicyRunner.learn's loop body drives it, 20 steps, against oracle/synth_env_oracle.py fed the stored actions: dones
    bit-exact, next observations / privileged observations / bootstrapped rewards 1e-5.
    Reference: /root/reference/humanoid/algo/ppo/on_policy_runner.py:129-141."""
    from humanoid.algo import PPO
    from humanoid.envs import task_registry
    from humanoid.utils import get_args
    PPO.precision = "bf16"
    monkeypatch.setenv("HGYM_GRAPH", "0")
    torch.manual_seed(98)
    np.random.seed(98)
    N, T = 8192, 20
    args = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", str(N), "--seed", "23"])
    task_registry.train_cfgs[args.task].seed = 23
    env, _ = task_registry.make_env(name=args.task, args=args)
    runner, _ = task_registry.make_alg_runner(env=env, name=args.task, args=args, log_root=None)
    alg, buf = runner.alg, env._buf
    assert not env.rollout_fused_supported(alg.net)          # two 32-row tiles per CU do not fit one round: the runner's other branch
    seed = int(env._ncfg.seed)
    g = torch.Generator().manual_seed(5)
    SC.plant(buf, None, g, csc=392)
    torch.cuda.synchronize()
    o = SC.oracle_from_buffers(buf)
    st = alg.storage
    obs_all, priv_all = st._obs_all, st._priv_all
    obs_all[0].copy_(env.get_observations())
    priv_all[0].copy_(env.get_privileged_observations())
    alg.env_stores_transitions = True
    try:
        with torch.inference_mode():
            obs, pobs, fin = obs_all[0], priv_all[0], None
            for i in range(T):                                  # on_policy_runner.py (this repo): rollout(), the defer_ok branch
                actions = alg.act(obs, pobs, env_fin=fin)
                env.bind_outputs(obs_all[i + 1], priv_all[i + 1])
                env.bind_transition(alg.transition_sink(), defer_finalize=True)
                obs, pobs, rewards, dones, infos = env.step(actions)
                fin = env.take_pending_finalize()
                assert fin is not None
                alg.process_env_step(rewards, dones, infos, stored=True)
            env.run_finalize(fin)
        torch.cuda.synchronize()
    finally:
        env.bind_outputs(None, None)
        env.bind_transition(None)
        alg.env_stores_transitions = False
    flips = [0]
    counts = dict(reset=0, timeout=0, push=0, boot=0)
    for i in range(T):
        a = st.actions[i].cpu()
        obs_o, priv_o, rew_o, reset_o, info = S.synth_step(o, seed, a)
        rew_dev_boot = st.rewards[i].view(-1).cpu()
        boot = alg.gamma * (st.values[i].view(-1).cpu() * o.extras_time_outs.float())        # ppo.py:107-108
        want = o.rew + boot
        d = (rew_dev_boot - want).abs()
        bad = (d > (EC.ATOL + EC.RTOL * want.abs())).nonzero().flatten().tolist()
        for e in bad:                               # low_speed threshold flips (tests/synth_common.py): counted, re-synchronised
            assert float(d[e]) <= SC.LOW_SPEED_QUANTUM, (i, e, float(d[e]))
            o.rew[e] = rew_dev_boot[e] - boot[e]
            o.episode_sums[e, K.REWARD_NAMES.index("low_speed")] += (rew_dev_boot[e] - want[e])
        flips[0] += len(bad)
        EC.exact(st.dones[i].view(-1), reset_o, "dones %d" % i)
        EC.close(obs_all[i + 1], obs_o, "next obs %d" % i)
        EC.close(priv_all[i + 1], priv_o, "next privileged obs %d" % i)
        counts["reset"] += int(reset_o.sum())
        counts["timeout"] += int(o.time_out.sum())
        counts["push"] += int(info["pushed"])
        counts["boot"] += int((boot != 0).sum())
    assert flips[0] <= 2, flips
    o.rew = buf.rew.cpu().clone() if flips[0] else o.rew
    EC.compare_state(SC.Holder(buf), o, "after the two-launch rollout", check_obs=False)
    SC.report("two-launch rollout (policy_act_fin<64-row tiles> + env_step_synth, deferred finaliser), N=8192, %d steps vs oracle: %s"
              % (T, counts), flips[0])
    assert counts["push"] == 1 and counts["timeout"] >= 3 and counts["reset"] > counts["timeout"] and counts["boot"] >= 3
    assert int(buf.counters[0]) == 392 + T and int(alg._sample_step) == T


@pytest.mark.gpu
@pytest.mark.parametrize("N", [8192, 4096])
def test_deferred_values_rollout_vs_oracle_gpu(monkeypatch, N):
    """the fused rollout launch WITHOUT critic tiles (hgym_rollout_step with values = NULL: one actor + env
    workgroup per 32 envs, 8192 envs in one round) and the critic ONCE over the stored rows afterwards (hgym_critic_values), driven as
    OnPolicyRunner.learn drives it, 20 steps, against oracle/synth_env_oracle.py fed the stored actions.
      * dones bit-exact, next observations / privileged observations 1e-5, the RAW rewards of the sink 1e-5 (threshold flips counted);
      * the bootstrap flags the finaliser stored == the oracle's stale-by-design extras["time_outs"], bit-exact;
      * storage.values (one 64-row-tile pass over all slots) == ActorCritic.evaluate of each slot (the per-step launches' kernel) to the
        bf16 path's tolerance, last_values likewise;
      * after compute_returns: rewards == raw + gamma * (V * flags) EXACTLY (ppo.py:107-108 in fp32), returns / advantages vs the
        oracle's GAE on those columns 1e-5 (rollout_storage.py:122-136).
    N = 4096 runs the same path where the inline form would also fit (HGYM_ROLLOUT_CRITIC=deferred).
    Reference: /root/reference/humanoid/algo/ppo/on_policy_runner.py:129-141,160-165."""
    from humanoid.algo import PPO
    from humanoid.envs import task_registry
    from humanoid.utils import get_args
    from oracle import ppo_oracle as P
    PPO.precision = "bf16"
    monkeypatch.setenv("HGYM_GRAPH", "0")
    monkeypatch.setenv("HGYM_ROLLOUT_CRITIC", "deferred" if N <= 4096 else "auto")
    torch.manual_seed(97)
    np.random.seed(97)
    T = 20
    args = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", str(N), "--seed", "29"])
    task_registry.train_cfgs[args.task].seed = 29
    task_registry.train_cfgs[args.task].runner.num_steps_per_env = T
    try:
        env, _ = task_registry.make_env(name=args.task, args=args)
        runner, _ = task_registry.make_alg_runner(env=env, name=args.task, args=args, log_root=None)
    finally:
        task_registry.train_cfgs[args.task].runner.num_steps_per_env = 60
    alg, buf = runner.alg, env._buf
    assert env.rollout_fused_mode(alg.net) == "deferred" and not env.rollout_fused_supported(alg.net)
    seed = int(env._ncfg.seed)
    g = torch.Generator().manual_seed(6)
    SC.plant(buf, None, g, csc=394)
    torch.cuda.synchronize()
    o = SC.oracle_from_buffers(buf)
    st = alg.storage
    assert st.num_transitions_per_env == T
    obs_all, priv_all = st._obs_all, st._priv_all
    obs_all[0].copy_(env.get_observations())
    priv_all[0].copy_(env.get_privileged_observations())
    alg.env_stores_transitions = True
    with torch.inference_mode():
        env.rollout_begin(alg._sample_step, T)
        obs, pobs = obs_all[0], priv_all[0]
        for i in range(T):
            alg.fused_rollout_step(env, i, obs, pobs, obs_all[i + 1], priv_all[i + 1], deferred=True)
            obs, pobs = obs_all[i + 1], priv_all[i + 1]
        env.rollout_end()
        torch.cuda.synchronize()
        raw = st.rewards.clone()
        assert float(st.values.abs().max()) == 0.0           # nothing has evaluated the critic yet
        alg.deferred_values()
        torch.cuda.synchronize()
    flips = [0]
    counts = dict(reset=0, timeout=0, push=0, boot=0)
    for i in range(T):
        a = st.actions[i].cpu()
        obs_o, priv_o, rew_o, reset_o, info = S.synth_step(o, seed, a)
        rew_dev = raw[i].view(-1).cpu()
        d = (rew_dev - o.rew).abs()
        bad = (d > (EC.ATOL + EC.RTOL * o.rew.abs())).nonzero().flatten().tolist()
        for e in bad:                               # low_speed threshold flips (tests/synth_common.py): counted, re-synchronised
            assert float(d[e]) <= SC.LOW_SPEED_QUANTUM, (i, e, float(d[e]))
            o.episode_sums[e, K.REWARD_NAMES.index("low_speed")] += (rew_dev[e] - o.rew[e])
            o.rew[e] = rew_dev[e]
        flips[0] += len(bad)
        EC.exact(st.dones[i].view(-1), reset_o, "dones %d" % i)
        EC.exact(st.time_outs[i].view(-1).bool(), o.extras_time_outs.bool(), "bootstrap flags %d" % i)
        EC.close(obs_all[i + 1], obs_o, "next obs %d" % i)
        EC.close(priv_all[i + 1], priv_o, "next privileged obs %d" % i)
        counts["reset"] += int(reset_o.sum())
        counts["timeout"] += int(o.time_out.sum())
        counts["push"] += int(info["pushed"])
        counts["boot"] += int(o.extras_time_outs.sum())
    assert flips[0] <= 2, flips
    o.rew = buf.rew.cpu().clone() if flips[0] else o.rew
    EC.compare_state(SC.Holder(buf), o, "after the deferred-values rollout", check_obs=False)
    # the one-pass critic against the per-step launches' kernel on the same rows, and its bf16 shadow rows
    with torch.inference_mode():
        worst = 0.0
        for i in (0, 1, T // 2, T - 1):
            v_step = alg.actor_critic.evaluate(priv_all[i]).view(-1)
            worst = max(worst, float((st.values[i].view(-1) - v_step).abs().max() / v_step.abs().max()))
        v_last = alg.actor_critic.evaluate(priv_all[T]).view(-1)
        worst = max(worst, float((st.last_values.view(-1) - v_last).abs().max() / v_last.abs().max()))
    BR.check("one-pass critic (64-row tiles) vs the per-step critic kernel", worst)
    assert all(st.shadow_valid) and torch.equal(st._priv_bf16[:, :, :219], st.privileged_observations.to(torch.bfloat16))
    assert torch.equal(st._obs_bf16[:, :, :705], st.observations.to(torch.bfloat16))
    # compute_returns: the bootstrap applied as the scan loads the rewards, the column written back
    alg.compute_returns(priv_all[T])
    torch.cuda.synchronize()
    gam = torch.tensor(alg.gamma, dtype=torch.float32)
    want_rew = raw.cpu() + gam * (st.values.cpu() * st.time_outs.cpu().float())
    assert torch.equal(st.rewards.cpu(), want_rew)
    assert int((want_rew != raw.cpu()).sum()) >= 3
    ret_o, adv_o = P.gae_returns(want_rew.view(T, N), st.values.cpu().view(T, N), st.dones.cpu().view(T, N).bool(), st.last_values.cpu().view(N),
                                 alg.gamma, alg.lam)
    np.testing.assert_allclose(st.returns.cpu().view(T, N).numpy(), ret_o.numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(st.advantages.cpu().view(T, N).numpy(), P.normalize_advantages(adv_o).numpy(), rtol=1e-4, atol=2e-5)
    SC.report("fused rollout step WITHOUT critic tiles + one critic pass (deferred values), N=%d, %d steps vs oracle: %s; one-pass vs per-step "
              "critic: %.2e" % (N, T, counts, worst), flips[0])
    assert counts["push"] == 1 and counts["timeout"] >= 3 and counts["reset"] > counts["timeout"] and counts["boot"] >= 3
    assert int(buf.counters[0]) == 394 + T and int(alg._sample_step) == T
