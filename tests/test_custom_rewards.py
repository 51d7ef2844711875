"""User-defined reward terms (/root/reference/humanoid/envs/base/legged_robot.py:518-541 finds `_reward_<name>` by name for every
non-zero scale and sums the terms in alphabetical order, :217-235): the two-launch env step (hgym_env_step_begin -> the caller's
terms -> hgym_env_step_end) against the oracle, which evaluates the same extra terms where the reference would.

  not gpu : the kernel source on the host (tests/hostcheck: hc_env_step_phase)
  gpu     : the HIP kernels through the C-ABI, and the reference-shaped surface (a task subclass with `_reward_*` methods)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import env_common as EC
from oracle import xbot_constants as K
from oracle.xbot_env_oracle import XBotEnvOracle

# name -> (scale, oracle fn, product fn over the env buffers): one term that sorts first, one in the middle, one last, and one
# that REPLACES a built-in term (a subclass overriding `_reward_torques`)
TERMS = {
    "aaa_speed": (0.3, lambda o: torch.square(o.base_lin_vel[:, 0]), lambda b: torch.square(b.view("base_lin_vel")[:, 0])),
    "heading_err": (-0.5, lambda o: torch.abs(o.commands[:, 2]), lambda b: torch.abs(b.view("commands")[:, 2])),
    "torques": (dict(K.REWARD_SCALES_RAW)["torques"], lambda o: torch.sum(torch.abs(o.torques), dim=1),
                lambda b: torch.sum(torch.abs(b.view("torques")), dim=1)),
    # `termination` is summed AFTER the only-positive clip (legged_robot.py:229-235): position HGYM_NUM_REWARDS + 1
    "termination": (-2.0, lambda o: (o.reset & ~o.time_out).float(), lambda b: (b.reset & ~b.time_out).float()),
    "zz_dof": (-0.01, lambda o: torch.sum(torch.square(o.sim.dof_pos), dim=1), lambda b: torch.sum(torch.square(b.dof_pos_view()), dim=1)),
}
NAMES = sorted(TERMS)


class SplitStepBackend:
    """begin / end of the two-launch step on either backend."""

    def __init__(self, be):
        self.be = be

    def begin(self, cfg, sim, st, out, noise):
        if self.be.name == "hip":
            L = self.be.L
            L.check(L.lib.hgym_env_step_begin(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), C.byref(noise), None, self.be.stream()), "begin")
        else:
            self.be.lib.hc_env_step_phase(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), C.byref(noise), None, 0, self.be.epb,
                                          self.be.nthreads, 1)

    def end(self, cfg, sim, st, out, noise):
        if self.be.name == "hip":
            L = self.be.L
            L.check(L.lib.hgym_env_step_end(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), C.byref(noise), self.be.stream()), "end")
        else:
            self.be.lib.hc_env_step_phase(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), C.byref(noise), None, 0, self.be.epb,
                                          self.be.nthreads, 2)


def run_custom_trace(be, N, steps, seed):
    from hgym.env_buffers import EnvBuffers
    from hgym import default_env_config
    g = torch.Generator().manual_seed(seed)
    fr, bm = 0.1 + 1.9 * torch.rand(N, 1, generator=g), 10.0 + 10.0 * torch.rand(N, 1, generator=g)
    o = XBotEnvOracle(N, frictions=fr, body_mass=bm, extra_rewards={n: (TERMS[n][1], TERMS[n][0]) for n in NAMES})
    env = EC.EnvUnderTest(be, N, fr, bm)
    b, cfg = env.buf, env.cfg
    kernel_names = list(K.REWARD_NAMES)
    b.set_custom_rewards([len(kernel_names) + 1 if n == "termination" else sum(1 for x in kernel_names if x < n) for n in NAMES])
    cfg.reward_scales[kernel_names.index("torques")] = 0.0          # overridden: the kernel's own term is switched off
    env.sim, env.st, env.out = b.sim_struct(), b.state_struct(), b.out_struct()
    sp = SplitStepBackend(be)
    u_dof, u_cmd3, z_obs = torch.rand(N, 12, generator=g), torch.rand(N, 3, generator=g), torch.randn(N, 47, generator=g)
    o.prime(u_dof, u_cmd3, z_obs)
    env.prime(u_dof, u_cmd3, z_obs)
    be.sync()
    ep = torch.randint(0, 2400, (N,), generator=g)
    ep[: min(N, 6)] = torch.tensor([2399, 2398, 799, 1598, 0, 2396])[: min(N, 6)]
    o.ep_len = ep.clone()
    b.episode_length.copy_(ep)
    o.common_step_counter = 397
    b.counters[0] = 397
    resets = 0
    for t in range(steps):
        a_in = torch.randn(N, 12, generator=g) * 1.5
        frame = EC.synth_frames(g, N)
        u_delay, z_act = torch.rand(N, generator=g), torch.randn(N, 12, generator=g)
        u_cmd, u_dof = torch.rand(N, 6, generator=g), torch.rand(N, 12, generator=g)
        u_push, z_obs = torch.rand(N, 5, generator=g), torch.randn(N, 47, generator=g)
        o.pre_physics(a_in.clone(), u_delay, z_act)
        o.pd_torques()
        o.sim.load(*frame)
        o.post_physics(u_cmd, u_dof, u_push, z_obs)
        a = a_in.to(be.device).float().contiguous().clone()
        be.pre_physics(cfg, env.st, a, env._noise(u_delay=u_delay, z_act=z_act))
        be.pd_torques(cfg, env.sim, env.st)
        be.sync()
        b.load_sim(*frame)
        nz = env._noise(u_cmd=u_cmd, u_dof=u_dof, u_push=u_push, z_obs=z_obs)
        sp.begin(cfg, env.sim, env.st, env.out, nz)
        be.sync()
        # what a user's `_reward_<name>` reads between the two launches is the state compute_reward starts from
        EC.close(b.view("base_lin_vel"), o.base_lin_vel, "derive: base_lin_vel %d" % t)
        keep = ~o.reset                                   # (the oracle has finished the step: resetting envs are back at 0 there)
        EC.exact(b.episode_length.cpu()[keep], o.ep_len[keep], "derive: episode length already incremented %d" % t)
        EC.exact(b.reset, o.reset, "derive: termination flags %d" % t)
        for j, n in enumerate(NAMES):
            b.custom_rew[j].copy_(TERMS[n][2](b) * (TERMS[n][0] * K.DT))
        sp.end(cfg, env.sim, env.st, env.out, nz)
        be.sync()
        EC.compare_state(env, o, "custom step %d" % t)
        for j, n in enumerate(NAMES):
            EC.close(b.custom_sums[j], o.extra_sums[n], "episode sum of %s, step %d" % (n, t))
            if o.extras_extra is not None:
                EC.close(b.extras_custom[j], o.extras_extra[n], "extras of %s, step %d" % (n, t), rtol=1e-5, atol=1e-7)
        resets += int(o.reset.sum())
    return resets


@pytest.mark.parametrize("N,epb,nthreads", [(37, 8, 64), (64, 16, 256)])
def test_user_defined_reward_terms_host(N, epb, nthreads):
    be = EC.HostBackend(envs_per_block=epb, nthreads=nthreads)
    assert run_custom_trace(be, N, steps=16, seed=40 + N) >= 3


def test_single_launch_entry_points_refuse_custom_terms_host():
    """With user-defined terms configured only the begin / end pair may run a step (checked in the product library's launch code;
    here: the configuration fields exist and default to none)."""
    from hgym import default_env_config
    cfg = default_env_config(8)
    assert cfg.num_custom_rewards == 0 and list(cfg.custom_reward_pos) == [0] * 24


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("N", [100, 4096])
def test_user_defined_reward_terms_gpu(N):
    be = EC.HipBackend()
    assert run_custom_trace(be, N, steps=12, seed=7 + N) >= 3
    # the one-launch entry points refuse the configuration instead of silently dropping the terms
    from hgym import EnvBuffers, default_env_config, _lib as L
    cfg = default_env_config(64)
    buf = EnvBuffers(cfg, "cuda")
    buf.set_custom_rewards([0])
    a = torch.zeros(64, 12, device="cuda")
    rc = L.lib.hgym_env_step_synth(C.byref(cfg), C.byref(buf.sim_struct()), C.byref(buf.state_struct()), C.byref(buf.out_struct()), L.fptr(a),
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == -1 and b"hgym_env_step_begin" in L.lib.hgym_last_error()


@pytest.mark.gpu
def test_task_subclass_with_reward_methods_gpu():
    """The reference-shaped surface: a task class adds `_reward_<name>` methods and lists their scales in its config (what a user of
    the reference does, legged_robot.py:518-541); make_env accepts it (it used to raise NotImplementedError), step() returns
    rewards that contain the terms, extras["episode"] and episode_sums carry them, and the runner trains on it."""
    import copy
    from humanoid.envs import task_registry, XBotLFreeEnv, XBotLCfg, XBotLCfgPPO
    from humanoid.utils import get_args

    class MyEnv(XBotLFreeEnv):
        def _reward_alive(self):
            return torch.ones(self.num_envs, device=self.device)

        def _reward_zz_dof(self):
            return torch.sum(torch.square(self.dof_pos), dim=1)

    class MyCfg(XBotLCfg):
        class rewards(XBotLCfg.rewards):
            class scales(XBotLCfg.rewards.scales):
                alive = 0.7
                zz_dof = -0.02

    class PlainCfg(XBotLCfg):
        pass
    task_registry.register("my_task", MyEnv, MyCfg(), XBotLCfgPPO())
    task_registry.register("plain_task", XBotLFreeEnv, PlainCfg(), XBotLCfgPPO())
    envs = {}
    for name in ("my_task", "plain_task"):
        torch.manual_seed(3)
        np.random.seed(3)
        args = get_args(["--task=" + name, "--headless", "--num_envs", "256", "--seed", "9"])
        task_registry.train_cfgs[name].seed = 9
        envs[name], _ = task_registry.make_env(name=name, args=args)
    mine, plain = envs["my_task"], envs["plain_task"]
    assert [n for n in mine.reward_names if n not in plain.reward_names] == ["alive", "zz_dof"]
    assert "rew_alive" in mine.extras["episode"] and "zz_dof" in mine.episode_sums
    g = torch.Generator().manual_seed(1)
    for t in range(30):
        a = (torch.randn(256, 12, generator=g) * 0.5).cuda()
        _, _, r1, d1, _ = mine.step(a.clone())
        _, _, r0, d0, _ = plain.step(a.clone())
        torch.cuda.synchronize()
        assert torch.equal(d1, d0)                       # same draws, same physics: the extra terms change rewards only
        # un-clipped sum: both rewards are clipped at 0 (only_positive_rewards), so compare where neither is clipped
        want = 0.7 * mine.dt + (-0.02 * mine.dt) * torch.sum(torch.square(mine.dof_pos), dim=1)
        ok = (r0 > 1e-6) & (r1 > 1e-6) & ~d0
        np.testing.assert_allclose((r1 - r0)[ok].cpu().numpy(), want[ok].cpu().numpy(), rtol=2e-4, atol=2e-6)
    assert float(mine.episode_sums["alive"].abs().max()) > 0
    task_registry.train_cfgs["my_task"].runner.max_iterations = 2
    args = get_args(["--task=my_task", "--headless", "--num_envs", "256", "--seed", "9"])
    runner, _ = task_registry.make_alg_runner(env=mine, name="my_task", args=args, log_root=None)
    p0 = runner.alg.net.params.clone()
    runner.learn(num_learning_iterations=2, init_at_random_ep_len=True)
    torch.cuda.synchronize()
    assert torch.isfinite(runner.alg.net.params).all() and not torch.equal(p0, runner.alg.net.params)
    assert not mine.rollout_fused_supported(runner.alg.net)
