"""-m gpu: the env kernels of libhgym_hip.so on a real MI355X, through the C-ABI, against the oracle
(identical seeded inputs and noise tables) and against the golden trace recorded from the reference."""
import os

import numpy as np
import pytest
import torch

import env_common as EC

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return EC.HipBackend()


@pytest.mark.parametrize("name", ["env_trace.npz", "env_trace_refact.npz", "env_trace_yawrate.npz"])
def test_golden_trace_gpu(hip, golden_dir, name):
    """The traces recorded from the reference's own XBotLFreeEnv.step: XBot-L defaults, and cfg.env.use_ref_actions = True."""
    G = np.load(os.path.join(golden_dir, name))
    N = G["friction"].shape[0]
    env = EC.EnvUnderTest(hip, N, T(G["friction"]), T(G["body_mass"]), sim_layout="aos", use_ref_actions=bool(G["use_ref_actions"]),
                          heading_command=bool(G["heading_command"]))
    env.prime(T(G["prime_u_dof"]), T(G["prime_u_cmd"]), T(G["prime_z_obs"]))
    hip.sync()
    EC.close(env.buf.obs, G["prime_obs"], "prime obs")
    EC.close(env.buf.priv_obs, G["prime_priv"], "prime priv")
    env.buf.episode_length.copy_(T(G["init_ep_len"]))
    env.buf.counters[0] = int(G["init_common_step_counter"])
    full = {int(s) for s in G["full_steps"]}
    for t in range(G["rew"].shape[0]):
        frame = (T(G["root"][t]), T(G["dof"][t]), T(G["contact"][t]), T(G["rigid"][t]))
        env.step(T(G["actions_in"][t]), frame, T(G["u_delay"][t]), T(G["z_act"][t]), T(G["u_cmd"][t]), T(G["u_dof"][t]),
                 T(G["u_push"][t]), T(G["z_obs"][t]))
        b = env.buf
        EC.close(env.actions_after, G["actions_in_after"][t], "caller's action tensor %d" % t)
        EC.exact(b.reset, G["reset"][t], "reset %d" % t)                       # bit-exact masks
        EC.exact(b.time_out, G["time_out"][t], "time_out %d" % t)
        EC.exact(b.episode_length, G["ep_len"][t], "ep_len %d" % t)
        EC.exact(b.extras_time_outs, G["extras_time_outs"][t], "extras time_outs %d" % t)
        EC.close(b.rew, G["rew"][t], "rew %d" % t)                              # 1e-5 relative
        EC.close(b.view("torques"), G["torques"][t], "torques %d" % t)
        EC.close(b.view("actions"), G["actions"][t], "actions %d" % t)
        EC.close(b.view("commands"), G["commands"][t], "commands %d" % t)
        EC.close(b.view("episode_sums"), G["episode_sums"][t], "episode_sums %d" % t)
        EC.close(b.root, G["root_after"][t], "root %d" % t)
        EC.close(b.dof_state, G["dof_after"][t], "dof %d" % t)
        EC.close(b.extras_episode, G["extras_episode"][t], "extras episode %d" % t, rtol=1e-5, atol=1e-7)
        if t in full:
            EC.close(b.obs, G["obs_step%d" % t], "obs %d" % t)
            EC.close(b.priv_obs, G["priv_step%d" % t], "priv %d" % t)


def test_reset_golden_trace_gpu(hip, golden_dir):
    """SURVEY.md 8a row E14 on the device: LeggedRobot.reset() = hgym_env_reset_all + a zero-action step, replayed from the trace recorded
    from the reference's own reset() (legged_robot.py:110-115; tests/golden/gen_fixtures.py::gen_env_reset_trace)."""
    EC.run_reset_golden(hip, golden_dir)


def test_reset_all_then_step_gpu(hip):
    """The same against the oracle on seeded inputs (the -m gpu twin of test_env_hostcheck.py::test_reset_all_then_step_host), at a
    size that spans several workgroups, plus reset() with the generic options on (terrain curriculum inside reset_idx(all))."""
    N = 300
    counts, env, o = EC.run_random_trace(hip, N, steps=3, seed=5)
    g = torch.Generator().manual_seed(1)
    u_dof, u_cmd3 = torch.rand(N, 12, generator=g), torch.rand(N, 3, generator=g)
    o._reset_masked(torch.ones(N, dtype=torch.bool), u_dof, u_cmd3)
    env.reset_all(u_dof, u_cmd3)
    hip.sync()
    assert float(env.buf.obs_ring.abs().max()) == 0.0 and float(env.buf.priv_ring.abs().max()) == 0.0
    EC.close(env.buf.view("commands"), o.commands, "commands after reset_all")
    EC.exact(env.buf.episode_length, o.ep_len, "ep_len after reset_all")
    EC.close(env.buf.root_view(), o.sim.root, "root after reset_all")
    frame = EC.synth_frames(g, N)
    a = torch.zeros(N, 12)
    nz = [torch.rand(N, generator=g), torch.randn(N, 12, generator=g), torch.rand(N, 6, generator=g),
          torch.rand(N, 12, generator=g), torch.rand(N, 5, generator=g), torch.randn(N, 47, generator=g)]
    o.pre_physics(a, nz[0], nz[1]); o.pd_torques(); o.sim.load(*frame); o.post_physics(*nz[2:])
    env.step(a, frame, *nz)
    EC.compare_state(env, o, "step after reset_all")
    # generic options: the terrain curriculum also runs in reset_idx(all)
    counts, env, o = EC.run_random_trace(hip, 44, steps=6, seed=321, sim_layout="soa", generic=True, track_sum=5.0)
    g = torch.Generator().manual_seed(2)
    N = 44
    u_dof, u_cmd3 = torch.rand(N, 12, generator=g), torch.rand(N, 3, generator=g)
    u_xy, r_level = torch.rand(N, 2, generator=g), torch.randint(0, 5, (N,), generator=g)
    o._reset_masked(torch.ones(N, dtype=torch.bool), u_dof, u_cmd3, u_xy, r_level)
    env.reset_all(u_dof, u_cmd3, u_xy, r_level)
    hip.sync()
    EC.exact(env.buf.terrain_levels, o.terrain.levels, "levels after reset_all")
    EC.close(env.buf.root_view(), o.sim.root, "root after reset_all (generic)")
    EC.close(env.buf.view("commands"), o.commands, "commands after reset_all (generic)")


@pytest.mark.parametrize("N,layout,steps", [(37, "soa", 20), (256, "aos", 16), (4096, "soa", 12), (8192, "soa", 8)])
def test_random_trace_gpu(hip, N, layout, steps):
    counts, env, o = EC.run_random_trace(hip, N, steps=steps, seed=100 + N, sim_layout=layout,
                                         check_every=1 if N < 1000 else 4)
    assert counts["push"] == 1 and counts["timeout"] >= 1 and counts["reset"] >= 3


def test_generic_frame_stack_gpu(hip):
    EC.run_random_trace(hip, 100, steps=10, seed=7, frame_stack=4, c_frame_stack=2)


def test_use_ref_actions_random_trace_gpu(hip):
    """SURVEY.md 8f item 3, cfg.env.use_ref_actions: component-wise entry points against the oracle, incl. the in-place
    mutation of the caller's action tensor."""
    EC.run_random_trace(hip, 300, steps=12, seed=21, use_ref_actions=True)


@pytest.mark.parametrize("N", [4096, 8192, 5000, 16384])
def test_fused_synthetic_step_properties(hip, N):
    """BASELINE configs[1] (4096 envs) and configs[3] (8192 envs/GPU, stack 15), a ragged count, and 16 384 envs (four
    workgroups per CU).  Full-size fused fast path (pre_physics + synthetic physics + post_physics in one launch, internal
    Philox): size-independent properties instead of an element-wise oracle --
    obs rows are the shifted previous rows plus a new frame, masks are consistent, counters advance."""
    import ctypes as C
    from hgym import EnvBuffers, default_env_config, _lib as L
    cfg = default_env_config(N, seed=123)
    buf = EnvBuffers(cfg, "cuda")
    sim, st, out = buf.sim_struct(), buf.state_struct(), buf.out_struct()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    noise = buf.noise_struct()
    L.check(L.lib.hgym_env_prime(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), C.byref(noise), s))
    buf.episode_length.copy_(torch.randint(0, 2400, (N,)).cuda())
    prev_obs = buf.obs.clone()
    prev_ep = buf.episode_length.clone()
    total_resets = 0
    for t in range(30):
        a = torch.randn(N, 12, device="cuda")
        L.check(L.lib.hgym_env_step_synth(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), L.fptr(a), s))
        torch.cuda.synchronize()
        obs, reset = buf.obs, buf.reset
        assert torch.isfinite(obs).all() and torch.isfinite(buf.priv_obs).all() and torch.isfinite(buf.rew).all()
        assert float(obs.abs().max()) <= 18.0 and float(buf.rew.min()) >= 0.0
        keep = ~reset
        # history shift: frames 0..13 of the new obs are frames 1..14 of the previous obs for non-reset envs
        assert torch.equal(obs[keep, : 14 * 47], prev_obs[keep, 47:])
        # reset envs: history zeroed, episode length back to 0, others advanced by exactly 1
        assert float(obs[reset, : 14 * 47].abs().max()) == 0.0 if bool(reset.any()) else True
        assert torch.equal(buf.episode_length[keep], prev_ep[keep] + 1)
        assert int(buf.episode_length[reset].abs().sum()) == 0
        assert bool((buf.time_out <= reset).all())
        total_resets += int(reset.sum())
        prev_obs, prev_ep = obs.clone(), buf.episode_length.clone()
    assert int(buf.counters[0]) == 30 and int(buf.counters[2]) == 31
    assert total_resets > 0


# ------------------------------------------------------------------------------------------------ generic options
def test_generic_options_golden_trace_gpu(hip, golden_dir):
    """terrain map (custom origins, terrain curriculum, height measurements) and command curriculum --
    the trace recorded from the reference with those options on, through the HIP kernels."""
    EC.run_generic_golden(hip, golden_dir)


@pytest.mark.parametrize("N,track_sum,moves", [(44, 40.0, 1), (1500, 40.0, 1), (300, 5.0, 0)])
def test_generic_options_random_trace_gpu(hip, N, track_sum, moves):
    counts, env, o = EC.run_random_trace(hip, N, steps=10, seed=321 + N, generic=True, track_sum=track_sum)
    assert counts["range_moves"] == moves and counts["level_up"] >= 3 and counts["level_down"] >= 3


def test_generic_options_fused_step_and_env_surface():
    """The options through the reference-shaped env (mesh_type='trimesh' with HumanoidTerrain, both curricula,
    measure_heights) on the fused fast path with internal Philox: properties instead of an element-wise oracle."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "humanoid-gym_amd"))
    from humanoid.envs import task_registry
    from humanoid.utils import get_args
    args = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", "512"])
    env_cfg, _ = task_registry.get_cfgs("humanoid_ppo")
    import copy
    env_cfg = copy.deepcopy(env_cfg)
    t = env_cfg.terrain
    t.mesh_type, t.curriculum, t.measure_heights, t.num_rows, t.num_cols, t.border_size = "trimesh", True, True, 5, 4, 5
    t.max_init_terrain_level = 2
    env_cfg.commands.curriculum = True
    np.random.seed(3)
    torch.manual_seed(3)
    env, _ = task_registry.make_env(name="humanoid_ppo", args=args, env_cfg=env_cfg)
    N = env.num_envs
    assert env.custom_origins and env.height_samples.shape == (env.terrain.tot_rows, env.terrain.tot_cols)
    assert env.measured_heights.shape == (N, 17 * 11) and env.terrain_origins.shape == (5, 4, 3)
    lv0 = env.terrain_levels.clone()
    assert int(lv0.max()) <= 2
    origin_of = lambda: env.terrain_origins[env.terrain_levels, env.terrain_types]
    assert torch.equal(env.env_origins, origin_of())
    # spawn jitter: every env within 1 m (per axis) of its tile origin, not on it
    d = env.root_states[:, :2] - env.env_origins[:, :2]
    assert float(d.abs().max()) <= 1.0 and float(d.abs().mean()) > 0.2
    env.episode_length_buf = torch.randint(2300, 2400, (N,), device="cuda")     # many time-outs soon
    env.common_step_counter = 2390                                               # command-curriculum check at step 10
    env.episode_sums["tracking_lin_vel"][:] = 400.0       # far above the 23.04 threshold even if some of the resetting envs reset before
    assert env.command_ranges["lin_vel_x"] == [-0.3, 0.6]
    saw_heights = 0.0
    for k in range(40):
        obs, priv, rew, dones, extras = env.step(torch.randn(N, 12, device="cuda") * 0.3)
        assert torch.equal(env.env_origins, origin_of())                        # origins follow the levels
        assert int(env.terrain_levels.min()) >= 0 and int(env.terrain_levels.max()) < 5
        saw_heights = max(saw_heights, float(env.measured_heights.abs().max()))
        # every sampled height is a value of the map (times the vertical scale)
        cells = torch.round(env.measured_heights / t.vertical_scale).to(torch.int16)
        assert bool(torch.isin(cells, env.height_samples.unique()).all())
    assert saw_heights > 0.0
    assert env.command_ranges["lin_vel_x"] == [-0.8, 1.0]                        # widened once, capped by max_curriculum = 1
    assert float(extras["episode"]["max_command_x"]) == 1.0
    assert abs(float(extras["episode"]["terrain_level"]) - float(env.terrain_levels.float().mean())) < 1e-6
    assert not torch.equal(env.terrain_levels, lv0)                              # the synthetic walkers stay put: demotions
    # host-callable _get_heights agrees with what the step sampled for envs that did not reset in the last step
    h = env._get_heights()
    keep = ~dones
    # (the step samples before the physics of the NEXT step moves the base: compare on the current pose directly)
    assert h.shape == env.measured_heights.shape and bool(torch.isfinite(h).all())


def test_yaw_rate_commands_random_trace_gpu(hip):
    """cfg.commands.heading_command = False, alone and together with the terrain / curriculum options."""
    EC.run_random_trace(hip, 300, steps=10, seed=77, heading_command=False)
    EC.run_random_trace(hip, 300, steps=8, seed=78, generic=True, heading_command=False)
