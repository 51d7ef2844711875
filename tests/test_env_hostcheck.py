"""CPU-only: the product's env kernel SOURCE (compiled for the host by tests/hostcheck) against the oracle and
against the golden trace recorded from the reference.  Catches arithmetic / indexing / ring-buffer mistakes
without a GPU; the `-m gpu` twin of this file (test_env_gpu.py) runs the real kernels."""
import os

import ctypes as C
import numpy as np
import pytest
import torch

import env_common as EC

T = lambda a: torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module", params=["chain", "split", "split3", "split3r"])
def host(request):
    """chain: the monolithic per-env chain; split: the phase sequence of the XBot-L fast kernels (per-joint work on (env, joint)
    lanes before and after a shorter chain).  Traces with generic options fall back to the chain inside the kernel source."""
    return EC.HostBackend(envs_per_block=8, nthreads=64, split={"chain": 0, "split": 1, "split3": 2, "split3r": 3}[request.param])


@pytest.mark.parametrize("name", ["env_trace.npz", "env_trace_refact.npz", "env_trace_yawrate.npz"])
def test_golden_trace_host(host, golden_dir, name):
    """The reference's own recorded traces (XBot-L defaults; cfg.env.use_ref_actions = True) through the kernel source."""
    G = np.load(os.path.join(golden_dir, name))
    N = G["friction"].shape[0]
    env = EC.EnvUnderTest(host, N, T(G["friction"]), T(G["body_mass"]), sim_layout="aos", use_ref_actions=bool(G["use_ref_actions"]),
                          heading_command=bool(G["heading_command"]))
    env.prime(T(G["prime_u_dof"]), T(G["prime_u_cmd"]), T(G["prime_z_obs"]))
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
        EC.exact(b.reset, G["reset"][t], "reset %d" % t)
        EC.exact(b.time_out, G["time_out"][t], "time_out %d" % t)
        EC.exact(b.episode_length, G["ep_len"][t], "ep_len %d" % t)
        EC.exact(b.extras_time_outs, G["extras_time_outs"][t], "extras time_outs %d" % t)
        EC.close(b.rew, G["rew"][t], "rew %d" % t)
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
    for k in ("feet_air_time", "feet_height", "last_feet_z", "last_actions", "last_last_actions", "last_dof_vel",
              "last_root_vel", "ref_dof_pos", "base_lin_vel", "base_ang_vel", "projected_gravity"):
        EC.close(env.buf.view(k), G["final_" + k], k)
    EC.close(env.buf.view("base_euler"), G["final_base_euler"], "base_euler")


def test_generic_options_golden_trace_host(host, golden_dir):
    """Terrain map (custom origins, terrain curriculum, height measurements) + command curriculum: the trace recorded from
    the reference with those options on."""
    EC.run_generic_golden(host, golden_dir)


@pytest.mark.parametrize("split", [0, 1, 2, 3])
@pytest.mark.parametrize("N,layout,epb,nthreads", [(37, "soa", 8, 64), (64, "aos", 16, 256), (5, "soa", 4, 32)])
def test_random_trace_host(N, layout, epb, nthreads, split):
    be = EC.HostBackend(envs_per_block=epb, nthreads=nthreads, split=split)
    counts, env, o = EC.run_random_trace(be, N, steps=24, seed=100 + N, sim_layout=layout)
    assert counts["push"] == 1 and counts["timeout"] >= 1 and counts["reset"] >= 3


@pytest.mark.parametrize("split", [0, 1, 2, 3])
@pytest.mark.parametrize("N,epb,nthreads", [(37, 8, 64), (64, 32, 512), (5, 4, 32)])
def test_rows_written_one_step_ahead_host(N, epb, nthreads, split):
    """hgym_rollout_step's rows-ahead protocol (HgymEnvOut.obs_ahead / priv_ahead / obs_older_ready, header v5) on the kernel source:
    every step writes the older frames of the rows AFTER next through the device's item geometry (hist_slot<H, F, 2>) plus its own frame,
    and from the second step on copies no older frames into its own rows.  The stacked observations are compared with the oracle
    after every step (a frame nobody wrote would be NaN), over time-outs / resets -- whose pre-written frames have to come out
    zero -- a push and command resampling; 34 steps: the 15-slot ring goes round more than twice."""
    be = EC.HostBackend(envs_per_block=epb, nthreads=nthreads, split=split)
    counts, env, o = EC.run_random_trace(be, N, steps=34, seed=500 + N, rows_ahead=True)
    assert counts["push"] == 1 and counts["timeout"] >= 1 and counts["reset"] >= 3
    assert env._k == 34 and env._ready


@pytest.mark.parametrize("track_sum,moves", [(40.0, 1), (5.0, 0)])
def test_generic_options_random_trace_host(track_sum, moves):
    """Oracle vs kernel source with the terrain map, both curricula and the height measurements on; the command range widens
    exactly when the resetting envs' mean tracking sum is above 80 % of the maximum (23.04)."""
    be = EC.HostBackend(envs_per_block=8, nthreads=64)
    counts, env, o = EC.run_random_trace(be, 44, steps=14, seed=321, sim_layout="soa", generic=True, track_sum=track_sum)
    assert counts["range_moves"] == moves and counts["level_up"] >= 3 and counts["level_down"] >= 3 and counts["push"] == 1
    assert o.cmd_range_x == ([-0.8, 1.1] if moves else [-0.3, 0.6])
    # LeggedRobot.reset() with the options on: the terrain curriculum also runs in reset_idx(all)
    g = torch.Generator().manual_seed(2)
    N = 44
    u_dof, u_cmd3 = torch.rand(N, 12, generator=g), torch.rand(N, 3, generator=g)
    u_xy, r_level = torch.rand(N, 2, generator=g), torch.randint(0, 5, (N,), generator=g)
    o._reset_masked(torch.ones(N, dtype=torch.bool), u_dof, u_cmd3, u_xy, r_level)
    env.reset_all(u_dof, u_cmd3, u_xy, r_level)
    EC.exact(env.buf.terrain_levels, o.terrain.levels, "levels after reset_all")
    EC.close(env.buf.root_view(), o.sim.root, "root after reset_all")
    EC.close(env.buf.view("commands"), o.commands, "commands after reset_all")


def test_yaw_rate_commands_random_trace_host():
    """cfg.commands.heading_command = False (legged_robot.py:311-314,333-334) together with the other generic options."""
    be = EC.HostBackend(envs_per_block=8, nthreads=64)
    EC.run_random_trace(be, 20, steps=10, seed=77, heading_command=False)
    EC.run_random_trace(be, 44, steps=8, seed=78, generic=True, heading_command=False)


def test_generic_frame_stack_host():
    """frame_stack / c_frame_stack other than 15/3 take the runtime-sized kernel instantiation."""
    be = EC.HostBackend(envs_per_block=4, nthreads=64)
    EC.run_random_trace(be, 12, steps=12, seed=7, frame_stack=4, c_frame_stack=2)


def test_reset_golden_trace_host(host, golden_dir):
    """LeggedRobot.reset() as recorded from the reference (tests/golden/env_reset_trace.npz) through the kernel source."""
    EC.run_reset_golden(host, golden_dir)


def test_reset_all_then_step_host():
    """LeggedRobot.reset(): reset_idx(all) + a zero-action step (legged_robot.py:112-117)."""
    be = EC.HostBackend()
    counts, env, o = EC.run_random_trace(be, 16, steps=3, seed=5)
    g = torch.Generator().manual_seed(1)
    N = 16
    u_dof, u_cmd3 = torch.rand(N, 12, generator=g), torch.rand(N, 3, generator=g)
    o._reset_masked(torch.ones(N, dtype=torch.bool), u_dof, u_cmd3)
    env.reset_all(u_dof, u_cmd3)
    assert float(env.buf.obs_ring.abs().max()) == 0.0 and float(env.buf.priv_ring.abs().max()) == 0.0
    EC.close(env.buf.view("commands"), o.commands, "commands after reset_all")
    EC.exact(env.buf.episode_length, o.ep_len, "ep_len after reset_all")
    frame = EC.synth_frames(g, N)
    a = torch.zeros(N, 12)
    nz = [torch.rand(N, generator=g), torch.randn(N, 12, generator=g), torch.rand(N, 6, generator=g),
          torch.rand(N, 12, generator=g), torch.rand(N, 5, generator=g), torch.randn(N, 47, generator=g)]
    o.pre_physics(a, nz[0], nz[1]); o.pd_torques(); o.sim.load(*frame); o.post_physics(*nz[2:])
    env.step(a, frame, *nz)
    EC.compare_state(env, o, "step after reset_all")


@pytest.mark.parametrize("N,nthreads,resets", [(4096, 512, 7), (4096, 1024, 0), (40, 64, 3), (8, 256, 1), (37, 64, 2)])
def test_finaliser_one_pass_form_equals_the_general_one_host(N, nthreads, resets):
    """csrc/hgym_finalize.hpp: fin_fused (a lane owns 8 consecutive envs, every load before its first store, 64- / 128-bit accesses; what the
    device runs when N % 8 == 0) against fin_part1 + fin_store on the same inputs: refreshed extras["time_outs"] (only when an env reset),
    extras["episode"], the emptied accumulators and the transition sink's bootstrapped rewards / dones -- bit for bit.  N = 37: declines."""
    from hgym import EnvBuffers, default_env_config
    be = EC.HostBackend(envs_per_block=8, nthreads=64)
    g = torch.Generator().manual_seed(N + nthreads)
    outs = []
    for fused in (0, 1):
        cfg = default_env_config(N, seed=1)
        buf = EnvBuffers(cfg, "cpu")
        gg = torch.Generator().manual_seed(N * 7 + 1)
        buf.rew.copy_(torch.randn(N, generator=gg))
        buf.reset.copy_(torch.rand(N, generator=gg) < 0.3)
        buf.time_out.copy_(torch.rand(N, generator=gg) < 0.4)
        buf.extras_time_outs.copy_(torch.rand(N, generator=gg) < 0.5)
        buf.episode_acc.copy_(torch.randn(24, generator=gg))
        buf.counters[1] = resets
        sink = dict(values=torch.randn(N, generator=gg), rewards=torch.full((N,), float("nan")), dones=torch.zeros(N, dtype=torch.bool),
                    step=torch.zeros(1, dtype=torch.int64), gamma=0.994)
        st, out = buf.state_struct(), buf.out_struct(sink=sink)
        rc = be.lib.hc_finalize_forms(C.byref(cfg), C.byref(st), C.byref(out), nthreads, fused)
        if fused and N % 8:
            assert rc == 1          # declined, nothing touched
            assert torch.isnan(sink["rewards"]).all()
            return
        assert rc == 0
        outs.append((buf.extras_time_outs.clone(), buf.extras_episode.clone(), buf.episode_acc.clone(), sink["rewards"].clone(), sink["dones"].clone(),
                     buf.time_out.clone(), buf.reset.clone(), buf.rew.clone(), sink["values"].clone()))
    a, b = outs
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    eto, eep, acc, rw, dn, to, rs, rew, val = a
    assert torch.equal(dn, rs)
    if resets > 0:
        assert torch.equal(eto, to) and bool((acc[:22] == 0).all())
    boot = val * eto.float()
    assert torch.equal(rw, rew + torch.tensor(0.994, dtype=torch.float32) * boot)


@pytest.mark.parametrize("N,nthreads,resets", [(4096, 512, 5), (4096, 1024, 0), (40, 64, 3), (37, 64, 2)])
def test_finaliser_deferred_sink_stores_raw_rewards_and_bootstrap_flags_host(N, nthreads, resets):
    """The transition sink of the deferred kind (HgymEnvOut.t_values NULL, t_time_outs set; header v7): both forms of the finaliser store
    the RAW reward, the dones and the flags the bootstrap would have used -- the stale-by-design extras["time_outs"] AFTER this step's
    refresh (only when an env reset: legged_robot.py:173-174) -- bit for bit the same, and `rew + gamma * (V * flags)` formed later is
    exactly what the immediate sink stores (ppo.py:107-108)."""
    from hgym import EnvBuffers, default_env_config
    be = EC.HostBackend(envs_per_block=8, nthreads=64)
    outs = []
    for kind in ("deferred general", "deferred one-pass", "immediate"):
        cfg = default_env_config(N, seed=1)
        buf = EnvBuffers(cfg, "cpu")
        gg = torch.Generator().manual_seed(N * 11 + 3)
        buf.rew.copy_(torch.randn(N, generator=gg))
        buf.reset.copy_(torch.rand(N, generator=gg) < 0.3)
        buf.time_out.copy_(torch.rand(N, generator=gg) < 0.4)
        buf.extras_time_outs.copy_(torch.rand(N, generator=gg) < 0.5)
        buf.episode_acc.copy_(torch.randn(24, generator=gg))
        buf.counters[1] = resets
        values = torch.randn(N, generator=gg)
        sink = dict(values=None if kind != "immediate" else values, rewards=torch.full((N,), float("nan")), dones=torch.zeros(N, dtype=torch.bool),
                    time_outs=torch.full((N,), 7, dtype=torch.uint8), step=torch.zeros(1, dtype=torch.int64), gamma=0.994)
        st, out = buf.state_struct(), buf.out_struct(sink=sink)
        rc = be.lib.hc_finalize_forms(C.byref(cfg), C.byref(st), C.byref(out), nthreads, int(kind == "deferred one-pass"))
        if kind == "deferred one-pass" and N % 8:
            assert rc == 1 and torch.isnan(sink["rewards"]).all() and bool((sink["time_outs"] == 7).all())
            outs.append(None)
            continue
        assert rc == 0
        outs.append((sink["rewards"].clone(), sink["dones"].clone(), sink["time_outs"].clone(), buf.extras_time_outs.clone(), buf.rew.clone(), values))
    gen, one, imm = outs
    rw, dn, to, eto, rew, val = gen
    assert torch.equal(rw, rew) and torch.equal(to.bool(), eto.bool()) and bool((to <= 1).all())
    if one is not None:
        for x, y in zip(gen[:4], one[:4]):
            assert torch.equal(x, y)
    assert torch.equal(imm[1], dn) and bool((imm[2] == 7).all())                     # the immediate sink never touches t_time_outs
    assert torch.equal(imm[0], rew + torch.tensor(0.994, dtype=torch.float32) * (val * to.float()))
