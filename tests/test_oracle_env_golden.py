"""Pins oracle/xbot_env_oracle.py against tests/golden/env_trace.npz (XBot-L defaults) and env_trace_refact.npz
(cfg.env.use_ref_actions = True), both recorded by running the reference's own XBotLFreeEnv.step
(tests/golden/gen_fixtures.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import xbot_constants as C
from oracle.xbot_env_oracle import XBotEnvOracle

T = lambda a: torch.from_numpy(np.asarray(a))


TRACES = ["env_trace.npz", "env_trace_refact.npz", "env_trace_yawrate.npz"]      # defaults | use_ref_actions | heading_command=False


def _load(golden_dir, name="env_trace.npz"):
    return np.load(os.path.join(golden_dir, name))


def _prime(G):
    N = G["friction"].shape[0]
    o = XBotEnvOracle(N, frictions=T(G["friction"]), body_mass=T(G["body_mass"]), use_ref_actions=bool(G["use_ref_actions"]),
                      heading_command=bool(G["heading_command"]))
    o.prime(T(G["prime_u_dof"]), T(G["prime_u_cmd"]), T(G["prime_z_obs"]))
    return o


@pytest.mark.parametrize("name", TRACES)
def test_prime_matches_reference(golden_dir, name):
    G = _load(golden_dir, name)
    o = _prime(G)
    assert torch.equal(o.obs, T(G["prime_obs"]))
    assert torch.equal(o.priv, T(G["prime_priv"]))
    assert torch.equal(o.commands, T(G["prime_commands"]))
    assert torch.equal(o.sim.dof_pos, T(G["prime_dof_pos"]))


@pytest.mark.parametrize("name", TRACES)
def test_trace_matches_reference(golden_dir, name):
    G = _load(golden_dir, name)
    o = _prime(G)
    o.ep_len = T(G["init_ep_len"]).clone()
    o.common_step_counter = int(G["init_common_step_counter"])
    S = G["rew"].shape[0]
    full = {int(s) for s in G["full_steps"]}
    saw = dict(reset=0, timeout=0, push=0, resample=0, stale=0)
    for t in range(S):
        a_in = T(G["actions_in"][t]).clone()
        a = o.pre_physics(a_in, T(G["u_delay"][t]), T(G["z_act"][t]))
        assert torch.equal(a_in, T(G["actions_in_after"][t])), t        # the caller's tensor: untouched, or += ref_action in place
        tq = o.pd_torques()
        o.sim.load(T(G["root"][t]), T(G["dof"][t]), T(G["contact"][t]), T(G["rigid"][t]))
        obs, priv, rew, reset, info = o.post_physics(T(G["u_cmd"][t]), T(G["u_dof"][t]), T(G["u_push"][t]), T(G["z_obs"][t]))
        # masks / indices: bit exact
        assert torch.equal(reset, T(G["reset"][t])), t
        assert torch.equal(o.time_out, T(G["time_out"][t])), t
        assert torch.equal(o.ep_len, T(G["ep_len"][t])), t
        assert info["any_reset"] == bool(G["any_reset"][t]) and info["pushed"] == bool(G["pushed"][t])
        # floats: the restatement uses the same torch ops in the same order -> identical bits
        assert torch.equal(tq, T(G["torques"][t])), t
        assert torch.equal(o.actions, T(G["actions"][t])), t
        assert torch.equal(info["frame"], T(G["frame"][t])), t
        assert torch.equal(info["priv_frame"], T(G["priv_frame"][t])), t
        assert torch.equal(rew, T(G["rew"][t])), t
        assert torch.equal(o.commands, T(G["commands"][t])), t
        assert torch.equal(o.episode_sums, T(G["episode_sums"][t])), t
        assert torch.equal(o.sim.root, T(G["root_after"][t])), t
        assert torch.equal(torch.stack((o.sim.dof_pos, o.sim.dof_vel), -1).view(-1, 2), T(G["dof_after"][t])), t
        assert torch.equal(o.extras_time_outs, T(G["extras_time_outs"][t])), t   # stale-by-design (App. A item 2)
        np.testing.assert_allclose(o.extras_episode.numpy(), G["extras_episode"][t], rtol=2e-6, atol=1e-9)
        if t in full:
            assert torch.equal(obs, T(G["obs_step%d" % t])), t
            assert torch.equal(priv, T(G["priv_step%d" % t])), t
        saw["reset"] += int(reset.sum()); saw["timeout"] += int(o.time_out.sum()); saw["push"] += int(info["pushed"])
        saw["resample"] += int((o.ep_len % C.RESAMPLE_STEPS == 0).sum() - reset.sum())
        saw["stale"] += int((not info["any_reset"]) and bool(o.extras_time_outs.any()))
    # the trace must have exercised every event class
    assert saw["reset"] > 10 and saw["timeout"] >= 2 and saw["push"] == 1
    if bool(G["use_ref_actions"]):
        assert float(np.abs(G["actions_in_after"] - G["actions_in"]).max()) > 0.5       # the feature really was on
    if not bool(G["heading_command"]):
        assert float(np.abs(G["commands"][:, :, 3]).max()) == 0.0                       # no heading target is ever drawn
        assert float(np.abs(G["commands"][:, :, 2]).max()) <= 0.3 + 1e-6
    for k in ("feet_air_time", "last_contacts", "feet_height", "last_feet_z", "last_actions", "last_last_actions",
              "last_dof_vel", "last_root_vel", "ref_dof_pos", "base_lin_vel", "base_ang_vel", "projected_gravity"):
        assert torch.equal(getattr(o, k), T(G["final_" + k])), k
    assert torch.equal(o.base_euler, T(G["final_base_euler"]))
    assert torch.equal(o.push_force, T(G["final_push_force"])) and torch.equal(o.push_torque, T(G["final_push_torque"]))


def test_constants_match_reference(golden_dir):
    import json
    K = json.load(open(os.path.join(golden_dir, "constants.json")))
    assert K["reward_names"] == C.REWARD_NAMES
    assert K["reward_scales_dt"] == C.REWARD_SCALES_DT
    assert K["dt"] == C.DT and K["max_episode_length"] == C.MAX_EPISODE_LENGTH
    assert K["resample_steps"] == C.RESAMPLE_STEPS and K["push_interval"] == C.PUSH_INTERVAL
    assert K["p_gains"] == C.P_GAINS and K["d_gains"] == C.D_GAINS
    assert np.allclose(K["torque_limits"], np.array(C.EFFORT, dtype=np.float32) * np.float32(C.TORQUE_LIMIT_FACTOR))
    assert K["base_init_state"] == [float(np.float32(x)) for x in C.BASE_INIT_STATE]
    assert K["param_counts"] == dict(actor=527244, critic=398849, std=12)
    assert K["num_obs"] == 705 and K["num_privileged_obs"] == 219
    assert (K["gamma"], K["lam"], K["clip_param"], K["entropy_coef"]) == (C.GAMMA, C.LAM, C.CLIP_PARAM, C.ENTROPY_COEF)
    assert (K["learning_rate"], K["desired_kl"], K["max_grad_norm"]) == (C.LEARNING_RATE, C.DESIRED_KL, C.MAX_GRAD_NORM)
    assert K["actor_hidden_dims"] == C.ACTOR_HIDDEN and K["critic_hidden_dims"] == C.CRITIC_HIDDEN
    assert K["num_steps_per_env"] == C.NUM_STEPS_PER_ENV and K["seed"] == C.SEED


# ------------------------------------------------------------------------------------------------
# third trace: the generic LeggedRobot options XBot-L leaves off -- terrain map with custom origins,
# terrain curriculum, height measurements, command curriculum
def terrain_spec(G):
    from oracle.xbot_env_oracle import TerrainSpec
    return TerrainSpec(T(G["terrain_origins"]), T(G["terrain_levels0"]), T(G["terrain_types"]), float(G["terrain_env_length"]), True,
                       height_samples=T(G["height_samples"]), height_points=T(G["height_points"]),
                       border_size=float(G["terrain_border"]), horizontal_scale=float(G["terrain_hscale"]),
                       vertical_scale=float(G["terrain_vscale"]))


def test_generic_trace_matches_reference(golden_dir):
    G = _load(golden_dir, "env_trace_generic.npz")
    N = G["friction"].shape[0]
    o = XBotEnvOracle(N, frictions=T(G["friction"]), body_mass=T(G["body_mass"]), terrain=terrain_spec(G), command_curriculum=True,
                      max_curriculum=float(G["max_curriculum"]))
    assert torch.equal(o.env_origins, T(G["env_origins0"]))
    o.prime(T(G["prime_u_dof"]), T(G["prime_u_cmd"]), T(G["prime_z_obs"]), T(G["prime_u_xy"]), T(G["prime_r_level"]))
    assert torch.equal(o.obs, T(G["prime_obs"])) and torch.equal(o.priv, T(G["prime_priv"]))
    assert torch.equal(o.sim.root, T(G["prime_root"]))                  # spawn jitter applied, levels untouched by the first reset
    assert torch.equal(o.terrain.levels, T(G["terrain_levels0"]))
    o.ep_len = T(G["init_ep_len"]).clone()
    o.common_step_counter = int(G["init_common_step_counter"])
    o.episode_sums = T(G["init_episode_sums"]).clone()
    ups = downs = redraws = 0
    for t in range(G["rew"].shape[0]):
        a = o.pre_physics(T(G["actions_in"][t]).clone(), T(G["u_delay"][t]), T(G["z_act"][t]))
        o.pd_torques()
        o.sim.load(T(G["root"][t]), T(G["dof"][t]), T(G["contact"][t]), T(G["rigid"][t]))
        before = o.terrain.levels.clone()
        obs, priv, rew, reset, info = o.post_physics(T(G["u_cmd"][t]), T(G["u_dof"][t]), T(G["u_push"][t]), T(G["z_obs"][t]),
                                                     T(G["u_xy"][t]), T(G["r_level"][t]))
        assert torch.equal(reset, T(G["reset"][t])), t
        assert torch.equal(o.terrain.levels, T(G["terrain_levels"][t])), t
        assert torch.equal(o.env_origins, T(G["env_origins"][t])), t
        assert torch.equal(o.measured_heights, T(G["measured_heights"][t])), t
        assert o.cmd_range_x == [float(v) for v in G["cmd_range_x"][t]], t
        assert torch.equal(o.commands, T(G["commands"][t])), t
        assert torch.equal(o.sim.root, T(G["root_after"][t])), t
        assert torch.equal(info["frame"], T(G["frame"][t])) and torch.equal(info["priv_frame"], T(G["priv_frame"][t])), t
        assert torch.equal(rew, T(G["rew"][t])) and torch.equal(o.episode_sums, T(G["episode_sums"][t])), t
        d = o.terrain.levels - before
        ups += int((d > 0).sum()); downs += int((d < 0).sum())
        redraws += int(((before == o.terrain.max_level - 1) & reset & (o.terrain.levels == T(G["r_level"][t])) & (d <= 0)).sum())
    assert ups >= 5 and downs >= 5                                          # both curriculum directions exercised
    assert o.cmd_range_x == [-0.8, 1.0] and [float(v) for v in G["cmd_range_x"][0]] == [-0.3, 0.6]   # the command range widened once
    assert float(np.abs(G["measured_heights"]).max()) > 0.01                # heights really sampled a non-flat map


# ------------------------------------------------------------------------------------------------
# LeggedRobot.reset(): reset_idx(all) + a zero-action step, recorded from the reference on an env that has stepped
def test_reset_trace_matches_reference(golden_dir):
    """legged_robot.py:110-115 / base_task.py:140-145 on env_reset_trace.npz: the warm-up steps, the state right after reset_idx(all)
    (what the reference's zero-action step finds at its entry) and everything that step leaves -- bit for bit."""
    G = _load(golden_dir, "env_reset_trace.npz")
    N = G["friction"].shape[0]
    o = XBotEnvOracle(N, frictions=T(G["friction"]), body_mass=T(G["body_mass"]))
    o.prime(T(G["prime_u_dof"]), T(G["prime_u_cmd"]), T(G["prime_z_obs"]))
    o.ep_len = T(G["init_ep_len"]).clone()
    o.common_step_counter = int(G["init_common_step_counter"])

    def step(pre, a_in):
        o.pre_physics(a_in, T(G[pre + "u_delay"]), T(G[pre + "z_act"]))
        tq = o.pd_torques()
        o.sim.load(T(G[pre + "root"]), T(G[pre + "dof"]), T(G[pre + "contact"]), T(G[pre + "rigid"]))
        obs, priv, rew, reset, info = o.post_physics(T(G[pre + "u_cmd"]), T(G[pre + "u_dof"]), T(G[pre + "u_push"]), T(G[pre + "z_obs"]))
        return tq, obs, priv, rew, reset

    def check(tag, res, g):
        tq, obs, priv, rew, reset = res
        assert torch.equal(reset, T(g("reset"))) and torch.equal(o.time_out, T(g("time_out"))) and torch.equal(o.ep_len, T(g("ep_len"))), tag
        assert torch.equal(tq, T(g("torques"))) and torch.equal(o.actions, T(g("actions"))), tag
        assert torch.equal(rew, T(g("rew"))) and torch.equal(o.commands, T(g("commands"))) and torch.equal(o.episode_sums, T(g("episode_sums"))), tag
        assert torch.equal(torch.clip(obs, -C.CLIP_OBS, C.CLIP_OBS), T(g("obs"))) and torch.equal(torch.clip(priv, -C.CLIP_OBS, C.CLIP_OBS), T(g("priv"))), tag
        assert torch.equal(o.sim.root, T(g("root_after"))), tag
        assert torch.equal(torch.stack((o.sim.dof_pos, o.sim.dof_vel), -1).view(-1, 2), T(g("dof_after"))), tag
        assert torch.equal(o.extras_time_outs, T(g("extras_time_outs"))), tag
        np.testing.assert_allclose(o.extras_episode.numpy(), g("extras_episode"), rtol=2e-6, atol=1e-9)

    S0 = G["warm_rew"].shape[0]
    for t in range(S0):
        o_pre = lambda k, t=t: G["warm_" + k][t]
        o.pre_physics(T(o_pre("actions_in")).clone(), T(o_pre("u_delay")), T(o_pre("z_act")))
        tq = o.pd_torques()
        o.sim.load(T(o_pre("root")), T(o_pre("dof")), T(o_pre("contact")), T(o_pre("rigid")))
        obs, priv, rew, reset, info = o.post_physics(T(o_pre("u_cmd")), T(o_pre("u_dof")), T(o_pre("u_push")), T(o_pre("z_obs")))
        check("warm-up step %d" % t, (tq, obs, priv, rew, reset), o_pre)
    assert int(G["warm_reset"].sum()) >= 2 and int(G["warm_time_out"].sum()) >= 1
    # reset_idx(arange(N)): legged_robot.py:163-215 for every env
    o._reset_masked(torch.ones(N, dtype=torch.bool), T(G["reset_u_dof"]), T(G["reset_u_cmd"]))
    assert torch.equal(o.commands, T(G["entry_commands"])) and torch.equal(o.ep_len, T(G["entry_ep_len"]))
    assert torch.equal(o.sim.root, T(G["entry_root"]))
    assert torch.equal(torch.stack((o.sim.dof_pos, o.sim.dof_vel), -1).view(-1, 2), T(G["entry_dof"]))
    assert torch.equal(o.episode_sums, T(G["entry_episode_sums"])) and float(o.episode_sums.abs().max()) == 0.0
    assert torch.equal(o.last_actions, T(G["entry_last_actions"])) and torch.equal(o.last_dof_vel, T(G["entry_last_dof_vel"]))
    assert torch.equal(o.feet_air_time, T(G["entry_feet_air_time"]))
    assert float(o.obs_hist.abs().max()) == float(G["entry_obs_history_absmax"]) == 0.0
    assert float(o.priv_hist.abs().max()) == float(G["entry_critic_history_absmax"]) == 0.0
    np.testing.assert_allclose(o.extras_episode.numpy(), G["entry_extras_episode"], rtol=2e-6, atol=1e-9)
    # ... + the zero-action step
    res = step("step_", torch.zeros(N, 12))
    check("zero-action step", res, lambda k: G["step_" + k])
    assert int(G["step_reset"].sum()) == 2 and int(G["step_ep_len"].min()) == 0 and int(G["step_ep_len"].max()) == 1
