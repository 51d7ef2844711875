#!/usr/bin/env python
"""Generate the committed golden vectors by RUNNING THE REFERENCE ITSELF (unmodified, on CPU).

    cd /root/repo && python tests/golden/gen_fixtures.py

Needs /root/reference (present only in the build container).  Outputs (all small, committed):
  tests/golden/constants.json       config-derived constants
  tests/golden/gae.npz              RolloutStorage.compute_returns: the §8c known answer + a seeded case
  tests/golden/policy_example.npz   weights of logs/XBot_ppo/exported/policies/policy_example.pt (trained
                                    actor, a data artefact not source) + its outputs on fixed inputs
  tests/golden/env_trace.npz        XBotLFreeEnv.step over a seeded synthetic sim trace: inputs, every RNG
                                    draw (scattered to full-N tables), outputs, final state
  tests/golden/env_trace_refact.npz the same with cfg.env.use_ref_actions
  tests/golden/env_trace_generic.npz  the same on a trimesh terrain map with terrain + command curricula and height measurements
  tests/golden/env_trace_yawrate.npz  the same with cfg.commands.heading_command = False
  tests/golden/env_reset_trace.npz  LeggedRobot.reset() (reset_idx(all) + a zero-action step) on an env that has already stepped
  tests/golden/ppo_update.npz       PPO.act / process_env_step / compute_returns / update on a small net
  tests/golden/ppo_update_full.npz  the same at the FULL XBot-L layer widths (inputs regenerated from a seed: ppo_full_case.py)
The oracle (oracle/*.py) is pinned against these in tests/test_oracle_env_golden.py and tests/test_oracle_algo_golden.py
(CPU); the HIP path replays the same vectors on the GPU (tests/test_env_gpu.py, test_net_gpu.py, test_gae_gpu.py).
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

torch.set_num_threads(1)  # deterministic reductions


def npy(t):
    return t.detach().cpu().numpy().copy()


# ------------------------------------------------------------------------------------------------
def gen_constants(R):
    e, cfg = H.make_ref_env(4)
    ppo = R.XBotLCfgPPO()
    d = dict(
        dt=float(e.dt), max_episode_length=float(e.max_episode_length),
        resample_steps=int(cfg.commands.resampling_time / e.dt),
        push_interval=float(cfg.domain_rand.push_interval),
        reward_names=list(e.reward_names),
        reward_scales_dt=[float(e.reward_scales[k]) for k in e.reward_names],
        p_gains=e.p_gains[0].tolist(), d_gains=e.d_gains[0].tolist(), torque_limits=e.torque_limits.tolist(),
        noise_scale_vec=e.noise_scale_vec.tolist(), commands_scale=e.commands_scale.tolist(),
        base_init_state=e.base_init_state.tolist(),
        num_obs=e.num_obs, num_privileged_obs=e.num_privileged_obs,
        gamma=ppo.algorithm.gamma, lam=ppo.algorithm.lam, clip_param=ppo.algorithm.clip_param,
        entropy_coef=ppo.algorithm.entropy_coef, learning_rate=ppo.algorithm.learning_rate,
        value_loss_coef=ppo.algorithm.value_loss_coef, max_grad_norm=ppo.algorithm.max_grad_norm,
        desired_kl=ppo.algorithm.desired_kl, schedule=ppo.algorithm.schedule,
        num_learning_epochs=ppo.algorithm.num_learning_epochs, num_mini_batches=ppo.algorithm.num_mini_batches,
        num_steps_per_env=ppo.runner.num_steps_per_env, seed=ppo.seed,
        actor_hidden_dims=ppo.policy.actor_hidden_dims, critic_hidden_dims=ppo.policy.critic_hidden_dims,
    )
    ac = R.ActorCritic(e.num_obs, e.num_privileged_obs, 12, **R.class_to_dict(ppo.policy))
    d["param_counts"] = dict(actor=sum(p.numel() for p in ac.actor.parameters()),
                             critic=sum(p.numel() for p in ac.critic.parameters()), std=ac.std.numel())
    d["state_dict_keys"] = list(ac.state_dict().keys())
    d["train_cfg_keys"] = sorted(R.class_to_dict(ppo).keys())
    json.dump(d, open(os.path.join(HERE, "constants.json"), "w"), indent=1)
    # full flattened configs (what task_registry hands the runner): the new repo's config classes must equal these
    json.dump(dict(env=R.class_to_dict(R.XBotLCfg()), train=R.class_to_dict(R.XBotLCfgPPO())),
              open(os.path.join(HERE, "config_dump.json"), "w"), indent=1, sort_keys=True)
    print("constants.json", d["param_counts"])


# ------------------------------------------------------------------------------------------------
def run_gae(R, rewards, values, dones, last_values, gamma, lam):
    T, N = rewards.shape
    st = R.RolloutStorage(N, T, [4], [4], [2])
    st.rewards[:] = rewards.unsqueeze(-1)
    st.values[:] = values.unsqueeze(-1)
    st.dones[:] = dones.unsqueeze(-1).byte()
    st.compute_returns(last_values.view(N, 1), gamma, lam)
    return st.returns.squeeze(-1), st.advantages.squeeze(-1)


def gen_gae(R):
    out = {}
    r = torch.tensor([[1, .5], [0, -1], [2, .25], [.5, 1]])
    v = torch.tensor([[.1, .2], [.3, .4], [.5, .6], [.7, .8]])
    d = torch.tensor([[0, 0], [1, 0], [0, 0], [0, 1]])
    lv = torch.tensor([0.9, 1.0])
    ret, adv = run_gae(R, r, v, d, lv, 0.994, 0.9)
    out.update(kat_rewards=npy(r), kat_values=npy(v), kat_dones=npy(d).astype(np.uint8), kat_last=npy(lv),
               kat_returns=npy(ret), kat_adv=npy(adv))
    g = torch.Generator().manual_seed(1234)
    T, N = 60, 24
    r = torch.rand(T, N, generator=g) * 0.3
    v = torch.randn(T, N, generator=g) * 2 + 3
    d = (torch.rand(T, N, generator=g) < 0.03)
    d[-1, 3] = True
    d[0, 5] = True
    lv = torch.randn(N, generator=g) * 2 + 3
    ret, adv = run_gae(R, r, v, d, lv, 0.994, 0.9)
    out.update(rnd_rewards=npy(r), rnd_values=npy(v), rnd_dones=npy(d).astype(np.uint8), rnd_last=npy(lv),
               rnd_returns=npy(ret), rnd_adv=npy(adv), gamma=0.994, lam=0.9)
    np.savez_compressed(os.path.join(HERE, "gae.npz"), **out)
    print("gae.npz kat returns", out["kat_returns"].ravel())


# ------------------------------------------------------------------------------------------------
def gen_policy_example():
    path = os.path.join(H.REFERENCE_ROOT, "logs/XBot_ppo/exported/policies/policy_example.pt")
    sha = hashlib.sha256(open(path, "rb").read()).hexdigest()
    pol = torch.jit.load(path, map_location="cpu")
    sd = pol.state_dict()
    out = {("w_" + k.replace(".", "_")): npy(v) for k, v in sd.items()}
    g = torch.Generator().manual_seed(7)
    x_rand = torch.randn(16, 705, generator=g).clamp(-18, 18)
    with torch.no_grad():
        out["y_zeros"] = npy(pol(torch.zeros(1, 705)))
        out["y_linspace"] = npy(pol(torch.linspace(-1, 1, 705)[None]))
        out["x_rand"] = npy(x_rand)
        out["y_rand"] = npy(pol(x_rand))
    out["sha256"] = np.frombuffer(sha.encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "policy_example.npz"), **out)
    print("policy_example.npz", sha[:8], list(sd.keys()), out["y_zeros"].ravel()[:4])


# ------------------------------------------------------------------------------------------------
def gen_env_trace(R, N=32, S=36, seed=11, use_ref_actions=False, name="env_trace.npz", generic=False, heading_command=True):
    """use_ref_actions=True records the second trace (env_trace_refact.npz): cfg.env.use_ref_actions, humanoid_env.py:190-191
    -- `actions += ref_action` IN PLACE on the caller's tensor, before the clip.
    generic=True records the third (env_trace_generic.npz): the LeggedRobot options XBot-L leaves off -- a trimesh terrain map
    (custom origins with spawn jitter, terrain curriculum, height measurements) and the command curriculum."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    g = torch.Generator().manual_seed(seed)
    fr = 0.1 + 1.9 * torch.rand(N, 1, generator=g)
    bm = 15.0 + 10.0 * torch.rand(N, 1, generator=g) - 5.0
    if generic:
        e, cfg = H.make_ref_env(N, frictions=fr, body_mass=bm, terrain=H.TERRAIN_OPTS, command_curriculum=True)
    else:
        e, cfg = H.make_ref_env(N, frictions=fr, body_mass=bm)
    cfg.env.use_ref_actions = bool(use_ref_actions)
    cfg.commands.heading_command = bool(heading_command)    # False: the yaw rate is sampled directly (legged_robot.py:333-334)
    ids_log = []
    orig_resample = e._resample_commands
    orig_reset_dofs = e._reset_dofs

    def resample(env_ids):
        ids_log.append(("cmd", env_ids.clone()))
        return orig_resample(env_ids)

    def reset_dofs(env_ids):
        ids_log.append(("dof", env_ids.clone()))
        return orig_reset_dofs(env_ids)

    e._resample_commands = resample
    e._reset_dofs = reset_dofs

    def take(log, tag_prefix):
        tag, t = log.pop(0)
        assert tag.startswith(tag_prefix), (tag, tag_prefix)
        return t

    def scatter(ids, vals, width):
        full = torch.zeros(N, width)
        if len(ids):
            full[ids] = vals.view(len(ids), width)
        return full

    if generic:
        levels0, origins0 = e.terrain_levels.clone(), e.env_origins.clone()
    H.RECORDER.enabled = True
    with H.recording_rng():
        H.finish_init(e)
    log = H.RECORDER.pop_all()
    ids = ids_log[:]
    del ids_log[:]
    assert [k for k, _ in ids] == ["dof", "cmd"]
    out = {}
    if generic:                                             # reset_idx order: terrain curriculum, dofs, root states, commands
        t0 = e.terrain
        out.update(terrain_origins=npy(e.terrain_origins), terrain_types=npy(e.terrain_types), terrain_levels0=npy(levels0),
                   height_samples=t0.heightsamples.astype(np.int16), height_points=npy(e.height_points[0]),
                   terrain_env_length=np.array(t0.env_length), terrain_border=np.array(float(cfg.terrain.border_size)),
                   terrain_hscale=np.array(cfg.terrain.horizontal_scale), terrain_vscale=np.array(cfg.terrain.vertical_scale),
                   max_curriculum=np.array(cfg.commands.max_curriculum), env_origins0=npy(origins0))
        out["prime_r_level"] = npy(take(log, "randint_like"))
    prime_u_dof = take(log, "rand_float")
    if generic:
        out["prime_u_xy"] = npy(take(log, "rand_float"))
    prime_u_cmd = torch.cat([take(log, "rand_float") for _ in range(3)], dim=1)
    prime_z_obs = take(log, "randn_like")
    assert not log
    out.update(friction=npy(fr), body_mass=npy(bm), prime_u_dof=npy(prime_u_dof), prime_u_cmd=npy(prime_u_cmd),
               prime_z_obs=npy(prime_z_obs), prime_obs=npy(e.obs_buf), prime_priv=npy(e.privileged_obs_buf),
               prime_commands=npy(e.commands), prime_dof_pos=npy(e.dof_pos), prime_root=npy(e.root_states))

    # plant episode lengths so that time-outs, command resampling and pushes all fire inside S steps
    ep = torch.randint(0, 2300, (N,), generator=g)
    ep[0:4] = torch.tensor([2399, 2398, 2396, 2390])       # time-outs at steps 2,3,5,11
    ep[4:8] = torch.tensor([799, 1598, 795, 2396])         # resamples at steps 1,2,5 ; 2396 -> t/o step 5
    ep[8] = 0
    e.episode_length_buf = ep.clone()                       # rebinding, as on_policy_runner.py:104-106 does
    e.common_step_counter = 388                             # push fires at step 12
    if generic:
        ep[9] = 2389                                        # a time-out at step 12 ...
        e.episode_length_buf = ep.clone()
        e.common_step_counter = 2388                        # ... where the command curriculum is examined (2400 % 2400 == 0)
        k = e.reward_names.index("tracking_lin_vel")
        e.episode_sums["tracking_lin_vel"][:] = 10.0 * torch.rand(N, generator=g) + 200.0   # "tracked well" by a wide margin: the range widens at step 12
        out["init_episode_sums"] = npy(torch.stack([e.episode_sums[n] for n in e.reward_names], dim=1))
    out["init_ep_len"] = npy(ep)
    out["init_common_step_counter"] = int(e.common_step_counter)

    frames = [H.synth_sim_state(g, N) for _ in range(S)]
    counter = {"n": 0, "t": 0}

    def place_on_terrain(frame):
        """Root positions relative to the env's CURRENT origin: far (promotes), near (demotes), in between."""
        root = frame[0]
        r = torch.rand(N, generator=g)
        rad = torch.where(r < 0.35, 4.2 + 2.8 * torch.rand(N, generator=g),
                          torch.where(r < 0.7, 0.3 * torch.rand(N, generator=g), 1.0 + 2.5 * torch.rand(N, generator=g)))
        ang = 6.2831853 * torch.rand(N, generator=g)
        root[:, 0] = e.env_origins[:, 0] + rad * torch.cos(ang)
        root[:, 1] = e.env_origins[:, 1] + rad * torch.sin(ang)
        root[:, 2] += e.env_origins[:, 2]
        extra = torch.rand(N, generator=g) < 0.08            # more falls than the shared frames carry: more curriculum moves
        frame[2].view(N, H.NUM_BODIES, 3)[extra, 0, 2] = 3.0

    def simulate(sim):
        counter["n"] += 1
        if counter["n"] % cfg.control.decimation == 0:
            H.write_sim_state(e, frames[counter["t"]])

    e.gym.simulate = simulate
    keys = ["actions_in", "actions_in_after", "u_delay", "z_act", "u_cmd", "u_dof", "u_push", "z_obs", "root", "dof", "contact", "rigid",
            "frame", "priv_frame", "rew", "reset", "time_out", "commands", "ep_len", "episode_sums", "torques",
            "actions", "any_reset", "pushed", "extras_time_outs", "extras_episode", "root_after", "dof_after"]
    if generic:
        keys += ["u_xy", "r_level", "terrain_levels", "env_origins", "measured_heights", "cmd_range_x"]
    rec = {k: [] for k in keys}
    full_steps = [0, 5, 14, 24, S - 1]
    for t in range(S):
        counter["t"] = t
        a_in = torch.randn(N, 12, generator=g) * 1.5
        if t % 7 == 3:
            a_in[t % N] *= 40.0                              # exercise the +-18 clip
        a_pass = a_in.clone()                                # the tensor the caller hands over (mutated when use_ref_actions)
        if generic:
            place_on_terrain(frames[t])
        with H.recording_rng():
            obs, priv, rew, reset, extras = e.step(a_pass)
        log = H.RECORDER.pop_all()
        ids = ids_log[:]
        del ids_log[:]
        u_delay = take(log, "rand(").view(N)
        z_act = take(log, "randn_like")
        kind, cb_ids = ids.pop(0)
        assert kind == "cmd"
        u_cmd = torch.zeros(N, 6)
        u_cmd[:, 0:3] = scatter(cb_ids, torch.cat([take(log, "rand_float") for _ in range(3)], dim=1), 3)
        pushed = (e.common_step_counter % cfg.domain_rand.push_interval == 0)
        u_push = torch.zeros(N, 5)
        if pushed:
            u_push[:, 0:2] = take(log, "rand_float")
            u_push[:, 2:5] = take(log, "rand_float")
        u_dof = torch.zeros(N, 12)
        u_xy, r_level = torch.zeros(N, 2), torch.zeros(N, dtype=torch.long)
        any_reset = bool(reset.any())
        if any_reset:
            kind, r_ids = ids.pop(0)
            assert kind == "dof"
            if generic:
                r_level[r_ids] = take(log, "randint_like")
            u_dof = scatter(r_ids, take(log, "rand_float"), 12)
            if generic:
                u_xy = scatter(r_ids, take(log, "rand_float"), 2)
            kind, r_ids2 = ids.pop(0)
            assert kind == "cmd" and torch.equal(r_ids, r_ids2)
            u_cmd[:, 3:6] = scatter(r_ids, torch.cat([take(log, "rand_float") for _ in range(3)], dim=1), 3)
        z_obs = take(log, "randn_like")
        assert not log and not ids, (log, ids)
        root, dof, contact, rigid = frames[t]
        vals = dict(actions_in=a_in, actions_in_after=a_pass, u_delay=u_delay, z_act=z_act, u_cmd=u_cmd, u_dof=u_dof, u_push=u_push, z_obs=z_obs,
                    root=root, dof=dof, contact=contact, rigid=rigid,
                    frame=e.obs_history[-1], priv_frame=e.critic_history[-1], rew=rew, reset=reset,
                    time_out=e.time_out_buf, commands=e.commands, ep_len=e.episode_length_buf,
                    episode_sums=torch.stack([e.episode_sums[k] for k in e.reward_names], dim=1),
                    torques=e.torques, actions=e.actions, any_reset=torch.tensor(any_reset),
                    pushed=torch.tensor(bool(pushed)), extras_time_outs=extras["time_outs"],
                    extras_episode=torch.stack([extras["episode"]["rew_" + k] for k in e.reward_names]),
                    root_after=e.root_states, dof_after=e.dof_state)
        if generic:
            vals.update(u_xy=u_xy, r_level=r_level, terrain_levels=e.terrain_levels, env_origins=e.env_origins,
                        measured_heights=e.measured_heights,
                        cmd_range_x=torch.tensor([float(v) for v in e.command_ranges["lin_vel_x"]], dtype=torch.float64))
        for k in keys:
            rec[k].append(npy(vals[k]))
        if t in full_steps:
            out["obs_step%d" % t] = npy(obs)
            out["priv_step%d" % t] = npy(priv)
    for k in keys:
        out[k] = np.stack(rec[k])
    out["full_steps"] = np.array(full_steps)
    out.update(final_feet_air_time=npy(e.feet_air_time), final_last_contacts=npy(e.last_contacts),
               final_feet_height=npy(e.feet_height), final_last_feet_z=npy(e.last_feet_z),
               final_last_actions=npy(e.last_actions), final_last_last_actions=npy(e.last_last_actions),
               final_last_dof_vel=npy(e.last_dof_vel), final_last_root_vel=npy(e.last_root_vel),
               final_ref_dof_pos=npy(e.ref_dof_pos), final_push_force=npy(e.rand_push_force),
               final_push_torque=npy(e.rand_push_torque), final_base_lin_vel=npy(e.base_lin_vel),
               final_base_ang_vel=npy(e.base_ang_vel), final_projected_gravity=npy(e.projected_gravity),
               final_base_euler=npy(e.base_euler_xyz))
    out["use_ref_actions"] = np.array(bool(use_ref_actions))
    out["heading_command"] = np.array(bool(heading_command))
    np.savez_compressed(os.path.join(HERE, name), **out)
    if generic:
        lv = np.concatenate([out["terrain_levels0"][None], out["terrain_levels"]])
        print("terrain moves: up %d down %d; command range %s -> %s" % (
            int((np.diff(lv, axis=0) > 0).sum()), int((np.diff(lv, axis=0) < 0).sum()), out["cmd_range_x"][0], out["cmd_range_x"][-1]))
    n_reset = int(out["reset"].sum())
    n_to = int(out["time_out"].sum())
    print("%s N=%d S=%d resets=%d timeouts=%d pushed_steps=%s resample_rows=%d size=%.2f MB" % (
        name, N, S, n_reset, n_to, np.nonzero(out["pushed"])[0].tolist(), int((out["u_cmd"][:, :, 0] != 0).sum()),
        os.path.getsize(os.path.join(HERE, name)) / 1e6))
    H.RECORDER.enabled = False


# ------------------------------------------------------------------------------------------------
def gen_env_reset_trace(R, N=16, S0=3, seed=15, name="env_reset_trace.npz"):
    """LeggedRobot.reset() = BaseTask.reset(): reset_idx(all envs) + ONE zero-action step (legged_robot.py:110-115, base_task.py:140-145),
    called on an env that has already stepped S0 times.  Records the S0 warm-up steps (inputs + draws), the draws
    of reset_idx(arange(N)), the state at the entry of the zero-action step (i.e. right after reset_idx) and everything the step leaves."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    g = torch.Generator().manual_seed(seed)
    fr = 0.1 + 1.9 * torch.rand(N, 1, generator=g)
    bm = 15.0 + 10.0 * torch.rand(N, 1, generator=g) - 5.0
    e, cfg = H.make_ref_env(N, frictions=fr, body_mass=bm)
    ids_log = []
    orig_resample, orig_reset_dofs = e._resample_commands, e._reset_dofs

    def resample(env_ids):
        ids_log.append(("cmd", env_ids.clone()))
        return orig_resample(env_ids)

    def reset_dofs(env_ids):
        ids_log.append(("dof", env_ids.clone()))
        return orig_reset_dofs(env_ids)

    e._resample_commands, e._reset_dofs = resample, reset_dofs

    def take(log, prefix):
        tag, t = log.pop(0)
        assert tag.startswith(prefix), (tag, prefix)
        return t

    def scatter(ids, vals, width):
        full = torch.zeros(N, width)
        if len(ids):
            full[ids] = vals.view(len(ids), width)
        return full

    H.RECORDER.enabled = True
    with H.recording_rng():
        H.finish_init(e)
    log = H.RECORDER.pop_all()
    del ids_log[:]
    out = dict(friction=npy(fr), body_mass=npy(bm), prime_u_dof=npy(take(log, "rand_float")))
    out["prime_u_cmd"] = npy(torch.cat([take(log, "rand_float") for _ in range(3)], dim=1))
    out["prime_z_obs"] = npy(take(log, "randn_like"))
    assert not log
    ep = torch.randint(5, 2000, (N,), generator=g)
    ep[0:3] = torch.tensor([2399, 798, 2397])               # a time-out and a command resample inside the warm-up
    e.episode_length_buf = ep.clone()
    e.common_step_counter = 40
    out["init_ep_len"] = npy(ep)
    out["init_common_step_counter"] = 40
    frames = [H.synth_sim_state(g, N) for _ in range(S0 + 1)]
    frames[S0][2].view(N, H.NUM_BODIES, 3)[[2, 7], 0, 2] = 3.0      # two base-link contacts: the zero-action step resets those envs AGAIN
    counter = {"n": 0, "t": 0}

    def simulate(sim):
        counter["n"] += 1
        if counter["n"] % cfg.control.decimation == 0:
            H.write_sim_state(e, frames[counter["t"]])

    e.gym.simulate = simulate

    def split_step_draws(log, ids, reset):
        """The draws of ONE XBotLFreeEnv.step, scattered to env-indexed tables (the layout of gen_env_trace)."""
        u_delay = take(log, "rand(").view(N)
        z_act = take(log, "randn_like")
        kind, cb_ids = ids.pop(0)
        assert kind == "cmd"
        u_cmd = torch.zeros(N, 6)
        u_cmd[:, 0:3] = scatter(cb_ids, torch.cat([take(log, "rand_float") for _ in range(3)], dim=1), 3)
        pushed = (e.common_step_counter % cfg.domain_rand.push_interval == 0)
        u_push = torch.zeros(N, 5)
        if pushed:
            u_push[:, 0:2] = take(log, "rand_float")
            u_push[:, 2:5] = take(log, "rand_float")
        u_dof = torch.zeros(N, 12)
        if bool(reset.any()):
            kind, r_ids = ids.pop(0)
            assert kind == "dof"
            u_dof = scatter(r_ids, take(log, "rand_float"), 12)
            kind, r_ids2 = ids.pop(0)
            assert kind == "cmd" and torch.equal(r_ids, r_ids2)
            u_cmd[:, 3:6] = scatter(r_ids, torch.cat([take(log, "rand_float") for _ in range(3)], dim=1), 3)
        z_obs = take(log, "randn_like")
        assert not log and not ids, (log, ids)
        return dict(u_delay=u_delay, z_act=z_act, u_cmd=u_cmd, u_dof=u_dof, u_push=u_push, z_obs=z_obs, pushed=torch.tensor(bool(pushed)))

    def outputs(obs, priv, rew, reset, extras):
        return dict(obs=obs, priv=priv, rew=rew, reset=reset, time_out=e.time_out_buf, commands=e.commands, ep_len=e.episode_length_buf,
                    episode_sums=torch.stack([e.episode_sums[k] for k in e.reward_names], dim=1), torques=e.torques, actions=e.actions,
                    extras_time_outs=extras["time_outs"],
                    extras_episode=torch.stack([extras["episode"]["rew_" + k] for k in e.reward_names]),
                    root_after=e.root_states, dof_after=e.dof_state)

    warm = {}
    for t in range(S0):
        counter["t"] = t
        a_in = torch.randn(N, 12, generator=g) * 1.5
        with H.recording_rng():
            res = e.step(a_in.clone())
        log, ids = H.RECORDER.pop_all(), ids_log[:]
        del ids_log[:]
        vals = dict(actions_in=a_in, root=frames[t][0], dof=frames[t][1], contact=frames[t][2], rigid=frames[t][3])
        vals.update(split_step_draws(log, ids, res[3]))
        vals.update(outputs(*res))
        for k, v in vals.items():
            warm.setdefault(k, []).append(npy(v))
    for k, v in warm.items():
        out["warm_" + k] = np.stack(v)

    # ---- reset(): snapshot the env at the entry of its zero-action step, i.e. right after reset_idx(all)
    counter["t"] = S0
    orig_step = e.step
    entry = {}

    def step_spy(actions):
        assert float(actions.abs().max()) == 0.0 and tuple(actions.shape) == (N, 12)
        entry.update(commands=npy(e.commands), ep_len=npy(e.episode_length_buf), root=npy(e.root_states), dof=npy(e.dof_state),
                     episode_sums=npy(torch.stack([e.episode_sums[k] for k in e.reward_names], dim=1)),
                     last_actions=npy(e.last_actions), last_dof_vel=npy(e.last_dof_vel), feet_air_time=npy(e.feet_air_time),
                     obs_history_absmax=np.array(max(float(f.abs().max()) for f in e.obs_history)),
                     critic_history_absmax=np.array(max(float(f.abs().max()) for f in e.critic_history)),
                     extras_episode=npy(torch.stack([e.extras["episode"]["rew_" + k] for k in e.reward_names])))
        return orig_step(actions)

    e.step = step_spy
    with H.recording_rng():
        obs, priv = e.reset()
    e.step = orig_step
    log, ids = H.RECORDER.pop_all(), ids_log[:]
    del ids_log[:]
    kind, r_ids = ids.pop(0)
    assert kind == "dof" and torch.equal(r_ids, torch.arange(N))
    out["reset_u_dof"] = npy(take(log, "rand_float"))
    kind, r_ids = ids.pop(0)
    assert kind == "cmd" and torch.equal(r_ids, torch.arange(N))
    out["reset_u_cmd"] = npy(torch.cat([take(log, "rand_float") for _ in range(3)], dim=1))
    for k, v in entry.items():
        out["entry_" + k] = v
    vals = dict(root=frames[S0][0], dof=frames[S0][1], contact=frames[S0][2], rigid=frames[S0][3])
    vals.update(split_step_draws(log, ids, e.reset_buf))
    vals.update(outputs(obs, priv, e.rew_buf, e.reset_buf, e.extras))
    for k, v in vals.items():
        out["step_" + k] = npy(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("%s N=%d warm-up steps=%d (resets %d) | resets in the zero-action step=%d size=%.2f MB" % (
        name, N, S0, int(out["warm_reset"].sum()), int(out["step_reset"].sum()), os.path.getsize(os.path.join(HERE, name)) / 1e6))
    H.RECORDER.enabled = False


# ------------------------------------------------------------------------------------------------
def gen_ppo_update(R, N=24, T=8, seed=3):
    """Drive the reference PPO through one full iteration on a small actor/critic (same 705/219/12 interface)."""
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    ah, ch = [48, 32, 16], [40, 32, 16]
    ac = R.ActorCritic(705, 219, 12, actor_hidden_dims=ah, critic_hidden_dims=ch, init_noise_std=1.0)
    alg = R.PPO(ac, num_learning_epochs=2, num_mini_batches=4, clip_param=0.2, gamma=0.994, lam=0.9,
                value_loss_coef=1.0, entropy_coef=0.001, learning_rate=1e-3, max_grad_norm=1.0,
                use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01, device="cpu")
    alg.init_storage(N, T, [705], [219], [12])
    out = {("p0_" + k.replace(".", "_")): npy(v) for k, v in ac.state_dict().items()}
    out["actor_hidden"] = np.array(ah)
    out["critic_hidden"] = np.array(ch)
    obs_l, priv_l, z_l, rew_l, done_l, to_l = [], [], [], [], [], []
    act_l, val_l, logp_l, mu_l, sig_l = [], [], [], [], []
    _orig_normal = torch.normal
    for t in range(T):
        obs = (torch.randn(N, 705, generator=g) * 1.2).clamp(-18, 18)
        priv = (torch.randn(N, 219, generator=g) * 1.2).clamp(-18, 18)
        z = torch.randn(N, 12, generator=g)
        # Normal.sample() = torch.normal(loc.expand, scale.expand); substitute the recorded standard draw
        torch.normal = lambda mean, std, **k: mean + std * z
        try:
            with torch.inference_mode():
                a = alg.act(obs, priv)
        finally:
            torch.normal = _orig_normal
        act_l.append(npy(a)); val_l.append(npy(alg.transition.values)); logp_l.append(npy(alg.transition.actions_log_prob))
        mu_l.append(npy(alg.transition.action_mean)); sig_l.append(npy(alg.transition.action_sigma))
        rew = torch.rand(N, generator=g) * 0.2
        done = torch.rand(N, generator=g) < 0.15
        tout = done & (torch.rand(N, generator=g) < 0.5)
        with torch.inference_mode():
            alg.process_env_step(rew, done, {"time_outs": tout})
        obs_l.append(npy(obs)); priv_l.append(npy(priv)); z_l.append(npy(z)); rew_l.append(npy(rew))
        done_l.append(npy(done)); to_l.append(npy(tout))
    last_priv = (torch.randn(N, 219, generator=g) * 1.2).clamp(-18, 18)
    with torch.inference_mode():
        alg.compute_returns(last_priv)
    st = alg.storage
    out.update(obs=np.stack(obs_l), priv=np.stack(priv_l), z=np.stack(z_l), rew_in=np.stack(rew_l), done=np.stack(done_l),
               time_outs=np.stack(to_l), actions=np.stack(act_l), values=np.stack(val_l), logp=np.stack(logp_l),
               mu=np.stack(mu_l), sigma=np.stack(sig_l), last_priv=npy(last_priv),
               st_rewards=npy(st.rewards), st_returns=npy(st.returns), st_advantages=npy(st.advantages))
    perm_holder = {}
    _orig_randperm = torch.randperm

    def randperm(n, **k):
        p = _orig_randperm(n, generator=g)
        perm_holder["p"] = p.clone()
        return p

    lrs, losses = [], []
    _orig_step = alg.optimizer.step

    def step(*a, **k):
        lrs.append(alg.optimizer.param_groups[0]["lr"])
        if len(lrs) == 1:   # gradients of the first minibatch, after clipping
            for kname, p in ac.named_parameters():
                out["g0_" + kname.replace(".", "_")] = npy(p.grad)
        r = _orig_step(*a, **k)
        if len(lrs) == 1:
            for kname, p in ac.named_parameters():
                out["p1_" + kname.replace(".", "_")] = npy(p.data)
        return r

    alg.optimizer.step = step
    torch.randperm = randperm
    try:
        mvl, msl = alg.update()
    finally:
        torch.randperm = _orig_randperm
    out["perm"] = npy(perm_holder["p"])
    out["lrs"] = np.array(lrs)
    out["mean_value_loss"] = np.array(mvl)
    out["mean_surrogate_loss"] = np.array(msl)
    out["final_lr"] = np.array(alg.learning_rate)
    for k, v in ac.state_dict().items():
        out["pF_" + k.replace(".", "_")] = npy(v)
    np.savez_compressed(os.path.join(HERE, "ppo_update.npz"), **out)
    print("ppo_update.npz lrs", lrs, "losses", mvl, msl)


def gen_ppo_update_full(R):
    """The reference's PPO (algo/ppo/ppo.py:91-184, unmodified) on the FULL XBot-L layer widths: one iteration on the inputs of
    tests/golden/ppo_full_case.py.  Stored: everything small in fp32 (per-step act outputs, rewards / returns / advantages, the
    learning-rate sequence, losses, per-tensor norms of the first minibatch's clipped gradient and of the parameter change);
    the 926 105-element tensors as fp16 (x a power-of-two scale; for rel-L2 / cosine checks) plus an fp32-exact sample of 4096
    entries per large tensor (for the element-wise fp32 checks)."""
    import ppo_full_case as CASE
    seed = CASE.SEED
    torch.manual_seed(seed)
    ac = R.ActorCritic(705, 219, 12, actor_hidden_dims=CASE.ACTOR_HIDDEN, critic_hidden_dims=CASE.CRITIC_HIDDEN, init_noise_std=1.0)
    p0 = CASE.initial_parameters(seed)
    ac.load_state_dict({k: torch.from_numpy(v) for k, v in p0.items()})
    alg = R.PPO(ac, device="cpu", **CASE.HYPER)
    N, T = CASE.N, CASE.T
    alg.init_storage(N, T, [705], [219], [12])
    I = {k: torch.from_numpy(v) for k, v in CASE.rollout_inputs(seed).items()}
    out = dict(seed=np.array(seed), N=np.array(N), T=np.array(T))
    act_l, val_l, logp_l, mu_l = [], [], [], []
    _orig_normal = torch.normal
    for t in range(T):
        z = I["z"][t]
        torch.normal = lambda mean, std, **k: mean + std * z          # Normal.sample() with the recorded standard draw
        try:
            with torch.inference_mode():
                a = alg.act(I["obs"][t], I["priv"][t])
        finally:
            torch.normal = _orig_normal
        act_l.append(npy(a)); val_l.append(npy(alg.transition.values)); logp_l.append(npy(alg.transition.actions_log_prob))
        mu_l.append(npy(alg.transition.action_mean))
        with torch.inference_mode():
            alg.process_env_step(I["rew_in"][t].clone(), I["done"][t], {"time_outs": I["time_outs"][t]})
    with torch.inference_mode():
        alg.compute_returns(I["last_priv"])
    st = alg.storage
    out.update(actions=np.stack(act_l), values=np.stack(val_l), logp=np.stack(logp_l), mu=np.stack(mu_l),
               st_rewards=npy(st.rewards), st_returns=npy(st.returns), st_advantages=npy(st.advantages))
    _orig_randperm = torch.randperm
    torch.randperm = lambda n, **k: I["perm"].clone()
    lrs = []
    _orig_step = alg.optimizer.step
    big = {}

    def keep(prefix, name, t, scale):
        a = npy(t).reshape(-1).astype(np.float32)
        key = name.replace(".", "_")
        out["%s_norm_%s" % (prefix, key)] = np.array(np.sqrt((a.astype(np.float64) ** 2).sum()))
        idx = CASE.sample_index(name, a.size, seed)
        out["%s_s32_%s" % (prefix, key)] = a[idx]                    # fp32-exact sample (whole tensor when small)
        if a.size > idx.size:
            out["%s_h16_%s" % (prefix, key)] = (a * scale).astype(np.float16)
            big[prefix] = scale

    def step(*a, **k):
        lrs.append(alg.optimizer.param_groups[0]["lr"])
        if len(lrs) == 1:   # gradients of the first minibatch, after clipping
            for kname, p in ac.named_parameters():
                keep("g0", kname, p.grad, 1024.0)
        return _orig_step(*a, **k)

    alg.optimizer.step = step
    try:
        mvl, msl = alg.update()
    finally:
        torch.randperm = _orig_randperm
    for kname, p in ac.named_parameters():
        keep("dP", kname, p.data - torch.from_numpy(p0[kname]), 64.0)           # parameter change over the 8 Adam steps
    out.update(lrs=np.array(lrs), mean_value_loss=np.array(mvl), mean_surrogate_loss=np.array(msl), final_lr=np.array(alg.learning_rate),
               g0_h16_scale=np.array(1024.0), dP_h16_scale=np.array(64.0))
    np.savez_compressed(os.path.join(HERE, "ppo_update_full.npz"), **out)
    print("ppo_update_full.npz lrs", lrs, "losses", mvl, msl, "size %.2f MB" % (os.path.getsize(os.path.join(HERE, "ppo_update_full.npz")) / 1e6))


if __name__ == "__main__":
    R = H.load_reference()
    if "--only-ppo-full" in sys.argv:
        gen_ppo_update_full(R)
        sys.exit(0)
    if "--only-generic" in sys.argv:
        gen_env_trace(R, N=32, S=20, seed=13, name="env_trace_generic.npz", generic=True)
        sys.exit(0)
    if "--only-reset" in sys.argv:
        gen_env_reset_trace(R)
        sys.exit(0)
    if "--only-yawrate" in sys.argv:
        gen_env_trace(R, N=16, S=14, seed=14, name="env_trace_yawrate.npz", heading_command=False)
        sys.exit(0)
    gen_constants(R)
    if "--constants-only" not in sys.argv:
        gen_gae(R)
        gen_policy_example()
        gen_env_trace(R)
        gen_env_trace(R, N=16, S=16, seed=12, use_ref_actions=True, name="env_trace_refact.npz")
        gen_env_trace(R, N=32, S=20, seed=13, name="env_trace_generic.npz", generic=True)
        gen_env_trace(R, N=16, S=14, seed=14, name="env_trace_yawrate.npz", heading_command=False)
        gen_env_reset_trace(R)
        gen_ppo_update(R)
        gen_ppo_update_full(R)
