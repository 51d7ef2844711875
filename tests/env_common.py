"""Shared driver for the env parity tests: runs the same seeded trace through a backend (the HIP library on a
GPU, or the host emulation of the same kernel source) and through the oracle, and compares everything.

Bars (BASELINE.json north_star / SURVEY.md §8c): termination / time-out / indexing masks and integer state
bit-exact; rewards, observations and float state within 1e-5 relative (+ a small absolute floor for values
that pass through the reference's add-2pi-subtract-2pi angle wrap, which quantises to ~5e-7).
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "humanoid-gym_amd"))

from oracle import xbot_constants as K  # noqa: E402
from oracle.xbot_env_oracle import XBotEnvOracle  # noqa: E402

RTOL, ATOL = 1e-5, 2e-6


def synth_frames(gen, N):
    """One Isaac-Gym-shaped frame of seeded synthetic sim state (same recipe as tests/golden/ref_harness.py)."""
    root = torch.zeros(N, 13)
    root[:, 0:2] = torch.randn(N, 2, generator=gen) * 0.5
    root[:, 2] = 0.9 + 0.02 * (2 * torch.rand(N, generator=gen) - 1)
    q = torch.cat([torch.randn(N, 3, generator=gen) * 0.15, torch.ones(N, 1)], dim=1)
    root[:, 3:7] = q / q.norm(dim=1, keepdim=True)
    root[:, 7:13] = torch.randn(N, 6, generator=gen) * 0.3
    dof = torch.zeros(N, 12, 2)
    dof[:, :, 0] = torch.randn(N, 12, generator=gen) * 0.2
    dof[:, :, 1] = torch.randn(N, 12, generator=gen) * 1.5
    contact = torch.zeros(N, K.NUM_BODIES, 3)
    feet, knees = list(K.FEET_BODIES), list(K.KNEE_BODIES)
    u = torch.rand(N, 2, generator=gen)
    on = (torch.rand(N, 2, generator=gen) > 0.4).float()
    contact[:, feet, 2] = 900.0 * u * on
    contact[:, feet, 0:2] = torch.randn(N, 2, 2, generator=gen) * 40.0 * on.unsqueeze(-1)
    hit = (torch.rand(N, generator=gen) < 0.04).float()
    contact[:, 0, :] = torch.randn(N, 3, generator=gen) * 2.0 * hit.unsqueeze(-1)
    small = (torch.rand(N, generator=gen) < 0.05).float() * (1 - hit)
    contact[:, 0, :] += torch.randn(N, 3, generator=gen) * 0.2 * small.unsqueeze(-1)
    rigid = torch.randn(N, K.NUM_BODIES, 13, generator=gen) * 0.2
    rigid[:, feet, 2] = 0.03 + 0.09 * torch.rand(N, 2, generator=gen)
    rigid[:, feet[0], 1] += 0.15
    rigid[:, feet[1], 1] -= 0.15
    rigid[:, knees[0], 1] += 0.12
    rigid[:, knees[1], 1] -= 0.12
    return root, dof.reshape(N * 12, 2), contact.reshape(N * K.NUM_BODIES, 3), rigid.reshape(N * K.NUM_BODIES, 13)


class HipBackend:
    """Calls libhgym_hip.so on the current CUDA(HIP) stream."""

    name = "hip"
    device = "cuda"

    def __init__(self):
        from hgym import _lib as L
        self.L = L

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def step_call(self, mode, cfg, sim, st, out, noise):
        L = self.L
        fn = {"prime": L.lib.hgym_env_prime, "reset_all": L.lib.hgym_env_reset_all, "post": L.lib.hgym_post_physics}[mode]
        L.check(fn(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), C.byref(noise), self.stream()), mode)

    def pre_physics(self, cfg, st, actions, noise):
        self.L.check(self.L.lib.hgym_pre_physics(C.byref(cfg), C.byref(st), self.L.fptr(actions), C.byref(noise), self.stream()))

    def pd_torques(self, cfg, sim, st):
        self.L.check(self.L.lib.hgym_pd_torques(C.byref(cfg), C.byref(sim), C.byref(st), self.stream()))

    def sync(self):
        torch.cuda.synchronize()


class HostBackend:
    """Drives tests/hostcheck/libhgym_hostcheck.so: the product's kernel source compiled for the host."""

    name = "host"
    device = "cpu"

    def __init__(self, envs_per_block=8, nthreads=64, split=False):
        """split: the phase sequence of the XBot-L fast kernels (per-joint work on (env, joint) lanes around a shorter per-env
        chain: env_step_phase_j / _a<split> / _f) instead of the monolithic per-env chain; default options only.  2 / 3: that chain on
        four wavefronts by role (env_step_phase_a3 + env_step_reward_sum, what the fused rollout launch runs), its roles emulated in ascending /
        descending lane order -- a role reading what another one writes during the phase would make the two differ."""
        sys.path.insert(0, os.path.join(ROOT, "tests", "hostcheck"))
        import build_hostcheck
        self.lib = C.CDLL(build_hostcheck.build())
        self.epb, self.nthreads, self.split = envs_per_block, nthreads, int(split)

    def step_call(self, mode, cfg, sim, st, out, noise):
        m = {"post": 0, "prime": 1, "reset_all": 2}[mode]
        self.lib.hc_env_step_ex(C.byref(cfg), C.byref(sim), C.byref(st), C.byref(out), C.byref(noise), None, m, 0,
                                self.epb, self.nthreads, self.split)

    def pre_physics(self, cfg, st, actions, noise):
        self.lib.hc_pre_physics(C.byref(cfg), C.byref(st), C.cast(actions.data_ptr(), C.POINTER(C.c_float)), C.byref(noise))

    def pd_torques(self, cfg, sim, st):
        self.lib.hc_pd_torques(C.byref(cfg), C.byref(sim), C.byref(st))

    def sync(self):
        pass


def close(a, b, what, rtol=RTOL, atol=ATOL):
    a = a.detach().cpu().float().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().float().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)


def exact(a, b, what):
    a = a.detach().cpu() if torch.is_tensor(a) else torch.as_tensor(a)
    b = b.detach().cpu() if torch.is_tensor(b) else torch.as_tensor(b)
    assert torch.equal(a.to(torch.int64), b.to(torch.int64)), what


class EnvUnderTest:
    """Product-side buffers + the call sequence of one vec-step in parity mode (external noise, external sim frames)."""

    def __init__(self, backend, N, friction, body_mass, sim_layout="soa", frame_stack=15, c_frame_stack=3, use_ref_actions=False,
                 terrain=None, command_curriculum=None, heading_command=True, rows_ahead=False):
        """terrain: an oracle TerrainSpec (its initial levels are copied); command_curriculum: max_curriculum or None.
        rows_ahead: the observation rows rotate through three buffers the way hgym_rollout_step's caller hands them in -- the step
        writes its observations to one, the older frames of the NEXT step's rows to another (HgymEnvOut.obs_ahead / priv_ahead),
        and from the second step on finds its own rows' older frames already written (obs_older_ready).  A buffer is filled with
        NaN before it is handed in as `ahead`, so a frame nobody wrote would show in the comparison with the oracle."""
        self.rows_ahead = bool(rows_ahead)
        from hgym import EnvBuffers, default_env_config
        self.be = backend
        self.cfg = default_env_config(N, frame_stack=frame_stack, c_frame_stack=c_frame_stack)
        self.cfg.use_ref_actions = int(bool(use_ref_actions))
        self.cfg.heading_command = int(bool(heading_command))
        self.buf = EnvBuffers(self.cfg, backend.device, sim_layout=sim_layout)
        self.buf.f["friction"].copy_(friction.view(1, N))
        self.buf.f["body_mass"].copy_(body_mass.view(1, N))
        if terrain is not None:
            self.buf.set_terrain(terrain.origins, terrain.levels, terrain.types, terrain.env_length, terrain.curriculum,
                                 height_samples=terrain.height_samples, height_points=terrain.height_points,
                                 border_size=terrain.border_size, horizontal_scale=terrain.hscale, vertical_scale=terrain.vscale)
        if command_curriculum is not None:
            self.buf.set_command_curriculum(K.CMD_LIN_VEL_X, command_curriculum)
        self.sim, self.st, self.out = self.buf.sim_struct(), self.buf.state_struct(), self.buf.out_struct()
        self.dev = backend.device

    def _noise(self, **kw):
        conv = lambda k, v: None if v is None else (v.to(self.dev).long().contiguous() if k == "r_level" else v.to(self.dev).float().contiguous())
        self._keep = {k: conv(k, v) for k, v in kw.items()}
        return self.buf.noise_struct(**self._keep)

    def prime(self, u_dof, u_cmd3, z_obs, u_xy=None, r_level=None):
        N = self.buf.N
        u_cmd = torch.zeros(N, 6)
        u_cmd[:, 3:6] = u_cmd3
        self.be.step_call("prime", self.cfg, self.sim, self.st, self.out,
                          self._noise(u_dof=u_dof, u_cmd=u_cmd, z_obs=z_obs, u_xy=u_xy, r_level=r_level))

    def reset_all(self, u_dof, u_cmd3, u_xy=None, r_level=None):
        N = self.buf.N
        u_cmd = torch.zeros(N, 6)
        u_cmd[:, 3:6] = u_cmd3
        self.be.step_call("reset_all", self.cfg, self.sim, self.st, self.out, self._noise(u_dof=u_dof, u_cmd=u_cmd, u_xy=u_xy, r_level=r_level))

    def step(self, actions_in, frame, u_delay, z_act, u_cmd, u_dof, u_push, z_obs, u_xy=None, r_level=None):
        a = actions_in.to(self.dev).float().contiguous().clone()
        self.be.pre_physics(self.cfg, self.st, a, self._noise(u_delay=u_delay, z_act=z_act))
        self.actions_after = a                  # the caller's tensor after the call (mutated only with use_ref_actions)
        self.be.pd_torques(self.cfg, self.sim, self.st)
        self.be.sync()
        self.buf.load_sim(*frame)
        if self.rows_ahead:
            fp = lambda t: C.cast(t.data_ptr(), C.POINTER(C.c_float))
            if not hasattr(self, "_rows"):
                b = self.buf
                self._rows = [(b.obs, b.priv_obs)] + [(torch.empty_like(b.obs), torch.empty_like(b.priv_obs)) for _ in range(2)]
                self._k, self._ready = 0, False
            nxt, ah = self._rows[(self._k + 1) % 3], self._rows[(self._k + 2) % 3]
            if not self._ready:
                nxt[0].fill_(float("nan")); nxt[1].fill_(float("nan"))
            ah[0].fill_(float("nan")); ah[1].fill_(float("nan"))
            self.out.obs, self.out.priv_obs = fp(nxt[0]), fp(nxt[1])
            self.out.obs_ahead, self.out.priv_ahead = fp(ah[0]), fp(ah[1])
            self.out.obs_older_ready = int(self._ready)
        self.be.step_call("post", self.cfg, self.sim, self.st, self.out,
                          self._noise(u_cmd=u_cmd, u_dof=u_dof, u_push=u_push, z_obs=z_obs, u_xy=u_xy, r_level=r_level))
        self.be.sync()
        if self.rows_ahead:
            self._k, self._ready = self._k + 1, True
            self.buf.obs, self.buf.priv_obs = nxt


def compare_state(env, o, tag, check_obs=True):
    """Every observable of the product env against the oracle object `o` (after the same step)."""
    b = env.buf
    exact(b.reset, o.reset, tag + " reset mask")
    exact(b.time_out, o.time_out, tag + " time_out mask")
    exact(b.episode_length, o.ep_len, tag + " episode_length")
    if o.extras_time_outs is not None:
        exact(b.extras_time_outs, o.extras_time_outs, tag + " extras time_outs (stale-by-design)")
    for name, ref in [("commands", o.commands), ("actions", o.actions), ("last_actions", o.last_actions),
                      ("last_last_actions", o.last_last_actions), ("last_dof_vel", o.last_dof_vel),
                      ("last_root_vel", o.last_root_vel), ("torques", o.torques), ("feet_air_time", o.feet_air_time),
                      ("feet_height", o.feet_height), ("last_feet_z", o.last_feet_z), ("ref_dof_pos", o.ref_dof_pos),
                      ("push_force", o.push_force), ("push_torque", o.push_torque), ("episode_sums", o.episode_sums),
                      ("base_lin_vel", o.base_lin_vel), ("base_ang_vel", o.base_ang_vel),
                      ("projected_gravity", o.projected_gravity), ("base_euler", o.base_euler)]:
        # torques = Kp*(0.25 a + q0 - q) - Kd*qd: terms of magnitude ~1e2 that can cancel to ~1e-1, so 1e-5 relative is taken
        # on the scale of the terms (1 ulp of an action moves the result by ~1e-5 absolute), not of the difference
        close(b.view(name), ref, tag + " " + name, atol=2e-5 if name == "torques" else ATOL)
    exact(b.view("last_contacts") > 0.5, o.last_contacts, tag + " last_contacts")
    close(b.rew, o.rew, tag + " rew")
    close(b.root_view(), o.sim.root, tag + " root_states (reset / push write-back)")
    close(b.dof_pos_view(), o.sim.dof_pos, tag + " dof_pos")
    close(b.dof_vel_view(), o.sim.dof_vel, tag + " dof_vel")
    if o.extras_episode is not None:
        close(b.extras_episode, o.extras_episode, tag + " extras episode means", rtol=1e-5, atol=1e-7)
    if check_obs:
        close(b.obs, torch.clip(o.obs, -K.CLIP_OBS, K.CLIP_OBS), tag + " obs")
        close(b.priv_obs, torch.clip(o.priv, -K.CLIP_OBS, K.CLIP_OBS), tag + " priv_obs")
    if o.terrain is not None:
        exact(b.terrain_levels, o.terrain.levels, tag + " terrain_levels")
        exact(b.view("env_origins").contiguous().view(torch.int32), o.env_origins.contiguous().view(torch.int32), tag + " env_origins (bits)")
        if o.measured_heights is not None:
            # indices come from a truncated fp32 quotient: bit-exact arithmetic, so the sampled cells must agree everywhere
            exact(b.measured_heights.contiguous().view(torch.int32), o.measured_heights.contiguous().view(torch.int32), tag + " measured_heights (bits)")
    if o.command_curriculum:
        assert [float(v) for v in b.command_range_x.cpu()] == o.cmd_range_x, tag + " command range"


def random_terrain_spec(g, N, rows=5, cols=4, points=(7, 5)):
    """A seeded terrain map with the shapes humanoid.utils.terrain produces (tile origins, int16 height field, sample grid)."""
    from oracle.xbot_env_oracle import TerrainSpec
    tile, border, hs = 8.0, 3.0, 0.1
    origins = torch.zeros(rows, cols, 3)
    origins[:, :, 0] = (torch.arange(rows).float().view(-1, 1) + 0.5) * tile
    origins[:, :, 1] = (torch.arange(cols).float().view(1, -1) + 0.5) * tile
    origins[:, :, 2] = 0.3 * torch.rand(rows, cols, generator=g)
    field = torch.randint(-60, 90, (int(rows * tile / hs + 2 * border / hs), int(cols * tile / hs + 2 * border / hs)), generator=g).to(torch.int16)
    px = torch.linspace(-0.6, 0.6, points[0])
    py = torch.linspace(-0.4, 0.4, points[1])
    gx, gy = torch.meshgrid(px, py, indexing="ij")
    hp = torch.zeros(points[0] * points[1], 3)
    hp[:, 0], hp[:, 1] = gx.flatten(), gy.flatten()
    levels = torch.randint(0, rows, (N,), generator=g)
    types = torch.div(torch.arange(N), N / cols, rounding_mode="floor").long()
    return TerrainSpec(origins, levels, types, tile, True, height_samples=field, height_points=hp, border_size=border,
                       horizontal_scale=hs, vertical_scale=0.005)


def run_random_trace(backend, N, steps, seed, sim_layout="soa", frame_stack=15, c_frame_stack=3, check_every=1,
                     use_ref_actions=False, generic=False, track_sum=40.0, heading_command=True, rows_ahead=False):
    """Seeded random trace through product + oracle with identical inputs; returns event counts.
    generic: a terrain map (custom origins, terrain curriculum, height measurements) and the command curriculum are on; the
    common step counter is planted so that the command curriculum is examined inside the trace and `track_sum` (the planted
    tracking_lin_vel episode sums) decides whether it moves the range."""
    g = torch.Generator().manual_seed(seed)
    fr = 0.1 + 1.9 * torch.rand(N, 1, generator=g)
    bm = 10.0 + 10.0 * torch.rand(N, 1, generator=g)
    spec = random_terrain_spec(g, N) if generic else None
    o = XBotEnvOracle(N, frictions=fr, body_mass=bm, frame_stack=frame_stack, c_frame_stack=c_frame_stack,
                      use_ref_actions=use_ref_actions, terrain=spec, command_curriculum=generic, max_curriculum=1.5,
                      heading_command=heading_command)
    env = EnvUnderTest(backend, N, fr, bm, sim_layout=sim_layout, frame_stack=frame_stack, c_frame_stack=c_frame_stack,
                       use_ref_actions=use_ref_actions, terrain=spec, command_curriculum=1.5 if generic else None,
                       heading_command=heading_command, rows_ahead=rows_ahead)
    u_dof, u_cmd3, z_obs = torch.rand(N, 12, generator=g), torch.rand(N, 3, generator=g), torch.randn(N, 47, generator=g)
    gen_draws = lambda: (torch.rand(N, 2, generator=g), torch.randint(0, spec.max_level, (N,), generator=g)) if generic else (None, None)
    u_xy, r_level = gen_draws()
    o.prime(u_dof, u_cmd3, z_obs, u_xy, r_level)
    env.prime(u_dof, u_cmd3, z_obs, u_xy, r_level)
    backend.sync()
    compare_state(env, o, "prime")
    ep = torch.randint(0, 2400, (N,), generator=g)
    ep[: min(N, 6)] = torch.tensor([2399, 2398, 799, 1598, 0, 2396])[: min(N, 6)]
    o.ep_len = ep.clone()
    env.buf.episode_length.copy_(ep)
    csc = 397
    if generic:
        csc = 2397                      # step 3 lands on 2400: a push step AND the command-curriculum check (time-out planted below)
        ep[min(N, 6):min(N, 8)] = 2397
        o.ep_len = ep.clone()
        env.buf.episode_length.copy_(ep)
        sums = o.episode_sums.clone()
        sums[:, K.REWARD_NAMES.index("tracking_lin_vel")] = track_sum * (0.9 + 0.2 * torch.rand(N, generator=g))
        o.episode_sums = sums.clone()
        env.buf.view("episode_sums").copy_(sums)
    o.common_step_counter = csc
    env.buf.counters[0] = csc
    counts = dict(reset=0, timeout=0, push=0, stale=0, level_up=0, level_down=0, range_moves=0)
    for t in range(steps):
        a_in = torch.randn(N, 12, generator=g) * 1.5
        if t % 5 == 2:
            a_in[t % N] *= 40.0
        frame = synth_frames(g, N)
        if generic:                     # base positions relative to the env's current tile origin: far, near, in between
            r = torch.rand(N, generator=g)
            rad = torch.where(r < 0.4, 4.2 + 2.5 * torch.rand(N, generator=g), 3.5 * torch.rand(N, generator=g))
            ang = 6.2831853 * torch.rand(N, generator=g)
            frame[0][:, 0] = o.env_origins[:, 0] + rad * torch.cos(ang)
            frame[0][:, 1] = o.env_origins[:, 1] + rad * torch.sin(ang)
            frame[0][:, 2] += o.env_origins[:, 2]
        u_delay, z_act = torch.rand(N, generator=g), torch.randn(N, 12, generator=g)
        u_cmd, u_dof = torch.rand(N, 6, generator=g), torch.rand(N, 12, generator=g)
        u_push, z_obs = torch.rand(N, 5, generator=g), torch.randn(N, 47, generator=g)
        u_xy, r_level = gen_draws()
        a_o = a_in.clone()
        o.pre_physics(a_o, u_delay, z_act)      # mutates a_o when use_ref_actions
        o.pd_torques()
        o.sim.load(*frame)
        before = None if spec is None else (spec.levels.clone(), list(o.cmd_range_x))
        _, _, _, _, info = o.post_physics(u_cmd, u_dof, u_push, z_obs, u_xy, r_level)
        env.step(a_in, frame, u_delay, z_act, u_cmd, u_dof, u_push, z_obs, u_xy, r_level)
        if before is not None:
            counts["level_up"] += int((spec.levels > before[0]).sum())
            counts["level_down"] += int((spec.levels < before[0]).sum())
            counts["range_moves"] += int(before[1] != o.cmd_range_x)
        if t % check_every == 0 or t == steps - 1:
            compare_state(env, o, "step %d" % t)
            close(env.actions_after, a_o, "step %d caller's action tensor" % t)
        counts["reset"] += int(o.reset.sum())
        counts["timeout"] += int(o.time_out.sum())
        counts["push"] += int(info["pushed"])
        counts["stale"] += int((not info["any_reset"]) and o.extras_time_outs is not None and bool(o.extras_time_outs.any()))
    return counts, env, o


# ------------------------------------------------------------------------------------------------ generic options
def terrain_spec_from_golden(G):
    from oracle.xbot_env_oracle import TerrainSpec
    T = lambda a: torch.from_numpy(np.asarray(a))
    return TerrainSpec(T(G["terrain_origins"]), T(G["terrain_levels0"]), T(G["terrain_types"]), float(G["terrain_env_length"]), True,
                       height_samples=T(G["height_samples"]), height_points=T(G["height_points"]),
                       border_size=float(G["terrain_border"]), horizontal_scale=float(G["terrain_hscale"]),
                       vertical_scale=float(G["terrain_vscale"]))


def run_generic_golden(backend, golden_dir):
    """tests/golden/env_trace_generic.npz (recorded from the reference: trimesh terrain map, terrain + command curricula,
    height measurements) through the product: integer state, origins and sampled heights bit-exact, floats to the usual bars."""
    T = lambda a: torch.from_numpy(np.asarray(a))
    G = np.load(os.path.join(golden_dir, "env_trace_generic.npz"))
    N = G["friction"].shape[0]
    env = EnvUnderTest(backend, N, T(G["friction"]), T(G["body_mass"]), sim_layout="aos", terrain=terrain_spec_from_golden(G),
                       command_curriculum=float(G["max_curriculum"]))
    b = env.buf
    env.prime(T(G["prime_u_dof"]), T(G["prime_u_cmd"]), T(G["prime_z_obs"]), T(G["prime_u_xy"]), T(G["prime_r_level"]))
    backend.sync()
    close(b.obs, G["prime_obs"], "prime obs")
    close(b.priv_obs, G["prime_priv"], "prime priv")
    close(b.root, G["prime_root"], "prime root (spawn jitter)")
    exact(b.terrain_levels, G["terrain_levels0"], "prime levels")
    b.episode_length.copy_(T(G["init_ep_len"]))
    b.counters[0] = int(G["init_common_step_counter"])
    b.view("episode_sums").copy_(T(G["init_episode_sums"]))
    moved = 0
    for t in range(G["rew"].shape[0]):
        frame = (T(G["root"][t]), T(G["dof"][t]), T(G["contact"][t]), T(G["rigid"][t]))
        env.step(T(G["actions_in"][t]), frame, T(G["u_delay"][t]), T(G["z_act"][t]), T(G["u_cmd"][t]), T(G["u_dof"][t]),
                 T(G["u_push"][t]), T(G["z_obs"][t]), T(G["u_xy"][t]), T(G["r_level"][t]))
        tag = "generic step %d " % t
        exact(b.reset, G["reset"][t], tag + "reset")
        exact(b.episode_length, G["ep_len"][t], tag + "ep_len")
        exact(b.terrain_levels, G["terrain_levels"][t], tag + "terrain_levels")
        exact(b.view("env_origins").contiguous().view(torch.int32), T(G["env_origins"][t]).view(torch.int32), tag + "env_origins (bits)")
        exact(b.measured_heights.contiguous().view(torch.int32), T(G["measured_heights"][t]).view(torch.int32), tag + "measured_heights (bits)")
        assert [float(v) for v in b.command_range_x.cpu()] == [float(v) for v in G["cmd_range_x"][t]], tag + "command range"
        close(b.view("commands"), G["commands"][t], tag + "commands")
        close(b.root, G["root_after"][t], tag + "root (reset write-back with jitter)")
        close(b.rew, G["rew"][t], tag + "rew")
        close(b.view("episode_sums"), G["episode_sums"][t], tag + "episode_sums")
        H, HC = 15, 3
        close(b.obs[:, (H - 1) * 47:], np.clip(G["frame"][t], -K.CLIP_OBS, K.CLIP_OBS), tag + "newest obs frame")
        close(b.priv_obs[:, (HC - 1) * 73:], np.clip(G["priv_frame"][t], -K.CLIP_OBS, K.CLIP_OBS), tag + "newest priv frame")
        close(b.obs_ring.view(N, H, 47)[:, (int(b.counters[2]) - 1) % H], G["frame"][t], tag + "ring frame")
        moved += int(t > 0 and list(G["cmd_range_x"][t]) != list(G["cmd_range_x"][t - 1]))
    assert moved == 1
    return env


def run_reset_golden(backend, golden_dir):
    """LeggedRobot.reset() (legged_robot.py:110-115, base_task.py:140-145) replayed from env_reset_trace.npz, the
    trace recorded from the reference: warm-up steps, hgym_env_reset_all with the reference's reset_idx(all) draws, the state the
    reference's zero-action step finds at its entry, and everything that step leaves (two envs reset AGAIN in it)."""
    import os
    G = np.load(os.path.join(golden_dir, "env_reset_trace.npz"))
    Tn = lambda a: torch.from_numpy(np.asarray(a))
    N = G["friction"].shape[0]
    env = EnvUnderTest(backend, N, Tn(G["friction"]), Tn(G["body_mass"]), sim_layout="aos")
    env.prime(Tn(G["prime_u_dof"]), Tn(G["prime_u_cmd"]), Tn(G["prime_z_obs"]))
    backend.sync()
    env.buf.episode_length.copy_(Tn(G["init_ep_len"]))
    env.buf.counters[0] = int(G["init_common_step_counter"])

    def step(g, a_in, tag):
        frame = (Tn(g("root")), Tn(g("dof")), Tn(g("contact")), Tn(g("rigid")))
        env.step(a_in, frame, Tn(g("u_delay")), Tn(g("z_act")), Tn(g("u_cmd")), Tn(g("u_dof")), Tn(g("u_push")), Tn(g("z_obs")))
        b = env.buf
        exact(b.reset, g("reset"), tag + " reset")
        exact(b.time_out, g("time_out"), tag + " time_out")
        exact(b.episode_length, g("ep_len"), tag + " ep_len")
        exact(b.extras_time_outs, g("extras_time_outs"), tag + " extras time_outs")
        close(b.rew, g("rew"), tag + " rew")
        close(b.view("torques"), g("torques"), tag + " torques", atol=2e-5)
        close(b.view("actions"), g("actions"), tag + " actions")
        close(b.view("commands"), g("commands"), tag + " commands")
        close(b.view("episode_sums"), g("episode_sums"), tag + " episode_sums")
        close(b.root, g("root_after"), tag + " root")
        close(b.dof_state, g("dof_after"), tag + " dof")
        close(b.extras_episode, g("extras_episode"), tag + " extras episode", rtol=1e-5, atol=1e-7)
        close(b.obs, g("obs"), tag + " obs")
        close(b.priv_obs, g("priv"), tag + " priv")

    for t in range(G["warm_rew"].shape[0]):
        step(lambda k, t=t: G["warm_" + k][t], Tn(G["warm_actions_in"][t]), "warm-up step %d" % t)
    env.reset_all(Tn(G["reset_u_dof"]), Tn(G["reset_u_cmd"]))
    backend.sync()
    b = env.buf
    assert float(b.obs_ring.abs().max()) == 0.0 and float(b.priv_ring.abs().max()) == 0.0      # both history rings zeroed
    close(b.view("commands"), G["entry_commands"], "commands after reset_all")
    exact(b.episode_length, G["entry_ep_len"], "ep_len after reset_all")
    close(b.root, G["entry_root"], "root after reset_all")
    close(b.dof_state, G["entry_dof"], "dof after reset_all")
    assert float(b.view("episode_sums").abs().max()) == 0.0
    close(b.view("last_actions"), G["entry_last_actions"], "last_actions after reset_all")
    close(b.view("last_dof_vel"), G["entry_last_dof_vel"], "last_dof_vel after reset_all")
    close(b.view("feet_air_time"), G["entry_feet_air_time"], "feet_air_time after reset_all")
    step(lambda k: G["step_" + k], torch.zeros(N, 12), "zero-action step of reset()")
    assert int(G["step_reset"].sum()) == 2
