"""Shared driver of the internal-Philox / synthetic-physics parity tests (the configuration bench.py times): the product's fused
env step -- host emulation of the kernel source, `hgym_env_step_synth`, or the env part of `hgym_rollout_step` -- against the
oracle consuming the SAME Philox stream (oracle/synth_env_oracle.py).

Bars: reset / time-out masks, episode lengths and the stale-by-design extras["time_outs"] bit-exact; floats 1e-5 relative
(env_common.RTOL / ATOL).  One caveat, stated and counted: the device's normals come from the hardware log2 / sqrt / sin / cos
(~1e-6 absolute from libm's), and `_reward_low_speed` (humanoid_env.py:469-500) is a step function of the base velocity those
normals feed -- an env within ~1e-6 of one of its thresholds can land on the other side.  Such an env shows up as a reward
difference of one low_speed quantum (<= 3.2 * 0.2 * dt); it is re-synchronised, counted, and the count is asserted small.
"""
import numpy as np
import torch

import env_common as EC
from oracle import synth_env_oracle as S
from oracle import xbot_constants as K
from oracle.xbot_env_oracle import XBotEnvOracle

LOW_SPEED_QUANTUM = 3.2 * 0.2 * K.DT + 1e-5      # largest jump of the low_speed term (-2 <-> 1.2) times its scale times dt


REPORT = []      # (what, forgiven low_speed threshold flips): printed by tests/conftest.py in the terminal summary, pass or fail


def report(what, flips):
    """Record a Philox-vs-oracle run's forgiven-flip count so that it shows in the driver's log even when the test passes
    (pytest -q swallows a passing test's stdout)."""
    REPORT.append((what, int(flips)))
    print("%s; low_speed threshold flips forgiven: %d" % (what, flips))


class Holder:
    def __init__(self, buf):
        self.buf = buf


def oracle_from_buffers(buf):
    """An XBotEnvOracle in exactly the state the product's buffers hold (any point between two steps)."""
    N = buf.N
    c = lambda t: t.detach().cpu().clone()
    o = XBotEnvOracle(N, frictions=c(buf.view("friction")), body_mass=c(buf.view("body_mass")))
    o.sim.root.copy_(c(buf.root_view()))
    o.sim.dof_pos.copy_(c(buf.dof_pos_view()))
    o.sim.dof_vel.copy_(c(buf.dof_vel_view()))
    o.sim.contact.copy_(c(buf.contact_view()))
    o.sim.rigid.copy_(c(buf.rigid_view()))
    o.env_origins = c(buf.view("env_origins")).contiguous()
    for name in ("commands", "actions", "last_actions", "last_last_actions", "last_dof_vel", "last_root_vel", "torques",
                 "feet_air_time", "feet_height", "last_feet_z", "ref_dof_pos", "push_force", "push_torque", "episode_sums",
                 "base_lin_vel", "base_ang_vel", "projected_gravity", "base_euler"):
        setattr(o, name, c(buf.view(name)).contiguous())
    o.last_contacts = c(buf.view("last_contacts")) > 0.5
    o.ep_len = c(buf.episode_length)
    o.common_step_counter = int(buf.counters[0])
    H, HC = o.H, o.Hc
    ring = int(buf.counters[2])
    order = [(ring + k) % H for k in range(H)]                       # oldest -> newest
    o.obs_hist = c(buf.obs_ring).view(N, H, K.NUM_SINGLE_OBS)[:, order].contiguous()
    order_c = [(ring + k) % HC for k in range(HC)]
    o.priv_hist = c(buf.priv_ring).view(N, HC, K.SINGLE_NUM_PRIV_OBS)[:, order_c].contiguous()
    o.extras_time_outs = c(buf.extras_time_outs).bool()
    o.extras_episode = c(buf.extras_episode)
    o.reset = c(buf.reset).bool()
    o.time_out = c(buf.time_out).bool()
    return o


def forgive_low_speed(rew_dev, sums_dev, o, budget):
    """Envs whose reward differs from the oracle's by one low_speed quantum (see the module docstring): copy the device's reward
    and episode sums of THAT env into the oracle.  Returns how many; raises beyond `budget` or for any other kind of difference."""
    r = rew_dev.detach().cpu().float()
    diff = (r - o.rew).abs()
    bad = (diff > (EC.ATOL + EC.RTOL * o.rew.abs())).nonzero().flatten().tolist()
    if not bad:
        return 0
    k = K.REWARD_NAMES.index("low_speed")
    sums = sums_dev.detach().cpu().float()
    for e in bad:
        assert float(diff[e]) <= LOW_SPEED_QUANTUM, "env %d: reward differs by %.3e (not a low_speed threshold flip)" % (e, float(diff[e]))
        other = torch.cat((sums[e, :k], sums[e, k + 1:])) - torch.cat((o.episode_sums[e, :k], o.episode_sums[e, k + 1:]))
        reset_now = bool(o.reset[e])
        if not reset_now:      # every other term's running sum must still agree
            assert float(other.abs().max()) <= 1e-4, "env %d: a term other than low_speed differs" % e
        o.rew[e] = r[e]
        o.episode_sums[e] = sums[e]
    assert len(bad) <= budget, "%d envs flipped a low_speed threshold (budget %d)" % (len(bad), budget)
    return len(bad)


def compare(buf, o, tag, flips, budget=2, check_obs=True):
    flips[0] += forgive_low_speed(buf.rew, buf.view("episode_sums"), o, budget - flips[0])
    EC.compare_state(Holder(buf), o, tag, check_obs=check_obs)


def plant(buf, o, gen, csc=397):
    """Episode lengths / step counter that put time-outs, command resamples and a push inside a short window."""
    N = buf.N
    ep = torch.randint(0, 2400, (N,), generator=gen)
    ep[: min(N, 8)] = torch.tensor([2399, 2398, 799, 1598, 0, 2396, 798, 2397])[: min(N, 8)]
    buf.episode_length.copy_(ep.to(buf.episode_length.device))
    buf.counters[0] = csc
    if o is not None:
        o.ep_len = ep.clone()
        o.common_step_counter = csc
