"""CPU oracle for the XBot-L env side of the hot path.

TEST INFRASTRUCTURE.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this; the product path (`humanoid-gym_amd/`) never does and fails loudly without its
HIP library.

This is a from-scratch, mask-driven restatement (no `nonzero`, no per-term method dispatch, one flat
state record) of what the reference computes in
  envs/custom/humanoid_env.py:83-142,189-269,272-540  and  envs/base/legged_robot.py:84-235,304-397
It is written with torch CPU fp32 ops because the reference *is* torch fp32 code: using the same
primitive ops in the same order makes the restatement pin bit-tight against fixtures recorded from
the reference itself (tests/golden/gen_fixtures.py -> tests/golden/env_trace.npz), which is how the
oracle is pinned (the reference ships no tests of its own).

Third-party arithmetic that is NOT under /root/reference: `isaacgym.torch_utils` (Isaac Gym Preview 4,
closed; pinned only by the comment at reference setup.py:43).  `quat_rotate_inverse`, `quat_apply`,
`get_euler_xyz`, `torch_rand_float` are restated below from their published definitions as standard
xyzw-quaternion math: parity is UNPINNED at that boundary; the in-container check
against scipy's Rotation (the library reference scripts/sim2sim.py:76-79 uses for the same quantities)
is tests/test_oracle_quat.py.

Randomness is external: every draw the reference takes from torch's global generator is an explicit
argument here (tables indexed by env id), so the oracle, the reference replay and the HIP kernels
consume identical numbers.
"""
import math

import torch

from . import xbot_constants as C

TWO_PI = 2 * math.pi


# ----------------------------------------------------------------------------------------------
# isaacgym.torch_utils restated (see module docstring)
# ----------------------------------------------------------------------------------------------
def quat_rotate_inverse(q, v):
    """v rotated by the inverse of unit quaternion q=(x,y,z,w).  Call sites legged_robot.py:133-135,215."""
    w = q[:, 3]
    u = q[:, :3]
    a = v * (2.0 * w ** 2 - 1.0).unsqueeze(-1)
    b = torch.cross(u, v, dim=-1) * w.unsqueeze(-1) * 2.0
    c = u * (u * v).sum(dim=-1, keepdim=True) * 2.0
    return a - b + c


def quat_apply(q, v):
    """v rotated by q.  Call site legged_robot.py:312."""
    u = q[:, :3]
    t = torch.cross(u, v, dim=-1) * 2
    return v + q[:, 3:] * t + torch.cross(u, t, dim=-1)


def euler_xyz_wrapped(q):
    """get_euler_xyz (each angle % 2pi) followed by the (-pi, pi] wrap of legged_robot.py:50-55."""
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    roll = torch.atan2(2.0 * (w * x + y * z), w * w - x * x - y * y + z * z)
    sp = 2.0 * (w * y - z * x)
    pitch = torch.where(sp.abs() >= 1, torch.sign(sp) * (math.pi / 2.0), torch.asin(sp))
    yaw = torch.atan2(2.0 * (w * z + x * y), w * w + x * x - y * y - z * z)
    e = torch.stack((roll % TWO_PI, pitch % TWO_PI, yaw % TWO_PI), dim=1)
    e[e > math.pi] -= TWO_PI
    return e


def uniform(lo, hi, u):
    """torch_rand_float: (hi-lo)*rand+lo, python-double scalars against an fp32 tensor."""
    return (hi - lo) * u + lo


def quat_apply_yaw(q, v):
    """utils/math.py:39-43: rotate (N,P,3) points by the yaw part of q (N,4) -- x,y zeroed, renormalised."""
    qy = q.clone()
    qy[:, :2] = 0.0
    qy = qy / qy.norm(p=2, dim=-1).clamp(min=1e-9).unsqueeze(-1)
    n, p = v.shape[0], v.shape[1]
    return quat_apply(qy.repeat(1, p).view(-1, 4), v.reshape(-1, 3)).view(n, p, 3)


class TerrainSpec:
    """What the env needs of a terrain map (utils/terrain.py) when cfg.terrain.mesh_type is heightfield / trimesh."""

    def __init__(self, origins, levels, types, env_length, curriculum, height_samples=None, height_points=None,
                 border_size=0.0, horizontal_scale=0.1, vertical_scale=0.005):
        self.origins = origins.float()                 # (rows, cols, 3)  legged_robot.py:696
        self.levels = levels.clone().long()            # (N,)             :693
        self.types = types.clone().long()              # (N,)             :694
        self.max_level = origins.shape[0]              # :695
        self.env_length = env_length                   # Terrain.env_length
        self.curriculum = bool(curriculum)
        self.height_samples = height_samples           # (tot_rows, tot_cols) int16, :586
        self.height_points = height_points             # (P, 3) base-frame sample points (z = 0), :743-759
        self.border_size, self.hscale, self.vscale = border_size, horizontal_scale, vertical_scale


# ----------------------------------------------------------------------------------------------
class SimState:
    """The four sim tensors of legged_robot.py:449-457, kept separately (dof state split in pos/vel)."""

    def __init__(self, n):
        self.root = torch.zeros(n, 13)
        self.dof_pos = torch.zeros(n, 12)
        self.dof_vel = torch.zeros(n, 12)
        self.contact = torch.zeros(n, C.NUM_BODIES, 3)
        self.rigid = torch.zeros(n, C.NUM_BODIES, 13)

    def load(self, root, dof_state, contact, rigid):
        n = self.root.shape[0]
        self.root.copy_(root)
        d = dof_state.view(n, 12, 2)
        self.dof_pos.copy_(d[..., 0])
        self.dof_vel.copy_(d[..., 1])
        self.contact.copy_(contact.view(n, C.NUM_BODIES, 3))
        self.rigid.copy_(rigid.view(n, C.NUM_BODIES, 13))


def grid_origins(n):
    """legged_robot.py:699-708 (plane terrain => regular grid)."""
    cols = math.floor(math.sqrt(n))
    rows = math.ceil(n / cols)
    xx, yy = torch.meshgrid(torch.arange(rows), torch.arange(cols), indexing="ij")
    o = torch.zeros(n, 3)
    o[:, 0] = C.ENV_SPACING * xx.flatten()[:n]
    o[:, 1] = C.ENV_SPACING * yy.flatten()[:n]
    return o


class XBotEnvOracle:
    """State record + the per-step functions.  All tensors are (N, k) fp32 unless noted."""

    def __init__(self, n, frictions=None, body_mass=None, frame_stack=C.FRAME_STACK,
                 c_frame_stack=C.C_FRAME_STACK, use_ref_actions=False, terrain=None, command_curriculum=False, max_curriculum=1.0,
                 heading_command=True, extra_rewards=None):
        self.n = n
        # user-defined reward terms, legged_robot.py:518-541: name -> (fn(oracle) -> (N,) raw value, scale).  Every non-zero scale
        # names a `_reward_<name>` found by getattr; the terms are evaluated and summed in the order of the scales dict, which
        # class_to_dict builds with dir(), i.e. alphabetically (helpers.py:41-56) -- so extra names interleave with the 22 built-in
        # ones.  A name that IS one of the 22 replaces that term (a subclass overriding the method).
        self.extra_rewards = dict(extra_rewards or {})
        self.extra_sums = {k: torch.zeros(n) for k in self.extra_rewards}
        self.extras_extra = None
        self.use_ref_actions = bool(use_ref_actions)      # cfg.env.use_ref_actions, humanoid_config.py:49 (False for XBot-L)
        # generic LeggedRobot options XBot-L leaves off
        self.terrain = terrain                            # TerrainSpec or None (plane)
        self.heading_command = bool(heading_command)         # cfg.commands.heading_command, legged_robot.py:311-314,331-334
        self.command_curriculum = bool(command_curriculum)   # cfg.commands.curriculum, legged_robot.py:179-180,422-431
        self.max_curriculum = max_curriculum
        self.cmd_range_x = [C.CMD_LIN_VEL_X[0], C.CMD_LIN_VEL_X[1]]   # python doubles, moved by the command curriculum
        self.measured_heights = None
        self.H = frame_stack
        self.Hc = c_frame_stack
        self.sim = SimState(n)
        self.env_origins = grid_origins(n) if terrain is None else terrain.origins[terrain.levels, terrain.types].clone()   # :697
        self.base_init = torch.tensor(C.BASE_INIT_STATE)
        self.friction = torch.ones(n, 1) if frictions is None else frictions.clone().view(n, 1)
        self.body_mass = torch.full((n, 1), 15.0) if body_mass is None else body_mass.clone().view(n, 1)
        self.p_gains = torch.tensor(C.P_GAINS)
        self.d_gains = torch.tensor(C.D_GAINS)
        self.torque_limits = torch.tensor(C.EFFORT) * C.TORQUE_LIMIT_FACTOR
        self.default_dof_pos = torch.tensor(C.DEFAULT_DOF_POS).unsqueeze(0)
        self.noise_vec = torch.zeros(C.NUM_SINGLE_OBS)          # humanoid_env.py:176-186
        self.noise_vec[5:17] = C.NOISE_DOF_POS * C.OBS_SCALE_DOF_POS
        self.noise_vec[17:29] = C.NOISE_DOF_VEL * C.OBS_SCALE_DOF_VEL
        self.noise_vec[41:44] = C.NOISE_ANG_VEL * C.OBS_SCALE_ANG_VEL
        self.noise_vec[44:47] = C.NOISE_QUAT * C.OBS_SCALE_QUAT
        self.commands_scale = torch.tensor([C.OBS_SCALE_LIN_VEL, C.OBS_SCALE_LIN_VEL, C.OBS_SCALE_ANG_VEL])
        self.gravity = torch.tensor([0.0, 0.0, -1.0]).repeat(n, 1)
        self.forward = torch.tensor([1.0, 0.0, 0.0]).repeat(n, 1)
        self.sim.root[:] = self.base_init
        self.sim.root[:, :3] += self.env_origins
        # --- per-env state (legged_robot.py:460-516, base_task.py:71-94, humanoid_env.py:78-79)
        z = lambda *s: torch.zeros(n, *s)
        self.ep_len = torch.zeros(n, dtype=torch.long)
        self.common_step_counter = 0
        self.commands = z(4)
        self.actions = z(12)
        self.last_actions = z(12)
        self.last_last_actions = z(12)
        self.last_dof_vel = z(12)
        self.last_root_vel = z(6)
        self.torques = z(12)
        self.feet_air_time = z(2)
        self.last_contacts = torch.zeros(n, 2, dtype=torch.bool)
        self.feet_height = z(2)
        self.last_feet_z = torch.full((n, 2), 0.05)     # python scalar 0.05 in the reference (broadcasts)
        self.ref_dof_pos = z(12)
        self.push_force = z(3)
        self.push_torque = z(3)
        self.episode_sums = z(C.NUM_REWARDS)
        self.base_lin_vel = quat_rotate_inverse(self.sim.root[:, 3:7], self.sim.root[:, 7:10])
        self.base_ang_vel = quat_rotate_inverse(self.sim.root[:, 3:7], self.sim.root[:, 10:13])
        self.projected_gravity = quat_rotate_inverse(self.sim.root[:, 3:7], self.gravity)
        self.base_euler = euler_xyz_wrapped(self.sim.root[:, 3:7])
        # history, oldest -> newest (legged_robot.py:509-516)
        self.obs_hist = torch.zeros(n, self.H, C.NUM_SINGLE_OBS)
        self.priv_hist = torch.zeros(n, self.Hc, C.SINGLE_NUM_PRIV_OBS)
        # outputs
        self.rew = z()
        self.reset = torch.ones(n, dtype=torch.bool)
        self.time_out = torch.zeros(n, dtype=torch.bool)
        self.extras_time_outs = None       # stale-by-design, SURVEY.md App. A item 2
        self.extras_episode = None
        self.obs = z(self.H * C.NUM_SINGLE_OBS)
        self.priv = z(self.Hc * C.SINGLE_NUM_PRIV_OBS)

    # ------------------------------------------------------------------ gait clock (E9)
    def _sin_phase(self):
        """humanoid_env.py:100-108: int64 * python double -> fp32, / 0.64, * 2pi, sin."""
        phase = self.ep_len * C.DT / C.CYCLE_TIME
        return phase, torch.sin(2 * torch.pi * phase)

    def _stance_mask(self):
        """humanoid_env.py:105-118."""
        _, s = self._sin_phase()
        m = torch.zeros(self.n, 2)
        m[:, 0] = s >= 0
        m[:, 1] = s < 0
        m[torch.abs(s) < 0.1] = 1
        return m

    def _ref_pose(self):
        """humanoid_env.py:121-142."""
        _, s = self._sin_phase()
        sl = torch.where(s > 0, torch.zeros_like(s), s)
        sr = torch.where(s < 0, torch.zeros_like(s), s)
        ref = torch.zeros(self.n, 12)
        s1 = C.TARGET_JOINT_POS_SCALE
        s2 = 2 * s1
        ref[:, 2] = sl * s1
        ref[:, 3] = sl * s2
        ref[:, 4] = sl * s1
        ref[:, 8] = sr * s1
        ref[:, 9] = sr * s2
        ref[:, 10] = sr * s1
        ref[torch.abs(s) < 0.1] = 0
        return ref

    # ------------------------------------------------------------------ E1, E2 (clip), E3
    def pre_physics(self, actions_in, u_delay, z_act):
        """humanoid_env.py:189-197 + legged_robot.py:90-91.  u_delay (N,), z_act (N,12).
        With use_ref_actions the reference adds ref_action = 2 * ref_dof_pos (the pose of the LAST compute_observations,
        humanoid_env.py:142) to the caller's tensor IN PLACE before the clip (:190-191): `actions_in` is mutated here too."""
        if self.use_ref_actions:
            actions_in += 2 * self.ref_dof_pos
        a = torch.clip(actions_in, -C.CLIP_ACTIONS, C.CLIP_ACTIONS)
        delay = u_delay.view(-1, 1) * C.ACTION_DELAY
        a = (1 - delay) * a + delay * self.actions
        a = a + C.ACTION_NOISE * z_act * a
        self.actions = torch.clip(a, -C.CLIP_ACTIONS, C.CLIP_ACTIONS)
        return self.actions

    def pd_torques(self):
        """legged_robot.py:340-356 evaluated on the current dof state."""
        t = self.p_gains * (self.actions * C.ACTION_SCALE + self.default_dof_pos - self.sim.dof_pos) \
            - self.d_gains * self.sim.dof_vel
        self.torques = torch.clip(t, -self.torque_limits, self.torque_limits)
        return self.torques

    # ------------------------------------------------------------------ commands (E5)
    def _resample_commands(self, mask, u3):
        """legged_robot.py:322-336 for envs where mask; u3 (N,3) = draws for x, y, heading."""
        m = mask
        cx = uniform(self.cmd_range_x[0], self.cmd_range_x[1], u3[:, 0])
        cy = uniform(C.CMD_LIN_VEL_Y[0], C.CMD_LIN_VEL_Y[1], u3[:, 1])
        self.commands[:, 0] = torch.where(m, cx, self.commands[:, 0])
        self.commands[:, 1] = torch.where(m, cy, self.commands[:, 1])
        if self.heading_command:
            self.commands[:, 3] = torch.where(m, uniform(C.CMD_HEADING[0], C.CMD_HEADING[1], u3[:, 2]), self.commands[:, 3])
        else:
            self.commands[:, 2] = torch.where(m, uniform(C.CMD_ANG_VEL_YAW[0], C.CMD_ANG_VEL_YAW[1], u3[:, 2]), self.commands[:, 2])
        keep = (torch.norm(self.commands[:, :2], dim=1) > 0.2).unsqueeze(1)
        self.commands[:, :2] = torch.where(m.unsqueeze(1), self.commands[:, :2] * keep, self.commands[:, :2])

    # ------------------------------------------------------------------ rewards (E8)
    def _rewards(self):
        """The 22 terms in alphabetical order; returns (N,22) raw values.  humanoid_env.py:272-540."""
        s = self.sim
        n = self.n
        feet = list(C.FEET_BODIES)
        knees = list(C.KNEE_BODIES)
        contact = s.contact[:, feet, 2] > 5.0
        r = []
        # action_smoothness :530-540
        t1 = torch.sum(torch.square(self.last_actions - self.actions), dim=1)
        t2 = torch.sum(torch.square(self.actions + self.last_last_actions - 2 * self.last_actions), dim=1)
        t3 = 0.05 * torch.sum(torch.abs(self.actions), dim=1)
        r.append(t1 + t2 + t3)
        # base_acc :386-393
        r.append(torch.exp(-torch.norm(self.last_root_vel - s.root[:, 7:13], dim=1) * 3))
        # base_height :374-384
        stance = self._stance_mask()
        mh = torch.sum(s.rigid[:, feet, 2] * stance, dim=1) / torch.sum(stance, dim=1)
        bh = s.root[:, 2] - (mh - 0.05)
        r.append(torch.exp(-torch.abs(bh - C.BASE_HEIGHT_TARGET) * 100))
        # collision :523-528
        r.append(torch.sum(1.0 * (torch.norm(s.contact[:, [C.BASE_BODY], :], dim=-1) > 0.1), dim=1))
        # default_joint_pos :362-372
        jd = s.dof_pos - self.default_dof_pos
        yr = torch.norm(jd[:, :2], dim=1) + torch.norm(jd[:, 6:8], dim=1)
        yr = torch.clamp(yr - 0.1, 0, 50)
        r.append(torch.exp(-yr * 100) - 0.01 * torch.norm(jd, dim=1))
        # dof_acc :516-521
        r.append(torch.sum(torch.square((self.last_dof_vel - s.dof_vel) / C.DT), dim=1))
        # dof_vel :509-514
        r.append(torch.sum(torch.square(s.dof_vel), dim=1))
        # feet_air_time :320-334 (stateful)
        filt = torch.logical_or(torch.logical_or(contact, stance), self.last_contacts)
        self.last_contacts = contact
        first = (self.feet_air_time > 0.0) * filt
        self.feet_air_time += C.DT
        air = self.feet_air_time.clamp(0, 0.5) * first
        self.feet_air_time *= ~filt
        r.append(air.sum(dim=1))
        # feet_clearance :446-467 (stateful)
        fz = s.rigid[:, feet, 2] - 0.05
        self.feet_height += fz - self.last_feet_z
        self.last_feet_z = fz
        swing = 1 - stance
        pos = torch.abs(self.feet_height - C.TARGET_FEET_HEIGHT) < 0.01
        r.append(torch.sum(pos * swing, dim=1))
        self.feet_height *= ~contact
        # feet_contact_forces :355-360
        r.append(torch.sum((torch.norm(s.contact[:, feet, :], dim=-1) - C.MAX_CONTACT_FORCE).clip(0, 400), dim=1))
        # feet_contact_number :336-344
        r.append(torch.mean(torch.where(contact == stance, 1.0, -0.3), dim=1))
        # feet_distance :282-292
        r.append(self._dist_reward(s.rigid[:, feet, :2], C.MAX_DIST))
        # foot_slip :308-318
        sp = torch.sqrt(torch.norm(s.rigid[:, feet, 7:9], dim=2))
        r.append(torch.sum(sp * contact, dim=1))
        # joint_pos :272-280 -- uses the PREVIOUS step's ref pose
        e = torch.norm(s.dof_pos - self.ref_dof_pos, dim=1)
        r.append(torch.exp(-2 * e) - 0.2 * e.clamp(0, 0.5))
        # knee_distance :295-305
        r.append(self._dist_reward(s.rigid[:, knees, :2], C.MAX_DIST / 2))
        # low_speed :469-500
        vx = self.base_lin_vel[:, 0]
        cx = self.commands[:, 0]
        av, ac = torch.abs(vx), torch.abs(cx)
        low = av < 0.5 * ac
        high = av > 1.2 * ac
        ok = ~(low | high)
        mism = torch.sign(vx) != torch.sign(cx)
        ls = torch.zeros(n)
        ls = torch.where(low, torch.full_like(ls, -1.0), ls)
        ls = torch.where(high, torch.zeros_like(ls), ls)
        ls = torch.where(ok, torch.full_like(ls, 1.2), ls)
        ls = torch.where(mism, torch.full_like(ls, -2.0), ls)
        r.append(ls * (cx.abs() > 0.1))
        # orientation :346-353
        qm = torch.exp(-torch.sum(torch.abs(self.base_euler[:, :2]), dim=1) * 10)
        og = torch.exp(-torch.norm(self.projected_gravity[:, :2], dim=1) * 20)
        r.append((qm + og) / 2.0)
        # torques :502-507
        r.append(torch.sum(torch.square(self.torques), dim=1))
        # track_vel_hard :408-425
        le = torch.norm(self.commands[:, :2] - self.base_lin_vel[:, :2], dim=1)
        ae = torch.abs(self.commands[:, 2] - self.base_ang_vel[:, 2])
        r.append((torch.exp(-le * 10) + torch.exp(-ae * 10)) / 2.0 - 0.2 * (le + ae))
        # tracking_ang_vel :436-444
        r.append(torch.exp(-torch.square(self.commands[:, 2] - self.base_ang_vel[:, 2]) * C.TRACKING_SIGMA))
        # tracking_lin_vel :427-434
        r.append(torch.exp(-torch.sum(torch.square(self.commands[:, :2] - self.base_lin_vel[:, :2]), dim=1)
                           * C.TRACKING_SIGMA))
        # vel_mismatch_exp :396-406
        lm = torch.exp(-torch.square(self.base_lin_vel[:, 2]) * 10)
        am = torch.exp(-torch.norm(self.base_ang_vel[:, :2], dim=1) * 5.0)
        r.append((lm + am) / 2.0)
        return torch.stack(r, dim=1)

    @staticmethod
    def _dist_reward(xy, max_df):
        d = torch.norm(xy[:, 0, :] - xy[:, 1, :], dim=1)
        d_min = torch.clamp(d - C.MIN_DIST, -0.5, 0.0)
        d_max = torch.clamp(d - max_df, 0, 0.5)
        return (torch.exp(-torch.abs(d_min) * 100) + torch.exp(-torch.abs(d_max) * 100)) / 2

    # ------------------------------------------------------------------ reset (E10)
    # ------------------------------------------------------------------ curricula / heights
    def _update_terrain_curriculum(self, m, r_level):
        """legged_robot.py:400-420 for envs where m; r_level (N,) int64 = the randint_like draw in [0, max_level)."""
        t = self.terrain
        s = self.sim
        distance = torch.norm(s.root[:, :2] - self.env_origins[:, :2], dim=1)
        move_up = distance > t.env_length / 2
        move_down = (distance < torch.norm(self.commands[:, :2], dim=1) * C.EPISODE_LENGTH_S * 0.5) * ~move_up
        lv = t.levels + (1 * move_up - 1 * move_down)
        lv = torch.where(lv >= t.max_level, r_level, torch.clip(lv, 0))
        t.levels = torch.where(m, lv, t.levels)
        self.env_origins = torch.where(m.unsqueeze(1), t.origins[t.levels, t.types], self.env_origins)

    def _update_command_curriculum(self, m):
        """legged_robot.py:422-431: widen lin_vel_x by 0.5 each way (capped) when the resetting envs tracked well."""
        k = C.REWARD_NAMES.index("tracking_lin_vel")
        mean_sum = self.episode_sums[m, k].mean()
        if bool(mean_sum / float(C.MAX_EPISODE_LENGTH) > 0.8 * C.REWARD_SCALES_DT[k]):     # fp32 tensor against python doubles
            lo = min(max(self.cmd_range_x[0] - 0.5, -self.max_curriculum), 0.0)
            hi = min(max(self.cmd_range_x[1] + 0.5, 0.0), self.max_curriculum)
            self.cmd_range_x = [lo, hi]
            return True
        return False

    def _get_heights(self):
        """legged_robot.py:761-795: min of three neighbouring height samples under each yaw-rotated sample point."""
        t = self.terrain
        s = self.sim
        pts = quat_apply_yaw(s.root[:, 3:7], t.height_points.unsqueeze(0).repeat(self.n, 1, 1)) + s.root[:, :3].unsqueeze(1)
        pts = pts + t.border_size
        pts = (pts / t.hscale).long()
        px = torch.clip(pts[:, :, 0].reshape(-1), 0, t.height_samples.shape[0] - 2)
        py = torch.clip(pts[:, :, 1].reshape(-1), 0, t.height_samples.shape[1] - 2)
        h = torch.min(torch.min(t.height_samples[px, py], t.height_samples[px + 1, py]), t.height_samples[px, py + 1])
        return h.view(self.n, -1) * t.vscale

    def _reset_masked(self, m, u_dof, u_cmd, u_xy=None, r_level=None):
        """legged_robot.py:163-215 + humanoid_env.py:264-269 for envs where m (bool N).
        u_xy (N,2): spawn jitter draws of custom origins (:385); r_level (N,): see _update_terrain_curriculum."""
        if not bool(m.any()):
            return False
        s = self.sim
        mc = m.unsqueeze(1)
        if self.terrain is not None and self.terrain.curriculum:
            self._update_terrain_curriculum(m, r_level)
        self.curriculum_moved = False
        if self.command_curriculum and self.common_step_counter % C.MAX_EPISODE_LENGTH == 0:
            self.curriculum_moved = self._update_command_curriculum(m)
        s.dof_pos[:] = torch.where(mc, self.default_dof_pos + uniform(-0.1, 0.1, u_dof), s.dof_pos)
        s.dof_vel[:] = torch.where(mc, torch.zeros_like(s.dof_vel), s.dof_vel)
        init = self.base_init.unsqueeze(0).repeat(self.n, 1)
        init[:, :3] += self.env_origins
        if self.terrain is not None:                      # custom origins: within 1 m of the tile centre (:382-385)
            init[:, :2] += uniform(-1.0, 1.0, u_xy)
        s.root[:] = torch.where(mc, init, s.root)
        self._resample_commands(m, u_cmd)
        for name in ("last_last_actions", "actions", "last_actions", "last_dof_vel", "feet_air_time"):
            t = getattr(self, name)
            setattr(self, name, torch.where(mc, torch.zeros_like(t), t))
        self.ep_len = torch.where(m, torch.zeros_like(self.ep_len), self.ep_len)
        self.reset = self.reset | m
        cnt = m.sum()
        self.extras_episode = (self.episode_sums * mc).sum(dim=0) / cnt / C.EPISODE_LENGTH_S
        self.episode_sums = torch.where(mc, torch.zeros_like(self.episode_sums), self.episode_sums)
        self.extras_extra = {k: (v * m).sum() / cnt / C.EPISODE_LENGTH_S for k, v in self.extra_sums.items()}
        self.extra_sums = {k: torch.where(m, torch.zeros_like(v), v) for k, v in self.extra_sums.items()}
        self.extras_time_outs = self.time_out.clone()
        self.base_euler = euler_xyz_wrapped(s.root[:, 3:7])
        g = quat_rotate_inverse(s.root[:, 3:7], self.gravity)
        self.projected_gravity = torch.where(mc, g, self.projected_gravity)
        self.obs_hist = torch.where(m.view(-1, 1, 1), torch.zeros_like(self.obs_hist), self.obs_hist)
        self.priv_hist = torch.where(m.view(-1, 1, 1), torch.zeros_like(self.priv_hist), self.priv_hist)
        return True

    # ------------------------------------------------------------------ observations (E11)
    def _observe(self, z_obs):
        """humanoid_env.py:200-262.  z_obs (N,47) standard normal draws."""
        s = self.sim
        phase, sp = self._sin_phase()
        self.ref_dof_pos = self._ref_pose()
        sin_pos = sp.unsqueeze(1)
        cos_pos = torch.cos(2 * torch.pi * phase).unsqueeze(1)
        stance = self._stance_mask()
        contact = s.contact[:, list(C.FEET_BODIES), 2] > 5.0
        cmd_in = torch.cat((sin_pos, cos_pos, self.commands[:, :3] * self.commands_scale), dim=1)
        q = (s.dof_pos - self.default_dof_pos) * C.OBS_SCALE_DOF_POS
        dq = s.dof_vel * C.OBS_SCALE_DOF_VEL
        diff = s.dof_pos - self.ref_dof_pos
        priv = torch.cat((cmd_in, q, dq, self.actions, diff,
                          self.base_lin_vel * C.OBS_SCALE_LIN_VEL,
                          self.base_ang_vel * C.OBS_SCALE_ANG_VEL,
                          self.base_euler * C.OBS_SCALE_QUAT,
                          self.push_force[:, :2], self.push_torque,
                          self.friction, self.body_mass / 30.0, stance, contact), dim=-1)
        frame = torch.cat((cmd_in, q, dq, self.actions,
                           self.base_ang_vel * C.OBS_SCALE_ANG_VEL,
                           self.base_euler * C.OBS_SCALE_QUAT), dim=-1)
        frame_noisy = frame + z_obs * self.noise_vec * C.NOISE_LEVEL
        self.obs_hist = torch.cat((self.obs_hist[:, 1:], frame_noisy.unsqueeze(1)), dim=1)
        self.priv_hist = torch.cat((self.priv_hist[:, 1:], priv.unsqueeze(1)), dim=1)
        self.obs = self.obs_hist.reshape(self.n, -1)
        self.priv = self.priv_hist.reshape(self.n, -1)
        return frame_noisy, priv

    # ------------------------------------------------------------------ post-physics (E4-E12)
    def post_physics(self, u_cmd, u_dof, u_push, z_obs, u_xy=None, r_level=None):
        """legged_robot.py:119-151 + the clip of :105-108.

        u_cmd (N,6): [0:3] draws of the callback resample, [3:6] draws of the reset resample;
        u_dof (N,12): reset joint offsets; u_push (N,5): push lin xy + ang xyz; z_obs (N,47);
        u_xy (N,2), r_level (N,): only with a terrain map (see _reset_masked)."""
        s = self.sim
        self.ep_len = self.ep_len + 1
        self.common_step_counter += 1
        quat = s.root[:, 3:7]
        self.base_lin_vel = quat_rotate_inverse(quat, s.root[:, 7:10])
        self.base_ang_vel = quat_rotate_inverse(quat, s.root[:, 10:13])
        self.projected_gravity = quat_rotate_inverse(quat, self.gravity)
        self.base_euler = euler_xyz_wrapped(quat)
        # callback :304-320
        self._resample_commands(self.ep_len % C.RESAMPLE_STEPS == 0, u_cmd[:, 0:3])
        if self.heading_command:
            fwd = quat_apply(quat, self.forward)
            heading = torch.atan2(fwd[:, 1], fwd[:, 0])
            ang = self.commands[:, 3] - heading
            ang = ang % TWO_PI                                   # utils/math.py:46-49 wrap_to_pi
            ang = ang - TWO_PI * (ang > math.pi)
            self.commands[:, 2] = torch.clip(0.5 * ang, -1.0, 1.0)
        if self.terrain is not None and self.terrain.height_samples is not None:
            self.measured_heights = self._get_heights()      # :316-317
        pushed = self.common_step_counter % C.PUSH_INTERVAL == 0
        if pushed:                                           # humanoid_env.py:83-98
            self.push_force[:, :2] = uniform(-C.MAX_PUSH_VEL_XY, C.MAX_PUSH_VEL_XY, u_push[:, 0:2])
            s.root[:, 7:9] = self.push_force[:, :2]
            self.push_torque = uniform(-C.MAX_PUSH_ANG_VEL, C.MAX_PUSH_ANG_VEL, u_push[:, 2:5])
            s.root[:, 10:13] = self.push_torque
        # termination :156-161
        self.reset = torch.any(torch.norm(s.contact[:, [C.BASE_BODY], :], dim=-1) > 1.0, dim=1)
        self.time_out = self.ep_len > C.MAX_EPISODE_LENGTH
        self.reset = self.reset | self.time_out
        # reward :217-235.  The reference evaluates the terms one after the other in alphabetical order, and two built-in terms
        # are stateful (feet_air_time: feet_air_time / last_contacts; feet_clearance: feet_height / last_feet_z): a user-defined
        # term sorting before "feet_air_time" sees those four buffers as compute_reward finds them, one sorting after
        # "feet_clearance" sees them updated.  (Names strictly between the two are not supported by this restatement.)
        sc = lambda name: self.extra_rewards[name][0](self) * (self.extra_rewards[name][1] * C.DT)
        assert not any("feet_air_time" <= n <= "feet_clearance" for n in self.extra_rewards)
        extra = {n: sc(n) for n in self.extra_rewards if n < "feet_air_time"}
        raw = self._rewards()
        extra.update({n: sc(n) for n in self.extra_rewards if n > "feet_clearance"})
        self.rew = torch.zeros(self.n)
        scales = torch.tensor(C.REWARD_SCALES_DT, dtype=torch.float64)
        self.reward_terms = torch.zeros(self.n, C.NUM_REWARDS)
        for name in sorted(set(C.REWARD_NAMES) | set(extra)):
            if name == "termination":      # legged_robot.py:533-534: not in the function list; added after the clip (:229-235)
                continue
            if name in extra:
                self.rew = self.rew + extra[name]
                self.extra_sums[name] = self.extra_sums[name] + extra[name]
                continue
            k = C.REWARD_NAMES.index(name)
            term = raw[:, k] * C.REWARD_SCALES_DT[k]
            self.rew = self.rew + term
            self.episode_sums[:, k] += term
            self.reward_terms[:, k] = term
        self.rew = torch.clip(self.rew, min=0.0)
        if "termination" in extra:         # legged_robot.py:231-235
            self.rew = self.rew + extra["termination"]
            self.extra_sums["termination"] = self.extra_sums["termination"] + extra["termination"]
        any_reset = self._reset_masked(self.reset.clone(), u_dof, u_cmd[:, 3:6], u_xy, r_level)
        frame, priv_frame = self._observe(z_obs)
        self.last_last_actions = self.last_actions.clone()
        self.last_actions = self.actions.clone()
        self.last_dof_vel = s.dof_vel.clone()
        self.last_root_vel = s.root[:, 7:13].clone()
        obs = torch.clip(self.obs, -C.CLIP_OBS, C.CLIP_OBS)
        priv = torch.clip(self.priv, -C.CLIP_OBS, C.CLIP_OBS)
        return obs, priv, self.rew, self.reset, dict(any_reset=any_reset, pushed=pushed,
                                                     frame=frame, priv_frame=priv_frame)

    # ------------------------------------------------------------------ construction tail / reset()
    def prime(self, u_dof, u_cmd3, z_obs, u_xy=None, r_level=None):
        """XBotLFreeEnv.__init__ tail, humanoid_env.py:80-81: reset_idx(all) then compute_observations."""
        self._reset_masked(torch.ones(self.n, dtype=torch.bool), u_dof, u_cmd3, u_xy, r_level)
        self._observe(z_obs)
