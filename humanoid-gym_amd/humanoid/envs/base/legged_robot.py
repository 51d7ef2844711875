"""LeggedRobot VecEnv on the MI355X hot path.

Same constructor, attributes and step/reset contract as the reference class (envs/base/legged_robot.py:57-235),
but the body of `step` is ONE fused HIP launch (`hgym_env_step_synth`: action processing, synthetic physics,
post-physics pipeline, rewards, mask-driven resets, observation stacking) plus a one-workgroup finaliser, on
env-major SoA buffers owned by this object.  PhysX is replaced by the synthetic backend of SURVEY.md §8d; an
external simulator can instead write the four sim tensors and call `post_physics_step()`.

There is no CPU implementation behind this class: without libhgym_hip.so and a gfx950 device it raises.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from humanoid.envs.base.base_task import BaseTask
from humanoid.utils.helpers import class_to_dict, shard_seed
from .legged_robot_config import LeggedRobotCfg

# reward terms the fused kernel implements, in the alphabetical order class_to_dict imposes
KERNEL_REWARD_TERMS = [
    "action_smoothness", "base_acc", "base_height", "collision", "default_joint_pos", "dof_acc", "dof_vel", "feet_air_time",
    "feet_clearance", "feet_contact_forces", "feet_contact_number", "feet_distance", "foot_slip", "joint_pos", "knee_distance",
    "low_speed", "orientation", "torques", "track_vel_hard", "tracking_ang_vel", "tracking_lin_vel", "vel_mismatch_exp"]

DOF_NAMES = ["%s_%s_joint" % (s, j) for s in ("left", "right")
             for j in ("leg_roll", "leg_yaw", "leg_pitch", "knee", "ankle_pitch", "ankle_roll")]
DOF_EFFORT = [100.0, 100.0, 250.0, 250.0, 100.0, 100.0] * 2          # urdf/XBot-L.urdf <limit effort=...>
DOF_LOWER = [-0.44, -1.05, -1.57, -1.05, -0.70, -0.44, -1.57, -1.05, -1.31, -1.10, -0.87, -0.44]
DOF_UPPER = [1.57, 1.05, 1.31, 1.10, 0.87, 0.44, 0.44, 1.05, 1.57, 1.05, 0.70, 0.44]
NOMINAL_BASE_MASS = 15.0   # synthetic backend: base link + collapsed upper body (URDF base_link alone is 9.96 kg)


class _CommandRanges(dict):
    """`env.command_ranges`: with cfg.commands.curriculum the lin_vel_x range lives on the device (the curriculum kernel moves
    it, legged_robot.py:422-431); reading that key reads it back."""
    live_x = None

    def __getitem__(self, key):
        if key == "lin_vel_x" and self.live_x is not None:
            return [float(v) for v in self.live_x.cpu()]
        return dict.__getitem__(self, key)


class LeggedRobot(BaseTask):
    def __init__(self, cfg: LeggedRobotCfg, sim_params, physics_engine, sim_device, headless):
        self.cfg = cfg
        self.sim_params = sim_params
        self.height_samples = None
        self.debug_viz = False
        self.init_done = False
        self._parse_cfg(self.cfg)
        super().__init__(self.cfg, sim_params, physics_engine, sim_device, headless)
        self._init_buffers()
        self._prepare_reward_function()
        self.init_done = True

    # ------------------------------------------------------------------ configuration
    def _parse_cfg(self, cfg):
        self.dt = self.cfg.control.decimation * self.sim_params.dt
        self.obs_scales = self.cfg.normalization.obs_scales
        self.reward_scales = class_to_dict(self.cfg.rewards.scales)
        self.command_ranges = _CommandRanges(class_to_dict(self.cfg.commands.ranges))
        if self.cfg.terrain.mesh_type not in ["heightfield", "trimesh"]:
            self.cfg.terrain.curriculum = False
        self.max_episode_length_s = self.cfg.env.episode_length_s
        self.max_episode_length = np.ceil(self.max_episode_length_s / self.dt)
        self.cfg.domain_rand.push_interval = np.ceil(self.cfg.domain_rand.push_interval_s / self.dt)

    def native_config_digest(self):
        """Hash of the HgymEnvConfig the next launches will carry (by value): lets a caller that captured launches into a HIP
        graph notice that the configuration was edited since."""
        return hash(bytes(self._ncfg)) if self._ncfg is not None else None

    def _native_config(self):
        """XBotLCfg -> HgymEnvConfig.  Scalars are combined in python double arithmetic and rounded to fp32 last,
        exactly where the reference's tensors round them."""
        from hgym import _lib as L, default_env_config
        cfg = self.cfg
        for need in ("frame_stack", "c_frame_stack", "num_single_obs", "single_num_privileged_obs"):
            if not hasattr(cfg.env, need):
                raise NotImplementedError("the MI355X hot path implements the XBot-L observation layout (cfg.env.%s missing)" % need)
        if cfg.env.num_single_obs != L.OBS_FRAME or cfg.env.single_num_privileged_obs != L.PRIV_FRAME:
            raise NotImplementedError("observation frame sizes other than 47/73 are not built")
        if cfg.terrain.mesh_type not in ("plane", "heightfield", "trimesh"):
            raise ValueError("Terrain mesh type not recognised. Allowed types are [plane, heightfield, trimesh]")
        if cfg.terrain.measure_heights and cfg.terrain.mesh_type == "plane":
            raise NotImplementedError("measure_heights on a plane returns zeros in the reference; there is nothing to sample")
        c = default_env_config(self.num_envs, seed=shard_seed(getattr(cfg, "seed", 5)), frame_stack=cfg.env.frame_stack,
                               c_frame_stack=cfg.env.c_frame_stack)
        c.decimation = cfg.control.decimation
        c.sim_dt = self.sim_params.dt
        c.dt = self.dt
        c.max_episode_length = int(self.max_episode_length)
        c.resample_steps = int(cfg.commands.resampling_time / self.dt)
        c.push_interval = int(cfg.domain_rand.push_interval)
        c.push_robots = int(bool(cfg.domain_rand.push_robots))
        c.add_noise = int(bool(cfg.noise.add_noise))
        c.use_ref_actions = int(bool(getattr(cfg.env, "use_ref_actions", False)))
        c.clip_actions = cfg.normalization.clip_actions
        c.clip_obs = cfg.normalization.clip_observations
        c.action_scale = cfg.control.action_scale
        c.action_delay = getattr(cfg.domain_rand, "action_delay", 0.0)
        c.action_noise = getattr(cfg.domain_rand, "action_noise", 0.0)
        c.noise_level = cfg.noise.noise_level
        ns_, os_ = cfg.noise.noise_scales, cfg.normalization.obs_scales
        vec = [0.0] * 5 + [ns_.dof_pos * os_.dof_pos] * 12 + [ns_.dof_vel * os_.dof_vel] * 12 + [0.0] * 12 + \
              [ns_.ang_vel * os_.ang_vel] * 3 + [ns_.quat * os_.quat] * 3
        for k, v in enumerate(vec):
            c.obs_noise[k] = v
        c.scale_lin_vel, c.scale_ang_vel, c.scale_dof_pos = os_.lin_vel, os_.ang_vel, os_.dof_pos
        c.scale_dof_vel, c.scale_quat = os_.dof_vel, os_.quat
        r = self.command_ranges
        c.cmd_x_lo, c.cmd_x_span = r["lin_vel_x"][0], r["lin_vel_x"][1] - r["lin_vel_x"][0]
        c.cmd_y_lo, c.cmd_y_span = r["lin_vel_y"][0], r["lin_vel_y"][1] - r["lin_vel_y"][0]
        c.cmd_h_lo, c.cmd_h_span = r["heading"][0], r["heading"][1] - r["heading"][0]
        c.cmd_yaw_lo, c.cmd_yaw_span = r["ang_vel_yaw"][0], r["ang_vel_yaw"][1] - r["ang_vel_yaw"][0]
        c.heading_command = int(bool(cfg.commands.heading_command))
        pv, pa = cfg.domain_rand.max_push_vel_xy, getattr(cfg.domain_rand, "max_push_ang_vel", 0.0)
        c.push_vel_lo, c.push_vel_span, c.push_ang_lo, c.push_ang_span = -pv, pv - (-pv), -pa, pa - (-pa)
        for j, name in enumerate(self.dof_names):
            c.default_dof_pos[j] = cfg.init_state.default_joint_angles[name]
            kp = kd = 0.0
            for key in cfg.control.stiffness:
                if key in name:
                    kp, kd = cfg.control.stiffness[key], cfg.control.damping[key]
            c.p_gains[j], c.d_gains[j] = kp, kd
            c.torque_limits[j] = float(np.float32(DOF_EFFORT[j]) * np.float32(cfg.safety.torque_limit))
            c.dof_lower[j], c.dof_upper[j] = DOF_LOWER[j], DOF_UPPER[j]
        st = cfg.init_state
        for i, v in enumerate(st.pos + st.rot + st.lin_vel + st.ang_vel):
            c.base_init_state[i] = v
        c.base_body, c.feet_bodies[0], c.feet_bodies[1] = 0, 6, 12
        c.knee_bodies[0], c.knee_bodies[1] = 4, 10
        rw = cfg.rewards
        c.only_positive_rewards = int(bool(rw.only_positive_rewards))
        c.base_height_target, c.min_dist, c.max_dist = rw.base_height_target, rw.min_dist, rw.max_dist
        c.target_joint_pos_scale, c.target_feet_height = rw.target_joint_pos_scale, rw.target_feet_height
        c.cycle_time, c.tracking_sigma, c.max_contact_force = rw.cycle_time, rw.tracking_sigma, rw.max_contact_force
        c.episode_length_s = self.max_episode_length_s
        return c

    # ------------------------------------------------------------------ construction
    terrain_class = None          # set by create_sim: humanoid.utils.terrain.Terrain (XBotLFreeEnv: HumanoidTerrain)

    def _build_terrain(self):
        """mesh_type heightfield / trimesh (legged_robot.py:543-586,683-697): the procedural map, each env's terrain level
        (row) and type (column), origins from the map.  The synthetic physics backend does not collide with the map; it
        feeds the reset origins, the terrain curriculum and the height measurements."""
        from humanoid.utils.terrain import Terrain
        tc = self.cfg.terrain
        self.terrain = (self.terrain_class or Terrain)(tc, self.num_envs)
        self.height_samples = torch.tensor(self.terrain.heightsamples).view(self.terrain.tot_rows, self.terrain.tot_cols).to(self.device)
        self.custom_origins = True
        max_init_level = tc.max_init_terrain_level if tc.curriculum else tc.num_rows - 1
        self.terrain_levels = torch.randint(0, max_init_level + 1, (self.num_envs,), device=self.device)
        self.terrain_types = torch.div(torch.arange(self.num_envs, device=self.device), (self.num_envs / tc.num_cols),
                                       rounding_mode="floor").to(torch.long)
        self.max_terrain_level = tc.num_rows
        self.terrain_origins = torch.from_numpy(self.terrain.env_origins).to(self.device).to(torch.float)

    def _init_height_points(self):
        """(num_envs, P, 3) base-frame sample grid, legged_robot.py:743-759 (the kernel keeps one (P, 3) copy)."""
        y = torch.tensor(self.cfg.terrain.measured_points_y, device=self.device)
        x = torch.tensor(self.cfg.terrain.measured_points_x, device=self.device)
        grid_x, grid_y = torch.meshgrid(x, y, indexing="ij")
        self.num_height_points = grid_x.numel()
        points = torch.zeros(self.num_envs, self.num_height_points, 3, device=self.device)
        points[:, :, 0] = grid_x.flatten()
        points[:, :, 1] = grid_y.flatten()
        return points

    def _get_heights(self, env_ids=None):
        """legged_robot.py:761-795 on the CURRENT base poses (the step itself samples before its resets; this is for callers
        such as play scripts)."""
        if self.cfg.terrain.mesh_type == "plane":
            return torch.zeros(self.num_envs, getattr(self, "num_height_points", 0), device=self.device)
        b = self._buf
        b.height_pose.copy_(self.root_states[:, :7])
        self._L.check(self._L.lib.hgym_measure_heights(C.byref(self._ncfg), C.byref(self._st_s), self._stream()), "hgym_measure_heights")
        h = b.measured_heights.clone()
        return h if env_ids is None else h[env_ids]

    def create_sim(self):
        """Where the reference builds the PhysX scene (legged_robot.py:588-708): allocate the device buffers,
        draw the per-env friction / base-mass randomisation, lay the envs out on the plane grid."""
        import hgym
        if not (str(self.device).startswith("cuda") and torch.cuda.is_available()):
            raise RuntimeError("LeggedRobot needs a gfx950 device (sim_device=%r); there is no CPU path" % (self.device,))
        self.up_axis_idx = 2
        self.num_dof = self.num_dofs = 12
        self.num_bodies = 13
        self.dof_names = list(DOF_NAMES)
        self._hgym = hgym
        self._L = hgym._lib
        self.cfg.seed = getattr(self.cfg, "seed", 5)
        self._ncfg = None
        self.custom_origins = False
        if self.cfg.terrain.mesh_type in ("heightfield", "trimesh"):
            self._build_terrain()
        self.feet_indices = torch.tensor([6, 12], dtype=torch.long, device=self.device)
        self.knee_indices = torch.tensor([4, 10], dtype=torch.long, device=self.device)
        self.penalised_contact_indices = torch.tensor([0], dtype=torch.long, device=self.device)
        self.termination_contact_indices = torch.tensor([0], dtype=torch.long, device=self.device)

    def _init_buffers(self):
        L = self._L
        self._ncfg = self._native_config()
        b = self._buf = self._hgym.EnvBuffers(self._ncfg, self.device)
        N = self.num_envs
        tc = self.cfg.terrain
        if self.custom_origins:
            measure = bool(tc.measure_heights)
            if measure:
                self.height_points = self._init_height_points()
                if not getattr(LeggedRobot, "_warned_heights", False):
                    LeggedRobot._warned_heights = True
                    print("measure_heights: heights are sampled into env.measured_heights every step; the XBot-L observation layout "
                          "is unchanged (the reference's concatenation, humanoid_env.py:246-248, does not fit its configured sizes)")
            b.set_terrain(self.terrain_origins, self.terrain_levels, self.terrain_types, self.terrain.env_length, tc.curriculum,
                          height_samples=self.height_samples if measure else None,
                          height_points=self.height_points[0] if measure else None, border_size=tc.border_size,
                          horizontal_scale=tc.horizontal_scale, vertical_scale=tc.vertical_scale)
            self.terrain_levels = b.terrain_levels           # the kernel's copy is the live one
            self.measured_heights = b.measured_heights if measure else 0
        else:
            self.measured_heights = 0
        if self.cfg.commands.curriculum:
            b.set_command_curriculum(self.command_ranges["lin_vel_x"], self.cfg.commands.max_curriculum)
            self.command_ranges.live_x = b.command_range_x
        dr = self.cfg.domain_rand
        if dr.randomize_friction:                         # legged_robot.py:257-269 (256 buckets)
            buckets = (dr.friction_range[1] - dr.friction_range[0]) * torch.rand(256, 1) + dr.friction_range[0]
            b.f["friction"].copy_(buckets[torch.randint(0, 256, (N,))].view(1, N))
        mass = torch.full((N,), NOMINAL_BASE_MASS)
        if dr.randomize_base_mass:                        # legged_robot.py:296-302
            mass += torch.from_numpy(np.random.uniform(dr.added_mass_range[0], dr.added_mass_range[1], N)).float()
        b.f["body_mass"].copy_(mass.view(1, N))
        # two output sets: consecutive steps never overwrite the tensors handed out by the previous step, so an
        # algorithm that keeps references for one step (reference ppo.py:99-100) stays correct
        z = lambda *s: torch.zeros(*s, device=self.device)
        self._outs = [(b.obs, b.priv_obs), (z(N, self.num_obs), z(N, self.num_privileged_obs))]
        self._flip = 0
        self._bound_out = None
        # A/B knob, read once (a captured rollout graph keeps the protocol active at capture; OnPolicyRunner's graph key includes it)
        self._rows_ahead = os.environ.get("HGYM_ROWS_AHEAD", "1") != "0"
        self._l0_ahead = os.environ.get("HGYM_L0_AHEAD", "1") != "0"      # the actor's first layer carried across launches (rollout_step)
        self._sim_s, self._st_s = b.sim_struct(), b.state_struct()
        self._noise_none = b.noise_struct()
        self.common_step_counter_buf = b.counters
        # views with the reference's names and shapes
        self.root_states = b.root_view()
        self.dof_pos, self.dof_vel = b.dof_pos_view(), b.dof_vel_view()
        self.base_quat = self.root_states[:, 3:7]
        self.contact_forces, self.rigid_state = b.contact_view(), b.rigid_view()
        for name in ("commands", "actions", "last_actions", "last_last_actions", "last_dof_vel", "last_root_vel", "torques",
                     "feet_air_time", "feet_height", "last_feet_z", "ref_dof_pos", "base_lin_vel", "base_ang_vel",
                     "projected_gravity", "env_origins"):
            setattr(self, name, b.view(name))
        self.base_euler_xyz = b.view("base_euler")
        self.rand_push_force, self.rand_push_torque = b.view("push_force"), b.view("push_torque")
        self.env_frictions, self.body_mass = b.view("friction"), b.view("body_mass")
        self.last_contacts = b.view("last_contacts")
        self.rew_buf, self.time_out_buf = b.rew, b.time_out
        self._reset_buf = b.reset
        self.obs_buf, self.privileged_obs_buf = self._outs[0]
        c = self._ncfg
        dev = self.device
        self.default_dof_pos = torch.tensor(list(c.default_dof_pos), device=dev).unsqueeze(0)
        self.default_joint_pd_target = self.default_dof_pos.clone()
        self.p_gains = torch.tensor(list(c.p_gains), device=dev).expand(N, 12)
        self.d_gains = torch.tensor(list(c.d_gains), device=dev).expand(N, 12)
        self.torque_limits = torch.tensor(list(c.torque_limits), device=dev)
        self.base_init_state = torch.tensor(list(c.base_init_state), device=dev)
        self.noise_scale_vec = torch.tensor(list(c.obs_noise), device=dev)
        self.commands_scale = torch.tensor([c.scale_lin_vel, c.scale_lin_vel, c.scale_ang_vel], device=dev)
        self.gravity_vec = torch.tensor([0.0, 0.0, -1.0], device=dev).repeat(N, 1)
        self.forward_vec = torch.tensor([1.0, 0.0, 0.0], device=dev).repeat(N, 1)
        self.extras = {}
        self._refresh_extras()

    def _prepare_reward_function(self):
        """legged_robot.py:518-541: drop zero scales, multiply by dt once; every surviving name is a reward term.  The 22 XBot-L terms
        are evaluated inside the env kernel.  Any OTHER name -- or one of the 22 that a task subclass overrides with a method of its
        own -- must be a `_reward_<name>` method of this object (as in the reference, which finds them all by name): those are
        evaluated in torch between the two launches of a split step (hgym_env_step_begin / hgym_env_step_end) and summed at their
        place in the reference's alphabetical order.  What such a method sees is the state compute_reward starts from
        (legged_robot.py:128-161 done); the four buffers two built-in terms update while the reference walks the list
        (feet_air_time, last_contacts, feet_height, last_feet_z) are read in their PRE-reward state even by a term that sorts
        after feet_air_time / feet_clearance."""
        for key in list(self.reward_scales.keys()):
            if self.reward_scales[key] == 0:
                self.reward_scales.pop(key)
            else:
                self.reward_scales[key] *= self.dt
        self.reward_names = [k for k in self.reward_scales]
        own = lambda name: getattr(type(self), "_reward_" + name, None) is not None      # LeggedRobot / XBotLFreeEnv define none themselves
        custom = [k for k in self.reward_names if k not in KERNEL_REWARD_TERMS or own(k)]
        if any(k in ("feet_air_time", "feet_clearance") for k in custom):
            raise NotImplementedError("the two stateful terms (feet_air_time, feet_clearance) cannot be overridden")
        self._custom_terms = []
        for name in custom:
            fn = getattr(self, "_reward_" + name)           # AttributeError for a scale without a method, as in the reference
            self._custom_terms.append((name, fn, self.reward_scales[name]))
        for k, name in enumerate(KERNEL_REWARD_TERMS):
            self._ncfg.reward_scales[k] = 0.0 if name in custom else self.reward_scales.get(name, 0.0)
        sums = self._buf.f["episode_sums"]
        self.episode_sums = {name: sums[KERNEL_REWARD_TERMS.index(name)] for name in self.reward_names if name not in custom}
        if custom:
            # custom term j is summed right before the first built-in term that sorts at or after it; `termination` is not in the
            # reference's function list at all: it is added after the only-positive clip (legged_robot.py:229-235, :533-534)
            self._buf.set_custom_rewards([len(KERNEL_REWARD_TERMS) + 1 if name == "termination" else
                                          sum(1 for b in KERNEL_REWARD_TERMS if b < name) for name in custom])
            self._st_s = self._buf.state_struct()
            for j, name in enumerate(custom):
                self.episode_sums[name] = self._buf.custom_sums[j]
        self._refresh_extras()

    # ------------------------------------------------------------------ runner-visible buffers
    @property
    def episode_length_buf(self):
        return self._buf.episode_length

    @episode_length_buf.setter
    def episode_length_buf(self, value):      # the runner REBINDS this attribute (on_policy_runner.py:104-106)
        self._buf.episode_length.copy_(value)

    @property
    def reset_buf(self):
        return self._reset_buf

    @reset_buf.setter
    def reset_buf(self, value):
        self._reset_buf.copy_(value.to(torch.bool))

    @property
    def common_step_counter(self):
        return int(self._buf.counters[0])

    @common_step_counter.setter
    def common_step_counter(self, v):
        self._buf.counters[0] = int(v)

    def _refresh_extras(self):
        b = self._buf
        cust = [c[0] for c in getattr(self, "_custom_terms", [])]
        self.extras["episode"] = {"rew_" + n: (b.extras_custom[cust.index(n)] if n in cust else b.extras_episode[KERNEL_REWARD_TERMS.index(n)])
                                  for n in self.reward_names} if hasattr(self, "reward_names") else {}
        if self.cfg.env.send_timeouts:
            self.extras["time_outs"] = b.extras_time_outs
        # curriculum info (legged_robot.py:203-207): live device scalars (the reference refreshes them on steps with a reset)
        if self.cfg.terrain.mesh_type == "trimesh" and getattr(self, "custom_origins", False) and self.extras["episode"]:
            if not hasattr(self, "_terrain_level_mean"):
                self._terrain_level_mean = torch.zeros((), device=self.device)
            self.extras["episode"]["terrain_level"] = self._terrain_level_mean
        if self.cfg.commands.curriculum and b.command_range_x is not None and self.extras["episode"]:
            self.extras["episode"]["max_command_x"] = b.command_range_x[1]

    def bind_outputs(self, obs, priv):
        """Native extension: make the next step write its observations straight into caller memory
        (the rollout storage slot), removing the add_transitions copy.  Pass None to go back."""
        self._bound_out = None if obs is None else (obs, priv)

    def bind_transition(self, sink, defer_finalize=False):
        """Native extension: `sink` = dict(values, rewards, dones, step, gamma) of caller tensors (or None).  While bound, the
        step finaliser also stores the scalar columns of the transition (what PPO.process_env_step would launch
        hgym_store_step for) and bumps the policy's sampling-step counter: one launch per vec-step instead of three.
        defer_finalize: step() does not launch the finaliser at all; the caller collects it with take_pending_finalize() and
        hands it to the next policy launch (PPO.act(env_fin=...)) or to run_finalize() -- before the next step()."""
        self._sink = sink
        # (a split step -- user-defined reward terms -- always runs its finaliser itself)
        self._defer = bool(defer_finalize) and sink is not None and not getattr(self, "_custom_terms", None)

    def bind_log_sink(self, on):
        """Native extension: while on, the step finaliser keeps the runner's per-step logging book-keeping on the device
        (HgymEnvOut.log_*: running episode return / length per env, the last-100-episodes rings, the per-step sums of
        extras["episode"]); read with log_sink_read().  Only the kernel's own reward terms are covered."""
        on = (bool(on) and not getattr(self, "_custom_terms", None)
              and set(self.extras.get("episode", {})) == {"rew_" + n for n in self.reward_names})
        self._buf.log_sink = on
        if on:
            self._buf.log_cur.zero_()
            self._buf.log_stats.zero_()
        return on

    def log_sink_read(self):
        """(episode means dict, returns of the last <= 100 finished episodes, their lengths) since the last call; one device
        read-back.  The per-step sums are cleared, the rings persist (they are the runner's rewbuffer / lenbuffer)."""
        ls = self._buf.log_stats.cpu()
        steps = max(float(ls[22]), 1.0)
        ep = {"rew_" + n: float(ls[KERNEL_REWARD_TERMS.index(n)]) / steps for n in self.reward_names}
        k = int(ls[25])
        self._buf.log_stats[:23].zero_()
        return ep, ls[32:32 + k].tolist(), ls[132:132 + k].tolist()

    def take_pending_finalize(self):
        p, self._pending_fin = getattr(self, "_pending_fin", None), None
        return p

    def run_finalize(self, fin):
        """Run a postponed step finaliser on its own (the last step of a rollout has no following policy launch)."""
        if fin is not None:
            self._L.check(self._L.lib.hgym_env_finalize(C.byref(fin[0]), C.byref(fin[1]), C.byref(fin[2]), self._stream()),
                          "hgym_env_finalize")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def seek(self, iteration, steps_per_iteration):
        """Native extension (OnPolicyRunner.load): put the common step counter -- the Philox counter word of every env draw
        (commands, pushes, noise, reset offsets) and the clock of the push / curriculum intervals -- where a run that has done
        `iteration` learning iterations has it, so that a resumed run continues the env's draw streams instead of replaying them
        from step 0.  The reference's checkpoint carries no generator state either (on_policy_runner.py:274-281); the counter is
        a function of the iteration number.  The history ring position is left alone."""
        if not hasattr(self, "_seek_base"):                    # no reset() yet: this IS the fresh env
            self._seek_base = int(self._buf.counters[0])
        self._buf.counters[0] = self._seek_base + int(iteration) * int(steps_per_iteration)

    # ------------------------------------------------------------------ fused rollout step (native extension)
    # One launch per vec-step: PPO.act, this env's step (synthetic-physics backend) and the previous step's finaliser
    # (include/hgym.h: hgym_rollout_begin / _step / _end).  Used by OnPolicyRunner when nothing on the host needs the per-step
    # results; every other caller keeps act() + step().
    def rollout_fused_supported(self, net):
        """The fused rollout launch WITH the critic's tiles inline serves this env / net (rollout_fused_mode == "inline")."""
        return self.rollout_fused_mode(net) == "inline"

    def rollout_fused_mode(self, net):
        """"inline": hgym_rollout_step with the critic's tiles beside the actor's (one actor + one critic workgroup per 32 envs fit the
        chip in one round: up to 4096 envs on 256 CUs); "deferred": the launch without critic tiles (values = NULL: one actor + env
        workgroup per 32 envs, up to 8192 envs) and the critic once over the stored rows after the rollout (PPO.deferred_values) --
        8192 envs: collection 3.68 -> 3.49 ms against the two-launch path, same call (profiles/r05d_deferred_values_ab.txt; round 3's
        attempt WITH the critic's tiles in a second round of workgroups lost: profiles/r03_seq_rollout_8192_negative_result.txt); at
        4096 envs the inline form wins (2.15 vs 2.76 ms: half the chip would idle); None: the two-launch path.  HGYM_ROLLOUT_CRITIC=inline|deferred|auto (default auto) restricts / forces the choice."""
        c, nc = self._ncfg, net.cfg
        generic = c.custom_origins or c.terrain_curriculum or c.num_height_points > 0 or c.command_curriculum or not c.heading_command
        cus = max(int(self._L.lib.hgym_device_cus()), 1)
        # the fused launch never calls step(): a task class that overrides step() / post_physics_step() (a wrapper, extra
        # book-keeping around the step) must keep getting its own code, i.e. the act() + step() path
        own_step = type(self).step is LeggedRobot.step and type(self).post_physics_step is LeggedRobot.post_physics_step
        own_step = own_step and not getattr(self, "_custom_terms", None)       # user-defined reward terms: the two-launch step
        ok = bool(own_step and not generic and not c.use_ref_actions and c.frame_stack == 15 and c.c_frame_stack == 3 and self.num_envs % 32 == 0
                  and nc.precision == self._L.BF16 and nc.actor_layers == 4 and nc.critic_layers == 4 and nc.actor_dims[1] == 512
                  and nc.critic_dims[1] == 768 and nc.num_actions == 12 and getattr(self.cfg.env, "send_timeouts", False))
        if not ok:
            return None
        want = os.environ.get("HGYM_ROLLOUT_CRITIC", "auto").lower()
        tiles = self.num_envs // 32
        if want != "deferred" and 2 * tiles <= cus:
            return "inline"
        if want != "inline" and tiles <= cus:
            return "deferred"
        return None

    def rollout_begin(self, step_counter, num_steps):
        if getattr(self, "_pending_fin", None) is not None:
            raise RuntimeError("a postponed step finaliser is pending; run it before a fused rollout")
        self._ro_T, self._ro_prev, self._ro_ahead, self._ro_l0 = int(num_steps), None, None, None
        self._L.check(self._L.lib.hgym_rollout_begin(C.byref(self._st_s), self._L.i64ptr(step_counter), C.c_void_p(self._buf.rollout_scratch.data_ptr()),
                                                     (self._ro_T - 1) & 1, self._stream()), "hgym_rollout_begin")

    def rollout_step(self, net, i, obs, priv, next_obs, next_priv, sink, seed, out, shadow=None, ahead=None, shadow_next=None):
        """Step i of the rollout begun with rollout_begin: actions / mu / sigma / logp / values of PPO.act into `out`, this env's
        step on those actions with the observations written to next_obs / next_priv, the transition sink of step i stored by
        the finaliser that rides in step i + 1 (or in rollout_end).  The last step uses the primary rew / reset / time_out
        buffers, so that they read as after a plain step() once the rollout is over.  shadow: optional (obs_bf16, priv_bf16)
        storage-slot tensors receiving the bf16 of `obs` / `priv` (HgymObsShadow).  ahead: (obs, priv) rows the NEXT step will write
        its observations to (HgymEnvOut.obs_ahead / priv_ahead): this launch writes their older frames off its critical path, and the
        next call -- recognised by its next_obs being that tensor -- skips the copy.  Same rows either way (HGYM_ROWS_AHEAD=0: never).
        shadow_next: the bf16 shadow tensor of next_obs (the NEXT call's shadow[0]), or None.  With `ahead` given there is a next
        launch: this one also forms 20 of the 24 k-steps of the actor's first layer for next_obs (HgymEnvOut.l0_ahead; the rows are
        this call's rows shifted by a frame) and writes columns [0, 640) of shadow_next; the next call -- recognised by its obs being
        this call's next_obs -- starts from those sums (l0_ready).  Bit-identical outputs (HGYM_L0_AHEAD=0: never)."""
        sh = None if shadow is None else net.shadow_struct(*shadow)
        L = self._L
        parity = (self._ro_T - 1 - i) & 1
        if out["values"] is None:           # deferred values (hgym_rollout_step with values = NULL): no critic tiles, hence none of their side jobs
            assert ahead is None and shadow_next is None and sink["values"] is None
            o = self._buf.out_struct(next_obs, next_priv, sink, True, alt=bool(parity))
            prev = self._ro_prev
            L.check(L.lib.hgym_rollout_step(C.byref(net.cfg), C.byref(net.struct), C.byref(self._ncfg), C.byref(self._sim_s), C.byref(self._st_s),
                                            C.byref(o), C.byref(prev[0]) if prev is not None else None, L.fptr(obs), L.fptr(priv),
                                            int(seed) & 0xFFFFFFFFFFFFFFFF, L.fptr(out["actions"]), L.fptr(out["mu"]), L.fptr(out["sigma"]),
                                            L.fptr(out["logp"]), None, C.c_void_p(self._buf.rollout_scratch.data_ptr()), parity,
                                            None if sh is None else C.byref(sh), self._stream()), "hgym_rollout_step")
            self._ro_prev = (o, parity, sink)
            self._ro_ahead = self._ro_l0 = None
            self.obs_buf, self.privileged_obs_buf = next_obs, next_priv
            return
        o = self._buf.out_struct(next_obs, next_priv, sink, True, alt=bool(parity))
        if not self._rows_ahead:
            ahead = None
        o.obs_older_ready = int(self._ro_prev is not None and self._ro_ahead is not None
                                and self._ro_ahead == (next_obs.data_ptr(), next_priv.data_ptr()))
        if ahead is not None:
            a_obs, a_priv = ahead
            assert a_obs.is_contiguous() and a_obs.shape == next_obs.shape and a_obs.data_ptr() != next_obs.data_ptr()
            assert a_priv.is_contiguous() and a_priv.shape == next_priv.shape and a_priv.data_ptr() != next_priv.data_ptr()
            o.obs_ahead, o.priv_ahead = L.fptr(a_obs), L.fptr(a_priv)
        self._ro_ahead = None if ahead is None else (ahead[0].data_ptr(), ahead[1].data_ptr())
        # the carried first layer rides with the rows-ahead protocol (the steady-state launch): ready if the previous launch formed the
        # sums for exactly these rows, with the same shadow arrangement
        l0 = self._ro_l0
        if (l0 is not None and o.obs_older_ready and l0[0] == obs.data_ptr()
                and l0[2] == (None if shadow is None else shadow[0].data_ptr())):
            o.l0_ready = L.fptr(self._buf.l0_partial(l0[1]))
        self._ro_l0 = None
        if self._l0_ahead and ahead is not None and (shadow is None) == (shadow_next is None):
            k = i & 1
            o.l0_ahead = L.fptr(self._buf.l0_partial(k))
            if shadow_next is not None:
                assert shadow_next.dtype == torch.bfloat16 and shadow_next.is_contiguous() and shadow_next.shape[0] == self.num_envs
                o.obs_bf16_ahead, o.ld_obs_bf16_ahead = C.c_void_p(shadow_next.data_ptr()), shadow_next.shape[-1]
            self._ro_l0 = (next_obs.data_ptr(), k, None if shadow_next is None else shadow_next.data_ptr())
        prev = self._ro_prev
        L.check(L.lib.hgym_rollout_step(C.byref(net.cfg), C.byref(net.struct), C.byref(self._ncfg), C.byref(self._sim_s), C.byref(self._st_s),
                                        C.byref(o), C.byref(prev[0]) if prev is not None else None, L.fptr(obs), L.fptr(priv),
                                        int(seed) & 0xFFFFFFFFFFFFFFFF, L.fptr(out["actions"]), L.fptr(out["mu"]), L.fptr(out["sigma"]),
                                        L.fptr(out["logp"]), L.fptr(out["values"]), C.c_void_p(self._buf.rollout_scratch.data_ptr()), parity,
                                        None if sh is None else C.byref(sh), self._stream()), "hgym_rollout_step")
        self._ro_prev = (o, parity, sink)          # keeps the struct (and the tensors it points at) alive for the next launch
        self.obs_buf, self.privileged_obs_buf = next_obs, next_priv

    def rollout_end(self):
        """The finaliser of the last step on its own."""
        o, parity, _ = self._ro_prev
        self._L.check(self._L.lib.hgym_rollout_end(C.byref(self._ncfg), C.byref(self._st_s), C.byref(o), C.c_void_p(self._buf.rollout_scratch.data_ptr()),
                                                   parity, self._stream()), "hgym_rollout_end")
        self._ro_prev = None

    def _next_out(self):
        if self._bound_out is not None:
            obs, priv = self._bound_out
        else:
            self._flip ^= 1
            obs, priv = self._outs[self._flip]
        return obs, priv, self._buf.out_struct(obs, priv, getattr(self, "_sink", None), getattr(self, "_defer", False))

    # ------------------------------------------------------------------ VecEnv API
    def step(self, actions):
        """legged_robot.py:84-109.  Returns (obs, privileged_obs, rewards, dones, extras)."""
        L = self._L
        a = actions.to(self.device, torch.float32)
        if not a.is_contiguous():
            a = a.contiguous()
        if getattr(self, "_pending_fin", None) is not None:
            raise RuntimeError("the previous step's finaliser was postponed (bind_transition(defer_finalize=True)) and never run")
        obs, priv, out = self._next_out()
        if self._custom_terms:
            self._split_step(out, L.fptr(a))
        else:
            L.check(L.lib.hgym_env_step_synth(C.byref(self._ncfg), C.byref(self._sim_s), C.byref(self._st_s), C.byref(out),
                                              L.fptr(a), self._stream()), "hgym_env_step_synth")
        if out.defer_finalize:
            self._pending_fin = (self._ncfg, self._st_s, out)
        if self._ncfg.use_ref_actions and a.data_ptr() != actions.data_ptr():
            actions.copy_(a)         # the reference mutates the caller's tensor (humanoid_env.py:190-191: actions += ref_action)
        if hasattr(self, "_terrain_level_mean"):
            torch.mean(self.terrain_levels.float(), dim=0, out=self._terrain_level_mean)
        self.obs_buf, self.privileged_obs_buf = obs, priv
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras

    def post_physics_step(self):
        """For an external simulator that has written root_states / dof / contact / rigid tensors itself."""
        L = self._L
        obs, priv, out = self._next_out()
        if self._custom_terms:
            self._split_step(out, None)
        else:
            L.check(L.lib.hgym_post_physics(C.byref(self._ncfg), C.byref(self._sim_s), C.byref(self._st_s), C.byref(out),
                                            C.byref(self._noise_none), self._stream()), "hgym_post_physics")
        self.obs_buf, self.privileged_obs_buf = obs, priv

    def _split_step(self, out, actions_ptr):
        """One step as two launches with the user-defined reward terms evaluated in between (legged_robot.py:217-235 with terms the
        kernel does not know): derive -> `_reward_<name>()` * scale for each -> finish."""
        L = self._L
        L.check(L.lib.hgym_env_step_begin(C.byref(self._ncfg), C.byref(self._sim_s), C.byref(self._st_s), C.byref(out),
                                          C.byref(self._noise_none), actions_ptr, self._stream()), "hgym_env_step_begin")
        for j, (name, fn, scale) in enumerate(self._custom_terms):
            self._buf.custom_rew[j].copy_(fn() * scale)
        L.check(L.lib.hgym_env_step_end(C.byref(self._ncfg), C.byref(self._sim_s), C.byref(self._st_s), C.byref(out),
                                        C.byref(self._noise_none), self._stream()), "hgym_env_step_end")

    def reset_idx(self, env_ids):
        """Only the all-envs form exists on the device (per-env resets are mask-driven inside the step kernel)."""
        if len(env_ids) == 0:
            return
        if len(env_ids) != self.num_envs:
            raise NotImplementedError("partial reset_idx from the host is not part of the hot path; resets are mask-driven on the device")
        L = self._L
        out = self._buf.out_struct(self.obs_buf, self.privileged_obs_buf)
        L.check(L.lib.hgym_env_reset_all(C.byref(self._ncfg), C.byref(self._sim_s), C.byref(self._st_s), C.byref(out),
                                         C.byref(self._noise_none), self._stream()), "hgym_env_reset_all")

    def reset(self):
        """legged_robot.py:112-117: reset every env, then one zero-action step."""
        self.reset_idx(torch.arange(self.num_envs, device=self.device))
        obs, privileged_obs, _, _, _ = self.step(torch.zeros(self.num_envs, self.num_actions, device=self.device))
        if not hasattr(self, "_seek_base"):
            # iteration 0 of seek(): the counter right after the FIRST reset (OnPolicyRunner.__init__ calls it), recorded here and not
            # lazily in seek() -- a load() after learn() would otherwise take the already-advanced counter for the base
            self._seek_base = int(self._buf.counters[0])
        return obs, privileged_obs

    def _prime(self):
        """XBotLFreeEnv.__init__ tail (humanoid_env.py:78-81)."""
        L = self._L
        out = self._buf.out_struct(self.obs_buf, self.privileged_obs_buf)
        L.check(L.lib.hgym_env_prime(C.byref(self._ncfg), C.byref(self._sim_s), C.byref(self._st_s), C.byref(out),
                                     C.byref(self._noise_none), self._stream()), "hgym_env_prime")
