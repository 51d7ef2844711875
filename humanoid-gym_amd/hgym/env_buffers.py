"""Caller-owned buffers of the env side: allocates the env-major SoA state, the sim tensors and the step
outputs as torch tensors and presents them to the C-ABI as the pointer structs of include/hgym.h."""
import ctypes as C

import torch

from . import _lib as L


def default_env_config(num_envs, seed=5, frame_stack=15, c_frame_stack=3):
    cfg = L.EnvConfig()
    L.check(L.lib.hgym_env_config_default(C.byref(cfg), int(num_envs)), "hgym_env_config_default")
    cfg.seed = int(seed)
    cfg.frame_stack = int(frame_stack)
    cfg.c_frame_stack = int(c_frame_stack)
    return cfg


def grid_origins(n, spacing=3.0):
    """Env origins of a plane terrain: regular grid (reference legged_robot.py:699-708)."""
    import math
    cols = math.floor(math.sqrt(n))
    rows = math.ceil(n / cols)
    xx, yy = torch.meshgrid(torch.arange(rows), torch.arange(cols), indexing="ij")
    o = torch.zeros(n, 3)
    o[:, 0] = spacing * xx.flatten()[:n]
    o[:, 1] = spacing * yy.flatten()[:n]
    return o


class EnvBuffers:
    """All device memory of one env shard.

    State fields are stored [C][N] (env-major SoA); `view(name)` returns the (N,C) transposed view the
    reference API exposes.  Sim tensors are either library-native SoA (`sim_layout="soa"`, used by the
    synthetic physics backend) or Isaac-Gym-shaped AoS (`"aos"`: root (N,13), dof_state (N*12,2),
    contact (N*13,3), rigid (N*13,13)), both described to the kernels by strides.
    """

    def __init__(self, cfg, device, sim_layout="soa", obs_out=None, priv_out=None):
        self.cfg = cfg
        self.device = torch.device(device)
        N = self.N = int(cfg.num_envs)
        H, HC = int(cfg.frame_stack), int(cfg.c_frame_stack)
        dev = self.device
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=dev)
        # one [136][N] allocation, fields adjacent in HgymEnvState order (lets the step kernel stage the state with
        # plain 16-byte row copies; separately allocated fields also work, through the per-field pointers)
        self._state = z(sum(c for _, c in L.ENV_STATE_FIELDS), N)
        self.f, off = {}, 0
        for name, c in L.ENV_STATE_FIELDS:
            self.f[name] = self._state[off:off + c]
            off += c
        self.episode_length = z(N, dtype=torch.int64)
        self.counters = z(4, dtype=torch.int64)
        self.obs_ring = z(N, H, L.OBS_FRAME)
        self.priv_ring = z(N, HC, L.PRIV_FRAME)
        self.episode_acc = z(24)
        # outputs
        self.obs = z(N, H * L.OBS_FRAME) if obs_out is None else obs_out
        self.priv_obs = z(N, HC * L.PRIV_FRAME) if priv_out is None else priv_out
        self.rew = z(N)
        self.reset = torch.ones(N, dtype=torch.bool, device=dev)
        self.time_out = z(N, dtype=torch.bool)
        self.extras_time_outs = z(N, dtype=torch.bool)
        self.extras_episode = z(L.NUM_REWARDS)
        # the fused rollout step (hgym_rollout_step) runs the finaliser of step t - 1 concurrently with the env phase of step t: a
        # second rew / reset / time_out set for the alternate steps, and the library's scratch block (zero-filled once)
        self.rew_alt, self.reset_alt, self.time_out_alt = z(N), torch.zeros(N, dtype=torch.bool, device=dev), z(N, dtype=torch.bool)
        self.rollout_scratch = torch.zeros(L.rollout_scratch_bytes(N), dtype=torch.uint8, device=dev)
        self._l0_partial = None      # (2, N, 512) fp32: the actor's first-layer partial sums handed from launch to launch (l0_partial())
        # logging sink of the step finaliser (HgymEnvOut.log_*): per-env running episode return / length, and the statistics block
        self.log_cur = z(2, N)
        self.log_stats = z(L.LOG_STATS)
        self.log_sink = False
        # sim tensors
        self.sim_layout = sim_layout
        if sim_layout == "soa":
            self.root = z(13, N)
            self.dof_pos = z(12, N)
            self.dof_vel = z(12, N)
            self.contact = z(L.NUM_BODIES * 3, N)
            self.rigid = z(L.NUM_BODIES * 13, N)
        elif sim_layout == "aos":
            self.root = z(N, 13)
            self.dof_state = z(N * 12, 2)
            self.contact = z(N * L.NUM_BODIES, 3)
            self.rigid = z(N * L.NUM_BODIES, 13)
        else:
            raise ValueError(sim_layout)
        # constants the reference draws at construction (legged_robot.py:257-302): friction, base mass, origins
        self.f["friction"].fill_(1.0)
        self.f["body_mass"].fill_(15.0)
        self.f["env_origins"].copy_(grid_origins(N).t())
        self.set_initial_root()
        self._structs = None
        # generic LeggedRobot options (set_terrain / set_command_curriculum); absent for XBot-L
        self.terrain_levels = self.terrain_types = self.terrain_origins = None
        self.height_samples = self.height_points = self.height_pose = self.measured_heights = None
        self.command_range_x = None
        self.custom_rew = self.custom_sums = self.custom_acc = self.extras_custom = None     # set_custom_rewards

    # ---- generic options --------------------------------------------------------
    def set_terrain(self, origins, levels, types, env_length, curriculum, height_samples=None, height_points=None, border_size=0.0,
                    horizontal_scale=0.1, vertical_scale=0.005):
        """A height-field / trimesh terrain map (humanoid.utils.terrain.Terrain): tile origins (rows, cols, 3), each env's
        level (row) and type (column); optionally the int16 height field and the (P, 3) base-frame sample points of the height
        measurements.  Sets the env origins from it (legged_robot.py:687-697) and fills the option fields of the config."""
        dev, cfg, N = self.device, self.cfg, self.N
        self.terrain_origins = torch.as_tensor(origins, dtype=torch.float32).to(dev).contiguous()
        self.terrain_levels = torch.as_tensor(levels, dtype=torch.int64).to(dev).contiguous().clone()
        self.terrain_types = torch.as_tensor(types, dtype=torch.int64).to(dev).contiguous()
        assert self.terrain_origins.dim() == 3 and self.terrain_levels.shape == (N,) and self.terrain_types.shape == (N,)
        cfg.custom_origins = 1
        cfg.terrain_curriculum = int(bool(curriculum))
        cfg.terrain_rows, cfg.terrain_cols = int(self.terrain_origins.shape[0]), int(self.terrain_origins.shape[1])
        cfg.terrain_env_length = float(env_length)
        self.f["env_origins"].copy_(self.terrain_origins[self.terrain_levels, self.terrain_types].t())
        self.set_initial_root()
        if height_samples is not None:
            self.height_samples = torch.as_tensor(height_samples, dtype=torch.int16).to(dev).contiguous()
            self.height_points = torch.as_tensor(height_points, dtype=torch.float32).to(dev).contiguous()
            P = int(self.height_points.shape[0])
            self.height_pose = torch.zeros(N, 7, device=dev)
            self.measured_heights = torch.zeros(N, P, device=dev)
            cfg.num_height_points = P
            cfg.height_rows, cfg.height_cols = int(self.height_samples.shape[0]), int(self.height_samples.shape[1])
            cfg.terrain_border, cfg.terrain_hscale, cfg.terrain_vscale = float(border_size), float(horizontal_scale), float(vertical_scale)

    def set_command_curriculum(self, lin_vel_x, max_curriculum):
        """cfg.commands.curriculum: the lin_vel_x range becomes device-resident ([lo, hi] doubles) and widens as the policy tracks."""
        self.command_range_x = torch.tensor([float(lin_vel_x[0]), float(lin_vel_x[1])], dtype=torch.float64, device=self.device)
        self.cfg.command_curriculum = 1
        self.cfg.max_curriculum = float(max_curriculum)

    def set_custom_rewards(self, positions):
        """User-defined reward terms (HgymEnvConfig.num_custom_rewards): positions[j] = how many built-in terms precede custom term j
        in the alphabetical order the reference sums in.  Allocates the (K, N) term / episode-sum buffers the caller fills between
        hgym_env_step_begin and hgym_env_step_end."""
        K = len(positions)
        if K > L.MAX_CUSTOM_REWARDS:
            raise NotImplementedError("at most %d user-defined reward terms" % L.MAX_CUSTOM_REWARDS)
        self.cfg.num_custom_rewards = K
        for j, p in enumerate(positions):
            self.cfg.custom_reward_pos[j] = int(p)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
        self.custom_rew, self.custom_sums, self.custom_acc, self.extras_custom = z(max(K, 1), self.N), z(max(K, 1), self.N), z(max(K, 1)), z(max(K, 1))

    # ---- views with the reference's shapes -------------------------------------------------------
    def view(self, name):
        """(N, C) view of a state field (transposed, non-contiguous)."""
        return self.f[name].t()

    def root_view(self):
        return self.root.t() if self.sim_layout == "soa" else self.root

    def dof_pos_view(self):
        return self.dof_pos.t() if self.sim_layout == "soa" else self.dof_state.view(self.N, 12, 2)[..., 0]

    def dof_vel_view(self):
        return self.dof_vel.t() if self.sim_layout == "soa" else self.dof_state.view(self.N, 12, 2)[..., 1]

    def contact_view(self):
        return (self.contact.t() if self.sim_layout == "soa" else self.contact).reshape(self.N, L.NUM_BODIES, 3)

    def rigid_view(self):
        return (self.rigid.t() if self.sim_layout == "soa" else self.rigid).reshape(self.N, L.NUM_BODIES, 13)

    def set_initial_root(self):
        init = torch.tensor(list(self.cfg.base_init_state), device=self.device).view(1, 13).repeat(self.N, 1)
        init[:, :3] += self.view("env_origins")
        self.root_view().copy_(init)

    def load_sim(self, root, dof_state, contact, rigid):
        """Copy Isaac-Gym-shaped host tensors into whatever layout this object uses."""
        N = self.N
        dev = self.device
        self.root_view().copy_(root.to(dev))
        d = dof_state.to(dev).view(N, 12, 2)
        self.dof_pos_view().copy_(d[..., 0])
        self.dof_vel_view().copy_(d[..., 1])
        if self.sim_layout == "soa":
            self.contact.copy_(contact.to(dev).view(N, -1).t())
            self.rigid.copy_(rigid.to(dev).view(N, -1).t())
        else:
            self.contact.copy_(contact.to(dev).view_as(self.contact))
            self.rigid.copy_(rigid.to(dev).view_as(self.rigid))

    # ---- C structs ---------------------------------------------------------------------------------
    def _strided(self, t, soa):
        N = self.N
        return L.Strided(L.fptr(t), 1 if soa else t.numel() // N, N if soa else 1)

    def sim_struct(self):
        if self.sim_layout == "soa":
            return L.SimTensors(self._strided(self.root, True), self._strided(self.dof_pos, True),
                                self._strided(self.dof_vel, True), self._strided(self.contact, True),
                                self._strided(self.rigid, True))
        dof = self.dof_state
        pos = L.Strided(L.fptr(dof), 24, 2)
        vel = L.Strided(C.cast(dof.data_ptr() + 4, L.c_float_p), 24, 2)
        return L.SimTensors(self._strided(self.root, False), pos, vel, self._strided(self.contact, False),
                            self._strided(self.rigid, False))

    def state_struct(self):
        st = L.EnvState()
        st.episode_length = L.i64ptr(self.episode_length)
        st.counters = L.i64ptr(self.counters)
        for name, _ in L.ENV_STATE_FIELDS:
            setattr(st, name, L.fptr(self.f[name]))
        st.obs_ring = L.fptr(self.obs_ring)
        st.priv_ring = L.fptr(self.priv_ring)
        st.episode_acc = L.fptr(self.episode_acc)
        if self.terrain_levels is not None:
            st.terrain_levels, st.terrain_types = L.i64ptr(self.terrain_levels), L.i64ptr(self.terrain_types)
            st.terrain_origins = L.fptr(self.terrain_origins)
        if self.height_samples is not None:
            st.height_samples = C.cast(self.height_samples.data_ptr(), C.POINTER(C.c_int16))
            st.height_points, st.height_pose = L.fptr(self.height_points), L.fptr(self.height_pose)
            st.measured_heights = L.fptr(self.measured_heights)
        if self.command_range_x is not None:
            st.command_range_x = C.cast(self.command_range_x.data_ptr(), L.c_f64_p)
        if self.custom_rew is not None:
            st.custom_rew, st.custom_sums, st.custom_acc = L.fptr(self.custom_rew), L.fptr(self.custom_sums), L.fptr(self.custom_acc)
        return st

    def out_struct(self, obs=None, priv=None, sink=None, defer_finalize=False, alt=False):
        """sink: optional dict(values, rewards, dones, step, gamma) -- HgymEnvOut's transition sink (caller keeps the tensors alive);
        with values = None and a `time_outs` slot it is the deferred-values kind (HgymEnvOut.t_time_outs).
        defer_finalize: the env-step call does not launch the step finaliser (see HgymEnvOut.defer_finalize).
        alt: the alternate rew / reset / time_out set (fused rollout step)."""
        rew, reset, time_out = (self.rew_alt, self.reset_alt, self.time_out_alt) if alt else (self.rew, self.reset, self.time_out)
        o = L.EnvOut(L.fptr(self.obs if obs is None else obs), L.fptr(self.priv_obs if priv is None else priv),
                     L.fptr(rew), L.u8ptr(reset), L.u8ptr(time_out), L.u8ptr(self.extras_time_outs),
                     L.fptr(self.extras_episode))
        if sink is not None:
            o.t_values, o.t_rewards = L.fptr(sink["values"]), L.fptr(sink["rewards"])
            if sink["values"] is None:
                o.t_time_outs = L.u8ptr(sink["time_outs"])
            o.t_dones, o.t_step, o.t_gamma = L.u8ptr(sink["dones"]), L.i64ptr(sink.get("step")), float(sink["gamma"])
        o.defer_finalize = 1 if defer_finalize else 0
        if self.log_sink:
            o.log_cur, o.log_stats = L.fptr(self.log_cur), L.fptr(self.log_stats)
        if self.extras_custom is not None:
            o.extras_custom = L.fptr(self.extras_custom)
        return o

    def l0_partial(self, k):
        """Buffer k (0 / 1) of the actor's first-layer partial pre-activations carried from one fused rollout launch to the next
        (HgymEnvOut.l0_ahead / l0_ready): (N, 512) fp32, allocated on first use."""
        if self._l0_partial is None:
            self._l0_partial = torch.zeros(2, self.N, 512, dtype=torch.float32, device=self.device)
        return self._l0_partial[k]

    @staticmethod
    def noise_struct(u_delay=None, z_act=None, u_cmd=None, u_dof=None, u_push=None, z_obs=None, u_xy=None, r_level=None):
        """Row-major (N,k) fp32 tables (or None -> internal Philox); r_level (N,) int64.  The caller keeps the tensors alive."""
        for t in (u_delay, z_act, u_cmd, u_dof, u_push, z_obs, u_xy):
            assert t is None or (t.is_contiguous() and t.dtype == torch.float32)
        assert r_level is None or (r_level.is_contiguous() and r_level.dtype == torch.int64)
        return L.EnvNoise(L.fptr(u_delay), L.fptr(z_act), L.fptr(u_cmd), L.fptr(u_dof), L.fptr(u_push), L.fptr(z_obs), L.fptr(u_xy),
                          L.i64ptr(r_level))
