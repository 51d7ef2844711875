"""ctypes binding of include/hgym.h (libhgym_hip.so).

There is deliberately NO fallback: if the HIP library is missing or fails to load, importing this module
raises, and every product class built on it is unusable.  (The CPU oracle under /oracle is test
infrastructure and is never imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HGYM_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libhgym_hip.so"))

NUM_DOF = 12
NUM_BODIES = 13
NUM_REWARDS = 22
OBS_FRAME = 47
PRIV_FRAME = 73
MAX_LAYERS = 8
MAX_CUSTOM_REWARDS = 24
F32, BF16 = 0, 1

c_float_p = C.POINTER(C.c_float)
c_u8_p = C.POINTER(C.c_uint8)
c_i64_p = C.POINTER(C.c_int64)
c_f64_p = C.POINTER(C.c_double)


class EnvConfig(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("frame_stack", C.c_int32), ("c_frame_stack", C.c_int32), ("decimation", C.c_int32),
        ("sim_dt", C.c_float), ("dt", C.c_float),
        ("max_episode_length", C.c_int32), ("resample_steps", C.c_int32), ("push_interval", C.c_int32),
        ("push_robots", C.c_int32), ("add_noise", C.c_int32), ("heading_command", C.c_int32), ("use_ref_actions", C.c_int32),
        ("clip_actions", C.c_float), ("clip_obs", C.c_float), ("action_scale", C.c_float),
        ("action_delay", C.c_float), ("action_noise", C.c_float), ("noise_level", C.c_float),
        ("obs_noise", C.c_float * OBS_FRAME),
        ("scale_lin_vel", C.c_float), ("scale_ang_vel", C.c_float), ("scale_dof_pos", C.c_float),
        ("scale_dof_vel", C.c_float), ("scale_quat", C.c_float),
        ("cmd_x_lo", C.c_float), ("cmd_x_span", C.c_float), ("cmd_y_lo", C.c_float), ("cmd_y_span", C.c_float),
        ("cmd_h_lo", C.c_float), ("cmd_h_span", C.c_float),
        ("dof_reset_lo", C.c_float), ("dof_reset_span", C.c_float),
        ("push_vel_lo", C.c_float), ("push_vel_span", C.c_float), ("push_ang_lo", C.c_float), ("push_ang_span", C.c_float),
        ("p_gains", C.c_float * NUM_DOF), ("d_gains", C.c_float * NUM_DOF), ("torque_limits", C.c_float * NUM_DOF),
        ("default_dof_pos", C.c_float * NUM_DOF), ("dof_lower", C.c_float * NUM_DOF), ("dof_upper", C.c_float * NUM_DOF),
        ("base_init_state", C.c_float * 13),
        ("base_body", C.c_int32), ("feet_bodies", C.c_int32 * 2), ("knee_bodies", C.c_int32 * 2),
        ("reward_scales", C.c_float * NUM_REWARDS), ("only_positive_rewards", C.c_int32),
        ("base_height_target", C.c_float), ("min_dist", C.c_float), ("max_dist", C.c_float),
        ("target_joint_pos_scale", C.c_float), ("target_feet_height", C.c_float), ("cycle_time", C.c_float),
        ("tracking_sigma", C.c_float), ("max_contact_force", C.c_float), ("episode_length_s", C.c_float),
        ("seed", C.c_uint64),
        # generic LeggedRobot options (all zero for XBot-L)
        ("custom_origins", C.c_int32), ("terrain_curriculum", C.c_int32), ("terrain_rows", C.c_int32), ("terrain_cols", C.c_int32),
        ("terrain_env_length", C.c_float), ("num_height_points", C.c_int32), ("height_rows", C.c_int32), ("height_cols", C.c_int32),
        ("terrain_border", C.c_float), ("terrain_hscale", C.c_float), ("terrain_vscale", C.c_float),
        ("cmd_yaw_lo", C.c_float), ("cmd_yaw_span", C.c_float),
        ("command_curriculum", C.c_int32), ("max_curriculum", C.c_float),
        # user-defined reward terms
        ("num_custom_rewards", C.c_int32), ("custom_reward_pos", C.c_int32 * MAX_CUSTOM_REWARDS),
    ]


class Strided(C.Structure):
    _fields_ = [("base", c_float_p), ("env_stride", C.c_int64), ("comp_stride", C.c_int64)]


class SimTensors(C.Structure):
    _fields_ = [("root", Strided), ("dof_pos", Strided), ("dof_vel", Strided), ("contact", Strided), ("rigid", Strided)]


ENV_STATE_FIELDS = [  # (name, components) in header order; all [C][N] fp32
    ("commands", 4), ("actions", 12), ("last_actions", 12), ("last_last_actions", 12), ("last_dof_vel", 12),
    ("last_root_vel", 6), ("torques", 12), ("feet_air_time", 2), ("last_contacts", 2), ("feet_height", 2),
    ("last_feet_z", 2), ("ref_dof_pos", 12), ("push_force", 3), ("push_torque", 3), ("episode_sums", NUM_REWARDS),
    ("base_lin_vel", 3), ("base_ang_vel", 3), ("projected_gravity", 3), ("base_euler", 3), ("friction", 1),
    ("body_mass", 1), ("env_origins", 3),
]


class EnvState(C.Structure):
    _fields_ = ([("episode_length", c_i64_p), ("counters", c_i64_p)] + [(n, c_float_p) for n, _ in ENV_STATE_FIELDS] +
                [("obs_ring", c_float_p), ("priv_ring", c_float_p), ("episode_acc", c_float_p),
                 ("terrain_levels", c_i64_p), ("terrain_types", c_i64_p), ("terrain_origins", c_float_p),
                 ("height_samples", C.POINTER(C.c_int16)), ("height_points", c_float_p), ("height_pose", c_float_p),
                 ("measured_heights", c_float_p), ("command_range_x", c_f64_p),
                 ("custom_rew", c_float_p), ("custom_sums", c_float_p), ("custom_acc", c_float_p)])


class EnvOut(C.Structure):
    _fields_ = [("obs", c_float_p), ("priv_obs", c_float_p), ("rew", c_float_p), ("reset", c_u8_p), ("time_out", c_u8_p),
                ("extras_time_outs", c_u8_p), ("extras_episode", c_float_p),
                ("t_values", c_float_p), ("t_rewards", c_float_p), ("t_dones", c_u8_p), ("t_step", c_i64_p),
                ("t_gamma", C.c_float), ("defer_finalize", C.c_int32), ("log_cur", c_float_p), ("log_stats", c_float_p),
                ("extras_custom", c_float_p), ("obs_ahead", c_float_p), ("priv_ahead", c_float_p), ("obs_older_ready", C.c_int32),
                ("l0_ahead", c_float_p), ("l0_ready", c_float_p), ("obs_bf16_ahead", C.c_void_p), ("ld_obs_bf16_ahead", C.c_int64),
                ("t_time_outs", c_u8_p)]


class EnvNoise(C.Structure):
    _fields_ = [("u_delay", c_float_p), ("z_act", c_float_p), ("u_cmd", c_float_p), ("u_dof", c_float_p),
                ("u_push", c_float_p), ("z_obs", c_float_p), ("u_xy", c_float_p), ("r_level", c_i64_p)]


class NetConfig(C.Structure):
    _fields_ = [("num_obs", C.c_int32), ("num_priv", C.c_int32), ("num_actions", C.c_int32),
                ("actor_layers", C.c_int32), ("critic_layers", C.c_int32),
                ("actor_dims", C.c_int32 * (MAX_LAYERS + 1)), ("critic_dims", C.c_int32 * (MAX_LAYERS + 1)),
                ("precision", C.c_int32), ("max_batch", C.c_int32),
                ("aux_layers", C.c_int32), ("aux_dims", C.c_int32 * (MAX_LAYERS + 1)), ("aux_target_offset", C.c_int32)]


class PPOConfig(C.Structure):
    _fields_ = [("clip_param", C.c_float), ("value_loss_coef", C.c_float), ("entropy_coef", C.c_float),
                ("max_grad_norm", C.c_float), ("desired_kl", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
                ("lr_min", C.c_double), ("lr_max", C.c_double), ("adaptive_lr", C.c_int32), ("world_size", C.c_int32),
                ("aux_coef", C.c_float), ("grad_norm_ready", C.c_int32)]


class Net(C.Structure):
    _fields_ = [("params", c_float_p), ("grads", c_float_p), ("adam_m", c_float_p), ("adam_v", c_float_p),
                ("opt_state", c_f64_p), ("workspace", C.c_void_p)]


class Comm(C.Structure):
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("data", c_float_p * 8), ("flags", C.POINTER(C.c_uint32) * 8),
                ("count", C.c_int64), ("status", c_i64_p), ("wait_ticks", C.c_int64), ("aux", c_f64_p * 8)]


class Batch(C.Structure):
    _fields_ = [("obs", c_float_p), ("priv", c_float_p), ("actions", c_float_p), ("values", c_float_p),
                ("advantages", c_float_p), ("returns", c_float_p), ("logp", c_float_p), ("mu", c_float_p),
                ("sigma", c_float_p), ("idx", c_i64_p), ("B", C.c_int32),
                ("obs_bf16", C.c_void_p), ("priv_bf16", C.c_void_p), ("num_rows", C.c_int64)]


class ObsShadow(C.Structure):
    """HgymObsShadow: where a policy launch leaves the bf16 of the observation rows it reads (row-major, ld in elements)."""
    _fields_ = [("obs", C.c_void_p), ("ld_obs", C.c_int64), ("priv", C.c_void_p), ("ld_priv", C.c_int64)]


STRUCTS = dict(HgymEnvConfig=EnvConfig, HgymStrided=Strided, HgymSimTensors=SimTensors, HgymEnvState=EnvState,
               HgymEnvOut=EnvOut, HgymEnvNoise=EnvNoise, HgymNetConfig=NetConfig, HgymPPOConfig=PPOConfig,
               HgymNet=Net, HgymBatch=Batch, HgymObsShadow=ObsShadow, HgymComm=Comm)

# every symbol include/hgym.h declares: name -> (restype, argtypes)
_P = C.POINTER
SYMBOLS = {
    "hgym_comm_alloc": (C.c_int32, [C.c_int64, C.POINTER(C.c_void_p)]),
    "hgym_comm_free": (C.c_int32, [C.c_void_p]),
    "hgym_comm_ipc_export": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "hgym_comm_ipc_open": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "hgym_comm_ipc_close": (C.c_int32, [C.c_void_p]),
    "hgym_comm_allreduce": (C.c_int32, [_P(Comm), C.c_uint32, C.c_void_p]),
    "hgym_comm_status": (C.c_int32, [_P(Comm), c_i64_p, C.c_void_p]),
    "hgym_comm_sum64": (C.c_int32, [_P(Comm), c_f64_p, C.c_int32, C.c_void_p]),
    "hgym_version": (C.c_int32, []),
    "hgym_last_error": (C.c_char_p, []),
    "hgym_device_cus": (C.c_int32, []),
    "hgym_sizeof": (C.c_int64, [C.c_char_p]),
    "hgym_env_config_default": (C.c_int32, [_P(EnvConfig), C.c_int32]),
    "hgym_env_prime": (C.c_int32, [_P(EnvConfig), _P(SimTensors), _P(EnvState), _P(EnvOut), _P(EnvNoise), C.c_void_p]),
    "hgym_env_reset_all": (C.c_int32, [_P(EnvConfig), _P(SimTensors), _P(EnvState), _P(EnvOut), _P(EnvNoise), C.c_void_p]),
    "hgym_pre_physics": (C.c_int32, [_P(EnvConfig), _P(EnvState), c_float_p, _P(EnvNoise), C.c_void_p]),
    "hgym_pd_torques": (C.c_int32, [_P(EnvConfig), _P(SimTensors), _P(EnvState), C.c_void_p]),
    "hgym_synth_physics": (C.c_int32, [_P(EnvConfig), _P(SimTensors), _P(EnvState), C.c_void_p]),
    "hgym_post_physics": (C.c_int32, [_P(EnvConfig), _P(SimTensors), _P(EnvState), _P(EnvOut), _P(EnvNoise), C.c_void_p]),
    "hgym_env_step_synth": (C.c_int32, [_P(EnvConfig), _P(SimTensors), _P(EnvState), _P(EnvOut), c_float_p, C.c_void_p]),
    "hgym_env_finalize": (C.c_int32, [_P(EnvConfig), _P(EnvState), _P(EnvOut), C.c_void_p]),
    "hgym_env_step_begin": (C.c_int32, [_P(EnvConfig), _P(SimTensors), _P(EnvState), _P(EnvOut), _P(EnvNoise), c_float_p, C.c_void_p]),
    "hgym_env_step_end": (C.c_int32, [_P(EnvConfig), _P(SimTensors), _P(EnvState), _P(EnvOut), _P(EnvNoise), C.c_void_p]),
    "hgym_measure_heights": (C.c_int32, [_P(EnvConfig), _P(EnvState), C.c_void_p]),
    "hgym_store_step": (C.c_int32, [C.c_int32, c_float_p, c_float_p, c_u8_p, c_u8_p, C.c_float, c_float_p, c_u8_p, C.c_void_p]),
    "hgym_randperm": (C.c_int32, [C.c_int64, C.c_uint64, C.c_uint64, c_i64_p, C.c_void_p]),
    "hgym_randperm_dev": (C.c_int32, [C.c_int64, C.c_uint64, c_i64_p, c_i64_p, C.c_void_p]),
    "hgym_gae": (C.c_int32, [C.c_int32, C.c_int32, c_float_p, c_float_p, c_u8_p, c_float_p, C.c_float, C.c_float,
                             c_float_p, c_float_p, c_f64_p, C.c_void_p]),
    "hgym_adv_normalize": (C.c_int32, [C.c_int64, c_float_p, c_f64_p, C.c_void_p]),
    "hgym_gae_bootstrap": (C.c_int32, [C.c_int32, C.c_int32, c_float_p, c_float_p, c_u8_p, c_u8_p, c_float_p, C.c_float, C.c_float,
                                       c_float_p, c_float_p, c_f64_p, C.c_void_p]),
    "hgym_critic_values": (C.c_int32, [_P(NetConfig), _P(Net), C.c_int64, c_float_p, c_float_p, _P(ObsShadow), C.c_void_p]),
    "hgym_net_param_count": (C.c_int64, [_P(NetConfig)]),
    "hgym_net_workspace_bytes": (C.c_int64, [_P(NetConfig)]),
    "hgym_net_sync_shadow": (C.c_int32, [_P(NetConfig), _P(Net), C.c_void_p]),
    "hgym_mlp_forward": (C.c_int32, [_P(NetConfig), _P(Net), C.c_int32, C.c_int32, c_float_p, C.c_int64, c_float_p, C.c_void_p]),
    "hgym_net_shadow_ld": (C.c_int64, [_P(NetConfig), C.c_int32]),
    "hgym_policy_act": (C.c_int32, [_P(NetConfig), _P(Net), C.c_int32, c_float_p, c_float_p, c_float_p, C.c_uint64, c_i64_p,
                                    c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, _P(ObsShadow), C.c_void_p]),
    "hgym_policy_act_fin": (C.c_int32, [_P(NetConfig), _P(Net), C.c_int32, c_float_p, c_float_p, c_float_p, C.c_uint64, c_i64_p,
                                    c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, _P(EnvConfig), _P(EnvState), _P(EnvOut),
                                    _P(ObsShadow), C.c_void_p]),
    "hgym_rollout_begin": (C.c_int32, [_P(EnvState), c_i64_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "hgym_rollout_step": (C.c_int32, [_P(NetConfig), _P(Net), _P(EnvConfig), _P(SimTensors), _P(EnvState), _P(EnvOut), _P(EnvOut), c_float_p,
                                      c_float_p, C.c_uint64, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, C.c_void_p, C.c_int32,
                                      _P(ObsShadow), C.c_void_p]),
    "hgym_rollout_end": (C.c_int32, [_P(EnvConfig), _P(EnvState), _P(EnvOut), C.c_void_p, C.c_int32, C.c_void_p]),
    "hgym_ppo_grad": (C.c_int32, [_P(NetConfig), _P(PPOConfig), _P(Net), _P(Batch), C.c_void_p]),
    "hgym_ppo_grad_part": (C.c_int32, [_P(NetConfig), _P(PPOConfig), _P(Net), _P(Batch), C.c_int32, C.c_void_p]),
    "hgym_net_param_offset": (C.c_int64, [_P(NetConfig), C.c_int32]),
    "hgym_ppo_apply": (C.c_int32, [_P(NetConfig), _P(PPOConfig), _P(Net), C.c_void_p]),
    "hgym_prof_enable": (C.c_int32, [C.c_int32]),
    "hgym_prof_phase_buffer": (C.c_int32, [C.c_void_p, C.c_int64]),
    "hgym_prof_summary": (C.c_int32, [C.c_int32, _P(C.c_int64), _P(C.c_double), _P(C.c_double)]),
}
PROF_GEMM, PROF_ENV_STEP, PROF_GAE, PROF_LOSS, PROF_MLP_FWD, PROF_MLP_BWD, PROF_DW, PROF_REDUCE, PROF_APPLY, PROF_POLICY, PROF_ROLLOUT, PROF_COMM = range(12)
ROLLOUT_SCRATCH_HEADER_BYTES = 512
ROLLOUT_DRAW_BYTES_PER_ENV = 1000


def rollout_scratch_bytes(num_envs):
    """HGYM_ROLLOUT_SCRATCH_BYTES(num_envs) of include/hgym.h."""
    return ROLLOUT_SCRATCH_HEADER_BYTES + ROLLOUT_DRAW_BYTES_PER_ENV * int(num_envs)

LOG_STATS = 256


def prof_summary(cls):
    n, ms, work = C.c_int64(), C.c_double(), C.c_double()
    rc = lib.hgym_prof_summary(cls, C.byref(n), C.byref(ms), C.byref(work))
    if rc != 0:
        raise HgymError("hgym_prof_summary failed: %s" % lib.hgym_last_error().decode())
    return n.value, ms.value, work.value


class HgymError(RuntimeError):
    pass


def _load(path):
    if not os.path.exists(path):
        raise ImportError(
            "libhgym_hip.so not found at %s -- build it with `python humanoid-gym_amd/build.py` "
            "(there is no CPU fallback for the hot path)" % path)
    lib = C.CDLL(path)
    missing = []
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise ImportError("libhgym_hip.so lacks symbols declared in include/hgym.h: %s" % ", ".join(missing))
    return lib


lib = _load(LIB_PATH)


def check(rc, what=""):
    if rc != 0:
        raise HgymError("%s failed (%d): %s" % (what or "hgym call", rc, lib.hgym_last_error().decode()))


def fptr(t):
    """float* of a torch tensor (or None)."""
    return None if t is None else C.cast(t.data_ptr(), c_float_p)


def u8ptr(t):
    return None if t is None else C.cast(t.data_ptr(), c_u8_p)


def i64ptr(t):
    return None if t is None else C.cast(t.data_ptr(), c_i64_p)


def f64ptr(t):
    return None if t is None else C.cast(t.data_ptr(), c_f64_p)


def gae_stats(n, device):
    """The `stats` buffer of hgym_gae / hgym_gae_bootstrap for n envs: HGYM_GAE_STATS_DOUBLES(n) zero-filled doubles (include/hgym.h)."""
    import torch
    return torch.zeros(4 + 2 * ((int(n) + 15) // 16), dtype=torch.float64, device=device)
