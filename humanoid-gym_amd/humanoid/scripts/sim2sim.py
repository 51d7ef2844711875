"""sim2sim: run an exported TorchScript actor in a 1 kHz single-robot simulation, policy at 100 Hz, on the CPU
(reference scripts/sim2sim.py:44-193; BASELINE.json configs[0] -- plumbing, no GPU).

Same entry points as the reference script (`cmd`, `quaternion_to_euler_array`, `get_obs`, `pd_control`, `run_mujoco`,
`Sim2simCfg` built in `__main__`, `--load_model` / `--terrain`).  The observation pipeline differs from the training
env's on purpose and is kept as the reference has it: no observation noise, joint positions
WITHOUT the default-pose offset, `omega` and the euler angles unscaled, clip to +-18, 15-frame history oldest first.

The control loop is split from the simulator: `ControlLoop` holds everything between "the simulator produced a
state" and "here are the joint torques" (observation frame, history, policy call, action clip, PD law).  `run_mujoco`
drives it from MuJoCo (imported lazily: the package is optional); `run_replay` drives it from a recorded state trace,
which is how the pipeline is tested against the reference's own loop where MuJoCo is not installed
(tests/test_sim2sim.py, fixture recorded by tests/golden/gen_sim2sim_fixture.py).
"""
import math
import os
import sys
from collections import deque

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from humanoid import LEGGED_GYM_ROOT_DIR  # noqa: E402


class cmd:                      # commanded base velocity (reference :44-47)
    vx = 0.4
    vy = 0.0
    dyaw = 0.0


def quaternion_to_euler_array(quat):
    """xyzw quaternion -> (roll, pitch, yaw) in radians (reference :50-68)."""
    x, y, z, w = quat
    roll = np.arctan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + y * y))
    pitch = np.arcsin(np.clip(2.0 * (w * y - z * x), -1.0, 1.0))
    yaw = np.arctan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
    return np.array([roll, pitch, yaw])


def get_obs(data):
    """(q, dq, quat xyzw, base-frame velocity, angular velocity, projected gravity) from a MuJoCo-shaped data object:
    `qpos`, `qvel`, `sensor('orientation').data` (wxyz), `sensor('angular-velocity').data` (reference :70-80)."""
    from scipy.spatial.transform import Rotation
    q = data.qpos.astype(np.double)
    dq = data.qvel.astype(np.double)
    quat = data.sensor("orientation").data[[1, 2, 3, 0]].astype(np.double)
    rot = Rotation.from_quat(quat)
    v = rot.apply(data.qvel[:3], inverse=True).astype(np.double)
    omega = data.sensor("angular-velocity").data.astype(np.double)
    gvec = rot.apply(np.array([0.0, 0.0, -1.0]), inverse=True).astype(np.double)
    return q, dq, quat, v, omega, gvec


def pd_control(target_q, q, kp, target_dq, dq, kd):
    """Joint torques of the position command (reference :82-85)."""
    return (target_q - q) * kp + (target_dq - dq) * kd


class ControlLoop:
    """Everything the reference's loop body does apart from stepping / rendering the simulator (:116-158)."""

    def __init__(self, policy, cfg):
        self.policy, self.cfg = policy, cfg
        n = cfg.env.num_actions
        self.target_q = np.zeros(n, dtype=np.double)
        self.action = np.zeros(n, dtype=np.double)
        self.history = deque(np.zeros([1, cfg.env.num_single_obs], dtype=np.double) for _ in range(cfg.env.frame_stack))
        self.count_lowlevel = 0
        self.last_policy_input = None

    def frame(self, q, dq, quat, omega):
        """One 47-float observation frame (:126-141)."""
        cfg = self.cfg
        scales = cfg.normalization.obs_scales
        phase = 2 * math.pi * self.count_lowlevel * cfg.sim_config.dt / 0.64      # same evaluation order as the reference
        euler = quaternion_to_euler_array(quat)
        euler[euler > math.pi] -= 2 * math.pi
        f = np.zeros([1, cfg.env.num_single_obs], dtype=np.float32)
        f[0, 0] = math.sin(phase)
        f[0, 1] = math.cos(phase)
        f[0, 2] = cmd.vx * scales.lin_vel
        f[0, 3] = cmd.vy * scales.lin_vel
        f[0, 4] = cmd.dyaw * scales.ang_vel
        f[0, 5:17] = q * scales.dof_pos
        f[0, 17:29] = dq * scales.dof_vel
        f[0, 29:41] = self.action
        f[0, 41:44] = omega
        f[0, 44:47] = euler
        lim = cfg.normalization.clip_observations
        return np.clip(f, -lim, lim)

    def torques(self, q, dq, quat, omega):
        """q, dq: the last num_actions entries of qpos / qvel.  Returns the clamped PD torques for this 1 ms step;
        every `decimation`-th call first queries the policy."""
        cfg = self.cfg
        n = cfg.env.num_actions
        if self.count_lowlevel % cfg.sim_config.decimation == 0:
            self.history.append(self.frame(q, dq, quat, omega))
            self.history.popleft()
            x = np.zeros([1, cfg.env.num_observations], dtype=np.float32)
            w = cfg.env.num_single_obs
            for i, fr in enumerate(self.history):
                x[0, i * w:(i + 1) * w] = fr[0, :]
            self.last_policy_input = x
            self.action[:] = self.policy(torch.tensor(x))[0].detach().numpy()
            lim = cfg.normalization.clip_actions
            self.action = np.clip(self.action, -lim, lim)
            self.target_q = self.action * cfg.control.action_scale
        rc = cfg.robot_config
        tau = pd_control(self.target_q, q, rc.kps, np.zeros(n, dtype=np.double), dq, rc.kds)
        tau = np.clip(tau, -rc.tau_limit, rc.tau_limit)
        self.count_lowlevel += 1
        return tau


def run_mujoco(policy, cfg, render=True):
    """Reference :87-164.  Needs the optional `mujoco` (and, when rendering, `mujoco_viewer`) packages."""
    try:
        import mujoco
    except ImportError as e:
        raise RuntimeError("sim2sim needs the `mujoco` package (reference setup.py: mujoco==2.3.6); "
                           "run_replay() exercises the same control loop from a recorded state trace") from e
    from tqdm import tqdm
    model = mujoco.MjModel.from_xml_path(cfg.sim_config.mujoco_model_path)
    model.opt.timestep = cfg.sim_config.dt
    data = mujoco.MjData(model)
    mujoco.mj_step(model, data)
    viewer = None
    if render:
        import mujoco_viewer
        viewer = mujoco_viewer.MujocoViewer(model, data)
    loop = ControlLoop(policy, cfg)
    n = cfg.env.num_actions
    for _ in tqdm(range(int(cfg.sim_config.sim_duration / cfg.sim_config.dt)), desc="Simulating..."):
        q, dq, quat, v, omega, gvec = get_obs(data)
        data.ctrl = loop.torques(q[-n:], dq[-n:], quat, omega)
        mujoco.mj_step(model, data)
        if viewer is not None:
            viewer.render()
    if viewer is not None:
        viewer.close()


def run_replay(policy, cfg, trace):
    """The control loop over a recorded state trace: `trace` has per 1 ms step `q` (S,12), `dq` (S,12), `quat` (S,4, xyzw)
    and `omega` (S,3).  Returns dict(policy_inputs (P,705), actions (P,12), tau (S,12))."""
    loop = ControlLoop(policy, cfg)
    S = len(trace["q"])
    inputs, actions, taus = [], [], []
    for s in range(S):
        polled = loop.count_lowlevel % cfg.sim_config.decimation == 0
        taus.append(loop.torques(np.asarray(trace["q"][s], np.double), np.asarray(trace["dq"][s], np.double),
                                 np.asarray(trace["quat"][s], np.double), np.asarray(trace["omega"][s], np.double)))
        if polled:
            inputs.append(loop.last_policy_input[0].copy())
            actions.append(loop.action.copy())
    return dict(policy_inputs=np.stack(inputs), actions=np.stack(actions), tau=np.stack(taus))


def mjcf_path(terrain=False, mjcf=None):
    """Where the MuJoCo model of XBot-L is.  The robot assets (URDF / MJCF / meshes: SURVEY.md §2, out of scope) are NOT shipped with
    this repo; they live in the reference checkout under resources/robots/XBot/mjcf/.  Resolution order: the explicit path, the
    HGYM_ROBOT_ASSETS directory (the reference's `resources/robots` or a copy of it), the reference's layout next to this package."""
    name = "XBot-L-terrain.xml" if terrain else "XBot-L.xml"
    if mjcf:
        return mjcf
    root = os.environ.get("HGYM_ROBOT_ASSETS")
    if root:
        return os.path.join(root, "XBot", "mjcf", name)
    return os.path.join(LEGGED_GYM_ROOT_DIR, "resources", "robots", "XBot", "mjcf", name)


def make_cfg(terrain=False, sim_duration=60.0, mjcf=None):
    """The reference's `Sim2simCfg` (:176-190): XBotLCfg + simulator and PD settings."""
    from humanoid.envs import XBotLCfg

    class Sim2simCfg(XBotLCfg):
        class sim_config:
            mujoco_model_path = mjcf_path(terrain, mjcf)
            dt = 0.001
            decimation = 10

        class robot_config:
            kps = np.array([200, 200, 350, 350, 15, 15] * 2, dtype=np.double)
            kds = np.full(12, 10.0, dtype=np.double)
            tau_limit = 200.0 * np.ones(12, dtype=np.double)

    Sim2simCfg.sim_config.sim_duration = sim_duration
    return Sim2simCfg()


if __name__ == "__main__":
    import argparse
    parser = argparse.ArgumentParser(description="Deployment script.")
    parser.add_argument("--load_model", type=str, required=True, help="Run to load from.")
    parser.add_argument("--terrain", action="store_true", help="terrain or plane")
    parser.add_argument("--replay", type=str, default=None, help="npz state trace (q, dq, quat, omega): run without MuJoCo")
    parser.add_argument("--mjcf", type=str, default=None,
                        help="MuJoCo model of XBot-L (default: $HGYM_ROBOT_ASSETS/XBot/mjcf/XBot-L[-terrain].xml, else the reference's "
                             "resources/robots/ layout next to this package; the assets are not shipped here)")
    args = parser.parse_args()
    pol = torch.jit.load(args.load_model)
    if args.replay:
        out = run_replay(pol, make_cfg(args.terrain), dict(np.load(args.replay)))
        print("replayed %d policy steps; |tau| max %.3f" % (len(out["actions"]), float(np.abs(out["tau"]).max())))
    else:
        cfg = make_cfg(args.terrain, mjcf=args.mjcf)
        if not os.path.exists(cfg.sim_config.mujoco_model_path):
            raise SystemExit("sim2sim: MuJoCo model not found at %s\nThe robot assets are not part of this repository: point --mjcf at "
                             "XBot-L.xml, or HGYM_ROBOT_ASSETS at the reference checkout's resources/robots directory "
                             "(roboterax/humanoid-gym), or use --replay <trace.npz> to run the control loop without a simulator."
                             % cfg.sim_config.mujoco_model_path)
        run_mujoco(pol, cfg)
