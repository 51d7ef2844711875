"""Records tests/golden/sim2sim_trace.npz by running the UNMODIFIED reference `scripts/sim2sim.py:run_mujoco` in this
container (BASELINE.json configs[0]: sim2sim, 1 env, CPU policy inference).

MuJoCo is not installed, so the three simulator calls the reference makes (`MjModel.from_xml_path`, `MjData`, `mj_step`)
and the viewer are stood in for by a tiny synthetic single-robot "simulator": joint state integrates the commanded torques
(unit inertia, light damping), the base orientation does a seeded random walk.  Everything between the simulator state
and the torques -- get_obs, the 47-float frame, the 15-frame history, the TorchScript policy call on the shipped
logs/XBot_ppo/exported/policies/policy_example.pt, action clip, PD law, torque clamp -- is the reference's own code.

Recorded per 1 ms step: q, dq (12), quat xyzw, omega as get_obs returned them, and the torques written to data.ctrl;
per policy step: the (1,705) policy input and the clipped action.

    python tests/golden/gen_sim2sim_fixture.py          (needs /root/reference; not run on the GPU box)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H   # noqa: E402

STEPS = 600        # 0.6 s of 1 kHz simulation = 60 policy steps (> frame_stack, so the history fills and rolls)


class _Sensor:
    def __init__(self, data):
        self.data = data


class FakeData:
    def __init__(self, model):
        self.rng = np.random.RandomState(2024)
        self.qpos = np.zeros(19)
        self.qpos[2] = 0.95
        self.qpos[3] = 1.0                      # wxyz
        self.qpos[7:] = self.rng.uniform(-0.1, 0.1, 12)
        self.qvel = np.zeros(18)
        self.ctrl = np.zeros(12)
        self._quat = np.array([1.0, 0.0, 0.0, 0.0])      # wxyz sensor reading
        self._omega = np.zeros(3)

    def sensor(self, name):
        return _Sensor(self._quat if name == "orientation" else self._omega)


class FakeModel:
    class opt:
        timestep = 0.001

    @staticmethod
    def from_xml_path(path):
        return FakeModel()


def fake_step(model, data):
    dt = model.opt.timestep
    tau = np.asarray(data.ctrl, dtype=np.double)
    data.qvel[6:] += dt * (tau - 2.0 * data.qvel[6:])
    data.qpos[7:] += dt * data.qvel[6:]
    data.qvel[:3] = 0.3 * data.rng.standard_normal(3)
    data._omega = 0.3 * data.rng.standard_normal(3)
    q = data._quat + 0.01 * data.rng.standard_normal(4)
    data._quat = q / np.linalg.norm(q)


def main():
    H.load_reference()
    mj = types.ModuleType("mujoco")
    mj.MjModel, mj.MjData, mj.mj_step = FakeModel, FakeData, fake_step
    mv = types.ModuleType("mujoco_viewer")

    class _Viewer:
        def __init__(self, *a):
            pass

        def render(self):
            pass

        def close(self):
            pass
    mv.MujocoViewer = _Viewer
    sys.modules["mujoco"], sys.modules["mujoco_viewer"] = mj, mv
    import importlib
    S = importlib.import_module("humanoid.scripts.sim2sim")
    from humanoid.envs import XBotLCfg

    class Sim2simCfg(XBotLCfg):             # reference scripts/sim2sim.py:176-190, shorter duration
        class sim_config:
            mujoco_model_path = "unused.xml"
            sim_duration = STEPS * 0.001
            dt = 0.001
            decimation = 10

        class robot_config:
            kps = np.array([200, 200, 350, 350, 15, 15, 200, 200, 350, 350, 15, 15], dtype=np.double)
            kds = np.array([10] * 12, dtype=np.double)
            tau_limit = 200. * np.ones(12, dtype=np.double)

    policy = torch.jit.load(os.path.join(H.REFERENCE_ROOT, "logs/XBot_ppo/exported/policies/policy_example.pt"))
    rec = dict(q=[], dq=[], quat=[], omega=[], tau=[], policy_inputs=[], actions=[])

    def policy_rec(x):
        y = policy(x)
        rec["policy_inputs"].append(x[0].numpy().copy())
        return y

    real_get_obs = S.get_obs

    def get_obs_rec(data):
        out = real_get_obs(data)
        q, dq, quat, v, omega, gvec = out
        rec["q"].append(q[-12:].copy()); rec["dq"].append(dq[-12:].copy())
        rec["quat"].append(quat.copy()); rec["omega"].append(omega.copy())
        return out
    S.get_obs = get_obs_rec

    class CtrlTap(FakeData):
        def __setattr__(self, k, v):
            if k == "ctrl" and "ctrl" in self.__dict__:
                rec["tau"].append(np.asarray(v, dtype=np.double).copy())
            object.__setattr__(self, k, v)
    mj.MjData = CtrlTap
    S.run_mujoco(policy_rec, Sim2simCfg())
    # the clipped action of every policy step = target_q / action_scale is not observable directly; recompute from the
    # recorded inputs with the same TorchScript module + the reference's clip (:148-149)
    for x in rec["policy_inputs"]:
        a = policy(torch.tensor(x[None]))[0].detach().numpy().astype(np.double)
        rec["actions"].append(np.clip(a, -18.0, 18.0))
    out = {k: np.stack(v) for k, v in rec.items()}
    assert out["q"].shape == (STEPS, 12) and out["tau"].shape == (STEPS, 12) and out["policy_inputs"].shape == (STEPS // 10, 705)
    np.savez_compressed(os.path.join(HERE, "sim2sim_trace.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
