"""The isaacgym.torch_utils restatements in the oracle (quat_rotate_inverse, quat_apply, get_euler_xyz) against
scipy's Rotation -- the library the reference itself uses for the same quantities in scripts/sim2sim.py:76-79 --
and against the reference's own euler restatement scripts/sim2sim.py:48-68 (formula restated here, the script
cannot be imported without mujoco).  Isaac Gym Preview 4 is absent, so this is the strongest pin available
for that boundary ("unpinned" in the strict sense)."""
import math

import numpy as np
import torch
from scipy.spatial.transform import Rotation

from oracle import xbot_env_oracle as O


def _rand_quats(n, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(n, 4, generator=g, dtype=torch.float64)
    return q / q.norm(dim=1, keepdim=True)


def test_quat_rotate_inverse_and_apply_match_scipy():
    q = _rand_quats(512, 0)
    v = torch.randn(512, 3, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    R = Rotation.from_quat(q.numpy())                      # scipy is xyzw like Isaac Gym
    np.testing.assert_allclose(O.quat_rotate_inverse(q, v).numpy(), R.apply(v.numpy(), inverse=True), atol=1e-12)
    np.testing.assert_allclose(O.quat_apply(q, v).numpy(), R.apply(v.numpy()), atol=1e-12)


def test_euler_xyz_wrapped_matches_sim2sim_formula_and_scipy():
    q = _rand_quats(512, 2)
    e = O.euler_xyz_wrapped(q).numpy()
    x, y, z, w = (q[:, i].numpy() for i in range(4))
    # scripts/sim2sim.py:48-68 quaternion_to_euler_array
    roll = np.arctan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y))
    pitch = np.arcsin(np.clip(2 * (w * y - z * x), -1, 1))
    yaw = np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))
    np.testing.assert_allclose(e, np.stack([roll, pitch, yaw], 1), atol=1e-9)
    # the same rotation: extrinsic xyz euler angles rebuild the quaternion (up to sign)
    R2 = Rotation.from_euler("xyz", e)
    q2 = R2.as_quat()
    sgn = np.sign((q2 * q.numpy()).sum(1, keepdims=True))
    np.testing.assert_allclose(q2 * sgn, q.numpy(), atol=1e-9)
    assert (e > -math.pi - 1e-12).all() and (e <= math.pi + 1e-12).all()


def test_fp32_path_close_to_fp64():
    q = _rand_quats(256, 3)
    v = torch.randn(256, 3, generator=torch.Generator().manual_seed(4), dtype=torch.float64)
    a = O.quat_rotate_inverse(q.float(), v.float()).double()
    b = O.quat_rotate_inverse(q, v)
    assert (a - b).abs().max() < 5e-6
