"""Small tensor helpers with the reference's names (utils/math.py:39-57).  The quaternion primitives that the
reference pulls from isaacgym.torch_utils are stated here as plain xyzw math."""
import numpy as np
import torch

__all__ = ["quat_apply", "normalize", "quat_apply_yaw", "wrap_to_pi", "torch_rand_sqrt_float", "torch_rand_float"]


def normalize(x, eps=1e-9):
    return x / x.norm(p=2, dim=-1).clamp(min=eps).unsqueeze(-1)


def quat_apply(q, v):
    shape = v.shape
    q, v = q.reshape(-1, 4), v.reshape(-1, 3)
    u = q[:, :3]
    t = torch.cross(u, v, dim=-1) * 2
    return (v + q[:, 3:] * t + torch.cross(u, t, dim=-1)).view(shape)


def quat_apply_yaw(quat, vec):
    yaw_only = quat.clone().view(-1, 4)
    yaw_only[:, :2] = 0.0
    return quat_apply(normalize(yaw_only), vec)


def wrap_to_pi(angles):
    """In place, like the reference: fold to (-pi, pi] via a python-style modulo."""
    angles %= 2 * np.pi
    angles -= 2 * np.pi * (angles > np.pi)
    return angles


def torch_rand_float(lower, upper, shape, device):
    return (upper - lower) * torch.rand(*shape, device=device) + lower


def torch_rand_sqrt_float(lower, upper, shape, device):
    r = 2 * torch.rand(*shape, device=device) - 1
    r = torch.where(r < 0.0, -torch.sqrt(-r), torch.sqrt(r))
    return (upper - lower) * ((r + 1.0) / 2.0) + lower
