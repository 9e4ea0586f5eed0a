"""Quaternion / angle helpers of the reference's `utils/math_utils.py:40-116` that the env layer uses on the host."""
from typing import Tuple

import numpy as np
import torch
from torch import Tensor

from .isaac_torch_utils import quat_apply, normalize, quat_mul


def _yaw_only(quat, sign):
    q = quat.clone().view(-1, 4)
    q[:, :2] = 0.
    if sign < 0:
        q[:, 2] = -1 * q[:, 2]
    return normalize(q)


def quat_apply_yaw(quat, vec):
    """Rotate `vec` by the yaw component of `quat` only (`math_utils.py:40-44`)."""
    return quat_apply(_yaw_only(quat, +1), vec)


def quat_apply_yaw_inverse(quat, vec):
    return quat_apply(_yaw_only(quat, -1), vec)


def wrap_to_pi(angles):
    """In place, like the reference (`math_utils.py:55-58`): the caller's tensor is modified."""
    angles %= 2 * np.pi
    angles -= 2 * np.pi * (angles > np.pi)
    return angles


def torch_rand_sqrt_float(lower, upper, shape, device):
    # type: (float, float, Tuple[int, int], str) -> Tensor
    r = 2 * torch.rand(*shape, device=device) - 1
    r = torch.where(r < 0., -torch.sqrt(-r), torch.sqrt(r))
    return (upper - lower) * (r + 1.) / 2. + lower


def ypr_to_quat(yaw: Tensor, pitch: Tensor, roll: Tensor) -> Tensor:
    """Yaw (Z), pitch (Y), roll (X) → quaternion (x, y, z, w) = q_roll * q_pitch * q_yaw (`math_utils.py:86-116`)."""
    yaw, pitch, roll = yaw.float() * 0.5, pitch.float() * 0.5, roll.float() * 0.5
    z = torch.zeros_like(yaw)
    q_roll = torch.stack([torch.sin(roll), z, z, torch.cos(roll)], dim=-1)
    q_pitch = torch.stack([z, torch.sin(pitch), z, torch.cos(pitch)], dim=-1)
    q_yaw = torch.stack([z, z, torch.sin(yaw), torch.cos(yaw)], dim=-1)
    return quat_mul(quat_mul(q_roll, q_pitch), q_yaw)
