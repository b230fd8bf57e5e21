"""Host-side (torch) mirror of the reference's RobotLearning/omniisaacgymenvs/utils/math.py, without the
scipy CPU round trip (utils/math.py:15-18,29-31 move every call device->CPU->device).  Scalar-first
quaternions.  The engine kernels do not call these; they exist because task code / user scripts written
against the reference import them."""
import torch


def quat_mul(a, b):
    w1, x1, y1, z1 = a.unbind(-1); w2, x2, y2, z2 = b.unbind(-1)
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dim=-1)


def quat_conjugate(a):
    return torch.cat((a[..., :1], -a[..., 1:]), dim=-1)


def rot_matrices_from_quat(orientations, device=None):
    """utils/math.py:7-19"""
    w, x, y, z = orientations.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.view(*orientations.shape[:-1], 3, 3)


def quat_rotate(q, v):
    return torch.einsum("...ij,...j->...i", rot_matrices_from_quat(q), v)


def quat_rotate_inverse(q, v):
    return torch.einsum("...ji,...j->...i", rot_matrices_from_quat(q), v)


def _positive_w(q):
    return torch.where((q[..., 0] < 0.0).unsqueeze(-1), -q, q)


def rotate_orientations(rotations, orientations, device=None):
    """quat of R(rot) R(orient), scalar part >= 0 (utils/math.py:33-57)."""
    return _positive_w(quat_mul(rotations, orientations))


def inverse_rotate_orientations(rotations, orientations, device=None):
    """quat of R(rot)^T R(orient), scalar part >= 0 (utils/math.py:59-83)."""
    return _positive_w(quat_mul(quat_conjugate(rotations), orientations))


def transform_vectors(quaternions, translations, vectors, device=None):
    """R(q) V + t on (N,m,3) point sets (utils/math.py:102-113)."""
    return torch.einsum("nij,nmj->nmi", rot_matrices_from_quat(quaternions), vectors) + translations.unsqueeze(1)


def inverse_transform_vectors(quaternions, translations, vectors, device=None):
    """R(q)^T (V - t) (utils/math.py:115-127)."""
    return torch.einsum("nji,nmj->nmi", rot_matrices_from_quat(quaternions), vectors - translations.unsqueeze(1))


def quat_from_euler_xyz(roll, pitch, yaw):
    cy, sy = torch.cos(yaw * 0.5), torch.sin(yaw * 0.5)
    cr, sr = torch.cos(roll * 0.5), torch.sin(roll * 0.5)
    cp, sp = torch.cos(pitch * 0.5), torch.sin(pitch * 0.5)
    return torch.stack([cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp,
                        cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp], dim=-1)


def rand_quaternions(num_envs, min_roll, max_roll, min_pitch, max_pitch, min_yaw, max_yaw, device, generator=None):
    """utils/math.py:176-193: one torch.rand((k,3)) draw mapped to roll/pitch/yaw."""
    r = torch.rand((num_envs, 3), device=device, generator=generator)
    return quat_from_euler_xyz(min_roll + (max_roll - min_roll) * r[:, 0], min_pitch + (max_pitch - min_pitch) * r[:, 1],
                               min_yaw + (max_yaw - min_yaw) * r[:, 2])


def unscale_transform(x, lower, upper):
    return x * (upper - lower) * 0.5 + (upper + lower) * 0.5
