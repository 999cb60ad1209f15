"""Tensor helpers with the names and semantics of the reference's `leibnizgym/utils/torch_utils.py` (the T12 row of
SURVEY.md section 8): range scaling, clamping and the xyzw quaternion algebra the task layer is written in.

Inside the native step the same arithmetic runs in the HIP kernel; these host-side versions exist so that code written
against the reference's utility module (`from leibnizgym.utils.torch_utils import quat_diff_rad`, custom rewards,
evaluation scripts) keeps working when the import root is switched.  They are ordinary batched torch expressions on
whatever device the inputs live on; `tests/test_torch_utils.py` pins them to the golden vectors produced by the
reference's own functions.  Quaternions are (x, y, z, w), last dimension 4.
"""
import torch


def scale_transform(x: torch.Tensor, lower: torch.Tensor, upper: torch.Tensor) -> torch.Tensor:
    """Map [lower, upper] affinely onto [-1, 1] (reference torch_utils.py:18-36)."""
    centre = 0.5 * (lower + upper)
    return 2.0 * (x - centre) / (upper - lower)


def unscale_transform(x: torch.Tensor, lower: torch.Tensor, upper: torch.Tensor) -> torch.Tensor:
    """Inverse of `scale_transform`: [-1, 1] back onto [lower, upper] (reference torch_utils.py:39-57)."""
    centre = 0.5 * (lower + upper)
    return x * (upper - lower) * 0.5 + centre


def saturate(x: torch.Tensor, lower: torch.Tensor, upper: torch.Tensor) -> torch.Tensor:
    """Element-wise clamp with tensor bounds (reference torch_utils.py:60-75)."""
    return torch.minimum(torch.maximum(x, lower), upper)


def quat_mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Hamilton product a (x) b of xyzw quaternions, any leading shape (reference torch_utils.py:83-113).

    Evaluated with the eight-multiplication factorisation, in the same operation order as the device function
    `quat_mul` of csrc/trifinger_hip.hip: `quat_diff_rad` feeds the result to asin near its steep end, where a
    differently rounded product shows up as ~1e-5 rad."""
    assert a.shape == b.shape
    shape = a.shape
    a, b = a.reshape(-1, 4), b.reshape(-1, 4)
    x1, y1, z1, w1 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    x2, y2, z2, w2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    ww = (z1 + x1) * (x2 + y2)
    yy = (w1 - y1) * (w2 + z2)
    zz = (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = 0.5 * (xx + (z1 - x1) * (x2 - y2))
    w = qq - ww + (z1 - y1) * (y2 - z2)
    x = qq - xx + (x1 + w1) * (x2 + w2)
    y = qq - yy + (w1 - x1) * (y2 + z2)
    z = qq - zz + (z1 + y1) * (w2 - x2)
    return torch.stack([x, y, z, w], dim=-1).view(shape)


def quat_conjugate(a: torch.Tensor) -> torch.Tensor:
    """(-x, -y, -z, w) (reference torch_utils.py:116-128)."""
    shape = a.shape
    a = a.reshape(-1, 4)
    return torch.cat([-a[:, :3], a[:, 3:4]], dim=-1).view(shape)


def quat_diff_rad(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Rotation angle between two orientations: 2 asin(min(|vec(a (x) conj(b))|, 1)) (reference torch_utils.py:131-150)."""
    rel = quat_mul(a, quat_conjugate(b))
    return 2.0 * torch.asin(torch.clamp(torch.norm(rel[..., 0:3], p=2, dim=-1), max=1.0))


def quaternion_from_euler_xyz(roll: torch.Tensor, pitch: torch.Tensor, yaw: torch.Tensor) -> torch.Tensor:
    """xyzw quaternion of the intrinsic roll-pitch-yaw rotation R = Rz(yaw) Ry(pitch) Rx(roll) (reference :153-180)."""
    cr, sr = torch.cos(0.5 * roll), torch.sin(0.5 * roll)
    cp, sp = torch.cos(0.5 * pitch), torch.sin(0.5 * pitch)
    cy, sy = torch.cos(0.5 * yaw), torch.sin(0.5 * yaw)
    x = cy * sr * cp - sy * cr * sp
    y = cy * cr * sp + sy * sr * cp
    z = sy * cr * cp - cy * sr * sp
    w = cy * cr * cp + sy * sr * sp
    return torch.stack([x, y, z, w], dim=-1)
