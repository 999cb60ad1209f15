"""Host-side samplers of object / goal poses with the names and signatures of the reference's
`leibnizgym/envs/trifinger/sample.py` (T11 of SURVEY.md section 8).

The native step draws its resets in-kernel from a counter-based generator (Philox keyed by the global env id); these
functions are the same distributions on the host, drawing from torch's global generator in the reference's order, for
code written against that module (curricula, evaluation scripts).  Each is a thin wrapper over a pure function of the
draws (`*_from_*`), which `tests/test_samplers_host.py` pins to the golden vectors of the reference's functions.
Quaternions are (x, y, z, w)."""
import math
from typing import Tuple

import torch

from ...utils.torch_utils import quaternion_from_euler_xyz


def xy_from_uniform(u_radius: torch.Tensor, u_theta: torch.Tensor, max_com_distance_to_center: float):
    """Uniform point of a disc: radius = R sqrt(u), angle = 2 pi u' (reference sample.py:22-34)."""
    radius = max_com_distance_to_center * torch.sqrt(u_radius)
    theta = 2 * math.pi * u_theta
    return radius * torch.cos(theta), radius * torch.sin(theta)


def z_from_uniform(u: torch.Tensor, min_height: float, max_height: float) -> torch.Tensor:
    return (max_height - min_height) * u + min_height


def orientation_from_normals(n: torch.Tensor) -> torch.Tensor:
    """Uniform rotation: a normalised 4-vector of standard normals (reference sample.py:55-65)."""
    return torch.nn.functional.normalize(n, p=2.0, dim=-1, eps=1e-12)


def angular_vel_from_normals(axis_normals: torch.Tensor, magnitude_normal: torch.Tensor, magnitude_stdev: float):
    axis = axis_normals / torch.norm(axis_normals, p=2, dim=-1).view(-1, 1)
    return (magnitude_normal * magnitude_stdev) * axis


def yaw_orientation_from_uniform(u: torch.Tensor) -> torch.Tensor:
    zero = torch.zeros_like(u)
    return quaternion_from_euler_xyz(zero, zero, 2 * math.pi * u)


# ---- the reference's entry points ------------------------------------------------------------------------------------
def random_xy(num: int, max_com_distance_to_center: float, device: str) -> Tuple[torch.Tensor, torch.Tensor]:
    u_radius = torch.rand(num, dtype=torch.float, device=device)       # radius first, then the angle (sample.py:26,29)
    u_theta = torch.rand(num, dtype=torch.float, device=device)
    return xy_from_uniform(u_radius, u_theta, max_com_distance_to_center)


def random_z(num: int, min_height: float, max_height: float, device: str) -> torch.Tensor:
    return z_from_uniform(torch.rand(num, dtype=torch.float, device=device), min_height, max_height)


def default_orientation(num: int, device: str) -> torch.Tensor:
    quat = torch.zeros((num, 4), dtype=torch.float, device=device)
    quat[..., -1] = 1.0
    return quat


def random_orientation(num: int, device: str) -> torch.Tensor:
    return orientation_from_normals(torch.randn((num, 4), dtype=torch.float, device=device))


def random_angular_vel(num: int, device: str, magnitude_stdev: float) -> torch.Tensor:
    axis = torch.randn((num, 3), dtype=torch.float, device=device)
    magnitude = torch.randn((num, 1), dtype=torch.float, device=device)
    return angular_vel_from_normals(axis, magnitude, magnitude_stdev)


def random_yaw_orientation(num: int, device: str) -> torch.Tensor:
    return yaw_orientation_from_uniform(torch.rand(num, dtype=torch.float, device=device))
