"""Host-side samplers (`leibnizgym_amd.envs.trifinger.sample`, counterparts of the reference's sample.py)
against the golden vectors of the reference's functions: as functions of the draws they consume, and as streams from
torch's global generator (same seeds as tests/golden/make_golden.py, so the draw ORDER is pinned too)."""
import os

import numpy as np
import torch

from leibnizgym_amd.envs.trifinger import sample as sm

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "samplers.npz"))
T = lambda k: torch.from_numpy(G[k])  # noqa: E731
N = 64


def close(a, b, atol=2e-6, rtol=1e-5):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, atol=atol, rtol=rtol), float((a - b).abs().max())


def test_functions_of_the_draws():
    x, y = sm.xy_from_uniform(T("xy_u_radius"), T("xy_u_theta"), float(G["xy_rmax"]))
    close(x, T("xy_x")), close(y, T("xy_y"))
    close(sm.z_from_uniform(T("z_u"), 0.0325, 0.1), T("z_d3"))
    close(sm.z_from_uniform(T("z_u"), 0.05629165124598851, 0.1), T("z_d4"))
    close(sm.orientation_from_normals(T("ori_normals")), T("ori_quat"))
    close(sm.yaw_orientation_from_uniform(T("yaw_u")), T("yaw_quat"))
    close(sm.angular_vel_from_normals(T("angvel_axis_normals"), T("angvel_mag_normal"), 0.5), T("angvel"))
    close(sm.default_orientation(4, "cpu"), T("default_quat"), atol=0.0, rtol=0.0)


def test_streams_draw_in_the_reference_order():
    torch.manual_seed(3003)
    x, y = sm.random_xy(N, float(G["xy_rmax"]), "cpu")
    close(x, T("xy_x")), close(y, T("xy_y"))
    torch.manual_seed(3004)
    close(sm.random_z(N, 0.0325, 0.1, "cpu"), T("z_d3"))
    torch.manual_seed(3005)
    close(sm.random_orientation(N, "cpu"), T("ori_quat"))
    torch.manual_seed(3006)
    close(sm.random_yaw_orientation(N, "cpu"), T("yaw_quat"))
    torch.manual_seed(3007)
    close(sm.random_angular_vel(N, "cpu", 0.5), T("angvel"))


def test_distributions():
    torch.manual_seed(0)
    x, y = sm.random_xy(20000, 0.1387, "cpu")
    r2 = (x * x + y * y) / 0.1387 ** 2
    assert float(r2.max()) <= 1.0 + 1e-6 and abs(float(r2.mean()) - 0.5) < 0.01      # uniform in area
    q = sm.random_orientation(20000, "cpu")
    assert torch.allclose(q.norm(dim=-1), torch.ones(20000), atol=1e-5)
    assert abs(float(q[:, 3].abs().mean()) - 0.4244) < 0.01                          # E|w| = 4/(3 pi) on S^3
