"""`leibnizgym_amd.utils.torch_utils` (host-side counterparts of the reference's utility functions)
against the golden vectors generated from the reference's own functions (tests/golden/math.npz)."""
import os

import numpy as np
import torch

from leibnizgym_amd.utils import torch_utils as tu

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "math.npz"))
T = lambda k: torch.from_numpy(G[k])  # noqa: E731


def close(a, b, atol=2e-6, rtol=1e-5):
    assert a.shape == b.shape and a.dtype == b.dtype
    assert torch.allclose(a, b, atol=atol, rtol=rtol), float((a - b).abs().max())


def test_scaling_and_saturation():
    lo, hi = T("scale_lo"), T("scale_hi")
    close(tu.scale_transform(T("scale_x"), lo, hi), T("scale_y"))
    close(tu.unscale_transform(T("scale_x"), lo, hi), T("unscale_y"))
    close(tu.saturate(T("scale_x"), lo, hi), T("saturate_y"), atol=0.0, rtol=0.0)
    close(tu.unscale_transform(tu.scale_transform(T("scale_x"), lo, hi), lo, hi), T("scale_x"), atol=1e-6)


def test_quaternion_algebra():
    a, b = T("quat_a"), T("quat_b")
    close(tu.quat_mul(a, b), T("quat_mul"))
    close(tu.quat_conjugate(a), T("quat_conj"), atol=0.0, rtol=0.0)
    # the golden set contains identical, antipodal and theta ~ pi pairs: asin is steep there (abs 2e-3)
    close(tu.quat_diff_rad(a, b), T("quat_diff_rad"), atol=2e-3)
    rpy = T("euler_rpy")
    close(tu.quaternion_from_euler_xyz(rpy[:, 0], rpy[:, 1], rpy[:, 2]), T("euler_quat"))
    # shapes other than [N, 4] keep their leading dimensions
    assert tu.quat_mul(a.view(8, 8, 4), b.view(8, 8, 4)).shape == (8, 8, 4)
    ident = torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(64, 4)
    close(tu.quat_mul(a, ident), a)


def test_star_import_gives_the_message_helpers():
    ns = {}
    exec("from leibnizgym_amd.utils import *", ns)
    for name in ("print_info", "print_warn", "print_error", "print_debug", "print_notify", "print_dict", "update_dict"):
        assert callable(ns[name])
