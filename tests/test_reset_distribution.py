"""Distribution tests of the in-kernel reset sampler (Philox4x32-10 keyed by (seed, global env id, reset count)).
The reference draws from torch's global RNG, so only the DISTRIBUTIONS are comparable (sample.py:22-84,
trifinger_env.py:1101-1265): uniform-in-disc xy, uniform z, uniform SO(3) goal orientation, uniform yaw."""
import numpy as np
from scipy import stats

from leibnizgym_amd.engine import TrifingerEngine, make_config

N = 20000
R_MAX = 0.13870835


def sample(backend, **kw):
    lib, device = backend
    eng = TrifingerEngine(make_config(lib, N, seed=2024, command_mode="torque", **kw), device=device, lib=lib)
    eng.reset()
    out = eng.state.cpu().numpy().copy(), eng
    return out


def test_goal_pose_distributions_difficulty4(backend):
    st, eng = sample(backend, task_difficulty=4)
    gx, gy, gz = st[31], st[32], st[33]
    r2 = (gx * gx + gy * gy) / R_MAX ** 2
    assert stats.kstest(r2, "uniform").pvalue > 1e-3                      # radius = r_max sqrt(U)
    theta = (np.arctan2(gy, gx) + 2 * np.pi) % (2 * np.pi)
    assert stats.kstest(theta / (2 * np.pi), "uniform").pvalue > 1e-3
    lo, hi = 0.05629165, 0.1
    assert gz.min() >= lo - 1e-7 and gz.max() <= hi + 1e-7
    assert stats.kstest((gz - lo) / (hi - lo), "uniform").pvalue > 1e-3
    q = st[34:38]
    assert np.abs(np.linalg.norm(q, axis=0) - 1).max() < 1e-6
    # uniform SO(3): each component of a uniform unit quaternion has density ~ (1 - x^2)^(1/2) -> x^2 ~ Beta(1/2, 3/2)
    for k in range(4):
        assert stats.kstest(q[k] ** 2, "beta", args=(0.5, 1.5)).pvalue > 1e-3
    assert abs(np.corrcoef(q)[0, 1]) < 0.03
    eng.close()


def test_object_pose_and_independence(backend):
    st, eng = sample(backend, task_difficulty=1)
    cx, cy = st[18], st[19]
    # the cube has gone through ONE simulate since the reset (env_base.py:336): positions moved by < 1 mm
    r2 = (cx * cx + cy * cy) / R_MAX ** 2
    assert stats.kstest(np.clip(r2, 0, 1), "uniform").pvalue > 1e-4
    yaw = 2 * np.arctan2(st[23], st[24])
    yaw = (yaw + 2 * np.pi) % (2 * np.pi)
    assert stats.kstest(yaw / (2 * np.pi), "uniform").pvalue > 1e-3
    assert np.abs(st[21]).max() < 1e-3 and np.abs(st[22]).max() < 1e-3    # yaw-only rotation
    # object and goal draws use different Philox stream tags: no correlation
    assert abs(np.corrcoef(cx, st[31])[0, 1]) < 0.03
    # difficulty 1: goal on the table, identity orientation (trifinger_env.py:1216-1220)
    assert np.all(st[33] == np.float32(0.0325)) and np.all(st[37] == 1.0) and np.all(st[34:37] == 0.0)
    eng.close()


def test_random_robot_reset_and_reproducibility(backend):
    kw = dict(task_difficulty=1, robot_reset="random", dof_pos_stddev=0.05, dof_vel_stddev=0.2)
    st, eng = sample(backend, **kw)
    # q = default + stddev * U(-1, 1) before the single simulate; the step moves it by O(h * qd)
    d = st[1] - 0.9
    assert abs(d.mean()) < 0.03 and 0.02 < d.std() < 0.05                  # one 20 ms simulate of gravity sag shifts the mean
    st2, eng2 = sample(backend, **kw)
    assert np.array_equal(st, st2)                                         # same seed -> same draws
    eng2.reset()                                                           # second reset: counter advanced
    assert not np.array_equal(eng2.state.cpu().numpy()[31:38], st[31:38])
    eng.close(), eng2.close()
