"""Physics known-answer tests.  The reference's physics is closed-source PhysX (parity unpinned):
the build's own physics spec is therefore checked against analytic answers and an independent fp64 numpy
model of the URDF chain.  They run on the oracle (CPU suite); the HIP path is bit-identical to the oracle
(tests/test_parity_hip_vs_oracle.py), and `test_finger_dynamics_leaf_gpu` repeats the leaf check on the GPU.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from leibnizgym_amd import _capi as capi
from leibnizgym_amd.engine import TrifingerEngine, make_config

G = 9.81
NO_REWARD = {k: {"activate": False} for k in capi.REWARD_TERM_ORDER}


# ---- independent fp64 model of one finger from the URDF numbers (trifingerpro.urdf) -----------------------
def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


# link inertials, joint origins: tests/golden/model.npz = the numbers of the reference's trifingerpro.urdf (tests/model_fixture.py)
from model_fixture import LINKS, J2, J3, TIP  # noqa: E402


def frames(q):
    R1 = rot_y(q[0]); p1 = np.zeros(3)
    R2 = R1 @ rot_x(q[1]); p2 = p1 + R1 @ J2
    R3 = R2 @ rot_x(q[2]); p3 = p2 + R2 @ J3
    return [(R1, p1), (R2, p2), (R3, p3), (R3, p3)]


def tip_position(q):
    R3, p3 = frames(q)[2]
    return p3 + R3 @ TIP


def potential(q):
    return sum(m * G * (p + R @ c)[2] for (m, c, _), (R, p) in zip(LINKS, frames(q)))


def kinetic_matrix(q, eps=1e-6):
    """M from T = 1/2 qd^T M qd with numerically differentiated COM positions / rotations."""
    M = np.zeros((3, 3))
    fr0 = frames(q)
    Jv, Jw = [], []
    for li in range(4):
        Jv.append(np.zeros((3, 3))); Jw.append(np.zeros((3, 3)))
    for j in range(3):
        dq = np.zeros(3); dq[j] = eps
        frp, frm = frames(q + dq), frames(q - dq)
        for li, (m, c, I) in enumerate(LINKS):
            Jv[li][:, j] = ((frp[li][1] + frp[li][0] @ c) - (frm[li][1] + frm[li][0] @ c)) / (2 * eps)
            dR = (frp[li][0] - frm[li][0]) / (2 * eps)
            W = dR @ fr0[li][0].T
            Jw[li][:, j] = np.array([W[2, 1], W[0, 2], W[1, 0]])
    for li, (m, c, I) in enumerate(LINKS):
        Iw = fr0[li][0] @ I @ fr0[li][0].T
        M += m * Jv[li].T @ Jv[li] + Jw[li].T @ Iw @ Jw[li]
    return M


def call_dynamics(lib, device, q, qd):
    n = q.shape[0]
    cfg = make_config(lib, 1, command_mode="torque")
    eng = TrifingerEngine(cfg, device=device, lib=lib)
    qt = torch.as_tensor(q, dtype=torch.float32).to(device).contiguous()
    qdt = torch.as_tensor(qd, dtype=torch.float32).to(device).contiguous()
    tip = torch.zeros(n, 3, device=device); mass = torch.zeros(n, 9, device=device); bias = torch.zeros(n, 3, device=device)
    rc = lib.tf_test_finger_dynamics(eng._handle, qt.data_ptr(), qdt.data_ptr(), tip.data_ptr(), mass.data_ptr(),
                                     bias.data_ptr(), n, None)
    assert rc == 0
    if device != "cpu":
        torch.cuda.synchronize()
    out = tip.cpu().numpy(), mass.cpu().numpy().reshape(n, 3, 3), bias.cpu().numpy()
    eng.close()
    return out


def _check_dynamics_leaf(lib, device):
    rng = np.random.default_rng(0)
    n = 64
    q = rng.uniform([-0.33, 0.0, -2.7], [1.0, 1.57, 0.0], (n, 3))
    q[0] = [0.0, 0.9, -1.7]
    tip, M, bias = call_dynamics(lib, device, q, np.zeros((n, 3)))
    for i in range(n):
        np.testing.assert_allclose(tip[i], tip_position(q[i]), atol=2e-6)
        Mi = kinetic_matrix(q[i])
        np.testing.assert_allclose(M[i], Mi, atol=2e-7, rtol=2e-4)
        np.testing.assert_allclose(M[i], M[i].T, atol=0)
        assert np.all(np.linalg.eigvalsh(M[i].astype(np.float64)) > 0)
        grad = np.array([(potential(q[i] + d) - potential(q[i] - d)) / 2e-6 for d in np.eye(3) * 1e-6])
        np.testing.assert_allclose(bias[i], grad, atol=3e-6, rtol=2e-4)      # zero velocity: bias = dV/dq
    # FK sanity of SURVEY 8a-P: default pose -> tip radius 0.1032 m, height 0.0773 m (base at z = 0.29)
    assert abs(np.hypot(tip[0][0], tip[0][1]) - 0.1032) < 2e-4
    assert abs(tip[0][2] + 0.29 - 0.0773) < 2e-4


def test_finger_dynamics_leaf(oracle):
    _check_dynamics_leaf(oracle, "cpu")


@pytest.mark.gpu
def test_finger_dynamics_leaf_gpu(hip):
    _check_dynamics_leaf(hip, "cuda:0")


def test_coriolis_bias_is_power_free(oracle):
    """qd^T C(q,qd) qd must equal 1/2 qd^T Mdot qd (passivity): check through d/dt of kinetic energy."""
    rng = np.random.default_rng(1)
    q = rng.uniform([-0.3, 0.1, -2.5], [0.9, 1.4, -0.2], (16, 3))
    qd = rng.uniform(-3, 3, (16, 3))
    _, M0, b = call_dynamics(oracle, "cpu", q, qd)
    _, _, g = call_dynamics(oracle, "cpu", q, np.zeros_like(qd))
    h = 1e-4
    _, Mp, _ = call_dynamics(oracle, "cpu", q + h * qd, qd)
    _, Mm, _ = call_dynamics(oracle, "cpu", q - h * qd, qd)
    for i in range(16):
        Mdot = (Mp[i].astype(np.float64) - Mm[i]) / (2 * h)
        c = (b[i] - g[i]).astype(np.float64)
        lhs = qd[i] @ c
        rhs = 0.5 * qd[i] @ Mdot @ qd[i]
        assert abs(lhs - rhs) < 2e-3 * max(1.0, abs(rhs)) + 2e-4, (lhs, rhs)


def test_bias_forces_match_the_lagrangian(oracle):
    """Full check of the velocity-dependent forces: h(q, qd) = Mdot qd - 1/2 d(qd^T M qd)/dq + dV/dq, with every
    derivative taken numerically on the independent fp64 model (pins all Coriolis / centrifugal terms, not only their
    projection on qd)."""
    rng = np.random.default_rng(7)
    q = rng.uniform([-0.3, 0.1, -2.5], [0.9, 1.4, -0.2], (12, 3))
    qd = rng.uniform(-4, 4, (12, 3))
    _, _, b = call_dynamics(oracle, "cpu", q, qd)
    eps = 1e-5
    for i in range(12):
        dM = [(kinetic_matrix(q[i] + eps * e) - kinetic_matrix(q[i] - eps * e)) / (2 * eps) for e in np.eye(3)]
        Mdot = sum(dM[k] * qd[i][k] for k in range(3))
        dT = np.array([0.5 * qd[i] @ dM[k] @ qd[i] for k in range(3)])
        dV = np.array([(potential(q[i] + eps * e) - potential(q[i] - eps * e)) / (2 * eps) for e in np.eye(3)])
        h = Mdot @ qd[i] - dT + dV
        np.testing.assert_allclose(b[i], h, atol=3e-4, rtol=2e-3)


# ---- whole-env known answers ---------------------------------------------------------------------------
def engine(lib, n=1, model_edit=None, device="cpu", **kw):
    m = lib.default_model()
    if model_edit:
        model_edit(m)
    base = dict(command_mode="torque", normalize_action=False, apply_safety_damping=False, reward_terms=NO_REWARD,
                success={"activate": False}, robot_reset="none", object_reset="none", episode_length=0, model=m)
    base.update(kw)
    eng = TrifingerEngine(make_config(lib, n, **base), device=device, lib=lib)
    # a sane initial state without calling reset(): fingers at default pose, cube at rest in the centre
    eng.q.copy_(torch.tensor([0.0, 0.9, -1.7] * 3, device=device).repeat(n, 1).T)
    eng.cube[0:3] = torch.tensor([0.0, 0.0, 0.0325], device=device)[:, None]
    eng.cube[3:7] = torch.tensor([0.0, 0.0, 0.0, 1.0], device=device)[:, None]
    return eng


def step_zero(eng, k=1, tau=None):
    act = torch.zeros(eng.num_envs, eng.action_dim) if tau is None else tau
    for _ in range(k):
        eng.step(act)


HOLD = dict(command_mode="position", normalize_action=False, apply_safety_damping=True)


def step_hold(eng, k=1):
    """PD-hold the fingers at the default pose (they would otherwise sag onto the table under gravity)."""
    act = torch.tensor([[0.0, 0.9, -1.7] * 3]).repeat(eng.num_envs, 1)
    for _ in range(k):
        eng.step(act)


def test_cube_free_fall_matches_symplectic_euler(oracle):
    def nodamp(m):
        m.cube_linear_damping = 0.0
    eng = engine(oracle, model_edit=nodamp)
    z0 = 0.25
    x0, y0 = 0.104, 0.06            # between two fingers (the upper links sit above the centre and radiate at 90, -30, 210 deg)
    eng.cube[0:3, 0] = torch.tensor([x0, y0, z0])
    h, steps = 0.01, 5          # 5 control steps = 10 substeps of 0.01 s
    step_zero(eng, steps)
    n = 2 * steps
    z_disc = z0 - G * h * h * n * (n + 1) / 2
    c = eng.cube[:, 0].numpy()
    assert abs(c[2] - z_disc) < 2e-6                      # the integrator's own closed form
    assert abs(c[2] - (z0 - 0.5 * G * (n * h) ** 2)) < 6e-3   # continuous answer, O(h) apart
    assert abs(c[9] + G * n * h) < 1e-5                   # v_z = -g t
    assert abs(c[0] - x0) < 1e-7 and abs(c[1] - y0) < 1e-7          # nothing touched it
    np.testing.assert_allclose(c[3:7], [0, 0, 0, 1], atol=1e-7)


def test_cube_rests_on_the_floor(oracle):
    eng = engine(oracle, **HOLD)
    eng.cube[0:2, 0] = torch.tensor([0.05, -0.03])
    step_hold(eng, 300)
    c = eng.cube[:, 0].numpy()
    assert abs(c[2] - 0.0325) < 1e-4                      # no sinking / hovering
    # PGS without warm start creeps a resting body by < 0.1 mm/s; 6 s of simulated time here
    assert abs(c[0] - 0.05) < 5e-4 and abs(c[1] + 0.03) < 5e-4
    assert np.abs(c[7:13]).max() < 2e-3
    assert abs(np.linalg.norm(c[3:7]) - 1) < 1e-6


def test_sliding_friction_stops_the_cube(oracle):
    """mu = 0.55 (average of cube 1.0 and floor 0.1): deceleration mu g, stopping distance v^2 / (2 mu g)."""
    eng = engine(oracle, **HOLD)
    v0 = 0.5
    eng.cube[0:2, 0] = torch.tensor([-0.05, 0.0])
    eng.cube[7, 0] = v0
    step_hold(eng, 1)
    v1 = float(eng.cube[7, 0])
    assert abs((v0 - v1) / 0.02 - 0.55 * G) < 0.08 * 0.55 * G       # Coulomb deceleration over the first step
    step_hold(eng, 30)
    c = eng.cube[:, 0].numpy()
    dist = c[0] + 0.05
    expect = v0 * v0 / (2 * 0.55 * G)
    assert abs(dist - expect) < 0.2 * expect, (dist, expect)
    assert np.abs(c[7:10]).max() < 2e-3 and abs(c[2] - 0.0325) < 2e-4  # stopped, did not tip or lift


def _quat_rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def wall_radius_at(z, m):
    """boundary profile of the model `m`: piecewise linear through its knots, nothing above the last"""
    wr, wz = list(m.wall_r), list(m.wall_z)
    return float(np.interp(z, wz, wr)) if z < wz[-1] else np.inf


def test_wall_keeps_the_cube_in_the_arena(oracle):
    """A fast slide into the boundary: no corner ever gets further out than the wall radius of its height (the vertical ring is
    32 mm high, above it the stage flares outwards: the 65 mm cube rocks against the cone and falls back), and the cube ends at
    rest inside."""
    eng = engine(oracle, **HOLD)
    m = oracle.default_model()
    eng.cube[0:2, 0] = torch.tensor([0.12, 0.0])
    eng.cube[7, 0] = 1.5                                              # fast slide towards the boundary
    corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) * 0.0325
    worst = -1.0
    for _ in range(60):
        step_hold(eng, 1)
        c = eng.cube[:, 0].numpy().astype(np.float64)
        pts = c[0:3] + corners @ _quat_rot(c[3:7]).T
        for p in pts:
            worst = max(worst, np.hypot(p[0], p[1]) - wall_radius_at(p[2], m))
    assert worst < 2.5e-3, worst                                      # speculative-contact slop at 1.5 m/s
    assert worst > -0.01                                              # and it did reach the wall
    c = eng.cube[:, 0].numpy()
    assert np.hypot(c[0], c[1]) < float(m.wall_r[0]) - 0.0325 + 1e-3 and abs(c[2] - 0.0325) < 2e-4 and np.abs(c[7:13]).max() < 5e-3
    assert np.isfinite(eng.state.numpy()).all()


def finger_energy(q, qd):
    return 0.5 * qd @ kinetic_matrix(q) @ qd + potential(q)


def _free_swing(oracle, q0, qd0, gravity, max_steps):
    """Zero torque, no damping, contacts disabled: a conservative 3-link chain.  Returns the energy error trace
    up to the first joint stop."""
    def edit(m):
        m.link_angular_damping = 0.0
        m.contact_margin = -1.0                                       # no contact rows at all
    eng = engine(oracle, model_edit=edit, gravity=gravity)
    eng.cube[0:3, 0] = torch.tensor([0.0, 0.15, 0.0325])
    eng.q[0:3, 0] = torch.tensor(q0, dtype=torch.float32)
    eng.qd[0:3, 0] = torch.tensor(qd0, dtype=torch.float32)
    g = -gravity[2]

    def energy(q, qd):
        return 0.5 * qd @ kinetic_matrix(q) @ qd + potential(q) * (g / G)
    e0 = energy(np.array(q0, dtype=np.float64), np.array(qd0, dtype=np.float64))
    errs, speeds, kin = [], [], []
    for _ in range(max_steps):
        step_zero(eng, 1)
        q = eng.q[0:3, 0].numpy().astype(np.float64)
        qd = eng.qd[0:3, 0].numpy().astype(np.float64)
        if (q <= [-0.33 + 1e-3, 1e-3, -2.7 + 1e-3]).any() or (q >= [1.0 - 1e-3, 1.57 - 1e-3, -1e-3]).any():
            break                                                     # reached a joint stop: no longer conservative
        if np.abs(qd).max() > 9.0:
            break                                                     # the 10 rad/s velocity limit dissipates
        errs.append(energy(q, qd) - e0)
        speeds.append(np.abs(qd).max())
        kin.append(0.5 * qd @ kinetic_matrix(q) @ qd)
    return e0, np.array(errs), np.array(speeds), np.array(kin)


def test_free_finger_kinetic_energy_is_conserved_without_gravity(oracle):
    """g = 0: T = 1/2 qd^T M(q) qd is an invariant; exercises the Coriolis/centrifugal terms and M^-1."""
    e0, errs, speeds, _ = _free_swing(oracle, [0.3, 0.8, -1.4], [1.5, -1.0, 2.0], (0.0, 0.0, 0.0), 60)
    assert len(errs) >= 8 and speeds.max() > 1.0
    assert np.abs(errs).max() < 0.03 * e0, (np.abs(errs).max(), e0)   # symplectic Euler at h = 0.01: O(h) wobble


def test_free_finger_energy_exchange_under_gravity(oracle):
    """Released from rest under gravity: potential energy turns into kinetic energy, total conserved."""
    e0, errs, speeds, kin = _free_swing(oracle, [0.0, 1.3, -2.4], [0.0, 0.0, 0.0], (0.0, 0.0, -G), 40)
    assert len(errs) >= 4 and speeds.max() > 4.0                      # it really fell
    # what was gained as kinetic energy was lost as potential energy.  Symplectic Euler started from rest lags by
    # exactly one half-step of kinetic energy: for free fall (T - dV) / T = -1/n after n substeps.
    n_sub = 2.0 * (np.arange(len(errs)) + 1)
    assert (errs < 0).all() and (np.abs(errs) < 1.35 * kin / n_sub + 1e-4).all(), (errs, kin)


def test_joint_and_velocity_limits_hold(oracle):
    eng = engine(oracle, n=4)
    eng.cube[0:3] = torch.tensor([0.0, 0.0, 0.25])[:, None] * 0 + torch.tensor([0.13, 0.13, 0.0325])[:, None]
    tau = torch.zeros(4, 9)
    tau[0, :] = 0.36
    tau[1, :] = -0.36
    tau[2, 0::3] = 0.36
    tau[3, 2::3] = -0.36
    lo = np.array([-0.33, 0.0, -2.7] * 3)[:, None]
    hi = np.array([1.0, 1.57, 0.0] * 3)[:, None]
    vmax = 0.0
    for _ in range(60):
        eng.step(tau)
        q, qd = eng.q.numpy(), eng.qd.numpy()
        assert (q >= lo - 1e-6).all() and (q <= hi + 1e-6).all()
        vmax = max(vmax, np.abs(qd).max())
        assert np.abs(qd).max() <= 10.0 + 1e-4
    assert vmax > 5.0                                                 # the limit was actually exercised
    q = eng.q.numpy()
    assert abs(q[0, 2] - 1.0) < 1e-3                                  # env 2: joint 0 driven into its upper stop
    assert abs(q[2, 3] + 2.7) < 1e-3                                  # env 3: joint 2 driven into its lower stop
    assert np.isfinite(eng.state.numpy()).all()


def test_fingertip_does_not_sink_through_the_floor(oracle):
    eng = engine(oracle)
    eng.cube[0:3, 0] = torch.tensor([0.0, 0.15, 0.0325])
    tau = torch.zeros(1, 9)
    tau[0, 1] = -0.36                                                 # swing finger 0 down towards the table
    tau[0, 2] = 0.2
    low = 1.0
    for _ in range(100):
        eng.step(tau)
        q = eng.q[0:3, 0].numpy().astype(np.float64)
        R3, p3 = frames(q)[2]
        ball = p3 + R3 @ np.array(list(oracle.default_model().cap_b), dtype=np.float64)
        low = min(low, ball[2] + 0.29 - 0.0102)
    assert low > -1.5e-3, low                                         # at most the speculative-contact slop
    assert low < 0.02                                                 # and it did reach the floor


def test_finger_pushes_cube_without_blowing_up(oracle):
    """Position-controlled fingers close on a centred cube: contact stays bounded and symmetric."""
    eng = engine(oracle, command_mode="position", normalize_action=False, apply_safety_damping=True)
    target = torch.tensor([[0.0, 0.9, -2.0] * 3])                     # curl the distal links towards the centre
    vmax = 0.0
    for _ in range(200):
        eng.step(target)
        c = eng.cube[:, 0].numpy()
        vmax = max(vmax, np.abs(c[7:10]).max())
    assert np.isfinite(eng.state.numpy()).all()
    assert vmax < 2.0 and abs(c[2] - 0.0325) < 0.02 and np.hypot(c[0], c[1]) < 0.05


def test_three_fold_symmetry(oracle):
    """The three fingers are copies rotated by 120 degrees about z: with a centred, yaw-symmetric scene the
    joint trajectories of the three fingers stay identical."""
    eng = engine(oracle, command_mode="position", normalize_action=False)
    eng.cube[0:3, 0] = torch.tensor([0.0, 0.0, 5.0])                  # cube out of reach for the whole test
    tgt = torch.tensor([[0.3, 1.1, -1.2] * 3])
    for _ in range(15):
        eng.step(tgt)
    q = eng.q[:, 0].numpy().reshape(3, 3)
    np.testing.assert_allclose(q[0], q[1], atol=1e-6)
    np.testing.assert_allclose(q[0], q[2], atol=1e-6)
