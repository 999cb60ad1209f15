"""Known-answer contact scenarios: three-finger pinch-and-lift, fingertip-cube impact
(impulse consistency, restitution above / below the bounce threshold), a middle link that cannot pass through the cube,
two fingers that stop at two capsule radii, a fingertip that stays inside the boundary.

Everything that is compared with the product is computed from the independent fp64 URDF model (tests/physics_ref.py,
tests/test_physics_analytic.py): tip positions, Jacobians, gravity torques, closest points.  The controllers are the
test's own (gravity compensation + Cartesian impedance through J^T, torque mode), so that the reference's PD gains
(`set for 250 Hz`, trifinger_env.py:218, applied at 50 Hz) are not part of what is tested.

The scenarios run on the oracle in the CPU suite and on the HIP library under `-m gpu`.
"""
import numpy as np
import pytest
import torch
from scipy.optimize import least_squares

import physics_ref as PR
import test_physics_analytic as T
from leibnizgym_amd import _capi as capi

R_TIP = PR.TIP_CAP[3]
TIP = PR.TIP_CAP[2]


def ik(f, target_world, q0=(0.0, 0.9, -1.7)):
    r = least_squares(lambda q: PR.link_point_world(f, q, 3, TIP) - target_world, q0,
                      bounds=(PR.Q_LO + 1e-3, PR.Q_HI - 1e-3), xtol=1e-12, ftol=1e-12)
    assert np.linalg.norm(r.fun) < 1e-7, "fingertip target out of reach"
    return r.x


def gravity_torque(q):
    return PR.bias_forces(q, np.zeros(3), -9.81)


def state_np(eng):
    return eng.state[:, 0].cpu().numpy().astype(np.float64)


def step_torque(eng, tau):
    eng.step(torch.tensor(np.clip(tau, -0.36, 0.36), dtype=torch.float32, device=eng.device)[None, :])


def torque_engine(lib, device, model_edit=None, **kw):
    base = dict(command_mode="torque", normalize_action=False, apply_safety_damping=False)
    base.update(kw)
    return T.engine(lib, device=device, model_edit=model_edit, **base)


def impedance_torques(st, targets, kp=200.0, kd=2.0):
    """gravity compensation + J^T (kp (target - tip) - kd tip velocity) for every finger with a target"""
    q, qd = st[0:9], st[9:18]
    tau = np.zeros(9)
    for f in range(3):
        qf = q[3 * f:3 * f + 3]
        tau[3 * f:3 * f + 3] = gravity_torque(qf)
        if targets[f] is not None:
            tip = PR.link_point_world(f, qf, 3, TIP)
            J = PR.point_jacobian(f, qf, 3, tip)
            tau[3 * f:3 * f + 3] += J.T @ (kp * (targets[f] - tip) - kd * (J @ qd[3 * f:3 * f + 3]))
    return tau


# ---- three-finger pinch and lift -----------------------------------------------------------------------------------
def _pinch_and_lift(lib, device, warm_start):
    def edit(m):
        m.warm_start = warm_start
    eng = torque_engine(lib, device, edit)
    d = PR.CUBE_HALF + R_TIP + 0.0005
    tips = [np.array([0.0, d, 0.0325]), np.array([d, -d * np.tan(np.pi / 6), 0.0325]), np.array([-d, -d * np.tan(np.pi / 6), 0.0325])]
    inward = [-t / np.linalg.norm(t[:2]) * np.array([1, 1, 0]) for t in tips]
    q0 = np.concatenate([ik(f, tips[f]) for f in range(3)])
    eng.q[:, 0] = torch.tensor(q0, dtype=torch.float32, device=device)
    lift, rel0, zs, slip = 0.0, None, [], []
    for i in range(450):                      # 50 steps squeeze, 55 steps lift by 55 mm, then hold
        if i >= 50:
            lift = min(0.055, lift + 0.001)
        targets = [tips[f] + 0.0075 * inward[f] + np.array([0, 0, lift]) for f in range(3)]   # 7.5 mm inside: 1.5 N
        step_torque(eng, impedance_torques(state_np(eng), targets))
        st = state_np(eng)
        c = st[capi.S_CUBE_P:capi.S_CUBE_P + 13]
        R = PR.quat_rot(c[3:7])
        rel = np.concatenate([R.T @ (PR.link_point_world(f, st[3 * f:3 * f + 3], 3, TIP) - c[0:3]) for f in range(3)])
        if i == 249:
            rel0 = rel.copy()
        if i >= 250:
            zs.append(c[2])
            slip.append(np.abs(rel - rel0).max())
    assert np.isfinite(st).all()
    eng.close()
    return np.array(zs), np.array(slip), st


def _check_pinch_and_lift(lib, device):
    zs, slip, st = _pinch_and_lift(lib, device, lib.default_model().warm_start)
    # held for 200 control steps (4 s) well above the table, fingertips do not creep on the faces, nothing drifts
    assert zs.min() > 0.06, zs.min()
    assert slip.max() < 1e-3, slip.max()
    assert abs(zs[-1] - zs[0]) < 5e-4
    c = st[capi.S_CUBE_P:capi.S_CUBE_P + 13]
    assert np.abs(c[7:10]).max() < 2e-3 and np.abs(c[10:13]).max() < 2e-2      # at rest in the grasp
    lam = st[capi.S_LAM_FC:capi.S_LAM_FC + 12:4]
    assert (st[capi.S_FC_LINK:capi.S_FC_LINK + 3] == 3).all() and (lam > 0.005).all()   # three live fingertip contacts


def test_pinch_and_lift(oracle):
    _check_pinch_and_lift(oracle, "cpu")


def test_pinch_and_lift_needs_the_warm_start(oracle):
    """the same grasp with cold-started sweeps creeps by millimetres: the known-answer test above does bite"""
    zs, slip, _ = _pinch_and_lift(oracle, "cpu", 0.0)
    assert slip.max() > 3e-3


@pytest.mark.gpu
def test_pinch_and_lift_gpu(hip):
    _check_pinch_and_lift(hip, "cuda:0")


# ---- in-hand rotation: the pinch transmits torque ---------------------------------------------------------------------
def _pinch_lift_twist(lib, device, yaw=0.45, tilt=0.25):
    """Pinch and lift as above, then the three fingertip targets are rotated about the VERTICAL axis through the grasp centre by `yaw`
    (and back to half of it), then about a HORIZONTAL axis by `tilt`: three frictional point contacts (one point per finger, no torsional
    friction - PhysX applies none either without a patch radius, trifinger_env.py:877) must carry the cube along in all of its six degrees
    of freedom.  Returns commanded and reached cube rotations."""
    eng = torque_engine(lib, device)
    d = PR.CUBE_HALF + R_TIP + 0.0005
    tips = [np.array([0.0, d, 0.0325]), np.array([d, -d * np.tan(np.pi / 6), 0.0325]), np.array([-d, -d * np.tan(np.pi / 6), 0.0325])]
    inward = [-t / np.linalg.norm(t[:2]) * np.array([1, 1, 0]) for t in tips]
    eng.q[:, 0] = torch.tensor(np.concatenate([ik(f, tips[f]) for f in range(3)]), dtype=torch.float32, device=device)
    grasp = [tips[f] + 0.0075 * inward[f] - np.array([0, 0, 0.0325]) for f in range(3)]      # relative to the grasp centre
    lift, a_yaw, a_tilt, out = 0.0, 0.0, 0.0, {}

    def rot(ax, a):
        c, s_ = np.cos(a), np.sin(a)
        return np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]]) if ax == "z" else np.array([[1, 0, 0], [0, c, -s_], [0, s_, c]])
    for i in range(700):
        if 50 <= i:
            lift = min(0.055, lift + 0.001)
        if 200 <= i < 350:
            a_yaw = yaw * (i - 199) / 150.0
        if 350 <= i < 425:
            a_yaw = yaw * (1.0 - 0.5 * (i - 349) / 75.0)
        if 450 <= i < 600:
            a_tilt = tilt * (i - 449) / 150.0
        Rt = rot("z", a_yaw) @ rot("x", a_tilt)
        centre = np.array([0.0, 0.0, 0.0325 + lift])
        targets = [centre + Rt @ grasp[f] for f in range(3)]
        step_torque(eng, impedance_torques(state_np(eng), targets))
        st = state_np(eng)
        if i in (199, 349, 424, 449, 699):
            c = st[capi.S_CUBE_P:capi.S_CUBE_P + 13]
            out[i] = (c[0:3].copy(), PR.quat_rot(c[3:7]), Rt.copy())
    assert np.isfinite(st).all()
    lam = st[capi.S_LAM_FC:capi.S_LAM_FC + 12:4]
    eng.close()
    return out, lam


def _check_twist(lib, device):
    out, lam = _pinch_lift_twist(lib, device)

    def rotvec(R):
        ang = float(np.arccos(np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)))
        ax = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        return ang, ax / max(np.linalg.norm(ax), 1e-12)
    R0 = out[199][1]
    # Fingertip SPHERES (radius r) roll on the faces of the cube (half width h): turning the ring of fingertip centres by theta about the
    # grasp centre turns the cube by about theta (h + r) / h = 1.31 theta when the distal links keep their orientation (rolling without
    # slipping: (h + r) dtheta = h dphi), a little less because they turn along.  Measured on this build: 1.23, the same on the way back.
    ratio = (PR.CUBE_HALF + R_TIP) / PR.CUBE_HALF
    a1, ax1 = rotvec(out[349][1] @ R0.T)
    a2, ax2 = rotvec(out[424][1] @ R0.T)
    assert 1.1 < a1 / 0.45 < ratio + 0.05 and 1.1 < a2 / 0.225 < ratio + 0.05, (a1, a2)
    assert abs(a1 / 0.45 - a2 / 0.225) < 0.06                                           # no slip on the way back: the same gearing
    assert ax1[2] > 0.995 and ax2[2] > 0.995                                             # about the vertical axis
    a3, ax3 = rotvec(out[699][1] @ out[449][1].T)                                        # then the tilt about a horizontal axis of the grasp
    want_axis = np.array([np.cos(0.225), np.sin(0.225), 0.0])
    assert 0.9 < a3 / 0.25 < ratio + 0.15 and float(ax3 @ want_axis) > 0.97, (a3, ax3)
    for key in (349, 424, 699):
        pos = out[key][0]
        assert pos[2] > 0.07 and np.hypot(pos[0], pos[1]) < 0.01, (key, pos)            # still held at the lifted grasp centre
    assert (lam > 0.005).all()


def test_three_fingertips_rotate_the_pinched_cube(oracle):
    """does the contact model let a grasp change the cube's ORIENTATION (the half of the difficulty-4 goal that the
    trained policy does not reach)?  Yes: a scripted pinch yaws the lifted cube by 0.55 rad, brings it half-way back and tilts it by 0.3 rad -
    through friction at three separated points with the gearing of fingertip spheres rolling on its faces; no torsional friction needed."""
    _check_twist(oracle, "cpu")


@pytest.mark.gpu
def test_three_fingertips_rotate_the_pinched_cube_gpu(hip):
    _check_twist(hip, "cuda:0")


# ---- fingertip - cube impact ---------------------------------------------------------------------------------------
def _impact(lib, device, speed, gap=0.001):
    """Finger 0's tip flies at `speed` along -y onto the +y face of a cube floating at rest (no gravity): one substep."""
    h = 0.01
    eng = torque_engine(lib, device, dt=h, substeps=1, gravity=(0.0, 0.0, 0.0))
    q = np.array([0.1, 0.8, -1.6])
    tip = PR.link_point_world(0, q, 3, TIP)
    centre = tip - np.array([0.004, PR.CUBE_HALF + R_TIP + gap, -0.006])    # off-centre hit: the cube must also spin
    J = PR.point_jacobian(0, q, 3, tip)
    qd = np.linalg.solve(J, np.array([0.0, -speed, 0.0]))
    f32 = dict(dtype=torch.float32, device=device)
    eng.q[0:3, 0] = torch.tensor(q, **f32)
    eng.qd[0:3, 0] = torch.tensor(qd, **f32)
    eng.cube[0:3, 0] = torch.tensor(centre, **f32)
    eng.simulate()
    st = state_np(eng)
    eng.close()
    qd1 = st[9:12]
    v1, w1 = st[capi.S_CUBE_V:capi.S_CUBE_V + 3], st[capi.S_CUBE_W:capi.S_CUBE_W + 3]
    n = np.array([0.0, 1.0, 0.0])                      # from the cube to the finger
    r = np.array([0.004, PR.CUBE_HALF, -0.006])        # contact point on the face, relative to the centre
    P_contact = tip - R_TIP * n                        # on the sphere
    Jc = PR.point_jacobian(0, q, 3, P_contact)
    # free velocity of the finger (what the solver starts from), from the independent model
    M = T.kinetic_matrix(q)
    qd_free = (qd + h * np.linalg.solve(M, -PR.bias_forces(q, qd, 0.0))) * (1.0 - h * PR.LINK_DAMP)
    return dict(q=q, qd=qd, qd1=qd1, v1=v1, w1=w1, n=n, r=r, Jc=Jc, M=M, qd_free=qd_free, speed=speed)


def _check_impact(lib, device):
    for speed, bounces in ((1.0, True), (0.3, False)):
        d = _impact(lib, device, speed)
        m, inertia = PR.CUBE_MASS, PR.CUBE_INERTIA
        P = m * d["v1"]                                            # impulse the cube received
        assert P @ d["n"] < -1e-4                                  # pushed away from the finger
        # the impulse acted AT the contact point: angular momentum about the centre = r x P
        np.testing.assert_allclose(inertia * d["w1"], np.cross(d["r"], P), rtol=2e-3, atol=2e-7)
        # and the finger received the opposite impulse through the contact Jacobian
        np.testing.assert_allclose(d["M"] @ (d["qd1"] - d["qd_free"]), d["Jc"].T @ (-P), rtol=2e-2, atol=2e-6)
        # friction pyramid: tangential part within mu (per axis) of the normal part
        Pn = -(P @ d["n"])
        assert np.abs(P - (P @ d["n"]) * d["n"]).max() <= PR.MU["fc"] * Pn * (1 + 1e-4)
        # relative normal velocity at the contact after the solve
        v_rel0 = d["n"] @ (d["Jc"] @ d["qd_free"])                 # approach speed the bias was computed from (< 0)
        v_rel1 = d["n"] @ (d["Jc"] @ d["qd1"] - (d["v1"] + np.cross(d["w1"], d["r"])))
        if bounces:      # above the 0.5 m/s threshold and inside contact_offset: separates with e = 0.4
            assert v_rel0 < -PR.BOUNCE
            assert abs(v_rel1 - (-PR.REST_F * v_rel0)) < 0.02 * abs(v_rel0), (v_rel1, v_rel0)
        else:            # below the threshold: no bounce, the remaining 1 mm gap closes exactly (speculative row)
            assert -PR.BOUNCE < v_rel0 < 0
            assert abs(v_rel1 - (-0.001 / 0.01)) < 5e-3, v_rel1
        # no energy is created
        ke0 = 0.5 * d["qd_free"] @ d["M"] @ d["qd_free"]
        ke1 = 0.5 * d["qd1"] @ d["M"] @ d["qd1"] + 0.5 * m * d["v1"] @ d["v1"] + 0.5 * inertia * d["w1"] @ d["w1"]
        assert ke1 <= ke0 * (1 + 1e-3)


def test_impact_impulse_and_restitution(oracle):
    _check_impact(oracle, "cpu")


@pytest.mark.gpu
def test_impact_impulse_and_restitution_gpu(hip):
    _check_impact(hip, "cuda:0")


# ---- middle link vs cube -------------------------------------------------------------------------------------------
def _capsule_gap(f, qf, link, cube):
    """smallest gap of the capsules of one link against the cube"""
    return min(g for g, _ in PR.finger_gaps(f, qf, cube[0:3], PR.quat_rot(cube[3:7]), np.full(3, PR.CUBE_HALF), links=(link,)))


def _middle_link_run(lib, device, contacts_on):
    def edit(m):
        if not contacts_on:
            m.contact_margin = -1.0
    eng = torque_engine(lib, device, edit, gravity=(0.0, 0.0, 0.0))
    q = np.array([-0.25, 0.35, -1.2])
    mid_local = np.array([0.028, 0.0, -0.08])          # centre of mass of the middle link (trifingerpro.urdf:114-118): mid-way along it
    mid = PR.link_point_world(0, q, 2, mid_local)
    # a floating cube beside the middle of finger 0's middle link, on the side joint 1 is about to swing it to, 1 cm from its capsules
    side = PR.link_point_world(0, q + np.array([0.3, 0, 0]), 2, mid_local) - mid
    side /= np.linalg.norm(side)
    lo_, hi_ = 0.0, 0.2
    for _ in range(50):                                 # bisection on the distance along `side` at which the link's gap is 1 cm
        d_ = 0.5 * (lo_ + hi_)
        if _capsule_gap(0, q, 2, np.concatenate([mid + side * d_, [0, 0, 0, 1]])) < 0.01:
            lo_ = d_
        else:
            hi_ = d_
    centre = mid + side * hi_
    f32 = dict(dtype=torch.float32, device=device)
    eng.q[0:3, 0] = torch.tensor(q, **f32)
    eng.cube[0:3, 0] = torch.tensor(centre, **f32)
    worst, links, moved = 1.0, set(), 0.0
    for _ in range(40):
        st = state_np(eng)
        tau = np.zeros(9)
        tau[0] = 0.2                                   # swing joint 1: the middle link sweeps sideways into the cube
        step_torque(eng, tau)
        st = state_np(eng)
        cube = st[capi.S_CUBE_P:capi.S_CUBE_P + 13]
        worst = min(worst, _capsule_gap(0, st[0:3], 2, cube))
        links.add(int(st[capi.S_FC_LINK]) & 3)         # activity code: link + 4 x (fingertip-wall contact pushing)
        moved = max(moved, np.linalg.norm(cube[0:3] - centre))
    eng.close()
    return worst, links, moved


def _check_middle_link(lib, device):
    worst, links, moved = _middle_link_run(lib, device, True)
    assert worst > -4e-3, worst                      # never deeper than the slop of an impact at ~1 m/s with 8 sweeps
    assert 2 in links                                # the contact slot was held by the middle link
    assert moved > 0.02                              # and the cube was pushed away
    ghost, _, still = _middle_link_run(lib, device, False)
    assert ghost < -0.012 and still < 1e-6           # without contacts the link passes straight through: the test bites


def test_middle_link_cannot_pass_through_the_cube(oracle):
    _check_middle_link(oracle, "cpu")


@pytest.mark.gpu
def test_middle_link_cannot_pass_through_the_cube_gpu(hip):
    _check_middle_link(hip, "cuda:0")


# ---- the thick upper part of the distal body vs cube (round 3: the link shapes follow the reference's collision hulls) ---------------
def _distal_body_run(lib, device, contacts_on, local_point):
    """finger 0 swings about joint 1 so that a point of its distal body - given in the lower-link frame, far from the fingertip -
    sweeps into a floating cube placed 1 cm beside it"""
    def edit(m):
        if not contacts_on:
            m.contact_margin = -1.0
    eng = torque_engine(lib, device, edit, gravity=(0.0, 0.0, 0.0))
    q = np.array([-0.25, 0.5, -1.0])
    p0 = PR.link_point_world(0, q, 3, local_point)
    side = PR.link_point_world(0, q + np.array([0.3, 0, 0]), 3, local_point) - p0
    side /= np.linalg.norm(side)
    lo_, hi_ = 0.0, 0.2
    for _ in range(50):                                 # bisection: distance along `side` at which the distal body's gap is 1 cm
        d_ = 0.5 * (lo_ + hi_)
        if _capsule_gap(0, q, 3, np.concatenate([p0 + side * d_, [0, 0, 0, 1]])) < 0.01:
            lo_ = d_
        else:
            hi_ = d_
    centre = p0 + side * hi_
    f32 = dict(dtype=torch.float32, device=device)
    eng.q[0:3, 0] = torch.tensor(q, **f32)
    eng.cube[0:3, 0] = torch.tensor(centre, **f32)
    worst, links, moved = 1.0, set(), 0.0
    tip_gap_at_contact = None
    for _ in range(40):
        tau = np.zeros(9)
        tau[0] = 0.2
        step_torque(eng, tau)
        st = state_np(eng)
        cube = st[capi.S_CUBE_P:capi.S_CUBE_P + 13]
        g = _capsule_gap(0, st[0:3], 3, cube)
        worst = min(worst, g)
        if (int(st[capi.S_FC_LINK]) & 3) == 3 and tip_gap_at_contact is None and st[capi.S_LAM_FC] > 0:
            tip = PR.link_point_world(0, st[0:3], 3, PR.TIP_CAP[2])
            loc = PR.quat_rot(cube[3:7]).T @ (tip - cube[0:3])
            tip_gap_at_contact = np.linalg.norm(np.maximum(np.abs(loc) - PR.CUBE_HALF, 0.0)) - PR.TIP_CAP[3]
        links.add(int(st[capi.S_FC_LINK]) & 3)
        moved = max(moved, np.linalg.norm(cube[0:3] - centre))
    eng.close()
    return worst, links, moved, tip_gap_at_contact


def _check_distal_body(lib, device):
    for local in (np.array([0.012, 0.0, -0.03]), np.array([0.011, 0.0, 0.0])):     # the thick part of the body; the joint-3 housing
        worst, links, moved, tip_gap = _distal_body_run(lib, device, True, local)
        assert worst > -8e-3, (local, worst)             # the shape stops at the cube: a finger swinging at 5-7 rad/s juggles a weightless cube
                                                         # spinning at 10 rad/s; single-step overlaps of a few mm (8 sweeps, one-point gap rule)
        assert 3 in links and moved > 0.02               # the distal body held the contact and pushed the cube away
        assert tip_gap is not None and tip_gap > 0.03    # ... with the fingertip sphere centimetres away: it was the body, not the tip
        ghost, _, still, _ = _distal_body_run(lib, device, False, local)
        assert ghost < -0.012 and still < 1e-6           # without contacts it passes straight through


def test_distal_body_cannot_pass_through_the_cube(oracle):
    _check_distal_body(oracle, "cpu")


@pytest.mark.gpu
def test_distal_body_cannot_pass_through_the_cube_gpu(hip):
    _check_distal_body(hip, "cuda:0")


# ---- finger vs finger ----------------------------------------------------------------------------------------------
def _finger_finger_run(lib, device, contacts_on):
    def edit(m):
        if not contacts_on:
            m.contact_margin = -1.0
    eng = torque_engine(lib, device, edit)
    f32 = dict(dtype=torch.float32, device=device)
    eng.cube[0:3, 0] = torch.tensor([0.0, 0.0, 5.0], **f32)          # cube out of the way
    meet = np.array([0.03, 0.02, 0.09])                               # both fingertips are told to go to the same point
    worst, dists = 1.0, []
    for i in range(150):
        step_torque(eng, impedance_torques(state_np(eng), [meet, meet, None], kp=120.0, kd=2.0))
        st = state_np(eng)
        segs = [(PR.link_point_world(f, st[3 * f:3 * f + 3], 3, PR.TIP_CAP[1]), PR.link_point_world(f, st[3 * f:3 * f + 3], 3, TIP))
                for f in (0, 1)]
        Pa, Pb = PR.segment_segment(segs[0][0], segs[0][1], segs[1][0], segs[1][1])
        dists.append(np.linalg.norm(Pa - Pb))
        worst = min(worst, dists[-1])
    assert np.isfinite(st).all()
    eng.close()
    return worst, st, np.array(dists)


def _check_finger_finger(lib, device):
    worst, st, dists = _finger_finger_run(lib, device, True)
    # pressed together the distal capsules come to rest at exactly two radii; two round fingertips pushed onto each
    # other slip off now and then, which shows as a transient of a few mm when the contact normal swings round
    assert worst > 2 * R_TIP - 3e-3, worst
    assert np.abs(dists[-30:] - 2 * R_TIP).max() < 3e-4, dists[-30:]
    assert np.abs(st[9:15]).max() < 0.05             # at rest
    ghost, _, _ = _finger_finger_run(lib, device, False)
    assert ghost < 2 * R_TIP - 8e-3                  # without the contact they interpenetrate


def test_two_fingers_stop_at_two_radii(oracle):
    _check_finger_finger(oracle, "cpu")


@pytest.mark.gpu
def test_two_fingers_stop_at_two_radii_gpu(hip):
    _check_finger_finger(hip, "cuda:0")


# ---- fingertip vs boundary -----------------------------------------------------------------------------------------
def _tip_wall_run(lib, device, contacts_on):
    def edit(m):
        if not contacts_on:
            m.contact_margin = -1.0
    eng = torque_engine(lib, device, edit)
    f32 = dict(dtype=torch.float32, device=device)
    eng.cube[0:3, 0] = torch.tensor([0.0, 0.0, 5.0], **f32)
    target = np.array([0.0, 0.205, 0.045])           # outside the 0.192 m wall, below its 60 mm step
    worst = -1.0
    for i in range(200):
        step_torque(eng, impedance_torques(state_np(eng), [target, None, None], kp=30.0, kd=1.5))
        st = state_np(eng)
        tip = PR.link_point_world(0, st[0:3], 3, TIP)
        wc, _ = PR.wall_tilt(tip[2])
        worst = max(worst, R_TIP - (PR.wall_radius_at(tip[2]) - np.hypot(tip[0], tip[1])) * wc)     # depth inside the (tilted) surface of the boundary
    eng.close()
    return worst, tip, st


def _check_tip_wall(lib, device):
    worst, tip, st = _tip_wall_run(lib, device, True)
    assert worst < 1.5e-3, worst                     # the fingertip never gets through the boundary
    assert worst > -2e-3                             # and it did reach it
    assert st[capi.S_LAM_TW] > 0                     # resting against the wall: a live wall contact
    ghost, _, _ = _tip_wall_run(lib, device, False)
    assert ghost > 0.01


def test_fingertip_stays_inside_the_boundary(oracle):
    _check_tip_wall(oracle, "cpu")


@pytest.mark.gpu
def test_fingertip_stays_inside_the_boundary_gpu(hip):
    _check_tip_wall(hip, "cuda:0")


# ---- the flared part of the boundary pushes a fingertip inward AND up -------------------------------------------------------------------
def _tip_on_cone_run(lib, device, frictionless):
    """Finger 0's tip is pressed radially outward (a soft spring towards a point outside the boundary, at the height it starts from) against the
    flared part of the stage at z = 70 mm, where the wall leans outward by 29.5 degrees (slope 0.565 between the knots at 60 and 100 mm)."""
    def edit(m):
        if frictionless:
            m.mu_tip_wall = 0.0
    eng = torque_engine(lib, device, edit)
    f32 = dict(dtype=torch.float32, device=device)
    eng.cube[0:3, 0] = torch.tensor([0.0, 0.0, 5.0], **f32)
    eng.q[0:3, 0] = torch.tensor(ik(0, np.array([0.0, 0.20, 0.07])), **f32)
    target, kp = np.array([0.0, 0.235, 0.07]), 20.0
    track = []
    for _ in range(200):
        step_torque(eng, impedance_torques(state_np(eng), [target, None, None], kp=kp, kd=1.0))
        st = state_np(eng)
        tip = PR.link_point_world(0, st[0:3], 3, TIP)
        wc, wsn = PR.wall_tilt(tip[2])
        dist = (PR.wall_radius_at(tip[2]) - np.hypot(tip[0], tip[1])) * wc - R_TIP
        track.append((tip[1], tip[2], dist, st[capi.S_LAM_TW], wc, wsn))
    eng.close()
    return np.array(track), target, kp


def _check_tip_on_cone(lib, device):
    tr, target, kp = _tip_on_cone_run(lib, device, True)
    # without friction the reaction of the leaning wall has an upward component: the tip climbs the cone until the spring, which now also pulls it
    # down, has no component along the surface any more - and it sits ON the tilted surface (a horizontal contact normal would leave it at 70 mm)
    y, z, dist, lam, wc, wsn = tr[-1]
    assert 0.078 < z < 0.092, z
    assert np.abs(tr[100:, 2]).max() < 5e-4 and lam > 0
    spring = kp * np.array([target[1] - y, target[2] - z])
    assert abs(spring @ np.array([wsn, wc])) < 0.03 * np.linalg.norm(spring), spring       # tangent of the surface in the (r, z) plane: (sin, cos)
    # with the friction of the pair (1.0 > tan 29.5 deg = 0.565) the tip stays where it touched
    tr2, _, _ = _tip_on_cone_run(lib, device, False)
    assert abs(tr2[-1, 1] - 0.07) < 2.5e-3 and np.abs(tr2[100:, 2]).max() < 5e-4 and tr2[-1, 3] > 0


def test_fingertip_on_the_flared_boundary_is_pushed_inward_and_up(oracle):
    """VERDICT round 3, missing 3 (boundary fidelity), the fingertip half: the contact normal of the fingertip - boundary contact follows the cone of
    the stage (high_table_boundary.urdf:20-259), and the gap is the distance to the tilted surface."""
    _check_tip_on_cone(oracle, "cpu")


@pytest.mark.gpu
def test_fingertip_on_the_flared_boundary_is_pushed_inward_and_up_gpu(hip):
    _check_tip_on_cone(hip, "cuda:0")


# ---- the finger - cube - floor chain in its steady state ------------------------------------------------------------------------------
# The slow case of the Gauss-Seidel sweeps (a heavy finger loading the light cube against the static floor) has exact
# steady states; the warm start carries the impulses there over the substeps, and these are what a policy that pushes or holds the cube lives in.
def _press_down(lib, device):
    """finger 0 presses on the top face of the resting cube with a soft spring: what the floor carries is the weight plus the finger's force"""
    eng = torque_engine(lib, device)
    f32 = dict(dtype=torch.float32, device=device)
    eng.cube[0:3, 0] = torch.tensor([0.0, 0.10, 0.0325], **f32)
    top = np.array([0.0, 0.10, 2 * PR.CUBE_HALF + R_TIP + 0.0005])
    eng.q[0:3, 0] = torch.tensor(ik(0, top), **f32)
    h = 0.01
    for _ in range(150):
        step_torque(eng, impedance_torques(state_np(eng), [top + np.array([0.0, 0.0, -0.010]), None, None], kp=100.0, kd=2.0))
    st = state_np(eng)
    eng.close()
    finger = st[capi.S_LAM_FC] / h
    floor = st[capi.S_LAM_CF:capi.S_LAM_CF + 12:3].sum() / h
    return finger, floor, st


def _check_press_down(lib, device):
    finger, floor, st = _press_down(lib, device)
    weight = PR.CUBE_MASS * 9.81
    assert 0.8 < finger < 1.1, finger                                    # 100 N/m x ~ 9.5 mm
    assert abs(floor - (finger + weight)) < 2e-3 * (finger + weight), (floor, finger, weight)
    assert np.abs(st[capi.S_CUBE_V:capi.S_CUBE_V + 6]).max() < 1e-3 and abs(st[capi.S_CUBE_P + 2] - PR.CUBE_HALF) < 3e-4


def _push_and_slide(lib, device, speed=0.02):
    """finger 0 pushes the cube towards the centre of the arena at a constant 2 cm/s (the spring target leads the fingertip): Coulomb sliding on the floor"""
    eng = torque_engine(lib, device)
    f32 = dict(dtype=torch.float32, device=device)
    eng.cube[0:3, 0] = torch.tensor([0.0, 0.08, 0.0325], **f32)
    start = np.array([0.0, 0.08 + PR.CUBE_HALF + R_TIP + 0.001, 0.0325])
    eng.q[0:3, 0] = torch.tensor(ik(0, start), **f32)
    h, rows = 0.01, []
    for i in range(200):
        target = start - np.array([0.0, 0.004 + speed * 0.02 * i, 0.0])
        step_torque(eng, impedance_torques(state_np(eng), [target, None, None], kp=150.0, kd=3.0))
        st = state_np(eng)
        rows.append((st[capi.S_LAM_FC] / h, st[capi.S_LAM_FC + 3] / h, st[capi.S_LAM_CF:capi.S_LAM_CF + 12:3].sum() / h, st[capi.S_CUBE_V + 1],
                     st[capi.S_FC_LINK]))
    eng.close()
    return np.array(rows)


def _check_push_and_slide(lib, device):
    r = _push_and_slide(lib, device)
    tail = r[120:]                                                       # the last 1.6 s: steady sliding
    push, lift, floor, vy = tail[:, 0].mean(), tail[:, 1].mean(), tail[:, 2].mean(), tail[:, 3]
    weight = PR.CUBE_MASS * 9.81
    assert (tail[:, 4] == 3).all()                                       # the distal link holds the contact
    assert np.abs(vy + 0.02).max() < 1e-3                                # the cube moves at the commanded speed
    assert abs(push - PR.MU["cf"] * floor) < 0.03 * push, (push, floor)  # Coulomb: pushing force = mu x what the floor carries
    assert abs(floor - lift - weight) < 0.02 * weight, (floor, lift)     # vertical balance: the fingertip's friction carries a part of the weight
    assert lift < -0.1                                                   # (the stored friction impulse is the one on the FINGER: the cube gets its negative, upward)
    assert 0.2 < push < 0.45


def test_chain_steady_states_press_down_and_push(oracle):
    _check_press_down(oracle, "cpu")
    _check_push_and_slide(oracle, "cpu")


@pytest.mark.gpu
def test_chain_steady_states_press_down_and_push_gpu(hip):
    _check_press_down(hip, "cuda:0")
    _check_push_and_slide(hip, "cuda:0")


# ---- the boundary as the cube's corners see it ------------------------------------------------------------------------------------------
# The stage is a bowl (high_table_boundary.urdf:20-259): a vertical ring up to 32 mm, above it a cone that leans outward by 29-35 degrees.  The FINGERTIP
# contact follows the tilted surface normal (test above).  The CUBE CORNERS keep the horizontal normal of the ring at every height, with the radius of
# the profile at the corner's height (include/trifinger.h: TfModel.wall_r) - that is the model, and these two tests say what it
# costs: (i) the deviation itself, pinned: a corner that meets the cone is stopped radially and gets NO vertical deflection, where the surface normal
# would turn cos(a) sin(a) of the radial speed into upward speed; (ii) how often the workload puts a pushing corner on the cone at all: a cube that
# lies on the table touches the boundary with its lower corners, 32 mm below the first knot, so only a lifted or tumbling cube at the boundary is affected.
def _flying_cube_hits_the_cone(lib, device, v_r=0.3):
    """a cube without weight, 110 mm up, flies outward: its two lower outward corners (77.5 mm: second segment of the cone) meet the boundary"""
    def edit(m):
        m.cube_linear_damping = 0.0
        m.cube_angular_damping = 0.0
        m.mu_cube_wall = 0.0                                   # the normal row alone (the friction rows act along the tangent and the vertical)
    eng = T.engine(lib, device=device, model_edit=edit, gravity=(0.0, 0.0, 0.0), **T.HOLD)
    m = lib.default_model()
    f32 = dict(dtype=torch.float32, device=device)
    r_at = T.wall_radius_at(0.11 - 0.0325, m)
    eng.cube[0:3, 0] = torch.tensor([r_at - 0.0325 - 0.004, 0.0, 0.11], **f32)      # the corners 4 mm inside the profile: they arrive within the step
    eng.cube[7, 0] = v_r
    act = torch.tensor([[0.0, 0.9, -1.7] * 3], **f32)
    before = eng.cube[:, 0].cpu().numpy().astype(np.float64)
    eng.step(act)
    after = eng.cube[:, 0].cpu().numpy().astype(np.float64)
    lam = eng.state[capi.S_LAM_CW:capi.S_LAM_CW + 12, 0].cpu().numpy().astype(np.float64)
    eng.close()
    return before, after, lam, m


def _check_corner_on_cone(lib, device):
    v_r = 0.3
    before, after, lam, m = _flying_cube_hits_the_cone(lib, device, v_r)
    assert (lam[0::3] > 0).sum() == 2, lam                                 # the two lower outward corners pushed
    dv = after[7:10] - before[7:10]
    # the model: the boundary pushes HORIZONTALLY - the cube loses radial speed (and starts to pitch about the corners, which sit below its centre) ...
    assert dv[0] < -0.3 * v_r and abs(dv[1]) < 1e-5, dv
    # ... and gets no vertical impulse at all, where the tilted surface normal (c n_h, s) would give dv_z = -(s / c) dv_x: THE deviation of this model
    assert abs(dv[2]) < 1e-6, dv
    assert after[11] != 0.0                                                # pitch rate about y
    sl = (float(m.wall_r[2]) - float(m.wall_r[1])) / (float(m.wall_z[2]) - float(m.wall_z[1]))
    assert 0.5 < sl < 0.62                                                 # tan of the lean of this segment: the missing dv_z is 0.57 |dv_x|
    zc = before[2] - 0.0325
    assert float(m.wall_z[1]) < zc < float(m.wall_z[2])


def test_cube_corner_on_the_cone_keeps_the_horizontal_normal(oracle):
    _check_corner_on_cone(oracle, "cpu")


@pytest.mark.gpu
def test_cube_corner_on_the_cone_keeps_the_horizontal_normal_gpu(hip):
    _check_corner_on_cone(hip, "cuda:0")


def cone_corner_census(eng, m, steps, every, step_fn):
    """share of (env, sample) pairs with a boundary corner that pushes (normal impulse > 0) and of those whose pushing corner sits ABOVE the vertical ring"""
    corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64) * float(m.cube_half)
    wz, wr = np.array(m.wall_z[:], dtype=np.float64), np.array(m.wall_r[:], dtype=np.float64)
    n_samples = n_push = n_cone = 0
    for k in range(steps):
        step_fn(k)
        if k % every:
            continue
        st = eng.state.cpu().numpy().astype(np.float64)
        lam = st[capi.S_LAM_CW:capi.S_LAM_CW + 12:3]                       # normal impulses of the four corner slots
        push = (st[capi.S_CW_FACE] != 0) & (lam > 0).any(0)
        n_samples += st.shape[1]
        n_push += int(push.sum())
        for i in np.nonzero(push)[0]:
            c = st[capi.S_CUBE_P:capi.S_CUBE_P + 7, i]
            pts = c[0:3] + corners @ PR.quat_rot(c[3:7]).T
            rho = np.hypot(pts[:, 0], pts[:, 1])
            r_at = np.interp(pts[:, 2], wz, wr, left=wr[0])
            near = (r_at - rho) < 1.5e-3                                   # corners at the surface (the pushing ones are among them)
            n_cone += int((near & (pts[:, 2] > wz[0])).any())
    return n_samples, n_push, n_cone


def test_random_actions_rarely_put_a_pushing_corner_on_the_cone(oracle):
    """Census of the bench workload (difficulty 4, random actions, 750-step episodes) on the oracle: cubes do reach the boundary (a few per cent of the
    envs have a pushing corner at any time), and they do it lying on the table - the share of env-steps in which a pushing corner sits on the cone, where
    the horizontal-normal model deviates from the surface, is below 0.3 % of all env-steps (measured 0.0-0.1 %; tools/wall_census.py prints the same
    figure for 65536 envs on the GPU)."""
    import parity_util as pu
    from leibnizgym_amd.engine import TrifingerEngine, make_config
    n = 384
    kw = dict(pu.CONFIGS["d4_torque_asym"])
    eng = TrifingerEngine(make_config(oracle, n, seed=5, episode_length=750, **kw), device="cpu", lib=oracle)
    eng.reset()
    n_samples, n_push, n_cone = cone_corner_census(eng, oracle.default_model(), 900, 20, lambda k: eng.step_random())
    eng.close()
    assert n_push > 0.005 * n_samples, (n_push, n_samples)                 # the rollout did reach the boundary
    assert n_cone < 0.003 * n_samples, (n_cone, n_push, n_samples)
