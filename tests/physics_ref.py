"""Independent fp64 restatement of ONE solver substep of this build's physics spec.

TEST INFRASTRUCTURE.  Nothing here shares code with oracle/tf_oracle.c or the HIP kernels: the kinematic chain is the
fp64 URDF model of tests/test_physics_analytic.py, contact Jacobians are finite differences of that model's forward
kinematics, closest points come from scipy's bounded minimisers (not from the alternating projections of the C code),
and the mixed complementarity problem of the substep is solved by projected Gauss-Seidel iterated to a fixed point
(1e-13) instead of the 8 sweeps the product runs.  tests/test_contact_lcp_reference.py checks that the oracle's result
approaches this solution as its sweep count grows, on random contact configurations.

What is restated from the spec (these are modelling choices, so they are part of what is compared): the collision
primitives and their selection rules, the tangent basis of a normal, the speculative / Baumgarte / restitution bias of a
normal row, the friction pyramid, the joint-limit box rows, the finger-finger pre-solve on the free
velocities (the three distal pairs, then the six middle-distal pairs of the default model), damping factors and symplectic Euler.
"""
import numpy as np
from scipy.optimize import minimize, minimize_scalar

from test_physics_analytic import LINKS, frames, kinetic_matrix, potential, G  # noqa: F401  fp64 URDF model

# robot and object numbers: the reference's asset files (tests/golden/model.npz via tests/model_fixture.py); the build's own
# choices - collision capsules, boundary steps - are read from tf_default_model of the oracle library (pinned against the fixture
# by tests/test_model_fixture.py), so that this model follows the spec instead of restating its literals
import model_fixture as MF
from oracle_util import load_oracle

H_BASE = MF.H_BASE
YAW = MF.YAW
_M = load_oracle().default_model()
SHAPES = MF.model_shapes(_M)                # the collision shapes of the three links in test order (tests/model_fixture.py)
TIP_CAP = (3, np.array(list(_M.cap_a), dtype=np.float64), np.array(list(_M.cap_b), dtype=np.float64), float(_M.cap_radius))
UPPER_CHECK_Z = float(_M.upper_check_z)
MIDDLE_CHECK_Z = float(_M.middle_check_z)
CUBE_HALF = MF.CUBE_SIZE / 2.0
CUBE_MASS = MF.CUBE_DENSITY * MF.CUBE_SIZE ** 3
CUBE_INERTIA = CUBE_MASS * MF.CUBE_SIZE ** 2 / 6.0
LINK_DAMP, CUBE_LIN_DAMP, CUBE_ANG_DAMP = 0.01, 0.0, 0.05
MU = dict(fc=1.0, cf=0.55, tf=0.55, cw=1.0, tw=1.0, ff=1.0)
REST_F, REST_FF, BOUNCE = 0.4, 0.8, 0.5
MARGIN, SLACK, OFFSET, ERP, MAX_DEPEN = 0.04, 0.005, 0.002, 0.2, 1000.0
Q_LO = np.array([-0.33, 0.0, -2.7])
Q_HI = np.array([1.0, 1.57, 0.0])
QD_MAX = 10.0
FF_ITERATIONS = 4
WALL_R = tuple(float(x) for x in _M.wall_r)
WALL_Z = tuple(float(x) for x in _M.wall_z)


def link_rotation_world(f, q, link):
    return rot_z(YAW[f]) @ frames(q)[link - 1][0]


def shape_candidates(f, qf, cube_p, R, hc, links=(3, 2, 1), high=True):
    """every collision shape of finger f (of the given links) against the box at cube_p / R with half extents hc:
    [(gap, link, x, y, ext)] with x the axis point (or sphere centre) and y the closest point of the box, both in the box frame, and
    ext what lies between x and the shape's surface along the line to y.  A tapered rounded box (include/trifinger.h: TfLinkShape):
    closest points of its AXIS and the box, then the support function of its cross-section along the direction to the box."""
    out = []
    first_sphere_of_2 = next(i for i, e in enumerate(SHAPES) if e[0] == "sphere" and e[1] == 2)
    for i, entry in enumerate(SHAPES):
        link = entry[1]
        if link not in links:
            continue
        if not high and (link == 1 or i == first_sphere_of_2):      # the upper link and the joint-2 housing of the middle link hang at the
            continue                                                  # height of the base: only a cube above UPPER_CHECK_Z reaches them
        if link == 2 and entry[0] == "shape" and not (cube_p[2] + float(np.abs(R[2, :]) @ hc)) > MIDDLE_CHECK_Z:
            continue                                                  # the middle link stays >= 0.12 m above the floor
        if entry[0] == "sphere":
            c = R.T @ (link_point_world(f, qf, link, entry[2]) - cube_p)
            yb = np.clip(c, -hc, hc)
            d = np.linalg.norm(c - yb)
            if d <= 1e-6:                           # centre inside the cube: outside the domain of the reference (reported as a deep overlap)
                out.append((-1.0, link, c, yb, entry[3]))
                continue
            out.append((d - entry[3], link, c, yb, entry[3]))
            continue
        sh = entry[2]
        a = R.T @ (link_point_world(f, qf, link, sh["a"]) - cube_p)
        b = R.T @ (link_point_world(f, qf, link, sh["b"]) - cube_p)
        x, yb = segment_box(a, b, hc)
        D = np.linalg.norm(x - yb)
        if D <= 1e-6:                               # axis inside the cube: outside the domain of the reference (reported as a deep overlap)
            out.append((-1.0, link, x, yb, 0.0))
            continue
        spar = float(np.clip((x - a) @ (b - a) / ((b - a) @ (b - a)), 0.0, 1.0))
        u_link = link_rotation_world(f, qf, link).T @ (R @ ((yb - x) / D))       # from the axis point towards the cube, link frame
        ext = MF.shape_extent(sh, spar, u_link)
        out.append((D - ext, link, x, yb, ext))
    return out


def finger_gaps(f, qf, cube_p, R, hc, links=(3, 2, 1)):
    """[(gap, link)] of every collision shape of finger f (of the given links) against the box"""
    return [(g, lk) for g, lk, _, _, _ in shape_candidates(f, qf, cube_p, R, hc, links)]


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def quat_rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def link_point_world(f, q, link, local):
    R, p = frames(q)[link - 1]
    return rot_z(YAW[f]) @ (p + R @ local) + np.array([0.0, 0.0, H_BASE])


def point_jacobian(f, q, link, world_point):
    """d(world position)/dq of the material point of `link` that currently sits at world_point (central differences)."""
    R, p = frames(q)[link - 1]
    local = R.T @ (rot_z(YAW[f]).T @ (world_point - np.array([0.0, 0.0, H_BASE])) - p)
    J = np.zeros((3, 3))
    eps = 1e-6
    for j in range(3):
        dq = np.zeros(3)
        dq[j] = eps
        J[:, j] = (link_point_world(f, q + dq, link, local) - link_point_world(f, q - dq, link, local)) / (2 * eps)
    return J


def bias_forces(q, qd, gravity_z):
    """h(q, qd) = Mdot qd - 1/2 d(qd^T M qd)/dq + dV/dq, every derivative numerical."""
    eps = 1e-5
    dM = [(kinetic_matrix(q + eps * e) - kinetic_matrix(q - eps * e)) / (2 * eps) for e in np.eye(3)]
    Mdot = sum(dM[k] * qd[k] for k in range(3))
    dT = np.array([0.5 * qd @ dM[k] @ qd for k in range(3)])
    dV = np.array([(potential(q + eps * e) - potential(q - eps * e)) / (2 * eps) for e in np.eye(3)]) * (-gravity_z / G)
    return Mdot @ qd - dT + dV


def tangent_basis(n):
    if abs(n[2]) < 0.9:
        t1 = np.array([-n[1], n[0], 0.0]) / np.hypot(n[0], n[1])
    else:
        t1 = np.array([0.0, -n[2], n[1]]) / np.hypot(n[1], n[2])
    return t1, np.cross(n, t1)


def contact_bias(gap, vn0, h, restitution):
    b = gap / h if gap >= 0.0 else max(ERP * gap / h, -MAX_DEPEN)
    if restitution > 0.0 and gap < OFFSET and vn0 < -BOUNCE:
        b = min(b, restitution * vn0)
    return b


def contact_live(gap, vn0, h):
    """a slot gets rows when its gap can close within the substep at the free approach speed, plus a slack"""
    return gap < MARGIN and gap < SLACK + h * max(0.0, -vn0)


def wall_radius_at(z):
    """the boundary profile of the spec: piecewise linear through the knots (WALL_Z[i], WALL_R[i]), a vertical ring below the first
    knot, no wall above the last"""
    if not z < WALL_Z[-1]:
        return 1000.0
    return float(np.interp(z, WALL_Z, WALL_R))


def wall_tilt(z):
    """(cos, sin) of the slope angle of the profile segment at height z: the inward surface normal of the boundary is (cos * n_h, sin) with n_h the
    inward horizontal unit vector; (1, 0) on the vertical ring below the first knot.  Spec: the FINGERTIP contact uses the tilted normal and the
    distance to the tilted surface, the cube corners the horizontal normal."""
    i = int(np.searchsorted(WALL_Z, z, side="left")) - 1          # z > WALL_Z[i]
    if i < 0:
        return 1.0, 0.0
    i = min(i, len(WALL_Z) - 2)
    sl = (WALL_R[i + 1] - WALL_R[i]) / (WALL_Z[i + 1] - WALL_Z[i])
    return 1.0 / np.sqrt(1.0 + sl * sl), sl / np.sqrt(1.0 + sl * sl)


def segment_box(a, b, hc):
    """closest points of the segment a-b and the box [-hc, hc]^3 (box frame): bounded scalar minimisation."""
    def dist2(s):
        x = a + s * (b - a)
        return float(np.sum((x - np.clip(x, -hc, hc)) ** 2))
    best = None
    for lo, hi in ((0.0, 0.5), (0.5, 1.0), (0.0, 1.0)):
        r = minimize_scalar(dist2, bounds=(lo, hi), method="bounded", options={"xatol": 1e-13})
        if best is None or r.fun < best.fun:
            best = r
    for s_end in (0.0, 1.0):
        if dist2(s_end) < best.fun:
            best = type("R", (), {"x": s_end, "fun": dist2(s_end)})
    x = a + best.x * (b - a)
    return x, np.clip(x, -hc, hc)


def segment_segment(p1, q1, p2, q2):
    d1, d2 = q1 - p1, q2 - p2

    def fun(st):
        return float(np.sum((p1 + st[0] * d1 - p2 - st[1] * d2) ** 2))
    best = None
    for s0 in (0.0, 0.5, 1.0):
        for t0 in (0.0, 0.5, 1.0):
            r = minimize(fun, [s0, t0], bounds=[(0, 1), (0, 1)], method="L-BFGS-B", options={"ftol": 1e-16, "gtol": 1e-14})
            if best is None or r.fun < best.fun:
                best = r
    s, t = best.x
    return p1 + s * d1, p2 + t * d2


class Row:
    """One scalar constraint row over the 15 velocity dofs [9 joint, cube v, cube w]."""
    def __init__(self, J, kind, bias=0.0, parent=None, mu=0.0, lo=0.0, hi=0.0):
        self.J, self.kind, self.bias, self.parent, self.mu, self.lo, self.hi = J, kind, bias, parent, mu, lo, hi
        self.lam = 0.0


def box_spec(size, density, gyroscopic=True):
    """mass properties of a solid box of `size` (x, y, z) and `density`, written from the textbook formulas"""
    size = np.asarray(size, dtype=np.float64)
    mass = density * float(np.prod(size))
    inertia = mass / 12.0 * np.array([size[1] ** 2 + size[2] ** 2, size[0] ** 2 + size[2] ** 2, size[0] ** 2 + size[1] ** 2])
    return {"half": size / 2.0, "mass": mass, "inertia": inertia, "gyroscopic": gyroscopic}


def ref_substep(q, qd, cube, tau, h, gravity=(0.0, 0.0, -9.81), tol=1e-13, max_sweeps=200000, box=None, ff_middle=True):
    """One substep of length h in fp64.  q, qd, tau: (9,), cube: (13,) [p, quat xyzw, v, w].  `box`: a `box_spec` for a
    general box object (full world-frame inertia tensor here - the product's inertia-scaled coordinates are not used).
    Returns (qd_new (9,), cube_v (3,), cube_w (3,), details)."""
    q, qd, cube, tau = (np.asarray(a, dtype=np.float64) for a in (q, qd, cube, tau))
    cp, cq, cv, cw = cube[0:3], cube[3:7], cube[7:10], cube[10:13]
    gz = gravity[2]
    Minv = np.zeros((15, 15))
    vfree = np.zeros(15)
    for f in range(3):
        sl = slice(3 * f, 3 * f + 3)
        M = kinetic_matrix(q[sl])
        Mi = np.linalg.inv(M)
        Minv[sl, sl] = Mi
        acc = Mi @ (tau[sl] - bias_forces(q[sl], qd[sl], gz))
        vfree[sl] = (qd[sl] + h * acc) * (1.0 - h * LINK_DAMP)
    R = quat_rot(cq)
    if box is None:
        Minv[9:12, 9:12] = np.eye(3) / CUBE_MASS
        Minv[12:15, 12:15] = np.eye(3) / CUBE_INERTIA
        w_free = cw
        hc = np.full(3, CUBE_HALF)
    else:
        Minv[9:12, 9:12] = np.eye(3) / box["mass"]
        Minv[12:15, 12:15] = R @ np.diag(1.0 / box["inertia"]) @ R.T
        w_free = cw
        if box["gyroscopic"]:                       # Euler's equations, explicit step in the body frame
            wb = R.T @ cw
            wb = wb + h * np.cross(box["inertia"] * wb, wb) / box["inertia"]
            w_free = R @ wb
        hc = np.asarray(box["half"], dtype=np.float64)
    vfree[9:12] = (cv + h * np.asarray(gravity)) * (1.0 - h * CUBE_LIN_DAMP)
    vfree[12:15] = w_free * (1.0 - h * CUBE_ANG_DAMP)
    rows = []
    details = {"fc": [], "te": [], "ff": [], "ffm": [], "n_floor": 0, "n_wall": 0}

    def add_contact(Jn_dofs, dirs, gap, restitution, mu, vref):
        """three rows for the 3x15 map `Jn_dofs` (velocity of body A minus body B at the contact, world) and dirs n,t1,t2"""
        Jr = [d @ Jn_dofs for d in dirs]
        vn0 = float(Jr[0] @ vref)
        if not contact_live(gap, vn0, h):
            return None
        n_row = Row(Jr[0], "normal", bias=contact_bias(gap, vn0, h, restitution))
        rows.append(n_row)
        rows.append(Row(Jr[1], "tangent", parent=n_row, mu=mu))
        rows.append(Row(Jr[2], "tangent", parent=n_row, mu=mu))
        return n_row

    def cube_map(r):      # velocity of the cube point with arm r: v + w x r
        Jc = np.zeros((3, 15))
        Jc[:, 9:12] = np.eye(3)
        Jc[:, 12:15] = -np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
        return Jc

    tipsphere = []
    distal = []
    for f in range(3):
        sl = slice(3 * f, 3 * f + 3)
        a3 = link_point_world(f, q[sl], 3, TIP_CAP[1])
        b3 = link_point_world(f, q[sl], 3, TIP_CAP[2])
        distal.append((a3, b3))
        tipsphere.append(b3)
    # ---- finger-finger pre-pass on the free velocities: pairs in turn, one frictionless normal row each ----
    v_ff = vfree.copy()
    for p in range(3):
        fa, fb = p, (p + 1) % 3
        Pa, Pb = segment_segment(distal[fa][0], distal[fa][1], distal[fb][0], distal[fb][1])
        dist = np.linalg.norm(Pa - Pb)
        if not dist > 1e-6:
            continue
        rad = TIP_CAP[3]
        gap = dist - 2 * rad
        if not gap < MARGIN:
            continue
        n = (Pa - Pb) / dist
        t1, t2 = tangent_basis(n)
        Jm = np.zeros((3, 15))
        Jm[:, 3 * fa:3 * fa + 3] = point_jacobian(fa, q[3 * fa:3 * fa + 3], 3, Pa - rad * n)
        Jm[:, 3 * fb:3 * fb + 3] = -point_jacobian(fb, q[3 * fb:3 * fb + 3], 3, Pb + rad * n)
        Jr = n @ Jm
        vn0 = float(Jr @ v_ff)
        if not contact_live(gap, vn0, h):
            continue
        lam_n = max(-(vn0 + contact_bias(gap, vn0, h, REST_FF)) / float(Jr @ Minv @ Jr), 0.0)     # frictionless: one row
        v_ff = v_ff + Minv @ Jr * lam_n
        details["ff"].append((fa, fb, gap, lam_n))
    # ---- ... then the middle link of every finger against the fingertip capsule of each other finger (the default model since API 8: the reference
    # keeps all robot links in one self-colliding group): six ordered pairs (fm; fm + 1), (fm; fm + 2), one frictionless row each, every one solved
    # on the FREE velocities (a Jacobi step: the rows see neither each other nor the distal pairs) and its velocity changes added to what the distal
    # pairs left.  The middle link is its finger-cube shape: closest points of its axis and the capsule's axis, its support function along the
    # line between them; the row moves joints 1 and 2 of the middle finger (the point is a point of link 2) and all three of the distal one ----
    if ff_middle:
        sh2 = next(e[2] for e in SHAPES if e[0] == "shape" and e[1] == 2)
        for o, fm in ((2, 0), (2, 1), (2, 2), (1, 0), (1, 1), (1, 2)):         # the order the spec lists them in (it only orders the sums)
            qm = q[3 * fm:3 * fm + 3]
            am, bm = link_point_world(fm, qm, 2, sh2["a"]), link_point_world(fm, qm, 2, sh2["b"])
            if True:
                fd = (fm + o) % 3
                Pm, Pd = segment_segment(am, bm, distal[fd][0], distal[fd][1])
                dist = np.linalg.norm(Pd - Pm)
                if not dist > 1e-6:
                    continue
                n = (Pd - Pm) / dist                                    # from the middle link to the capsule
                sp = float(np.clip((Pm - am) @ (bm - am) / ((bm - am) @ (bm - am)), 0.0, 1.0))
                ext = MF.shape_extent(sh2, sp, link_rotation_world(fm, qm, 2).T @ n)
                gap = dist - ext - TIP_CAP[3]
                if not gap < MARGIN:
                    continue
                Jm_ = np.zeros((3, 15))
                Jm_[:, 3 * fd:3 * fd + 3] = point_jacobian(fd, q[3 * fd:3 * fd + 3], 3, Pd - TIP_CAP[3] * n)
                Jm_[:, 3 * fm:3 * fm + 3] = -point_jacobian(fm, qm, 2, Pm + ext * n)
                Jr = n @ Jm_
                vn0 = float(Jr @ vfree)
                if not contact_live(gap, vn0, h):
                    continue
                lam_n = max(-(vn0 + contact_bias(gap, vn0, h, REST_FF)) / float(Jr @ Minv @ Jr), 0.0)
                v_ff = v_ff + Minv @ Jr * lam_n
                details["ffm"].append((fm, fd, gap, lam_n))
    # ---- finger contacts ----
    for f in range(3):
        sl = slice(3 * f, 3 * f + 3)
        qf = q[sl]
        best = None
        for gap_c, cand, xc, yc, ext in shape_candidates(f, qf, cp, R, hc, links=(3, 2, 1), high=cp[2] > UPPER_CHECK_Z):
            if best is None or gap_c < best[0]:       # the shape with the smallest gap holds the contact (first wins a tie)
                best = (gap_c, cand, xc, yc, ext)
        gap, link, x, y, rad = best
        if gap <= -1.0:
            raise ValueError("a shape's axis lies inside the cube: outside the domain of the reference")
        if gap < MARGIN:
            n = R @ ((x - y) / np.linalg.norm(x - y))
            t1, t2 = tangent_basis(n)
            Pw = cp + R @ x - rad * n
            rc = R @ y
            Jm = -cube_map(rc)
            Jm[:, sl] = point_jacobian(f, qf, link, Pw)
            nr = add_contact(Jm, (n, t1, t2), gap, REST_F, MU["fc"], vfree)
            if nr is not None:
                details["fc"].append((f, link, gap, nr))
        # fingertip sphere against the floor and against the boundary wall
        B = tipsphere[f]
        rad = TIP_CAP[3]
        rho = np.hypot(B[0], B[1])
        for kind in ("floor", "wall"):
            if kind == "floor":
                gp, n = B[2] - rad, np.array([0.0, 0.0, 1.0])
            else:
                if not rho > 1e-6:
                    continue
                wc, wsn = wall_tilt(B[2])
                gp, n = (wall_radius_at(B[2]) - rho) * wc - rad, np.array([-B[0] / rho * wc, -B[1] / rho * wc, wsn])
            if gp < MARGIN:
                t1, t2 = tangent_basis(n)
                Jm = np.zeros((3, 15))
                Jm[:, sl] = point_jacobian(f, qf, 3, B - rad * n)
                nr = add_contact(Jm, (n, t1, t2), gp, REST_F, MU["tf"] if kind == "floor" else MU["tw"], vfree)
                if nr is not None:
                    details["te"].append((f, kind, gp, nr))
    # ---- cube corners against the floor ----
    k = int(np.argmax(np.abs(R[2, :]) * hc))       # the face holding the four lowest corners
    sk = -1.0 if R[2, k] > 0 else 1.0
    others = [i for i in range(3) if i != k]
    for idx in range(4):
        yv = np.zeros(3)
        yv[k] = sk * hc[k]
        yv[others[0]] = hc[others[0]] if idx & 1 else -hc[others[0]]
        yv[others[1]] = hc[others[1]] if idx & 2 else -hc[others[1]]
        r = R @ yv
        gap = cp[2] + r[2]
        if gap < MARGIN:
            Jm = cube_map(r)
            ez, ex, ey = np.eye(3)[2], np.eye(3)[0], np.eye(3)[1]
            if add_contact(Jm, (ez, ex, ey), gap, 0.0, MU["cf"], vfree) is not None:
                details["n_floor"] += 1
    # ---- cube corners against the boundary wall ----
    rho_c = np.hypot(cp[0], cp[1])
    if rho_c > 1e-6:
        dvec = np.array([cp[0] / rho_c, cp[1] / rho_c, 0.0])
        pr = R.T @ dvec
        k = int(np.argmax(np.abs(pr) * hc))
        sk = -1.0 if pr[k] < 0 else 1.0
        others = [i for i in range(3) if i != k]
        for idx in range(4):
            yv = np.zeros(3)
            yv[k] = sk * hc[k]
            yv[others[0]] = hc[others[0]] if idx & 1 else -hc[others[0]]
            yv[others[1]] = hc[others[1]] if idx & 2 else -hc[others[1]]
            r = R @ yv
            P = cp + r
            rho = np.hypot(P[0], P[1])
            gap = wall_radius_at(P[2]) - rho
            if gap < MARGIN and rho > 1e-6:
                n = np.array([-P[0] / rho, -P[1] / rho, 0.0])
                t = np.array([-n[1], n[0], 0.0])
                if add_contact(cube_map(r), (n, t, np.array([0.0, 0.0, 1.0])), gap, 0.0, MU["cw"], vfree) is not None:
                    details["n_wall"] += 1
    # ---- joint / velocity limit rows ----
    for j in range(9):
        Jr = np.zeros(15)
        Jr[j] = 1.0
        lo = float(np.clip((Q_LO[j % 3] - q[j]) / h, -QD_MAX, QD_MAX))
        hi = float(np.clip((Q_HI[j % 3] - q[j]) / h, -QD_MAX, QD_MAX))
        rows.append(Row(Jr, "limit", lo=lo, hi=hi))
    # ---- projected Gauss-Seidel to a fixed point ----
    v = v_ff.copy()
    W = [Minv @ r.J for r in rows]
    D = [float(r.J @ w) for r, w in zip(rows, W)]
    sweeps = 0
    for sweeps in range(1, max_sweeps + 1):
        change = 0.0
        for r, w, d in zip(rows, W, D):
            if d <= 0.0:
                continue
            vrel = float(r.J @ v)
            if r.kind == "normal":
                new = max(r.lam - (vrel + r.bias) / d, 0.0)
            elif r.kind == "tangent":
                lim = r.mu * r.parent.lam
                new = float(np.clip(r.lam - vrel / d, -lim, lim))
            else:
                v0 = vrel - d * r.lam
                new = (float(np.clip(v0, r.lo, r.hi)) - v0) / d
            dl = new - r.lam
            if dl != 0.0:
                v = v + w * dl
                r.lam = new
                change = max(change, abs(dl) * np.sqrt(d))
        if change < tol:
            break
    details["sweeps"] = sweeps
    details["rows"] = rows
    details["Minv"] = Minv                    # (developer experiments, tests/dev: the substep's contact problem as data)
    details["v_start"] = v_ff
    return v[0:9].copy(), v[9:12].copy(), v[12:15].copy(), details
