"""The physical spec pinned to the reference's ASSET FILES (tests/golden/model.npz, generated from trifingerpro.urdf, the link and
boundary meshes and the object URDFs by tests/golden/make_model_golden.py - VERDICT round 2, item 1):

  * every URDF-derived field of `tf_default_model` equals the fixture (exact to fp32 rounding; the distal link = lower link + tip
    link merged in fp64 by the generator, independently of the C code);
  * the build's collision primitives COVER the convex hulls the reference loads: no point of a link's hull surface lies more than
    3 mm outside the union of its capsules (and the over-coverage the capsule family costs is stated, not hidden);
  * the boundary steps lie inside the band of the 40 convex pieces' inner surface;
  * both libraries ship the same model (CPU: oracle; `-m gpu`: the HIP library, whose default model needs no device)."""
import numpy as np
import pytest
from scipy.spatial import ConvexHull

import model_fixture as MF
from leibnizgym_amd import _capi as capi

F = MF.FIXTURE


def _check_urdf_numbers(m):
    f32 = lambda x: np.asarray(x, dtype=np.float64).astype(np.float32)      # noqa: E731
    assert np.float32(m.base_height) == f32(F["base_height"])
    yaw = F["finger_yaw"]
    assert np.allclose(list(m.base_yaw_cos), np.cos(yaw), atol=1e-7) and np.allclose(list(m.base_yaw_sin), np.sin(yaw), atol=1e-7)
    assert np.allclose(list(m.base_half_yaw_cos), np.cos(yaw / 2), atol=1e-7) and np.allclose(list(m.base_half_yaw_sin), np.sin(yaw / 2), atol=1e-7)
    assert np.array_equal(np.array(list(m.j2_origin), dtype=np.float32), f32(F["j2_origin"]))
    assert np.array_equal(np.array(list(m.j3_origin), dtype=np.float32), f32(F["j3_origin"]))
    assert np.array_equal(np.array(list(m.tip_origin), dtype=np.float32), f32(F["tip_origin"]))
    assert np.allclose(F["j1_origin"], 0) and tuple(F["j1_axis"]) == (0, 1, 0) and tuple(F["j2_axis"]) == (1, 0, 0) and tuple(F["j3_axis"]) == (1, 0, 0)
    # upper and middle link inertials verbatim; the distal body = lower + tip merged (generator, fp64)
    for i in range(2):
        assert np.float32(m.link_mass[i]) == f32(F["link_mass"][i])
        assert np.array_equal(np.array(list(m.link_com[i]), dtype=np.float32), f32(F["link_com"][i]))
        assert np.array_equal(np.array(list(m.link_inertia[i]), dtype=np.float32), f32(F["link_inertia"][i]))
    assert np.isclose(m.link_mass[2], float(F["distal_mass"]), rtol=2e-7)
    assert np.allclose(list(m.link_com[2]), F["distal_com"], rtol=2e-7, atol=1e-10)
    assert np.allclose(list(m.link_inertia[2]), F["distal_inertia"], rtol=3e-7, atol=1e-12)
    # object: cube_multicolor_rrc.urdf
    s, rho = float(F["cube_size"][0]), float(F["cube_density"])
    assert np.all(F["cube_size"] == s)
    assert np.float32(m.cube_half) == np.float32(s / 2) and np.isclose(m.cube_mass, rho * s ** 3, rtol=2e-7)
    assert np.isclose(m.cube_inertia, rho * s ** 5 / 6.0, rtol=2e-7)
    # fingertip sphere: least-squares sphere of SIM__BL-Finger_Tip_actual_tip.obj in the tip-link frame
    centre = F["tip_origin"] + F["tip_sphere_centre"]
    assert np.allclose(list(m.cap_b), centre, atol=1.5e-4) and abs(m.cap_radius - float(F["tip_sphere_radius"])) < 1e-4
    assert float(F["tip_sphere_residual"]) < 4e-4


def test_urdf_numbers_oracle(oracle):
    _check_urdf_numbers(oracle.default_model())


def test_phase3_object_numbers(oracle):
    m = oracle.box_model(MF.PHASE3_SIZE, MF.PHASE3_DENSITY)
    assert np.allclose(list(m.box_half), np.array(MF.PHASE3_SIZE) / 2, rtol=1e-6)
    assert np.isclose(m.cube_mass, MF.PHASE3_DENSITY * np.prod(MF.PHASE3_SIZE), rtol=1e-6)


# ---- coverage of the collision hulls by the capsules -----------------------------------------------------------------------
def _surface(V, n, seed):
    h = ConvexHull(V)
    rng = np.random.default_rng(seed)
    tri = V[h.simplices]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    idx = rng.choice(len(tri), n, p=area / area.sum())
    u = rng.random((n, 2))
    flip = u.sum(1) > 1
    u[flip] = 1 - u[flip]
    P = tri[idx, 0] + u[:, :1] * (tri[idx, 1] - tri[idx, 0]) + u[:, 1:] * (tri[idx, 2] - tri[idx, 0])
    return np.vstack([P, V[h.vertices]]), h


def _inside(P, h):
    return (P @ h.equations[:, :3].T + h.equations[:, 3]).max(1)


S_GRID = np.linspace(0.0, 1.0, 201)


def _shape_dist(P, sh):
    """signed distance of points P (link frame) to a tapered rounded box: min over the axis parameter s of
    |(planar excess over the core rectangle, offset along the axis)| - rho(s)"""
    a, b, e1, e2 = sh["a"], sh["b"], sh["e1"], sh["e2"]
    d = b - a
    ax = d / np.linalg.norm(d)
    out = np.full(len(P), np.inf)
    for s in S_GRID:
        lerp = lambda p: p[0] + s * (p[1] - p[0])      # noqa: E731
        rho = lerp(sh["rho"])
        rel = P - (a + s * d) - lerp(sh["o1"]) * e1 - lerp(sh["o2"]) * e2
        q1 = np.maximum(np.abs(rel @ e1) - (lerp(sh["w1"]) - rho), 0.0)
        q2 = np.maximum(np.abs(rel @ e2) - (lerp(sh["w2"]) - rho), 0.0)
        ua = rel @ ax
        out = np.minimum(out, np.sqrt(q1 * q1 + q2 * q2 + ua * ua) - rho)
    return out


def _shape_surface(sh, n, rng):
    s, th = rng.random(n), rng.random(n) * 2 * np.pi
    u1, u2 = np.cos(th), np.sin(th)
    lerp = lambda p: p[0] + s * (p[1] - p[0])      # noqa: E731
    rho = lerp(sh["rho"])
    p1 = np.sign(u1) * (lerp(sh["w1"]) - rho) + rho * u1 + lerp(sh["o1"])
    p2 = np.sign(u2) * (lerp(sh["w2"]) - rho) + rho * u2 + lerp(sh["o2"])
    return sh["a"] + s[:, None] * (sh["b"] - sh["a"]) + p1[:, None] * sh["e1"] + p2[:, None] * sh["e2"], s


def coverage(m):
    """per link: (largest distance of a hull VERTEX outside the link's shapes, of any hull SURFACE point; largest distance of a point
    of the tapered box's side surface outside the hull; the same for the housing spheres)"""
    shapes = MF.model_shapes(m)
    out = {}
    bodies = {1: [F["hull_upper"]], 2: [F["hull_middle"]], 3: [F["hull_lower"], F["hull_tip_in_lower"]]}
    rng = np.random.default_rng(3)
    for link, hull_sets in bodies.items():
        mine = [e for e in shapes if e[1] == link]
        assert mine and mine[0][0] == "shape", f"link {link} has no collision shape"
        hulls, S_all, V_all = [], [], []
        for i, V in enumerate(hull_sets):
            V = V.astype(np.float64)
            S, h = _surface(V, 20000, i)
            hulls.append(h); S_all.append(S); V_all.append(V)
        keepS, keepV = [], []
        for i in range(len(hulls)):                    # surface of the union of the hulls of the body
            ks, kv = np.ones(len(S_all[i]), bool), np.ones(len(V_all[i]), bool)
            for j, h in enumerate(hulls):
                if j != i:
                    ks &= _inside(S_all[i], h) > -1e-9
                    kv &= _inside(V_all[i], h) > -1e-9
            keepS.append(S_all[i][ks]); keepV.append(V_all[i][kv])
        S, Vv = np.vstack(keepS), np.vstack(keepV)

        def dist(P):
            d = _shape_dist(P, mine[0][2])
            for e in mine[1:]:
                d = np.minimum(d, np.linalg.norm(P - e[2], axis=1) - e[3])
            return d
        Q, _ = _shape_surface(mine[0][2], 20000, rng)
        over_body = np.min([_inside(Q, h) for h in hulls], axis=0).max()
        over_sph = 0.0
        for e in mine[1:]:
            v = rng.normal(size=(4000, 3))
            Qs = e[2] + e[3] * v / np.linalg.norm(v, axis=1)[:, None]
            over_sph = max(over_sph, np.min([_inside(Qs, h) for h in hulls], axis=0).max())
        out[link] = (dist(Vv).max(), dist(S).max(), over_body, over_sph)
    return out


def _check_coverage(m):
    cov = coverage(m)
    for link, (vert_out, surf_out, over_body, over_sph) in cov.items():
        print(f"link {link}: hull vertices at most {vert_out * 1e3:.2f} mm outside the shape, hull surface {surf_out * 1e3:.2f} mm; "
              f"tapered box at most {over_body * 1e3:.2f} mm outside the hull, housing spheres {over_sph * 1e3:.2f} mm")
        assert vert_out <= 0.003, (link, vert_out)        # the bar of the review: no hull vertex more than 3 mm outside
        assert surf_out <= 0.0031, (link, surf_out)       # and no point of its faces either (sampled, 20000 points per hull)
    # what the shape family costs the other way round is bounded and stated: the body of the distal link follows its
    # hull to 4 mm, the middle link to 8 mm; the housing spheres bulge by up to 16 mm over the flat faces of the joint housings
    assert cov[3][2] <= 0.0045 and cov[2][2] <= 0.0085 and cov[1][2] <= 0.0095, cov
    assert max(c[3] for c in cov.values()) <= 0.017, cov
    return cov


def test_shapes_cover_the_collision_hulls(oracle):
    _check_coverage(oracle.default_model())


def test_fingertip_region_is_exact(oracle):
    """The distal shape has the fingertip capsule's axis and ends in the fingertip sphere; over the last 4 cm of the distal body (where
    nearly every contact of the task happens) its surface is within 2.5 mm of the hull BOTH ways."""
    m = oracle.default_model()
    sh = MF.model_shapes(m)[0][2]
    assert np.array_equal(sh["a"], np.array(list(m.cap_a), dtype=np.float64)) and np.array_equal(sh["b"], np.array(list(m.cap_b), dtype=np.float64))
    r = float(m.cap_radius)
    assert sh["w1"][1] == r and sh["w2"][1] == r and sh["rho"][1] == r and sh["o1"][1] == 0.0 and sh["o2"][1] == 0.0
    hulls = [ConvexHull(F["hull_lower"].astype(np.float64)), ConvexHull(F["hull_tip_in_lower"].astype(np.float64))]
    Q, s = _shape_surface(sh, 40000, np.random.default_rng(0))
    near_tip = Q[:, 2] < -0.12
    assert np.min([_inside(Q[near_tip], h) for h in hulls], axis=0).max() <= 0.0025
    S = np.vstack([_surface(F["hull_lower"].astype(np.float64), 20000, 0)[0], _surface(F["hull_tip_in_lower"].astype(np.float64), 20000, 1)[0]])
    S = S[S[:, 2] < -0.12]
    v = np.array(list(m.cap_b), dtype=np.float64)
    d = np.minimum(_shape_dist(S, sh), np.linalg.norm(S - v, axis=1) - r)
    assert d.max() <= 0.0025, d.max()


def test_shape_gap_is_close_to_its_minimum_over_the_axis():
    """The gap of a link shape is evaluated at the point of the axis that is closest to the cube (one closed-form query), not minimised
    over the axis: with the taper the two can differ.  Over random near-contact poses of the distal body: median < 0.1 mm, p99 < 4 mm,
    never more than 6 mm; an overlap that the one-point rule does not see is never deeper than 4 mm."""
    import physics_ref as PR
    rng = np.random.default_rng(0)
    sh = [e for e in PR.SHAPES if e[0] == "shape" and e[1] == 3][0][2]
    hc = np.full(3, PR.CUBE_HALF)
    diff, hidden = [], 0.0
    while len(diff) < 300:
        q = rng.uniform(PR.Q_LO + 0.05, PR.Q_HI - 0.05)
        mid = PR.link_point_world(0, q, 3, sh["a"] + rng.uniform(0, 1) * (sh["b"] - sh["a"]))
        v = rng.normal(size=4)
        R = PR.quat_rot(v / np.linalg.norm(v))
        d = rng.normal(size=3)
        c = mid + d / np.linalg.norm(d) * rng.uniform(0.035, 0.075)
        g_model = PR.shape_candidates(0, q, c, R, hc, links=(3,))[0][0]
        if not -0.004 < g_model < 0.03:
            continue
        a = R.T @ (PR.link_point_world(0, q, 3, sh["a"]) - c)
        b = R.T @ (PR.link_point_world(0, q, 3, sh["b"]) - c)
        Rl = PR.link_rotation_world(0, q, 3)
        best = np.inf
        for t in np.linspace(0, 1, 101):
            x = a + t * (b - a)
            y = np.clip(x, -hc, hc)
            D = np.linalg.norm(x - y)
            if D < 1e-6:
                best = -1.0
                break
            best = min(best, D - MF.shape_extent(sh, t, Rl.T @ (R @ ((y - x) / D))))
        if best <= -1.0:
            continue
        diff.append(g_model - best)
        if best < 0.0 < g_model:
            hidden = max(hidden, -best)
    diff = np.array(diff)
    assert np.median(diff) < 1e-4 and np.percentile(diff, 99) < 4e-3 and diff.max() < 6e-3, (np.median(diff), np.percentile(diff, 99), diff.max())
    assert hidden < 4e-3, hidden


def test_only_the_fingertip_can_reach_floor_and_boundary(oracle):
    """The step gives floor and boundary contacts to the fingertip sphere only (review round 2, item 8: "non-tip links vs floor and
    boundary").  With the reference's kinematics and joint limits nothing else CAN touch them: over the whole joint range the upper
    link, the middle link and the joint housings stay more than 8 cm above the floor and more than 5 cm inside the boundary, and
    wherever a point of the distal body comes within 3 mm of the floor the fingertip sphere is at least 8 mm lower - the fingertip
    contact stops the finger first.  Against the BOUNDARY the same holds to within 3 mm only: in outstretched poses the thick part
    of the distal body can reach the flared wall about as early as the fingertip (not modelled: DESIGN.md section 10)."""
    import physics_ref as PR
    from test_physics_analytic import frames
    m = oracle.default_model()
    wr, wz = np.array(list(m.wall_r)), np.array(list(m.wall_z))
    wall = lambda z: float(np.interp(z, wz, wr)) if z < wz[-1] else 1e3      # noqa: E731
    rng = np.random.default_rng(1)
    low = {1: 9.0, 2: 9.0, "housing": 9.0}
    wgap = {1: 9.0, 2: 9.0, "housing": 9.0}
    floor_margin, wall_margin = 9.0, 9.0
    ss = np.linspace(0.0, 1.0, 11)
    for _ in range(4000):
        q = rng.uniform(PR.Q_LO, PR.Q_HI)
        fr = frames(q)
        for e in PR.SHAPES:
            link = e[1]
            Rl, pl = fr[link - 1]
            if e[0] == "sphere":
                c = pl + Rl @ e[2]
                low["housing"] = min(low["housing"], c[2] + PR.H_BASE - e[3])
                wgap["housing"] = min(wgap["housing"], wall(c[2] + PR.H_BASE) - (np.hypot(c[0], c[1]) + e[3]))
                continue
            sh = e[2]
            zl, wl = [], []
            for t in ss:
                x = pl + Rl @ (sh["a"] + t * (sh["b"] - sh["a"]))
                zc = x[2] + PR.H_BASE
                rn = np.hypot(x[0], x[1])
                rdir = np.array([x[0], x[1], 0.0]) / rn if rn > 1e-9 else np.array([1.0, 0.0, 0.0])
                zl.append(zc - MF.shape_extent(sh, t, Rl.T @ np.array([0.0, 0.0, -1.0])))
                wl.append(wall(zc) - (rn + MF.shape_extent(sh, t, Rl.T @ rdir)))
            if link != 3:
                low[link] = min(low[link], min(zl))
                wgap[link] = min(wgap[link], min(wl))
            else:                                   # distal body: s <= 0.9 against the fingertip sphere (s = 1)
                if min(zl[:-1]) < 0.003:
                    floor_margin = min(floor_margin, min(zl[:-1]) - zl[-1])
                if min(wl[:-1]) < 0.003:
                    wall_margin = min(wall_margin, min(wl[:-1]) - wl[-1])
    assert low[1] > 0.25 and low[2] > 0.11 and low["housing"] > 0.08, low
    assert low[2] - float(m.contact_margin) - 0.004 >= float(m.middle_check_z)      # what the middle-link height gate of the step relies on (4 mm: base-offset DR)
    assert wgap[1] > 0.06 and wgap[2] > 0.06 and wgap["housing"] > 0.05, wgap
    assert floor_margin > 0.008 and wall_margin > -0.003, (floor_margin, wall_margin)


def test_boundary_profile_lies_in_the_band_of_the_convex_pieces(oracle):
    """high_table_boundary.urdf loads 40 convex pieces; each spans up to ~50 degrees of arc, so the inner surface is polygonal:
    at height z its distance to the axis runs from the chord value `boundary_profile_r` (fixture) to that value / cos(25 deg) at
    the piece corners.  The build's profile r(z) - a vertical ring, then the cone of the stage, piecewise linear through four knots -
    must lie inside that band (1.5 mm tolerance) at every height."""
    m = oracle.default_model()
    assert int(F["boundary_num_pieces"]) == 40
    z, lo = F["boundary_profile_z"], F["boundary_profile_r"]
    hi = lo / np.cos(np.radians(25.0))
    wr, wz = list(m.wall_r), list(m.wall_z)
    assert abs(wz[3] - float(F["boundary_z_range"][1])) < 2e-3
    worst = 0.0
    for zz, l, h in zip(z, lo, hi):
        if zz <= 0.0 or zz >= wz[3]:
            continue
        r = float(np.interp(zz, wz, wr))
        assert l - 0.0015 <= r <= h + 0.0015, (zz, r, l, h)
        worst = max(worst, abs(r - 0.5 * (l + h)))
    assert worst < 0.010          # never more than 1 cm from the middle of the band (the band itself is ~17 mm wide: chord sag)


def test_both_libraries_ship_the_same_model(oracle):
    hip = capi.TfLib(capi.hip_library_path())              # loading and tf_default_model need no device
    a, b = hip.default_model(), oracle.default_model()
    import ctypes
    assert bytes(ctypes.string_at(ctypes.byref(a), ctypes.sizeof(a))) == bytes(ctypes.string_at(ctypes.byref(b), ctypes.sizeof(b)))
    _check_urdf_numbers(a)


@pytest.mark.gpu
def test_urdf_numbers_and_coverage_hip(hip):
    m = hip.default_model()
    _check_urdf_numbers(m)
    _check_coverage(m)


# ---- the contact geometry of the step against the reference's collision hulls themselves -----------------------------------------
def _hull_box_distance(V, hc, iters=400):
    """distance between the convex hull of the points V and the box [-hc, hc]^3 (both in the box frame): Frank-Wolfe with exact line
    search on p -> dist(p, box)^2 over the hull (the linear sub-problem is an argmin over the hull vertices); returns 0 on overlap"""
    p = V[np.argmin(np.linalg.norm(V, axis=1))].copy()
    for _ in range(iters):
        g = p - np.clip(p, -hc, hc)
        if g @ g < 1e-16:
            return 0.0
        v = V[np.argmin(V @ g)]
        d = v - p
        if -(g @ d) < 1e-12:
            break
        # exact line search on the segment p + t d: the squared distance to the box is piecewise quadratic and convex in t
        lo, hi = 0.0, 1.0
        for _ in range(40):
            m1, m2 = lo + (hi - lo) / 3, hi - (hi - lo) / 3
            f1 = np.sum((p + m1 * d - np.clip(p + m1 * d, -hc, hc)) ** 2)
            f2 = np.sum((p + m2 * d - np.clip(p + m2 * d, -hc, hc)) ** 2)
            if f1 < f2:
                hi = m2
            else:
                lo = m1
        p = p + 0.5 * (lo + hi) * d
    return float(np.linalg.norm(p - np.clip(p, -hc, hc)))


def test_contact_gap_against_the_distance_of_the_reference_hulls():
    """What the step takes for the gap between the distal body and the cube - one axis-box query and the support function of the fitted
    cross-section, or the housing sphere - against the true distance between the REFERENCE's collision hulls (lower link + fingertip,
    tests/golden/model.npz) and the cube, over random near-contact poses: the geometric error of the contact model, end to end.
    Median below 1.5 mm, 90 % within 3.5 mm, never more than 9 mm (the bulge of the housing sphere is the worst case)."""
    import physics_ref as PR
    rng = np.random.default_rng(4)
    hulls = [F["hull_lower"].astype(np.float64), F["hull_tip_in_lower"].astype(np.float64)]
    hc = np.full(3, PR.CUBE_HALF)
    err = []
    while len(err) < 80:
        q = rng.uniform(PR.Q_LO + 0.05, PR.Q_HI - 0.05)
        on_link = rng.uniform(0, 1) * np.array([0.005, 0.0, -0.1592]) + np.array([0.0135, 0.0, 0.0])
        v = rng.normal(size=4)
        R = PR.quat_rot(v / np.linalg.norm(v))
        d = rng.normal(size=3)
        c = PR.link_point_world(0, q, 3, on_link) + d / np.linalg.norm(d) * rng.uniform(0.035, 0.08)
        cands = PR.shape_candidates(0, q, c, R, hc, links=(3,))
        g_model = min(x[0] for x in cands)
        if not 0.0005 < g_model < 0.02:
            continue
        Rl, pl = PR.link_rotation_world(0, q, 3), PR.link_point_world(0, q, 3, np.zeros(3))
        d_true = min(_hull_box_distance((R.T @ ((Rl @ H.T).T + pl - c).T).T, hc) for H in hulls)
        err.append(g_model - d_true)
    err = np.array(err)
    print("\ncontact-model gap minus hull distance [mm]: median %.2f  p10 %.2f  p90 %.2f  min %.2f  max %.2f" % (
        np.median(err) * 1e3, np.percentile(err, 10) * 1e3, np.percentile(err, 90) * 1e3, err.min() * 1e3, err.max() * 1e3))
    assert np.median(np.abs(err)) < 1.5e-3 and np.percentile(np.abs(err), 90) < 3.5e-3 and np.abs(err).max() < 9e-3, err


def test_finger_cube_gap_of_the_engine_against_the_independent_geometry(oracle):
    """a STATIC check of the contact geometry, no dynamics - the gap and the link of the finger-cube contact candidate exactly as the
    substep selects it (oracle entry tfo_finger_gap, the function substep() calls; the HIP kernels are bit-identical to it) against
    tests/physics_ref.shape_candidates (fp64, scipy closest points, the support function written from the spec) on sampled poses with the cube
    within a few centimetres of the finger.  A wrong sign in the support function, a swapped width direction or a wrong link frame shows up as
    millimetres here; agreement is held to 0.05 mm."""
    import ctypes as C
    import physics_ref as PR
    m = oracle.default_model()
    fn = oracle.dll.tfo_finger_gap
    fn.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    fn.restype = None
    rng = np.random.default_rng(5)
    hc = np.full(3, PR.CUBE_HALF)
    checked, by_link, worst = 0, {1: 0, 2: 0, 3: 0}, 0.0
    while checked < 400:
        f = int(rng.integers(3))
        q = rng.uniform(PR.Q_LO + 0.02, PR.Q_HI - 0.02)
        link = int(rng.choice([3, 3, 2, 1]))
        # a cube placed near a random point of the chosen link
        sh = [e for e in PR.SHAPES if e[0] == "shape" and e[1] == link][0][2]
        pt = PR.link_point_world(f, q, link, sh["a"] + rng.uniform(0, 1) * (sh["b"] - sh["a"]))
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        cp = pt + d * (PR.CUBE_HALF * rng.uniform(1.0, 1.7) + rng.uniform(0.01, 0.05))
        if cp[2] < 0.0325:
            continue
        cq = rng.normal(size=4); cq /= np.linalg.norm(cq)
        R = PR.quat_rot(cq)
        cands = PR.shape_candidates(f, q, cp, R, hc, links=(3, 2, 1), high=cp[2] > PR.UPPER_CHECK_Z)
        if any(c[0] <= -1.0 for c in cands):
            continue                                   # an axis inside the cube: outside the domain of the reference
        gaps = sorted(c[0] for c in cands)
        if len(gaps) > 1 and gaps[1] - gaps[0] < 2e-4:
            continue                                   # two shapes tie: which one holds the contact is a rounding matter
        best = min(cands, key=lambda c: c[0])
        if best[0] > 0.06:
            continue
        gap, lk = C.c_float(), C.c_int32()
        fn(C.byref(m), f, (C.c_float * 3)(*q.astype(np.float32)), (C.c_float * 3)(*cp.astype(np.float32)), (C.c_float * 4)(*cq.astype(np.float32)),
           C.byref(gap), C.byref(lk))
        assert lk.value == best[1], (f, q, cp, lk.value, best[1])
        worst = max(worst, abs(gap.value - best[0]))
        assert abs(gap.value - best[0]) < 5e-5, (f, q, cp, gap.value, best[0], best[1])
        by_link[best[1]] += 1
        checked += 1
    print(f"\n400 poses: worst gap difference {worst * 1e6:.1f} um; contacts held by link 3 / 2 / 1: {by_link[3]} / {by_link[2]} / {by_link[1]}")
    assert by_link[3] >= 100 and by_link[2] >= 50 and by_link[1] >= 10
