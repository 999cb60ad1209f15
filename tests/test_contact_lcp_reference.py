"""The contact physics against an INDEPENDENT fp64 solution.

tests/physics_ref.py restates one solver substep in fp64 numpy without sharing any code with the C oracle or the HIP
kernels (finite-difference Jacobians of the fp64 URDF model, scipy closest points, projected Gauss-Seidel iterated to
a fixed point).  For 120 random contact configurations - fingertips and link capsules against the cube, the cube on
the floor under a finger (finger-cube-floor chains), against the boundary, floating, penetrating by up to 4 mm or
separated by up to 4 mm, with random velocities and torques - the product's substep must converge to that solution as
its sweep count grows.  What the shipped 8 sweeps leave is measured here and quoted in DESIGN.md section 2.

The cases run on the oracle in the CPU suite; the `-m gpu` variant pushes the same cases through the HIP library
(which is bit-identical to the oracle: tests/test_parity_hip_vs_oracle.py).
"""
import numpy as np
import pytest
import torch

from leibnizgym_amd import _capi as capi
import physics_ref as PR
import test_physics_analytic as T

H = 0.01
N_CASES = 120
SWEEPS = (8, 64, 1024)


def _rand_quat(rng):
    v = rng.normal(size=4)
    return v / np.linalg.norm(v)


def make_case(rng):
    """A random state with at least one link capsule within +-4 mm of the cube and nothing penetrating deeper."""
    while True:
        q = np.concatenate([rng.uniform(PR.Q_LO + 0.05, PR.Q_HI - 0.05) for _ in range(3)])
        f0 = rng.integers(3)
        tip = PR.link_point_world(f0, q[3 * f0:3 * f0 + 3], 3, PR.TIP_CAP[2])
        if tip[2] < 0.012:
            continue
        if rng.random() < 0.6:        # cube flat on the floor next to the fingertip
            yaw = rng.uniform(0, 2 * np.pi)
            cq = np.array([0, 0, np.sin(yaw / 2), np.cos(yaw / 2)])
            d = rng.normal(size=3)
            d[2] = abs(d[2]) * 0.3
            d /= np.linalg.norm(d)
            c = tip - d * (PR.CUBE_HALF * rng.uniform(1.0, 1.35) + 0.0102)
            c[2] = 0.0325 + rng.uniform(-0.0005, 0.001)
        else:                         # cube in the air, any orientation
            cq = _rand_quat(rng)
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            c = tip - d * (PR.CUBE_HALF * rng.uniform(1.0, 1.6) + 0.0102)
            if c[2] < 0.06:
                continue
        if np.hypot(c[0], c[1]) > 0.15:
            continue
        R = PR.quat_rot(cq)
        ok, near = True, False
        for f in range(3):
            qf = q[3 * f:3 * f + 3]
            for g, _ in PR.finger_gaps(f, qf, c, R, np.full(3, PR.CUBE_HALF)):
                ok &= g >= -0.004
                near |= g < 0.004
            tipf = PR.link_point_world(f, qf, 3, PR.TIP_CAP[2])
            ok &= tipf[2] - 0.0102 >= -0.003
            ok &= PR.wall_radius_at(tipf[2]) - np.hypot(tipf[0], tipf[1]) - 0.0102 >= -0.003
        if not ok or not near:
            continue
        qd = rng.uniform(-2, 2, 9)
        tau = rng.uniform(-0.36, 0.36, 9)
        cube = np.concatenate([c, cq, rng.uniform(-0.3, 0.3, 3), rng.uniform(-2, 2, 3)])
        return q, qd, cube, tau


def product_substep(lib, device, q, qd, cube, tau, sweeps):
    eng = T.engine(lib, device=device, dt=H, substeps=1, solver_iterations=sweeps)
    f32 = dict(dtype=torch.float32, device=device)
    eng.q[:, 0] = torch.tensor(q, **f32)
    eng.qd[:, 0] = torch.tensor(qd, **f32)
    eng.cube[:, 0] = torch.tensor(cube, **f32)
    eng.tau[:, 0] = torch.tensor(tau, **f32)
    eng.simulate()
    st = eng.state[:, 0].cpu().numpy().astype(np.float64)
    eng.close()
    return st[9:18], st[25:28], st[28:31]


def scaled_error(got, want):
    """max over dofs of |error| in units of the velocity limits' order: joints / 10 rad/s, cube m/s, cube rad/s / 20"""
    return max(np.abs(got[0] - want[0]).max() / 10.0, np.abs(got[1] - want[1]).max(), np.abs(got[2] - want[2]).max() / 20.0)


def _run(lib, device, n_cases, sweeps_list):
    rng = np.random.default_rng(20261002)
    errs = []
    kinds = {"fc": 0, "chain": 0, "te": 0, "ff": 0, "wall": 0, "link2": 0}
    skipped = 0
    while len(errs) < n_cases:
        q, qd, cube, tau = make_case(rng)
        ref = PR.ref_substep(q, qd, cube, tau, H, max_sweeps=50000)
        det = ref[3]
        if det["sweeps"] >= 50000:               # a (rare) configuration whose fp64 Gauss-Seidel has not settled to 1e-13: no reference value
            skipped += 1
            assert skipped <= max(2, n_cases // 50), "the reference fails to reach its fixed point too often"
            continue
        live = [x for x in det["fc"] if x[3].lam > 0]
        kinds["fc"] += bool(live)
        kinds["chain"] += bool(live) and det["n_floor"] > 0
        kinds["te"] += any(x[3].lam > 0 for x in det["te"])
        kinds["ff"] += any(x[3] > 0 for x in det["ff"])
        kinds["wall"] += det["n_wall"] > 0
        kinds["link2"] += any(x[1] != 3 for x in live)
        errs.append([scaled_error(product_substep(lib, device, q, qd, cube, tau, k), ref[:3]) for k in sweeps_list])
    return np.array(errs), kinds


def _check(errs, kinds, sweeps_list, n_cases):
    med = np.median(errs, axis=0)
    p90 = np.percentile(errs, 90, axis=0)
    print("\nsweeps  median      p90         max   (scaled velocity error vs the fp64 fixed point)")
    for k, s in enumerate(sweeps_list):
        print(f"{s:6d}  {med[k]:.3e}  {p90[k]:.3e}  {errs[:, k].max():.3e}")
    print("case mix:", kinds)
    # the random cases do exercise what they are meant to
    if n_cases >= 120:
        assert kinds["fc"] >= 36 and kinds["chain"] >= 14 and kinds["te"] >= 3 and kinds["ff"] >= 1
    last = len(sweeps_list) - 1
    # converged: every case agrees with the independent solution (a handful of cases with redundant corner contacts
    # have a non-unique friction split, hence 1e-3 and not 1e-5)
    assert errs[:, last].max() < 1e-3, errs[:, last].max()
    assert np.percentile(errs[:, last], 90) < 2e-5
    # and the error shrinks with the sweep count
    assert p90[0] >= p90[1] >= p90[last] and errs[:, 1].max() < 0.3 * max(errs[:, 0].max(), 1e-3)
    # what the shipped 8 sweeps (cold start) leave: documented in DESIGN.md section 2
    assert med[0] < 2e-4 and p90[0] < 5e-2 and errs[:, 0].max() < 0.3


def _run_warm(lib, device, n_cases, k_warm=4, sweeps=8):
    """Residual of the SHIPPED sweep count in the warm state: the product runs `k_warm` substeps from a random contact state (its
    warm-start rows fill), then substep k_warm + 1 is compared with the fp64 fixed point computed from the state it started
    from - once with the warm rows the product carried there, once cold (rows cleared) from the same state."""
    rng = np.random.default_rng(20261003)
    warm, cold, kept = [], [], 0
    f32 = dict(dtype=torch.float32, device=device)
    while kept < n_cases:
        q, qd, cube, tau = make_case(rng)
        eng = T.engine(lib, device=device, dt=H, substeps=1, solver_iterations=sweeps)
        eng.q[:, 0] = torch.tensor(q, **f32); eng.qd[:, 0] = torch.tensor(qd, **f32)
        eng.cube[:, 0] = torch.tensor(cube, **f32); eng.tau[:, 0] = torch.tensor(tau, **f32)
        for _ in range(k_warm):
            eng.simulate()
        st = eng.state[:, 0].cpu().numpy().astype(np.float64)
        q1, qd1, cube1 = st[0:9], st[9:18], st[18:31]
        try:
            ref = PR.ref_substep(q1, qd1, cube1, tau, H, max_sweeps=50000)
        except ValueError:                      # a capsule axis ended up inside the cube: outside the domain of the reference
            eng.close()
            continue
        det = ref[3]
        persistent = any(x[3].lam > 0 for x in det["fc"]) or any(x[3].lam > 0 for x in det["te"])
        if det["sweeps"] >= 50000 or not persistent:
            eng.close()
            continue
        saved = eng.state[:, 0].clone()
        eng.simulate()
        s2 = eng.state[:, 0].cpu().numpy().astype(np.float64)
        warm.append(scaled_error((s2[9:18], s2[25:28], s2[28:31]), ref[:3]))
        eng.state[:, 0] = saved
        eng.state[capi.S_LAM_FC:, 0] = 0.0      # every warm-start row
        eng.simulate()
        s3 = eng.state[:, 0].cpu().numpy().astype(np.float64)
        cold.append(scaled_error((s3[9:18], s3[25:28], s3[28:31]), ref[:3]))
        eng.close()
        kept += 1
    return np.array(warm), np.array(cold)


def _report_warm(warm, cold):
    print("\nshipped sweeps, substep 5 of a persistent contact: scaled velocity error vs the fp64 fixed point")
    for name, e in (("warm start", warm), ("cold start", cold)):
        print(f"{name}:  median {np.median(e):.3e}  p90 {np.percentile(e, 90):.3e}  p99 {np.percentile(e, 99):.3e}  max {e.max():.3e}")


def test_warm_state_residual_of_the_shipped_sweeps(oracle):
    """what 8 sweeps leave once the warm start has had four substeps to fill."""
    warm, cold = _run_warm(oracle, "cpu", 10)        # (the fp64 reference needs ~5 s per case; the 120-case table is in profiles/r3_b_pgs_variants.txt)
    _report_warm(warm, cold)
    assert np.median(warm) <= np.median(cold) * 2.0 + 1e-5
    assert np.median(warm) < 5e-3 and warm.max() < 0.5, (np.median(warm), warm.max())


def test_substep_converges_to_the_independent_lcp_solution(oracle):
    errs, kinds = _run(oracle, "cpu", N_CASES, SWEEPS)
    _check(errs, kinds, SWEEPS, N_CASES)


@pytest.mark.gpu
def test_substep_converges_to_the_independent_lcp_solution_gpu(hip):
    errs, kinds = _run(hip, "cuda:0", 60, SWEEPS)
    _check(errs, kinds, SWEEPS, 60)


# ---- the boundary above its vertical ring: cube corners and fingertips against the flared part ---------------------------------------
def _boundary_cases(rng, n):
    """cube thrown at the flared part of the boundary (any orientation, some corner within -2 .. +3 mm of the surface, above the
    vertical ring), and fingertips stretched out to it; everything else far from any contact"""
    out = []
    hc = PR.CUBE_HALF
    corners = np.array([[sx, sy, sz] for sx in (-hc, hc) for sy in (-hc, hc) for sz in (-hc, hc)])
    q_rest = np.array([0.0, 0.9, -1.7] * 3)
    while len(out) < n:
        if len(out) % 4 != 3:
            cq = _rand_quat(rng)
            R = PR.quat_rot(cq)
            phi, zc, g0 = rng.uniform(0, 2 * np.pi), rng.uniform(0.06, 0.13), rng.uniform(-0.002, 0.003)
            ed = np.array([np.cos(phi), np.sin(phi), 0.0])

            def min_gap(rc):
                P = rc * ed + np.array([0.0, 0.0, zc]) + corners @ R.T
                return min(PR.wall_radius_at(p[2]) - np.hypot(p[0], p[1]) for p in P)
            lo, hi = 0.05, 0.30
            for _ in range(50):
                mid = 0.5 * (lo + hi)
                lo, hi = (mid, hi) if min_gap(mid) > g0 else (lo, mid)
            c = lo * ed + np.array([0.0, 0.0, zc])
            if min((c + corners @ R.T)[:, 2]) < 0.034:          # keep the contact on the cone, not on the ring or the floor
                continue
            try:                                                # and the resting fingers out of it
                if min(g for f in range(3) for g, _ in PR.finger_gaps(f, q_rest[3 * f:3 * f + 3], c, R, np.full(3, hc))) < 0.01:
                    continue
            except ValueError:
                continue
            v = ed * rng.uniform(0.2, 1.0) + rng.normal(size=3) * 0.1
            cube = np.concatenate([c, cq, v, rng.uniform(-3, 3, 3)])
            out.append((q_rest.copy(), np.zeros(9), cube, np.zeros(9)))
        else:
            q = q_rest.copy()
            q[0:3] = rng.uniform(PR.Q_LO + 0.05, PR.Q_HI - 0.05)
            tip = PR.link_point_world(0, q[0:3], 3, PR.TIP_CAP[2])
            g = PR.wall_radius_at(tip[2]) - np.hypot(tip[0], tip[1]) - 0.0102
            if not (-0.002 < g < 0.003 and tip[2] > 0.04):
                continue
            qd = np.zeros(9)
            qd[0:3] = rng.uniform(-2, 2, 3)
            tau = np.zeros(9)
            tau[0:3] = rng.uniform(-0.36, 0.36, 3)
            cube = np.array([0.0, 0.0, 0.0325, 0, 0, 0, 1.0, 0, 0, 0, 0, 0, 0])
            out.append((q, qd, cube, tau))
    return out


def _check_boundary(lib, device, n):
    errs, hits = [], 0
    for q, qd, cube, tau in _boundary_cases(np.random.default_rng(77), n):
        ref = PR.ref_substep(q, qd, cube, tau, H, max_sweeps=50000)
        det = ref[3]
        assert det["sweeps"] < 50000
        hits += det["n_wall"] > 0 or any(x[1] == "wall" and x[3].lam > 0 for x in det["te"])
        errs.append([scaled_error(product_substep(lib, device, q, qd, cube, tau, k), ref[:3]) for k in (8, 1024)])
    errs = np.array(errs)
    print(f"\nboundary cases with a pushing contact on the cone: {hits} of {n};  8 sweeps median {np.median(errs[:, 0]):.2e} max {errs[:, 0].max():.2e};"
          f"  1024 sweeps median {np.median(errs[:, 1]):.2e} max {errs[:, 1].max():.2e}")
    assert hits >= n // 2
    # converged, the rows agree with the reference; 8 cold sweeps leave up to O(1) on an edge of the flying cube that hits the cone with two
    # corners at once (friction rows of the two corners against each other - the slow case of Gauss-Seidel)
    assert errs[:, 1].max() < 1e-3 and np.median(errs[:, 1]) < 2e-5 and np.median(errs[:, 0]) < 5e-3, errs


def test_boundary_rows_on_the_flared_part_agree_with_the_independent_solution(oracle):
    """Cube corners (any orientation, flying) and fingertips against the flared part of the boundary - radius from the piecewise-linear
    profile at the height of the contact, horizontal inward normal, rows n, t, +z - against the fp64 reference, which builds them from its
    own geometry (the random cases of the main test hold only a handful of boundary contacts)."""
    _check_boundary(oracle, "cpu", 16)


@pytest.mark.gpu
def test_boundary_rows_on_the_flared_part_agree_with_the_independent_solution_gpu(hip):
    _check_boundary(hip, "cuda:0", 16)


def test_truncation_residual_of_the_shipped_sweeps_on_many_states(oracle):
    """What the shipped 8 sweeps leave, measured on MANY states without the fp64 reference in the loop: the product's substep against the same
    substep iterated 2048 times from the same state (same warm-start rows).  A regression guard at the measured level (round 4,
    profiles/r4_a_solver_variants.txt: 939 states, warm p90 3.3e-2, p99 0.19; 16 sweeps: 9.6e-3 / 0.11) - the review's bar (warm p90 <= 5e-3,
    p99 <= 5e-2) is NOT met by this iteration at any affordable sweep count."""
    rng = np.random.default_rng(20261004)
    n = 256
    cases = [make_case(rng) for _ in range(n)]
    f32 = dict(dtype=torch.float32)
    res = {}
    for sweeps in (8, 16, (8, 2)):                       # (8, 2): solver_iterations 8 with solver_inner 2 - the cube block twice per sweep (API 6)
        it, inner = sweeps if isinstance(sweeps, tuple) else (sweeps, 1)
        eng = T.engine(oracle, n=n, dt=H, substeps=1, solver_iterations=it, solver_inner=inner)
        ref = T.engine(oracle, n=n, dt=H, substeps=1, solver_iterations=2048)
        for k, name in enumerate(("q", "qd", "cube", "tau")):
            getattr(eng, name).copy_(torch.tensor(np.array([c[k] for c in cases]).T, **f32))
        for _ in range(4):
            eng.simulate()
        saved = eng.state.clone()
        eng.simulate()
        ref.state.copy_(saved)
        ref.simulate()
        got, want = eng.state, ref.state
        d = (got - want).abs()
        err = torch.maximum(torch.maximum(d[9:18].max(0).values / 10.0, d[25:28].max(0).values), d[28:31].max(0).values / 20.0).numpy()
        lam = want[capi.S_LAM_FC:]
        live = ((((lam[12:15].to(torch.int32) & 3) != 0) & (lam[0:12:4] > 0)).any(0) | (lam[15:24:3] > 0).any(0)).numpy()
        res[sweeps] = err[live]
        eng.close(); ref.close()
        print(f"\n{sweeps} sweeps, {int(live.sum())} states with a live finger contact: median {np.median(err[live]):.2e}  p90 {np.percentile(err[live], 90):.2e}  "
              f"p99 {np.percentile(err[live], 99):.2e}  max {err[live].max():.2e}")
    assert len(res[8]) >= 80
    assert np.median(res[8]) < 5e-3 and np.percentile(res[8], 90) < 8e-2 and res[8].max() < 1.0
    assert np.median(res[16]) < np.median(res[8]) and np.percentile(res[16], 90) < np.percentile(res[8], 90)     # more sweeps, less residual
    # solver_inner = 2 (what the block study of round 4 pointed at: the unresolved part is the block of all rows that touch the cube): well below the 8 plain
    # sweeps - and no better than 16 plain sweeps, which cost the same in this kernel (every pass is an exchange between the cube role and the finger roles)
    assert np.percentile(res[(8, 2)], 90) < 0.6 * np.percentile(res[8], 90) and np.median(res[(8, 2)]) < np.median(res[8])
