#!/usr/bin/env python3
"""Fit the link collision shapes of TfModel to the collision hulls the reference loads (tests/golden/model.npz, generated from the
reference's URDF / OBJ files by tests/golden/make_model_golden.py).

The hulls are TAPERED PRISMS with rounded-rectangular cross-sections (motor housings, the D-shaped distal link) and a puck (the joint
housing) at one or both ends.  Shape family of the build: a "tapered rounded box" swept along the link axis - at parameter s of the
axis a -> b the cross-section is a rectangle of half widths (w1(s), w2(s)) along two fixed directions of the link frame, with corner
rounding rho(s) and centre offset (o1(s), o2(s)), all linear in s - plus a sphere per joint housing.  With w1 = w2 = rho it is a
capsule; the distal shape ends in the fingertip sphere (w = rho = 0.0102 at s = 1), so the fingertip region is exact.

Objective per link: no point of the hull surface more than 3 mm outside the shape (the bar of the round-2 review), with as little
over-coverage as the family allows (weighted: x 4 near the fingertip, x 0.4 on the flat faces of the housings, where the neighbouring
link sits).

    python tools/fit_link_shapes.py {distal|middle|upper}      # prints the parameters (link frame, metres) and both coverages
"""
import os
import sys
import time

import numpy as np
from scipy.optimize import minimize
from scipy.spatial import ConvexHull

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(REPO, "tests", "golden", "model.npz"))
UNDER_MAX = 0.0027
S_GRID = np.linspace(0.0, 1.0, 81)


def surface_samples(V, n, seed):
    h = ConvexHull(V)
    rng = np.random.default_rng(seed)
    tri = V[h.simplices]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    idx = rng.choice(len(tri), n, p=area / area.sum())
    u = rng.random((n, 2))
    f = u.sum(1) > 1
    u[f] = 1 - u[f]
    P = tri[idx, 0] + u[:, :1] * (tri[idx, 1] - tri[idx, 0]) + u[:, 1:] * (tri[idx, 2] - tri[idx, 0])
    return np.vstack([P, V[h.vertices]]), h


def sd_hull(P, h):
    """plane-max signed distance to a convex hull: exact inside (<= 0), a lower bound of the distance outside"""
    return (P @ h.equations[:, :3].T + h.equations[:, 3]).max(1)


def shape_dist(P, a, b, e1, e2, prm):
    """signed distance of points P to the swept shape: min over s of |(planar excess over the core rectangle, axial offset)| - rho(s)"""
    w10, w11, w20, w21, r0, r1, o10, o11, o20, o21 = prm
    d = b - a
    ax = d / np.linalg.norm(d)
    out = np.full(len(P), np.inf)
    for s in S_GRID:
        o1, o2 = o10 + s * (o11 - o10), o20 + s * (o21 - o20)
        rel = P - (a + s * d) - o1 * e1 - o2 * e2
        rho = r0 + s * (r1 - r0)
        h1, h2 = max(w10 + s * (w11 - w10) - rho, 0.0), max(w20 + s * (w21 - w20) - rho, 0.0)
        q1, q2 = np.maximum(np.abs(rel @ e1) - h1, 0.0), np.maximum(np.abs(rel @ e2) - h2, 0.0)
        ua = rel @ ax
        out = np.minimum(out, np.sqrt(q1 * q1 + q2 * q2 + ua * ua) - rho)
    return out


def shape_surface(a, b, e1, e2, prm, n, rng):
    """points on the side surface of the swept shape"""
    w10, w11, w20, w21, r0, r1, o10, o11, o20, o21 = prm
    s, th = rng.random(n), rng.random(n) * 2 * np.pi
    u1, u2 = np.cos(th), np.sin(th)
    rho = r0 + s * (r1 - r0)
    h1, h2 = np.maximum(w10 + s * (w11 - w10) - rho, 0.0), np.maximum(w20 + s * (w21 - w20) - rho, 0.0)
    p1 = np.sign(u1) * h1 + rho * u1 + (o10 + s * (o11 - o10))
    p2 = np.sign(u2) * h2 + rho * u2 + (o20 + s * (o21 - o20))
    return a + s[:, None] * (b - a) + p1[:, None] * e1 + p2[:, None] * e2


def sphere_surface(c, r, n, rng):
    v = rng.normal(size=(n, 3))
    return c + r * v / np.linalg.norm(v, axis=1)[:, None]


def coverage(S, hulls, a, b, e1, e2, prm, spheres, rng, nsurf=600, weight=None):
    """(largest distance of a hull-surface point outside the shape, largest (weighted) distance of a shape-surface point outside the hull)"""
    dist = shape_dist(S, a, b, e1, e2, prm)
    for c, r in spheres:
        dist = np.minimum(dist, np.linalg.norm(S - c, axis=1) - r)
    over = 0.0
    for Q in [shape_surface(a, b, e1, e2, prm, nsurf, rng)] + [sphere_surface(c, r, 200, rng) for c, r in spheres]:
        sd = np.min([sd_hull(Q, h) for h in hulls], axis=0)
        over = max(over, (sd * weight(Q) if weight is not None else sd).max())
    return dist.max(), over


LINKS = {
    # name: (hull vertex sets, axis a, axis b, width directions e1 / e2, start vector, unpack(p) -> (prm, spheres), weight)
    "distal": dict(
        hulls=("hull_lower", "hull_tip_in_lower"), a=(0.0135, 0.0, 0.0), b=(0.0185, 0.0, -0.1592), e1=(1, 0, 0), e2=(0, 1, 0),
        x0=[0.0123, 0.0216, 0.008, -0.0014, 0.0112, 0.0, 0.0218],
        # s = 1 is the fingertip sphere (w = rho = 0.0102, no offset); free: widths, rounding and x offset at s = 0, the housing sphere
        unpack=lambda p: ((p[0], 0.0102, p[1], 0.0102, p[2], 0.0102, p[3], 0.0, 0.0, 0.0), [(np.array([p[4], 0.0, p[5]]), abs(p[6]))]),
        weight=lambda Q: np.where(Q[:, 2] < -0.10, 4.0, np.where((Q[:, 2] > -0.012) & ((Q[:, 0] < 0.0) | (Q[:, 0] > 0.022)), 0.4, 1.0))),
    "middle": dict(
        hulls=("hull_middle",), a=(0.026, -0.003, -0.012), b=(0.035, 0.0, -0.150), e1=(1, 0, 0), e2=(0, 1, 0),
        x0=[0.024, 0.0145, 0.0255, 0.023, 0.008, 0.008, 0, 0, 0, 0, 0.018, -0.0023, -0.0028, 0.029, 0.036, 0.0, -0.166, 0.024],
        unpack=lambda p: (tuple(p[:10]), [(np.array(p[10:13]), abs(p[13])), (np.array(p[14:17]), abs(p[17]))]),
        weight=lambda Q: np.where(((Q[:, 2] > -0.025) | (Q[:, 2] < -0.145)) & ((Q[:, 0] < 0.0) | (Q[:, 0] > 0.049)), 0.4, 1.0)),
    "upper": dict(
        hulls=("hull_upper",), a=(0.004, 0.040, 0.0), b=(0.004, 0.218, 0.0), e1=(1, 0, 0), e2=(0, 0, 1),
        x0=[0.016, 0.025, 0.022, 0.024, 0.008, 0.008, -0.002, 0.002, 0, 0],
        unpack=lambda p: (tuple(p[:10]), []), weight=None),
}


def fit(which):
    L = LINKS[which]
    a, b, e1, e2 = (np.array(L[k], dtype=np.float64) for k in ("a", "b", "e1", "e2"))
    hulls, parts = [], []
    for i, key in enumerate(L["hulls"]):
        S, h = surface_samples(G[key].astype(np.float64), 3000, i)
        hulls.append(h)
        parts.append(S)
    keep = []
    for i, S in enumerate(parts):                      # surface of the union of the hulls of the body
        k = np.ones(len(S), bool)
        for j, h in enumerate(hulls):
            if j != i:
                k &= sd_hull(S, h) > -1e-9
        keep.append(S[k])
    S = np.vstack(keep)

    def cost(p):
        prm, sph = L["unpack"](p)
        u, o = coverage(S, hulls, a, b, e1, e2, prm, sph, np.random.default_rng(1), weight=L["weight"])
        return max(o, 0.0) + 30.0 * max(0.0, u - UNDER_MAX)
    best = None
    for trial in range(3):
        x = np.array(L["x0"]) * (1.0 + (0.05 * np.random.default_rng(trial).normal(size=len(L["x0"])) if trial else 0.0))
        res = minimize(cost, x, method="Nelder-Mead", options=dict(maxiter=3000, xatol=2e-5, fatol=2e-6, adaptive=True))
        if best is None or res.fun < best.fun:
            best = res
    prm, sph = L["unpack"](best.x)
    return S, hulls, a, b, e1, e2, prm, sph


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "distal"
    t0 = time.time()
    S, hulls, a, b, e1, e2, prm, sph = fit(which)
    u, o = coverage(S, hulls, a, b, e1, e2, prm, sph, np.random.default_rng(2), nsurf=6000)
    print(f"{which}: hull surface at most {u * 1e3:.2f} mm outside the shape, shape surface at most {o * 1e3:.2f} mm outside the hull ({time.time() - t0:.0f} s)")
    print("  axis a", a.tolist(), "b", b.tolist(), "e1", e1.tolist(), "e2", e2.tolist())
    print("  w1 %.4f -> %.4f   w2 %.4f -> %.4f   rho %.4f -> %.4f   o1 %.4f -> %.4f   o2 %.4f -> %.4f" % prm)
    for c, r in sph:
        print("  sphere c", np.round(c, 4).tolist(), "r", round(float(r), 4))
