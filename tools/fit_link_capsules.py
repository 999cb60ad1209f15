#!/usr/bin/env python3
"""Fit the link collision primitives of TfModel (capsules, csrc/trifinger_hip.hip:tf_default_model) to the collision hulls the
reference loads (tests/golden/model.npz, generated from the reference's URDF / OBJ files by tests/golden/make_model_golden.py).

Objective per link: the union of K capsules must COVER the convex hull - no point of the hull surface more than 3 mm outside the
union (the bar of the round-2 review) - with as little over-coverage (capsule surface outside the hull) as the family allows; the
fingertip capsule (tube + tip sphere of the distal link) is kept exactly as it is: it carries nearly every contact of the task.
The hulls are tapered prisms with pucks (the joint housings) at their ends, so each link gets capsules ALONG the link for the body
and, where a housing is wider than the body, one capsule ACROSS the link along the joint axis.

    python tools/fit_link_capsules.py [distal|middle|upper] [seed]     # prints the capsules (link frame, metres) and both coverages
"""
import os
import sys
import time

import numpy as np
from scipy.optimize import minimize
from scipy.spatial import ConvexHull

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(REPO, "tests", "golden", "model.npz"))
UNDER_MAX = 0.0028


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


def seg_dist(P, a, b):
    d = b - a
    s = np.clip((P - a) @ d / max(d @ d, 1e-12), 0, 1)
    return np.linalg.norm(P - (a + s[:, None] * d), axis=1)


_DIRS = None


def capsule_surface(a, b, r, n):
    global _DIRS
    if _DIRS is None or len(_DIRS) != n:
        rng = np.random.default_rng(1)
        v = rng.normal(size=(n, 3))
        _DIRS = (v / np.linalg.norm(v, axis=1)[:, None], rng.random(n))
    v, s = _DIRS
    d = b - a
    L = np.linalg.norm(d)
    e = d / max(L, 1e-9)
    perp = v - (v @ e)[:, None] * e
    perp /= np.maximum(np.linalg.norm(perp, axis=1), 1e-9)[:, None]
    side = a + s[:, None] * d + r * perp
    dn = (v @ e)[:, None]
    return np.vstack([side, a + r * np.where(dn < 0, v, -v), b + r * np.where(dn > 0, v, -v)])


class Body:
    def __init__(self, hull_vertex_sets):
        self.hulls = []
        parts = []
        for i, V in enumerate(hull_vertex_sets):
            V = np.asarray(V, dtype=np.float64)
            S, h = surface_samples(V, 2500, i)
            self.hulls.append(h)
            parts.append(S)
        keep = []
        for i, S in enumerate(parts):                  # hull surface inside another hull of the body is not surface of the body
            k = np.ones(len(S), bool)
            for j, h in enumerate(self.hulls):
                if j != i:
                    k &= sd_hull(S, h) > -1e-9
            keep.append(S[k])
        self.S = np.vstack(keep)

    def coverage(self, caps, nsurf=300, weight=None):
        """(largest distance of a hull-surface point outside the union, largest distance of a capsule-surface point outside the body;
        `weight(points)` scales the latter per point - the fit uses it to keep the fingertip region tight)"""
        gap = np.min([seg_dist(self.S, np.asarray(a, float), np.asarray(b, float)) - r for a, b, r in caps], axis=0)
        over = 0.0
        for a, b, r in caps:
            Q = capsule_surface(np.asarray(a, float), np.asarray(b, float), r, nsurf)
            sd = np.min([sd_hull(Q, h) for h in self.hulls], axis=0)
            if weight is not None:
                sd = sd * weight(Q)
            over = max(over, sd.max())
        return gap.max(), over


def fit(body, build, x0, seed, iters=2500, weight=None):
    def cost(p):
        u, o = body.coverage(build(p), weight=weight)
        return max(o, 0.0) + 30.0 * max(0.0, u - UNDER_MAX)
    best = None
    rng = np.random.default_rng(seed)
    for trial in range(3):
        x = np.array(x0) * (1.0 + (0.04 * rng.normal(size=len(x0)) if trial else 0.0))
        res = minimize(cost, x, method="Nelder-Mead", options=dict(maxiter=iters, xatol=2e-5, fatol=2e-6, adaptive=True))
        if best is None or res.fun < best.fun:
            best = res
    return best.x


TIP_C = (0.0185, 0.0, -0.1592)        # fingertip sphere centre in the lower-link frame: tip_origin + fitted centre (0, 0, 0.0034)
TIP_CAP = ((0.0135, 0.0, 0.0), TIP_C, 0.0102)


def distal(seed):
    body = Body([G["hull_lower"], G["hull_tip_in_lower"]])

    def build(p):   # fingertip capsule (fixed) + a symmetric pair fanning out from the tube to the joint housing + two capsules ACROSS the
        x1, y1, z1, x2, y2, z2, r, hx, hy, hz1, hz2, hr = p     # housing (a 22 mm thick puck of radius 22 mm about the joint axis)
        return [TIP_CAP, ((x1, y1, z1), (x2, y2, z2), abs(r)), ((x1, -y1, z1), (x2, -y2, z2), abs(r)),
                ((hx, -hy, hz1), (hx, hy, hz1), abs(hr)), ((hx, -hy, hz2), (hx, hy, hz2), abs(hr))]
    w = lambda Q: np.where(Q[:, 2] < -0.10, 4.0, 1.0)      # noqa: E731  the fingertip region must not be inflated
    x = fit(body, build, [0.012, 0.0115, -0.002, 0.017, 0.003, -0.095, 0.0105, 0.011, 0.010, 0.010, -0.010, 0.011], seed, weight=w)
    return body, build(x)


def middle(seed):
    body = Body([G["hull_middle"]])

    def build(p):   # 2 x 2 capsules along the body (the hull is not symmetric in y: own centre), one sphere-like capsule across each joint housing
        xa1, xa2, xb1, xb2, ya, yb, yc, z1, z2, ra, rb, tx1, tx2, tz, tr, bx1, bx2, bz, br = p
        return [((xa1, yc + ya, z1), (xa2, yc + ya, z2), abs(ra)), ((xa1, yc - ya, z1), (xa2, yc - ya, z2), abs(ra)),
                ((xb1, yc + yb, z1), (xb2, yc + yb, z2), abs(rb)), ((xb1, yc - yb, z1), (xb2, yc - yb, z2), abs(rb)),
                ((tx1, yc, tz), (tx2, yc, tz), abs(tr)), ((bx1, 0.0, bz), (bx2, 0.0, bz), abs(br))]
    # over-coverage at the flat faces of the two joint housings (where the neighbouring links sit) is harmless: weight 0.4 there
    w = lambda Q: np.where((Q[:, 2] > -0.022) | (Q[:, 2] < -0.150), 0.4, 1.0)      # noqa: E731
    x = fit(body, build, [0.012, 0.031, 0.036, 0.037, 0.011, 0.011, -0.003, -0.020, -0.150, 0.0135, 0.0135,
                          0.017, 0.019, -0.003, 0.029, 0.037, 0.033, -0.166, 0.024], seed, iters=3500, weight=w)
    return body, build(x)


def upper(seed):
    body = Body([G["hull_upper"]])

    def build(p):   # two symmetric pairs along the link (it is never reached by the cube on the table; its capsules cost nothing)
        xa1, xa2, xb1, xb2, za, zb, y1, y2, ra, rb = p
        return [((xa1, y1, za), (xa2, y2, za), abs(ra)), ((xa1, y1, -za), (xa2, y2, -za), abs(ra)),
                ((xb1, y1, zb), (xb2, y2, zb), abs(rb)), ((xb1, y1, -zb), (xb2, y2, -zb), abs(rb))]
    x = fit(body, build, [-0.004, -0.008, 0.010, 0.016, 0.010, 0.009, 0.042, 0.218, 0.014, 0.015], seed)
    return body, build(x)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "distal"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    body, caps = {"distal": distal, "middle": middle, "upper": upper}[which](seed)
    u, o = body.coverage(caps, 4000)
    print(f"{which}: hull surface at most {u * 1e3:.2f} mm outside the capsules, capsule surface at most {o * 1e3:.2f} mm outside the hull "
          f"({time.time() - t0:.0f} s)")
    for a, b, r in caps:
        print("   a", np.round(a, 4).tolist(), "b", np.round(b, 4).tolist(), "r", round(float(r), 4))
