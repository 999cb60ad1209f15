"""The physical spec as the reference's asset files hold it (tests/golden/model.npz, written by tests/golden/make_model_golden.py
from trifingerpro.urdf, the link / boundary meshes and the object URDFs) in the shape the fp64 test models use.  Everything the
independent models of tests/test_physics_analytic.py and tests/physics_ref.py know about the robot comes from here, not from
literals; what is the BUILD's own choice (collision capsules, boundary steps, solver constants) comes from `tf_default_model`
of the library under test and is pinned against this fixture by tests/test_model_fixture.py."""
import os

import numpy as np

_G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model.npz"))
FIXTURE = {k: _G[k] for k in _G.files}


def _sym(i6):
    xx, yy, zz, xy, xz, yz = i6
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]], dtype=np.float64)


H_BASE = float(FIXTURE["base_height"])
YAW = tuple(float(a) for a in FIXTURE["finger_yaw"])
J2 = FIXTURE["j2_origin"].astype(np.float64)
J3 = FIXTURE["j3_origin"].astype(np.float64)
TIP = FIXTURE["tip_origin"].astype(np.float64)
# (mass, COM in the frame of the link the body moves with, inertia about the COM): upper, middle, lower, tip (rigidly on the lower link)
LINKS = [(float(FIXTURE["link_mass"][i]), FIXTURE["link_com"][i].astype(np.float64) + (TIP if i == 3 else 0.0), _sym(FIXTURE["link_inertia"][i]))
         for i in range(4)]
CUBE_SIZE = float(FIXTURE["cube_size"][0])
CUBE_DENSITY = float(FIXTURE["cube_density"])
PHASE3_SIZE = tuple(float(x) for x in FIXTURE["phase3_size"])
PHASE3_DENSITY = float(FIXTURE["phase3_density"])


E12 = {3: (np.array([1.0, 0, 0]), np.array([0, 1.0, 0])), 2: (np.array([1.0, 0, 0]), np.array([0, 1.0, 0])), 1: (np.array([1.0, 0, 0]), np.array([0, 0, 1.0]))}


def model_shapes(m):
    """the collision shapes of a TfModel in the order the step tests them: [("shape", link, dict) | ("sphere", link, centre, radius)];
    a shape dict holds the axis a, b, the width directions e1, e2 of the link frame and the (s = 0, s = 1) pairs w1, w2, rho, o1, o2"""
    v = lambda x: np.array(list(x), dtype=np.float64)      # noqa: E731
    out = []
    for link, sh, spheres in ((3, m.shape3, m.sph3), (2, m.shape2, m.sph2), (1, m.shape1, ())):
        out.append(("shape", link, dict(a=v(sh.a), b=v(sh.b), e1=E12[link][0], e2=E12[link][1], w1=v(sh.w1), w2=v(sh.w2), rho=v(sh.rho),
                                        o1=v(sh.o1), o2=v(sh.o2))))
        for sp in spheres:
            out.append(("sphere", link, v(sp.c), float(sp.radius)))
    return out


def shape_extent(sh, s, u_link):
    """support function of the cross-section of a link shape at parameter s of its axis along the unit direction u_link (link frame)"""
    lerp = lambda p: p[0] + s * (p[1] - p[0])      # noqa: E731
    rho = lerp(sh["rho"])
    u1, u2 = float(u_link @ sh["e1"]), float(u_link @ sh["e2"])
    return (lerp(sh["w1"]) - rho) * abs(u1) + (lerp(sh["w2"]) - rho) * abs(u2) + rho + lerp(sh["o1"]) * u1 + lerp(sh["o2"]) * u2
