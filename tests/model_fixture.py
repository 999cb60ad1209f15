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


def model_capsules(m):
    """[(link, a, b, radius)] of a TfModel: the fingertip capsule first, then its table - the order the step tests them in"""
    caps = [(3, np.array(list(m.cap_a), dtype=np.float64), np.array(list(m.cap_b), dtype=np.float64), float(m.cap_radius))]
    for i in range(m.n_caps):
        c = m.caps[i]
        caps.append((int(c.link), np.array(list(c.a), dtype=np.float64), np.array(list(c.b), dtype=np.float64), float(c.radius)))
    return caps
