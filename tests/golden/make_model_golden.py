#!/usr/bin/env python3
"""Generate tests/golden/model.npz: the PHYSICAL SPEC the reference hands to IsaacGym, read from its asset files.

Build container only (reads /root/reference; the fixture travels, the assets do not).  Numbers only - no text of any
reference file is kept.  Sources (paths relative to the reference checkout, resources/assets/trifinger/):

  robot_properties_fingers/urdf/pro/trifingerpro.urdf
      :51-55,161-190,461-475   kinematic tree: base height, finger yaws, joint origins and axes
      :94-98,114-118,134-138,155-159   link inertials (mass, COM, inertia)
      :88-93,108-113,128-133,149-153   collision origins of the link meshes
  robot_properties_fingers/meshes/stl/pro/SIM__BL-Finger_{Proximal,Intermediate,Tip_without_tip,Tip_actual_tip}.obj
      one convex hull per link is what the reference loads (leibnizgym/envs/trifinger/trifinger_env.py:859-879:
      no V-HACD): the fixture holds the hull VERTICES in the frame of the link they move with
  robot_properties_fingers/urdf/high_table_boundary.urdf:20-259 + meshes/convex_table_boundary/convex_*.obj
      the 40 convex pieces of the boundary annulus: inner radius per height band
  objects/urdf/cube_multicolor_rrc.urdf:10-18, cube_multicolor_rrc_phase3.urdf    object box size and density

    python tests/golden/make_model_golden.py            # writes tests/golden/model.npz
"""
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np
from scipy.spatial import ConvexHull

REF = os.environ.get("TF_REFERENCE", "/root/reference")
ASSETS = os.path.join(REF, "resources", "assets", "trifinger")
RPF = os.path.join(ASSETS, "robot_properties_fingers")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model.npz")


def floats(s):
    return np.array([float(x) for x in s.split()], dtype=np.float64)


def rpy_matrix(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return rz @ ry @ rx


def origin_of(el):
    o = el.find("origin") if el is not None else None
    xyz = floats(o.get("xyz", "0 0 0")) if o is not None else np.zeros(3)
    rpy = floats(o.get("rpy", "0 0 0")) if o is not None else np.zeros(3)
    return xyz, rpy


def obj_vertices(path):
    return np.array([[float(x) for x in line.split()[1:4]] for line in open(path) if line.startswith("v ")], dtype=np.float64)


def mesh_path(filename):
    return filename.replace("package://robot_properties_fingers", RPF)


def main():
    out = {}
    root = ET.parse(os.path.join(RPF, "urdf", "pro", "trifingerpro.urdf")).getroot()
    links = {l.get("name"): l for l in root.findall("link")}
    joints = {j.get("name"): j for j in root.findall("joint")}

    # ---- kinematic tree ----
    out["base_height"] = origin_of(joints["base_to_upper_holder_joint"])[0][2]
    yaws = []
    for suffix in ("0", "120", "240"):
        j = [jj for jj in joints.values() if jj.find("child").get("link") == f"finger_base_link_{suffix}"][0]
        xyz, rpy = origin_of(j)
        assert np.allclose(xyz, 0) and np.allclose(rpy[:2], 0)
        yaws.append(rpy[2])
    out["finger_yaw"] = np.array(yaws)
    for key, jn in (("j1", "finger_base_to_upper_joint_0"), ("j2", "finger_upper_to_middle_joint_0"), ("j3", "finger_middle_to_lower_joint_0"),
                    ("tip", "finger_lower_to_tip_joint_0")):
        xyz, rpy = origin_of(joints[jn])
        assert np.allclose(rpy, 0)
        out[f"{key}_origin"] = xyz
        ax = joints[jn].find("axis")
        if ax is not None:
            out[f"{key}_axis"] = floats(ax.get("xyz"))
    # the three fingers are copies of one another (only the yaw differs)
    for suffix in ("120", "240"):
        for jn in ("finger_base_to_upper_joint", "finger_upper_to_middle_joint", "finger_middle_to_lower_joint", "finger_lower_to_tip_joint"):
            a, b = origin_of(joints[f"{jn}_0"]), origin_of(joints[f"{jn}_{suffix}"])
            assert np.allclose(a[0], b[0]) and np.allclose(a[1], b[1])

    # ---- inertials ----
    names = ("upper", "middle", "lower", "tip")
    mass, com, inertia = [], [], []
    for n in names:
        ine = links[f"finger_{n}_link_0"].find("inertial")
        xyz, rpy = origin_of(ine)
        assert np.allclose(rpy, 0)
        mass.append(float(ine.find("mass").get("value")))
        com.append(xyz)
        i = ine.find("inertia")
        inertia.append([float(i.get(k)) for k in ("ixx", "iyy", "izz", "ixy", "ixz", "iyz")])
        for suffix in ("120", "240"):      # identical on every finger
            i2 = links[f"finger_{n}_link_{suffix}"].find("inertial")
            assert float(i2.find("mass").get("value")) == mass[-1] and np.allclose(origin_of(i2)[0], xyz)
    out["link_mass"], out["link_com"], out["link_inertia"] = np.array(mass), np.array(com), np.array(inertia)
    # distal body of the model: lower link with the rigidly attached tip link merged (parallel axis theorem, fp64)
    m2 = np.array([mass[2], mass[3]])
    c2 = np.array([com[2], com[3] + out["tip_origin"]])
    mm = m2.sum()
    cc = (m2[:, None] * c2).sum(0) / mm
    I = np.zeros((3, 3))
    for k in range(2):
        ik = inertia[2 + k]
        I += np.array([[ik[0], ik[3], ik[4]], [ik[3], ik[1], ik[5]], [ik[4], ik[5], ik[2]]])
        d = c2[k] - cc
        I += m2[k] * ((d @ d) * np.eye(3) - np.outer(d, d))
    out["distal_mass"], out["distal_com"] = mm, cc
    out["distal_inertia"] = np.array([I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]])

    # ---- collision hulls, in the frame of the link they move with (the tip hull also in the lower-link frame) ----
    for n in names:
        col = links[f"finger_{n}_link_0"].find("collision")
        xyz, rpy = origin_of(col)
        V = obj_vertices(mesh_path(col.find("geometry/mesh").get("filename"))) @ rpy_matrix(rpy).T + xyz
        H = V[ConvexHull(V).vertices]
        out[f"hull_{n}"] = H.astype(np.float32)
        out[f"collision_origin_{n}"] = np.concatenate([xyz, rpy])
    out["hull_tip_in_lower"] = (out["hull_tip"].astype(np.float64) + out["tip_origin"]).astype(np.float32)
    # fingertip: the actual-tip mesh is a sphere (least squares on its hull vertices, tip-link frame)
    T = out["hull_tip"].astype(np.float64)
    A = np.hstack([2 * T, np.ones((len(T), 1))])
    sol = np.linalg.lstsq(A, (T ** 2).sum(1), rcond=None)[0]
    centre = sol[:3]
    radius = np.sqrt(sol[3] + centre @ centre)
    out["tip_sphere_centre"], out["tip_sphere_radius"] = centre, radius
    out["tip_sphere_residual"] = np.abs(np.linalg.norm(T - centre, axis=1) - radius).max()

    # ---- boundary: 40 convex pieces; inner radius (closest approach of a piece to the z axis) per height band ----
    stage = ET.parse(os.path.join(RPF, "urdf", "high_table_boundary.urdf")).getroot()
    cols = stage.find("link").findall("collision")
    pieces = []
    for c in cols:
        xyz, rpy = origin_of(c)
        assert np.allclose(xyz, 0) and np.allclose(rpy, 0)
        pieces.append(obj_vertices(mesh_path(c.find("geometry/mesh").get("filename"))))
    out["boundary_num_pieces"] = len(pieces)
    allv = np.vstack(pieces)
    out["boundary_z_range"] = np.array([allv[:, 2].min(), allv[:, 2].max()])
    out["boundary_outer_radius"] = np.hypot(allv[:, 0], allv[:, 1]).max()
    # the inner profile r(z): for a grid of heights, the smallest radius at which any piece has material.  A convex piece's
    # cross-section at height z is the hull of its edge intersections; its distance to the axis is evaluated on that polygon.
    zs = np.linspace(out["boundary_z_range"][0] + 1e-4, out["boundary_z_range"][1] - 1e-4, 177)
    prof = np.full(len(zs), np.inf)
    for V in pieces:
        h = ConvexHull(V)
        edges = set()
        for s in h.simplices:
            for i in range(3):
                edges.add((min(s[i], s[(i + 1) % 3]), max(s[i], s[(i + 1) % 3])))
        E = np.array(sorted(edges))
        a, b = V[E[:, 0]], V[E[:, 1]]
        for k, z in enumerate(zs):
            cross = (a[:, 2] - z) * (b[:, 2] - z) < 0
            if cross.sum() < 3:
                continue
            t = (z - a[cross, 2]) / (b[cross, 2] - a[cross, 2])
            P = (a[cross] + t[:, None] * (b[cross] - a[cross]))[:, :2]
            hp = P[ConvexHull(P).vertices]
            # distance from the origin to the polygon's boundary (the axis is outside every piece)
            q, r_ = hp, np.roll(hp, -1, axis=0)
            d = r_ - q
            tt = np.clip(-(q * d).sum(1) / (d * d).sum(1), 0, 1)
            prof[k] = min(prof[k], np.linalg.norm(q + tt[:, None] * d, axis=1).min())
    out["boundary_profile_z"], out["boundary_profile_r"] = zs, prof

    # ---- objects ----
    for key, fn in (("cube", "cube_multicolor_rrc.urdf"), ("phase3", "cube_multicolor_rrc_phase3.urdf")):
        link = ET.parse(os.path.join(ASSETS, "objects", "urdf", fn)).getroot().find("link")
        out[f"{key}_size"] = floats(link.find("collision/geometry/box").get("size"))
        out[f"{key}_density"] = float(link.find("inertial/density").get("value"))
    np.savez_compressed(OUT, **{k: np.asarray(v) for k, v in out.items()})
    print("wrote", OUT, {k: np.asarray(v).shape for k, v in out.items()})
    print("tip sphere", out["tip_sphere_centre"], out["tip_sphere_radius"], "residual", out["tip_sphere_residual"])
    for z0, z1 in ((0.0, 0.06), (0.06, 0.10), (0.10, 0.14), (0.14, 0.176)):
        sel = (zs > z0 + 0.002) & (zs < z1 - 0.002)
        print(f"boundary inner radius for z in [{z0}, {z1}]: {prof[sel].min():.4f} .. {prof[sel].max():.4f}")
    print("distal merged:", mm, cc, out["distal_inertia"])


if __name__ == "__main__":
    sys.exit(main())
