"""The object as a general box (the reference's objects/urdf/cube_multicolor_rrc_phase3.urdf, 20 x 80 x 20 mm,
density 500; `CuboidalObject` of envs/trifinger/utils.py:57-131 takes any (x, y, z) size).  `TfModel.box` switches the
native step to per-axis half extents, the principal moments of inertia and - optionally - the gyroscopic term; its solve
runs in inertia-scaled angular coordinates.  Checked here: the mass / CuboidalObject numbers, rest on
every kind of face, torque-free motion against the analytic symmetric top, the contact solve against the independent fp64
reference (tests/physics_ref.py, which uses the full world-frame inertia tensor instead of the scaled coordinates), task
constants of the resets, and - under `-m gpu` - the HIP box kernels against the oracle bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

import parity_util as pu
import physics_ref as PR
import test_contact_lcp_reference as L
import test_physics_analytic as T
from leibnizgym_amd import _capi as capi
from leibnizgym_amd.engine import TrifingerEngine, make_config
from leibnizgym_amd.envs import TrifingerEnv
from oracle_util import golden

PHASE3 = (0.02, 0.08, 0.02)
DENSITY = 500.0


def box_edit(lib, size=PHASE3, density=DENSITY, **fields):
    def edit(m):
        lib.tf_model_set_box(C.byref(m), (C.c_float * 3)(*size), float(density))
        for k, v in fields.items():
            setattr(m, k, v)
    return edit


def test_box_model_numbers(oracle):
    m = oracle.box_model(PHASE3, DENSITY)
    spec = PR.box_spec(PHASE3, DENSITY)
    assert m.box == 1 and m.box_gyroscopic == 1
    assert np.allclose(list(m.box_half), spec["half"], rtol=1e-6) and np.isclose(m.cube_mass, spec["mass"], rtol=1e-6)
    assert np.isclose(m.cube_mass, 0.016, rtol=1e-6)                               # 500 kg/m^3 x 32 cm^3
    assert np.allclose(list(m.box_inertia), spec["inertia"], rtol=1e-6)
    assert np.isclose(m.cube_inertia, spec["inertia"].mean(), rtol=1e-6)           # I_ref of the scaled solve
    # CuboidalObject((0.02, 0.08, 0.02)) of the imported reference (tests/golden/constants.npz)
    g = golden("constants")
    assert np.isclose(m.obj_radius_3d, g["phase3_radius_3d"], rtol=1e-6)
    assert np.isclose(m.obj_max_com_dist, g["phase3_max_com_dist"], rtol=1e-6)
    assert np.isclose(m.obj_min_height, g["phase3_min_height"], rtol=1e-6)
    assert np.isclose(m.obj_span_min_height, g["phase3_max_height"] - g["phase3_min_height"], rtol=1e-6)
    assert np.isclose(m.obj_span_radius, g["phase3_max_height"] - g["phase3_radius_3d"], rtol=1e-6)
    d = oracle.default_model()                                                      # the cube keeps its literals
    assert d.box == 0 and d.obj_radius_3d == np.float32(0.05629165) and d.obj_max_com_dist == np.float32(0.13870835)
    assert d.obj_min_height == np.float32(0.0325) and d.obj_span_radius == np.float32(0.04370835)


def test_default_cube_asked_for_as_a_box_stays_the_cube(oracle):
    """include/trifinger.h: "a cubic size keeps box = 0 when it equals the default cube" - `native.object_size = [0.065] * 3`
    must not push an env onto the EXT kernels and the inertia-scaled arithmetic."""
    import ctypes
    d, m = oracle.default_model(), oracle.box_model((0.065, 0.065, 0.065), 291.3)
    assert m.box == 0 and bytes(ctypes.string_at(ctypes.byref(m), ctypes.sizeof(m))) == bytes(ctypes.string_at(ctypes.byref(d), ctypes.sizeof(d)))
    assert oracle.box_model((0.065, 0.065, 0.065), 500.0).box == 1          # another density is another object
    assert oracle.box_model((0.065, 0.065, 0.0651), 291.3).box == 1


def _rest(lib, device):
    """the box lying on each kind of face stays where it is, at the height of that half extent"""
    for up_axis, quat in ((2, (0, 0, 0, 1)), (1, (np.sin(np.pi / 4), 0, 0, np.cos(np.pi / 4))),
                          (0, (0, np.sin(np.pi / 4), 0, np.cos(np.pi / 4)))):
        eng = T.engine(lib, device=device, model_edit=box_edit(lib))
        half = np.array(PHASE3) / 2
        # the fingers are not driven and come to rest with their tips on the floor 7 cm from the centre: the box sits inside
        eng.cube[0:3, 0] = torch.tensor([0.004, -0.003, half[up_axis]], dtype=torch.float32, device=device)
        eng.cube[3:7, 0] = torch.tensor(quat, dtype=torch.float32, device=device)
        for _ in range(100):
            eng.step(torch.zeros(1, 9, device=device))
        c = eng.cube[:, 0].cpu().numpy()
        assert abs(c[2] - half[up_axis]) < 2e-4, (up_axis, c[2])
        assert np.abs(c[0:2] - [0.004, -0.003]).max() < 2e-4 and np.abs(c[7:13]).max() < 2e-2
        assert np.abs(c[3:7] - np.array(quat) * np.sign(np.dot(c[3:7], quat))).max() < 2e-3
        eng.close()


def test_box_rests_on_every_face(oracle):
    _rest(oracle, "cpu")


@pytest.mark.gpu
def test_box_rests_on_every_face_gpu(hip):
    _rest(hip, "cuda:0")


def _free_rotation(lib, device, gyroscopic):
    size = (0.03, 0.09, 0.03)                                   # symmetric top: I_x = I_z
    eng = T.engine(lib, device=device, model_edit=box_edit(lib, size, 500.0, box_gyroscopic=int(gyroscopic),
                                                          cube_angular_damping=0.0), gravity=(0.0, 0.0, 0.0),
                   dt=0.004, substeps=4)
    spec = PR.box_spec(size, 500.0)
    w0 = np.array([3.0, 4.0, 1.0])                              # body = world at t = 0
    eng.cube[0:3, 0] = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float32, device=device)       # far from everything
    eng.cube[10:13, 0] = torch.tensor(w0, dtype=torch.float32, device=device)
    steps = 125                                                 # 0.5 s in substeps of 1 ms
    for _ in range(steps):
        eng.step(torch.zeros(1, 9, device=device))
    c = eng.cube[:, 0].cpu().numpy().astype(np.float64)
    eng.close()
    R = PR.quat_rot(c[3:7])
    return spec, w0, R, c[10:13], steps * 0.004


def _check_free_rotation(lib, device):
    spec, w0, R, w, t = _free_rotation(lib, device, True)
    Ib = spec["inertia"]
    L0 = Ib * w0
    L = R @ (Ib * (R.T @ w))
    assert np.linalg.norm(L - L0) < 0.01 * np.linalg.norm(L0)                      # angular momentum is conserved
    wb = R.T @ w
    assert abs(wb[1] - w0[1]) < 0.01 * abs(w0[1])                                  # spin about the symmetry axis
    # analytic symmetric top (symmetry axis y): dwx/dt = -k wz, dwz/dt = k wx with k = (I_t - I_s) / I_t * w_s, i.e. the
    # transverse body component keeps its length and its angle atan2(wx, wz) decreases at the rate k
    rate = (Ib[0] - Ib[1]) / Ib[0] * w0[1]
    ang0 = np.arctan2(w0[0], w0[2])
    want = np.hypot(w0[0], w0[2]) * np.array([np.sin(ang0 - rate * t), np.cos(ang0 - rate * t)])
    assert np.abs(np.array([wb[0], wb[2]]) - want).max() < 0.03 * np.hypot(w0[0], w0[2]), (wb, want)
    # without the term (what PhysX does unless asked): the world angular velocity just stays
    spec, w0, R, w, t = _free_rotation(lib, device, False)
    assert np.abs(w - w0).max() < 1e-3                      # 500 fp32 round trips through the scaled coordinates


def test_torque_free_symmetric_top(oracle):
    _check_free_rotation(oracle, "cpu")


@pytest.mark.gpu
def test_torque_free_symmetric_top_gpu(hip):
    _check_free_rotation(hip, "cuda:0")


def make_box_case(rng, half):
    """a random state with at least one link capsule within +-4 mm of the box and nothing penetrating deeper"""
    while True:
        q = np.concatenate([rng.uniform(PR.Q_LO + 0.05, PR.Q_HI - 0.05) for _ in range(3)])
        f0 = rng.integers(3)
        tip = PR.link_point_world(f0, q[3 * f0:3 * f0 + 3], 3, PR.TIP_CAP[2])
        if tip[2] < 0.012:
            continue
        d = rng.normal(size=3)
        if rng.random() < 0.6:                                   # lying on one of its faces next to the fingertip
            up = rng.integers(3)
            yaw = rng.uniform(0, 2 * np.pi)
            base = {2: (0, 0, 0, 1), 1: (np.sin(np.pi / 4), 0, 0, np.cos(np.pi / 4)), 0: (0, np.sin(np.pi / 4), 0, np.cos(np.pi / 4))}[up]
            qz = np.array([0, 0, np.sin(yaw / 2), np.cos(yaw / 2)])
            b = np.array(base, dtype=np.float64)
            cq = np.array([qz[3] * b[0] + qz[0] * b[3] + qz[1] * b[2] - qz[2] * b[1],
                           qz[3] * b[1] - qz[0] * b[2] + qz[1] * b[3] + qz[2] * b[0],
                           qz[3] * b[2] + qz[0] * b[1] - qz[1] * b[0] + qz[2] * b[3],
                           qz[3] * b[3] - qz[0] * b[0] - qz[1] * b[1] - qz[2] * b[2]])
            d[2] = abs(d[2]) * 0.3
            d /= np.linalg.norm(d)
            c = tip - d * (half.max() * rng.uniform(0.3, 1.3) + 0.0102)
            c[2] = half[up] + rng.uniform(-0.0005, 0.001)
        else:
            cq = L._rand_quat(rng)
            d /= np.linalg.norm(d)
            c = tip - d * (half.max() * rng.uniform(0.3, 1.5) + 0.0102)
            if c[2] < 0.06:
                continue
        if np.hypot(c[0], c[1]) > 0.15:
            continue
        R = PR.quat_rot(cq)
        ok, near = True, False
        for f in range(3):
            qf = q[3 * f:3 * f + 3]
            for g, _ in PR.finger_gaps(f, qf, c, R, half):
                ok &= g >= -0.004
                near |= g < 0.004
            tipf = PR.link_point_world(f, qf, 3, PR.TIP_CAP[2])
            ok &= tipf[2] - 0.0102 >= -0.003
            ok &= PR.wall_radius_at(tipf[2]) - np.hypot(tipf[0], tipf[1]) - 0.0102 >= -0.003
        if not ok or not near:
            continue
        return q, rng.uniform(-2, 2, 9), np.concatenate([c, cq, rng.uniform(-0.3, 0.3, 3), rng.uniform(-3, 3, 3)]), rng.uniform(-0.36, 0.36, 9)


def _box_substep(lib, device, q, qd, cube, tau, sweeps):
    eng = T.engine(lib, device=device, model_edit=box_edit(lib), dt=L.H, substeps=1, solver_iterations=sweeps)
    f32 = dict(dtype=torch.float32, device=device)
    eng.q[:, 0] = torch.tensor(q, **f32)
    eng.qd[:, 0] = torch.tensor(qd, **f32)
    eng.cube[:, 0] = torch.tensor(cube, **f32)
    eng.tau[:, 0] = torch.tensor(tau, **f32)
    eng.simulate()
    st = eng.state[:, 0].cpu().numpy().astype(np.float64)
    eng.close()
    return st[9:18], st[25:28], st[28:31]


def _box_lcp(lib, device, n_cases):
    rng = np.random.default_rng(424242)
    spec = PR.box_spec(PHASE3, DENSITY)
    errs, live, floor = [], 0, 0
    for _ in range(n_cases):
        q, qd, cube, tau = make_box_case(rng, spec["half"])
        ref = PR.ref_substep(q, qd, cube, tau, L.H, max_sweeps=100000, box=spec)
        assert ref[3]["sweeps"] < 100000
        live += any(x[3].lam > 0 for x in ref[3]["fc"])
        floor += ref[3]["n_floor"] > 0
        errs.append([L.scaled_error(_box_substep(lib, device, q, qd, cube, tau, k), ref[:3]) for k in (8, 64, 2048)])
    errs = np.array(errs)
    print("\nbox: sweeps 8 / 64 / 2048  median", np.median(errs, axis=0), " p90", np.percentile(errs, 90, axis=0), " max", errs.max(axis=0),
          f" live finger contacts {live}, floor {floor} of {n_cases}")
    assert live >= n_cases // 4 and floor >= n_cases // 4
    # converged: the inertia-scaled solve agrees with the full-tensor fp64 solution (a 16 g bar spins up to hundreds of rad/s
    # under a finger: the angular error is measured relative to 20 rad/s like everywhere else)
    assert np.percentile(errs[:, 2], 90) < 1e-4 and errs[:, 2].max() < 5e-3, errs[:, 2].max()
    assert np.median(errs[:, 0]) >= np.median(errs[:, 2]) and np.percentile(errs[:, 1], 90) <= np.percentile(errs[:, 0], 90) + 1e-9


def test_box_substep_converges_to_the_full_tensor_solution(oracle):
    _box_lcp(oracle, "cpu", 60)


@pytest.mark.gpu
def test_box_substep_converges_to_the_full_tensor_solution_gpu(hip):
    _box_lcp(hip, "cuda:0", 20)


def test_env_with_the_phase3_cuboid(backend):
    """`native.object_size` through the public API: spawn height, goal heights and the xy range follow the CuboidalObject
    numbers of the box (trifinger_env.py:1171-1172, 1228-1236 with utils.py:122-131)."""
    lib, dev = backend
    env = TrifingerEnv(config={"num_instances": 2048, "command_mode": "torque", "task_difficulty": 4, "seed": 5,
                               "native": {"object_size": list(PHASE3), "object_density": DENSITY}},
                       device=dev, verbose=False, lib=lib)
    env.reset()
    st = env._engine.state.cpu().numpy()
    r3 = max(PHASE3) * np.sqrt(3) / 2
    assert np.abs(st[capi.S_CUBE_P + 2] - 0.01).max() < 1e-3                       # lying on its 2 cm side
    assert np.hypot(st[capi.S_CUBE_P], st[capi.S_CUBE_P + 1]).max() <= 0.195 - r3 + 1e-3
    gz = st[capi.S_GOAL_P + 2]
    assert gz.min() >= r3 - 1e-6 and gz.max() <= 0.1 + 1e-6 and gz.std() > 0.005  # random_z(radius_3d, max_height)
    for _ in range(20):
        obs, rew, _, _ = env.step(torch.rand(2048, 9, device=dev) * 2 - 1)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    env.close()


@pytest.mark.gpu
def test_box_rollout_bit_exact(hip, oracle):
    """the HIP box kernels against the oracle, every per-env output bit for bit, with contacts, time-outs and DR"""
    for cfg_name in ("d4_torque_asym", "d4_domain_randomization"):
        got = pu.rollout(hip, "cuda:0", 600, 90, cfg_name, extra=dict(model=hip.box_model(PHASE3, DENSITY)))
        want = pu.rollout(oracle, "cpu", 600, 90, cfg_name, extra=dict(model=oracle.box_model(PHASE3, DENSITY)))
        for t, (a, b) in enumerate(zip(got, want)):
            pu.assert_bit_equal(a, b, f"box {cfg_name} step {t}")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 65])
def test_ext_kernels_ragged_sizes(hip, oracle, n):
    """the EXT kernel instantiations (box object + extended domain randomisation together) on a single env and on a ragged
    last workgroup, split path included: bit for bit"""
    extra = dict(model=None, domain_randomization={"activate": True, "cube_mass": (0.8, 1.2), "cube_size": (0.9, 1.1), "friction": (0.7, 1.3),
                                                   "robot_base_position": (0.01, 0.01, 0.003), "stage_position": (0.01, 0.02),
                                                   "friction_robot": (0.8, 1.2), "friction_object": (0.6, 1.4), "friction_stage": (0.7, 1.3)})
    got = pu.rollout(hip, "cuda:0", n, 60, "d4_torque_asym", episode_length=25, extra=dict(extra, model=hip.box_model(PHASE3, DENSITY)))
    want = pu.rollout(oracle, "cpu", n, 60, "d4_torque_asym", episode_length=25, extra=dict(extra, model=oracle.box_model(PHASE3, DENSITY)))
    for t, (a, b) in enumerate(zip(got, want)):
        pu.assert_bit_equal(a, b, f"EXT n={n} step {t}")
