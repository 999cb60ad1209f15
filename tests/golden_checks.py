"""Checks of a trifinger C-ABI library against the committed golden fixtures (tests/golden/*.npz).

Every function takes `(lib, device)`: the oracle on 'cpu' (CPU suite: pins the oracle) or the HIP
library on 'cuda:0' (GPU suite: pins the product).  All calls go through the C ABI.

Tolerances (fp32): abs 2e-6 / rel 1e-5 unless noted; quat_diff_rad near pi 2e-3 (asin slope);
samplers 2e-6 (own sincos vs torch's).
"""
import ctypes as C

import numpy as np
import torch

from leibnizgym_amd import _capi as capi
from leibnizgym_amd.engine import TrifingerEngine, make_config
from oracle_util import golden

ATOL, RTOL = 2e-6, 1e-5


def T(a, device, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(device).contiguous()


def P(t):
    return C.c_void_p(t.data_ptr())


def sync(device):
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()


def close(a, b, atol=ATOL, rtol=RTOL, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b) - (atol + rtol * np.abs(b))
    assert np.all(err <= 0), f"{what}: max excess {err.max():.3e}, max abs diff {np.abs(a - b).max():.3e}"


# ---------------------------------------------------------------------------------------------
def check_math(lib, device):
    g = golden("math")
    a, b = T(g["quat_a"], device), T(g["quat_b"], device)
    n = a.shape[0]
    out4 = torch.empty_like(a)
    assert lib.tf_test_quat_mul(P(a), P(b), P(out4), n, None) == 0
    sync(device)
    close(out4.cpu().numpy(), g["quat_mul"], what="quat_mul")
    out1 = torch.empty(n, device=device)
    assert lib.tf_test_quat_diff_rad(P(a), P(b), P(out1), n, None) == 0
    sync(device)
    got, want = out1.cpu().numpy(), g["quat_diff_rad"]
    near_pi = want > 3.0
    close(got[~near_pi], want[~near_pi], atol=5e-6, what="quat_diff_rad")
    close(got[near_pi], want[near_pi], atol=2e-3, what="quat_diff_rad near pi")
    # yaw-only euler rows == sample_yaw_quat(u) with yaw = 2 pi u
    rpy = g["euler_rpy"][:8]
    u = T(rpy[:, 2] / np.float32(6.2831855), device)
    q = torch.empty(8, 4, device=device)
    assert lib.tf_test_sample_yaw_quat(P(u), P(q), 8, None) == 0
    sync(device)
    close(q.cpu().numpy(), g["euler_quat"][:8], atol=1e-6, what="quaternion_from_euler_xyz (yaw)")


def check_lgsk(lib, device):
    g = golden("rewards")
    x = T(g["lgsk_x"], device)
    y = torch.empty_like(x)
    for scale, key in ((50.0, "lgsk_y50"), (3.0, "lgsk_y3")):
        assert lib.tf_test_lgsk(P(x), scale, P(y), x.numel(), None) == 0
        sync(device)
        close(y.cpu().numpy(), g[key], atol=1e-12, rtol=2e-5, what=f"lgsk scale {scale}")


def check_samplers(lib, device):
    g = golden("samplers")
    n = g["xy_x"].shape[0]
    ur, ut = T(g["xy_u_radius"], device), T(g["xy_u_theta"], device)
    x, y = torch.empty(n, device=device), torch.empty(n, device=device)
    assert lib.tf_test_sample_xy(P(ur), P(ut), float(g["xy_rmax"]), P(x), P(y), n, None) == 0
    sync(device)
    close(x.cpu().numpy(), g["xy_x"], atol=2e-7, rtol=2e-6, what="random_xy.x")
    close(y.cpu().numpy(), g["xy_y"], atol=2e-7, rtol=2e-6, what="random_xy.y")
    u = T(g["yaw_u"], device)
    q = torch.empty(n, 4, device=device)
    assert lib.tf_test_sample_yaw_quat(P(u), P(q), n, None) == 0
    sync(device)
    close(q.cpu().numpy(), g["yaw_quat"], atol=1e-6, what="random_yaw_orientation")
    nrm = T(g["ori_normals"], device)
    assert lib.tf_test_normalize_quat(P(nrm), P(q), n, None) == 0
    sync(device)
    close(q.cpu().numpy(), g["ori_quat"], atol=2e-7, rtol=2e-6, what="random_orientation")


# ---------------------------------------------------------------------------------------------
def _engine(lib, device, n, **kw):
    kw.setdefault("success", {"activate": False})
    cfg = make_config(lib, n, **kw)
    return TrifingerEngine(cfg, device=device, lib=lib)


def check_torque(lib, device):
    """T4: (action, q, qd) -> applied torque, 3 command modes x normalize x safety damping."""
    g = golden("torque")
    n = g["q"].shape[0]
    for mode in ("torque", "position", "position_impedance"):
        for norm in (1, 0):
            for safe in (0, 1):
                eng = _engine(lib, device, n, command_mode=mode, normalize_action=bool(norm),
                              apply_safety_damping=bool(safe))
                eng.q.copy_(T(g["q"].T, device))
                eng.qd.copy_(T(g["qd"].T, device))
                eng.action_buf.copy_(T(g[f"{mode}_action"], device))
                eng.pre_step()
                sync(device)
                close(eng.tau.T.cpu().numpy(), g[f"{mode}_norm{norm}_safe{safe}"], atol=1e-6,
                      what=f"torque {mode} norm={norm} safe={safe}")
                eng.close()


def check_obs(lib, device):
    """T5: obs[41|50] / states[113|122] assembly + scale tables, for the slots that are inputs of the
    native step (q, qd, object pose/vel, goal, last action, dof force).  Fingertip slots are produced by
    the build's own FK and are checked in test_physics_analytic.py; here they are compared after
    re-normalising the FK tips with the golden scale table."""
    g = golden("obs")
    n = g["q"].shape[0]
    for mode in ("torque", "position", "position_impedance"):
        for na in (1, 0):
            for norm_obs in (True, False):
                eng = _engine(lib, device, n, command_mode=mode, normalize_action=bool(na), normalize_obs=norm_obs,
                              asymmetric_obs=True)
                tag = f"{mode}_na{na}"
                eng.q.copy_(T(g["q"].T, device))
                eng.qd.copy_(T(g["qd"].T, device))
                eng.cube.copy_(T(g["obj"].T, device))
                eng.goal.copy_(T(g["goal"].T, device))
                eng.action_buf.copy_(T(g[f"{mode}_action"], device))
                eng.tau.copy_(T(g["dof_force"].T, device))
                eng.view(capi.S_PREV_OBJ_P, 7).copy_(T(g["obj"][:, :7].T, device))
                eng.post_step()
                sync(device)
                obs = eng.obs.cpu().numpy()
                sts = eng.states.cpu().numpy()
                want_obs = g[f"{tag}_obs_norm"] if norm_obs else g[f"{tag}_obs_raw"]
                want_sts = g[f"{tag}_states_norm"] if norm_obs else g[f"{tag}_states_raw"]
                od = want_obs.shape[1]
                close(obs, want_obs, atol=2e-6, what=f"obs {tag} norm_obs={norm_obs}")
                # states: common prefix, object velocity, dof force
                close(sts[:, :od + 6], want_sts[:, :od + 6], atol=2e-6, what=f"states[:obs+6] {tag}")
                close(sts[:, od + 45:od + 54], want_sts[:, od + 45:od + 54], atol=2e-6, what=f"states dof force {tag}")
                # fingertip block: the scale table must be the golden one -> unscale with golden table gives raw FK
                if norm_obs:
                    lo, hi = g[f"{tag}_states_lo"], g[f"{tag}_states_hi"]
                    raw = sts * (hi - lo) * 0.5 + (lo + hi) * 0.5
                    eng2 = _engine(lib, device, n, command_mode=mode, normalize_action=bool(na), normalize_obs=False,
                                   asymmetric_obs=True)
                    eng2.state.copy_(eng.state)
                    eng2.action_buf.copy_(eng.action_buf)
                    eng2.post_step()
                    sync(device)
                    close(raw[:, od + 6:od + 45], eng2.states.cpu().numpy()[:, od + 6:od + 45], atol=5e-6,
                          what=f"fingertip block scale table {tag}")
                    eng2.close()
                eng.close()


REWARD_CFGS = {
    "d1": {
        "finger_move_penalty": {"activate": True, "weight": -0.1},
        "finger_reach_object_rate": {"activate": True, "norm_p": 2, "weight": -750},
        "object_dist": {"activate": True, "weight": 2000},
        "object_rot": {"activate": False, "weight": 300},
        "object_rot_delta": {"activate": False, "weight": -250},
        "object_move": {"activate": False, "weight": -750},
    },
    "d4": {
        "finger_move_penalty": {"activate": True, "weight": -0.1},
        "finger_reach_object_rate": {"activate": True, "norm_p": 2, "weight": -250,
                                     "thresh_sched_start": 0, "thresh_sched_end": 1e7},
        "object_dist": {"activate": True, "weight": 2000, "thresh_sched_start": 0, "thresh_sched_end": 10e10},
        "object_rot": {"activate": True, "weight": 2000, "epsilon": 0.01, "scale": 3.0,
                       "thresh_sched_start": 1e7, "thresh_sched_end": 1e10},
        "object_rot_delta": {"activate": False, "weight": -250},
        "object_move": {"activate": False, "weight": -750},
    },
    "envdef": {
        "finger_reach_object_rate": {"activate": True, "weight": -750, "norm_p": 2},
        "finger_move_penalty": {"activate": True, "weight": -0.1},
        "object_dist": {"activate": True, "weight": 2000},
        "object_rot": {"activate": True, "weight": 300},
        "object_rot_delta": {"activate": True, "weight": -250},
        "object_move": {"activate": True, "weight": -750},
    },
    "linsched": {
        "finger_reach_object_rate": {"activate": True, "weight": -750, "norm_p": 2},
        "finger_move_penalty": {"activate": True, "weight": -0.1},
        "object_dist": {"activate": True, "weight": 2000},
        "object_rot": {"activate": True, "weight": 300, "scale": 1.0},
        "object_rot_delta": {"activate": True, "weight": -250,
                             "linear_schedule_start": 5e6, "linear_schedule_end": 1.5e7},
        "object_move": {"activate": True, "weight": -750},
    },
}


def check_rewards(lib, device):
    """T6/T7: per-env total reward and per-term means for 4 configs x 5 schedule points.

    Object terms (object_dist, object_rot, object_rot_delta, object_move) are compared directly with
    the golden values of the imported reference terms.  The two fingertip terms take FK tips and are
    covered by `check_finger_rewards`."""
    g = golden("rewards")
    obj, obj_prev, goal = g["obj"], g["obj_prev"], g["goal"]
    n = obj.shape[0]
    steps = g["sched_steps"]
    order = list(capi.REWARD_TERM_ORDER)
    assert [str(s) for s in g["term_order"]] == order
    for cname, terms in REWARD_CFGS.items():
        per_all, means_all = g[f"{cname}_per_term"], g[f"{cname}_means"]
        for si, step in enumerate(steps):
            # object-only config: switch the two finger terms off, keep every object term as configured
            obj_terms = {k: dict(v) for k, v in terms.items()}
            obj_terms["finger_reach_object_rate"]["activate"] = False
            obj_terms["finger_move_penalty"]["activate"] = False
            # env_steps_count = frame_count * global_num_envs (env_base.py:287-289): with global_num_envs=1 the
            # frame counter itself is the schedule step
            cfg = make_config(lib, n, command_mode="torque", reward_terms=obj_terms, dt=float(g["dt"]),
                              success={"activate": False}, global_num_envs=1)
            eng = TrifingerEngine(cfg, device=device, lib=lib)
            eng.cube.copy_(T(obj.T, device))
            eng.goal.copy_(T(goal.T, device))
            eng.view(capi.S_PREV_OBJ_P, 7).copy_(T(obj_prev[:, :7].T, device))
            eng.frame_count = int(step)
            eng.post_step()
            sync(device)
            want = np.zeros(n, dtype=np.float64)
            for k, name in enumerate(order):
                if obj_terms[name]["activate"]:
                    want += per_all[si, k].astype(np.float64)
            close(eng.reward.cpu().numpy(), want, atol=2e-4, rtol=2e-5, what=f"object rewards {cname} step={step}")
            info = eng.info.cpu().numpy()
            for k, name in enumerate(order):
                if obj_terms[name]["activate"]:
                    close(info[k], means_all[si, k], atol=2e-4, rtol=2e-5, what=f"mean {name} {cname} step={step}")
            eng.close()


def check_finger_rewards(lib, device):
    """T7 fingertip terms against the IMPORTED reference classes (tests/golden/finger_rewards.npz): the fixture holds joint
    positions, the fingertip positions the native FK gives for them, and the values of FingerReachObjectRatePenalty
    (norm_p in {2, 1, 3, inf}, rewards.py:187-235) and FingertipMovementPenalty (:238-270) on those fingertips.  The test
    replays the joint positions - previous frame, then current frame - and compares fingertips and rewards."""
    g = golden("finger_rewards")
    n = g["q"].shape[0]
    for k, p in enumerate(g["norm_ps"]):
        for active in ("reach", "move"):
            terms = {name: {"activate": False} for name in capi.REWARD_TERM_ORDER}
            if active == "reach":
                terms["finger_reach_object_rate"] = {"activate": True, "weight": float(g["reach_weight"]), "norm_p": float(p)}
            elif k == 0:
                terms["finger_move_penalty"] = {"activate": True, "weight": float(g["move_weight"])}
            else:
                continue
            cfg = make_config(lib, n, command_mode="torque", reward_terms=terms, dt=float(g["dt"]), success={"activate": False})
            eng = TrifingerEngine(cfg, device=device, lib=lib)
            eng.q.copy_(T(g["q_prev"].T, device))
            eng.post_step()                                   # leaves the previous frame's fingertips in the history row
            sync(device)
            close(eng.tip_pos.T.cpu().numpy().reshape(n, 3, 3), g["tips_prev"], atol=1e-7, rtol=0, what="FK tips (previous frame)")
            eng.q.copy_(T(g["q"].T, device))
            eng.cube.copy_(T(g["obj"].T, device))
            eng.view(capi.S_PREV_OBJ_P, 7).copy_(T(g["obj_prev"][:, :7].T, device))
            eng.post_step()
            sync(device)
            close(eng.tip_pos.T.cpu().numpy().reshape(n, 3, 3), g["tips"], atol=1e-7, rtol=0, what="FK tips")
            want = g[f"reach_{k}"] if active == "reach" else g["move"]
            # values of order 10 built from differences of distances of order 0.1 m times 750: 2e-4 abs is fp32 rounding
            close(eng.reward.cpu().numpy(), want, atol=3e-4, rtol=2e-5, what=f"{active} term vs imported reference, norm_p={p}")
            eng.close()


def check_finger_reach_small_distances(lib, device):
    """FingerReachObjectRatePenalty with a large p when the fingertips are millimetres from the object centre (rewards.py:203-235:
    torch.norm(d, p)): d^p of a 2-4 mm distance underflows fp32 from p = 10 on, so the p-norm is formed on d / max|d|.  Reference
    value: the same expression in float64."""
    g = golden("finger_rewards")
    n = g["q"].shape[0]
    rng = np.random.default_rng(5)
    for p in (3, 10, 16):
        terms = {name: {"activate": False} for name in capi.REWARD_TERM_ORDER}
        terms["finger_reach_object_rate"] = {"activate": True, "weight": -750.0, "norm_p": float(p)}
        cfg = make_config(lib, n, command_mode="torque", reward_terms=terms, dt=0.02, success={"activate": False})
        eng = TrifingerEngine(cfg, device=device, lib=lib)
        eng.q.copy_(T(g["q_prev"].T, device))
        eng.post_step()
        eng.q.copy_(T(g["q"].T, device))
        # the object centre 1-4 mm from fingertip 0 in both frames (a physically impossible pose: the reward arithmetic alone is tested)
        off_prev = rng.uniform(-1.0, 1.0, (n, 3)) * 0.002 + 0.001
        off_now = rng.uniform(-1.0, 1.0, (n, 3)) * 0.003 + 0.0005
        obj_prev = (g["tips_prev"][:, 0, :] + off_prev).astype(np.float32)
        obj_now = (g["tips"][:, 0, :] + off_now).astype(np.float32)
        eng.view(capi.S_CUBE_P, 3).copy_(T(obj_now.T, device))
        eng.view(capi.S_PREV_OBJ_P, 3).copy_(T(obj_prev.T, device))
        eng.post_step()
        sync(device)
        tips, tips_prev = g["tips"].astype(np.float64), g["tips_prev"].astype(np.float64)
        norm = lambda d: (np.abs(d) ** p).sum(-1) ** (1.0 / p)      # noqa: E731
        want = -750.0 * sum(norm(tips[:, f] - obj_now.astype(np.float64)) - norm(tips_prev[:, f] - obj_prev.astype(np.float64)) for f in range(3))
        d0 = norm(tips[:, 0] - obj_now.astype(np.float64))
        assert d0.min() < 0.004 and (np.abs(tips[:, 0] - obj_now) ** p).sum(-1).min() < (1e-38 if p >= 16 else 1.0)
        close(eng.reward.cpu().numpy(), want, atol=3e-4, rtol=2e-5, what=f"reach term at mm distances, norm_p={p}")
        eng.close()


def check_termination(lib, device):
    """T8: flags, counts, bonus, successes for difficulty {1,4,5} x activate {T,F}."""
    g = golden("termination")
    n = g["obj_p"].shape[0]
    none = {k: {"activate": False} for k in capi.REWARD_TERM_ORDER}
    for diff in (1, 4, 5):
        for act in (1, 0):
            cfg = make_config(lib, n, command_mode="torque", task_difficulty=diff, reward_terms=none,
                              success={"activate": bool(act), "bonus": float(g["bonus"]),
                                       "position_tolerance": float(g["pos_tol"]),
                                       "orientation_tolerance": float(g["ori_tol"])})
            eng = TrifingerEngine(cfg, device=device, lib=lib)
            eng.view(capi.S_CUBE_P, 3).copy_(T(g["obj_p"].T, device))
            eng.view(capi.S_CUBE_Q, 4).copy_(T(g["obj_q"].T, device))
            eng.view(capi.S_GOAL_P, 3).copy_(T(g["goal_p"].T, device))
            eng.view(capi.S_GOAL_Q, 4).copy_(T(g["goal_q"].T, device))
            eng.view(capi.S_PREV_OBJ_P, 3).copy_(T(g["obj_p"].T, device))
            eng.view(capi.S_PREV_OBJ_Q, 4).copy_(T(g["obj_q"].T, device))
            eng.successes.copy_(torch.as_tensor(g["successes_in"]).to(device))
            eng.goal_reset_buf.copy_(torch.as_tensor(g["goal_reset_in"]).to(device))
            eng.post_step()
            sync(device)
            tag = f"d{diff}_act{act}"
            # the golden adds the bonus to a random reward_in; all reward terms are off here -> bonus only
            want_bonus = g[f"{tag}_reward"] - g["reward_in"]
            close(eng.reward.cpu().numpy(), want_bonus, atol=1e-3, what=f"bonus {tag}")
            assert np.array_equal(eng.goal_reset_buf.cpu().numpy(), g[f"{tag}_goal_reset"]), tag
            assert np.array_equal(eng.successes.cpu().numpy(), g[f"{tag}_successes"]), tag
            info = eng.info.cpu().numpy()
            assert info[capi.INFO_POS_COUNT] == float(g["pos_count"])
            assert info[capi.INFO_ORI_COUNT] == float(g["ori_count"])
            close(info[capi.INFO_SUCCESS_MEAN], g[f"{tag}_succ_mean"], atol=1e-6, what=f"succ mean {tag}")
            eng.close()


def check_constants(lib, device):
    """T13: sampling radii/heights used by the native resets equal CuboidalObject(0.065)'s numbers."""
    g = golden("constants")
    n = 4096
    for diff, zlo, zhi in ((1, g["min_height"], g["min_height"]), (3, g["min_height"], g["max_height"]),
                           (4, g["radius_3d"], g["max_height"]), (2, 0.0825, 0.0825)):
        cfg = make_config(lib, n, command_mode="torque", task_difficulty=diff, seed=11)
        eng = TrifingerEngine(cfg, device=device, lib=lib)
        eng.reset()
        sync(device)
        goal = eng.goal.cpu().numpy()
        rad = np.hypot(goal[0], goal[1])
        if diff == 2:
            assert np.all(rad == 0)
        else:
            assert rad.max() <= float(g["max_com_distance"]) * (1 + 1e-6)
            assert rad.max() > 0.98 * float(g["max_com_distance"])
        assert goal[2].min() >= float(zlo) - 1e-7 and goal[2].max() <= float(zhi) + 1e-7
        cube = eng.cube.cpu().numpy()
        # after one simulate the cube has barely moved from its spawn pose
        assert np.hypot(cube[0], cube[1]).max() <= float(g["max_com_distance"]) + 2e-3
        eng.close()
