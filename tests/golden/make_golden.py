#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own functions.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference's physics lives in closed-source IsaacGym and cannot be imported, but
its task-layer leaf modules can be loaded by path (tests/golden/_ref_loader.py).  Every
array stored here is either a seeded input or the output of an *imported reference
function* (or a composition of imported reference functions in the order of the cited
env-method lines).  Nothing from the reference's source text is stored - only numbers.

Fixtures written (all float32 unless noted):
  math.npz        T12  scale/unscale/saturate/quat_mul/quat_conjugate/quat_diff_rad/euler
  rewards.npz     T7   lgsk_kernel + six reward terms, difficulty-1 and difficulty-4 configs,
                       at env_steps_count in {0, 9.99e6, 1e7, 1.0001e7, 2e10}
  samplers.npz    T11  samplers as functions of the uniform/normal draws they consume
  torque.npz      T4   pre-step torque law (3 command modes x safety on/off)
  obs.npz         T5   obs[41]/states[113] assembly + scale tables
  termination.npz T8   success flags / counts / bonus for difficulty {1,4,5} x activate {T,F}
  constants.npz   T13  CuboidalObject(0.065) numbers, dimension enum values
  finger_rewards.npz T7 the two fingertip terms (FingerReachObjectRatePenalty with norm_p in {2, 1, 3, inf},
                       FingertipMovementPenalty) of the imported reference evaluated on fingertip positions that the
                       native step itself produced (forward kinematics of the stored joint positions, run here through
                       the CPU oracle): the fixture pins term(tips), the tips are whatever the engine's FK gives
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_loader import load_reference  # noqa: E402

ref = load_reference()
tu = ref.torch_utils
rw = ref.rewards
sm = ref.sample

N = 64
F32 = torch.float32


def npf(t):
    return t.detach().cpu().numpy().astype(np.float32)


def unit_quats(n, gen):
    q = torch.randn(n, 4, generator=gen, dtype=F32)
    return q / q.norm(dim=-1, keepdim=True)


# --------------------------------------------------------------------------------------
# Reference constants (data transcribed from trifinger_env.py:149-224; numbers only)
# --------------------------------------------------------------------------------------
MAX_TORQUE = 0.36
MAX_VEL = 10.0
Q_LO = torch.tensor([-0.33, 0.0, -2.7] * 3, dtype=F32)
Q_HI = torch.tensor([1.0, 1.57, 0.0] * 3, dtype=F32)
QD_LO = torch.full((9,), -MAX_VEL, dtype=F32)
QD_HI = torch.full((9,), MAX_VEL, dtype=F32)
TAU_LO = torch.full((9,), -MAX_TORQUE, dtype=F32)
TAU_HI = torch.full((9,), MAX_TORQUE, dtype=F32)
STIFF_LO = torch.tensor([1.0] * 9, dtype=F32)
STIFF_HI = torch.tensor([50.0] * 9, dtype=F32)
KP = torch.tensor([10.0] * 9, dtype=F32)
KD = torch.tensor([0.1, 0.3, 0.001] * 3, dtype=F32)
KS = torch.tensor([0.08, 0.08, 0.04] * 3, dtype=F32)
TIP_POS_LO = torch.tensor([-0.4, -0.4, 0.0], dtype=F32)
TIP_POS_HI = torch.tensor([0.4, 0.4, 0.5], dtype=F32)
ORI_LO = -torch.ones(4, dtype=F32)
ORI_HI = torch.ones(4, dtype=F32)
TIP_VEL_LO = torch.full((6,), -0.2, dtype=F32)
TIP_VEL_HI = torch.full((6,), 0.2, dtype=F32)
WRENCH_LO = torch.full((6,), -1.0, dtype=F32)
WRENCH_HI = torch.full((6,), 1.0, dtype=F32)
OBJ_POS_LO = torch.tensor([-0.3, -0.3, 0.0], dtype=F32)
OBJ_POS_HI = torch.tensor([0.3, 0.3, 0.3], dtype=F32)
OBJ_VEL_LO = torch.full((6,), -0.5, dtype=F32)
OBJ_VEL_HI = torch.full((6,), 0.5, dtype=F32)


def gen_math():
    g = torch.Generator().manual_seed(1001)
    out = {}
    # scale / unscale / saturate on 9-dim rows with the joint-position limits
    x = (torch.rand(N, 9, generator=g, dtype=F32) * 2 - 1) * 3.0
    out["scale_x"] = npf(x)
    out["scale_lo"] = npf(Q_LO)
    out["scale_hi"] = npf(Q_HI)
    out["scale_y"] = npf(tu.scale_transform(x, Q_LO, Q_HI))
    out["unscale_y"] = npf(tu.unscale_transform(x, Q_LO, Q_HI))
    out["saturate_y"] = npf(tu.saturate(x, Q_LO, Q_HI))
    # quaternions (xyzw).  Edge rows: identical, antipodal (-q), theta ~ pi, tiny angle
    a = unit_quats(N, g)
    b = unit_quats(N, g)
    b[0] = a[0]
    b[1] = -a[1]
    a[2] = torch.tensor([0.0, 0.0, 0.0, 1.0])
    b[2] = torch.tensor([0.0, 0.0, 1.0, 0.0])            # theta = pi
    a[3] = torch.tensor([0.0, 0.0, 0.0, 1.0])
    b[3] = torch.tensor([1e-4, 0.0, 0.0, 1.0]) / np.sqrt(1 + 1e-8)   # tiny angle
    a[4] = torch.tensor([0.0, 0.0, 0.0, 1.0])
    b[4] = torch.tensor([0.0, 0.70710678, 0.0, 0.70710678])          # 90 deg
    a[5] = torch.tensor([0.0, 0.0, 0.0, 1.0])
    b[5] = torch.tensor([0.0, 0.0, 0.9999999, 0.0004472])          # just short of pi
    out["quat_a"] = npf(a)
    out["quat_b"] = npf(b)
    out["quat_mul"] = npf(tu.quat_mul(a, b))
    out["quat_conj"] = npf(tu.quat_conjugate(a))
    out["quat_diff_rad"] = npf(tu.quat_diff_rad(a, b))
    # euler -> quat
    roll = (torch.rand(N, generator=g, dtype=F32) * 2 - 1) * np.pi
    pitch = (torch.rand(N, generator=g, dtype=F32) * 2 - 1) * np.pi
    yaw = torch.rand(N, generator=g, dtype=F32) * 2 * np.pi
    roll[:8] = 0.0
    pitch[:8] = 0.0
    out["euler_rpy"] = npf(torch.stack([roll, pitch, yaw], -1))
    out["euler_quat"] = npf(tu.quaternion_from_euler_xyz(roll, pitch, yaw))
    np.savez(os.path.join(HERE, "math.npz"), **out)


# reward-term configs: (name -> kwargs) as in scripts/rlg_hydra.py:83-109 (difficulty 1-3)
REWARDS_D1 = {
    "finger_move_penalty": {"activate": True, "weight": -0.1},
    "finger_reach_object_rate": {"activate": True, "norm_p": 2, "weight": -750},
    "object_dist": {"activate": True, "weight": 2000},
    "object_rot": {"activate": False, "weight": 300},
    "object_rot_delta": {"activate": False, "weight": -250},
    "object_move": {"activate": False, "weight": -750},
}
# scripts/rlg_hydra.py:140-174 (difficulty 4)
REWARDS_D4 = {
    "finger_move_penalty": {"activate": True, "weight": -0.1},
    "finger_reach_object_rate": {"activate": True, "norm_p": 2, "weight": -250,
                                 "thresh_sched_start": 0, "thresh_sched_end": 1e7},
    "object_dist": {"activate": True, "weight": 2000,
                    "thresh_sched_start": 0, "thresh_sched_end": 10e10},
    "object_rot": {"activate": True, "weight": 2000, "epsilon": 0.01, "scale": 3.0,
                   "thresh_sched_start": 1e7, "thresh_sched_end": 1e10},
    "object_rot_delta": {"activate": False, "weight": -250},
    "object_move": {"activate": False, "weight": -750},
}
# env default dict (trifinger_env.py:76-104): everything active
REWARDS_ENV_DEFAULT = {
    "finger_reach_object_rate": {"activate": True, "weight": -750, "norm_p": 2},
    "finger_move_penalty": {"activate": True, "weight": -0.1},
    "object_dist": {"activate": True, "weight": 2000},
    "object_rot": {"activate": True, "weight": 300},
    "object_rot_delta": {"activate": True, "weight": -250},
    "object_move": {"activate": True, "weight": -750},
}
# an extra config exercising the linear schedule of object_rot_delta (rewards.py:14-17,174-177)
REWARDS_LINSCHED = {
    "finger_reach_object_rate": {"activate": True, "weight": -750, "norm_p": 2},
    "finger_move_penalty": {"activate": True, "weight": -0.1},
    "object_dist": {"activate": True, "weight": 2000},
    "object_rot": {"activate": True, "weight": 300, "scale": 1.0},
    "object_rot_delta": {"activate": True, "weight": -250,
                         "linear_schedule_start": 5e6, "linear_schedule_end": 1.5e7},
    "object_move": {"activate": True, "weight": -750},
}
TERM_ORDER = ["finger_reach_object_rate", "finger_move_penalty", "object_dist",
              "object_rot", "object_rot_delta", "object_move"]
SCHED_STEPS = [0.0, 9.99e6, 1e7, 1.0001e7, 2e10]


def reward_inputs(g):
    tips = torch.zeros(N, 3, 13, dtype=F32)
    tips[:, :, 0:3] = (torch.rand(N, 3, 3, generator=g, dtype=F32) - 0.5) * 0.3
    tips[:, :, 2] = tips[:, :, 2].abs()
    tips[:, :, 3:7] = unit_quats(N * 3, g).view(N, 3, 4)
    tips[:, :, 7:13] = (torch.rand(N, 3, 6, generator=g, dtype=F32) - 0.5) * 0.4
    tips_prev = tips.clone()
    tips_prev[:, :, 0:3] += (torch.rand(N, 3, 3, generator=g, dtype=F32) - 0.5) * 0.01
    tips_prev[0] = 0.0                        # post-reset style zeroed history row
    obj = torch.zeros(N, 13, dtype=F32)
    obj[:, 0:3] = (torch.rand(N, 3, generator=g, dtype=F32) - 0.5) * 0.3
    obj[:, 2] = obj[:, 2].abs() + 0.0325
    obj[:, 3:7] = unit_quats(N, g)
    obj[:, 7:13] = (torch.rand(N, 6, generator=g, dtype=F32) - 0.5)
    obj_prev = obj.clone()
    obj_prev[:, 0:3] += (torch.rand(N, 3, generator=g, dtype=F32) - 0.5) * 0.004
    dq = unit_quats(N, g) * 0.02
    dq[:, 3] = 1.0
    dq = dq / dq.norm(dim=-1, keepdim=True)
    obj_prev[:, 3:7] = tu.quat_mul(obj[:, 3:7].contiguous(), dq)
    goal = torch.zeros(N, 7, dtype=F32)
    goal[:, 0:3] = (torch.rand(N, 3, generator=g, dtype=F32) - 0.5) * 0.3
    goal[:, 2] = goal[:, 2].abs() + 0.0325
    goal[:, 3:7] = unit_quats(N, g)
    # edge rows: zero distance / identical orientation / dist = 0.4 / antipodal quats
    goal[1, 0:3] = obj[1, 0:3]
    goal[2, 3:7] = obj[2, 3:7]
    goal[3, 0:3] = obj[3, 0:3] + torch.tensor([0.4, 0.0, 0.0])
    goal[4, 3:7] = -obj[4, 3:7]
    return tips, tips_prev, obj, obj_prev, goal


def eval_terms(cfg, step, dt, tips, tips_prev, obj, obj_prev, goal, scripted):
    terms = {}
    for name, kw in cfg.items():
        t = rw.REWARD_TERMS_MAPPING[name](name, **dict(kw))
        terms[name] = torch.jit.script(t) if scripted else t
    r = {
        "finger_reach_object_rate": terms["finger_reach_object_rate"].compute(step, tips, tips_prev, obj, obj_prev),
        "finger_move_penalty": terms["finger_move_penalty"].compute(dt, tips, tips_prev),
        "object_dist": terms["object_dist"].compute(dt, step, obj, goal),
        "object_rot": terms["object_rot"].compute(dt, step, obj, goal),
        "object_rot_delta": terms["object_rot_delta"].compute(dt, step, obj, obj_prev, goal),
        "object_move": terms["object_move"].compute(obj, obj_prev, goal),
    }
    per_term = torch.stack([r[k] for k in TERM_ORDER], dim=0)      # [6, N]
    total = torch.zeros(N, dtype=F32)
    means = []
    for k in TERM_ORDER:                                           # trifinger_env.py:551-554
        if terms[k].activate:
            total = total + r[k]
            means.append(r[k].mean())
        else:
            means.append(torch.tensor(float("nan")))
    return per_term, total, torch.stack(means)


def gen_rewards():
    g = torch.Generator().manual_seed(2002)
    out = {}
    x = torch.cat([torch.linspace(0, 0.45, 48), torch.tensor([0.0, 1e-4, 0.01, 0.02, 0.05, 0.1, 0.2, 0.4]),
                   torch.rand(8, generator=g) * 3.2]).to(F32)
    out["lgsk_x"] = npf(x)
    out["lgsk_y50"] = npf(rw.lgsk_kernel(x, 50.0))
    out["lgsk_y3"] = npf(rw.lgsk_kernel(x, 3.0))
    tips, tips_prev, obj, obj_prev, goal = reward_inputs(g)
    out["tips"], out["tips_prev"] = npf(tips), npf(tips_prev)
    out["obj"], out["obj_prev"], out["goal"] = npf(obj), npf(obj_prev), npf(goal)
    out["dt"] = np.float32(0.02)
    out["sched_steps"] = np.array(SCHED_STEPS, dtype=np.float64)
    out["term_order"] = np.array(TERM_ORDER)
    dt = 0.02
    for cname, cfg in (("d1", REWARDS_D1), ("d4", REWARDS_D4), ("envdef", REWARDS_ENV_DEFAULT),
                       ("linsched", REWARDS_LINSCHED)):
        per, tot, mean = [], [], []
        for step in SCHED_STEPS:
            p, t, m = eval_terms(cfg, step, dt, tips, tips_prev, obj, obj_prev, goal, scripted=True)
            p2, t2, _ = eval_terms(cfg, step, dt, tips, tips_prev, obj, obj_prev, goal, scripted=False)
            assert torch.equal(p, p2) and torch.equal(t, t2), "jit vs eager mismatch in the reference"
            per.append(npf(p)); tot.append(npf(t)); mean.append(npf(m))
        out[f"{cname}_per_term"] = np.stack(per)      # [S, 6, N]
        out[f"{cname}_total"] = np.stack(tot)         # [S, N]
        out[f"{cname}_means"] = np.stack(mean)        # [S, 6] (nan where inactive)
    np.savez(os.path.join(HERE, "rewards.npz"), **out)


def gen_samplers():
    out = {}
    n = N
    r_max = 0.1387083487540115
    # random_xy: draws radius-uniform then theta-uniform (sample.py:26,29)
    torch.manual_seed(3003)
    u_r = torch.rand(n, dtype=F32)
    u_t = torch.rand(n, dtype=F32)
    torch.manual_seed(3003)
    x, y = sm.random_xy(n, r_max, "cpu")
    out["xy_u_radius"], out["xy_u_theta"] = npf(u_r), npf(u_t)
    out["xy_x"], out["xy_y"] = npf(x), npf(y)
    out["xy_rmax"] = np.float64(r_max)
    # random_z
    torch.manual_seed(3004)
    u = torch.rand(n, dtype=F32)
    torch.manual_seed(3004)
    out["z_u"] = npf(u)
    out["z_d3"] = npf(sm.random_z(n, 0.0325, 0.1, "cpu"))
    torch.manual_seed(3004)
    out["z_d4"] = npf(sm.random_z(n, 0.05629165124598851, 0.1, "cpu"))
    # default / random orientation
    out["default_quat"] = npf(sm.default_orientation(4, "cpu"))
    torch.manual_seed(3005)
    nrm = torch.randn(n, 4, dtype=F32)
    torch.manual_seed(3005)
    out["ori_normals"] = npf(nrm)
    out["ori_quat"] = npf(sm.random_orientation(n, "cpu"))
    # yaw orientation
    torch.manual_seed(3006)
    u = torch.rand(n, dtype=F32)
    torch.manual_seed(3006)
    out["yaw_u"] = npf(u)
    out["yaw_quat"] = npf(sm.random_yaw_orientation(n, "cpu"))
    # angular velocity: randn(n,3) axis then randn(n,1) magnitude (sample.py:71-75)
    torch.manual_seed(3007)
    ax = torch.randn(n, 3, dtype=F32)
    mg = torch.randn(n, 1, dtype=F32)
    torch.manual_seed(3007)
    out["angvel_axis_normals"], out["angvel_mag_normal"] = npf(ax), npf(mg)
    out["angvel"] = npf(sm.random_angular_vel(n, "cpu", 0.5))
    np.savez(os.path.join(HERE, "samplers.npz"), **out)


def gen_torque():
    """T4: trifinger_env.py:442-494 composed from the imported unscale_transform/saturate."""
    g = torch.Generator().manual_seed(4004)
    out = {}
    q = Q_LO + (Q_HI - Q_LO) * torch.rand(N, 9, generator=g, dtype=F32)
    qd = (torch.rand(N, 9, generator=g, dtype=F32) * 2 - 1) * 8.0
    out["q"], out["qd"] = npf(q), npf(qd)
    modes = {
        "torque": (TAU_LO, TAU_HI, 9),
        "position": (Q_LO, Q_HI, 9),
        "position_impedance": (torch.cat([Q_LO, STIFF_LO]), torch.cat([Q_HI, STIFF_HI]), 18),
    }
    for mode, (lo, hi, adim) in modes.items():
        a = torch.rand(N, adim, generator=g, dtype=F32) * 2.4 - 1.2
        a[0] = 0.0
        a[1] = 1.0
        a[2] = -1.0
        out[f"{mode}_action"] = npf(a)
        for normalize in (True, False):
            at = tu.unscale_transform(a, lo, hi) if normalize else a
            if mode == "torque":
                tau = at
            elif mode == "position":
                tau = KP * (at - q)
                tau = tau - KD * qd
            else:
                tau = at[:, 9:18] * (at[:, 0:9] - q)
                tau = tau - KD * qd
            applied = tu.saturate(tau, TAU_LO, TAU_HI)
            out[f"{mode}_norm{int(normalize)}_safe0"] = npf(applied)
            applied2 = applied - KS * qd
            applied2 = tu.saturate(applied2, TAU_LO, TAU_HI)
            out[f"{mode}_norm{int(normalize)}_safe1"] = npf(applied2)
    np.savez(os.path.join(HERE, "torque.npz"), **out)


def gen_obs():
    """T5: trifinger_env.py:996-1051 (concat order) + :663-710 (scale tables) + scale_transform."""
    g = torch.Generator().manual_seed(5005)
    out = {}
    q = Q_LO + (Q_HI - Q_LO) * torch.rand(N, 9, generator=g, dtype=F32)
    qd = (torch.rand(N, 9, generator=g, dtype=F32) * 2 - 1) * 10.0
    obj = torch.zeros(N, 13, dtype=F32)
    obj[:, 0:3] = (torch.rand(N, 3, generator=g, dtype=F32) - 0.5) * 0.5
    obj[:, 3:7] = unit_quats(N, g)
    obj[:, 7:13] = (torch.rand(N, 6, generator=g, dtype=F32) - 0.5) * 1.2
    goal = torch.zeros(N, 7, dtype=F32)
    goal[:, 0:3] = (torch.rand(N, 3, generator=g, dtype=F32) - 0.5) * 0.5
    goal[:, 3:7] = unit_quats(N, g)
    tips = torch.zeros(N, 3, 13, dtype=F32)
    tips[:, :, 0:3] = (torch.rand(N, 3, 3, generator=g, dtype=F32) - 0.5) * 0.6
    tips[:, :, 3:7] = unit_quats(N * 3, g).view(N, 3, 4)
    tips[:, :, 7:13] = (torch.rand(N, 3, 6, generator=g, dtype=F32) - 0.5) * 0.6
    dof_force = (torch.rand(N, 9, generator=g, dtype=F32) - 0.5) * 0.8
    ft = (torch.rand(N, 18, generator=g, dtype=F32) - 0.5) * 3.0
    for k, v in dict(q=q, qd=qd, obj=obj, goal=goal, tips=tips, dof_force=dof_force, ft=ft).items():
        out[k] = npf(v)
    tip_lo = torch.cat([TIP_POS_LO, ORI_LO, TIP_VEL_LO])
    tip_hi = torch.cat([TIP_POS_HI, ORI_HI, TIP_VEL_HI])
    modes = {"torque": (TAU_LO, TAU_HI, 9), "position": (Q_LO, Q_HI, 9),
             "position_impedance": (torch.cat([Q_LO, STIFF_LO]), torch.cat([Q_HI, STIFF_HI]), 18)}
    for mode, (alo, ahi, adim) in modes.items():
        act = torch.rand(N, adim, generator=g, dtype=F32) * 2 - 1
        out[f"{mode}_action"] = npf(act)
        for norm_action in (True, False):
            if norm_action:
                oa_lo, oa_hi = torch.full((adim,), -1.0), torch.full((adim,), 1.0)
            else:
                oa_lo, oa_hi = alo, ahi
            obs_lo = torch.cat([Q_LO, QD_LO, OBJ_POS_LO, ORI_LO, OBJ_POS_LO, ORI_LO, oa_lo])
            obs_hi = torch.cat([Q_HI, QD_HI, OBJ_POS_HI, ORI_HI, OBJ_POS_HI, ORI_HI, oa_hi])
            st_lo = torch.cat([obs_lo, OBJ_VEL_LO, tip_lo.repeat(3), TAU_LO, WRENCH_LO.repeat(3)])
            st_hi = torch.cat([obs_hi, OBJ_VEL_HI, tip_hi.repeat(3), TAU_HI, WRENCH_HI.repeat(3)])
            raw_obs = torch.cat([q, qd, obj[:, 0:7], goal, act], dim=-1)
            raw_states = torch.cat([raw_obs, obj[:, 7:13], tips.reshape(N, 39), dof_force, ft], dim=-1)
            tag = f"{mode}_na{int(norm_action)}"
            out[f"{tag}_obs_lo"], out[f"{tag}_obs_hi"] = npf(obs_lo), npf(obs_hi)
            out[f"{tag}_states_lo"], out[f"{tag}_states_hi"] = npf(st_lo), npf(st_hi)
            out[f"{tag}_obs_raw"], out[f"{tag}_states_raw"] = npf(raw_obs), npf(raw_states)
            out[f"{tag}_obs_norm"] = npf(tu.scale_transform(raw_obs, obs_lo, obs_hi))
            out[f"{tag}_states_norm"] = npf(tu.scale_transform(raw_states, st_lo, st_hi))
    np.savez(os.path.join(HERE, "obs.npz"), **out)


def gen_termination():
    """T8: trifinger_env.py:1053-1099 composed from torch.norm + imported quat_diff_rad."""
    g = torch.Generator().manual_seed(6006)
    out = {}
    obj_p = (torch.rand(N, 3, generator=g, dtype=F32) - 0.5) * 0.2
    obj_q = unit_quats(N, g)
    goal_p = obj_p + (torch.rand(N, 3, generator=g, dtype=F32) - 0.5) * 0.03
    dq = unit_quats(N, g) * 0.12
    dq[:, 3] = 1.0
    dq = dq / dq.norm(dim=-1, keepdim=True)
    goal_q = tu.quat_mul(obj_q, dq)
    goal_p[0] = obj_p[0]; goal_q[0] = obj_q[0]
    goal_q[1] = -obj_q[1]
    reward_in = torch.rand(N, generator=g, dtype=F32) * 10 - 5
    succ_in = torch.rand(N, generator=g) > 0.5
    goal_reset_in = torch.rand(N, generator=g) > 0.7
    out.update(obj_p=npf(obj_p), obj_q=npf(obj_q), goal_p=npf(goal_p), goal_q=npf(goal_q),
               reward_in=npf(reward_in), successes_in=succ_in.numpy(), goal_reset_in=goal_reset_in.numpy())
    pos_tol, ori_tol, bonus = 0.01, 0.2, 5000.0
    out.update(pos_tol=np.float32(pos_tol), ori_tol=np.float32(ori_tol), bonus=np.float32(bonus))
    d_p = torch.norm(goal_p - obj_p, p=2, dim=-1)
    pos_ok = torch.le(d_p, pos_tol)
    d_q = tu.quat_diff_rad(obj_q, goal_q)
    ori_ok = torch.le(d_q, ori_tol)
    out["pos_count"] = np.int64(pos_ok.sum().item())
    out["ori_count"] = np.int64(ori_ok.sum().item())
    out["pos_ok"], out["ori_ok"] = pos_ok.numpy(), ori_ok.numpy()
    for diff in (1, 4, 5):
        if diff < 4:
            done = pos_ok
        elif diff == 4:
            done = torch.logical_and(pos_ok, ori_ok)
        else:
            done = ori_ok
        for act in (True, False):
            reward = reward_in.clone()
            goal_reset = goal_reset_in.clone()
            succ = succ_in.clone()
            if act:
                ids = torch.nonzero(done).squeeze()
                reward[ids] += bonus
                goal_reset = done
                succ = succ + goal_reset
            else:
                succ = torch.logical_and(goal_reset, succ)
            tag = f"d{diff}_act{int(act)}"
            out[f"{tag}_reward"] = npf(reward)
            out[f"{tag}_goal_reset"] = goal_reset.numpy()
            out[f"{tag}_successes"] = succ.numpy()
            out[f"{tag}_succ_mean"] = np.float64(np.mean(succ.cpu().numpy()))
    np.savez(os.path.join(HERE, "termination.npz"), **out)


def gen_constants():
    o = ref.tf_utils.CuboidalObject(0.065)
    d = ref.tf_utils.TrifingerDimensions
    np.savez(os.path.join(HERE, "constants.npz"),
             radius_3d=np.float64(o.radius_3d), max_com_distance=np.float64(o.max_com_distance_to_center),
             min_height=np.float64(o.min_height), max_height=np.float64(o.max_height),
             arena_radius=np.float64(ref.tf_utils.ARENA_RADIUS), size=np.array(o.size, dtype=np.float64),
             state_dim=np.int64(d.StateDim.value), wrench_dim=np.int64(d.WrenchDim.value),
             num_fingers=np.int64(d.NumFingers.value), joint_dim=np.int64(d.JointPositionDim.value),
             object_pose_dim=np.int64(d.ObjectPoseDim.value), object_vel_dim=np.int64(d.ObjectVelocityDim.value),
             # the same class for the phase-3 cuboid of the reference's assets (objects/urdf/cube_multicolor_rrc_phase3.urdf)
             **{f"phase3_{k}": np.float64(v) for k, v in (lambda c: dict(radius_3d=c.radius_3d, max_com_dist=c.max_com_distance_to_center,
                                                                          min_height=c.min_height, max_height=c.max_height))(
                 ref.tf_utils.CuboidalObject((0.02, 0.08, 0.02))).items()})


NORM_PS = (2, 1, 3, float("inf"))


def gen_finger_rewards():
    """Fingertip reward terms on FK-produced tips.  The joint positions are seeded inputs; the fingertip positions are the
    output of the native step's forward kinematics (CPU oracle here; the HIP library reproduces them bit for bit), the
    expected values are the outputs of the imported reference classes on those fingertips."""
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle_util import load_oracle
    from leibnizgym_amd import _capi as capi
    from leibnizgym_amd.engine import TrifingerEngine, make_config
    lib = load_oracle()
    rng = np.random.default_rng(77)
    n = N
    base = np.tile(np.array([0.0, 0.9, -1.7], dtype=np.float32), (n, 3))
    q_prev = base + rng.uniform(-0.3, 0.3, (n, 9)).astype(np.float32)
    q = q_prev + rng.uniform(-0.02, 0.02, (n, 9)).astype(np.float32)
    obj = np.zeros((n, 13), dtype=np.float32)
    obj[:, 0:3] = rng.uniform(-0.1, 0.1, (n, 3)); obj[:, 2] = np.abs(obj[:, 2]) + 0.0325; obj[:, 6] = 1.0
    obj_prev = obj.copy()
    obj_prev[:, 0:3] += rng.uniform(-0.002, 0.002, (n, 3)).astype(np.float32)
    off = {k: {"activate": False} for k in capi.REWARD_TERM_ORDER}
    eng = TrifingerEngine(make_config(lib, n, command_mode="torque", reward_terms=off, success={"activate": False}),
                          device="cpu", lib=lib)
    tips_of = {}
    for name, qq in (("prev", q_prev), ("now", q)):
        eng.q.copy_(torch.from_numpy(np.ascontiguousarray(qq.T)))
        eng.post_step()
        tips_of[name] = eng.tip_pos.T.numpy().reshape(n, 3, 3).copy()
    eng.close()
    tips = torch.zeros(n, 3, 13, dtype=F32); tips[:, :, 0:3] = torch.from_numpy(tips_of["now"])
    tips_prev = torch.zeros(n, 3, 13, dtype=F32); tips_prev[:, :, 0:3] = torch.from_numpy(tips_of["prev"])
    out = {"q": q, "q_prev": q_prev, "obj": obj, "obj_prev": obj_prev, "tips": tips_of["now"], "tips_prev": tips_of["prev"],
           "dt": np.float32(0.02), "norm_ps": np.array(NORM_PS, dtype=np.float64),
           "reach_weight": np.float32(-750.0), "move_weight": np.float32(-0.1)}
    o, op = torch.from_numpy(obj), torch.from_numpy(obj_prev)
    for k, p in enumerate(NORM_PS):
        term = rw.FingerReachObjectRatePenalty(activate=True, weight=-750.0, norm_p=p)
        out[f"reach_{k}"] = npf(term.compute(0.0, tips, tips_prev, o, op))
    out["move"] = npf(rw.FingertipMovementPenalty(activate=True, weight=-0.1).compute(0.02, tips, tips_prev))
    np.savez(os.path.join(HERE, "finger_rewards.npz"), **out)


if __name__ == "__main__":
    gen_math()
    gen_rewards()
    gen_samplers()
    gen_torque()
    gen_obs()
    gen_termination()
    gen_constants()
    gen_finger_rewards()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
