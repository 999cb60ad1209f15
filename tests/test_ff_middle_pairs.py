"""TfModel.ff_middle_pairs (API 7; the default model has it ON since API 8: VERDICT r5 item 2): the middle link of a finger against the distal capsule of another finger, beyond the three
distal pairs.  The reference keeps every robot link in one collision group with self-collision on (trifinger_env.py:811-812).

* scenario (oracle here, HIP under -m gpu): finger 2 is told to put its fingertip INSIDE the middle link of finger 0.  The gap between the two
  bodies - closest points of the middle link's axis and the fingertip capsule, the support function of the middle link's cross-section, the capsule
  radius: all from the independent fp64 model (tests/physics_ref.py, tests/model_fixture.py) - stays at zero with the pairs on and goes 4 cm
  negative with them off (`native.ff_middle_pairs: false`, the opt-out).
* the parity rollout of this switch (tests/parity_util.py CONFIGS["ff_middle_pairs"], compared bit for bit in test_parity_hip_vs_oracle.py) is one
  in which the switch matters: the same rollout without it ends elsewhere."""
import numpy as np
import pytest
import torch

import model_fixture as MF
import parity_util as pu
import physics_ref as PR
import test_contact_scenarios as S

TIP = PR.TIP_CAP[2]
SH2 = next(e[2] for e in PR.SHAPES if e[0] == "shape" and e[1] == 2)


def middle_distal_gap(fm, qm, fd, qd):
    """fp64: middle link of finger fm (its finger-cube shape) against the fingertip capsule of finger fd"""
    a, b = PR.link_point_world(fm, qm, 2, SH2["a"]), PR.link_point_world(fm, qm, 2, SH2["b"])
    A, B = PR.link_point_world(fd, qd, 3, PR.TIP_CAP[1]), PR.link_point_world(fd, qd, 3, TIP)
    Pm, Pd = PR.segment_segment(a, b, A, B)
    D = np.linalg.norm(Pd - Pm)
    s = float(np.clip((Pm - a) @ (b - a) / ((b - a) @ (b - a)), 0.0, 1.0))
    u = PR.link_rotation_world(fm, qm, 2).T @ ((Pd - Pm) / D)
    return D - MF.shape_extent(SH2, s, u) - PR.TIP_CAP[3]


Q0 = np.array([0.51715904, 0.42356514, -2.58937149])      # finger 0 leans over the arena: finger 2's fingertip reaches the middle of its middle link


def _run(lib, device, on, steps=120):
    def edit(m):
        m.ff_middle_pairs = 1 if on else 0
    eng = S.torque_engine(lib, device, edit)
    target = PR.link_point_world(0, Q0, 2, np.array([0.028, 0.0, -0.08]))      # centre of mass of finger 0's middle link: inside its body
    away = PR.link_point_world(2, np.array([0.0, 0.9, -1.7]), 3, TIP) - target
    away /= np.linalg.norm(away)
    q2 = None
    for d_ in np.arange(0.02, 0.2, 0.005):          # start: the first pose on the way back to finger 2's rest pose that is 2 cm clear of the middle link
        try:
            qq = S.ik(2, target + d_ * away, q0=(0.175, 0.26, -2.38))
        except AssertionError:
            continue
        if middle_distal_gap(0, Q0, 2, qq) > 0.02:
            q2 = qq
            break
    assert q2 is not None
    f32 = dict(dtype=torch.float32, device=device)
    eng.cube[0:3, 0] = torch.tensor([0.0, 0.0, 5.0], **f32)          # cube out of the way
    eng.q[0:3, 0] = torch.tensor(Q0, **f32)
    eng.q[6:9, 0] = torch.tensor(q2, **f32)
    gaps = []
    for _ in range(steps):
        st = S.state_np(eng)
        tau = S.impedance_torques(st, [None, None, target], kp=40.0, kd=2.0)     # finger 2: fingertip to the target; everyone: gravity compensation
        tau[0:3] += 8.0 * (Q0 - st[0:3]) - 0.3 * st[9:12]                         # finger 0 holds its pose
        S.step_torque(eng, tau)
        st = S.state_np(eng)
        gaps.append(middle_distal_gap(0, st[0:3], 2, st[6:9]))
    assert np.isfinite(st).all()
    eng.close()
    return np.array(gaps)


def _check(lib, device):
    g = _run(lib, device, True)
    assert g[0] > 0.005 and g.min() > -4e-3, (g[0], g.min())        # approaches from outside, never deeper than a transient of a few mm
    assert np.abs(np.median(g[-60:])) < 1e-3, np.median(g[-60:])      # ... and stays pressed against the middle link: the gap is zero
    ghost = _run(lib, device, False)
    assert ghost.min() < -0.03, ghost.min()                            # the opt-out (distal pairs only): the fingertip goes 4 cm into the body


def test_fingertip_stops_at_the_middle_link_of_another_finger(oracle):
    _check(oracle, "cpu")


@pytest.mark.gpu
def test_fingertip_stops_at_the_middle_link_of_another_finger_gpu(hip):
    _check(hip, "cuda:0")


def test_the_parity_rollout_of_the_switch_is_one_in_which_it_matters(oracle):
    n, steps = 256, 60
    on = pu.rollout(oracle, "cpu", n, steps, "ff_middle_pairs")
    off = pu.rollout(oracle, "cpu", n, steps, "ff_middle_pairs", extra={"_model_edit": dict(ff_middle_pairs=0)})
    changed = np.any(on[-1]["state"][0:18] != off[-1]["state"][0:18], axis=0)
    assert 0.02 < changed.mean() < 0.9, changed.mean()      # a share of the envs saw such a contact (the rest are bit-identical: the rows are only added)


def test_the_default_model_holds_the_reference_contact_set(oracle):
    """API 8: tf_default_model() has the middle-distal pairs on (the reference keeps every robot link in one self-colliding group, trifinger_env.py:811-812)"""
    assert oracle.default_model().ff_middle_pairs == 1
