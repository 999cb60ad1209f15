"""A non-finite env must not poison the batch: the guard in the post-step parks it at the default pose, flags it for
reset and counts it in info[10]; every other env is untouched and the flagged env is reset (and finite) one step later.
DESIGN.md section 5, last bullet.  Checked on the oracle (CPU) and, bit for bit against it, on the GPU."""
import numpy as np
import pytest
import torch

import parity_util as pu
from leibnizgym_amd import _capi as capi
from leibnizgym_amd.engine import TrifingerEngine, make_config

N, BAD = 200, (3, 77, 199)


def run(lib, device, poison):
    kw = dict(pu.CONFIGS["d4_torque_asym"])
    eng = TrifingerEngine(make_config(lib, N, seed=4, episode_length=50, **kw), device=device, lib=lib)
    eng.reset()
    g = torch.Generator().manual_seed(8)
    snaps = []
    for t in range(6):
        act = (torch.rand(N, 9, generator=g) * 2 - 1)
        if t == 2:
            if poison == "action":
                act[list(BAD), 4] = float("nan")
            else:                                   # a corrupted state row (cube position of the three envs)
                eng.state[capi.S_CUBE_P, list(BAD)] = float("inf")
        eng.step(act.to(device))
        snaps.append(pu.snapshot(eng))
    eng.close()
    return snaps


def test_nan_action_is_absorbed_by_the_torque_clamp(oracle):
    """A NaN command never reaches the physics: the torque saturation (v_min / v_max semantics: the non-NaN operand
    wins) turns it into the limit torque.  Only the slot of the observation that reports the command shows the NaN."""
    snaps, ref = run(oracle, "cpu", "action"), run(oracle, "cpu", "nothing")
    s2 = snaps[2]
    assert s2["info"][capi.INFO_NUM_NONFINITE] == 0 and np.isfinite(s2["state"]).all()
    tau = s2["state"][capi.S_TAU + 4, list(BAD)]                     # 0.36 from the first clamp, then the safety damping
    assert np.isfinite(tau).all() and np.all(np.abs(tau) <= 0.36 + 1e-7)
    # the slot of the observation that reports the command: every emitted value passes a clamp whose bounds win over NaN
    # (+-clip_obs when the wrapper's clipping is fused, +-FLT_MAX otherwise), so the batch stays free of NaN
    assert not np.isnan(s2["obs"]).any() and np.all(s2["obs"][list(BAD), 32 + 4] < -1e38)
    ok = np.ones(N, bool)
    ok[list(BAD)] = False
    assert np.array_equal(s2["obs"][ok], ref[2]["obs"][ok]) and np.array_equal(s2["state"][:, ok], ref[2]["state"][:, ok])
    assert np.isfinite(snaps[3]["obs"]).all()                        # the next command overwrites the slot


def test_guard_isolates_and_recovers(oracle):
    snaps = run(oracle, "cpu", "state")
    ref = run(oracle, "cpu", "nothing")            # same seeds, no poison: the unaffected envs must be identical
    ok = np.ones(N, bool)
    ok[list(BAD)] = False
    s2, s3 = snaps[2], snaps[3]
    assert s2["info"][capi.INFO_NUM_NONFINITE] == len(BAD)
    assert np.all(s2["reset_buf"][list(BAD)] == 1) and np.isfinite(s2["state"][:capi.S_TIP_P]).all()
    # parked at the default pose
    assert np.allclose(s2["state"][capi.S_Q:capi.S_Q + 3, BAD[0]], [0.0, 0.9, -1.7])
    assert np.all(s2["state"][capi.S_QD:capi.S_QD + 9][:, list(BAD)] == 0)
    for k in ("obs", "states", "reward"):
        assert np.isfinite(s2[k]).all(), k                          # also the guarded envs hand out finite values
        assert np.array_equal(s2[k][ok], ref[2][k][ok]), k          # neighbours never noticed
    assert np.all(s2["reward"][list(BAD)] == 0.0) and np.isfinite(s2["info"]).all()
    assert np.array_equal(s2["state"][:, ok], ref[2]["state"][:, ok])
    # one step later the flagged envs have been reset like any timed-out env: finite everywhere, guard count back to 0
    assert s3["info"][capi.INFO_NUM_NONFINITE] == 0 and s3["info"][capi.INFO_NUM_RESETS] == len(BAD)
    for k in ("obs", "states", "reward", "state"):
        assert np.isfinite(s3[k]).all(), k
    assert np.all(s3["steps"][list(BAD)] == 1)


@pytest.mark.gpu
@pytest.mark.parametrize("poison", ["action", "state"])
def test_guard_hip_equals_oracle(hip, oracle, poison):
    got, want = run(hip, "cuda:0", poison), run(oracle, "cpu", poison)
    for t, (a, b) in enumerate(zip(got, want)):
        pu.assert_bit_equal(a, b, f"nan guard ({poison}) step {t}")
