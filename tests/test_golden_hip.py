"""GPU suite: the HIP library against the fixtures generated from the reference's own functions."""
import numpy as np
import pytest
import torch

import golden_checks as gc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_math(hip):
    gc.check_math(hip, DEV)


def test_lgsk(hip):
    gc.check_lgsk(hip, DEV)


def test_samplers(hip):
    gc.check_samplers(hip, DEV)


def test_torque_law(hip):
    gc.check_torque(hip, DEV)


def test_obs_states(hip):
    gc.check_obs(hip, DEV)


def test_object_rewards(hip):
    gc.check_rewards(hip, DEV)


def test_finger_rewards(hip):
    gc.check_finger_rewards(hip, DEV)


def test_finger_reach_small_distances(hip):
    gc.check_finger_reach_small_distances(hip, DEV)


def test_termination(hip):
    gc.check_termination(hip, DEV)


def test_constants(hip):
    gc.check_constants(hip, DEV)


def test_philox_matches_oracle(hip, oracle):
    n = 4096
    rng = np.random.default_rng(0)
    env = rng.integers(0, 2**31, n, dtype=np.uint32)
    ctr = rng.integers(0, 2**31, n, dtype=np.uint32)
    out_o = np.zeros((n, 4), dtype=np.uint32)
    oracle.tf_test_philox(0x1234567890ABCDEF, env.ctypes.data, ctr.ctypes.data, 3, out_o.ctypes.data, n, None)
    e_d = torch.as_tensor(env.view(np.int32)).to(DEV)
    c_d = torch.as_tensor(ctr.view(np.int32)).to(DEV)
    o_d = torch.zeros(n, 4, dtype=torch.int32, device=DEV)
    assert hip.tf_test_philox(0x1234567890ABCDEF, e_d.data_ptr(), c_d.data_ptr(), 3, o_d.data_ptr(), n, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(o_d.cpu().numpy().view(np.uint32), out_o)
