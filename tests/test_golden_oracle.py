"""CPU suite: pins the ORACLE (oracle/tf_oracle.c) against fixtures generated from the reference's own
functions (tests/golden/make_golden.py).  The GPU suite runs the same checks on the HIP library."""
import golden_checks as gc


def test_math(oracle):
    gc.check_math(oracle, "cpu")


def test_lgsk(oracle):
    gc.check_lgsk(oracle, "cpu")


def test_samplers(oracle):
    gc.check_samplers(oracle, "cpu")


def test_torque_law(oracle):
    gc.check_torque(oracle, "cpu")


def test_obs_states(oracle):
    gc.check_obs(oracle, "cpu")


def test_object_rewards(oracle):
    gc.check_rewards(oracle, "cpu")


def test_finger_rewards(oracle):
    gc.check_finger_rewards(oracle, "cpu")


def test_finger_reach_small_distances(oracle):
    gc.check_finger_reach_small_distances(oracle, "cpu")


def test_termination(oracle):
    gc.check_termination(oracle, "cpu")


def test_constants(oracle):
    gc.check_constants(oracle, "cpu")
