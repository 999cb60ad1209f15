#!/usr/bin/env python3
"""Random-action driver of the TriFinger env - the role of the reference's scripts/trifinger_random_action.py: build the
env from a plain config dict (8192 instances, torque mode, decimation 1), reset, then step with actions 2 U[0,1) - 1.

    python scripts/trifinger_random_action.py [steps]      # the reference loops forever; here `steps` bounds the run
                                                           # (default 2000) and the rate is printed every 500 steps
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from leibnizgym_amd.envs import TrifingerEnv  # noqa: E402
from leibnizgym_amd.utils.helpers import print_info  # noqa: E402


def main(steps):
    # the reference's dict; the `sim` block selects PhysX pipelines there and has no counterpart here (it is accepted
    # and ignored by the config merge, like every other simulator-only key)
    env_config = {
        "num_instances": 8192,
        "aggregrate_mode": True,
        "control_decimation": 1,
        "command_mode": "torque",
        "sim": {"use_gpu_pipeline": True, "physx": {"use_gpu": False}},
    }
    env = TrifingerEnv(config=env_config, device="cuda:0", verbose=True, visualize=False)
    env.reset()
    print_info("Trifinger environment creation successful.")
    n = env.get_num_instances()
    t0, last = time.perf_counter(), 0
    for k in range(1, steps + 1):
        action = 2 * torch.rand(env.get_action_shape(), dtype=torch.float, device=env.device) - 1
        env.step(action)
        env.render()
        if k % 500 == 0 or k == steps:
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print_info(f"step {k}: {n * (k - last) / dt:.3e} env-steps/s")
            t0, last = time.perf_counter(), k
    env.close()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2000)
