#!/usr/bin/env python3
"""Launcher with the reference's command line (scripts/rlg_hydra.py):

    python scripts/rlg_hydra.py gym=trifinger_difficulty_4 args.num_envs=65536 args.headless=True
    python scripts/rlg_hydra.py gym=trifinger_difficulty_4 args.num_envs=8192 args.play=True args.checkpoint=<run>/nn/trifinger.pth

Composes the same configuration (leibnizgym_amd/config.py) and hands it to `run_rlg_hydra`
(leibnizgym_amd/utils/rlg_train.py): time-stamped run directory with `agent_config.yaml` / `env_config.yaml`, the env-info
observer, then RL-Games' `Runner` when `rl_games` is installed and the in-repo PPO runner otherwise.  `rollout=N` runs a
random-action rollout instead (scripts/trifinger_random_action.py of the reference) and prints env-steps/s."""
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from leibnizgym_amd.config import compose  # noqa: E402
from leibnizgym_amd.utils import rlg_train  # noqa: E402


def main(argv):
    rollout = None
    rest = []
    for a in argv:
        if a.startswith("rollout="):
            rollout = int(a.split("=", 1)[1])
        else:
            rest.append(a)
    cfg = compose(rest)
    if rollout is None:
        return rlg_train.run_rlg_hydra(cfg)
    args = SimpleNamespace(**cfg["args"])
    rlg_train.configure(cfg["gym"], args, None, cfg["rlg"])
    env = rlg_train.parse_vec_task(args, cfg["gym"])
    env.reset()
    n, a = env.num_envs, env.num_actions
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rollout):
        env.step(2 * torch.rand((n, a), device="cuda:0") - 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"random-action rollout, {n} envs x {rollout} steps: {n * rollout / dt:.3e} env-steps/s")


if __name__ == "__main__":
    main(sys.argv[1:])
