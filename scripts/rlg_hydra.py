#!/usr/bin/env python3
"""Launcher with the reference's command line (scripts/rlg_hydra.py):

    python scripts/rlg_hydra.py gym=trifinger_difficulty_4 args.num_envs=65536 args.headless=True

Composes the same configuration (leibnizgym_amd/config.py), then hands the env to RL-Games when `rl_games`
is installed; without it, runs a random-action rollout (scripts/trifinger_random_action.py of the reference)
and prints env-steps/s."""
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from leibnizgym_amd.config import compose  # noqa: E402
from leibnizgym_amd.utils import rlg_train  # noqa: E402


def main(argv):
    cfg = compose(argv)
    args = SimpleNamespace(**cfg["args"])
    rlg_train.configure(cfg["gym"], args, args.logdir)
    if rlg_train.HAVE_RL_GAMES and args.train:
        from rl_games.torch_runner import Runner
        runner = Runner()
        runner.load(cfg["rlg"])
        runner.reset()
        runner.run(cfg["args"])
        return
    env = rlg_train.parse_vec_task(args, cfg["gym"])
    env.reset()
    n, a = env.num_envs, env.num_actions
    steps = int(os.environ.get("TF_ROLLOUT_STEPS", "500"))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        env.step(2 * torch.rand((n, a), device="cuda:0") - 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rl_games not installed: random-action rollout, {n} envs x {steps} steps: {n * steps / dt:.3e} env-steps/s")


if __name__ == "__main__":
    main(sys.argv[1:])
