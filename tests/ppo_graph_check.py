"""Helper of tests/test_ppo.py (run as a script in a FRESH process): an eager trainer first, then a graph-replaying one, at the
BASELINE configs[4] shape - the order in which HIP graphs with memset / memcpy nodes corrupted the update."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from leibnizgym_amd.config import compose  # noqa: E402
from leibnizgym_amd.envs import TrifingerEnv  # noqa: E402
from leibnizgym_amd.ppo import PPOConfig, PPOTrainer  # noqa: E402
from leibnizgym_amd.utils.rlg_train import RlGamesGpuEnvAdapter  # noqa: E402
from leibnizgym_amd.wrappers import VecTaskPython  # noqa: E402


def run(use_graphs, epochs):
    cfg = compose(["gym=trifinger_difficulty_4", "args.num_envs=8192", "args.headless=True"])
    env = TrifingerEnv(config=cfg["gym"], device="cuda:0", verbose=False)
    ad = RlGamesGpuEnvAdapter("rlgpu", 8192, env=VecTaskPython(env, rl_device="cuda:0"))
    tr = PPOTrainer(ad, 41, 113, 9, PPOConfig.from_rlg(cfg["rlg"], num_envs=8192, use_graphs=use_graphs), device="cuda:0")
    st = tr.train(epochs)
    env.close()
    return st


if __name__ == "__main__":
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    eager, graph = run(False, epochs), run(True, epochs)
    for a, b in zip(eager, graph):
        assert a["lr"] == b["lr"], (a["lr"], b["lr"])
        assert abs(a["kl"] - b["kl"]) < 0.1 * a["kl"] and abs(a["c_loss"] - b["c_loss"]) < 0.05 * a["c_loss"] + 1e-4, (a, b)
    assert 0.004 < graph[-1]["kl"] < 0.02
    print("graph == eager over", epochs, "epochs")
