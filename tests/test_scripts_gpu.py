"""The launch scripts run end to end on the GPU box: the reference's demo driver, the Hydra-style launcher (random-action
fallback when rl_games is absent) and a two-epoch PPO run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable] + args, cwd=REPO, env=e, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return p.stdout


def test_random_action_driver(hip):
    out = run(["scripts/trifinger_random_action.py", "120"])
    assert "Trifinger environment creation successful." in out and "env-steps/s" in out


def test_hydra_style_launcher(hip, tmp_path):
    """reference command line: train two epochs through run_rlg (run directory, config dumps, checkpoint), then
    args.play=True args.checkpoint=... from that checkpoint; rollout=N is the random-action mode"""
    import glob
    base = ["scripts/rlg_hydra.py", "gym=trifinger_difficulty_4", "args.num_envs=512", "args.headless=True", f"args.logdir={tmp_path}/logs"]
    out = run(base, env={"TF_MAX_EPOCHS": "2"})
    assert "Saving logs at" in out and len([ln for ln in out.splitlines() if ln.startswith("epoch")]) == 2
    runs = glob.glob(f"{tmp_path}/logs/*")
    assert len(runs) == 1 and all(os.path.isfile(os.path.join(runs[0], f)) for f in ("agent_config.yaml", "env_config.yaml", "nn/trifinger.pth"))
    out = run(base + ["args.play=True", f"args.checkpoint={runs[0]}/nn/trifinger.pth"], env={"TF_PLAY_STEPS": "20"})
    assert "Restoring checkpoint" in out and "play: 20 steps" in out
    out = run(base + ["rollout=40"])
    assert "env-steps/s" in out


def test_ppo_training_script(hip):
    out = run(["scripts/train_ppo.py", "gym=trifinger_difficulty_4", "args.num_envs=512", "epochs=2"])
    lines = [ln for ln in out.splitlines() if ln.startswith("epoch")]
    assert len(lines) == 2 and "frames/s" in lines[-1]
