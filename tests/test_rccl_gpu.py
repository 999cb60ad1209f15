"""RCCL for real on the one GPU a test box has: the multi-rank tests of tests/test_bench_multiproc_gpu.py put every rank on cuda:0 and
therefore swap the backend to gloo (TF_BENCH_SINGLE_DEVICE_TEST=1) - RCCL itself never ran in them.  Here a world of ONE rank runs the `nccl` backend
(= RCCL on ROCm) through the same code paths an 8-GPU job takes: communicator set-up with `device_id`, the device-side all-reduces."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TF_BENCH_SINGLE_DEVICE_TEST", None)
    return env


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_world_of_one_reduces_the_episode_statistics_through_rccl(hip):
    """`bench.py --gpus 1 --stats-every 4`: EpisodeStatsReducer's side-stream all-reduce on the nccl backend; in a world of one the reduced statistics
    are the rank's own (means x local envs / global envs: the same bits)."""
    env = dict(_env(), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()))
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "40", "--warmup", "3", "--envs", "8192", "--settle", "30", "--stats-every", "4",
                        "--no-cpu-baseline", "--no-fast-contact-leg"], cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    red, own = d["episode_stats_all_reduced"], d["episode_stats_rank0"]
    assert len(red) == 11 and red == own, (red, own)
    assert red[2] > 0.0                                         # object_dist mean over the 8192 envs: a positive kernel value
    assert "all-reduced every 4 steps" in d["config"]["parallelism"]


def test_trainer_world_of_one_exchanges_gradients_through_rccl(hip):
    """scripts/train_ppo.py under `torch.distributed.run --nproc-per-node 1` with the nccl backend: the weight broadcast, one gradient all-reduce per
    minibatch and one KL all-reduce per mini-epoch execute on RCCL; the run learns as a plain single-process one does (finite losses, same counts)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_port()),
           "scripts/train_ppo.py", "gym=trifinger_difficulty_4", "args.num_envs=1024", "epochs=2"]
    p = subprocess.run(cmd, cwd=REPO, env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("epoch")]
    assert len(lines) == 2 and all("nan" not in ln.lower() for ln in lines), p.stdout[-800:]
    coll = [ln for ln in p.stdout.splitlines() if ln.startswith("collectives:")]
    assert len(coll) == 1, p.stdout[-800:]
    # horizon 32 x 1024 envs, minibatch = 1024 envs x ... : asymm.yaml gives 32 minibatches x 4 mini-epochs per epoch
    assert "backend nccl" in coll[0] and "world 1" in coll[0]
    n_grad = int(coll[0].split("gradient all-reduces ")[1].split(",")[0])
    n_kl = int(coll[0].split("KL all-reduces ")[1])
    assert n_grad > 0 and n_grad % n_kl == 0 and n_kl == 2 * 4, coll[0]
