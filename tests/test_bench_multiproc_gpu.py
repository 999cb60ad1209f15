"""bench.py's multi-process path (the driver launches it with torch.distributed.run, one rank per GPU) exercised on a box
with ONE GPU: two ranks share cuda:0 and the timing collectives go through gloo (TF_BENCH_SINGLE_DEVICE_TEST=1).  Checks
what the driver relies on: exactly one JSON line, from rank 0, whole-job aggregate, the contract's keys."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(script_args, nproc=2):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, TF_BENCH_SINGLE_DEVICE_TEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port)] + script_args
    p = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    return p


def test_single_process_line_carries_the_contract(hip):
    """`python bench.py` (N = 1): one JSON line with the contract's keys, `value` = envs x steps / elapsed of the step with on-device action
    generation, `roofline` (bound / achieved / peak / unit / frac / traffic, measured kernel time) and what the steady-state prelude did."""
    p = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "3", "--envs", "8192", "--settle", "30", "--no-cpu-baseline"],
                       cwd=REPO, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-1500:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "value_resident_actions", "value_with_torch_action_generation", "steady_state_prelude"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 3 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["strong_scaling"]["value"] == d["value"] and d["roofline"]["kernel_variant"] == "wide_helpers"      # one rank: no second leg; 8192 envs: the 256-register kernel with helper wavefronts
    assert abs(d["value"] - 8192 * 40 / (d["ms_per_step"] * 1e-3 * 40)) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_avg_us", "kernel_launches_timed"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel_launches_timed"] == 40
    assert "127" in r["kernel"]                                            # the instantiation with the action source fused in is what `value` launches
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["kernel_avg_us"] < d["ms_per_step"] * 1e3 * 1.05
    assert "30 steps" in d["steady_state_prelude"] and "workload" in d["config"]
    q = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "2", "--envs", "4096", "--settle", "0", "--no-cpu-baseline"],
                       cwd=REPO, capture_output=True, text=True, timeout=600)
    assert q.returncode == 0 and json.loads([ln for ln in q.stdout.splitlines() if ln.startswith("{")][0])["steady_state_prelude"].startswith("none")


def test_two_ranks_print_one_aggregate_line(hip):
    p = launch(["bench.py", "--gpus", "2", "--steps", "100", "--warmup", "5", "--envs", "8192"])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 100 and d["warmup"] == 5 and d["scaling"] == "weak"
    assert d["config"]["envs_per_gpu"] == 8192 and d["config"]["global_envs"] == 16384
    # the strong-scaling leg beside it: --envs (8192) in total, 4096 per rank, the 256-register instantiation of the step kernel (with helper wavefronts)
    s = d["strong_scaling"]
    assert s["global_envs"] == 8192 and s["envs_per_gpu"] == 4096 and s["kernel_variant"] == "wide_helpers" and s["kernel"].endswith("true, true>")
    assert abs(s["value"] - 8192 * 100 / (s["ms_per_step"] * 1e-3 * 100)) < 1e-6 * s["value"]
    assert d["value_strong_65536_total"] is None               # only the headline total carries that key
    assert abs(d["value"] - 16384 * 100 / (d["ms_per_step"] * 1e-3 * 100)) < 1e-6 * d["value"]     # whole-job aggregate
    assert "cpu_baseline" not in d and d["roofline"]["kernel_launches_timed"] > 0
    for k in ("metric", "unit", "higher_is_better", "vs_baseline", "dtype", "data", "roofline"):
        assert k in d


def test_eight_ranks_on_one_device(hip):
    """The driver's 8-GPU command line (`--nproc-per-node 8 ... bench.py --gpus 8`) with all eight ranks on cuda:0 and small shards:
    rank offsets 0, 2048, ..., one aggregate JSON line from rank 0 - so that the first real 8-GPU run cannot fail on plumbing."""
    p = launch(["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5", "--envs", "2048", "--settle", "50", "--strong-total", "65536"], nproc=8)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 20 and d["scaling"] == "weak"
    assert d["config"]["envs_per_gpu"] == 2048 and d["config"]["global_envs"] == 16384
    # the strong-scaling reading of the metric at the driver's 8-GPU command: the headline's 65536 envs partitioned into 8 x 8192
    s = d["strong_scaling"]
    assert s["global_envs"] == 65536 and s["envs_per_gpu"] == 8192 and s["kernel_variant"] == "wide_helpers"
    assert d["value_strong_65536_total"] == s["value"] and abs(s["value"] - 65536 * 20 / (s["ms_per_step"] * 1e-3 * 20)) < 1e-6 * s["value"]
    assert abs(d["value"] - 16384 * 20 / (d["ms_per_step"] * 1e-3 * 20)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d


def test_two_ranks_domain_randomisation_with_the_stats_all_reduce(hip):
    """BASELINE configs[3] per GPU (difficulty 4 + every DR feature, 16384 envs) with the optional episode-statistics
    all-reduce every 4 steps (EpisodeStatsReducer on its side stream): the exchange the RCCL path runs on a real node."""
    p = launch(["bench.py", "--gpus", "2", "--steps", "60", "--warmup", "5", "--envs", "16384", "--dr", "--stats-every", "4"])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["envs_per_gpu"] == 16384 and d["config"]["global_envs"] == 32768
    st = d["episode_stats_all_reduced"]
    assert len(st) == 11 and all(x == x for x in st)
    assert st[2] > 0.0                                         # object_dist mean over ALL 32768 envs: a positive kernel value
    assert 0 <= st[6] <= 32768 and 0 <= st[7] <= 32768         # goal counts are sums over both shards


def test_two_rank_ppo_training(hip):
    """BASELINE configs[4] shape (PPO, env shards, one fused gradient all-reduce per minibatch) with two ranks on one device:
    both ranks train, rank 0 logs, the job ends cleanly."""
    p = launch(["scripts/train_ppo.py", "gym=trifinger_difficulty_4", "args.num_envs=1024", "epochs=2"])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("epoch")]
    assert len(lines) == 2 and "frames/s" in lines[-1]
    frames = int(lines[-1].split("frames")[1].split()[0])
    assert frames == 2 * 32 * 1024 * 2                         # epochs x horizon x envs per rank x ranks
