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


def test_two_ranks_print_one_aggregate_line(hip):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, TF_BENCH_SINGLE_DEVICE_TEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "100", "--warmup", "5", "--envs", "8192"]
    p = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 100 and d["warmup"] == 5 and d["scaling"] == "weak"
    assert d["config"]["envs_per_gpu"] == 8192 and d["config"]["global_envs"] == 16384
    assert abs(d["value"] - 16384 * 100 / (d["ms_per_step"] * 1e-3 * 100)) < 1e-6 * d["value"]     # whole-job aggregate
    assert "cpu_baseline" not in d and d["roofline"]["kernel_launches_timed"] > 0
    for k in ("metric", "unit", "higher_is_better", "vs_baseline", "dtype", "data", "roofline"):
        assert k in d
