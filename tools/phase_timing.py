"""Developer tool (GPU box): per-phase cycle counts of k_step from s_memtime stamps (build: make -C leibnizgym_amd/csrc
libtrifinger_hip_timing.so; lane 0 of every wave appends a stamp at each PHASE_STAMP site)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import bench
from leibnizgym_amd.engine import TrifingerEngine, make_config
from leibnizgym_amd import _capi

SUB = ["free motion (FK, dynamics, M^-1)", "finger contacts + rows", "cube-floor corners", "cube-wall corners",
       "limit rows", "PGS sweeps", "wrench + integrate"]
LABELS = (["issue all loads", "wait for loads, action via LDS", "apply_resets", "action_buf store, torque, park"]
          + [f"sub0: {x}" for x in SUB] + [f"sub1: {x}" for x in SUB]
          + ["(stamp 2)", "unpark, tip FK, NaN guard", "tip history, rewards, termination, statistics atomic",
             "obs emit + store", "states emit + store", "(stamp)", "state stores, finish", "statistics ticket check / fold"])

lib = _capi.TfLib(os.path.join(REPO, "leibnizgym_amd", "csrc", "libtrifinger_hip_timing.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
stride = int(lib.tf_scratch_floats(64))
for asym in (True, False):
    eng = TrifingerEngine(make_config(lib, n, seed=7, **bench.workload_kwargs(asym)), device="cuda:0", lib=lib)
    g = torch.Generator(device="cuda:0").manual_seed(1)
    ring = [(torch.rand(n, 9, device="cuda:0", generator=g) * 2 - 1) for _ in range(8)]
    eng.reset()
    acc = []
    for k in range(60):
        eng.step(ring[k % 8])
        if k >= 10:
            torch.cuda.synchronize()
            st = eng.scratch.view(torch.int32).view(-1, stride)[:, 16:16 + len(LABELS) + 1].cpu().numpy().astype(np.int64) & 0xffffffff
            acc.append(np.diff(st, axis=1) & 0xffffffff)
    d = np.stack(acc).astype(np.float64)           # [steps, waves, phases]
    med = np.median(d, axis=(0, 1))
    tot = med.sum()
    print(f"asym={asym} N={n}: median s_memtime ticks per wave, total {tot:.0f}")
    for lab, m in zip(LABELS, med):
        print(f"  {lab:40s} {m:9.0f}  {100 * m / tot:5.1f} %")
    eng.close()
