"""Developer tool (GPU box): per-phase cycle counts of k_step from s_memtime stamps (build: make -C leibnizgym_amd/csrc
libtrifinger_hip_timing.so).  Phases: A = loads/resets/torque, B = physics substeps, C = outputs/rewards/stats."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import bench
from leibnizgym_amd.engine import TrifingerEngine, make_config
from leibnizgym_amd import _capi
lib = _capi.TfLib(os.path.join(REPO, "leibnizgym_amd", "csrc", "libtrifinger_hip_timing.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
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
            st = eng.scratch.view(torch.int32).view(-1, 16)[:, 11:15].cpu().numpy().astype(np.int64) & 0xffffffff
            d = np.diff(st, axis=1) & 0xffffffff
            acc.append(d)
    d = np.stack(acc).astype(np.float64)           # [steps, waves, 3]
    med = np.median(d, axis=(0, 1))
    print(f"asym={asym} N={n}: median cycles per wave  A {med[0]:.0f}  B {med[1]:.0f}  C {med[2]:.0f}  (100 MHz s_memtime ticks x?) "
          f"ratio A:B:C = {med[0]/med.sum():.2f}:{med[1]/med.sum():.2f}:{med[2]/med.sum():.2f}; start skew over waves "
          f"{np.ptp(st[:, 0] & 0xffffffff):.0f}")
    eng.close()
