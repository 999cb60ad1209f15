"""Fit  t(k_step) = fixed + substeps * (sub_fixed + iters * per_iter)  from a few (substeps, iterations) timings."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import bench
from leibnizgym_amd.engine import TrifingerEngine, make_config
from leibnizgym_amd import _capi
lib = _capi.load_hip_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rows = []
for asym in (True, False):
    A, y = [], []
    for sub, it in ((1, 1), (1, 8), (2, 1), (2, 4), (2, 8), (4, 8), (4, 2)):
        kw = bench.workload_kwargs(asym); kw.update(substeps=sub, solver_iterations=it)
        eng = TrifingerEngine(make_config(lib, n, seed=7, **kw), device="cuda:0", lib=lib)
        g = torch.Generator(device="cuda:0").manual_seed(1)
        ring = [(torch.rand(n, 9, device="cuda:0", generator=g) * 2 - 1) for _ in range(8)]
        eng.reset()
        for k in range(10): eng.step(ring[k % 8])
        eng.enable_kernel_timing(300)
        for k in range(300): eng.step(ring[k % 8])
        ms, cnt = eng.kernel_time_ms()
        t = ms / cnt * 1e3
        print(f"asym={asym} substeps={sub} iters={it}: k_step {t:.1f} us", flush=True)
        A.append([1, sub, sub * it]); y.append(t)
        eng.close()
    c, *_ = np.linalg.lstsq(np.array(A, float), np.array(y), rcond=None)
    print(f"asym={asym}: fixed {c[0]:.1f} us + substeps x ({c[1]:.1f} us + iters x {c[2]:.2f} us); residual {np.abs(np.array(A)@c-np.array(y)).max():.2f} us")
