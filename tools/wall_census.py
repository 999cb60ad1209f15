"""Developer tool (GPU box): how many boundary corners of the cube push, per env and per wavefront, under the bench workload."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import bench
from leibnizgym_amd.engine import TrifingerEngine, make_config
from leibnizgym_amd import _capi as capi
lib = capi.load_hip_library()
n = 65536
eng = TrifingerEngine(make_config(lib, n, seed=7, **bench.workload_kwargs(True)), device="cuda:0", lib=lib)
g = torch.Generator(device="cuda:0").manual_seed(7)
ring = [(torch.rand(n, 9, device="cuda:0", generator=g) * 2 - 1) for _ in range(16)]
eng.reset()
hist = np.zeros(5); waves = np.zeros(5); zs = []
for k in range(1500):
    eng.step(ring[k % 16])
    if k >= 300 and k % 50 == 0:
        lam = eng.state[capi.S_LAM_CW:capi.S_LAM_CW + 12].view(4, 3, n)[:, 0, :]      # normal impulses of the four corner slots
        face = eng.state[capi.S_CW_FACE]
        act = (lam > 0) & (face != 0)[None, :]
        cnt = act.sum(0)
        hist += np.bincount(cnt.cpu().numpy(), minlength=5)
        per_wave = act.view(4, n // 64, 64).any(2).sum(0)          # corner slots with at least one pushing lane, per wavefront
        waves += np.bincount(per_wave.cpu().numpy(), minlength=5)
        slot = act.float().mean(1).cpu().numpy()
        cz = eng.state[capi.S_CUBE_P + 2][cnt >= 3]
        zs.append(cz.cpu().numpy())
print("envs by number of pushing boundary corners (0..4):", (hist / hist.sum()).round(5))
print("wavefronts by number of corner slots with a pushing lane (0..4):", (waves / waves.sum()).round(4))
print("share of envs pushing per slot (last sample):", slot.round(5))
z = np.concatenate(zs)
print("cube centre height of envs with >= 3 pushing corners: n", len(z), "median", np.median(z) if len(z) else None, "p10/p90", np.percentile(z, [10, 90]) if len(z) else None)
