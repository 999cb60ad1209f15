"""Developer tool (GPU box, timing build): where the hardware put the four role wavefronts of every workgroup."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
from collections import Counter
import bench
from leibnizgym_amd.engine import TrifingerEngine, make_config
from leibnizgym_amd import _capi
lib = _capi.TfLib(os.path.join(REPO, "leibnizgym_amd", "csrc", "libtrifinger_hip_timing.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
eng = TrifingerEngine(make_config(lib, n, seed=7, **bench.workload_kwargs(True)), device="cuda:0", lib=lib)
eng.reset()
a = torch.zeros(n, 9, device="cuda:0")
for _ in range(3):
    eng.step(a)
torch.cuda.synchronize()
st = eng.scratch.view(torch.int32).view(-1, 4, 64).cpu().numpy().astype(np.int64) & 0xffffffff
hw, xcc = st[:, :, 40], st[:, :, 41] & 0xf
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; wave = hw & 0xf
print("first workgroups: (role: simd cu sh se xcc wave_slot)")
for b in range(12):
    print(b, [(int(simd[b, r]), int(cu[b, r]), int(sh[b, r]), int(se[b, r]), int(xcc[b, r]), int(wave[b, r])) for r in range(4)])
same = sum(len(set(simd[b])) == 4 for b in range(st.shape[0]))
print("workgroups whose 4 waves sit on 4 distinct SIMDs:", same, "of", st.shape[0])
print("SIMD of the cube role:", Counter(simd[:, 3].tolist()))
key = [(int(xcc[b, 3]), int(se[b, 3]), int(sh[b, 3]), int(cu[b, 3])) for b in range(st.shape[0])]
per_cu = {}
for b, k in enumerate(key):
    per_cu.setdefault(k, []).append((b, int(simd[b, 3])))
print("CUs used:", len(per_cu))
hist = Counter(tuple(sorted(Counter(s for _, s in v).values(), reverse=True)) for v in per_cu.values())
print("per CU: multiset of how many cube waves share a SIMD:", hist)
print("example CUs:", list(per_cu.items())[:4])
