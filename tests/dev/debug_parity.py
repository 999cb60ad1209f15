"""Debug helper (GPU box): first step at which the fused HIP rollout leaves the oracle, which state rows differ and how."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import parity_util as pu
from oracle_util import load_oracle
from leibnizgym_amd import _capi
cfg = sys.argv[1] if len(sys.argv) > 1 else "d1_torque_sym"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
got = pu.rollout(_capi.load_hip_library(), "cuda:0", n, steps, cfg)
want = pu.rollout(load_oracle(), "cpu", n, steps, cfg)
for t, (a, b) in enumerate(zip(got, want)):
    x, y = a["state"], b["state"]
    same = (x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))
    if same.all():
        continue
    rows = np.argwhere((~same).any(axis=1)).ravel()
    print("step", t, "differing rows", rows.tolist())
    for r in rows[:6]:
        e = np.argwhere(~same[r]).ravel()
        print("  row", r, "n", len(e), "env", e[0], "hip", repr(x[r, e[0]]), "orc", repr(y[r, e[0]]))
        env = e[0]
    print("  env", env, "rows 90..105 hip", x[90:105, env], "\n                    orc", y[90:105, env])
    break
else:
    print("no difference in state over", steps, "steps")
