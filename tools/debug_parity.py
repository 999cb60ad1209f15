"""Debug helper (GPU box): per-field / per-row mismatch census between the HIP path and the oracle."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import parity_util as pu
from oracle_util import load_oracle
from leibnizgym_amd import _capi

cfg = sys.argv[1] if len(sys.argv) > 1 else "impedance_random_moving"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
hip, orc = _capi.load_hip_library(), load_oracle()
got = pu.rollout(hip, "cuda:0", n, steps, cfg)
want = pu.rollout(orc, "cpu", n, steps, cfg)
for t, (a, b) in enumerate(zip(got, want)):
    for k in pu.PER_ENV_FIELDS:
        x, y = a[k], b[k]
        if x.dtype.kind == "f":
            same = x.view(np.uint32) == y.view(np.uint32)
        else:
            same = x == y
        if not same.all():
            if k == "state":
                rows = np.argwhere((~same).any(axis=1)).ravel()
                print(f"step {t} state rows differing: {rows.tolist()}")
                r = rows[0]; e = np.argwhere(~same[r]).ravel()[0]
                print(f"   first: row {r} env {e} got {x[r, e]!r} want {y[r, e]!r}; envs differing in that row: {int((~same[r]).sum())}")
            else:
                idx = np.argwhere(~same)
                print(f"step {t} {k}: {len(idx)} differ; first {tuple(idx[0])} got {x[tuple(idx[0])]!r} want {y[tuple(idx[0])]!r}")
    print(f"step {t} info hip {a['info'][:11]}\n          orc {b['info'][:11]}")
