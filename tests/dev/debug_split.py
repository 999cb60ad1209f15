"""Debug helper (GPU box): walk the split path hook by hook on both libraries and report the first divergence."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import parity_util as pu
from oracle_util import load_oracle
from leibnizgym_amd import _capi
from leibnizgym_amd.engine import TrifingerEngine, make_config

cfgname = sys.argv[1] if len(sys.argv) > 1 else "d1_torque_sym"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
hip, orc = _capi.load_hip_library(), load_oracle()
kw = dict(pu.CONFIGS[cfgname])
eh = TrifingerEngine(make_config(hip, n, seed=3, episode_length=40, **kw), device="cuda:0", lib=hip)
eo = TrifingerEngine(make_config(orc, n, seed=3, episode_length=40, **kw), device="cpu", lib=orc)
eh.reset(); eo.reset()

def cmp(tag):
    torch.cuda.synchronize()
    a, b = eh.state.cpu().numpy(), eo.state.numpy()
    same = a.view(np.uint32) == b.view(np.uint32)
    rows = np.argwhere((~same).any(axis=1)).ravel().tolist()
    nan_h = np.argwhere(~np.isfinite(a).all(axis=0)).ravel()
    print(f"{tag}: differing rows {rows[:20]}{'...' if len(rows) > 20 else ''}; non-finite envs on hip: {nan_h[:8].tolist()} ({len(nan_h)})")
    if rows:
        r = rows[0]; e = np.argwhere(~same[r]).ravel()[0]
        print(f"    first row {r} env {e}: hip {a[r, e]!r} orc {b[r, e]!r}")
        return int(e)
    return None

cmp("after reset")
for t in range(steps):
    act = pu.actions_for(t, n, eh.action_dim, 3)
    eh.action_buf.copy_(act.to("cuda:0")); eo.action_buf.copy_(act)
    eh.apply_resets(); eo.apply_resets(); cmp(f"step {t} apply_resets")
    eh.pre_step(); eo.pre_step(); cmp(f"step {t} pre_step")
    for s in range(eh.cfg.control_decimation):
        eh.simulate(); eo.simulate()
        e = cmp(f"step {t} simulate {s}")
        if e is not None:
            print("    hip state col:", eh.state[:59, e].cpu().numpy())
            print("    orc state col:", eo.state[:59, e].numpy())
            sys.exit(0)
    eh.post_step(); eo.post_step(); cmp(f"step {t} post_step")
    eh.finish_step(); eo.finish_step()
