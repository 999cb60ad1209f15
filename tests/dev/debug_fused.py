import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import parity_util as pu
from oracle_util import load_oracle
from leibnizgym_amd import _capi
from leibnizgym_amd.engine import TrifingerEngine, make_config
np.set_printoptions(precision=9, suppress=False, linewidth=220)
cfgname = sys.argv[1] if len(sys.argv) > 1 else "d1_torque_sym"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
hip, orc = _capi.load_hip_library(), load_oracle()
kw = dict(pu.CONFIGS[cfgname])
eh = TrifingerEngine(make_config(hip, n, seed=3, episode_length=40, **kw), device="cuda:0", lib=hip)
eo = TrifingerEngine(make_config(orc, n, seed=3, episode_length=40, **kw), device="cpu", lib=orc)
eh.reset(); eo.reset()
for t in range(3):
    pre = eo.state.numpy().copy()
    act = pu.actions_for(t, n, eh.action_dim, 3)
    eh.step(act.to("cuda:0")); eo.step(act)
    torch.cuda.synchronize()
    a, b = eh.state.cpu().numpy(), eo.state.numpy()
    same = (a.view(np.uint32) == b.view(np.uint32))[:59]
    bad = np.argwhere(~same.all(axis=0)).ravel()
    print(f"step {t}: envs differing {bad[:30].tolist()} ({len(bad)}); hip info {eh.info[:11].cpu().numpy()}")
    for e in bad[:3]:
        print(f"  env {e}: pre-state q {pre[0:9, e]} qd {pre[9:18, e]}\n     cube {pre[18:31, e]}\n     act {act[e].numpy()}")
        print(f"     hip post q {a[0:9, e]} cube {a[18:25, e]}\n     orc post q {b[0:9, e]} cube {b[18:25, e]}")
        rows = np.argwhere(~same[:, e]).ravel().tolist()
        print(f"     rows differing {rows}")
    if len(bad): break
