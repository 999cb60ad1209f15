"""Bit-exact parity over LONG episodes (GPU box): the HIP step and the CPU oracle side by side, step by step, compared every K steps and at the
end.  With 750-step episodes under random actions the rollout reaches the steady state of the bench workload - about 5 % of the cubes rest
against the boundary - which the 40..60-step episodes of the test-suite and of parity_stress.py never do (boundary-corner slot order, per-slot
flags, the branch-free boundary block of the sweeps).
    python tests/dev/parity_stress_long.py [envs] [steps] [compare_every]"""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import parity_util as pu
from oracle_util import load_oracle
from leibnizgym_amd import _capi
from leibnizgym_amd.engine import TrifingerEngine, make_config

hip, orc = _capi.load_hip_library(), load_oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
every = int(sys.argv[3]) if len(sys.argv) > 3 else 50
PHASE3, DENSITY = (0.02, 0.08, 0.02), 500.0
RUNS = [("d4_torque_asym", None), ("d1_torque_sym", None), ("d4_domain_randomization", None), ("d4_domain_randomization_extended", None),
        ("d4_torque_asym", "box"), ("d4_domain_randomization", "box")]
for cfg_name, obj in RUNS:
    engs = []
    for lib, dev in ((hip, "cuda:0"), (orc, "cpu")):
        kw = dict(pu.CONFIGS[cfg_name])
        kw.pop("_clipping", None)
        if obj == "box":
            kw["model"] = lib.box_model(PHASE3, DENSITY)
        engs.append(TrifingerEngine(make_config(lib, n, seed=21, episode_length=750, **kw), device=dev, lib=lib))
    for e in engs:
        e.reset()
    wall = 0.0
    for t in range(steps):
        act = pu.actions_for(t, n, engs[0].action_dim, 21)
        engs[0].step(act.to("cuda:0"))
        engs[1].step(act)
        if t % every == every - 1 or t == steps - 1:
            a, b = pu.snapshot(engs[0]), pu.snapshot(engs[1])
            pu.assert_bit_equal(a, b, f"{cfg_name} {obj or 'cube'} step {t}")
            wall = max(wall, float((b["state"][_capi.S_CW_FACE] != 0).mean()))
    for e in engs:
        e.close()
    print(f"{cfg_name:34s} {obj or 'cube':4s}: {n} envs x {steps} steps (750-step episodes), compared every {every} steps: bit-identical; "
          f"up to {100 * wall:.1f} % of the envs with a live boundary contact", flush=True)
