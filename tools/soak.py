"""Soak run on the GPU box: many steps at the headline size, invariants checked every 1000 steps."""
import sys, os, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from leibnizgym_amd.engine import TrifingerEngine, make_config
from leibnizgym_amd import _capi as capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
lib = capi.load_hip_library()
kw = bench.workload_kwargs(True)
kw.update(domain_randomization={"activate": True}, episode_length=300)
box = len(sys.argv) > 3 and sys.argv[3] == "box"            # the phase-3 cuboid instead of the cube
if box:
    kw.update(model=lib.box_model((0.02, 0.08, 0.02), 500.0))
eng = TrifingerEngine(make_config(lib, n, seed=11, **kw), device="cuda:0", lib=lib)
g = torch.Generator(device="cuda:0").manual_seed(3)
ring = [(torch.rand(n, 9, device="cuda:0", generator=g) * 2 - 1) for _ in range(32)]
if os.environ.get("VARIANT"):                               # force one instantiation of the step kernel (same bits: a check that an event is the physics')
    eng.kernel_variant = os.environ["VARIANT"]
eng.reset()
nonfinite = torch.zeros((), device="cuda:0"); resets = torch.zeros((), device="cuda:0")
t0 = time.perf_counter()
deep_box = []
soft_box = []
for k in range(steps):
    eng.step(ring[k % 32])
    nonfinite += eng.info[capi.INFO_NUM_NONFINITE]; resets += eng.info[capi.INFO_NUM_RESETS]
    if k % 1000 == 999:
        st = eng.state
        ok = bool(torch.isfinite(st).all()) and bool(torch.isfinite(eng.obs).all()) and bool(torch.isfinite(eng.states).all())
        zmin = float(st[capi.S_CUBE_P + 2].min()); rmax = float(torch.hypot(st[capi.S_CUBE_P], st[capi.S_CUBE_P + 1]).max())
        zmax = float(st[capi.S_CUBE_P + 2].max()); vmax = float(st[capi.S_CUBE_V:capi.S_CUBE_V + 3].abs().max())
        tipz = float(st[capi.S_TIP_P + 2:capi.S_TIP_P + 9:3].min())
        if k % 10000 == 9999 or not ok:
            print(f"step {k+1}: finite={ok} cube z [{zmin:.4f}, {zmax:.4f}] r max {rmax:.4f} |v| max {vmax:.2f} tip z min {tipz:.4f} "
                  f"|qd| max {float(st[9:18].abs().max()):.2f} nonfinite so far {float(nonfinite):.0f} resets {float(resets):.0f} "
                  f"mean reward {float(eng.reward.mean()):.3f}", flush=True)
        # nothing leaves the arena, sinks through the floor or the table, or runs away.  The thin bar (half thickness 10 mm) is watched separately: its
        # deepest samples are reported (a finger can press the 16 g bar several millimetres into the table for a few steps - the slow case of the
        # solver), a centre below 2 mm fails
        if box and zmin < 0.006:
            deep_box.append((k + 1, zmin))
        # the round-3 bound (centre above 4 mm) stays as a COUNTED soft failure: more than 2 % of the samples below it fails the soak, so that a
        # regression of the penetration it was written for is still caught; a centre below 2 mm fails at once
        if box and zmin <= 0.004:
            soft_box.append((k + 1, zmin))
            assert len(soft_box) <= max(2, (steps // 1000) // 50), ("bar pressed more than 6 mm into the table too often", soft_box[-5:])
        assert ok and rmax < 0.27 and zmin > (0.002 if box else 0.02) and zmax < 1.0 and vmax < 20.0 and tipz > -0.003, (zmin, zmax, rmax, vmax, tipz)
torch.cuda.synchronize()
print(f"{n} envs x {steps} steps in {time.perf_counter()-t0:.1f} s; non-finite envs caught: {float(nonfinite):.0f}")
if box:
    print(f"samples (every 1000 steps, minimum over the envs) with the bar's centre below 6 mm (4 mm into the table): {len(deep_box)} of {steps // 1000}"
          + (f"; deepest: centre at {min(z for _, z in deep_box) * 1e3:.1f} mm (step {min(deep_box, key=lambda t: t[1])[0]})" if deep_box else "")
          + f"; below 4 mm (the counted soft bound, at most {max(2, (steps // 1000) // 50)} allowed): {len(soft_box)}")
