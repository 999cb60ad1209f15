"""Developer tool (GPU box): per-phase cycle counts of the fused step from s_memtime stamps (build: make -C leibnizgym_amd/csrc
libtrifinger_hip_timing.so; lane 0 of every wavefront writes a stamp at each STAMP site of tf_roles.h)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import bench
from leibnizgym_amd.engine import TrifingerEngine, make_config
from leibnizgym_amd import _capi

lib = _capi.TfLib(os.environ.get("TF_LIB") or os.path.join(REPO, "leibnizgym_amd", "csrc", "libtrifinger_hip_timing.so"))     # TF_LIB: another -DTF_PHASE_TIMING build
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
asym = True
dr = len(sys.argv) > 3 and sys.argv[3] == "dr"          # phase_timing.py N WARMUP dr: every domain-randomisation feature (the EXT kernel)
eng = TrifingerEngine(make_config(lib, n, seed=7, **bench.workload_kwargs(asym, 4, dr)), device="cuda:0", lib=lib)
g = torch.Generator(device="cuda:0").manual_seed(1)
ring = [(torch.rand(n, 9, device="cuda:0", generator=g) * 2 - 1) for _ in range(8)]
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 10      # steps after the reset before sampling starts (600: the cubes have reached the boundary)
if os.environ.get("VARIANT"):                             # narrow / wide: force one instantiation of the step kernel (default: what tf_create picks for N)
    eng.kernel_variant = os.environ["VARIANT"]
eng.reset()
acc = []
for k in range(warm + 30):
    eng.step(ring[k % 8])
    if k >= warm:
        torch.cuda.synchronize()
        st = eng.scratch.view(torch.int32).view(-1, 4, 64).cpu().numpy().astype(np.int64) & 0xffffffff
        acc.append(st)
S = np.stack(acc)                 # [steps, wg, role, id]
def d(a, b, role):                # median over steps and workgroups of stamp[b] - stamp[a] for a role (0..2 fingers pooled)
    sel = S[:, :, role, :] if role == 3 else S[:, :, 0:3, :].reshape(S.shape[0], -1, 64)
    return np.median((sel[..., b] - sel[..., a]) & 0xffffffff)
def v(a, role):
    sel = S[:, :, role, :] if role == 3 else S[:, :, 0:3, :].reshape(S.shape[0], -1, 64)
    return np.median(sel[..., a])
print(f"kernel variant: {eng.kernel_variant}")
print(f"N={n} asym={asym}{' + every DR feature (EXT kernel)' if dr else ''}, sampled {warm}..{warm + 29} steps after the reset: median s_memtime ticks; finger role | cube role")
rows = [("loads + action tile (to #1)", 0, 1), ("resets, action_buf, torque", 1, 2)]
for s in (0, 1):
    b = 4 + 12 * s
    rows += [(f"sub{s}: free motion / corners (to S1 arrive)", (2 if s == 0 else 4 + 7), b + 0), (f"sub{s}: S1 wait", b + 0, b + 1),
             (f"sub{s}: contact generation / finger-finger pass", b + 1, b + 2), (f"sub{s}: S2 wait", b + 2, b + 3),
             (f"sub{s}: publish record (finger) + S3", b + 3, b + 5), (f"sub{s}: sweeps incl. barriers", b + 5, b + 6),
             (f"sub{s}: wrench / integrate", b + 6, b + 7)]
rows += [("to post start", 4 + 12 + 7, 30), ("tip FK / nan flags", 30, 31), ("P1 wait", 31, 32), ("emit tile", 32, 33), ("P3 wait", 33, 34), ("rewards, stores, finish", 34, 35), ("TOTAL", 0, 35)]
for lab, a, b in rows:
    print(f"  {lab:55s} {d(a, b, 0):9.0f} | {d(a, b, 3):9.0f}")
print(f"  cube role behind P3: rewards {d(34, 36, 3):.0f} | statistics fold issued {d(36, 37, 3):.0f} | state rows {d(37, 38, 3):.0f} | counters, time-out, dones {d(38, 39, 3):.0f} | "
      f"statistics fold consumed, end {d(39, 35, 3):.0f}")
for s in (0, 1):
    b = 4 + 12 * s
    print(f"  sub{s}: barrier wait inside sweeps: finger {v(b + 8, 0):.0f} | cube {v(b + 8, 3):.0f};  cube finger-cube rows {v(b + 10, 3):.0f}; up to the end of the floor rows (incl. W1) {v(b + 11, 3):.0f}")
# how the workgroups spread around their median: the launch is one round of workgroups, so it ends with its slowest one
cube = S[:, :, 3, :]
tot = (cube[:, :, 35] - cube[:, :, 0]) & 0xffffffff
pct = lambda a: "  ".join(f"p{q} {np.percentile(a, q):.0f}" for q in (1, 50, 90, 99, 100))      # noqa: E731
print(f"  per workgroup (cube role) start -> end: {pct(tot)}      (s_memtime of different XCDs is not synchronised: no cross-workgroup spans)")
for s_ in (0, 1):
    b = 4 + 12 * s_
    sw = (cube[:, :, b + 6] - cube[:, :, b + 5]) & 0xffffffff
    print(f"  sub{s_} sweeps per workgroup (cube role): {pct(sw)}")
# what the slow workgroups spend their sweeps on (cube role, substep 0): finger-cube rows | W1 + floor rows | boundary rows + W2
b = 4
sw = ((cube[:, :, b + 6] - cube[:, :, b + 5]) & 0xffffffff).astype(np.float64).ravel()
fc = cube[:, :, b + 10].astype(np.float64).ravel()
fl = cube[:, :, b + 11].astype(np.float64).ravel() - fc
rest = sw - fc - fl
for lab, lo, hi in (("all", 0, 100), ("p40-p60", 40, 60), ("p85-p95", 85, 95), ("p98-p100", 98, 100)):
    a, z = np.percentile(sw, lo), np.percentile(sw, hi)
    m = (sw >= a) & (sw <= z)
    print(f"  sweeps of sub0, workgroups {lab:9s}: total {np.median(sw[m]):6.0f} = finger-cube rows {np.median(fc[m]):6.0f} + W1 and floor rows {np.median(fl[m]):6.0f} + boundary rows and W2 {np.median(rest[m]):6.0f}")
eng.close()
