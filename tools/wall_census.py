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

# share of env-steps with a pushing corner ON THE CONE (above the vertical ring): where the horizontal-normal model of the cube corners deviates from the
# tilted surface (tests/test_contact_scenarios.py: test_cube_corner_on_the_cone_..., test_random_actions_rarely_put_a_pushing_corner_on_the_cone)
m = lib.default_model()
wz = torch.tensor(list(m.wall_z), device="cuda:0"); wr = torch.tensor(list(m.wall_r), device="cuda:0")
sgn = torch.tensor([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=torch.float32, device="cuda:0") * float(m.cube_half)
def quat_rot(q):                      # xyzw -> [n, 3, 3]
    x, y, z, w = q[0], q[1], q[2], q[3]
    return torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                        torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                        torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)
tot = push_n = cone_n = 0
for k in range(600):
    eng.step(ring[k % 16])
    if k % 25 == 0:
        st = eng.state
        push = (st[capi.S_CW_FACE] != 0) & (st[capi.S_LAM_CW:capi.S_LAM_CW + 12:3] > 0).any(0)
        R = quat_rot(st[capi.S_CUBE_Q:capi.S_CUBE_Q + 4])
        pts = st[capi.S_CUBE_P:capi.S_CUBE_P + 3].T[:, None, :] + torch.einsum("nij,cj->nci", R, sgn)
        rho = torch.hypot(pts[..., 0], pts[..., 1]); z = pts[..., 2]
        r_at = torch.full_like(z, float(wr[0]))
        for i in range(3):
            seg = z > wz[i]
            r_at = torch.where(seg, wr[i] + (z - wz[i]) * (wr[i + 1] - wr[i]) / (wz[i + 1] - wz[i]), r_at)
        on_cone = ((r_at - rho) < 1.5e-3) & (z > wz[0])
        tot += n; push_n += int(push.sum()); cone_n += int((push & on_cone.any(1)).sum())
print(f"env-steps sampled {tot}: with a pushing boundary corner {push_n} ({100.0 * push_n / tot:.2f} %), of those with a corner at the surface of the CONE "
      f"(above the {float(wz[0]) * 1e3:.0f} mm ring) {cone_n} ({100.0 * cone_n / tot:.4f} % of all env-steps)")
