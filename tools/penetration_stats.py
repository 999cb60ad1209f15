"""Developer tool (GPU box): penetration-depth statistics of a headline-size rollout: fingertip sphere vs
floor, cube corners vs floor, fingertip sphere vs cube - percentiles over every env and sampled step, random actions and (optionally)
domain randomisation.   python tools/penetration_stats.py [num_envs] [steps] [dr|-] [contact_slack]   (a slack other than the model's: the
experiment of DESIGN.md section 4; the fused-step time is printed beside the depths)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from leibnizgym_amd.engine import TrifingerEngine, make_config
from leibnizgym_amd import _capi as capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
dr = len(sys.argv) > 3 and sys.argv[3] == "dr"
lib = capi.load_hip_library()
dev = "cuda:0"
kw = bench.workload_kwargs(True)
kw.update(episode_length=300)
slack = float(sys.argv[4]) if len(sys.argv) > 4 else None
if slack is not None:
    mm = lib.default_model()
    mm.contact_slack = slack
    kw.update(model=mm)
if dr:
    kw.update(domain_randomization={"activate": True})
eng = TrifingerEngine(make_config(lib, n, seed=11, **kw), device=dev, lib=lib)
m = lib.default_model()
g = torch.Generator(device=dev).manual_seed(3)
ring = [(torch.rand(n, 9, device=dev, generator=g) * 2 - 1) for _ in range(32)]
eng.reset()
eng.enable_kernel_timing(steps)
R_TIP, HALF = float(m.cap_radius), float(m.cube_half)
J2, J3, CB = (torch.tensor(list(x), device=dev) for x in (m.j2_origin, m.j3_origin, m.cap_b))
YC, YS, HB = list(m.base_yaw_cos), list(m.base_yaw_sin), float(m.base_height)


def tip_spheres(q):
    """centres of the three fingertip spheres, world frame: [3][N, 3] (FK of the URDF chain: y, x, x joints)"""
    out = []
    for f in range(3):
        q1, q2, q3 = q[3 * f], q[3 * f + 1], q[3 * f + 2]
        def rot_x(a, v):
            c, s = torch.cos(a), torch.sin(a)
            return torch.stack([v[0].expand_as(a) if v[0].dim() == 0 else v[0], c * v[1] - s * v[2], s * v[1] + c * v[2]])
        def rot_y(a, v):
            c, s = torch.cos(a), torch.sin(a)
            return torch.stack([c * v[0] + s * v[2], v[1].expand_as(a) if v[1].dim() == 0 else v[1], -s * v[0] + c * v[2]])
        p = rot_x(q3, CB) + J3[:, None]
        p = rot_x(q2, p) + J2[:, None]
        p = rot_y(q1, p)
        out.append(torch.stack([YC[f] * p[0] - YS[f] * p[1], YS[f] * p[0] + YC[f] * p[1], p[2] + HB], dim=1))
    return out


def quat_rot_T(qt, v):      # R(q)^T v, q = xyzw rows [4, N], v [N, 3]
    x, y, z, w = qt
    R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)]),
                     torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)]),
                     torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)])])       # [3, 3, N]
    return torch.einsum("ijn,ni->nj", R, v), R


pen = {"fingertip-floor": [], "cube-floor": [], "fingertip-cube": []}
for k in range(steps):
    eng.step(ring[k % 32])
    if k % 25 == 24:
        st = eng.state
        size = st[capi.S_DR + 1] if dr else torch.ones(n, device=dev)
        tips = tip_spheres(st[capi.S_Q:capi.S_Q + 9])
        cp = st[capi.S_CUBE_P:capi.S_CUBE_P + 3].T
        hc = (HALF * size)[:, None]
        tf, tc = [], []
        for t in tips:
            tf.append(R_TIP - t[:, 2])
            loc, R = quat_rot_T(st[capi.S_CUBE_Q:capi.S_CUBE_Q + 4], t - cp)
            d = loc.abs() - hc
            outside = torch.clamp(d, min=0).norm(dim=1) + torch.clamp(d.max(dim=1).values, max=0)      # signed distance to the box
            tc.append(R_TIP - outside)
        pen["fingertip-floor"].append(torch.stack(tf).flatten())
        # lowest corner of the cube: z_c - sum_k hc |R_zk|
        pen["cube-floor"].append((hc[:, 0] * R[2].abs().sum(0)) - cp[:, 2])
        pen["fingertip-cube"].append(torch.stack(tc).flatten())
torch.cuda.synchronize()
ms_, cnt_ = eng.kernel_time_ms()
print(f"{n} envs x {steps} steps, random actions{', domain randomisation' if dr else ''}, sampled every 25 steps; penetration depth in mm "
      f"(positive = inside; contact_offset is 2 mm); contact_slack {slack if slack is not None else float(m.contact_slack):.4f} m; "
      f"fused step {ms_ / max(cnt_, 1) * 1e3:.2f} us (with a host synchronisation every 25 steps)")
for name, v in pen.items():
    x = torch.cat(v).float() * 1e3
    touching = x > -1.0
    qs = torch.quantile(x[touching][:8_000_000], torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev)) if touching.any() else torch.zeros(4)
    print(f"{name:16s}: in or near contact {100 * float(touching.float().mean()):5.1f} % of samples; of those: median {qs[0]:6.2f}  p90 {qs[1]:6.2f}  "
          f"p99 {qs[2]:6.2f}  p99.9 {qs[3]:6.2f}  max {float(x.max()):6.2f}")
