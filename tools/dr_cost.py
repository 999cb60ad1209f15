"""Developer tool (GPU box): what each domain-randomisation feature of BASELINE configs[3] costs in the fused step (HIP events, 65536 envs unless given).
    python tools/dr_cost.py [envs]"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from leibnizgym_amd.engine import TrifingerEngine, make_config
from leibnizgym_amd import _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
lib = _capi.load_hip_library()
full = dict(bench.FULL_DR)
neutral = {"robot_base_position": (0.0, 0.0, 0.0), "stage_position": (0.0, 0.0), "friction_robot": (1.0, 1.0), "friction_object": (1.0, 1.0), "friction_stage": (1.0, 1.0)}
variants = [("no domain randomisation (headline kernel)", None),
            ("six scale factors only (headline kernel, DR rows loaded)", dict(full, obs_noise=0.0, action_repeat_prob=0.0, **neutral)),
            ("+ observation noise", dict(full, action_repeat_prob=0.0, **neutral)),
            ("+ action repeat", dict(full, **neutral)),
            ("+ base / stage offsets, friction per body = every feature (EXT kernel)", full),
            ("every feature but the observation noise (EXT kernel)", dict(full, obs_noise=0.0))]
for name, dr in variants:
    kw = bench.workload_kwargs(True, 4, False)
    if dr is not None:
        kw["domain_randomization"] = dr
    eng = TrifingerEngine(make_config(lib, n, seed=7, **kw), device="cuda:0", lib=lib)
    eng.reset()
    eng.steps.copy_(torch.randint(0, 750, (n,), device="cuda:0", dtype=torch.int64))
    for _ in range(750):
        eng.step_random()
    best = 1e9
    for rep in range(3):
        eng.enable_kernel_timing(1000, 8)
        for _ in range(1000):
            eng.step_random()
        torch.cuda.synchronize()
        ms, cnt = eng.kernel_time_ms()
        best = min(best, ms / cnt * 1e3)
    print(f"{name:75s} {best:7.2f} us", flush=True)
    eng.close()
