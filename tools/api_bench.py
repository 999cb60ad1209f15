"""Developer tool (GPU box): throughput through the PUBLIC API layers at the headline size - raw engine, TrifingerEnv.step,
VecTaskPython.step (+ action / observation clipping) and the RL-Games adapter (+ get_state) - same workload as bench.py."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from leibnizgym_amd.config import gym_config
from leibnizgym_amd.envs import TrifingerEnv
from leibnizgym_amd.utils.rlg_train import RlGamesGpuEnvAdapter
from leibnizgym_amd.wrappers import VecTaskPython

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = "cuda:0"
cfg = gym_config("trifinger_difficulty_4")
cfg.update(num_instances=n, seed=7, physics_engine="physx", asymmetric_obs=True, command_mode="torque")
env = TrifingerEnv(config=cfg, device=dev, verbose=False)
vec = VecTaskPython(env, rl_device=dev)
ad = RlGamesGpuEnvAdapter("rlgpu", n, env=vec)
g = torch.Generator(device=dev).manual_seed(7)
ring = [(torch.rand(n, 9, device=dev, generator=g) * 2 - 1).contiguous() for _ in range(16)]


def timed(label, fn):
    for k in range(10):
        fn(ring[k % 16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        fn(ring[k % 16])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label:44s} {n * steps / dt:.4e} env-steps/s  {dt / steps * 1e6:7.1f} us/step", flush=True)


env.reset()
timed("engine.step (C ABI, what bench.py times)", lambda a: env._engine.step(a))
timed("TrifingerEnv.step", lambda a: env.step(a))
timed("VecTaskPython.step", lambda a: vec.step(a))
timed("RlGamesGpuEnvAdapter.step (obs + states dict)", lambda a: ad.step(a))
