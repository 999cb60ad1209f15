"""Developer tool (GPU box): train the in-repo PPO on difficulty 1 (cube to a goal position on the table; default) or any other
difficulty (4 = the headline task: lift to a 6-DoF pose goal) and then roll the policy, reporting what the cube physically does -
distance to the goal, speeds, heights, fraction of cubes lifted, goal counts - so that a learning curve cannot hide an exploit of
the contact model.   python tools/ppo_learning_check.py [epochs] [num_envs] [seed] [fused|plain] [difficulty] [hydra-style overrides ...]
(overrides, e.g. gym.reward_terms.object_rot.weight=10000  gym.sim.physx.num_position_iterations=16  gym.native.solver=tgs)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from leibnizgym_amd import _capi as capi
from leibnizgym_amd.config import compose
from leibnizgym_amd.envs import TrifingerEnv
from leibnizgym_amd.ppo import PPOConfig, PPOTrainer
from leibnizgym_amd.utils.rlg_train import RlGamesGpuEnvAdapter
from leibnizgym_amd.wrappers import VecTaskPython

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 7
fused = (sys.argv[4] != "plain") if len(sys.argv) > 4 else True
difficulty = int(sys.argv[5]) if len(sys.argv) > 5 else 1
overrides = sys.argv[6:]
cfg = compose([f"gym=trifinger_difficulty_{difficulty}", f"args.num_envs={n}", "args.headless=True"] + overrides)
if overrides:
    print("overrides:", " ".join(overrides), flush=True)
_variant = os.environ.get("TF_LIB")                           # a developer build of the product library (A/B of a physics change through the learner)
env = TrifingerEnv(config=cfg["gym"], device="cuda:0", verbose=False, **({"lib": capi.TfLib(os.path.abspath(_variant))} if _variant else {}))
if _variant:
    print("library:", _variant, flush=True)
tr = PPOTrainer(RlGamesGpuEnvAdapter("rlgpu", n, env=VecTaskPython(env, rl_device="cuda:0")), env.get_obs_dim(), env.get_state_dim(),
                env.get_action_dim(), PPOConfig.from_rlg(cfg["rlg"], num_envs=n, seed=seed, fused_kernels=fused), device="cuda:0")
t0 = time.perf_counter()
every = 10 if epochs <= 300 else 50


def log(st):                                   # called by the trainer after every epoch: the state rows are this epoch's
    if st["epoch"] % every == every - 1:
        s_ = env._engine.state
        lifted = float((s_[capi.S_CUBE_P + 2] > 0.05).float().mean())
        d_ = (s_[capi.S_CUBE_P:capi.S_CUBE_P + 3] - s_[capi.S_GOAL_P:capi.S_GOAL_P + 3]).norm(dim=0)
        info_ = env._engine.info
        print(f"epoch {st['epoch']:4d} frames {st['frames']:10d} reward/step {st['mean_reward']:8.3f} kl {st['kl']:.4f} "
              f"{st['frames'] / (time.perf_counter() - t0):.3e} frames/s  cube above 5 cm {100 * lifted:5.1f} %  "
              f"dist to goal median {float(d_.median()) * 1e3:6.1f} mm  envs at the position / orientation goal "
              f"{int(info_[capi.INFO_POS_COUNT])} / {int(info_[capi.INFO_ORI_COUNT])}  sigma {float(tr.net.log_std.exp().mean()):.3f}", flush=True)


tr.train(epochs, log=log)
for DET in (False, True):
    print(f'---- play, deterministic={DET}, log_std {tr.net.log_std.detach().cpu().numpy().round(2)}')
    eng = env._engine
    dist, vmax, wmax, zmax, tipmin = [], 0.0, 0.0, 0.0, 1.0
    env.reset()
    tr.last = tr._unpack(tr.env.reset())
    rsum = 0.0
    for k in range(700):                                   # one episode (750 steps) without its time-out
        r_, info_ = tr.play(1, deterministic=DET)
        rsum += r_
        if k in (0, 1, 2, 50, 300, 699):
            print(f"play step {k}: mean reward {r_:.3f}", {kk: round(float(v), 4) for kk, v in info_.items() if "rewards" in kk})
        s = eng.state
        d = (s[capi.S_CUBE_P:capi.S_CUBE_P + 3] - s[capi.S_GOAL_P:capi.S_GOAL_P + 3]).norm(dim=0)
        vmax = max(vmax, float(s[capi.S_CUBE_V:capi.S_CUBE_V + 3].norm(dim=0).max()))
        wmax = max(wmax, float(s[capi.S_CUBE_W:capi.S_CUBE_W + 3].norm(dim=0).max()))
        zmax = max(zmax, float(s[capi.S_CUBE_P + 2].max()))
        tipmin = min(tipmin, float(s[capi.S_TIP_P + 2:capi.S_TIP_P + 9:3].min()))
        if k % 100 == 99:
            dist.append((k + 1, float(d.mean()), float(d.median()), float((d < 0.02).float().mean()),
                         float((s[capi.S_CUBE_P + 2] > 0.05).float().mean()),
                         {kk.split("/")[-2] if kk.endswith("count") else kk.split("/")[-1]: round(float(v), 1) for kk, v in info_.items() if "goal" in kk or "success" in kk}))
    for k, mean, med, frac, lifted, counts in dist:
        print(f"play step {k:4d}: distance to the goal mean {mean * 1e3:6.1f} mm  median {med * 1e3:6.1f} mm  within 2 cm {100 * frac:5.1f} %  "
              f"cube above 5 cm {100 * lifted:5.1f} %  env info {counts}")
    print(f"mean reward per step in play {rsum / 700:.3f}")
    print(f"over the episode: cube speed max {vmax:.2f} m/s, spin max {wmax:.1f} rad/s, height max {zmax * 1e3:.1f} mm (rest 32.5), "
          f"fingertip height min {tipmin * 1e3:.1f} mm (radius 10.2)")
