#!/usr/bin/env python3
"""End-to-end PPO on the native TriFinger env (BASELINE config 5: difficulty 4, 8192 envs per GPU).

    python scripts/train_ppo.py gym=trifinger_difficulty_4 args.num_envs=8192 [epochs=20] [args.checkpoint=run/nn/trifinger.pth] [save_dir=run/nn]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_ppo.py ...

One process per GPU; envs are sharded (no data-path collective), gradients are averaged with one fused RCCL
all-reduce per minibatch.  Uses RL-Games' hyper-parameters of resources/config/rlg/asymm.yaml (leibnizgym_amd/ppo.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from leibnizgym_amd.config import compose  # noqa: E402
from leibnizgym_amd.envs import TrifingerEnv  # noqa: E402
from leibnizgym_amd.ppo import PPOConfig, PPOTrainer  # noqa: E402
from leibnizgym_amd.utils.rlg_train import RlGamesGpuEnvAdapter  # noqa: E402
from leibnizgym_amd.wrappers import VecTaskPython  # noqa: E402


def main(argv):
    epochs, save_dir = 20, None
    rest = []
    for a in argv:
        if a.startswith("epochs="):
            epochs = int(a.split("=", 1)[1])
        elif a.startswith("save_dir="):
            save_dir = a.split("=", 1)[1]
        else:
            rest.append(a)
    cfg = compose(rest)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook for boxes with ONE GPU (tests/test_bench_multiproc_gpu.py): every rank on cuda:0, gradients through gloo
    one_device = os.environ.get("TF_BENCH_SINGLE_DEVICE_TEST", "") == "1"
    if one_device:
        local = 0
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    # launched by torch.distributed.run (RANK is set): a process group also for a world of ONE, so that `--nproc-per-node 1` runs the very code path of
    # an 8-GPU job - RCCL initialisation, the gradient all-reduce of every minibatch, the KL all-reduce of every mini-epoch (tests/test_rccl_gpu.py)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if world > 1 or launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))
    n = cfg["gym"]["num_instances"]                 # envs per GPU
    env = TrifingerEnv(config=cfg["gym"], device=dev, verbose=False, env_id_offset=rank * n,
                       global_num_instances=world * n)
    adapter = RlGamesGpuEnvAdapter("rlgpu", n, env=VecTaskPython(env, rl_device=dev))
    # hyper-parameters from the agent tree (asymm.yaml); minibatch = envs per GPU, as update_cfg sets it
    tr = PPOTrainer(adapter, env.get_obs_dim(), env.get_state_dim(), env.get_action_dim(),
                    PPOConfig.from_rlg(cfg["rlg"], num_envs=n), device=dev)
    if cfg["args"]["checkpoint"]:
        tr.restore(cfg["args"]["checkpoint"])
    t0 = time.perf_counter()
    marks = []                                      # (time, frames) after every epoch: the steady rate excludes graph capture and warm-up

    def log(st):
        if rank == 0:
            now = time.perf_counter()
            marks.append((now, st["frames"] * world))
            back = marks[max(0, len(marks) - 11)]
            steady = (marks[-1][1] - back[1]) / max(now - back[0], 1e-9) if len(marks) > 1 else 0.0
            print(f"epoch {st['epoch']:4d} frames {st['frames'] * world:10d} reward/step {st['mean_reward']:9.3f} "
                  f"kl {st['kl']:.4f} lr {st['lr']:.2e} loss {st['loss']:.4f}  {st['frames'] * world / (now - t0):.3e} frames/s "
                  f"since start, {steady:.3e} over the last {min(len(marks) - 1, 10)} epochs", flush=True)
    tr.train(epochs, log, checkpoint_dir=save_dir if rank == 0 else None)
    if world > 1 or launched:
        if rank == 0:
            print(f"collectives: backend {dist.get_backend()}, world {dist.get_world_size()}, gradient all-reduces {tr.n_grad_allreduce}, "
                  f"KL all-reduces {tr.n_kl_allreduce}", flush=True)
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
