"""The in-repo PPO (BASELINE config 5 glue) on the CPU with the oracle env injected: shapes, finite
losses, parameters move, GAE against a hand computation, and 2-rank gradient averaging (gloo)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from leibnizgym_amd.config import gym_config
from leibnizgym_amd.envs import TrifingerEnv
from leibnizgym_amd.ppo import ActorCritic, PPOConfig, PPOTrainer
from leibnizgym_amd.utils.rlg_train import RlGamesGpuEnvAdapter
from leibnizgym_amd.wrappers import VecTaskPython


def make(oracle, n=32, **kw):
    cfg = gym_config("trifinger_difficulty_4")
    cfg.update(num_instances=n, seed=1, physics_engine="physx", asymmetric_obs=True, episode_length=20)
    env = TrifingerEnv(config=cfg, device="cpu", verbose=False, lib=oracle, **kw)
    return env, RlGamesGpuEnvAdapter("rlgpu", n, env=VecTaskPython(env, rl_device="cpu"))


def test_network_shapes_match_asymm_yaml():
    net = ActorCritic(41, 113, 9, [400, 200, 100])
    actor = sum(p.numel() for p in net.actor.parameters()) + net.log_std.numel()
    critic = sum(p.numel() for p in net.critic.parameters())
    assert actor == 41 * 400 + 400 + 400 * 200 + 200 + 200 * 100 + 100 + 100 * 9 + 9 + 9      # ~118 k
    assert critic == 113 * 400 + 400 + 400 * 200 + 200 + 200 * 100 + 100 + 100 + 1            # ~146 k


def test_two_epochs_run_and_learn_something(oracle):
    env, ad = make(oracle)
    tr = PPOTrainer(ad, 41, 113, 9, PPOConfig(horizon=8, minibatches=4, mini_epochs=2), device="cpu")
    before = [p.detach().clone() for p in tr.net.parameters()]
    stats = tr.train(2)
    assert len(stats) == 2 and all(torch.isfinite(torch.tensor([s["loss"], s["kl"]])).all() for s in stats)
    assert any(not torch.equal(a, b) for a, b in zip(before, tr.net.parameters()))
    assert tr.frames == 2 * 8 * 32 and stats[-1]["c_loss"] >= 0


def test_gae_matches_hand_computation(oracle):
    env, ad = make(oracle, n=4)
    c = PPOConfig(horizon=3, minibatches=1, mini_epochs=1)
    tr = PPOTrainer(ad, 41, 113, 9, c, device="cpu")
    buf = tr.rollout()
    r, v, d = buf["rew"], buf["val"], buf["done"]
    adv = torch.zeros(3, 4)
    last = torch.zeros(4)
    for t in (2, 1, 0):
        nd = 1 - d[t]
        delta = r[t] + c.gamma * v[t + 1] * nd - v[t]
        last = delta + c.gamma * c.tau * nd * last
        adv[t] = last
    assert torch.allclose(buf["adv"], adv) and torch.allclose(buf["ret"], adv + v[:3])


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_util import load_oracle
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    env, ad = make(load_oracle(), n=16, env_id_offset=rank * 16, global_num_instances=32)
    tr = PPOTrainer(ad, 41, 113, 9, PPOConfig(horizon=4, minibatches=2, mini_epochs=1, seed=5), device="cpu")
    tr.train(1)
    torch.save([p.detach().clone() for p in tr.net.parameters()], os.path.join(out, f"w{rank}.pt"))
    dist.destroy_process_group()


def test_data_parallel_ranks_stay_in_sync(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w0, w1 = (torch.load(os.path.join(tmp_path, f"w{r}.pt")) for r in range(2))
    # different env shards and different sampled actions, one averaged gradient: identical weights afterwards
    assert all(torch.allclose(a, b, atol=1e-6) for a, b in zip(w0, w1))


@pytest.mark.gpu
def test_split_update_matches_unsplit(hip):
    """The minibatch step split at the gradient exchange (flat gradient buffer -> all-reduce -> optimiser; what every rank of a
    data-parallel run executes) takes the same optimisation steps as the one-rank form."""
    def run(split):
        cfg = gym_config("trifinger_difficulty_4")
        cfg.update(num_instances=256, seed=1, physics_engine="physx", asymmetric_obs=True, episode_length=20)
        env = TrifingerEnv(config=cfg, device="cuda:0", verbose=False)
        ad = RlGamesGpuEnvAdapter("rlgpu", 256, env=VecTaskPython(env, rl_device="cuda:0"))
        tr = PPOTrainer(ad, 41, 113, 9, PPOConfig(horizon=8, minibatches=4, mini_epochs=2), device="cuda:0")
        if split:                                     # one rank: the average of one gradient is itself
            from types import SimpleNamespace
            tr.dist_on = True
            tr.dist = SimpleNamespace(all_reduce=lambda *a, **k: None, get_world_size=lambda g=None: 1,
                                      ReduceOp=SimpleNamespace(SUM=None))
        torch.manual_seed(11)
        stats = tr.train(2)
        return [p.detach().clone() for p in tr.net.parameters()], stats
    eager, s0 = run(False)
    split, s2 = run(True)
    for a, c in zip(eager, split):
        assert torch.allclose(a, c, atol=2e-5, rtol=1e-4)
    assert abs(s0[-1]["kl"] - s2[-1]["kl"]) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("writer,reader", [(False, True), (True, False), (True, True)])
def test_checkpoint_restores_across_optimiser_paths(hip, tmp_path, writer, reader):
    """A checkpoint written by the torch optimiser (fused_kernels=False, the CPU form) restores into the flat hand-written
    optimiser of the GPU path and the reverse: moments per parameter, step counter and learning rates arrive, and the next
    update is the one the writer itself would have taken."""
    def make(fused):
        cfg = gym_config("trifinger_difficulty_4")
        cfg.update(num_instances=256, seed=1, physics_engine="physx", asymmetric_obs=True, episode_length=20)
        env = TrifingerEnv(config=cfg, device="cuda:0", verbose=False)
        ad = RlGamesGpuEnvAdapter("rlgpu", 256, env=VecTaskPython(env, rl_device="cuda:0"))
        return PPOTrainer(ad, 41, 113, 9, PPOConfig(horizon=8, minibatches=4, mini_epochs=2, fused_kernels=fused), device="cuda:0")
    a = make(writer)
    a.train(2)
    path = a.save(os.path.join(tmp_path, "ck.pth"))
    b = make(reader)
    b.restore(path)
    sa, sb = a._optimizer_state(), b._optimizer_state()
    assert sa["step"] == sb["step"] > 0 and sa["lrs"] == pytest.approx(sb["lrs"])
    for k in sa["exp_avg"]:
        assert torch.equal(sa["exp_avg"][k], sb["exp_avg"][k]) and torch.equal(sa["exp_avg_sq"][k], sb["exp_avg_sq"][k]), k
    assert float(sa["exp_avg_sq"]["log_std"].abs().sum()) > 0.0
    for (ka, pa), (kb, pb) in zip(a.net.state_dict().items(), b.net.state_dict().items()):
        assert torch.equal(pa, pb), ka
    # the same minibatch through both trainers: same parameters afterwards (two implementations of clip + Adam: tolerance)
    buf = a.rollout()
    torch.manual_seed(5); a.update(buf)
    torch.manual_seed(5); b.update(buf)
    for pa, pb in zip(a.net.state_dict().values(), b.net.state_dict().values()):
        assert torch.allclose(pa, pb, atol=3e-5, rtol=1e-4)


def test_split_k_linear_matches_linear():
    """The batch-parallel weight gradient of the trainer's linear layers is the ordinary one."""
    from leibnizgym_amd.ppo import _SplitKLinear
    torch.manual_seed(0)
    x = torch.randn(2048, 41, requires_grad=True)
    w = torch.randn(400, 41, requires_grad=True)
    b = torch.randn(400, requires_grad=True)
    g = torch.randn(2048, 400)
    _SplitKLinear.apply(x, w, b).backward(g)
    got = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    torch.nn.functional.linear(x, w, b).backward(g)
    for a, t in zip(got, (x, w, b)):
        assert torch.allclose(a, t.grad, rtol=1e-5, atol=1e-3 * float(t.grad.abs().max()))
    # a batch the slices do not divide falls back to the single GEMM
    x2 = torch.randn(100, 41, requires_grad=True)
    _SplitKLinear.apply(x2, w, b).sum().backward()
    assert x2.grad.shape == (100, 41)


@pytest.mark.parametrize("central", [True, False])
def test_legacy_optimizer_states_restore_and_bad_ones_are_rejected_before_the_weights_move(oracle, tmp_path, central):
    """The two round-2 checkpoint formats of the optimiser state - FlatClipAdam's flat buffers (parameters padded to 4 floats; actor group
    then critic group for a central-value net, net.parameters() order - log_std FIRST - otherwise) and torch.optim.Adam.state_dict() -
    restore to the same per-parameter moments as the current format; a state of another network is rejected before a weight is
    overwritten."""
    env, ad = make(oracle)
    cfg = PPOConfig(horizon=8, minibatches=4, mini_epochs=2)
    sdim = 113 if central else 0                      # state_dim 0: no central value network (the critic reads obs)
    a = PPOTrainer(ad, 41, sdim, 9, cfg, device="cpu")
    assert a.net.central == central
    a.train(1)
    want = a._optimizer_state()
    assert float(want["exp_avg_sq"]["log_std"].abs().sum()) > 0.0
    # torch format
    torch_sd = a.opt.state_dict()
    # flat format, written here the way round 2's FlatClipAdam laid it out
    order = a.net.actor_parameters() + a.net.critic_parameters() if a.net.central else list(a.net.parameters())
    names = {id(p): n for n, p in a.net.named_parameters()}
    if not central:
        assert names[id(order[0])] == "log_std" and names[id(a.net.actor_parameters()[-1])] == "log_std"     # the two orders differ
    size = sum((p.numel() + 3) & ~3 for p in order)
    fm, fv, off = torch.zeros(size), torch.zeros(size), 0
    for p in order:
        fm[off:off + p.numel()] = want["exp_avg"][names[id(p)]].reshape(-1)
        fv[off:off + p.numel()] = want["exp_avg_sq"][names[id(p)]].reshape(-1)
        off += (p.numel() + 3) & ~3
    flat_sd = {"kind": "flat_clip_adam", "m": fm, "v": fv, "step": torch.tensor(want["step"]), "lr": torch.tensor(want["lrs"])}
    for sd in (torch_sd, flat_sd):
        ck = a.state_dict(); ck["optimizer"] = sd
        path = os.path.join(tmp_path, "legacy.pth"); torch.save(ck, path)
        env2, ad2 = make(oracle)
        b = PPOTrainer(ad2, 41, sdim, 9, cfg, device="cpu")
        b.restore(path)
        got = b._optimizer_state()
        assert got["step"] == want["step"] and got["lrs"] == pytest.approx(want["lrs"])
        for k in want["exp_avg"]:
            assert torch.equal(got["exp_avg"][k], want["exp_avg"][k]) and torch.equal(got["exp_avg_sq"][k], want["exp_avg_sq"][k]), k
    # mismatched states: rejected, and the model of the reader is untouched
    env3, ad3 = make(oracle)
    c = PPOTrainer(ad3, 41, sdim, 9, cfg, device="cpu")
    before = {k: v.clone() for k, v in c.net.state_dict().items()}
    bad_flat = dict(flat_sd, m=fm[:-4], v=fv[:-4])
    bad_named = {"kind": "adam_per_parameter", "exp_avg": dict(want["exp_avg"], log_std=torch.zeros(3)), "exp_avg_sq": want["exp_avg_sq"],
                 "step": 1.0, "lrs": want["lrs"]}
    bad_torch = {"state": {}, "param_groups": [{"params": [0, 1], "lr": 1e-3}]}
    for sd in (bad_flat, bad_named, bad_torch, {"kind": "something else"}):
        ck = a.state_dict(); ck["optimizer"] = sd
        path = os.path.join(tmp_path, "bad.pth"); torch.save(ck, path)
        with pytest.raises(ValueError):
            c.restore(path)
        for k, v in c.net.state_dict().items():
            assert torch.equal(v, before[k]), k
