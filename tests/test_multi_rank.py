"""World-size-2 and world-size-8 runs on CPU (gloo): the shards of the env range reproduce one big engine bit for bit, and the
optional episode-statistics all-reduce returns the big engine's info.  The 8-rank case is the shape of BASELINE configs[3]
(131072 envs over 8 GPUs: shard_range(131072, r, 8)) at 8 x 64 envs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import parity_util as pu
from leibnizgym_amd.sharding import shard_range

TOTAL, STEPS, CFG = 300, 60, "envdefault_position"


def test_shard_range_partitions():
    for total, world in ((300, 2), (65536, 8), (10, 3), (7, 8)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (o0, c0), (o1, _) in zip(spans, spans[1:]):
            assert o0 + c0 == o1


def _worker(rank, world, port, out_dir, TOTAL=TOTAL, STEPS=STEPS, CFG=CFG):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from oracle_util import load_oracle
    from leibnizgym_amd.engine import TrifingerEngine, make_config
    from leibnizgym_amd.sharding import EpisodeStatsReducer, shard_range as sr
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = load_oracle()
    off, cnt = sr(TOTAL, rank, world)
    kw = dict(pu.CONFIGS[CFG])
    eng = TrifingerEngine(make_config(lib, cnt, seed=3, episode_length=25, env_id_offset=off, global_num_envs=TOTAL, **kw),
                          device="cpu", lib=lib)
    red = EpisodeStatsReducer(eng, TOTAL, every=1)
    eng.reset()
    infos = []
    for t in range(STEPS):
        act = pu.actions_for(t, TOTAL, eng.action_dim, 3)[off:off + cnt].contiguous()
        eng.step(act)
        assert red.step()
        infos.append(red.result().numpy().copy())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), state=eng.state.numpy(), obs=eng.obs.numpy(),
             states=eng.states.numpy(), reward=eng.reward.numpy(), steps=eng.steps.numpy(),
             reset_count=eng.reset_count.numpy(), info=np.stack(infos), off=off, cnt=cnt)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,TOTAL,STEPS,CFG", [(2, TOTAL, STEPS, CFG), (8, 512, 40, "d4_domain_randomization_extended")])
def test_shards_equal_one_engine(oracle, tmp_path, world, TOTAL, STEPS, CFG):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path), TOTAL, STEPS, CFG), nprocs=world, join=True)
    # the single-engine run
    from leibnizgym_amd.engine import TrifingerEngine, make_config
    kw = dict(pu.CONFIGS[CFG])
    eng = TrifingerEngine(make_config(oracle, TOTAL, seed=3, episode_length=25, **kw), device="cpu", lib=oracle)
    eng.reset()
    infos = []
    for t in range(STEPS):
        eng.step(pu.actions_for(t, TOTAL, eng.action_dim, 3))
        infos.append(eng.info.numpy().copy())
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert [int(p["off"]) for p in parts] == [shard_range(TOTAL, r, world)[0] for r in range(world)]
    assert [int(p["cnt"]) for p in parts] == [TOTAL // world] * world
    for key, axis in (("state", 1), ("obs", 0), ("states", 0), ("reward", 0), ("steps", 0), ("reset_count", 0)):
        got = np.concatenate([p[key] for p in parts], axis=axis)
        want = getattr(eng, key).numpy()
        assert got.shape == want.shape
        assert np.array_equal(got.view(np.uint32) if got.dtype == np.float32 else got,
                              want.view(np.uint32) if want.dtype == np.float32 else want), key
    assert int(eng.reset_count.max()) >= 2          # resets (and, with success termination on, goal resets) happened inside the window
    for r in range(world):
        np.testing.assert_allclose(parts[r]["info"][:, :11], np.stack(infos)[:, :11], rtol=2e-5, atol=2e-4)
