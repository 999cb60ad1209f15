"""Developer experiment (GPU box): does splitting the 65536 envs into two engines stepped concurrently on two streams
overlap the start/end latencies of the launches?  Prints aggregate env-steps/s for 1 x 65536, 2 x 32768 (two streams),
2 x 32768 (one stream) and 4 x 16384 (four streams)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from leibnizgym_amd.engine import TrifingerEngine, make_config
from leibnizgym_amd import _capi

lib = _capi.load_hip_library()
dev = "cuda:0"
steps = 2000


def run(parts, nstreams, join=False):
    n = 65536 // parts
    engs, rings, streams = [], [], [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    for p in range(parts):
        e = TrifingerEngine(make_config(lib, n, seed=7, env_id_offset=p * n, global_num_envs=65536, **bench.workload_kwargs(True)),
                            device=dev, lib=lib)
        g = torch.Generator(device=dev).manual_seed(7 + p)
        rings.append([(torch.rand(n, 9, device=dev, generator=g) * 2 - 1).contiguous() for _ in range(16)])
        e.reset()
        engs.append(e)
    torch.cuda.synchronize()

    main = torch.cuda.current_stream(dev)

    def loop(k0, k1):
        for k in range(k0, k1):
            if join:                            # fork: every part stream waits for what the caller queued so far
                ev = torch.cuda.Event(); ev.record(main)
                for st in streams:
                    st.wait_event(ev)
            for p, e in enumerate(engs):
                with torch.cuda.stream(streams[p % nstreams]):
                    e.step(rings[p][k % 16])
            if join:                            # join: the caller's stream continues after every part has finished
                for st in streams:
                    ev = torch.cuda.Event(); ev.record(st)
                    main.wait_event(ev)
    loop(0, 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(20, 20 + steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{parts} x {n} envs on {nstreams} stream(s){' fork-join every step' if join else ''}: {65536 * steps / dt:.4e} env-steps/s  {dt / steps * 1e6:6.1f} us per 65536-env step", flush=True)
    for e in engs:
        e.close()


run(1, 1)
run(4, 4)
run(4, 4, join=True)
run(2, 2, join=True)
run(8, 8)
run(8, 8, join=True)
run(1, 1)
