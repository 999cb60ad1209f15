"""helpers of walk_bench.py / walk_decompose.py"""
import os

import torch

from leibnizgym_amd import ppo_kernels as _pk

dev = "cuda:0"
if os.environ.get("TFP_LIB"):          # a developer build of the trainer's kernels (timing variants): the package itself takes no environment override
    _pk.library_path = lambda: os.environ["TFP_LIB"]


def t_us(f, n=10, reps=5):
    """time per call of f inside a HIP graph of n calls (launch overhead of the host excluded), microseconds"""
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): f()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / (n * reps) * 1e3


def net(dims):
    n = len(dims) - 1
    return [(torch.randn(dims[i + 1], dims[i], device=dev) * dims[i] ** -0.5, torch.randn(dims[i + 1], device=dev) * 0.1, 1 if i < n - 1 else 0,
             (torch.zeros(dims[i + 1], dims[i], device=dev), torch.zeros(dims[i + 1], device=dev))) for i in range(n)]
