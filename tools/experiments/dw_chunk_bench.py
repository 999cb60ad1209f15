"""Developer experiment (GPU box): the grouped weight-gradient launch of a minibatch step (8 products, 8192 rows) + the chunk sums, over row-chunk sizes."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from leibnizgym_amd import ppo_kernels as pk
dev = "cuda:0"
def t_us(f, n=10, reps=5):
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
M = 8192
shapes = [(400, 41), (200, 400), (100, 200), (9, 100), (400, 113), (200, 400), (100, 200), (1, 100)]
as_ = [torch.randn(M, n1, device=dev) for n1, _ in shapes]; bs = [torch.randn(M, n2, device=dev) for _, n2 in shapes]
outs = [(torch.empty(n1, n2, device=dev), torch.empty(n1, device=dev)) for n1, n2 in shapes]
for chunk in (128, 256, 512, 1024, 2048):
    def f():
        assert pk.gemm_tn_bias_group(as_, bs, None, outs, chunk=chunk)
        pk.flush_partial_sums()
    print(f"chunk {chunk:5d}: products + sums {t_us(f):7.1f} us")
