"""Developer experiment (GPU box): the eight weight-gradient products of a minibatch step - the direct kernel (csrc/ppo_dw_direct.hip) against the grouped
LDS-staged one, and the slab sums on their own; per call inside a HIP graph."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))); sys.path.insert(0, os.path.dirname(__file__))
import torch
from leibnizgym_amd import ppo_kernels as pk
from walk_bench_util import t_us
dev = "cuda:0"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
shapes = [(400, 41), (200, 400), (100, 200), (9, 100), (400, 113), (200, 400), (100, 200), (1, 100)]
as_ = [torch.randn(rows, n1, device=dev) for n1, _ in shapes]
bs = [torch.randn(rows, n2, device=dev) for _, n2 in shapes]
outs = [(torch.zeros(n1, n2, device=dev), torch.zeros(n1, device=dev)) for n1, n2 in shapes]
flops = 2.0 * rows * sum(n1 * (n2 + 1) for n1, n2 in shapes)
def direct():
    pk.gemm_tn_bias_direct(as_, bs, outs); pk.discard_partial_sums()
def grouped():
    pk.gemm_tn_bias_group(as_, bs, None, outs); pk.discard_partial_sums()
def direct_sum():
    pk.gemm_tn_bias_direct(as_, bs, outs); pk.flush_partial_sums()
def grouped_sum():
    pk.gemm_tn_bias_group(as_, bs, None, outs); pk.flush_partial_sums()
for name, f in (("direct products", direct), ("grouped products", grouped), ("direct + sums", direct_sum), ("grouped + sums", grouped_sum)):
    t = t_us(f)
    print(f"{name}: {t:.1f} us = {flops / t * 1e-6:.1f} TFLOP/s", flush=True)
for sel in ([1], [1, 5], [0, 4], [2, 6], [3, 7]):
    a2, b2, o2 = [as_[i] for i in sel], [bs[i] for i in sel], [outs[i] for i in sel]
    def f():
        pk.gemm_tn_bias_direct(a2, b2, o2); pk.discard_partial_sums()
    t = t_us(f)
    fl = 2.0 * rows * sum(shapes[i][0] * (shapes[i][1] + 1) for i in sel)
    print(f"direct, problems {[shapes[i] for i in sel]}: {t:.1f} us = {fl / t * 1e-6:.1f} TFLOP/s", flush=True)
