"""Developer experiment (GPU box): the network walk (csrc/ppo_mlp_walk.hip: all layers of both MLPs in one launch per direction) against the per-layer
grouped launches, at the trainer's minibatch shapes; times per call inside a HIP graph, max error against float64 torch.
    python3 tools/experiments/walk_bench.py [M]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import walk_bench_util  # noqa: F401  (honours TFP_LIB)
from leibnizgym_amd import ppo_kernels as pk

dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192


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


def net(dims):
    n = len(dims) - 1
    return [(torch.randn(dims[i + 1], dims[i], device=dev) * dims[i] ** -0.5, torch.randn(dims[i + 1], device=dev) * 0.1, 1 if i < n - 1 else 0,
             (torch.zeros(dims[i + 1], dims[i], device=dev), torch.zeros(dims[i + 1], device=dev))) for i in range(n)]


torch.manual_seed(0)
la, lc = net([41, 400, 200, 100, 9]), net([113, 400, 200, 100, 1])
xa, xc, gya, gyc = torch.randn(M, 41, device=dev), torch.randn(M, 113, device=dev), torch.randn(M, 9, device=dev), torch.randn(M, 1, device=dev)
flops_f = 2.0 * M * (41 * 400 + 113 * 400 + 2 * (400 * 200 + 200 * 100) + 100 * 10)
flops_b = 2.0 * M * (2 * (400 * 200 + 200 * 100) + 100 * 10)
out = {}
for walk, direct in ((False, False), (True, False), (True, True)):
    pk.USE_WALK, pk.USE_DIRECT_DW = walk, direct
    ya, yc = pk.mlp_forward_pair(xa, la, xc, lc)
    out[walk] = (ya, yc)
    f_train = t_us(lambda: pk.mlp_forward_pair(xa, la, xc, lc))
    f_roll = t_us(lambda: pk.mlp_forward_pair(xa, la, xc, lc, store_hidden=False)) if walk else float("nan")
    def bwd():
        pk.mlp_backward_pair(xa, ya, gya, la, xc, yc, gyc, lc); pk.flush_partial_sums()
    b_all = t_us(bwd)
    b_dx = t_us(lambda: pk.mlp_walk_backward([(gya, ya, la), (gyc, yc, lc)])) if walk else float("nan")
    print(f"{'walk' if walk else 'per-layer grouped'}{' + direct dW' if direct else ''}: forward (train) {f_train:.1f} us = {flops_f / f_train * 1e-6:.1f} TFLOP/s, forward (rollout) {f_roll:.1f} us, "
          f"backward incl. dW {b_all:.1f} us, input-gradient walk {b_dx:.1f} us = {flops_b / b_dx * 1e-6 if walk else float('nan'):.1f} TFLOP/s, "
          f"minibatch GEMM total {f_train + b_all:.1f} us", flush=True)
# error against float64
h = xa.double()
for (w, b, act, _), y in zip(la, out[True][0]):
    h = torch.nn.functional.linear(h, w.double(), b.double())
    h = torch.nn.functional.elu(h) if act else h
    print(f"  actor layer out {tuple(y.shape)}: walk max |err| {float((y.double() - h).abs().max()):.2e}", flush=True)
for a, b in zip(out[True][0] + out[True][1], out[False][0] + out[False][1]):
    print(f"  walk vs per-layer {tuple(a.shape)}: max |diff| {float((a - b).abs().max()):.2e}")
