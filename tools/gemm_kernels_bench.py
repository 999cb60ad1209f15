"""Developer tool (GPU box): per-kernel times of the trainer's fp32 MFMA GEMMs against torch (rocBLAS) on the MLP shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leibnizgym_amd import ppo_kernels as pk
if os.environ.get("PPO_LIB"):                                  # a developer build of csrc/ppo_kernels.hip (compile-flag experiments)
    pk.library_path = lambda: os.path.abspath(os.environ["PPO_LIB"])
dev = "cuda:0"
def t_us(f, n=20, reps=5):
    """device time per call: n calls captured in one HIP graph (no host time between the launches), replayed"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
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
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
CH = int(os.environ.get("CHUNK", "512"))
tot = [0.0, 0.0]
SHAPES = ((41, 400), (113, 400), (400, 200), (200, 100), (100, 9), (100, 1))
if os.environ.get("SHAPES"): SHAPES = tuple(tuple(int(v) for v in sh.split("x")) for sh in os.environ["SHAPES"].split(","))   # e.g. SHAPES=400x200
for K, N in SHAPES:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    gy = torch.randn(M, N, device=dev); y = torch.randn(M, N, device=dev)
    fl = 2.0 * M * N * K * 1e-6
    # the trainer's flow (ppo_kernels.mlp_backward): gy arrives as dZ of this layer (the product above multiplied it by elu'(y) where it was stored), the
    # input gradient leaves as dZ of the layer below (times elu' of this layer's input x, itself an ELU output); OPERAND=1: the former flow, the factor
    # elu'(y) formed in the operand loads of both products
    eg = lambda t: torch.where(t > 0, 1.0, t + 1.0)           # noqa: E731
    if os.environ.get("OPERAND"):
        r = [("fwd ", t_us(lambda: pk.linear_fwd(x, w, b, 1)), t_us(lambda: torch.nn.functional.elu(torch.addmm(b, x, w.t())))),
             ("dX  ", t_us(lambda: pk.gemm_nn(gy, w, y)), t_us(lambda: (gy * eg(y)) @ w)),
             ("dW+b", t_us(lambda: pk.gemm_tn_bias(gy, x, y, chunk=CH)), t_us(lambda: ((gy * eg(y)).t() @ x, gy.sum(0))))]
    else:
        r = [("fwd ", t_us(lambda: pk.linear_fwd(x, w, b, 1)), t_us(lambda: torch.nn.functional.elu(torch.addmm(b, x, w.t())))),
             ("dX  ", t_us(lambda: pk.gemm_nn(gy, w, None, y_out=x)), t_us(lambda: (gy @ w) * eg(x))),
             ("dW+b", t_us(lambda: pk.gemm_tn_bias(gy, x, None, chunk=CH)), t_us(lambda: (gy.t() @ x, gy.sum(0))))]
    dz = gy * eg(y)
    e = lambda u, v: float((u - v).abs().max() / v.abs().max())
    gw, gb = pk.gemm_tn_bias(gy, x, y)
    print(f"   rel err fwd {e(pk.linear_fwd(x, w, b, 1), torch.nn.functional.elu(torch.addmm(b, x, w.t()))):.1e} dX {e(pk.gemm_nn(gy, w, y), dz @ w):.1e} "
          f"dW {e(gw, dz.t() @ x):.1e} db {e(gb, dz.sum(0)):.1e}")
    for name, a, c in r:
        print(f"K={K:4d} N={N:4d} {name}  mfma {a:7.1f} us ({fl / a:6.1f} TF)   torch {c:7.1f} us")
        tot[0] += a; tot[1] += c
print(f"sum over the six layers: mfma {tot[0]:.1f} us, torch {tot[1]:.1f} us")
