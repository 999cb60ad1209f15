"""Developer tool (GPU box): the trainer's MFMA linear layer (forward + backward) against torch (rocBLAS / hipBLASLt + elementwise
launches) on the shapes of the two MLPs; run under rocprofv3 --kernel-trace --stats for kernel-level times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leibnizgym_amd import ppo_kernels as pk
dev = "cuda:0"
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for K, N, act in ((41, 400, 1), (113, 400, 1), (400, 200, 1), (200, 100, 1), (100, 9, 0), (100, 1, 0)):
    x0 = torch.randn(M, K, device=dev); w0 = torch.randn(N, K, device=dev) * K ** -0.5; b0 = torch.randn(N, device=dev)
    gy = torch.randn(M, N, device=dev)
    res = []
    for fused in (False, True):
        x, w, b = (t.clone().requires_grad_(True) for t in (x0, w0, b0))
        def run():
            for t in (x, w, b): t.grad = None
            y = pk.mfma_linear(x, w, b, act) if fused else (torch.nn.functional.elu(torch.nn.functional.linear(x, w, b)) if act else torch.nn.functional.linear(x, w, b))
            y.backward(gy)
            return y
        y = run()
        res.append((y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone(), timeit(run)))
    errs = [float((a - c).abs().max() / a.abs().max()) for a, c in zip(res[0][:4], res[1][:4])]
    print(f"K={K:4d} N={N:4d} act={act}  fwd+bwd torch {res[0][4]:6.1f} us  mfma {res[1][4]:6.1f} us   rel err y/dx/dw/db {errs[0]:.1e} {errs[1]:.1e} {errs[2]:.1e} {errs[3]:.1e}")
