"""Developer experiment: per-layer error of the network walk against float64 torch, forward with and without hidden stores, then backward"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))); sys.path.insert(0, os.path.dirname(__file__))
import torch
from leibnizgym_amd import ppo_kernels as pk
from walk_bench_util import net
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
la, lc = net([41, 400, 200, 100, 9]), net([113, 400, 200, 100, 1])
xa, xc, gya, gyc = torch.randn(M, 41, device=dev), torch.randn(M, 113, device=dev), torch.randn(M, 9, device=dev), torch.randn(M, 1, device=dev)
def ref(x, layers):
    h, ys = x.double(), []
    for w, b, act, _ in layers:
        h = torch.nn.functional.linear(h, w.double(), b.double()); h = torch.nn.functional.elu(h) if act else h; ys.append(h)
    return ys
for store in (False, True):
    outs = pk.mlp_walk_forward([(xa, la), (xc, lc)], store_hidden=store)
    torch.cuda.synchronize()
    for name, ys, r in (("actor", outs[0], ref(xa, la)), ("value", outs[1], ref(xc, lc))):
        for l, (y, y64) in enumerate(zip(ys, r)):
            if y is None: continue
            err = (y.double() - y64).abs()
            bad = (err > 1e-4).nonzero()
            if bad.shape[0] and y.shape[1] % 4 == 0:
                cpr = y.shape[1] // 4
                ids = sorted(set(((bad[:, 0] % 32) * cpr + bad[:, 1] // 4).tolist()))
                print("   bad chunk ids (in block):", ids[:80], "blocks", sorted(set((bad[:, 0] // 32).tolist())))
            print(f"store_hidden={store} {name} layer {l}: max err {float(err.max()):.3e}, bad {bad.shape[0]}" + (f", first bad {bad[0].tolist()} last bad {bad[-1].tolist()} rows {sorted(set(bad[:, 0].tolist()))[:8]} cols {sorted(set(bad[:, 1].tolist()))[:12]}" if bad.shape[0] else ""), flush=True)
if len(sys.argv) > 2:
    dz = pk.mlp_walk_backward([(gya, outs[0], la), (gyc, outs[1], lc)])
    torch.cuda.synchronize()
    print("backward ran", [tuple(d.shape) for d in dz[0]])
