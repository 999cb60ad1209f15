"""Developer experiment (GPU box): times of the network walk alone - both networks, the actor alone, the value network alone, forward (no hidden stores)
and backward - for the library named by TFP_LIB (the product build, or a WALK_DBG timing variant of csrc/ppo_mlp_walk.hip):
    TFP_LIB=leibnizgym_amd/csrc/variants/libppo_walk_dbg1.so python3 tools/experiments/walk_decompose.py [M]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from leibnizgym_amd import ppo_kernels as pk
from walk_bench_util import t_us, net

dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
torch.manual_seed(0)
la, lc = net([41, 400, 200, 100, 9]), net([113, 400, 200, 100, 1])
xa, xc, gya, gyc = torch.randn(M, 41, device=dev), torch.randn(M, 113, device=dev), torch.randn(M, 9, device=dev), torch.randn(M, 1, device=dev)
ya, yc = pk.mlp_walk_forward([(xa, la), (xc, lc)])
row = [os.path.basename(pk.library_path())]
for name, nets_f, nets_b in (("both", [(xa, la), (xc, lc)], [(gya, ya, la), (gyc, yc, lc)]), ("actor", [(xa, la)], [(gya, ya, la)]), ("value", [(xc, lc)], [(gyc, yc, lc)])):
    f0 = t_us(lambda: pk.mlp_walk_forward(nets_f, store_hidden=False))
    f1 = t_us(lambda: pk.mlp_walk_forward(nets_f))
    b = t_us(lambda: pk.mlp_walk_backward(nets_b))
    row.append(f"{name}: fwd {f0:.1f} / with stores {f1:.1f} / bwd {b:.1f} us")
print(" | ".join(row), flush=True)
