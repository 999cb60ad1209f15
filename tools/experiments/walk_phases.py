"""Developer experiment (GPU box): where a workgroup of the network walk spends its cycles (WALK_TIMING build of csrc/ppo_mlp_walk.hip: s_memtime stamps
of one workgroup, per wavefront):  python3 tools/experiments/walk_phases.py   (builds nothing: make the variant first, see tools/README.md)"""
import os, sys, ctypes as C
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
os.environ["TFP_LIB"] = os.path.join(REPO, "leibnizgym_amd", "csrc", "variants", "libppo_walk_timing.so")      # picked up by walk_bench_util
sys.path.insert(0, REPO); sys.path.insert(0, os.path.dirname(__file__))
import torch
from leibnizgym_amd import ppo_kernels as pk
from walk_bench_util import net

dev = "cuda:0"
M = 8192
la, lc = net([41, 400, 200, 100, 9]), net([113, 400, 200, 100, 1])
xa, xc, gya, gyc = torch.randn(M, 41, device=dev), torch.randn(M, 113, device=dev), torch.randn(M, 9, device=dev), torch.randn(M, 1, device=dev)
lib = pk.load()
buf = (C.c_ulonglong * 128)()


def show(tag, nsteps):
    torch.cuda.synchronize()
    assert lib.tfp_walk_debug_read(buf) == 0
    for w in range(4):
        t = [buf[w * 32 + i] for i in range(2 + 3 * nsteps)]
        d = [t[i + 1] - t[i] for i in range(len(t) - 1)]
        names = ["input"] + [f"L{s} {p}" for s in range(nsteps) for p in ("K loop", "epilogue", "barrier")]
        print(f"{tag} wave {w}: total {t[-1] - t[0]} cycles | " + ", ".join(f"{n} {v}" for n, v in zip(names, d)), flush=True)


for _ in range(3):
    ya, yc = pk.mlp_walk_forward([(xa, la), (xc, lc)])
show("forward (workgroup 200 = value network)", 4)
for _ in range(3):
    pk.mlp_walk_backward([(gya, ya, la), (gyc, yc, lc)])
show("backward", 3)
