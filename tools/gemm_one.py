"""Developer tool (GPU box): one GEMM of the trainer in a loop (for rocprofv3 --pmc): gemm_one.py fwd|dx|dw M K N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leibnizgym_amd import ppo_kernels as pk
kind, M, K, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = "cuda:0"
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
gy = torch.randn(M, N, device=dev); y = torch.randn(M, N, device=dev)
for _ in range(30):
    if kind == "fwd": pk.linear_fwd(x, w, b, 1)
    elif kind == "dx": pk.gemm_nn(gy, w, y)
    else: pk.gemm_tn_bias(gy, x, y)
torch.cuda.synchronize()
