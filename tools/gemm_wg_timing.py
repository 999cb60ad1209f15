"""Developer tool (GPU box): start / end wall time of every workgroup of one forward GEMM (needs a -DGEMM_TIMING build of
csrc/ppo_kernels.hip:  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DGEMM_TIMING -shared -o /tmp/ppo_timing.so ppo_kernels.hip;
python tools/gemm_wg_timing.py /tmp/ppo_timing.so)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from leibnizgym_amd import ppo_kernels as pk
if len(sys.argv) > 1: pk.library_path = lambda: sys.argv[1]
dev="cuda:0"
M,K,N=8192,400,200
x=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev); b=torch.randn(N,device=dev)
for _ in range(5): pk.linear_fwd(x,w,b,1)
torch.cuda.synchronize()
out=(C.c_ulonglong*4096)(); pk.load().tfp_debug_read_wg(out)
a=np.array(list(out),dtype=np.int64).reshape(2048,2)[:512]
t0=a[:,0].min(); st=(a[:,0]-t0)*10e-3; en=(a[:,1]-t0)*10e-3   # us
print("start us: min %.2f median %.2f max %.2f" % (st.min(), np.median(st), st.max()))
print("end   us: min %.2f median %.2f max %.2f" % (en.min(), np.median(en), en.max()))
print("duration us: min %.2f median %.2f max %.2f" % ((en-st).min(), np.median(en-st), (en-st).max()))
print("histogram of start times (us):", np.histogram(st, bins=8)[0], np.histogram(st, bins=8)[1].round(1))
dbg = (C.c_uint * 64)(); pk.load().tfp_debug_read(dbg)
d = np.array(list(dbg), dtype=np.int64).reshape(8, 8)
for wv in range(8):
    print("wave %d (%s): total %6d cycles, in barriers %6d, tiles %d" % (wv, "multiplies" if wv < 4 else "stages", d[wv, 0], d[wv, 1], d[wv, 2]))
