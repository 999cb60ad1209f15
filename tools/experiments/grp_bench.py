"""Developer experiment (GPU box): a pair of forward products as ONE grouped launch against two single launches, straight through ctypes
(works with any build of csrc/ppo_kernels.hip):  PPO_LIB=path python3 tools/experiments/grp_bench.py"""
import os, sys, ctypes as C
import torch
lib = C.CDLL(os.path.abspath(os.environ.get("PPO_LIB", "leibnizgym_amd/csrc/libtrifinger_ppo.so")))
lib.tfp_linear_fwd.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p]
lib.tfp_linear_fwd_group.argtypes = [C.c_void_p] * 7 + [C.c_int32, C.c_int32, C.c_void_p]
dev = "cuda:0"
def t_us(f, n=20, reps=5):
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
vp = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
ip = lambda vs: (C.c_int32 * len(vs))(*vs)
for K, N in ((400, 200), (200, 100), (41, 400), (113, 400)):
    xs = [torch.randn(M, K, device=dev) for _ in range(2)]; ws = [torch.randn(N, K, device=dev) for _ in range(2)]; bs = [torch.randn(N, device=dev) for _ in range(2)]
    ys = [torch.empty(M, N, device=dev) for _ in range(2)]
    st = lambda: torch.cuda.current_stream().cuda_stream
    def grp():
        rc = lib.tfp_linear_fwd_group(vp(xs), vp(ws), vp(bs), vp(ys), ip([M, M]), ip([N, N]), ip([K, K]), 1, 2, st()); assert rc == 0, rc
    def two():
        for x, w, b, y in zip(xs, ws, bs, ys):
            rc = lib.tfp_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, 1, st()); assert rc == 0, rc
    print(f"fwd pair {K}->{N}: group {t_us(grp):.1f} us, two singles {t_us(two):.1f} us")
