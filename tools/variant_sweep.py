"""Developer tool (GPU box): fused-step time (HIP events) of the headline workload for the two instantiations of the step kernel
(tf_set_kernel_variant) over a range of population sizes, with a state checksum (identical between the variants: same arithmetic).
    python tools/variant_sweep.py [lib.so] [N ...]      env: SETTLE, SOLVER=iterations,inner, NO_TIMEOUT=1, FF_MIDDLE=0|1, ASYM=0|1 (default: both)"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from leibnizgym_amd.engine import TrifingerEngine, make_config
from leibnizgym_amd import _capi

args = sys.argv[1:]
lib = _capi.load_hip_library()
if args and args[0].endswith(".so"):
    lib = _capi.TfLib(os.path.abspath(args.pop(0)))
sizes = [int(a) for a in args] or [8192, 16384, 32768, 65536]
SETTLE = int(os.environ.get("SETTLE", "400"))
SOLVER = [int(x) for x in os.environ.get("SOLVER", "8,1").split(",")]      # SOLVER=16,1 / SOLVER=8,2: solver_iterations, solver_inner
for asym in ([bool(int(os.environ["ASYM"]))] if os.environ.get("ASYM") else (True, False)):
    for n in sizes:
        for variant in ("narrow", "wide", "wide_helpers"):
            kw = bench.workload_kwargs(asym)
            kw.update(solver_iterations=SOLVER[0], solver_inner=SOLVER[1])
            if os.environ.get("FF_MIDDLE"):                   # FF_MIDDLE=0: the fast contact set of API <= 7 (TfModel.ff_middle_pairs = 0; default model: 1)
                kw["model"] = lib.default_model()
                kw["model"].ff_middle_pairs = int(os.environ["FF_MIDDLE"])
            if os.environ.get("NO_TIMEOUT"):                  # diagnostic: no time-out resets inside the timed steps (what the reset path of a launch costs)
                kw.update(episode_length=0)
            eng = TrifingerEngine(make_config(lib, n, seed=7, **kw), device="cuda:0", lib=lib)
            eng.kernel_variant = variant
            eng.reset()
            eng.steps.copy_(torch.randint(0, 750, (n,), device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(11)))      # spread the time-outs, as bench.py does
            for k in range(SETTLE):
                eng.step_random()
            best = 1e9
            for rep in range(3):
                eng.enable_kernel_timing(64, window=8)
                for k in range(512):
                    eng.step_random()
                torch.cuda.synchronize()
                ms, cnt = eng.kernel_time_ms()
                best = min(best, ms / cnt * 1e3)
            chk = float(eng.state.double().abs().sum())
            print(f"N={n:6d} asym={int(asym)} {variant:12s}: k_env {best:7.2f} us   {n / best:8.1f} env-steps/us   state checksum {chk:.9e}", flush=True)
            eng.close()
