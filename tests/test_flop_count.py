"""The exact fp32 operation count of one env-step ("to be replaced by an exact count from the CPU restatement's
instrumented build").  oracle/tf_flops.h compiles the oracle as C++ with every `float` a wrapper that performs the same IEEE operation
and counts it; this test holds that build to the ordinary oracle bit for bit and to the figure bench.py reports.

    python tests/test_flop_count.py        # prints the table kept in profiles/ (r3_l_flops.txt)
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import bench  # noqa: E402
from leibnizgym_amd._capi import TfLib  # noqa: E402
from leibnizgym_amd.engine import TrifingerEngine, make_config  # noqa: E402
from oracle_util import ORACLE_DIR, load_oracle  # noqa: E402

NAMES = ("add", "mul", "fma", "div", "sqrt", "cmp", "cvt")
N_ENVS, N_STEPS = 256, 1000


def counting_library():
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "flops"], stdout=subprocess.DEVNULL)
    lib = TfLib(os.path.join(ORACLE_DIR, "_build", "libtrifinger_oracle_flops.so"))
    lib.dll.tf_flop_counts.argtypes = [C.POINTER(C.c_uint64)]
    lib.dll.tf_flop_reset.argtypes = []
    return lib


def count(lib, asym, dr=False, n=N_ENVS, steps=N_STEPS, counting=True):
    """per env-step operation counts of `steps` steps of the bench workload with random actions (resets inside the steps included,
    the initial reset excluded), and the final state"""
    eng = TrifingerEngine(make_config(lib, n, seed=7, **bench.workload_kwargs(asym, dr=dr)), device="cpu", lib=lib)
    g = torch.Generator().manual_seed(7)
    eng.reset()
    if counting:
        lib.dll.tf_flop_reset()
    for _ in range(steps):
        eng.step(torch.rand(n, 9, generator=g) * 2 - 1)
    out = (C.c_uint64 * len(NAMES))()
    if counting:
        lib.dll.tf_flop_counts(out)
    state = eng.state.clone()
    eng.close()
    return {k: out[i] / (n * steps) for i, k in enumerate(NAMES)}, state


def flops(c):
    """the usual convention: add, mul, div, sqrt one each, a fused multiply-add two; comparisons / min / max / abs and conversions
    are operations of the vector ALU but not floating-point operations"""
    return c["add"] + c["mul"] + 2.0 * c["fma"] + c["div"] + c["sqrt"]


def test_counting_build_is_the_oracle_and_bench_reports_its_count():
    lib = counting_library()
    for asym in (True, False):
        c, s_count = count(lib, asym, n=64, steps=120)
        _, s_plain = count(load_oracle(), asym, n=64, steps=120, counting=False)
        assert torch.equal(s_count.view(torch.int32), s_plain.view(torch.int32)), "the counting build must be the same arithmetic"
    c, _ = count(lib, True, steps=300)
    print("\nfp32 operations per env-step (asymmetric obs): " + "  ".join(f"{k} {v:.0f}" for k, v in c.items()) + f"  -> {flops(c):.0f} FLOP")
    assert abs(flops(c) / bench.FLOPS_PER_ENV_STEP - 1.0) < 0.03, (flops(c), bench.FLOPS_PER_ENV_STEP)


if __name__ == "__main__":
    lib = counting_library()
    print(f"exact fp32 operation count of the oracle (oracle/tf_flops.h), per env-step; {N_ENVS} envs x {N_STEPS} steps from reset, random actions, "
          "bench.py workload (difficulty 4, 2 substeps, 8 sweeps); FLOP = add + mul + div + sqrt + 2 fma")
    for label, asym, dr in (("asymmetric obs (headline)", True, False), ("symmetric obs", False, False), ("asymmetric + every DR feature", True, True)):
        c, _ = count(lib, asym, dr)
        print(f"{label:32s} " + "  ".join(f"{k} {v:8.1f}" for k, v in c.items()) + f"   FLOP {flops(c):9.1f}   all counted operations {sum(c.values()):9.1f}")
