"""Test-side access to the CPU oracle (oracle/ is test infrastructure; never imported by the package)."""
import ctypes as C
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libtrifinger_oracle.so")
GOLDEN_DIR = os.path.join(REPO, "tests", "golden")

_lib = None


def build_oracle():
    if os.environ.get("TF_ORACLE_SO"):          # another build of the same source (tests/test_oracle_sanitizer.py: the -fsanitize=address,undefined one)
        return os.environ["TF_ORACLE_SO"]
    src = os.path.join(ORACLE_DIR, "tf_oracle.c")
    hdr = os.path.join(REPO, "include", "trifinger.h")
    if (not os.path.isfile(ORACLE_SO)
            or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


def load_oracle():
    global _lib
    if _lib is None:
        from leibnizgym_amd._capi import TfLib
        _lib = TfLib(build_oracle())
        d = _lib.dll
        fp = C.POINTER(C.c_float)
        d.tfo_sincos.argtypes = [fp, fp, fp, C.c_int32]
        for n in ("tfo_exp", "tfo_asin", "tfo_log"):
            getattr(d, n).argtypes = [fp, fp, C.c_int32]
        d.tfo_philox_raw.argtypes = [C.POINTER(C.c_uint32)] * 3
    return _lib


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def vptr(a):
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)
