"""The C-ABI libraries load (no GPU needed) and export every symbol include/trifinger.h declares; the ctypes
mirrors of the structs have the C compiler's sizes; both libraries ship the same default model."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from leibnizgym_amd import _capi as capi
from oracle_util import REPO

HEADER = os.path.join(REPO, "include", "trifinger.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tf_[a-z0-9_]+)\s*\(", src)))


def hip_lib():
    path = capi.hip_library_path()
    if not os.path.isfile(path):
        subprocess.check_call(["make", "-j8", "-C", os.path.dirname(path), "-s"])
    return capi.TfLib(path)


def test_header_and_binding_agree():
    names = declared_functions()
    assert len(names) >= 25
    assert sorted(capi.SYMBOLS) == names, set(names) ^ set(capi.SYMBOLS)


def test_hip_library_exports_every_symbol():
    lib = hip_lib()                      # TfLib raises if a symbol is missing
    assert lib.backend == "hip-gfx950"
    assert lib.tf_api_version() == capi.TF_API_VERSION
    assert lib.tf_action_dim(0) == 9 and lib.tf_action_dim(2) == 18 and lib.tf_action_dim(7) == capi.TF_ERR_COMMAND_MODE
    assert lib.tf_scratch_floats(65536) == 1024 * 16


def test_oracle_library_exports_every_symbol(oracle):
    assert oracle.backend == "oracle-c"


def test_struct_sizes_match_the_c_compiler():
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "trifinger.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu\n", sizeof(TfRewardTerm), sizeof(TfModel), sizeof(TfConfig), sizeof(TfBuffers),
           offsetof(TfConfig, model), offsetof(TfConfig, reward));
    return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "s.c"), os.path.join(d, "s")
        open(src, "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), "-o", exe, src])
        out = subprocess.check_output([exe]).decode().split()
    sizes = [int(x) for x in out]
    assert sizes == [C.sizeof(capi.TfRewardTerm), C.sizeof(capi.TfModel), C.sizeof(capi.TfConfig),
                     C.sizeof(capi.TfBuffers), capi.TfConfig.model.offset, capi.TfConfig.reward.offset]


def test_kernel_variant_names_match_the_header(oracle):
    """TrifingerEngine.KERNEL_VARIANTS against the TF_KERNEL_* enumerators of include/trifinger.h; the oracle accepts every one of them (and ignores it),
    rejects what lies outside"""
    from leibnizgym_amd.engine import TrifingerEngine, make_config
    text = open(os.path.join(REPO, "include", "trifinger.h")).read()
    enum = re.search(r"enum \{ (TF_KERNEL_AUTO[^}]*) \};", text).group(1)
    vals = {k.strip().split(" = ")[0][len("TF_KERNEL_"):].lower(): int(k.strip().split(" = ")[1]) for k in enum.split(",")}
    assert vals == TrifingerEngine.KERNEL_VARIANTS, (vals, TrifingerEngine.KERNEL_VARIANTS)
    assert int(re.search(r"#define TF_HELPERS_MAX_ENVS (\d+)", text).group(1)) * 2 == int(re.search(r"#define TF_WIDE_MAX_ENVS (\d+)", text).group(1))
    eng = TrifingerEngine(make_config(oracle, 8), device="cpu", lib=oracle)
    for name in vals:
        eng.kernel_variant = name
    assert oracle.tf_set_kernel_variant(eng._handle, max(vals.values()) + 1) != 0
    eng.close()


def test_default_models_are_identical(oracle):
    a, b = hip_lib().default_model(), oracle.default_model()
    assert bytes(a) == bytes(b)
    assert abs(a.cube_mass - 291.3 * 0.065 ** 3) < 1e-7          # cube_multicolor_rrc.urdf: density x volume
    assert abs(a.link_mass[2] - 0.052) < 1e-7                     # lower link + tip link


def test_product_fails_loudly_without_a_gpu_device():
    from leibnizgym_amd.engine import TrifingerEngine, make_config
    lib = hip_lib()
    with pytest.raises(RuntimeError, match="no CPU path"):
        TrifingerEngine(make_config(lib, 4), device="cpu")
    missing = os.path.join(REPO, "leibnizgym_amd", "csrc", "does_not_exist.so")
    with pytest.raises(capi.TfLibraryError, match="no fallback"):
        capi.TfLib(missing)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, "leibnizgym_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                for line in text.splitlines():
                    code = line.split("#")[0].split("//")[0]
                    assert "oracle_util" not in code and "tf_oracle" not in code and "libtrifinger_oracle" not in code, \
                        (f, line)


def test_create_rejects_bad_configs(oracle):
    """Empty / invalid inputs at the C boundary: status codes, not crashes."""
    import ctypes as C
    from leibnizgym_amd.engine import make_config
    for lib in (oracle, hip_lib()):
        h = C.c_void_p()
        cfg = make_config(lib, 4)
        cfg.num_envs = 0                                   # empty batch
        assert lib.tf_create(C.byref(cfg), C.byref(h)) == capi.TF_ERR_INVALID_ARG
        cfg.num_envs = 2097152 + 1                         # above TF_MAX_ENVS (32-bit offsets into the state block)
        assert lib.tf_create(C.byref(cfg), C.byref(h)) == capi.TF_ERR_INVALID_ARG
        cfg = make_config(lib, 4); cfg.command_mode = 9
        assert lib.tf_create(C.byref(cfg), C.byref(h)) == capi.TF_ERR_COMMAND_MODE
        cfg = make_config(lib, 4); cfg.task_difficulty = 0
        assert lib.tf_create(C.byref(cfg), C.byref(h)) == capi.TF_ERR_DIFFICULTY
        cfg = make_config(lib, 4); cfg.robot_reset_type = 5
        assert lib.tf_create(C.byref(cfg), C.byref(h)) == capi.TF_ERR_ROBOT_RESET
        cfg = make_config(lib, 4); cfg.object_reset_type = -1
        assert lib.tf_create(C.byref(cfg), C.byref(h)) == capi.TF_ERR_OBJECT_RESET
        for bad_p in (0, 17, -2):                          # built: 1..16 and TF_NORM_INF (-1)
            cfg = make_config(lib, 4); cfg.finger_reach_norm_p = bad_p
            assert lib.tf_create(C.byref(cfg), C.byref(h)) == capi.TF_ERR_UNSUPPORTED
        cfg = make_config(lib, 4); cfg.model.wall_z[2] = cfg.model.wall_z[1]      # boundary knots must rise strictly (piecewise-linear profile)
        assert lib.tf_create(C.byref(cfg), C.byref(h)) == capi.TF_ERR_INVALID_ARG
        cfg = make_config(lib, 4); cfg.model.wall_r[3] = float("nan")
        assert lib.tf_create(C.byref(cfg), C.byref(h)) == capi.TF_ERR_INVALID_ARG
        cfg = make_config(lib, 4); cfg.api_version = 99
        assert lib.tf_create(C.byref(cfg), C.byref(h)) == capi.TF_ERR_INVALID_ARG
        assert lib.tf_step(None, None, None) == capi.TF_ERR_INVALID_ARG
    # unbound handle on the oracle (no device needed)
    h = C.c_void_p()
    assert oracle.tf_create(C.byref(make_config(oracle, 4)), C.byref(h)) == 0
    assert oracle.tf_step(h, C.c_void_p(1), None) == capi.TF_ERR_NOT_BOUND
    oracle.tf_destroy(h)


def test_default_model_is_the_same_in_both_libraries(oracle):
    """tf_default_model of the product and of the oracle are written twice (C++ / C): every field must agree bit for bit,
    otherwise the parity tests would compare two different physical models (no GPU needed: a host function)."""
    import ctypes as C
    hip = capi.TfLib(capi.hip_library_path())
    a, b = hip.default_model(), oracle.default_model()
    assert bytes(C.string_at(C.addressof(a), C.sizeof(a))) == bytes(C.string_at(C.addressof(b), C.sizeof(b))), \
        [n for n, _ in capi.TfModel._fields_ if bytes(getattr(a, n)) != bytes(getattr(b, n))]


def test_ppo_library_exports_every_symbol_its_header_declares():
    """include/trifinger_ppo.h <-> libtrifinger_ppo.so <-> the ctypes binding (no GPU needed: the library only has to load)."""
    from leibnizgym_amd import ppo_kernels as pk
    src = open(os.path.join(REPO, "include", "trifinger_ppo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(tfp_[a-z0-9_]+)\s*\(", src)))
    assert len(names) == 22, names
    path = pk.library_path()
    if not os.path.isfile(path):
        subprocess.check_call(["make", "-C", os.path.dirname(path), "-s"])
    lib = C.CDLL(path)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.tfp_api_version() == 3
    bound = pk.load()                                   # the binding sets argtypes for every declared entry point
    for n in names:
        assert getattr(bound, n).restype is C.c_int, n
