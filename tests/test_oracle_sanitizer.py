"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers on the CPU build - GPU ASan is not available
on the pool).  `make -C oracle asan` builds oracle/_build/libtrifinger_oracle_asan.so; a child python with libasan preloaded runs the oracle-side tests
that drive every entry point of the C ABI (golden vectors, the env API contract, the NaN guard, the box object, the reset distribution) against it.
A heap overflow, a use after free, a signed overflow or a misaligned access in the restatement aborts the child."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_tests_pass_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "-s", "asan"], stdout=subprocess.DEVNULL)
    so = os.path.join(REPO, "oracle", "_build", "libtrifinger_oracle_asan.so")
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    assert os.path.isfile(so) and os.path.isfile(libasan), (so, libasan)
    env = dict(os.environ, TF_ORACLE_SO=so, LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",        # (CPython itself leaks by design: leak checking off)
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    sel = ["tests/test_golden_oracle.py", "tests/test_nan_guard.py", "tests/test_env_api.py", "tests/test_box_object.py::test_box_model_numbers", "tests/test_box_object.py::test_box_rests_on_every_face",
           "tests/test_reset_distribution.py"]
    sel = [s for s in sel if os.path.exists(os.path.join(REPO, s.split("::")[0]))]
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + sel, cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "passed" in p.stdout and "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-2000:]
