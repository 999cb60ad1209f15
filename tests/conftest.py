import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_util import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def hip():
    """The product library on a GPU box.  Fails (never skips silently) if it is not built."""
    import torch
    from leibnizgym_amd import _capi
    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    return _capi.load_hip_library()


BACKENDS = [pytest.param("oracle", id="oracle-cpu"), pytest.param("hip", id="hip-gpu", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def backend(request):
    """(library, device) for tests of the host-side layers that must hold on both sides: the CPU oracle injected through
    the `lib=` test hook (CPU suite) and the HIP product library on cuda:0 (`-m gpu`)."""
    if request.param == "oracle":
        from oracle_util import load_oracle
        return load_oracle(), "cpu"
    import torch
    from leibnizgym_amd import _capi
    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    return _capi.load_hip_library(), "cuda:0"
