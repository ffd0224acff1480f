import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ddsp-svc-official_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _lib_built():
    return os.path.exists(os.path.join(PKG, "hipddsp", "libddsp_amd.so"))


@pytest.fixture(scope="session")
def lib_path():
    """Path of the in-tree shared library; builds it with hipcc when it is missing (cross-compiles on CPU)."""
    if not _lib_built():
        from hipddsp.build import build_lib
        build_lib(verbose=False)
    return os.path.join(PKG, "hipddsp", "libddsp_amd.so")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a gpu-marked test ran without a HIP device (there is no CPU fallback to hide behind)")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def ctx(dev, lib_path):
    import hipddsp
    return hipddsp.context_for(dev)


def rms(x):
    import torch
    return float(torch.sqrt(torch.mean(x.double() ** 2)))
