import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ntt-cuda_amd"), os.path.join(ROOT, "oracle"), ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def native():
    """The HIP library through ctypes; built in tree on first use (hipcc cross-compiles without a GPU)."""
    import ntt_cuda_amd
    ntt_cuda_amd.build()
    ntt_cuda_amd.lib()
    return ntt_cuda_amd


@pytest.fixture(scope="session")
def gpu(native):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (run -m gpu only on the GPU box)")
    return torch.device("cuda:0")
