import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if needed) and load libaha_hip.so; a missing library is a hard failure, never a skip."""
    from aha_amd import build as _build
    _build.build()
    from aha_amd import _lib
    return _lib.lib()


@pytest.fixture(scope="session")
def gpu(hip_lib):
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch.device("cuda:0")
