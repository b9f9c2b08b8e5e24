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
    # The oracle runs on the GPU box's host cores.  With torch's default of one intra-op thread per logical CPU (256 there) its skinny
    # matmuls thrash -- measured in bench.py's cpu_baseline leg: 1.9 s per layer at 256 threads vs 26 ms at 16-64 -- so cap the pool.
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, 32)))
    return torch.device("cuda:0")
