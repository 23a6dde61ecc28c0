import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The ctypes binding; gpu tests fail loudly if the extension or the device is missing."""
    from csparse3_amd import csc_hip
    csc_hip.lib()
    return csc_hip


@pytest.fixture(scope="session")
def gpu(hip):
    # torch brings its own copy of the HIP runtime; the tests that use torch tensors for device buffers need it
    # initialised in this process too, and doing that first keeps the order independent of test selection
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    if hip.device_count() < 1:
        pytest.fail("no HIP device visible: gpu-marked tests must run on the GPU box")
    return hip


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle
