import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def gpu():
    import torch
    from livescan3d_amd import native
    if not torch.cuda.is_available() or native.device_count() <= 0:
        pytest.fail("-m gpu tests need a HIP device and libNativeUtils.so; there is no CPU fallback to test")
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)
