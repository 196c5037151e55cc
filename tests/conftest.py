import os
import sys

import pytest

# PyTorch's wheel carries its own ROCm runtime libraries.  A process that uses both PyTorch and
# libraytracer_amd.so must load PyTorch's first, so that one ROCr instance serves both (the second
# instance to open the GPU finds none); the tests that pass torch tensors to the library rely on it.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/liboracle.so), built on first use."""
    from oracle import oracle as orc

    orc.lib()
    return orc
