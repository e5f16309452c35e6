import os
import sys

import pytest

# torch (when a test needs it: streams, tensors as exchange buffers, torch.distributed) must enter the process BEFORE libvxba.so:
# it bundles its own HIP runtime under the same SONAME as /opt/rocm's, and only the first one loaded can see the GPU.
# bench.py imports it first for the same reason.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for the CPU suite
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests._oracle import Oracle
    return Oracle
