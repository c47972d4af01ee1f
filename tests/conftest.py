import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Contexts on cuda:0 through the C ABI; the HIP library must already be built in-tree."""
    from raytracer_amd import abi
    made = []

    def make(width, height):
        c = abi.Context(width, height, 0)
        made.append(c)
        return c

    yield make
    for c in made:
        c.close()
