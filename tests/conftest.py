import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def alice():
    return golden("alice29.txt")


@pytest.fixture(scope="session")
def gpu():
    """torch + a visible device + the HIP library; the gpu-marked tests fail loudly without them."""
    import torch

    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    import compu_amd

    assert compu_amd.lib().chip_device_count() > 0
    return torch
