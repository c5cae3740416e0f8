import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("distancetransform-depthcompletion_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_op(pkg):
    """The HIP operator on cuda:0.  Fails (does not skip) when the extension is missing."""
    import torch

    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    pkg._lib.load()
    return pkg.device.DtFill(device="cuda:0")
