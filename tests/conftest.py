import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mpg():
    import mpgan_amd
    return mpgan_amd


@pytest.fixture(scope="session")
def gpu_ops(mpg):
    """The operator layer on cuda:0; fails (does not skip) when the HIP path is unusable."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mpgan_amd import ops, _lib
    cu, arch = _lib.device_info()
    assert arch.startswith("gfx950"), arch
    return ops


def rel_l2(a, b):
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
