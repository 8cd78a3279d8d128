import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def rel_err(a, b):
    """max |a-b| / max |b| : the 1e-3 'rel-tol fp32' gate of BASELINE.json is applied on this scale-normalised error."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


@pytest.fixture(scope="session")
def noise_tape():
    import torch

    class Tape:
        def __init__(self, seed):
            self.g = torch.Generator().manual_seed(int(seed))

        def __call__(self, shape):
            return torch.randn(tuple(shape), generator=self.g)

    return Tape
