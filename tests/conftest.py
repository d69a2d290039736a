import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """tests/golden/tiny.npz -- outputs of the unmodified reference (see tests/golden/make_golden.py)."""
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "tiny.npz"), allow_pickle=False))
