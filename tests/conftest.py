import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_PARENT = os.path.join(ROOT, "incagg-gnn_amd")
for p in (ROOT, PKG_PARENT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_relabel():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "relabel_golden.npz"))
