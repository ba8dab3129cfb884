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
def elev_mask_g20():
    """mask/thk/topg of the reference's tests/test_conserv/elev_mask.cdl (see tests/golden/)."""
    d = np.load(os.path.join(ROOT, "tests", "golden", "elev_mask_g20.npz"))
    return {k: d[k] for k in ("mask", "thk", "topg")}
