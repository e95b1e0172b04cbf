import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def load_golden(name):
    """Golden fixtures are plain arrays (np.load, allow_pickle=False)."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="session")
def c_oracle():
    from oracle import c_oracle as co
    co.lib()
    return co


@pytest.fixture(scope="session")
def np_oracle():
    from oracle import qps_oracle_np
    return qps_oracle_np


@pytest.fixture(scope="session")
def qps():
    """The product package with libqps_hip.so built (no GPU needed to build or load)."""
    import quadraticprogramsolver_amd as q
    from quadraticprogramsolver_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    _lib.lib()
    return q


@pytest.fixture(scope="session")
def gpu(qps):
    from quadraticprogramsolver_amd import _lib
    if _lib.lib().qps_device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (there is no CPU fallback to hide behind)")
    return qps
