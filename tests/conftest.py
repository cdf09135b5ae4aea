import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)          # tests/nifti_util.py


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from frog_amd import _abi
        return _abi.hip_lib().frog_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a device must fail loudly, not skip: the driver
    # selects the marker itself, so nothing is deselected here.
    pass


@pytest.fixture(scope="session")
def small_pairs():
    """6 images x 3000 points, ~1500 pairs per image pair (all pairs linked)."""
    from frog_amd.pairs import Pairs
    return Pairs.synthetic(6, 3000, 1500, seed=7)


@pytest.fixture(scope="session")
def tiny_pairs():
    """4 images x 600 points, ~300 pairs per image pair."""
    from frog_amd.pairs import Pairs
    return Pairs.synthetic(4, 600, 300, seed=3)
