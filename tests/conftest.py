import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def hsk():
    import housescan_amd
    return housescan_amd


@pytest.fixture(scope="session")
def synth_frames(hsk):
    """First frames of the deterministic synthetic stream with ground-truth poses."""
    cache = {}

    def get(k):
        if k not in cache:
            p = hsk.synth_pose(k)
            cache[k] = (p, hsk.synth_depth(p))
        return cache[k]

    return get
