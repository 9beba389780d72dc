import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cores():
    """the cores this process may really use: its affinity mask cut down to the container's CPU quota (cgroup cpu.max).
    The GPU boxes show 256 CPUs behind a 16-core quota: the OpenMP oracle on 256 threads burns the quota on spinning
    barriers and everything in the container -- single-threaded code too -- is throttled (a 3-minute suite took 19)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


# before libgomp is loaded by the oracle's OpenMP build
os.environ.setdefault("OMP_NUM_THREADS", str(_usable_cores()))


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
    # the library under test must have been built from THIS tree with the default flags (built .so files travel to the GPU
    # box with the snapshot; a stale or experimental one would be tested in good faith otherwise)
    import importlib.util
    spec = importlib.util.spec_from_file_location("hsk_build_id", os.path.join(ROOT, "housescan_amd", "csrc", "build_id.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    have = housescan_amd._lib.load().hsk_build_id().decode()
    want = mod.build_id()
    assert have == want, (f"housescan_amd/libhskinfu.so was built from other sources or flags (library {have}, tree {want}): "
                          "run `python -c 'import __graft_entry__ as g; g.build()'`")
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
