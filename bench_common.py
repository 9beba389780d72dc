"""What bench.py (the single-GPU line) and bench_launcher.py (the N > 1 launcher and its workers) share."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
W, H = 640, 480
SLAB_FORMS = ("rccl", "rccl_icp_allreduce", "direct")   # the forms of the N > 1 z-slab path (bench_launcher.py)


def make_frames(hsk, first, count):
    poses = [hsk.synth_pose(k) for k in range(first, first + count)]
    return poses, [hsk.synth_depth(p) for p in poses]


def check_build(args):
    """The measured library must be the default build of THIS tree: "+exp" marks other compiler flags (timing experiments,
    some of which give wrong results by construction), a different hash a stale .so.  tests/conftest.py refuses both too."""
    from housescan_amd import _lib
    from housescan_amd.csrc import build_id as tree_id
    have, want = _lib.load().hsk_build_id().decode(), tree_id.build_id()
    if have != want and not args.allow_exp:
        raise SystemExit("bench.py: housescan_amd/libhskinfu.so is build %s, the tree is %s -- rebuild with "
                         "`python -c 'import __graft_entry__ as g; g.build()'` (or pass --allow-exp for a timing experiment)" % (have, want))
    return have


def emit(out, have=None, want=None):
    """the JSON line is the LAST thing on stdout: RCCL's version banner sits in the C library's buffer until then"""
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(json.dumps(out))
    sys.stdout.flush()
