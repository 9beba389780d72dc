#!/usr/bin/env python3
"""The suite's two fuzz tests (tests/test_gpu_parity.py: random depth images from arbitrary poses into cubic / non-cubic / ragged
volumes with the raycast of the result; streams that mix good frames with garbage, empty frames and jumps through the tracker,
synchronous and pipelined) over MANY more seeds than the suite holds.  The tests' two seed-specific expectations (so many updates, so many lost frames: about the suite's own seeds) are switched off,
so every comparison of bits, verdicts and counts runs for every seed, and whatever fails is a parity failure.   usage: tools/fuzz_campaign.py FIRST_SEED COUNT"""
import os
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))
import sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import housescan_amd as hsk
from housescan_amd import _lib
from oracle import oracle
import test_gpu_parity as T
oracle.build()
first, count = int(sys.argv[1]), int(sys.argv[2])
cache = {}
def synth_frames(k):
    if k not in cache:
        p = hsk.synth_pose(k)
        cache[k] = (p, hsk.synth_depth(p))
    return cache[k]
T.SEED_EXPECTATIONS = False   # parity only: the suite's seed-specific expectations are not in the path of any comparison here
integrate_fuzz = getattr(T.test_integrate_and_raycast_fuzz, "__wrapped__", T.test_integrate_and_raycast_fuzz)
tracker_fuzz = getattr(T.test_tracker_fuzz_vs_oracle, "__wrapped__", T.test_tracker_fuzz_vs_oracle)
bad, other = [], []
t0 = time.time()
for seed in range(first, first + count):
    for name, fn, args in (("integrate+raycast", integrate_fuzz, (hsk, oracle, seed)), ("tracker", tracker_fuzz, (hsk, oracle, synth_frames, seed))):
        try:
            fn(*args)
        except AssertionError as e:
            import traceback
            line = (traceback.extract_tb(e.__traceback__)[-1].line or "")
            msg = (str(e).split("\n")[0] or line)[:200]
            bad.append((name, seed, msg))   # (the seed-specific expectations are switched off above: whatever fails here is parity)
print(f"build {_lib.load().hsk_build_id().decode()}  fuzz seeds {first}..{first + count - 1} (two tests each, every comparison of every seed run): parity failures {len(bad)}, {time.time() - t0:.0f} s")
for b in bad: print("  PARITY", b)
sys.exit(1 if bad else 0)
