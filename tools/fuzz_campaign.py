#!/usr/bin/env python3
"""The suite's two fuzz tests (tests/test_gpu_parity.py: random depth images from arbitrary poses into cubic / non-cubic / ragged
volumes with the raycast of the result; streams that mix good frames with garbage, empty frames and jumps through the tracker,
synchronous and pipelined) over MANY more seeds than the suite holds.  A seed counts as a parity failure when an assertion
about bits, verdicts or counts fails; the tests' seed-specific expectations (so many updates, so many lost frames) are reported
apart.   usage: tools/fuzz_campaign.py FIRST_SEED COUNT"""
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
            # the two expectations that depend on what a seed's random frames happen to be; everything else is parity
            expectation = ("seed in (0, 3)" in line) or ("total > 30000" in line)
            (other if expectation else bad).append((name, seed, msg))
print(f"build {_lib.load().hsk_build_id().decode()}  fuzz seeds {first}..{first + count - 1} (two tests each): parity failures {len(bad)}, "
      f"seed-specific expectations not met {len(other)} (not parity: e.g. a seed whose garbage frames happen to be tracked), {time.time() - t0:.0f} s")
for b in bad: print("  PARITY", b)
for o in other[:6]: print("  other", o)
sys.exit(1 if bad else 0)
