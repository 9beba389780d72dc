#!/usr/bin/env python3
"""Integrate-only bench on a fixed stream: hsk_integrate with the ground-truth poses of the synthetic stream (no
tracking, so timing experiments that give wrong voxels cannot derail it).  Run under rocprofv3 --kernel-trace --stats
(tools/experiments/int_ab.sh) and read the integrate kernels' average durations.  usage: int_bench.py [volume] [first] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import housescan_amd as hsk  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 26
trk = hsk.KinfuTracker(n=n)
for k in range(first, first + count):
    p = hsk.synth_pose(k)
    trk.integrate(hsk.synth_depth(p), p)
trk.close()
print("done", n, first, count)
