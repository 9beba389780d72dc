#!/usr/bin/env python3
"""Replays the first <count> frames of the synthetic stream through one tracker (one integrate per frame, frame 0
included) and exits: the child process bench.py runs under `rocprofv3 --pmc` to read the integrate kernels' HBM
counters over exactly the frames of its timed region.  usage: replay_frames.py <volume> <count>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import housescan_amd as hsk  # noqa: E402

n, count = int(sys.argv[1]), int(sys.argv[2])
trk = hsk.KinfuTracker(n=n)
for k in range(count):
    trk.process_frame(hsk.synth_depth(hsk.synth_pose(k)))
trk.close()
