#!/usr/bin/env python3
"""Verdicts of integrate's coarse level per frame of the tracked synthetic stream (hsk_integrate_coarse_counts) beside pass
B's queue length.  usage: coarse_counts.py [N=512] [frames=30]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import housescan_amd as hsk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
trk = hsk.KinfuTracker(n=n)
for k in range(frames):
    pose, ok = trk.process_frame(hsk.synth_depth(hsk.synth_pose(k)))
    trk.lib.hsk_synchronize(trk.h)
    mixed, settled, free_worked, quiet = trk.integrate_coarse_counts()
    print(f"frame {k:3d} tracked {int(ok)}: mixed {mixed:6d} settled {settled:6d} free-but-worked {free_worked:6d} quiet {quiet:6d} "
          f"queue {trk.integrate_queue_entries():7d}", flush=True)
trk.close()
