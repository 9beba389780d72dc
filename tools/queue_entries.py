#!/usr/bin/env python3
"""queue entries of integrate (pass A -> pass B) against the voxels the rule rewrites, over the first frames of the
synthetic stream.  usage: queue_entries.py [volume] [frames]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import housescan_amd as h
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 26
trk = h.KinfuTracker(n=n)
ent = []
for k in range(frames):
    d = h.synth_depth(h.synth_pose(k))
    trk.process_frame(d)
    ent.append(trk.integrate_queue_entries())
print("volume", n, "queue entries per frame: first", ent[0], "mean of frames 6..", int(np.mean(ent[6:])), "last", ent[-1])
