#!/usr/bin/env python3
"""Integrate parity on depth images WITH HOLES, in seconds (the light class of pass A / pass B, round 6): frames of the noise run,
of the sensor-holes stream and of a room seen by a sensor, integrated at their ground-truth poses into n^3 volumes -- TSDF
(after the flush) and update counts against the CPU oracle, and how many lane-blocks went which way.
usage: light_parity.py [n ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import housescan_amd as hsk
from oracle import oracle
bad = 0
for n in [int(a) for a in sys.argv[1:]] or [64, 128, 256]:
    for name in ("noise", "holes", "room-sensor"):
        if name == "noise":
            gts, fr = hsk.synth_noisy_frames(24)
        elif name == "holes":
            gts, fr = hsk.synth_sensor_frames(24, absorbing=True)
        else:
            gts, fr = hsk.synth_sensor_frames(24, room=0, absorbing=True)
        cfg = oracle.default_config(n, omp=True)
        trk = hsk.KinfuTracker(n=n)
        vol = np.zeros((n, n, n, 2), np.int16)
        for k in (0, 1, 2, 9, 16, 23):
            pose, depth = gts[k], fr[k]
            nu = oracle.integrate(cfg, vol, oracle.scale_depth(cfg, depth), pose, omp=True)
            cu = trk.count_updates(depth, pose)
            trk.integrate(depth, pose)
            q, l = trk.integrate_queue_entries(), trk.integrate_light_entries()
            got = trk.download_tsdf() if k in (2, 23) else None
            d = int((got != vol).any(axis=-1).sum()) if got is not None else 0
            print(f"n={n} {name:11s} frame {k:2d}: updates {cu} vs {nu}, per-voxel entries {q}, light entries {l}" + (f", differing voxels {d}" if got is not None else ""), flush=True)
            bad += d + (cu != nu)
        trk.close()
print("PARITY OK" if bad == 0 else "PARITY BROKEN")
sys.exit(1 if bad else 0)
