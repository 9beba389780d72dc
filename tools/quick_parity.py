#!/usr/bin/env python3
"""Integrate parity in seconds, for experimental builds (the pytest suite refuses a library whose build id says +exp):
frames 0, 7, 14, 21 of the synthetic stream into n^3 volumes, TSDF and update counts against the CPU oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import housescan_amd as hsk
from oracle import oracle
bad = 0
for n in [int(a) for a in sys.argv[1:]] or [64, 128, 256]:  # (64: one x block per row, the smallest grid)
    cfg = oracle.default_config(n, omp=True)
    trk = hsk.KinfuTracker(n=n)
    vol = np.zeros((n, n, n, 2), np.int16)
    for k in (0, 7, 14, 21):
        pose = hsk.synth_pose(k)
        depth = hsk.synth_depth(pose)
        nu = oracle.integrate(cfg, vol, oracle.scale_depth(cfg, depth), pose, omp=True) if "omp" in oracle.integrate.__code__.co_varnames else oracle.integrate(cfg, vol, oracle.scale_depth(cfg, depth), pose)
        cu = trk.count_updates(depth, pose)
        trk.integrate(depth, pose)
        got = trk.download_tsdf()
        d = int((got != vol).any(axis=-1).sum())
        print(f"n={n} frame {k}: updates {cu} vs {nu}, differing voxels {d}")
        bad += d + (cu != nu)
    trk.close()
print("PARITY OK" if bad == 0 else "PARITY BROKEN")
sys.exit(1 if bad else 0)
