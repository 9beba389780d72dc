#!/usr/bin/env python3
"""The noise run (sigma = 1.2 mm z^2, 2 % dropout) through one tracker: coarse-level verdicts and pass B's
queue per frame.  Under `rocprofv3 --kernel-trace --stats` it gives the kernels' times on that stream (tools/noise_kstats.sh).
usage: noise_run.py [N=512] [frames=30] [dropout=0.02] [sigma_mm=1.2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import housescan_amd as hsk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dropout = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
sigma = float(sys.argv[4]) if len(sys.argv) > 4 else 1.2
gts, fr = hsk.synth_noisy_frames(frames, sigma_mm=sigma, dropout=dropout)
trk = hsk.KinfuTracker(n=n)
for k, d in enumerate(fr):
    pose, ok = trk.process_frame(d)
    if k % 5 == 4 or k == frames - 1:
        trk.lib.hsk_synchronize(trk.h)
        mixed, settled, free_worked, quiet = trk.integrate_coarse_counts()
        print(f"frame {k:3d} tracked {int(ok)}: mixed {mixed:6d} settled {settled:6d} free-but-worked {free_worked:6d} quiet {quiet:6d} queue {trk.integrate_queue_entries():8d}", flush=True)
trk.close()
