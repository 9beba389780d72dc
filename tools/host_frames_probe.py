#!/usr/bin/env python3
"""Where does a host-frame stream stall?  Frames of the scripted stream through hsk_submit_frame / hsk_wait_frame in a fresh
process: host time of every submit and every wait, the outliers listed (frame, which call, ms).
usage: host_frames_probe.py [N=512] [frames=300] [--warm-sync W]   (--warm-sync: W frames through hsk_process_frame first, as bench.py's warm-up does)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import housescan_amd as hsk
pos = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(pos[0]) if pos else 512
frames = int(pos[1]) if len(pos) > 1 else 300
warm = int(sys.argv[sys.argv.index("--warm-sync") + 1]) if "--warm-sync" in sys.argv else 0
fr = [hsk.synth_depth(hsk.synth_pose(k)) for k in range(frames)]
idle = None
if "--idle-ctx" in sys.argv:   # a second context, alive and idle, beside the one that is fed (bench.py keeps its first tracker for the read-outs)
    idle = hsk.KinfuTracker(n=n)
    for k in range(4):
        idle.process_frame(fr[k])
    idle.synchronize()
if "--torch" in sys.argv:
    import torch
    x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
trk = hsk.KinfuTracker(n=n)
for k in range(1 + warm):
    trk.process_frame(fr[k])
trk.synchronize()
ts, tw = [], []
t00 = time.perf_counter()
t0 = time.perf_counter(); trk.submit_frame(fr[1 + warm]); ts.append(time.perf_counter() - t0)
for k in range(2 + warm, frames):
    t0 = time.perf_counter(); trk.submit_frame(fr[k]); t1 = time.perf_counter(); trk.wait_frame(); t2 = time.perf_counter()
    ts.append(t1 - t0); tw.append(t2 - t1)
trk.wait_frame(); trk.synchronize()
total = time.perf_counter() - t00
ts, tw = np.array(ts) * 1e3, np.array(tw) * 1e3
print(f"{frames - 1 - warm} pipelined host frames at {n}^3: {(frames - 1 - warm) / total:.1f} frames/s; submit ms median {np.median(ts):.3f} p99 {np.percentile(ts, 99):.3f} max {ts.max():.3f}; "
      f"wait ms median {np.median(tw):.3f} p99 {np.percentile(tw, 99):.3f} max {tw.max():.3f}")
for name, a in (("submit", ts), ("wait", tw)):
    for i in np.argsort(-a)[:6]:
        if a[i] > 0.6:
            print(f"  outlier: {name} of timed frame {i}: {a[i]:.3f} ms")
trk.close()
