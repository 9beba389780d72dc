#!/usr/bin/env python3
"""A stream with holes through one tracker: coarse-level verdicts and pass B's two queue classes per frame, stage times at the end.
Under `rocprofv3 --kernel-trace --stats` it gives the kernels' times on that stream (tools/noise_kstats.sh).
usage: noise_run.py [N=512] [frames=30] [stream=noise]      stream: noise | holes | scripted | room0 (tools/replay_frames.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import housescan_amd as hsk
from replay_frames import stream_frames
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
stream = sys.argv[3] if len(sys.argv) > 3 else "noise"
gts, fr, init = stream_frames(hsk, stream, frames)
print(f"stream {stream}: invalid pixels {np.mean([(d == 0).mean() for d in fr]):.4f}", flush=True)
trk = hsk.KinfuTracker(n=n) if init is None else hsk.KinfuTracker(n=n, init_pose=init)
trk.set_profiling(True)
for k, d in enumerate(fr):
    pose, ok = trk.process_frame(d)
    if k == 5:
        trk.stage_ms(reset=True)
    if k % 5 == 4 or k == frames - 1:
        trk.synchronize()
        mixed, settled, free_worked, quiet = trk.integrate_coarse_counts()
        print(f"frame {k:3d} tracked {int(ok)}: mixed {mixed:6d} settled {settled:6d} free-but-worked {free_worked:6d} quiet {quiet:6d} "
              f"queue {trk.integrate_queue_entries():8d} light {trk.integrate_light_entries():8d}", flush=True)
ms, nf = trk.stage_ms()
print(f"us/frame over frames 6..{frames - 1}: pre {ms[0] / nf * 1e3:.1f} icp {ms[1] / nf * 1e3:.1f} integrate {ms[2] / nf * 1e3:.1f} raycast {ms[3] / nf * 1e3:.1f}")
trk.close()
