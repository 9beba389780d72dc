#!/usr/bin/env python3
"""Long-horizon parity: FRAMES frames of the scripted synthetic stream through the pipelined tracker at N^3 against the CPU
oracle's tracker -- every pose and the final TSDF bit for bit.  (The pytest suite holds shorter runs at these sizes: the
oracle takes 0.14 s per frame at 512^3 and 0.5 s at 1024^3 on 16 cores.)   usage: tools/long_parity.py N FRAMES [--noise]
--noise: SURVEY.md 8(d)'s noise run instead of the exact render (sigma = 1.2 mm z^2 on every pixel, 2 % dropout; seeds 1234 / 5678)"""
import os
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))  # (a 256-CPU box behind a 16-core quota)
import sys, time, numpy as np
sys.path.insert(0, '.')
import housescan_amd as hsk
from oracle import oracle
n, frames = int(sys.argv[1]), int(sys.argv[2])
cfg_o = oracle.default_config(n, omp=True)
ot = oracle.Tracker(cfg_o, omp=True)
trk = hsk.KinfuTracker(n=n)
noise = "--noise" in sys.argv[3:]
fr = hsk.synth_noisy_frames(frames)[1] if noise else [hsk.synth_depth(hsk.synth_pose(k)) for k in range(frames)]
t0 = time.time()
want = [ot.process(d) for d in fr]
t1 = time.time()
got = []
trk.submit_frame(fr[0])
for d in fr[1:]:
    trk.submit_frame(d)
    got.append(trk.wait_frame())
got.append(trk.wait_frame())
bad = 0
for k, ((p, ok), (po, oko)) in enumerate(zip(got, want)):
    if ok != oko or np.ascontiguousarray(p, np.float32).tobytes() != np.ascontiguousarray(po, np.float32).tobytes():
        bad += 1
        if bad < 4: print("frame", k, "differs", ok, oko)
vol = trk.download_tsdf()
dv = int((vol != ot.volume()).any(axis=-1).sum())
from housescan_amd import _lib
print(f"build {_lib.load().hsk_build_id().decode()}  n={n} frames={frames}{' NOISY stream (1.2 mm z^2, 2 % dropout)' if noise else ''}: pose mismatches {bad} of {frames}, differing voxels {dv} of {vol.shape[0] * vol.shape[1] * vol.shape[2]}, "
      f"lost frames {sum(1 for _, ok in got[1:] if not ok)}, oracle {t1 - t0:.1f} s")
sys.exit(1 if (bad or dv) else 0)
