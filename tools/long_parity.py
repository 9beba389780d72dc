#!/usr/bin/env python3
"""Long-horizon parity: FRAMES frames of the scripted synthetic stream through the pipelined tracker at N^3 against the CPU
oracle's tracker -- every pose and the final TSDF bit for bit.  (The pytest suite holds shorter runs at these sizes: the
oracle takes 0.14 s per frame at 512^3 and 0.5 s at 1024^3 on 16 cores.)
usage: tools/long_parity.py N FRAMES [--noise | --holes | --room V [--first K] [--scan F] [--sensor]]
--noise: SURVEY.md 8(d)'s noise run instead of the exact render (sigma = 1.2 mm z^2 on every pixel, 2 % dropout; seeds 1234 / 5678)
--holes: the scripted stream with holes as a sensor makes them (hsk.synth_sensor_frames: grazing rays, shadow bands, range cut, sigma)
--room V: the ROOM SCAN -- camera inside the volume: frames K .. K + FRAMES - 1 of the F-frame (default 720) three-turn scan of
          closed room V (hsk_synth_room_*), the tracker started at the ground-truth pose of frame K"""
import os
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))  # (a 256-CPU box behind a 16-core quota)
import sys, time, numpy as np
sys.path.insert(0, '.')
import housescan_amd as hsk
from oracle import oracle
n, frames = int(sys.argv[1]), int(sys.argv[2])
opts = sys.argv[3:]
noise, holes = "--noise" in opts, "--holes" in opts
room = int(opts[opts.index("--room") + 1]) if "--room" in opts else None
what = ""
if room is not None:
    first = int(opts[opts.index("--first") + 1]) if "--first" in opts else 0
    scan = int(opts[opts.index("--scan") + 1]) if "--scan" in opts else 720
    gts = [hsk.synth_room_pose(room, first + k, scan) for k in range(frames)]
    if "--sensor" in opts:   # the room as a structured-light sensor sees it: grazing rays, shadow bands, absorbing furniture, noise
        fr = [hsk.synth_sensor_depth(p, scene=room, seed=1234 + first + i, absorbing=True)[0] for i, p in enumerate(gts)]
    else:
        fr = [hsk.synth_room_depth(room, p) for p in gts]
    cfg_o = oracle.default_config(n, omp=True, init_R=gts[0][:3, :3], init_t=gts[0][:3, 3])
    trk = hsk.KinfuTracker(n=n, init_pose=gts[0])
    what = f" ROOM SCAN (camera inside the volume{', as a SENSOR sees it: %.1f %% of the pixels invalid' % (100.0 * np.mean([(d == 0).mean() for d in fr])) if '--sensor' in opts else ''}): room {room}, frames {first}..{first + frames - 1} of a {scan}-frame three-turn scan"
else:
    cfg_o = oracle.default_config(n, omp=True)
    trk = hsk.KinfuTracker(n=n)
    if noise:
        gts, fr = hsk.synth_noisy_frames(frames)
        what = " NOISY stream (1.2 mm z^2, 2 % dropout)"
    elif holes:
        gts, fr = hsk.synth_sensor_frames(frames, absorbing=True)
        what = " SENSOR-HOLES stream (grazing rays, shadow bands, range cut, absorbing block, 1.2 mm z^2; %.1f %% of the pixels invalid)" % (100.0 * np.mean([(d == 0).mean() for d in fr]))
    else:
        gts = [hsk.synth_pose(k) for k in range(frames)]
        fr = [hsk.synth_depth(p) for p in gts]
ot = oracle.Tracker(cfg_o, omp=True)
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
worst = max(float(np.linalg.norm(p[:3, 3] - g[:3, 3])) for (p, _), g in zip(got, gts)) * 1e3
print(f"build {_lib.load().hsk_build_id().decode()}  n={n} frames={frames}{what}: pose mismatches {bad} of {frames}, differing voxels {dv} of {vol.shape[0] * vol.shape[1] * vol.shape[2]}, "
      f"lost frames {sum(1 for _, ok in got[1:] if not ok)}, worst translation error vs ground truth {worst:.2f} mm, oracle {t1 - t0:.1f} s")
sys.exit(1 if (bad or dv) else 0)
