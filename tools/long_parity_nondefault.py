#!/usr/bin/env python3
"""Long-horizon parity with EVERY configuration field off its default at once (what tests/test_gpu_configs.py holds for ten
frames of a small volume): a non-cubic, non-power-of-two volume with anisotropic cells, a longer truncation distance,
fx != fy and an off-centre principal point, other ICP iteration counts and gates, another start pose -- FRAMES pipelined
frames against the CPU oracle's tracker, every pose and the final TSDF bit for bit.   usage: tools/long_parity_nondefault.py [FRAMES] [--holes]
--holes: the frames as a structured-light sensor returns them (hsk_synth_render_sensor with THIS configuration's intrinsics: grazing rays,
         shadow bands, range cut, absorbing block, noise) -- the light class of pass A / pass B away from every default"""
import os
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))
import sys, time, numpy as np
sys.path.insert(0, '.')
import housescan_amd as hsk
from housescan_amd import _lib
from oracle import oracle
frames = int([a for a in sys.argv[1:] if not a.startswith("--")][0]) if [a for a in sys.argv[1:] if not a.startswith("--")] else 300
holes = "--holes" in sys.argv[1:]
W, H = 640, 480
fx, fy, cx, cy = 525.0 * 1.04, 525.0 * 0.97, W / 2 - 0.5 + 6.5, H / 2 - 0.5 - 5.0
vol, size = (384, 320, 448), (3.0, 2.6, 3.2)            # cells 7.8 / 8.1 / 7.1 mm
iters = (8, 4, 3)
start = hsk.synth_pose(0).copy()
start[:3, 3] += np.array([0.02, -0.03, 0.01], np.float32)
sine = float(np.sin(np.radians(15.0)))
cfg_o = oracle.default_config(vol[0], vol=vol, size=size, trunc=0.045, W=W, H=H, fx=fx, fy=fy, cx=cx, cy=cy, icp_iters=iters,
                              dist_thresh=0.07, angle_thresh=sine, init_R=start[:3, :3], init_t=start[:3, 3], omp=True)
cfg_h = hsk.default_config(vol[0], vol_y=vol[1], vol_z=vol[2], own_z1=vol[2], vol_size_m=size, trunc_dist_m=0.045, width=W, height=H,
                           fx=fx, fy=fy, cx=cx, cy=cy, icp_iters=iters, icp_dist_thresh_m=0.07, icp_angle_thresh_sin=sine, init_pose=start)
ot = oracle.Tracker(cfg_o, omp=True)
trk = hsk.KinfuTracker(cfg_h)
if holes:
    fr = [hsk.synth_sensor_depth(hsk.synth_pose(k), -1, 1234 + k, 1.2, 3.5, True, W, H, fx, fy, cx, cy)[0] for k in range(frames)]
else:
    fr = [hsk.synth_depth(hsk.synth_pose(k), W, H, fx, fy, cx, cy) for k in range(frames)]
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
v = trk.download_tsdf()
dv = int((v != ot.volume()).any(axis=-1).sum())
print(f"build {_lib.load().hsk_build_id().decode()}  volume {vol[0]}x{vol[1]}x{vol[2]} voxels of {size} m, fx {fx:.1f} fy {fy:.1f} cx {cx} cy {cy}, trunc 0.045, "
      f"ICP {iters}, frames={frames}{' SENSOR-HOLES stream (%.1f %% of the pixels invalid)' % (100.0 * np.mean([(d == 0).mean() for d in fr])) if holes else ''}: pose mismatches {bad} of {frames}, differing voxels {dv} of {v.shape[0] * v.shape[1] * v.shape[2]}, "
      f"lost frames {sum(1 for _, ok in got[1:] if not ok)}, tracked by the oracle {sum(1 for _, ok in want[1:] if ok)}, oracle {t1 - t0:.1f} s")
sys.exit(1 if (bad or dv) else 0)
