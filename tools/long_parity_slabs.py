#!/usr/bin/env python3
"""Long-horizon parity of the z-slab group: FRAMES frames of the scripted stream through `hsk_group_*`
with SLABS slabs -- all on the one device of a gpurun box, the exchange in the form FORM -- against the CPU oracle's
single-volume tracker: every pose, and the final TSDF assembled from the slabs' owned planes, bit for bit.  (The pytest
suite holds 12 - 40 frames of this; tools/long_parity.py is the single-volume run.)
usage: tools/long_parity_slabs.py N FRAMES SLABS [direct|composite|icp_allreduce] [--noise | --holes]
--noise / --holes: SURVEY.md 8(d)'s noise run / the sensor-holes stream (tools/long_parity.py) -- the light class inside slab contexts"""
import os
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))
import sys, time, numpy as np
sys.path.insert(0, '.')
import housescan_amd as hsk
from housescan_amd import _lib
from oracle import oracle
n, frames, slabs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
form = sys.argv[4] if len(sys.argv) > 4 and not sys.argv[4].startswith("--") else "direct"
stream = "noise" if "--noise" in sys.argv else "holes" if "--holes" in sys.argv else "scripted"
flags = {"direct": 4, "composite": 0, "icp_allreduce": 2}[form]  # HSK_GROUP_DIRECT / the staged composites / HSK_GROUP_ICP_ALLREDUCE
ot = oracle.Tracker(oracle.default_config(n, omp=True), omp=True)
grp = hsk.KinfuGroup(n=n, device_ids=(0,) * slabs, flags=flags)
fr = (hsk.synth_noisy_frames(frames)[1] if stream == "noise" else hsk.synth_sensor_frames(frames, absorbing=True)[1] if stream == "holes"
      else [hsk.synth_depth(hsk.synth_pose(k)) for k in range(frames)])
t0 = time.time()
want = [ot.process(d) for d in fr]
t1 = time.time()
got = []
grp.submit_frame(fr[0])
for d in fr[1:]:
    grp.submit_frame(d)
    got.append(grp.wait_frame())
got.append(grp.wait_frame())
t2 = time.time()
bad = 0
for k, ((p, ok), (po, oko)) in enumerate(zip(got, want)):
    if ok != oko or np.ascontiguousarray(p, np.float32).tobytes() != np.ascontiguousarray(po, np.float32).tobytes():
        bad += 1
        if bad < 4: print("frame", k, "differs", ok, oko)
vol = grp.download_tsdf()
dv = int((vol != ot.volume()).any(axis=-1).sum())
print(f"build {_lib.load().hsk_build_id().decode()}  n={n} frames={frames} slabs={grp.n_slabs()} ({form}, one device; {stream} stream, {100.0 * np.mean([(d == 0).mean() for d in fr]):.1f} % of the pixels invalid): pose mismatches {bad} of {frames}, "
      f"differing voxels {dv} of {vol.shape[0] * vol.shape[1] * vol.shape[2]}, lost frames {sum(1 for _, ok in got[1:] if not ok)}, "
      f"group {frames / (t2 - t1):.0f} frames/s, oracle {t1 - t0:.1f} s")
grp.close()
sys.exit(1 if (bad or dv) else 0)
