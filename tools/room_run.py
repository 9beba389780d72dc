#!/usr/bin/env python3
"""The room scan -- camera INSIDE the volume (hsk_synth_room_*, the three-turn turntable scan of synth.cpp) -- through one
tracker: stage times, V_upd, coarse-level verdicts and pass B's queue per window of frames, then the pipelined frame rate
over the same frames.  Under `rocprofv3 --kernel-trace --stats` it gives the kernels' times on that stream.
usage: room_run.py [N=512] [frames=120] [variant=0] [first=0] [scan_frames=720] [--parity]
--parity: every pose and the final TSDF against the CPU oracle's tracker (same init_pose), bit for bit"""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import housescan_amd as hsk

pos = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(pos[0]) if len(pos) > 0 else 512
frames = int(pos[1]) if len(pos) > 1 else 120
variant = int(pos[2]) if len(pos) > 2 else 0
first = int(pos[3]) if len(pos) > 3 else 0
scan = int(pos[4]) if len(pos) > 4 else 720
gts = [hsk.synth_room_pose(variant, first + k, scan) for k in range(frames)]
fr = [hsk.synth_room_depth(variant, p) for p in gts]
print(f"room {variant} frames {first}..{first + frames - 1} of a {scan}-frame scan at {n}^3; invalid pixels: "
      f"{np.mean([(d == 0).mean() for d in fr]):.4f}", flush=True)

trk = hsk.KinfuTracker(n=n, init_pose=gts[0])
trk.set_profiling(True)
win = 10
poses = []
for k, d in enumerate(fr):
    pose, ok = trk.process_frame(d)
    poses.append((pose.copy(), ok))
    if k % win == win - 1 or k == frames - 1:
        ms, nf = trk.stage_ms(reset=True)
        mixed, settled, free_worked, quiet = trk.integrate_coarse_counts()
        q = trk.integrate_queue_entries()
        vupd = trk.count_updates(d, pose)
        err = np.linalg.norm(pose[:3, 3] - gts[k][:3, 3]) * 1e3
        print(f"frame {k:3d} ok {int(ok)} err {err:6.2f} mm | us/frame pre {ms[0] / nf * 1e3:6.1f} icp {ms[1] / nf * 1e3:6.1f} int {ms[2] / nf * 1e3:6.1f} "
              f"ray {ms[3] / nf * 1e3:6.1f} | V_upd {vupd / 1e6:6.2f} M queue {q:8d} mixed {mixed:6d} settled {settled:6d} free-worked {free_worked:6d} quiet {quiet:6d}",
              flush=True)
trk.set_profiling(False)
vol = trk.download_tsdf() if "--parity" in sys.argv else None
trk.close()

# the pipelined rate over the same frames (host frames: hsk_submit_frame / hsk_wait_frame)
trk = hsk.KinfuTracker(n=n, init_pose=gts[0])
trk.process_frame(fr[0])
t0 = time.perf_counter()
trk.submit_frame(fr[1])
lost = 0
for d in fr[2:]:
    trk.submit_frame(d)
    lost += not trk.wait_frame()[1]
lost += not trk.wait_frame()[1]
trk.synchronize()
dt = time.perf_counter() - t0
print(f"pipelined, host frames: {(frames - 1) / dt:.1f} frames/s, lost {lost}", flush=True)
trk.close()

if "--parity" in sys.argv:
    from oracle import oracle
    cfg = oracle.default_config(n, omp=True, init_R=gts[0][:3, :3], init_t=gts[0][:3, 3])
    ot = oracle.Tracker(cfg, omp=True)
    bad = 0
    t0 = time.time()
    for k, d in enumerate(fr):
        po, oko = ot.process(d)
        p, ok = poses[k]
        if ok != oko or np.ascontiguousarray(p, np.float32).tobytes() != np.ascontiguousarray(po, np.float32).tobytes():
            bad += 1
            if bad < 4:
                print("frame", k, "differs", ok, oko)
    dv = int((vol != ot.volume()).any(axis=-1).sum())
    print(f"parity vs oracle: pose mismatches {bad} of {frames}, differing voxels {dv}, oracle {time.time() - t0:.1f} s")
    sys.exit(1 if (bad or dv) else 0)
