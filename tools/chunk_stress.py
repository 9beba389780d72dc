#!/usr/bin/env python3
"""Stress of the two-level summaries (integrate.hip: lane-block bytes + wave-chunk bytes) at a size where the coarse level
fires: N^3 (256 by default), per seed a walk of camera poses -- a few anchor poses revisited many times in a row (chunk bytes
count up), jumps between them and random poses in between (pending counts pushed down under frustums that cut the chunks),
smooth depth with holes and blocky garbage mixed -- no read-back until the checks: the volume against the oracle's at a few
points and at the end, a cloud taken mid-way (no flush).   usage: tools/chunk_stress.py [N=256] [first_seed=0] [seeds=10] [steps=90]"""
import os
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))
import sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import housescan_amd as hsk
from housescan_amd import _lib
from oracle import oracle
import test_gpu_parity as T
oracle.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 10
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 90
bad = 0
t0 = time.time()
for seed in range(first, first + seeds):
    rng = np.random.default_rng(7000 + seed)
    cfg_o = oracle.default_config(n)
    trk = hsk.KinfuTracker(n=n)
    vol = np.zeros((n, n, n, 2), np.int16)
    anchors = [hsk.synth_pose(int(k)) for k in rng.integers(0, 150, 3)] + [T._lookat_pose(rng, np.array([1.5, 1.5, 1.5]), 1.2)]
    frames = []
    for a in anchors:
        d = hsk.synth_depth(a).astype(np.int64)
        d[rng.random(d.shape) < rng.choice([0.0, 0.0, 0.02, 0.1])] = 0           # holes: none, sparse, dense
        hy, hx = rng.integers(0, 400), rng.integers(0, 520)
        d[hy:hy + 60, hx:hx + 90] = 0
        frames.append(np.clip(d, 0, 65535).astype(np.uint16))
    cur, quiet_max, checks = 0, 0, 0
    for i in range(steps):
        r = rng.random()
        if r < 0.70:
            pose, depth = anchors[cur], frames[cur]
        elif r < 0.85:
            cur = int(rng.integers(0, len(anchors)))
            pose, depth = anchors[cur], frames[cur]
        elif r < 0.95:
            pose, depth = T._lookat_pose(rng, np.array([1.5, 1.5, 1.0]), 1.0), frames[cur]   # the same image from elsewhere
        else:
            pose, depth = anchors[cur], T._random_depth(rng)                               # blocky garbage
        oracle.integrate(cfg_o, vol, oracle.scale_depth(cfg_o, depth), pose, omp=True)
        trk.integrate(depth, pose)
        if i % 30 == 29:
            quiet_max = max(quiet_max, trk.integrate_coarse_counts()[3])
        if i == steps // 2:
            pts, total = trk.extract_cloud()
            opts, ototal = oracle.extract_cloud(cfg_o, vol)
            if total != ototal or not np.array_equal(pts.view(np.uint32), opts.view(np.uint32)):
                bad += 1
                print(f"seed {seed}: cloud differs at step {i}")
        if i in (steps // 3, steps - 1):
            got = trk.download_tsdf()
            dv = int((got != vol).any(axis=-1).sum())
            checks += 1
            if dv:
                bad += 1
                print(f"seed {seed}: {dv} voxels differ at step {i}")
    trk.close()
    print(f"seed {seed}: ok so far ({checks} volume checks, most quiet chunks seen {quiet_max})", flush=True)
print(f"build {_lib.load().hsk_build_id().decode()}  chunk stress {n}^3, seeds {first}..{first + seeds - 1}, {steps} integrations each: failures {bad}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
