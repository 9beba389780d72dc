#!/usr/bin/env python3
"""Generates tests/golden/kinfu_golden.npz from the CPU oracle (oracle/kinfu_oracle.c) on the deterministic
synthetic stream.  These vectors are SELF-GENERATED: the reference repository holds no KinFu source, golden
TSDF, trajectory or recorded depth stream (parity unpinned), so they pin self-consistency,
regressions and cross-machine reproducibility -- not PCL equivalence.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import housescan_amd as hsk  # noqa: E402  (host-only synth renderer)
from oracle import oracle as O  # noqa: E402

W, H, FX, CX, CY = 160, 120, 131.25, 79.75, 59.75
N = 32
FRAMES = [0, 2, 4, 6, 8]


def depth(k):
    return hsk.synth_depth(hsk.synth_pose(k), W, H, FX, FX, CX, CY)


def main():
    cfg = O.default_config(N, W=W, H=H, fx=FX, fy=FX, cx=CX, cy=CY)
    trk = O.Tracker(cfg)
    poses = [trk.process(depth(k))[0] for k in FRAMES]
    d = depth(4)
    vm = O.vmap(cfg, O.bilateral(cfg, d))
    icp27, _ = O.icp_accumulate(cfg, 0, vm, O.nmap(vm), trk.model_map(2, 0), trk.model_map(3, 0), poses[-1], poses[-1])
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kinfu_golden.npz")
    np.savez_compressed(out, n=N, frames=np.array(FRAMES), poses=np.stack(poses),
                        tsdf_crop=trk.volume()[8:24, 8:24, 8:24].copy(),
                        vmap_crop=trk.model_map(2, 0)[:, 40:70, 60:100].copy(),
                        nmap_crop=trk.model_map(3, 0)[:, 40:70, 60:100].copy(),
                        depth1=depth(FRAMES[1]), icp27=icp27)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
