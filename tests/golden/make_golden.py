#!/usr/bin/env python3
"""Generates tests/golden/kinfu_golden.npz from the NUMPY TWIN (tests/np_twin.py) on the deterministic synthetic
stream -- not from the C oracle, which the vectors are then checked against (tests/test_oracle_pins.py), and not from
the HIP path (tests/test_gpu_parity.py::test_golden_vectors_gpu).  The vectors remain SELF-GENERATED: the reference
repository holds no KinFu source, golden TSDF, trajectory or recorded depth stream (parity unpinned),
so they pin the agreement of three separately written restatements, regressions and cross-machine reproducibility --
not PCL equivalence.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
import housescan_amd as hsk  # noqa: E402  (host-only synthetic renderer: produces the INPUT frames)
import np_twin as T  # noqa: E402

W, H, FX, CX, CY = 160, 120, 131.25, 79.75, 59.75
N = 32
FRAMES = [0, 2, 4, 6, 8]


def depth(k):
    return hsk.synth_depth(hsk.synth_pose(k), W, H, FX, FX, CX, CY)


def main():
    trk = T.Tracker(N, W, H, FX, FX, CX, CY)
    poses = [trk.process(depth(k))[0] for k in FRAMES]
    d = depth(4)
    vm = T.vmap(T.bilateral(d), FX, FX, CX, CY)
    icp27, _ = T.icp_sums(vm, T.nmap(vm), trk.vmod[0], trk.nmod[0], FX, FX, CX, CY, poses[-1], poses[-1], np.float32(0.10),
                          np.float32(0.3420201433256687))
    x6, ok = T.icp_solve(icp27)
    assert ok
    cloud = T.extract_cloud(trk.vol, (3.0, 3.0, 3.0))
    _, _, keys, _ = T.raycast(trk.vol, (3.0, 3.0, 3.0), 0.03, W, H, FX, FX, CX, CY, poses[-1])
    out = os.path.join(HERE, "kinfu_golden.npz")
    np.savez_compressed(out, n=N, frames=np.array(FRAMES), poses=np.stack(poses),
                        tsdf_crop=trk.vol[8:24, 8:24, 8:24].copy(),
                        vmap_crop=trk.vmod[0][:, 40:70, 60:100].copy(),
                        nmap_crop=trk.nmod[0][:, 40:70, 60:100].copy(),
                        vmap2=trk.vmod[2].copy(), nmap2=trk.nmod[2].copy(),
                        depth1=depth(FRAMES[1]), bilateral1=T.bilateral(depth(FRAMES[1])), icp27=icp27, solve6=x6,
                        pose_after_solve=T.pose_update(poses[-1], x6), keys_crop=keys[40:70, 60:100].copy(),
                        cloud_head=cloud[:256].copy(), cloud_count=len(cloud))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
