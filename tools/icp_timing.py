#!/usr/bin/env python3
"""Per-block phase times of k_icp_iter (debug build with -DHSK_ICP_TIMING): entry -> sums reduced -> solved ->
pixels accumulated -> block sums added, for each of the 19 iterations of one frame (us, s_memrealtime at 100 MHz)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import housescan_amd as hsk
from housescan_amd import _lib
trk = hsk.KinfuTracker(n=512)
for k in range(40):
    trk.process_frame(hsk.synth_depth(hsk.synth_pose(k)))
lib = C.CDLL(_lib.LIB_PATH)
t = np.zeros((20, 256, 10), np.uint64)
rc = lib.hsk_debug_icp_times(C.c_void_p(t.ctypes.data), t.size)
t = t.astype(np.float64) / 100.0
nb = [75] * 4 + [150] * 5 + [240] * 10
print("rc", rc)
print("iter blocks | first start  start spread | reduce  solve  pixels  sums | block life mean/max | kernel span | gap to next first start")
prev_end = None
for i in range(19):
    b = t[i, :nb[i]]
    t0 = b[:, 0].min()
    d = np.diff(b[:, :5], axis=1)
    end = b[:, 4].max()
    gap = (t[i + 1, :nb[i + 1], 0].min() - end) if i < 18 else float('nan')
    if i > 0 and b[:, 5:9].min() > 0:   # (the solve-step stamps exist only when hsk_icp_dev.h saw ICP_STAMP defined)
        inner = np.diff(np.concatenate([b[:, 1:2], b[:, 5:9], b[:, 2:3]], axis=1), axis=1).mean(axis=0)
        extra = "  || sums->regs %.2f solve6 %.2f shfl+sincos %.2f pose %.2f publish+barrier %.2f" % tuple(inner)
    else:
        extra = ""
    print(f"{i:4d} {nb[i]:5d} | {t0 - t[0, :75, 0].min():9.2f} {b[:, 0].max() - t0:9.2f} | " + " ".join(f"{x:6.2f}" for x in d.mean(axis=0)) +
          f" | {(b[:, 4] - b[:, 0]).mean():6.2f} {(b[:, 4] - b[:, 0]).max():6.2f} | {end - t0:6.2f} | {gap:6.2f}" + extra)
