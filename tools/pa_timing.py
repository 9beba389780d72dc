#!/usr/bin/env python3
"""Per-wave phase stamps of integrate pass A (debug build with -DHSK_PA_TIMING; s_memrealtime at 100 MHz), round-5 form:
only the waves whose wave-chunk the coarse level left to pass A stamp (slot = wave-chunk).  Prints, for the last integrate
of a short fixed-pose run: the span, the life of a working wave by phase, and what the chunks held."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import housescan_amd as hsk
from housescan_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
trk = hsk.KinfuTracker(n=n)
lib = C.CDLL(_lib.LIB_PATH)
for k in range(12):
    p = hsk.synth_pose(k)
    if k == 11:
        trk.lib.hsk_synchronize(trk.h)
        lib.hsk_debug_pa_clear()
    trk.integrate(hsk.synth_depth(p), p)
print("coarse counts (mixed, settled, free-but-worked, quiet):", trk.integrate_coarse_counts())
raw = np.zeros((65536, 8), np.uint64)
rc = lib.hsk_debug_pa_times(C.c_void_p(raw.ctypes.data), raw.size)
t = raw[:, :6].astype(np.float64) / 100.0
ran = raw[:, 5] > 0
tt = t[ran]
t0 = tt[:, 0].min()
life = tt[:, 5] - tt[:, 0]
print(f"working waves {ran.sum()}  first start .. last end {tt[:, 5].max() - t0:.1f} us; starts p50 {np.percentile(tt[:, 0] - t0, 50):.2f} p90 {np.percentile(tt[:, 0] - t0, 90):.2f} max {(tt[:, 0] - t0).max():.2f}; "
      f"life mean {life.mean():.2f} p50 {np.percentile(life, 50):.2f} p90 {np.percentile(life, 90):.2f} max {life.max():.2f}")
ph = np.diff(tt, axis=1)
for i, nm in enumerate(["prologue (z range, pose)", "stage 1 (16-px table, summaries)", "stage 2 (pixel box, fine table)",
                        "tickets + volume loads / stores acked", "queue write"]):
    print(f"   {nm:45s} mean {ph[:, i].mean():6.2f} us   p90 {np.percentile(ph[:, i], 90):6.2f}")
nf = (raw[ran, 7] & np.uint64(0xffffffff)).astype(np.int64)
no = (raw[ran, 7] >> np.uint64(32)).astype(np.int64)
print(f"   free lane-blocks per wave {nf.mean():.1f}, uncertain {no.mean():.1f}; waves with none of either {(nf + no == 0).sum()}")
ts = np.arange(0, tt[:, 5].max() - t0, 1.0)
alive = [int(((tt[:, 0] - t0 <= x) & (tt[:, 5] - t0 > x)).sum()) for x in ts]
print("working waves alive at each microsecond:", alive)
