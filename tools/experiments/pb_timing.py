#!/usr/bin/env python3
"""Per-wave stamps of integrate pass B (debug build with -DHSK_PB_TIMING; s_memrealtime at 100 MHz): start, prologue
done (256 queue counters read and scanned), end of each trip.  usage (GPU box): tools/pb_timing.sh [volume]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import housescan_amd as hsk
from housescan_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
trk = hsk.KinfuTracker(n=n)
for k in range(26):
    trk.process_frame(hsk.synth_depth(hsk.synth_pose(k)))
trk.synchronize()
print("queue entries", trk.integrate_queue_entries())
lib = C.CDLL(_lib.LIB_PATH)
raw = np.zeros((8192, 8), np.uint64)
rc = lib.hsk_debug_pb_times(C.c_void_p(raw.ctypes.data), raw.size)
t = raw.astype(np.float64) / 100.0
ran = raw[:, 0] > 0
t0 = t[ran, 0].min()
ntrips = (raw[:, 2:7] > 0).sum(axis=1)
end = np.where(ntrips > 0, t[np.arange(len(t)), 1 + ntrips], t[:, 1])
print(f"rc {rc} waves {ran.sum()}  span {end[ran].max() - t0:.1f} us  first-to-last start {t[ran, 0].max() - t0:.1f}")
print(f"prologue mean {(t[ran, 1] - t[ran, 0]).mean():.2f} p90 {np.percentile(t[ran, 1] - t[ran, 0], 90):.2f}")
for k in range(1, 5):
    m = ran & (ntrips >= k)
    if m.any():
        d = t[m, 1 + k] - t[m, k]
        print(f"trip {k}: waves {m.sum():5d}  mean {d.mean():6.2f} p10 {np.percentile(d, 10):6.2f} p90 {np.percentile(d, 90):6.2f} max {d.max():6.2f}")
for k in range(0, 5):
    m = ran & (ntrips == k)
    if m.any():
        print(f"waves with {k} trips: {m.sum():5d}, life mean {(end[m] - t[m, 0]).mean():6.2f}, end mean {(end[m] - t0).mean():6.2f} max {(end[m] - t0).max():6.2f}")
