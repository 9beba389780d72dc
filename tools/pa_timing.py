#!/usr/bin/env python3
"""Per-wave phase stamps of integrate pass A (debug build with -DHSK_PA_TIMING; s_memrealtime at 100 MHz).
Prints, for the last integrate of a short fixed-pose run: the kernel span, how long waves live by what they had to do,
where that life goes (phase means), and how many waves a SIMD holds at a time (occupancy over the span)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import housescan_amd as hsk
from housescan_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
trk = hsk.KinfuTracker(n=n)
for k in range(12):
    p = hsk.synth_pose(k)
    trk.integrate(hsk.synth_depth(p), p)
lib = C.CDLL(_lib.LIB_PATH)
nw = 65536
raw = np.zeros((nw, 8), np.uint64)
rc = lib.hsk_debug_pa_times(C.c_void_p(raw.ctypes.data), raw.size)
t = raw[:, :6].astype(np.float64) / 100.0
ran = raw[:, 0] > 0
t0 = t[ran, 0].min()
end = np.where(t[:, 5] > 0, t[:, 5], t[:, 1])
span = end[ran].max() - t0
nf = (raw[:, 7] & 0xffffffff).astype(np.int64)
no = (raw[:, 7] >> 32).astype(np.int64)
work = ran & (t[:, 2] > 0)
life = end - t[:, 0]
print(f"rc {rc} waves {ran.sum()}  kernel span {span:.1f} us  first-to-last start {t[ran,0].max()-t0:.1f} us")
print(f"waves with work {work.sum()} ({100*work.sum()/ran.sum():.0f} %), life mean {life[work].mean():.2f} p50 {np.percentile(life[work],50):.2f} p90 {np.percentile(life[work],90):.2f} max {life[work].max():.2f}; idle waves life {life[ran & ~work].mean():.2f}")
ph = np.diff(t[work], axis=1)
names = ["prologue (z range, bounds)", "stage 1 (16-px table)", "stage 2 (pixel box, 8-px table)", "tickets + free loads/update/stores acked", "queue write"]
for i, nm in enumerate(names):
    print(f"   {nm:45s} mean {ph[:, i].mean():6.2f} us   p90 {np.percentile(ph[:, i], 90):6.2f}")
for lo, hi, nm in ((0, 0, "no free lane-block"), (1, 64, "1-64 free"), (65, 128, "65-128 free")):
    m = work & (nf >= lo) & (nf <= hi)
    if m.any():
        print(f"   waves with {nm:20s}: {m.sum():6d}, life {life[m].mean():5.2f}, phase 3 {np.diff(t[m], axis=1)[:, 3].mean():5.2f}; uncertain lane-blocks per wave {no[m].mean():.1f}")
print(f"free lane-blocks {nf[work].sum()}  uncertain {no[work].sum()}")
hw = raw[:, 6]
slot = ((hw >> 32) & 0xf) * 4096 + (hw & 0xffff & ~np.uint64(0xf))   # xcc, then HW_ID without the wave slot
# occupancy: waves alive per SIMD, sampled every 0.5 us
keys, inv = np.unique(slot[ran], return_inverse=True)
ts = np.arange(0, span, 0.5)
s_, e_ = t[ran, 0] - t0, end[ran] - t0
alive = np.array([((s_ <= x) & (e_ > x)).sum() for x in ts])
print(f"SIMD slots seen {len(keys)}; waves alive chip-wide: mean {alive.mean():.0f} max {alive.max()}  => per SIMD mean {alive.mean()/max(1,len(keys)):.2f}")
q = [alive[int(len(alive) * f)] for f in (0.1, 0.3, 0.5, 0.7, 0.9)]
print("waves alive at 10/30/50/70/90 % of the span:", q)
starts = np.sort(s_)
print("wave starts per us over the span: mean %.0f, in the first 5 us %.0f" % (len(starts) / span, (starts < 5).sum() / 5))
