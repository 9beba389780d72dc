#!/usr/bin/env python3
"""Per-wave life of k_integrate_detail (debug build with -DHSK_INT_TIMING): start / end stamps and the queue entries
each wave processed; shows how evenly the 256 queues load the chip (us, s_memrealtime at 100 MHz)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import housescan_amd as hsk
from housescan_amd import _lib
trk = hsk.KinfuTracker(n=int(sys.argv[1]) if len(sys.argv) > 1 else 512)
for k in range(40):
    trk.process_frame(hsk.synth_depth(hsk.synth_pose(k)))
lib = C.CDLL(_lib.LIB_PATH)
nw = 7 * 256 * 4
t = np.zeros((nw, 3), np.uint64)
rc = lib.hsk_debug_detail_times(C.c_void_p(t.ctypes.data), t.size)
ent = t[:, 2].astype(np.int64)
t = t[:, :2].astype(np.float64) / 100.0
t0 = t[:, 0].min()
life = t[:, 1] - t[:, 0]
print("rc", rc, "kernel span %.1f us, first-start spread %.1f us" % (t[:, 1].max() - t0, (t[:, 0] - t0).max()))
print("wave life   mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (life.mean(), *np.percentile(life, [50, 90, 99]), life.max()))
print("wave end    mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % ((t[:, 1] - t0).mean(), *np.percentile(t[:, 1] - t0, [50, 90, 99]), (t[:, 1] - t0).max()))
print("entries per wave: mean %.0f  p50 %.0f  p90 %.0f  max %.0f   total %d" % (ent.mean(), *np.percentile(ent, [50, 90]), ent.max(), ent.sum()))
q = ent.reshape(256, 28).sum(axis=1)
print("entries per queue: mean %.0f  min %d  p90 %.0f  max %d" % (q.mean(), q.min(), np.percentile(q, 90), q.max()))
print("us per 64 entries (wave life / trips): %.2f" % (life.sum() / max(1, (ent / 64.0).sum())))
order = np.argsort(-t[:, 1])[:8]
for w in order:
    print("  wave %5d queue %3d  start %.1f  end %.1f  entries %d" % (w, w // 28, t[w, 0] - t0, t[w, 1] - t0, ent[w]))
