#!/usr/bin/env python3
"""Per-wave phase times of k_raycast (debug build with -DHSK_RC_TIMING): staging / march / refine, and where the
slowest tiles are in the image."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import housescan_amd as hsk
from housescan_amd import _lib
trk = hsk.KinfuTracker(n=int(sys.argv[1]) if len(sys.argv) > 1 else 512, use_graph=0)
for k in range(40):
    trk.process_frame(hsk.synth_depth(hsk.synth_pose(k)))
lib = C.CDLL(_lib.LIB_PATH)
tall = np.zeros((8192, 8), np.uint64)
rc = lib.hsk_debug_rc_times(C.c_void_p(tall.ctypes.data), 8192 * 8)
t = tall[:4800].copy()
helpers = tall[4800:4800 + 320]
hl = helpers[(helpers[:, 3] > 0) & (helpers[:, 0] + np.uint64(200) >= tall[:4800, 0].min())]   # (this launch's: the stamps are never cleared)
if len(hl):   # helper waves of the splitting raycast (workgroups behind the tiles' own)
    h0 = tall[:4800, 0].min()
    hs = hl[:, :4].astype(np.float64) / 100.0
    print("helper waves that marched: %d; start %.1f..%.1f us, life mean %.1f max %.1f, end mean %.1f max %.1f; trips mean %.0f" % (
        len(hl), (hs[:, 0] - h0 / 100.0).min(), (hs[:, 0] - h0 / 100.0).max(), (hs[:, 3] - hs[:, 0]).mean(), (hs[:, 3] - hs[:, 0]).max(),
        (hs[:, 3] - h0 / 100.0).mean(), (hs[:, 3] - h0 / 100.0).max(), hl[:, 4].mean()))
adopted = (t[:, 4] >> np.uint64(32)).astype(np.int64) & 1
print("tiles that adopted their helper's results:", int(adopted.sum()))
trips = (t[:, 4] & np.uint64(0xffffffff)).astype(np.int64)
gtrips = (t[:, 5] & np.uint64(0xffff)).astype(np.int64)
it_all = ((t[:, 5] >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64)
it_skip = ((t[:, 5] >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)
it_empty = ((t[:, 5] >> np.uint64(48)) & np.uint64(0xffff)).astype(np.int64)
it_reg = it_all - it_skip
print("loop iterations per wave: mean %.1f p90 %.0f max %d = crossings %.1f + regular trips %.1f (of which no lane gathered: %.1f)" % (
    it_all.mean(), np.percentile(it_all, 90), it_all.max(), it_skip.mean(), it_reg.mean(), it_empty.mean()))
setup = (t[:, 6].astype(np.float64) - t[:, 1].astype(np.float64)) / 100.0   # ray set-up: after staging until the march loop starts
print("ray set-up (before the loop): mean %.1f p90 %.1f max %.1f us" % (setup.mean(), np.percentile(setup, 90), setup.max()))
t = t[:, :4].astype(np.float64) / 100.0   # s_memrealtime ticks at 100 MHz -> us
t0 = t[:, 0].min()
stage, march, refine, life = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
print("rc", rc, "span", t[:, 3].max() - t0, "first start spread", (t[:, 0] - t0).max())
for name, v in (("staging", stage), ("march", march), ("refine", refine), ("life", life), ("end", t[:, 3] - t0)):
    print(f"{name:8s} mean {v.mean():6.1f}  p50 {np.percentile(v, 50):6.1f}  p90 {np.percentile(v, 90):6.1f}  p99 {np.percentile(v, 99):6.1f}  max {v.max():6.1f}")
order = np.argsort(-(t[:, 3] - t0))[:12]
for i in order:
    row = i // 80
    row = (60 - 1 - (row >> 1)) if (row & 1) else (row >> 1)   # k_raycast dispatches tile rows from the edges inwards
    print("trips %4d gather-trips %4d us/trip %.3f |" % (trips[i], gtrips[i], march[i] / max(1, trips[i])), "tile", i, "xy", (i % 80) * 8, row * 8, "start %.1f stage %.1f march %.1f refine %.1f end %.1f" % (t[i, 0] - t0, stage[i], march[i], refine[i], t[i, 3] - t0))

print("trips: mean %.0f p50 %.0f p90 %.0f max %d; gather-trips mean %.0f max %d" % (trips.mean(), np.percentile(trips, 50), np.percentile(trips, 90), trips.max(), gtrips.mean(), gtrips.max()))
# march time against trips: least squares us = a * trips + b * gather_trips
A = np.stack([trips, gtrips, np.ones_like(trips)], axis=1).astype(np.float64)
coef, *_ = np.linalg.lstsq(A, march, rcond=None)
print("march us ~ %.3f * trips + %.3f * gather_trips + %.1f" % tuple(coef))
A2 = np.stack([it_skip, it_reg - it_empty, it_empty, np.ones_like(trips)], axis=1).astype(np.float64)
c2, *_ = np.linalg.lstsq(A2, march, rcond=None)
print("march us ~ %.2f * crossings + %.2f * gathering trips + %.2f * empty trips + %.1f" % tuple(c2))
late = t[:, 2] - t0 > 70
print("waves whose march ends after 70 us: %d, their trips mean %.0f gather-trips mean %.0f" % (late.sum(), trips[late].mean() if late.any() else 0, gtrips[late].mean() if late.any() else 0))
