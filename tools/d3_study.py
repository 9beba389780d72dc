#!/usr/bin/env python3
"""CPU study for deviation D3: where does the interpolated hit time T* = t - step F_t / (F_{t+step} - F_t)
of a raycast zero crossing fall relative to the march step [t, t + step] that found it?  The specification kept only
T* in [t - step/2, t + 3 step/2]; Appendix A.6 as written keeps whatever the interpolation gives.  Runs the scripted synthetic
stream through the oracle built with A.6's acceptance (ORA_LIT_D3) and a histogram of (T* - t) / step (ORA_D3_STATS).
usage: tools/d3_study.py [volume] [frames]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 100
O.build_variant("d3stats", "-DORA_LIT_D3 -DORA_D3_STATS")
L = O.lib("var:d3stats")
import housescan_amd as hsk  # noqa: E402  (host-only synthetic stream)
trk = O.Tracker(O.default_config(n, omp="var:d3stats"), omp="var:d3stats")
for k in range(frames):
    trk.process(hsk.synth_depth(hsk.synth_pose(k)))
out = (C.c_longlong * 12)()
L.ora_d3_stats(out, 0)
names = ["< -4", "[-4,-2)", "[-2,-1)", "[-1,-0.5)", "[-0.5,0)", "[0,1]", "(1,1.5]", "(1.5,2]", "(2,3]", "(3,5]", "> 5", "NaN"]
tot = sum(out)
print("volume %d^3, %d frames, %d zero crossings with two valid trilinear samples; (T* - t) / step:" % (n, frames, tot))
for nm, c in zip(names, out):
    print("  %-10s %12d  %9.5f %%" % (nm, c, 100.0 * c / max(1, tot)))
kept = out[4] + out[5] + out[6]
print("  kept by the specification's [-0.5, 1.5]: %.5f %%;  by [-1, 2]: %.5f %%;  by [-2, 3]: %.5f %%" %
      (100.0 * kept / tot, 100.0 * (kept + out[3] + out[7]) / tot, 100.0 * (kept + out[3] + out[7] + out[2] + out[8]) / tot))
