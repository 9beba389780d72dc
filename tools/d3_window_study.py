#!/usr/bin/env python3
"""D3 study, part two: the specification with a WIDER hit-time acceptance window against the build that accepts every hit time
(`-DORA_LIT_D3`), CPU only, 256^3 x 300 frames of the scripted stream.  Says what a wider window (and the wider slab halo that
goes with it) would buy; results in profiles/r04/spec_vs_literal.md.   usage: tools/d3_window_study.py"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import numpy as np
from oracle import oracle as O
import housescan_amd as hsk
import spec_vs_literal as S
n, frames = 256, 300
O.build_variant("d3", "-DORA_LIT_D3")
for name, lo, hi in (("w23", 2.0, 3.0), ("w1525", 1.5, 2.5)):
    O.build_variant(name, "-DORA_D3_LO=%.1ff -DORA_D3_HI=%.1ff" % (lo, hi))
    a, b = O.Tracker(O.default_config(n, omp="var:" + name), omp="var:" + name), O.Tracker(O.default_config(n, omp="var:d3"), omp="var:d3")
    worst = [0.0, 0.0]
    for k in range(frames):
        d = hsk.synth_depth(hsk.synth_pose(k))
        pa, _ = a.process(d); pb, _ = b.process(d)
        mm, deg = S.pose_delta(pa, pb)
        worst = [max(worst[0], mm), max(worst[1], deg)]
    print("window [-%.1f, %.1f] vs every hit time accepted, %d^3 x %d: max %.4f mm / %.5f deg" % (lo, hi, n, frames, worst[0], worst[1]), flush=True)
    a.close(); b.close()
