#!/usr/bin/env python3
"""How far does this build's bit-reproducible SPECIFICATION (oracle/kinfu_oracle.c, what the HIP kernels equal bit for
bit) move TSDF values and poses from Appendix-A-LITERAL arithmetic (the same file built with -DORA_LITERAL: FMA
contraction allowed, one expf per bilateral tap, plain binary64 ICP sums, LLT Cholesky, libm sinf / cosf, no D3 / D6)?

CPU only.  Two parts:
  stages   identical inputs through one stage at a time in both forms (what the arithmetic alone changes);
  tracker  the scripted synthetic stream through both trackers, frame by frame: pose difference in
           mm / degrees, both against the ground truth, and at checkpoints the TSDF |difference| histogram in LSB
           (1 LSB = tau / 32767), the fraction of observed voxels that differ, and the model maps' hit-mask difference.

usage: tools/spec_vs_literal.py [--volume 256] [--frames 300] [--every 50] [--out profiles/r03/spec_vs_literal_256.json]
The north_star's "TSDF within a stated float tolerance, trajectory within stated mm/deg" is stated from these numbers
(DESIGN.md section 4); tests/test_spec_vs_literal.py asserts the bound on a short stream.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rot_angle_deg(Ra, Rb):
    # angle of Ra^T Rb from the chord |Ra - Rb|_F = 2 sqrt(2) sin(angle / 2): accurate for tiny angles, where acos of a
    # trace is not
    f = np.linalg.norm(Ra.astype(np.float64) - Rb.astype(np.float64))
    return float(np.degrees(2.0 * np.arcsin(min(1.0, f / (2.0 * np.sqrt(2.0))))))


def pose_delta(a, b):
    return float(np.linalg.norm(a[:3, 3].astype(np.float64) - b[:3, 3].astype(np.float64)) * 1e3), rot_angle_deg(a[:3, :3], b[:3, :3])


def tsdf_delta(va, vb):
    """va, vb: int16 [Z, Y, X, 2].  Over voxels either form has observed (weight > 0)."""
    obs = (va[..., 1] > 0) | (vb[..., 1] > 0)
    n_obs = int(obs.sum())
    d = np.abs(va[..., 0].astype(np.int32) - vb[..., 0].astype(np.int32))[obs]
    band = ((np.abs(va[..., 0]) < 32767) | (np.abs(vb[..., 0]) < 32767))[obs]  # inside the truncation band in either form
    edges = [0, 1, 2, 4, 8, 16, 64, 256, 1024, 4096, 16384, 65536]
    hist = np.histogram(d, bins=edges)[0]
    only = int(((va[..., 1] > 0) != (vb[..., 1] > 0)).sum())
    wdiff = int((va[..., 1] != vb[..., 1])[obs].sum())
    db = d[band]
    return {
        "observed_voxels": n_obs,
        "differing_fraction": round(float((d > 0).mean()) if n_obs else 0.0, 6),
        "band_voxels": int(band.sum()),
        "band_differing_fraction": round(float((db > 0).mean()) if db.size else 0.0, 6),
        "abs_lsb": {"max": int(d.max()) if n_obs else 0, "mean": round(float(d.mean()), 4) if n_obs else 0.0,
                    "p50_band": int(np.percentile(db, 50)) if db.size else 0, "p99_band": int(np.percentile(db, 99)) if db.size else 0,
                    "p999_band": int(np.percentile(db, 99.9)) if db.size else 0},
        "hist_lsb_edges": edges, "hist_lsb_counts": [int(x) for x in hist],
        "observed_in_one_form_only": only, "weights_differ": wdiff,
    }


def maps_delta(va, na, vb, nb, cell_mm):
    ha, hb = ~np.isnan(va[0]), ~np.isnan(vb[0])
    both = ha & hb
    dv = np.linalg.norm((va[:, both] - vb[:, both]).astype(np.float64), axis=0) * 1e3
    nboth = both & ~np.isnan(na[0]) & ~np.isnan(nb[0])
    cosn = np.clip((na[:, nboth].astype(np.float64) * nb[:, nboth].astype(np.float64)).sum(0), -1, 1)
    ang = np.degrees(np.arccos(cosn))
    return {"hit_pixels_spec": int(ha.sum()), "hit_pixels_literal": int(hb.sum()), "hit_mask_differs": int((ha != hb).sum()),
            "vertex_mm": {"mean": round(float(dv.mean()), 4), "p99": round(float(np.percentile(dv, 99)), 4), "max": round(float(dv.max()), 4)} if dv.size else None,
            "vertex_max_in_cells": round(float(dv.max()) / cell_mm, 4) if dv.size else None,
            "normal_deg": {"mean": round(float(ang.mean()), 4), "p99": round(float(np.percentile(ang, 99)), 4), "max": round(float(ang.max()), 4)} if ang.size else None}


def stage_section(n, hsk, O):
    """one stage at a time on identical inputs: what the arithmetic alone changes"""
    out = {}
    cfg_s, cfg_l = O.default_config(n, omp=True), O.default_config(n, omp="literal")
    pose0, pose1 = hsk.synth_pose(0), hsk.synth_pose(5)
    d0, d1 = hsk.synth_depth(pose0), hsk.synth_depth(pose1)
    # bilateral
    bs, bl = O.bilateral(cfg_s, d1, omp=True), O.bilateral(cfg_l, d1, omp="literal")
    dd = np.abs(bs.astype(np.int32) - bl.astype(np.int32))
    out["bilateral_mm"] = {"pixels": int(dd.size), "differing": int((dd > 0).sum()), "max": int(dd.max()),
                           "differing_excluding_last_row_col": int((dd[:-1, :-1] > 0).sum()),
                           "note": "literal: one expf per tap, window exclusive of the image's last row / column"}
    # integrate: two frames from an empty volume, same poses
    X = n
    vs = np.zeros((X, X, X, 2), np.int16)
    vl = np.zeros((X, X, X, 2), np.int16)
    for (p, d) in ((pose0, d0), (pose1, d1)):
        sc = O.scale_depth(cfg_s, d)
        O.integrate(cfg_s, vs, sc, p, omp=True)
        O.integrate(cfg_l, vl, sc, p, omp="literal")
    out["integrate_two_frames_same_poses"] = tsdf_delta(vs, vl)
    # raycast of the SAME volume from the same pose
    vm_s, nm_s, _, _ = O.raycast(cfg_s, vs, pose1, omp=True)
    vm_l, nm_l, _, _ = O.raycast(cfg_l, vs, pose1, omp="literal")
    out["raycast_same_volume"] = maps_delta(vm_s, nm_s, vm_l, nm_l, 3000.0 / n)
    # ICP sums + solve on the same maps: current maps of frame 5 against the model maps raycast above, estimate = frame 4's pose
    vcur = O.vmap(cfg_s, O.bilateral(cfg_s, d1, omp=True), 0)
    ncur = O.nmap(vcur)
    est = hsk.synth_pose(4)
    ss, _ = O.icp_accumulate(cfg_s, 0, vcur, ncur, vm_s, nm_s, est, pose1, omp=True)
    sl, _ = O.icp_accumulate(cfg_l, 0, vcur, ncur, vm_s, nm_s, est, pose1, omp="literal")
    xs, oks = O.icp_solve(ss, omp=True)
    xl, okl = O.icp_solve(sl, omp="literal")
    xl_same, _ = O.icp_solve(ss, omp="literal")
    out["icp_same_maps"] = {"sums_max_rel_diff": float(np.max(np.abs(ss - sl) / np.maximum(np.abs(ss), 1e-300))),
                            "increment_spec": [float(v) for v in xs], "increment_literal": [float(v) for v in xl],
                            "increment_max_abs_diff": float(np.max(np.abs(xs.astype(np.float64) - xl.astype(np.float64)))),
                            "increment_max_abs_diff_same_sums_LDLT_vs_LLT": float(np.max(np.abs(xs.astype(np.float64) - xl_same.astype(np.float64)))),
                            "both_solved": bool(oks and okl)}
    return out


VARIANTS = [("fma", "", "fast", "FMA contraction allowed, nothing else (gcc -ffp-contract=fast -mfma)"),
            ("d1", "-DORA_LIT_D1", "off", "D1: bilateral with one expf per tap, exclusive window clip, zero centre filtered"),
            ("d3", "-DORA_LIT_D3", "off", "D3: extrapolated raycast hit times accepted"),
            ("d4", "-DORA_LIT_D4", "off", "D4: plain binary64 ICP sums, LLT Cholesky, libm sinf / cosf"),
            ("d6", "-DORA_LIT_D6", "off", "D6: 1 / z < 0 as integrate's in-front test"),
            ("inc", "-DORA_LIT_INC", "off", "integrate: camera coordinates advanced incrementally along z (A.4's closing note; includes D6)")]


def attribute_section(n, frames, hsk, O, log=None):
    """the tracker with ONE deviation taken back at a time, against the specification: which of them moves the result"""
    out = {}
    for name, defs, contract, what in VARIANTS:
        O.build_variant(name, defs, contract)
        r = tracker_section(n, frames, frames, hsk, O, log=None, other="var:" + name)
        cp = r["checkpoints"][str(frames)]
        out[name] = {"what": what, "pose_translation_mm_max": r["spec_vs_literal_pose"]["translation_mm"]["max"],
                     "pose_rotation_deg_max": r["spec_vs_literal_pose"]["rotation_deg"]["max"],
                     "tsdf_differing_fraction": cp["differing_fraction"], "tsdf_band_differing_fraction": cp["band_differing_fraction"],
                     "tsdf_p99_band_lsb": cp["abs_lsb"]["p99_band"], "tsdf_max_lsb": cp["abs_lsb"]["max"],
                     "hit_mask_differs": cp["model_maps"]["hit_mask_differs"]}
        if log:
            log("  only %-3s: %s" % (name, json.dumps(out[name])))
    return out


def tracker_section(n, frames, every, hsk, O, log=None, other="literal"):
    cfg_s, cfg_l = O.default_config(n, omp=True), O.default_config(n, omp=other)
    ts, tl = O.Tracker(cfg_s, omp=True), O.Tracker(cfg_l, omp=other)
    per = []
    checkpoints = {}
    lost_s = lost_l = 0
    t0 = time.time()
    for k in range(frames):
        gt = hsk.synth_pose(k)
        d = hsk.synth_depth(gt)
        ps, oks = ts.process(d)
        pl, okl = tl.process(d)
        if k > 0:
            lost_s += (not oks)
            lost_l += (not okl)
        dmm, ddeg = pose_delta(ps, pl)
        gs = pose_delta(ps, gt)
        gl = pose_delta(pl, gt)
        per.append((dmm, ddeg, gs[0], gs[1], gl[0], gl[1]))
        if (k + 1) % every == 0 or k == frames - 1:
            cp = tsdf_delta(ts.volume(), tl.volume())
            cp["model_maps"] = maps_delta(ts.model_map(2, 0), ts.model_map(3, 0), tl.model_map(2, 0), tl.model_map(3, 0), 3000.0 / n)
            cp["pose_diff_mm_deg"] = [round(dmm, 5), round(ddeg, 6)]
            checkpoints[str(k + 1)] = cp
            if log:
                log("  frame %d: pose diff %.4f mm %.5f deg; TSDF differing %.4f of observed, %.4f of band, p99 band %d LSB  (%.0f s)" % (
                    k + 1, dmm, ddeg, cp["differing_fraction"], cp["band_differing_fraction"], cp["abs_lsb"]["p99_band"], time.time() - t0))
    ts.close()
    tl.close()
    a = np.array(per)
    tau_mm = float(O.tau(cfg_s)) * 1e3

    def ate(col):
        return {"rmse_mm": round(float(np.sqrt((a[:, col] ** 2).mean())), 4), "max_mm": round(float(a[:, col].max()), 4)}
    return {
        "volume": n, "frames": frames, "tau_mm": tau_mm, "lsb_in_micrometres": round(tau_mm * 1e3 / 32767, 4),
        "lost_frames": {"spec": int(lost_s), "literal": int(lost_l)},
        "spec_vs_literal_pose": {"translation_mm": {"max": round(float(a[:, 0].max()), 5), "rmse": round(float(np.sqrt((a[:, 0] ** 2).mean())), 5),
                                                    "final": round(float(a[-1, 0]), 5)},
                                 "rotation_deg": {"max": round(float(a[:, 1].max()), 6), "final": round(float(a[-1, 1]), 6)}},
        "vs_ground_truth": {"spec": {"ate": ate(2), "max_angle_deg": round(float(a[:, 3].max()), 5)},
                            "literal": {"ate": ate(4), "max_angle_deg": round(float(a[:, 5].max()), 5)}},
        "per_frame_every_10": [[k] + [round(float(v), 5) for v in a[k]] for k in range(0, frames, 10)],
        "per_frame_columns": ["frame", "spec-literal mm", "spec-literal deg", "spec-gt mm", "spec-gt deg", "literal-gt mm", "literal-gt deg"],
        "checkpoints": checkpoints,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--volume", type=int, default=256)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--stages-volume", type=int, default=128)
    ap.add_argument("--no-stages", action="store_true")
    ap.add_argument("--attribute", type=int, default=0, metavar="FRAMES",
                    help="also run the tracker with one deviation taken back at a time over this many frames")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import housescan_amd as hsk   # the synthetic stream's generator lives in the library (host-only entry points)
    from oracle import oracle as O
    O.build()
    res = {"tool": "tools/spec_vs_literal.py", "literal_build": "gcc -O3 -ffp-contract=fast -mfma -DORA_LITERAL (oracle/Makefile: literal)"}
    if not args.no_stages:
        res["stages"] = stage_section(args.stages_volume, hsk, O)
        print("stages:", json.dumps(res["stages"])[:2000], flush=True)
    if args.attribute:
        res["attribution"] = {"volume": args.volume, "frames": args.attribute,
                              "one_deviation_at_a_time": attribute_section(args.volume, args.attribute, hsk, O, log=lambda s: print(s, flush=True))}
    res["tracker"] = tracker_section(args.volume, args.frames, args.every, hsk, O, log=lambda s: print(s, flush=True))
    t = res["tracker"]
    print(json.dumps({k: t[k] for k in ("volume", "frames", "lost_frames", "spec_vs_literal_pose", "vs_ground_truth")}))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
