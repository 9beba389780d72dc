"""GPU parity tests at the BASELINE.json configurations' own shapes (SURVEY.md 8(d) "Per-config shapes") and of the
paths round 1 left without oracle coverage:

  * configs[3] volume, 1024^3, BIT-EXACT against the oracle (this size alone takes the non-temporal streaming path of
    integrate pass A and the 32^3-voxel bricks of the raycast's skip structure);
  * configs[3] partition at its own size: eight z-slab contexts of the 1024^3 volume, composited the way the collectives
    do (MIN of step keys, SUM of bit patterns), bit-identical to the single 1024^3 context;
  * a10, the integration gate: a tracker run in which some frames gate off;
  * configs[4]: four concurrent 512^3 contexts driven from four OS threads (SURVEY.md 8(b) threading contract:
    "distinct contexts may be driven from different OS threads"), each bit-identical to its own sequential run.
"""
import threading

import numpy as np
import pytest

from test_gpu_parity import assert_same_bits

pytestmark = pytest.mark.gpu


def test_config3_1024_integrate_and_raycast_bit_exact(hsk, oracle, synth_frames):
    """1024^3 (4 GiB): two integrations and a raycast, every voxel / pixel / step key against the OpenMP oracle"""
    n = 1024
    cfg_o = oracle.default_config(n, omp=True)
    trk = hsk.KinfuTracker(n=n)
    vol = np.zeros((n, n, n, 2), np.int16)
    for k in (0, 9):
        pose, depth = synth_frames(k)
        scaled = oracle.scale_depth(cfg_o, depth)
        n_upd = oracle.integrate(cfg_o, vol, scaled, pose, omp=True)
        assert trk.count_updates(depth, pose) == n_upd
        trk.integrate(depth, pose)
    got = trk.download_tsdf()
    for z0 in range(0, n, 128):   # slab by slab: a mismatch is reported with its position, and no 4 GiB temporaries
        assert_same_bits(got[z0:z0 + 128], vol[z0:z0 + 128], f"1024^3 tsdf planes {z0}..{z0 + 127}")
    assert (vol[..., 1] == 2).sum() > 0.05 * n ** 3
    del got
    for pose in (synth_frames(9)[0], synth_frames(30)[0]):
        vm, nm, keys = trk.raycast(pose, want_keys=True)
        ovm, onm, okeys, _ = oracle.raycast(cfg_o, vol, pose, omp=True)
        assert np.array_equal(keys, okeys)
        assert_same_bits(vm, ovm, "1024^3 raycast vmap")
        assert_same_bits(nm, onm, "1024^3 raycast nmap")
        assert (~np.isnan(vm[0])).mean() > 0.5
    trk.close()


def test_config3_eight_slabs_of_1024_compose_to_single_volume(hsk, synth_frames):
    """BASELINE configs[3]'s partition at its own size, on one GPU: eight z-slab contexts (128 owned planes + halo each)
    of the 1024^3 volume, driven through the hsk_mgpu_* building blocks and composited as the collectives do; poses,
    owned planes and model maps bit-identical to the single 1024^3 context"""
    import torch
    from housescan_amd.sharded import HipSlabEngine, slab_halo, slab_range
    n, world = 1024, 8
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(dev)
    ref = hsk.KinfuTracker(n=n)
    engines = []
    for r in range(world):
        cfg = hsk.default_config(n)
        cfg.own_z0, cfg.own_z1 = slab_range(r, world, n)
        cfg.halo = slab_halo(max(0.03, 2.1 * 3.0 / n), 3.0 / n)
        engines.append(HipSlabEngine(hsk.KinfuTracker(cfg), torch, dev, side))
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for k in range(4):
            _, depth = synth_frames(k)
            pref, okref = ref.process_frame(depth)
            d_dev = torch.from_numpy(depth.view(np.int16)).to(dev)
            first = engines[0].frame_index() == 0
            if k < 2:
                for e in engines:
                    e.frame_begin(d_dev)
                if not first:
                    for e in engines:
                        e.icp_replicated()
                        e.integrate()
                    keys = [e.raycast_local().clone() for e in engines]
            else:
                keys = [e.frame_front(d_dev).clone() for e in engines]   # the one-call frame front
            if first:
                outs = [e.frame_end(None, None) for e in engines]
            else:
                kmin = torch.stack(keys).amin(dim=0)
                bits = [e.raycast_resolve(kmin).clone() for e in engines]
                bsum = torch.stack(bits).sum(dim=0, dtype=torch.int32)
                outs = [e.frame_end(kmin, bsum) for e in engines]
            for p, ok in outs:
                assert ok == okref
                assert_same_bits(p, pref, f"1024^3 slab pose frame {k}")
    torch.cuda.synchronize()
    full = ref.download_tsdf()
    for r, e in enumerate(engines):
        z0, z1 = slab_range(r, world, n)
        got = e.t.download_tsdf()
        assert_same_bits(got[z0 - e.t.stored_z0:z1 - e.t.stored_z0], full[z0:z1], f"slab {r} owned planes")
        # the redundantly integrated halo planes agree too (no halo exchange is ever needed)
        assert_same_bits(got, full[e.t.stored_z0:e.t.stored_z0 + e.t.stored_nz], f"slab {r} stored planes")
        for level in range(3):
            assert_same_bits(e.t.download_map(2, level), ref.download_map(2, level), f"slab {r} model vmap {level}")
            assert_same_bits(e.t.download_map(3, level), ref.download_map(3, level), f"slab {r} model nmap {level}")
        e.t.close()
    ref.close()


def _gate_metric(p, q):
    """(rotation angle + translation) / 2 between two poses, in float64 (only used to design the test sequence)"""
    c = (np.trace(p[:3, :3].astype(np.float64).T @ q[:3, :3].astype(np.float64)) - 1) / 2
    return (np.arccos(np.clip(c, -1, 1)) + np.linalg.norm(p[:3, 3].astype(np.float64) - q[:3, 3])) / 2


@pytest.mark.parametrize("api", ["sync", "submit"])
def test_integration_gate_some_frames_gate_off(hsk, oracle, synth_frames, api):
    """a10: integrate only when (|rodrigues(R^-1 R_prev)| + |t - t_prev|) / 2 >= threshold.
    The stream repeats frames (camera at rest => gate off) between moving frames (gate on); poses, TSDF and model maps
    bit-identical to the oracle, the weights count exactly the integrated frames, and a lost frame resets"""
    n = 128
    thr = 0.004
    seq = [0, 1, 2, 2, 2, 3, 4, 4, 5, 6]
    # the design of the sequence: moving frames are above the threshold, repeated frames far below it
    assert _gate_metric(hsk.synth_pose(1), hsk.synth_pose(2)) > 1.5 * thr
    ot = oracle.Tracker(oracle.default_config(n, move_thresh=thr), omp=True)
    trk = hsk.KinfuTracker(hsk.default_config(n, integrate_move_thresh=thr))
    passed, prev = 1, None   # the first frame always integrates
    for i, k in enumerate(seq):
        depth = synth_frames(k)[1]
        po, oko = ot.process(depth)
        if api == "sync":
            ph, okh = trk.process_frame(depth)
        else:
            trk.submit_frame(depth)
            ph, okh = trk.wait_frame()
        assert oko == okh == (i > 0)
        assert_same_bits(ph, po, f"gated pose, step {i} (frame {k})")
        if prev is not None:
            m = _gate_metric(po, prev)
            assert not (0.7 * thr < m < 1.3 * thr), "test design: keep the metric away from the threshold"
            passed += m >= thr
        prev = po
    assert passed == 1 + 6      # frames 1, 2, 3, 4, 5, 6 moved; the three repeats did not
    vol = trk.download_tsdf()
    assert_same_bits(vol, ot.volume(), "gated tsdf")
    assert int(vol[..., 1].max()) == passed, "weights count the integrated frames only"
    for level in range(3):
        assert_same_bits(trk.download_map(2, level), ot.model_map(2, level), f"gated model vmap {level}")
        assert_same_bits(trk.download_map(3, level), ot.model_map(3, level), f"gated model nmap {level}")
    # lost branch in gated mode: an empty frame makes the system singular; both sides reset to the initial pose
    zero = np.zeros_like(synth_frames(0)[1])
    po, oko = ot.process(zero)
    ph, okh = trk.process_frame(zero)
    assert not oko and not okh
    assert_same_bits(ph, po, "pose after the lost frame")
    assert not trk.download_tsdf().any()
    trk.close()
    ot.close()


def test_config5_four_concurrent_512_contexts_from_four_threads(hsk, synth_frames):
    """BASELINE configs[4] shape on one device: four 512^3 contexts, each fed its own trajectory from its own OS thread
    at the same time (ctypes releases the GIL inside the C ABI calls); every context must reproduce, bit for bit, what
    the same frames give when that context runs alone"""
    n, nctx, nframes = 512, 4, 10
    streams = [[synth_frames(20 * c + k)[1] for k in range(nframes)] for c in range(nctx)]

    def run(trk, frames, out, pipelined):
        res = []
        if pipelined:     # two of the four threads use the submit / wait pair, two the synchronous call
            res.append(trk.process_frame(frames[0]))
            trk.submit_frame(frames[1])
            for d in frames[2:]:
                trk.submit_frame(d)
                res.append(trk.wait_frame())
            res.append(trk.wait_frame())
        else:
            for d in frames:
                res.append(trk.process_frame(d))
        out.append(res)

    # sequential reference runs, one context at a time
    want = []
    for c in range(nctx):
        trk = hsk.KinfuTracker(n=n)
        out = []
        run(trk, streams[c], out, False)
        want.append((out[0], trk.download_tsdf()))
        trk.close()
    # concurrent runs
    trks = [hsk.KinfuTracker(n=n) for _ in range(nctx)]
    outs = [[] for _ in range(nctx)]
    errs = []

    def guarded(c):
        try:
            run(trks[c], streams[c], outs[c], c % 2 == 1)
        except Exception as e:  # noqa: BLE001
            errs.append((c, e))

    threads = [threading.Thread(target=guarded, args=(c,)) for c in range(nctx)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for c in range(nctx):
        poses_ref, vol_ref = want[c]
        for k, ((p, ok), (pr, okr)) in enumerate(zip(outs[c][0], poses_ref)):
            assert ok == okr == (k > 0), (c, k)
            assert_same_bits(p, pr, f"context {c} frame {k}: concurrent vs alone")
        assert_same_bits(trks[c].download_tsdf(), vol_ref, f"context {c} tsdf: concurrent vs alone")
        trks[c].close()
    # the four trajectories really differ
    assert not np.array_equal(want[0][0][-1][0], want[1][0][-1][0])


def test_tracker_with_every_parameter_off_its_default(hsk, oracle):
    """every field of the configuration away from its default at once -- anisotropic cells in a non-cubic volume, a longer
    truncation distance, fx != fy and an off-centre principal point on a 320x240 image, other ICP iteration counts and
    gates, a start pose that is not the default -- through the tracker, synchronously and pipelined: poses, verdicts, TSDF
    and model maps the oracle's, bit for bit (the kernels take these as data; the tests elsewhere vary them one or two at
    a time)"""
    W, H = 320, 240
    fx, fy, cx, cy = 262.5 * 1.04, 262.5 * 0.97, W / 2 - 0.5 + 3.25, H / 2 - 0.5 - 2.5
    vol, size = (160, 96, 200), (3.0, 2.4, 3.2)            # cells 18.75 / 25 / 16 mm
    iters = (6, 3, 2)
    start = hsk.synth_pose(0).copy()
    start[:3, 3] += np.array([0.02, -0.03, 0.01], np.float32)
    kw_o = dict(vol=vol, size=size, trunc=0.07, W=W, H=H, fx=fx, fy=fy, cx=cx, cy=cy, icp_iters=iters, dist_thresh=0.07,
                angle_thresh=float(np.sin(np.radians(15.0))), init_R=start[:3, :3], init_t=start[:3, 3])
    kw_h = dict(vol_y=vol[1], vol_z=vol[2], own_z1=vol[2], vol_size_m=size, trunc_dist_m=0.07, width=W, height=H, fx=fx, fy=fy, cx=cx, cy=cy,
                icp_iters=iters, icp_dist_thresh_m=0.07, icp_angle_thresh_sin=float(np.sin(np.radians(15.0))), init_pose=start)
    cfg_o = oracle.default_config(vol[0], **kw_o)
    ot = oracle.Tracker(cfg_o, omp=True)
    trk = hsk.KinfuTracker(hsk.default_config(vol[0], **kw_h))
    pipe = hsk.KinfuTracker(hsk.default_config(vol[0], **kw_h))
    frames = [hsk.synth_depth(hsk.synth_pose(2 * k), W, H, fx, fy, cx, cy) for k in range(10)]
    want = []
    for k, d in enumerate(frames):
        po, oko = ot.process(d)
        ph, okh = trk.process_frame(d)
        assert oko == okh, k
        assert_same_bits(ph, po, f"pose frame {k}")
        want.append((po, oko))
    assert sum(ok for _, ok in want) >= 8                   # it does track
    assert_same_bits(trk.download_tsdf(), ot.volume(), "TSDF, every parameter off its default")
    for kind in (2, 3):
        for level in range(3):
            assert_same_bits(trk.download_map(kind, level), ot.model_map(kind, level), f"model map {kind} level {level}")
    pipe.submit_frame(frames[0])
    got = []
    for d in frames[1:]:
        pipe.submit_frame(d)
        got.append(pipe.wait_frame())
    got.append(pipe.wait_frame())
    for k, ((p, ok), (po, oko)) in enumerate(zip(got, want)):
        assert ok == oko
        assert_same_bits(p, po, f"pipelined pose frame {k}")
    assert_same_bits(pipe.download_tsdf(), ot.volume(), "pipelined TSDF")
    for t in (trk, pipe):
        t.close()
    ot.close()
