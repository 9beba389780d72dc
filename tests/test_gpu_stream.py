"""GPU tests of SURVEY.md 8(f) rank 4 (a RECORDED depth stream through the tracker -- BASELINE configs[2] is "scan from
recorded stream"; the frame contract is takeDepthSnapshot's, /root/reference/housescan/HoniHelper.hs:20-36) and of
SURVEY.md 8(d)'s cfg2 trajectory report: the scripted 300-frame stream, recorded to the library's HSKD container and
replayed by hsk_track_stream (the frame feed inside the library, host frames, pipelined), at 256^3 (configs[1]) and 512^3
(configs[2]).  Every pose must equal the in-memory run's bit for bit; the trajectory error against the scripted ground
truth (ATE rmse / max, max angle) is asserted within stated bounds and printed.  Plus: reading the volume out in the
MIDDLE of a pipelined stream (frames in flight, free-space weights pending in the side table) returns the oracle's volume.
"""
import json
import os

import numpy as np
import pytest

from test_gpu_parity import assert_same_bits

pytestmark = pytest.mark.gpu

N_FRAMES = 300


def angle_deg(Ra, Rb):
    f = np.linalg.norm(Ra.astype(np.float64) - Rb.astype(np.float64))
    return float(np.degrees(2.0 * np.arcsin(min(1.0, f / (2.0 * np.sqrt(2.0))))))


@pytest.fixture(scope="module")
def recorded_stream(hsk, tmp_path_factory):
    """the scripted synthetic stream (300 frames, 10 s @ 30 Hz) recorded to an HSKD file"""
    path = str(tmp_path_factory.mktemp("hskd") / "synthetic_300.hskd")
    w = hsk.DepthStreamWriter(path)
    poses = []
    for k in range(N_FRAMES):
        p = hsk.synth_pose(k)
        poses.append(p)
        w.write(hsk.synth_depth(p))
    w.close()
    assert os.path.getsize(path) == 36 + N_FRAMES * 640 * 480 * 2
    return path, np.stack(poses)


@pytest.mark.parametrize("n,ate_rmse_mm,ate_max_mm,angle_max_deg", [(256, 4.0, 8.0, 0.25), (512, 2.5, 5.0, 0.15)])
def test_recorded_stream_replay_and_trajectory(hsk, recorded_stream, n, ate_rmse_mm, ate_max_mm, angle_max_deg):
    path, gt = recorded_stream
    rd = hsk.DepthStreamReader(path)
    assert len(rd) == N_FRAMES and (rd.w, rd.hgt) == (640, 480)
    # in memory, one synchronous call per frame
    ref = hsk.KinfuTracker(n=n)
    ref_poses, ref_ok = [], []
    for k in range(N_FRAMES):
        p, ok = ref.process_frame(rd[k])
        ref_poses.append(p.copy())
        ref_ok.append(ok)
    ref_poses = np.stack(ref_poses)
    # from the file, the frame feed inside the library (two calls: the second continues the scan)
    trk = hsk.KinfuTracker(n=n)
    pa, oka = trk.track_stream(rd, 0, 100)
    pb, okb = trk.track_stream(rd, 100, N_FRAMES - 100)
    poses = np.concatenate([pa, pb])
    ok = np.concatenate([oka, okb])
    assert list(ok) == ref_ok and not ok[0] and ok[1:].all(), "frame 0 is the untracked first frame; no frame may be lost"
    assert_same_bits(poses, ref_poses, f"{n}^3: poses of the replayed recording vs the in-memory run")
    assert_same_bits(trk.download_tsdf(), ref.download_tsdf(), f"{n}^3: TSDF after the replayed recording")
    rd.close()
    ref.close()
    trk.close()
    # trajectory against the scripted ground truth
    err = np.linalg.norm(poses[:, :3, 3].astype(np.float64) - gt[:, :3, 3].astype(np.float64), axis=1) * 1e3
    ang = np.array([angle_deg(poses[k, :3, :3], gt[k, :3, :3]) for k in range(N_FRAMES)])
    report = {"volume": n, "frames": N_FRAMES, "ate_rmse_mm": round(float(np.sqrt((err ** 2).mean())), 3),
              "ate_max_mm": round(float(err.max()), 3), "final_mm": round(float(err[-1]), 3), "angle_max_deg": round(float(ang.max()), 4)}
    print("\ntrajectory " + json.dumps(report))
    assert report["ate_rmse_mm"] <= ate_rmse_mm and report["ate_max_mm"] <= ate_max_mm and report["angle_max_deg"] <= angle_max_deg, report


def test_track_stream_errors(hsk, recorded_stream, tmp_path):
    path, _ = recorded_stream
    rd = hsk.DepthStreamReader(path)
    trk = hsk.KinfuTracker(n=64)
    with pytest.raises(hsk.KinfuError, match="frame range"):
        trk.track_stream(rd, 290, 20)
    small = hsk.KinfuTracker(hsk.default_config(64, width=320, height=240, fx=262.5, fy=262.5, cx=159.5, cy=119.5))
    with pytest.raises(hsk.KinfuError, match="frame size"):
        small.track_stream(rd, 0, 2)
    trk.submit_frame(rd[0])
    with pytest.raises(hsk.KinfuError, match="in flight"):
        trk.track_stream(rd, 1, 2)
    trk.wait_frame()
    p, ok = trk.track_stream(rd, 1, 3)   # continues the scan the submitted frame started
    assert ok.all()
    small.close()
    trk.close()
    rd.close()


@pytest.mark.parametrize("n", [128, 512])
def test_readout_in_the_middle_of_a_pipelined_stream(hsk, oracle, synth_frames, n):
    """download / cloud extraction between submissions, with a frame still in flight and free-space weights pending in
    the lane-block summaries: the volume read out is the oracle's after exactly the frames submitted so far, and the
    stream goes on to the oracle's poses"""
    cfg_o = oracle.default_config(n, omp=True)
    ot = oracle.Tracker(cfg_o, omp=True)
    trk = hsk.KinfuTracker(n=n)
    frames = [synth_frames(k)[1] for k in range(9)]
    want = [ot.process(frames[0])]
    trk.submit_frame(frames[0])
    got = [trk.wait_frame()]
    for k in range(1, 9):
        trk.submit_frame(frames[k])
        want.append(ot.process(frames[k]))
        if k in (3, 6):   # frame k is in flight (not waited for): the read-out is ordered behind it on the stream
            vol = trk.download_tsdf()
            assert_same_bits(vol, ot.volume(), f"{n}^3: volume read out with frame {k} in flight")
            if k == 6:
                cloud, total = trk.extract_cloud()
                ocloud, ototal = oracle.extract_cloud(cfg_o, np.ascontiguousarray(ot.volume()))
                assert total == ototal
                assert_same_bits(cloud, ocloud, f"{n}^3: cloud extracted with frame {k} in flight")
        got.append(trk.wait_frame())
    for k, ((p, ok), (po, oko)) in enumerate(zip(got, want)):
        assert ok == oko
        assert_same_bits(p, po, f"{n}^3: pose of frame {k} around the read-outs")
    assert_same_bits(trk.download_tsdf(), ot.volume(), f"{n}^3: final volume")
    ot.close()
    trk.close()


def test_track_stream_restarts_on_the_frame_after_a_lost_one(hsk, synth_frames, tmp_path):
    """hsk_track_stream keeps a frame in flight; when frame i loses tracking, frame i + 1 has already been
    dropped on the device -- the feed must read it again and resubmit it as the first frame of the restarted scan, so that
    poses, verdicts and the volume stay exactly those of hsk_process_frame on the same frames (lost frames back to back, a
    lost frame second to last and last included)."""
    zero = np.zeros((480, 640), np.uint16)
    seq = [synth_frames(k)[1] for k in range(6)] + [zero] + [synth_frames(k)[1] for k in range(6, 10)] + [zero, zero] + \
          [synth_frames(k)[1] for k in range(10, 14)] + [zero] + [synth_frames(14)[1]] + [zero]
    path = str(tmp_path / "with_holes.hskd")
    w = hsk.DepthStreamWriter(path)
    for d in seq:
        w.write(d)
    w.close()
    rd = hsk.DepthStreamReader(path)
    for n in (64, 128):
        ref = hsk.KinfuTracker(n=n)
        want = [ref.process_frame(d) for d in seq]
        trk = hsk.KinfuTracker(n=n)
        pa, oka = trk.track_stream(rd, 0, 7)       # ends ON the lost frame: the next call starts on the restart frame
        pb, okb = trk.track_stream(rd, 7, len(seq) - 7)
        poses, ok = np.concatenate([pa, pb]), np.concatenate([oka, okb])
        assert [bool(o) for o in ok] == [o for _, o in want], n
        assert sum(1 for _, o in want if not o) >= 9   # frame 0, four lost frames and the frame after each
        assert_same_bits(poses, np.stack([p for p, _ in want]), f"{n}^3: poses of a stream with lost frames, feed vs hsk_process_frame")
        assert_same_bits(trk.download_tsdf(), ref.download_tsdf(), f"{n}^3: TSDF after a stream with lost frames")
        one = hsk.KinfuTracker(n=n)                # ... and in ONE call
        pc, okc = one.track_stream(rd, 0, len(seq))
        assert_same_bits(pc, poses, "one call vs two")
        assert list(okc) == list(ok)
        assert_same_bits(one.download_tsdf(), ref.download_tsdf(), "one call: TSDF")
        for t in (ref, trk, one):
            t.close()
    rd.close()
