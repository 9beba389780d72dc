"""The z-slab group over MORE THAN ONE device (BASELINE configs[3] / [4]): ncclCommInitAll over two devices
in one process, and the rank form -- two OS processes, one GPU each, a real two-rank RCCL communicator built from
hsk_group_unique_id.  These need a box with at least two GPUs and are SKIPPED on the one-GPU boxes this repository has
been developed on: the paths below have never run on hardware.  RCCL refuses two ranks on one
device ("Duplicate GPU detected"), so the two-process form cannot be emulated on a single GPU.
"""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from test_gpu_parity import assert_same_bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_devices():
    import torch
    return torch.cuda.device_count()


needs_two = pytest.mark.skipif(_n_devices() < 2, reason="needs at least two GPUs")


def _pipelined(t, frames):
    res = []
    t.submit_frame(frames[0])
    for d in frames[1:]:
        t.submit_frame(d)
        res.append(t.wait_frame())
    res.append(t.wait_frame())
    return res


@needs_two
@pytest.mark.parametrize("flags", [0, 2, 4])   # RCCL composites, + all-reduced ICP, the direct exchange over peer access
def test_group_over_two_devices_single_process(hsk, synth_frames, flags):
    n = 128
    frames = [synth_frames(k)[1] for k in range(8)]
    ref = hsk.KinfuTracker(n=n)
    want = _pipelined(ref, frames)
    grp = hsk.KinfuGroup(hsk.default_config(n), device_ids=[0, 1], flags=flags)
    got = _pipelined(grp, frames)
    for k, ((p, ok), (pr, okr)) in enumerate(zip(got, want)):
        assert ok == okr
        assert_same_bits(p, pr, f"two-device group pose frame {k} (flags {flags})")
    assert_same_bits(grp.download_tsdf(), ref.download_tsdf(), "two-device group tsdf")
    grp.close()
    ref.close()


RANK_SCRIPT = textwrap.dedent("""
    import os, sys, time
    import numpy as np
    sys.path.insert(0, {root!r})
    import housescan_amd as hsk
    rank, idfile, n = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    if rank == 0:
        uid = hsk.KinfuGroup.unique_id()
        open(idfile + ".tmp", "wb").write(uid)
        os.rename(idfile + ".tmp", idfile)
    else:
        while not os.path.exists(idfile):
            time.sleep(0.05)
        uid = open(idfile, "rb").read()
    grp = hsk.KinfuGroup(hsk.default_config(n, device_id=rank), rank=rank, world=2, comm_id=uid, flags=int(sys.argv[4]))
    assert grp.ranks_seen() == 2
    frames = [hsk.synth_depth(hsk.synth_pose(k)) for k in range(8)]
    grp.submit_frame(frames[0])
    out = []
    for d in frames[1:]:
        grp.submit_frame(d)
        out.append(grp.wait_frame())
    out.append(grp.wait_frame())
    for p, ok in out:
        print("pose", int(ok), np.ascontiguousarray(p, np.float32).tobytes().hex())
    vol = grp.download_tsdf()
    np.save(idfile + ".vol%d.npy" % rank, vol)
    grp.close()
    print("done")
""")


@needs_two
@pytest.mark.parametrize("flags", [0, 2, 4])   # 4: hipIpc-mapped peer buffers across two DEVICES, flags raised over xGMI
def test_rank_form_two_processes_two_devices(hsk, synth_frames, tmp_path, flags):
    n = 128
    idfile = str(tmp_path / "comm_id")
    script = RANK_SCRIPT.format(root=ROOT)
    env = dict(os.environ, HSK_FRAME_TIMEOUT_S="30")
    procs = [subprocess.Popen([sys.executable, "-c", script, str(r), idfile, str(n), str(flags)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and so.strip().endswith("done"), (r, se[-2000:])
    ref = hsk.KinfuTracker(n=n)
    want = _pipelined(ref, [synth_frames(k)[1] for k in range(8)])
    for r, (so, _) in enumerate(outs):
        rows = [ln.split() for ln in so.splitlines() if ln.startswith("pose ")]
        assert len(rows) == len(want)
        for k, (row, (pr, okr)) in enumerate(zip(rows, want)):
            assert int(row[1]) == int(okr)
            assert bytes.fromhex(row[2]) == np.ascontiguousarray(pr, np.float32).tobytes(), f"rank {r} pose frame {k}"
    full = ref.download_tsdf()
    got = np.load(idfile + ".vol0.npy") | np.load(idfile + ".vol1.npy")   # each rank fills only the planes it owns
    assert_same_bits(got, full, "the two ranks' owned planes together")
    ref.close()


@needs_two
def test_bench_bare_on_two_devices():
    """`python bench.py --gpus 2` as the driver would type it, on two real devices: every form must run, match the single
    context and see two ranks"""
    import json
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "12", "--warmup", "3", "--volume", "256"], cwd=ROOT,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["launcher"]["failed_forms"] == {}, out["launcher"]
    for f in ("rccl", "rccl_icp_allreduce", "direct"):
        assert out["forms"][f]["matches_single_gpu"] is True and out["forms"][f]["ranks_seen"] == 2, (f, out["forms"][f])
    assert out["rooms_weak"]["rooms"] == 2 and out["matches_single_gpu"] is True
