"""Multi-GPU behind the C ABI (SURVEY.md 8(b) n_devices / device_ids, 8(e)): `hsk_group_*` runs the z-slab frame loop and
its collectives inside the library.  One GPU is enough to test it: several slabs on device 0 composite through the
library's own kernels, and HSK_GROUP_FORCE_RCCL sends the same data through ncclAllReduce (a one-rank communicator).
Everything must be bit-identical to a single context.  The last test drives the library from a plain C program."""
import os
import subprocess

import numpy as np
import pytest

from test_gpu_parity import assert_same_bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reference(hsk, synth_frames, n, frames):
    ref = hsk.KinfuTracker(n=n)
    poses = [ref.process_frame(synth_frames(k)[1]) for k in frames]
    return ref, poses


@pytest.mark.parametrize("slabs,flags", [(2, 0), (3, 2), (3, 1), (2, 3)])
def test_group_matches_single_context(hsk, synth_frames, slabs, flags):
    """slabs on one device (flags: 1 = collectives through RCCL, 2 = ICP sums all-reduced every iteration), synchronous
    and pipelined calls mixed: poses, the owned planes of every slab and the model maps equal the single context's"""
    n, frames = 64, list(range(9))
    ref, want = _reference(hsk, synth_frames, n, frames)
    grp = hsk.KinfuGroup(hsk.default_config(n), device_ids=[0] * slabs, flags=flags)
    assert grp.n_slabs() == slabs
    got = []
    for k in frames[:4]:
        got.append(grp.process_frame(synth_frames(k)[1]))
    grp.submit_frame(synth_frames(4)[1])
    for k in frames[5:]:
        grp.submit_frame(synth_frames(k)[1])
        got.append(grp.wait_frame())
    got.append(grp.wait_frame())
    for k, ((p, ok), (pr, okr)) in enumerate(zip(got, want)):
        assert ok == okr == (k > 0)
        assert_same_bits(p, pr, f"group pose frame {k} ({slabs} slabs, flags {flags})")
    assert_same_bits(grp.download_tsdf(), ref.download_tsdf(), "group tsdf")
    for i in range(slabs):
        s = grp.slab(i)
        for level in range(3):
            assert_same_bits(s.download_map(2, level), ref.download_map(2, level), f"slab {i} model vmap {level}")
            assert_same_bits(s.download_map(3, level), ref.download_map(3, level), f"slab {i} model nmap {level}")
    grp.close()
    ref.close()


@pytest.mark.parametrize("stream,slabs,flags", [("noise", 3, 4), ("holes", 8, 4), ("noise", 2, 0)])
def test_group_on_streams_with_holes(hsk, stream, slabs, flags):
    """round 6: the light class of pass A / pass B (free space over holes in the depth image: queue tails, pending counts in the
    entries, summary bytes rewritten by pass B) inside SLAB contexts -- stored planes that start at own_z0 - halo, slabs thinner
    than a pass-A chunk -- and the fused frame end (k_adopt_pyramid): a group on the noise run and on the sensor-holes stream,
    256^3, direct exchange (4) and staged composites (0), equals one context: poses, every owned plane, the model maps"""
    n, count = 256, 8
    frames = hsk.synth_noisy_frames(count)[1] if stream == "noise" else hsk.synth_sensor_frames(count, absorbing=True)[1]
    ref = hsk.KinfuTracker(n=n)
    want = [ref.process_frame(d) for d in frames]
    assert ref.integrate_light_entries() > 1000, "the stream must exercise the light class"
    grp = hsk.KinfuGroup(hsk.default_config(n), device_ids=[0] * slabs, flags=flags)
    got = [grp.process_frame(frames[0]), grp.process_frame(frames[1])]
    grp.submit_frame(frames[2])
    for d in frames[3:]:
        grp.submit_frame(d)
        got.append(grp.wait_frame())
    got.append(grp.wait_frame())
    for k, ((p, ok), (pr, okr)) in enumerate(zip(got, want)):
        assert ok == okr == (k > 0)
        assert_same_bits(p, pr, f"{stream} group pose frame {k} ({slabs} slabs, flags {flags})")
    assert_same_bits(grp.download_tsdf(), ref.download_tsdf(), f"{stream} group tsdf ({slabs} slabs)")
    assert sum(grp.slab(i).integrate_light_entries() for i in range(slabs)) > 1000, "the slabs' own light class must have run"
    for i in (0, slabs - 1):
        s = grp.slab(i)
        for level in range(3):
            assert_same_bits(s.download_map(2, level), ref.download_map(2, level), f"slab {i} model vmap {level}")
            assert_same_bits(s.download_map(3, level), ref.download_map(3, level), f"slab {i} model nmap {level}")
    grp.close()
    ref.close()


def test_group_eight_slabs_256(hsk, synth_frames):
    """configs[3]'s eight-way partition through the group call (256^3 here; the 1024^3 partition is composed slab by
    slab in test_gpu_configs.py)"""
    n, frames = 256, list(range(4))
    ref, want = _reference(hsk, synth_frames, n, frames)
    grp = hsk.KinfuGroup(hsk.default_config(n), device_ids=[0] * 8)
    for k, (pr, okr) in zip(frames, want):
        p, ok = grp.process_frame(synth_frames(k)[1])
        assert ok == okr
        assert_same_bits(p, pr, f"8-slab pose frame {k}")
    assert_same_bits(grp.download_tsdf(), ref.download_tsdf(), "8-slab tsdf")
    grp.close()
    ref.close()


def test_group_rank_form_and_tracking_loss(hsk, synth_frames):
    """the one-process-per-GPU constructor (world of one rank, communicator from a unique id) and the loss / reset path"""
    n = 64
    ref = hsk.KinfuTracker(n=n)
    grp = hsk.KinfuGroup(hsk.default_config(n), rank=0, world=1, comm_id=hsk.KinfuGroup.unique_id(), flags=hsk.GROUP_FORCE_RCCL)
    zero = np.zeros_like(synth_frames(0)[1])
    seq = [synth_frames(0)[1], synth_frames(1)[1], zero, synth_frames(2)[1], synth_frames(3)[1]]
    for i, d in enumerate(seq):
        pr, okr = ref.process_frame(d)
        p, ok = grp.process_frame(d)
        assert ok == okr, i
        assert_same_bits(p, pr, f"rank-form pose step {i}")
    assert_same_bits(grp.download_tsdf(), ref.download_tsdf(), "rank-form tsdf after a loss and a restart")
    with pytest.raises(hsk.KinfuError, match="size"):
        grp.process_frame(np.zeros((10, 10), np.uint16))
    grp.reset()
    assert not grp.download_tsdf().any()
    grp.close()
    ref.close()


@pytest.mark.parametrize("slabs,flags", [(2, 0), (3, 1)])
def test_group_pipelined_loss_restarts_behind_the_frames_in_flight(hsk, synth_frames, slabs, flags):
    """a frame that loses tracking in the MIDDLE of a pipelined stream (one frame always submitted ahead): the frame in
    flight behind it is dropped on the device, the next submission restarts the scan and parks its synchronous result
    behind it -- the same sequence of poses and verdicts as hsk_submit_frame / hsk_wait_frame on one context, the same
    volume, and the stream tracks on afterwards"""
    n = 64
    zero = np.zeros_like(synth_frames(0)[1])
    seq = [synth_frames(k)[1] for k in range(4)] + [zero] + [synth_frames(k)[1] for k in range(4, 11)]

    def pipelined(t):
        res = []
        t.submit_frame(seq[0])
        for d in seq[1:]:
            t.submit_frame(d)
            res.append(t.wait_frame())
        res.append(t.wait_frame())
        return res

    ref = hsk.KinfuTracker(n=n)
    want = pipelined(ref)
    grp = hsk.KinfuGroup(hsk.default_config(n), device_ids=[0] * slabs, flags=flags)
    got = pipelined(grp)
    verdicts = [ok for _, ok in want]
    assert verdicts[:4] == [False, True, True, True] and verdicts[4:7] == [False, False, False] and all(verdicts[7:]), verdicts
    for i, ((p, ok), (pr, okr)) in enumerate(zip(got, want)):
        assert ok == okr, i
        assert_same_bits(p, pr, f"pipelined group pose step {i} around a lost frame ({slabs} slabs, flags {flags})")
    assert_same_bits(grp.download_tsdf(), ref.download_tsdf(), "group tsdf after a pipelined loss and restart")
    # an error in the middle of a call (here: a frame of the wrong size is refused BEFORE anything is enqueued) leaves the
    # group usable; too many submissions are refused without side effects as well
    with pytest.raises(hsk.KinfuError, match="size"):
        grp.submit_frame(np.zeros((10, 10), np.uint16))
    p, ok = grp.process_frame(synth_frames(11)[1])
    pr, okr = ref.process_frame(synth_frames(11)[1])
    assert ok == okr
    assert_same_bits(p, pr, "group pose after a refused submission")
    grp.close()
    ref.close()


def test_c_harness_drives_the_library_without_python(hsk, synth_frames, tmp_path):
    """tests/abi_harness.c: a C program linked against libhskinfu (no Python, no torch in that process) runs 4 frames
    through one context and through a two-slab group; its poses equal the ones this process gets"""
    exe = str(tmp_path / "abi_harness")
    lib_dir = os.path.join(ROOT, "housescan_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_harness.c"),
                           "-L" + lib_dir, "-lhskinfu", "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe, "64", "4"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    ref, want = _reference(hsk, synth_frames, 64, range(4))
    _, total = ref.extract_cloud(cap=0)
    lines = out.stdout.strip().splitlines()
    assert lines[-1] == "done"
    for tag in ("single", "group"):
        rows = [ln.split() for ln in lines if ln.startswith(tag + " ")]
        assert len(rows) == 4
        for k, r in enumerate(rows):
            assert int(r[2]) == int(want[k][1])
            words = np.array([int(x, 16) for x in r[3:]], np.uint32)
            assert np.array_equal(words, np.ascontiguousarray(want[k][0][:3, :4]).view(np.uint32).reshape(-1)), (tag, k)
    assert int([ln for ln in lines if ln.startswith("cloud ")][0].split()[1]) == total
    ref.close()


def test_config5_four_rooms_each_a_two_slab_group_from_four_threads(hsk, synth_frames):
    """BASELINE configs[4] as specified -- four concurrent 512^3 volumes, a PAIR of slabs each -- on the one device of the
    box: four two-slab groups driven from four OS threads at once; every room must reproduce, bit for bit, what a single
    context gives for its frames (the pairs' composites must not leak into each other)"""
    import threading
    n, rooms, nframes = 512, 4, 6
    streams = [[synth_frames(25 * r + k)[1] for k in range(nframes)] for r in range(rooms)]
    want = []
    for r in range(rooms):
        ref = hsk.KinfuTracker(n=n)
        want.append(([ref.process_frame(d) for d in streams[r]], ref.download_tsdf()))
        ref.close()
    groups = [hsk.KinfuGroup(hsk.default_config(n), device_ids=[0, 0]) for _ in range(rooms)]
    outs, errs = [None] * rooms, []

    def run(r):
        try:
            g, res = groups[r], []
            res.append(g.process_frame(streams[r][0]))
            g.submit_frame(streams[r][1])
            for d in streams[r][2:]:
                g.submit_frame(d)
                res.append(g.wait_frame())
            res.append(g.wait_frame())
            outs[r] = res
        except Exception as e:  # noqa: BLE001
            errs.append((r, e))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(rooms)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    full = np.zeros((n, n, n, 2), np.int16)
    for r in range(rooms):
        for k, ((p, ok), (pr, okr)) in enumerate(zip(outs[r], want[r][0])):
            assert ok == okr == (k > 0)
            assert_same_bits(p, pr, f"room {r} frame {k}: pair of slabs, four rooms at once, vs one context alone")
        assert_same_bits(groups[r].download_tsdf(full), want[r][1], f"room {r} tsdf")
        groups[r].close()


# ---- the direct (one-hop, no RCCL) exchange: HSK_GROUP_DIRECT -------------------------------------------------------
def _pipelined(t, frames):
    res = []
    t.submit_frame(frames[0])
    for d in frames[1:]:
        t.submit_frame(d)
        res.append(t.wait_frame())
    res.append(t.wait_frame())
    return res


@pytest.mark.parametrize("slabs,n", [(2, 64), (3, 64), (8, 256)])
def test_direct_exchange_single_process(hsk, synth_frames, slabs, n):
    """peer-write exchange with the slabs of one process (here all on device 0: the same kernels and stream wait / write
    operations as between devices): poses, volume and model maps equal a single context's, pipelined, with a lost frame"""
    zero = np.zeros_like(synth_frames(0)[1])
    frames = [synth_frames(k)[1] for k in range(5)] + [zero] + [synth_frames(k)[1] for k in range(5, 10)]
    ref = hsk.KinfuTracker(n=n)
    want = _pipelined(ref, frames)
    grp = hsk.KinfuGroup(hsk.default_config(n), device_ids=[0] * slabs, flags=hsk.GROUP_DIRECT | hsk.GROUP_PROFILE)
    got = _pipelined(grp, frames)
    for i, ((p, ok), (pr, okr)) in enumerate(zip(got, want)):
        assert ok == okr, i
        assert_same_bits(p, pr, f"direct exchange, {slabs} slabs: pose step {i}")
    assert_same_bits(grp.download_tsdf(), ref.download_tsdf(), "direct exchange: tsdf")
    for i in range(slabs):
        for level in range(3):
            assert_same_bits(grp.slab(i).download_map(2, level), ref.download_map(2, level), f"direct: slab {i} model vmap {level}")
            assert_same_bits(grp.slab(i).download_map(3, level), ref.download_map(3, level), f"direct: slab {i} model nmap {level}")
    ms, front, cnt = grp.exchange_ms()
    assert cnt >= 6 and ms > 0.0 and front > 0.0
    print(f"\ndirect exchange, {slabs} slabs on one device at {n}^3: {1e3 * ms / cnt:.1f} us per frame (slab work before it {1e3 * front / cnt:.1f} us)")
    with pytest.raises(hsk.KinfuError, match="DIRECT"):
        hsk.KinfuGroup(hsk.default_config(n), device_ids=[0, 0], flags=hsk.GROUP_DIRECT | hsk.GROUP_ICP_ALLREDUCE)
    grp.close()
    ref.close()


DIRECT_RANK_SCRIPT = """
import os, sys, time
import numpy as np
sys.path.insert(0, {root!r})
import housescan_amd as hsk
rank, world, idfile, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
uid = open(idfile, "rb").read()
grp = hsk.KinfuGroup(hsk.default_config(n, device_id=0), rank=rank, world=world, comm_id=uid, flags=hsk.GROUP_DIRECT)
zero = np.zeros((480, 640), np.uint16)
frames = [hsk.synth_depth(hsk.synth_pose(k)) for k in range(5)] + [zero] + [hsk.synth_depth(hsk.synth_pose(k)) for k in range(5, 10)]
grp.submit_frame(frames[0])
out = []
for d in frames[1:]:
    grp.submit_frame(d)
    out.append(grp.wait_frame())
out.append(grp.wait_frame())
for p, ok in out:
    print("pose", int(ok), np.ascontiguousarray(p, np.float32).tobytes().hex())
np.save(idfile + ".vol%d.npy" % rank, grp.download_tsdf())
grp.close()
print("done")
"""


@pytest.mark.parametrize("world,n", [(2, 64), (3, 64), (8, 256)])   # (8 ranks: the size of the node the scaling run will use)
def test_direct_exchange_between_processes_sharing_the_gpu(hsk, synth_frames, tmp_path, world, n):
    """the RANK form of the group across OS processes (one slab each; here they share device 0, which RCCL refuses --
    "Duplicate GPU detected" -- and the direct form does not): hipIpc-mapped peer buffers, the POSIX shared-memory flag page,
    stream waits on flags another process raises.  Every rank must report the single context's poses; the ranks' owned
    planes together are the single context's volume."""
    import sys
    idfile = str(tmp_path / "comm_id")
    open(idfile, "wb").write(os.urandom(128))
    script = DIRECT_RANK_SCRIPT.format(root=ROOT)
    procs = [subprocess.Popen([sys.executable, "-c", script, str(r), str(world), idfile, str(n)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=420))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and so.strip().endswith("done"), (r, se[-3000:])
    zero = np.zeros_like(synth_frames(0)[1])
    frames = [synth_frames(k)[1] for k in range(5)] + [zero] + [synth_frames(k)[1] for k in range(5, 10)]
    ref = hsk.KinfuTracker(n=n)
    want = _pipelined(ref, frames)
    for r, (so, _) in enumerate(outs):
        rows = [ln.split() for ln in so.splitlines() if ln.startswith("pose ")]
        assert len(rows) == len(want)
        for k, (row, (pr, okr)) in enumerate(zip(rows, want)):
            assert int(row[1]) == int(okr), (r, k)
            assert bytes.fromhex(row[2]) == np.ascontiguousarray(pr, np.float32).tobytes(), f"rank {r} pose step {k}"
    got = np.load(idfile + ".vol0.npy")
    for r in range(1, world):
        got = got | np.load(idfile + ".vol%d.npy" % r)   # each rank fills only the planes it owns
    assert_same_bits(got, ref.download_tsdf(), "the ranks' owned planes together")
    ref.close()


DEAD_PEER_SCRIPT = """
import os, sys, time
import numpy as np
sys.path.insert(0, {root!r})
import housescan_amd as hsk
rank, world, idfile, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
uid = open(idfile, "rb").read()
grp = hsk.KinfuGroup(hsk.default_config(n, device_id=0), rank=rank, world=world, comm_id=uid, flags=hsk.GROUP_DIRECT)
assert grp.ranks_seen() == world
frames = [hsk.synth_depth(hsk.synth_pose(k)) for k in range(40)]
grp.submit_frame(frames[0])
t_fail = None
try:
    for k, d in enumerate(frames[1:], 1):
        grp.submit_frame(d)
        grp.wait_frame()
        if k == 6 and rank == world - 1:
            os._exit(9)          # this rank dies in the middle of the stream, a frame in flight, no teardown
        t_fail = time.time()
    print("no error")
except hsk.KinfuError as e:
    print("error after %.1f s: %s" % (time.time() - t_fail, e))
    try:
        grp.wait_frame()
    except hsk.KinfuError as e2:
        print("then: %s" % e2)
    try:
        grp.reset()
    except hsk.KinfuError as e3:
        print("reset refused: %s" % e3)
    t0 = time.time()
    grp.close()                  # must return: the streams' waits were released when the group was poisoned
    print("closed in %.1f s" % (time.time() - t0))
    sys.exit(3)
"""


def test_direct_exchange_survives_a_dead_peer(hsk, tmp_path):
    """VERDICT r03 item 7 / ADVICE r03: one of three ranks of a direct-exchange group dies mid-stream.  The flags it would
    have raised never come; every other rank must get a negative code (HSK_ERR_TIMEOUT) from hsk_group_wait_frame within
    HSK_FRAME_TIMEOUT_S, find the group poisoned (reset refused: several ranks), and hsk_group_destroy must return -- the
    waits queued on its streams were released from the host -- so the process exits by itself, non-zero; the shared-memory
    page's name is gone from /dev/shm."""
    import glob
    import sys
    import time
    world, n = 3, 64
    before = set(glob.glob("/dev/shm/hskx_*"))
    idfile = str(tmp_path / "comm_id")
    open(idfile, "wb").write(os.urandom(128))
    script = DEAD_PEER_SCRIPT.format(root=ROOT)
    env = dict(os.environ, HSK_FRAME_TIMEOUT_S="4")
    procs = [subprocess.Popen([sys.executable, "-c", script, str(r), str(world), idfile, str(n)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env) for r in range(world)]
    t0 = time.time()
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=180))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a surviving rank hung behind the dead peer")
    assert procs[world - 1].returncode == 9
    for r in range(world - 1):
        so, se = outs[r]
        assert procs[r].returncode == 3, (r, procs[r].returncode, so, se[-2000:])
        assert "error after" in so and "-5" in so and "did not report" in so, so
        assert "then:" in so and "poisoned" in so and "reset refused" in so and "closed in" in so, so
        waited = float(so.split("error after ")[1].split(" s")[0])
        assert 3.0 < waited < 30.0, so
    assert time.time() - t0 < 150
    assert set(glob.glob("/dev/shm/hskx_*")) <= before, "the direct exchange left its shared-memory page behind"
