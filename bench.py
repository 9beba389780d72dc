#!/usr/bin/env python3
"""bench.py -- frames/sec fused (640x480 depth into a 512^3 TSDF): integrate + ICP + raycast per frame.

One "step" = one whole tracker step (bilateral / pyramid / maps, 19 ICP iterations, TSDF integrate, TSDF raycast, model
pyramid) on one synthetic 640x480 depth frame that is already resident in HBM when the timed region starts.  Prints ONE
JSON line (the task contract) with, beside the contract's keys (SURVEY.md 8(d), row by row):

  roofline        integrate stage vs the HBM roofline: algorithmic bytes = 8 B x V_upd + 2 B x W x H per launch (V_upd
                  counted by the library's own count-only kernels), duration from HIP events on the library's stream over a
                  replay of exactly the timed frames; `traffic` = HBM bytes per launch from FETCH_SIZE / WRITE_SIZE passes of
                  rocprofv3 over the same frames, run as child processes of THIS invocation (null when rocprofv3 is absent)
  roofline_1024   the same on a 1024^3 volume (the HBM measurement: 4 GiB, sixteen times the Infinity Cache)
  stage_us, icp_us_per_iter, frame_ms (median / p10 / p90), raycast (rays/s, algorithmic GB/s from oracle-counted steps)
  pcie_inclusive  the same frames handed over as HOST buffers (the real shape of HoniHelper.hs:20), pipelined
  cpu_baseline    the CPU oracle (kind "port") on this box's host cores: all cores and one thread, built -O3 -march=native
  concurrent_rooms_one_gpu   2 and 4 independent rooms scanned at once on the one GPU (BASELINE configs[4] on a single device)

  python bench.py                                   # 1 GPU, 512^3
  python bench.py --gpus N                          # bare, or under python -m torch.distributed.run --nproc-per-node N
      ONE volume sharded as z-slabs over N GPUs through the C ABI (hsk_group_*), every form on the same frames, each on
      fresh worker processes: "rccl" (two ncclAllReduce composites per frame), "rccl_icp_allreduce" (the north_star's
      literal form: + the 27 ICP sums all-reduced at each of the 19 iterations), "direct" (one-hop peer writes); value =
      the fastest form whose poses and TSDF planes equal a single context's ("matches_single_gpu"), with "ranks_seen"
      from the communicator, the weak-scaling partitions (rooms_weak, pairs_weak), the 1024^3 slabs (slabs_1024) and
      DESIGN.md section 6's predicted microseconds beside the measured stage_us.  A form that fails or hangs is killed
      by the launcher and recorded under launcher.failed_forms; the others still run.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
W, H = 640, 480


def make_frames(hsk, first, count):
    poses = [hsk.synth_pose(k) for k in range(first, first + count)]
    return poses, [hsk.synth_depth(p) for p in poses]


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (the only place, with the raycast step count, where bench.py touches oracle/)
# ---------------------------------------------------------------------------------------------------------------------
def host_cores():
    """the CPU cores this process can really use: its affinity mask, cut down to the container's CPU quota (cgroup v2
    cpu.max / v1 cfs quota) -- 256 OpenMP threads on a 16-core quota run 30 times slower than 16"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n



def cpu_baseline(volume, hsk, budget_s=12.0):
    """Oracle (CPU restatement, kind "port") on a bounded sample of the same workload, all host cores, then one thread."""
    from oracle import oracle as O
    threads = host_cores()
    native = O.build_native() is not None
    mode = "native" if native else True
    build = "gcc -O3 -march=native -fopenmp" if native else "gcc -O3 -fopenmp (portable build: the native one could not be built here)"

    def run(nthreads, budget, min_frames):
        O.lib(mode).ora_set_threads(int(nthreads))
        cfg = O.default_config(volume, omp=mode)
        trk = O.Tracker(cfg, omp=mode)
        _, frames = make_frames(hsk, 0, 64)
        trk.process(frames[0])  # frame 0 is the untracked first frame (integrate only)
        t0 = time.perf_counter()
        done = 0
        for d in frames[1:]:
            trk.process(d)
            done += 1
            if done >= min_frames and time.perf_counter() - t0 > budget:
                break
        dt = time.perf_counter() - t0
        stages = trk.stage_seconds()
        trk.close()
        return done, dt, stages

    n_all, dt_all, st_all = run(threads, budget_s, 8)
    out = {
        "value": round(n_all / dt_all, 4), "unit": "frames/s", "cores": threads, "kind": "port",
        "sample": f"{n_all} tracked frames of the same synthetic stream into a {volume}^3 TSDF (oracle/kinfu_oracle.c, {build}, "
                  f"{threads} threads = the cores of this container's CPU quota; about {budget_s:.0f} s of CPU work)",
        "stage_seconds": {"preprocess": round(st_all[0], 3), "icp": round(st_all[1], 3), "integrate": round(st_all[2], 3),
                          "raycast": round(st_all[3], 3)},
    }
    n_one, dt_one, _ = run(1, 0.6 * budget_s, 2)
    out["single_thread"] = {"value": round(n_one / dt_one, 4), "unit": "frames/s", "cores": 1,
                            "sample": f"{n_one} tracked frames, same build, one OpenMP thread"}
    return out


def raycast_algorithmic_bytes(volume, trk, pose):
    """B_ray = sum over rays (n_steps x 4 B + [hit] x 64 taps x 4 B) + 2 x 3 x W x H x 4 B, with the march
    steps counted by the oracle on the very volume and pose the GPU raycast ran on"""
    from oracle import oracle as O
    cfg = O.default_config(volume, omp=True)
    vol = trk.download_tsdf()
    _, _, keys, n_steps = O.raycast(cfg, vol, pose, omp=True)
    hits = int(((keys != O.KEY_NONE) & ((keys & 1) == 0)).sum())
    return int(n_steps) * 4 + hits * 64 * 4 + 2 * 3 * W * H * 4, int(n_steps), hits


# ---------------------------------------------------------------------------------------------------------------------
# HBM traffic of the integrate stage from PMC counters, collected by child processes of this run
# ---------------------------------------------------------------------------------------------------------------------
def pmc_counters(volume, total, passes, timeout_s=240):
    """rocprofv3 --pmc passes (one child run of tools/replay_frames.py each, over the same frames as the timed region);
    returns {kernel name: {counter: [value per launch]}} or (None, reason)"""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    per = {}
    tmp = tempfile.mkdtemp(prefix="hsk_pmc_")
    try:
        for i, ctrs in enumerate(passes):
            out = os.path.join(tmp, "p%d" % i)
            cmd = [exe, "--pmc"] + ctrs.split() + ["--output-format", "csv", "-d", out, "--", sys.executable,
                                                   os.path.join(ROOT, "tools", "replay_frames.py"), str(volume), str(total)]
            try:
                subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=timeout_s, check=True)
            except (subprocess.SubprocessError, OSError) as e:
                return None, f"rocprofv3 --pmc {ctrs} failed: {type(e).__name__}"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, f"rocprofv3 --pmc {ctrs} wrote no counter file"
            for r in csv.DictReader(open(files[0])):
                name = r["Kernel_Name"].split("(")[0]
                per.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        return per, None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def kernel_durations(volume, total, window_first, timeout_s=240):
    """mean duration (us) of every kernel over the launches of the timed window: a child run of tools/replay_frames.py under
    rocprofv3 --kernel-trace (the per-dispatch trace, so that the window can be cut out); {name prefix: us} or (None, reason)"""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    tmp = tempfile.mkdtemp(prefix="hsk_kt_")
    try:
        out = os.path.join(tmp, "kt")
        cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.join(ROOT, "tools", "replay_frames.py"),
               str(volume), str(total)]
        try:
            subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
        except (subprocess.SubprocessError, OSError) as e:
            return None, f"rocprofv3 --kernel-trace failed: {type(e).__name__}"
        files = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)
        if not files:
            return None, "rocprofv3 --kernel-trace wrote no trace"
        per = {}
        for r in csv.DictReader(open(files[0])):
            per.setdefault(r["Kernel_Name"].split("(")[0], []).append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
        return {k: float(np.mean(v[window_first:total])) for k, v in per.items() if len(v) >= total and v[window_first:total]}, None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def issue_util_block(volume, total, window_first):
    """how busy pass A and pass B keep the two issue pipes -- SQ_INSTS_VALU x 2 cycles (a wave64 instruction on a
    SIMD-32: MI355X_MICROARCH.md) over 1024 SIMDs x the kernel's cycles, SQ_INSTS_SALU over 256 scalar units x cycles (2.4 GHz);
    counters and durations from child runs over the frames of the timed window"""
    per, why = pmc_counters(volume, total, ["SQ_INSTS_VALU SQ_INSTS_SALU"])
    if per is None:
        return {"note": why}
    dur, why = kernel_durations(volume, total, window_first)
    if dur is None:
        return {"note": why}
    out = {"note": "SQ_INSTS_VALU x 2 / (1024 SIMDs x cycles), SQ_INSTS_SALU / (256 x cycles), cycles = mean kernel duration x 2.4 GHz; "
                   "chip-wide averages over the launch: the CUs inside the frustum are busier than that"}
    for label, prefix in (("pass_a", "void k_integrate<false"), ("pass_b", "void k_integrate_detail2<false"), ("k_column_zrange", "k_column_zrange")):
        pn = [k for k in per if k.startswith(prefix)]
        dn = [k for k in dur if k.startswith(prefix)]
        if not pn or not dn:
            continue
        valu = float(np.mean(per[pn[0]].get("SQ_INSTS_VALU", [0])[window_first:total]))
        salu = float(np.mean(per[pn[0]].get("SQ_INSTS_SALU", [0])[window_first:total]))
        cyc = dur[dn[0]] * 2400.0
        out[label] = {"us": round(dur[dn[0]], 2), "insts_valu": int(valu), "insts_salu": int(salu),
                      "valu": round(valu * 2.0 / (1024.0 * cyc), 3), "salu": round(salu / (256.0 * cyc), 3)}
    return out


INTEGRATE_KERNELS = ("k_column_zrange", "void k_integrate<false", "void k_integrate_detail2<false")   # (name prefixes: pass A is k_integrate<false, 2 | 4>)


def pmc_traffic(volume, total, window_first, timeout_s=240, with_raycast=True):
    """HBM-side bytes of the integrate stage per frame: FETCH_SIZE and WRITE_SIZE (separate rocprofv3 passes: they do not
    fit one), mean over the launches of the timed window.  gfx950: FETCH_SIZE counts 64 B per 128-B request on wide
    streams, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.  KiB units.  Also returns the
    raycast's counters (its 4-B gathers are an access width the guide calls uncalibrated: raw and doubled both given)."""
    passes = ["FETCH_SIZE", "WRITE_SIZE"] + (["TCC_HIT_sum TCC_MISS_sum"] if with_raycast else [])
    per, why = pmc_counters(volume, total, passes, timeout_s)
    if per is None:
        return None, why, None

    def mean(kernel, ctr, first, last):
        names = [k for k in per if k.startswith(kernel)]
        vals = per[names[0]].get(ctr, [])[first:last] if names else []
        return float(np.mean(vals)) if vals else 0.0
    # one launch of each integrate kernel per frame, frame 0 included: index = frame number
    kernels = {}
    fetch = write = 0.0
    for name in INTEGRATE_KERNELS:
        f = mean(name, "FETCH_SIZE", window_first, total) * 1024 * 2
        w = mean(name, "WRITE_SIZE", window_first, total) * 1024
        kernels[name.replace("void ", "") + (">" if name.endswith("<false") else "")] = {"fetch_bytes_corrected_x2": int(f), "write_bytes": int(w)}
        fetch += f
        write += w
    info = {"fetch_bytes_corrected_x2": int(fetch), "write_bytes": int(write), "per_kernel": kernels,
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, child runs of tools/replay_frames.py over the same frames as the "
                      "timed region; mean per frame of the three integrate kernels"}
    ray = None
    if with_raycast:   # the raycast runs from frame 1 on: launch index = frame - 1
        rk = "void k_raycast<false"
        fr = mean(rk, "FETCH_SIZE", window_first - 1, total - 1) * 1024
        hit, miss = mean(rk, "TCC_HIT_sum", window_first - 1, total - 1), mean(rk, "TCC_MISS_sum", window_first - 1, total - 1)
        ray = {"fetch_bytes_raw": int(fr), "fetch_bytes_x2": int(2 * fr), "write_bytes": int(mean(rk, "WRITE_SIZE", window_first - 1, total - 1) * 1024),
               "l2_requests_hit": int(hit), "l2_requests_miss": int(miss), "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None,
               "note": "FETCH_SIZE of 4-B gathers is uncalibrated on gfx950 (exactly half on 16-B/lane streams): the true figure lies between raw and x2"}
    return int(fetch + write), info, ray


# ---------------------------------------------------------------------------------------------------------------------
# work the timed region defers or leaves out: reading the scan's products out of the volume
# ---------------------------------------------------------------------------------------------------------------------
def readout_block(trk, n, with_download=True):
    """After the timed frames: the deferred free-space weights written back into the volume (k_summaries<true>, what
    every read-out does first), the TSDF zero-crossing cloud (hsk_extract_cloud), the two meshes and the whole volume
    (hsk_download_tsdf into pageable host memory) -- host wall clock around each call of the Python binding, i.e. around the
    caller's protocol: a size query, then the fill.  Every product is taken twice: the first call of a context also pays for
    what the library allocates on first use (the pinned staging pair, the row tables, the product buffer: `first_call_ms`);
    the second follows a DIFFERENT product, so its count pass runs again (the cache only serves query + fill of one product)."""
    trk.synchronize()
    t0 = time.perf_counter()
    trk.flush_weights()
    trk.synchronize()
    t1 = time.perf_counter()
    out = {"flush_weights_ms": round((t1 - t0) * 1e3, 3),
           "note": "host clock; flush = the free-space weights held in the summaries written back into the volume: only hsk_download_tsdf "
                   "needs it (the products ask whether a weight is zero, which no deferred weight is); the timed frames never pay it"}
    first = {}

    def timed(name, fn):
        ta = time.perf_counter()
        res, cnt = fn()
        tb = time.perf_counter()
        del res
        return round((tb - ta) * 1e3, 3), int(cnt)

    for rnd in (0, 1):
        for name, key, fn in (("extract_cloud_ms", "cloud_points", trk.extract_cloud), ("extract_mesh_ms", "mesh_triangles", trk.extract_mesh),
                              ("extract_mesh_cubes_ms", "mesh_cubes_triangles", lambda: trk.extract_mesh(cubes=True))):
            ms, cnt = timed(name, fn)
            if rnd == 0:
                first[name] = ms
            else:
                out[name], out[key] = ms, cnt
    out["first_call_ms"] = first
    if with_download:
        t2 = time.perf_counter()
        vol = trk.download_tsdf()   # (a fresh host array: the page faults of its 512 MiB / 4 GiB are inside)
        t3 = time.perf_counter()
        trk.download_tsdf(out=vol)  # (the same array again: the copy alone)
        t4 = time.perf_counter()
        out["download_tsdf_ms"] = round((t4 - t3) * 1e3, 1)
        out["download_GBps_pageable_host"] = round(vol.nbytes / (t4 - t3) / 1e9, 2)
        out["download_tsdf_fresh_array_ms"] = round((t3 - t2) * 1e3, 1)
        del vol
    return out


def concurrent_rooms(hsk, n, dev_frames, local_rank, counts=(2, 4)):
    """M independent rooms on ONE GPU at once (M contexts in this process, a host thread and a stream pair each; BASELINE
    configs[4] scans four 512^3 rooms concurrently): one room's frame is a chain of dependent stages that leaves most of the
    chip idle for half of its time (19 ICP iterations at one wave per SIMD), so rooms interleave -- frames/s in all"""
    import threading
    total = len(dev_frames)
    out = {}
    for M in counts:
        trks = [hsk.KinfuTracker(n=n, device_id=local_rank) for _ in range(M)]
        lost = [0] * M
        for t in trks:
            for k in range(min(6, total - 2)):
                t.process_frame_dev(dev_frames[k].data_ptr())
        first = min(6, total - 2)
        gate = threading.Barrier(M + 1)

        def run(i):
            t = trks[i]
            gate.wait()
            t.submit_frame_dev(dev_frames[first].data_ptr())
            for k in range(first + 1, total):
                t.submit_frame_dev(dev_frames[k].data_ptr())
                lost[i] += not t.wait_frame()[1]
            lost[i] += not t.wait_frame()[1]
            t.synchronize()   # (the last wait returns with the pose: that frame's integrate and raycast are part of the work timed)

        threads = [threading.Thread(target=run, args=(i,)) for i in range(M)]
        for th in threads:
            th.start()
        gate.wait()
        t0 = time.perf_counter()
        for th in threads:
            th.join()
        dt = time.perf_counter() - t0
        for t in trks:
            t.close()
        out["%d_rooms" % M] = {"frames_per_s_in_all": round(M * (total - first) / dt, 1), "per_room": round((total - first) / dt, 1), "lost_frames": int(sum(lost))}
    out["note"] = ("M independent %d^3 rooms scanned at once on ONE GPU (contexts of one process, a host thread each, %d pipelined frames per room): "
                   "the stages of different rooms overlap, those of one room cannot" % (n, total - first))
    return out


def angle_deg(Ra, Rb):
    f = np.linalg.norm(Ra.astype(np.float64) - Rb.astype(np.float64))
    return float(np.degrees(2.0 * np.arcsin(min(1.0, f / (2.0 * np.sqrt(2.0))))))


def is_synthetic_stream(hsk, rd):
    """does the file hold the scripted synthetic stream?  Then its ground-truth poses are known."""
    if (rd.w, rd.hgt) != (W, H) or len(rd) < 1:
        return False
    last = len(rd) - 1
    return np.array_equal(rd[0], hsk.synth_depth(hsk.synth_pose(0))) and np.array_equal(rd[last], hsk.synth_depth(hsk.synth_pose(last)))


def record_synthetic(hsk, path, frames):
    wr = hsk.DepthStreamWriter(path)
    for k in range(frames):
        wr.write(hsk.synth_depth(hsk.synth_pose(k)))
    wr.close()


def stream_replay(hsk, n, path, local_rank=0, first=0, count=None):
    """a recorded HSKD stream through hsk_track_stream (host frames from the file, the frame feed inside the library,
    one frame in flight ahead); frames/s over the whole replay and, for the synthetic stream, the trajectory error"""
    rd = hsk.DepthStreamReader(path)
    count = len(rd) - first if count is None else count
    trk = hsk.KinfuTracker(n=n, device_id=local_rank)
    trk.track_stream(rd, first, min(count, 2))   # page the library in, first-frame path
    trk.reset()
    t0 = time.perf_counter()
    poses, ok = trk.track_stream(rd, first, count)
    trk.synchronize()
    dt = time.perf_counter() - t0
    out = {"volume": n, "frames": int(count), "frames_per_s_file_and_pcie_inclusive": round(count / dt, 2), "lost_frames": int((~ok[1:]).sum()),
           "api": "hsk_track_stream (HSKD file -> pinned ring -> H2D -> tracker, one frame in flight ahead)"}
    if is_synthetic_stream(hsk, rd):
        gt = np.stack([hsk.synth_pose(first + k) for k in range(count)])
        err = np.linalg.norm(poses[:, :3, 3].astype(np.float64) - gt[:, :3, 3].astype(np.float64), axis=1) * 1e3
        ang = np.array([angle_deg(poses[k, :3, :3], gt[k, :3, :3]) for k in range(count)])
        out["trajectory_vs_ground_truth"] = {"ate_rmse_mm": round(float(np.sqrt((err ** 2).mean())), 3), "ate_max_mm": round(float(err.max()), 3),
                                             "final_mm": round(float(err[-1]), 3), "angle_max_deg": round(float(ang.max()), 4),
                                             "note": "scripted synthetic stream (SURVEY.md 8(d)): yaw 12 deg, pitch 4 deg, 0.15 m circle, 10 s @ 30 Hz"}
    rd.close()
    trk.close()
    return out, poses


# ---------------------------------------------------------------------------------------------------------------------
def timed_single(hsk, torch, n, K, Wm, ahead, dev_frames, local_rank, graph=0, sync_api=False):
    """the timed region at one GPU: warm-up, then K frames through hsk_submit_frame_dev / hsk_wait_frame"""
    trk = hsk.KinfuTracker(n=n, device_id=local_rank, use_graph=graph)
    total = 1 + Wm + K
    lost = 0
    for i in range(1 + Wm):
        trk.process_frame_dev(dev_frames[i].data_ptr())
    torch.cuda.synchronize()
    stamps = []
    t0 = time.perf_counter()
    if sync_api:
        for i in range(1 + Wm, total):
            pose, ok = trk.process_frame_dev(dev_frames[i].data_ptr())
            lost += (not ok)
            stamps.append(time.perf_counter())
    else:
        ahead = max(1, min(ahead, 2, K - 1)) if K > 1 else 1
        for i in range(1 + Wm, min(1 + Wm + ahead, total)):
            trk.submit_frame_dev(dev_frames[i].data_ptr())
        for i in range(1 + Wm + ahead, total):
            trk.submit_frame_dev(dev_frames[i].data_ptr())
            pose, ok = trk.wait_frame()
            lost += (not ok)
            stamps.append(time.perf_counter())
        for _ in range(min(ahead, K)):
            pose, ok = trk.wait_frame()
            lost += (not ok)
            stamps.append(time.perf_counter())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per = np.diff(np.array([t0] + stamps)) * 1e3
    return trk, pose, lost, elapsed, per


def replay_with_events(hsk, n, K, Wm, frames, dev_frames, local_rank, final_pose):
    """stage times, ICP level times and V_upd over exactly the timed frames (a second context with events around every
    stage: the timed loop carries none, they would sit in the pipelined stream)"""
    total = 1 + Wm + K
    rep = hsk.KinfuTracker(n=n, device_id=local_rank, use_graph=0)
    rep.set_profiling(True)
    rep_poses = {}
    all_poses = []
    sample = list(range(1 + Wm, total, max(1, K // 10)))
    p = None
    entries = []
    per_frame = []   # stage times of every timed frame (the library sums them; read and reset after each frame)
    for i in range(total):
        if i == 1 + Wm:
            rep.stage_ms(reset=True)  # warm-up frames are not part of the timed region
        p, _ = rep.process_frame_dev(dev_frames[i].data_ptr())
        all_poses.append(p.copy())
        if i >= 1 + Wm:
            m1, n1 = rep.stage_ms(reset=True)
            if n1 == 1:
                per_frame.append(m1)
        if i in sample:
            rep_poses[i] = p.copy()
            entries.append(rep.integrate_queue_entries())   # lane-blocks pass A handed to pass B in this frame
    per_frame = np.array(per_frame) if per_frame else np.zeros((1, 4))
    ms, nf = [float(v) for v in per_frame.sum(axis=0)], len(per_frame)
    replay_with_events.per_frame_us = per_frame * 1e3
    replay_with_events.poses = all_poses
    rep.set_profiling(False)
    if final_pose is not None:
        assert np.array_equal(p, final_pose), "the replay must reproduce the timed run's final pose bit for bit"
    vupd = [rep.count_updates(frames[i], rep_poses[i]) for i in sample]
    # ICP time per pyramid level: a few more frames with an event at every level (those events cost about 4 us each, so
    # they stay out of the stage times above); count_updates left the tracker state alone
    rep.set_profiling(2)
    for _ in range(16):   # the last frame again (camera at rest: the iteration counts are fixed, so the times are the same)
        rep.process_frame_dev(dev_frames[total - 1].data_ptr())
    icp_sum = rep.icp_level_ms()
    _, nf2 = rep.stage_ms(reset=True)
    rep.set_profiling(False)
    icp_ms = [v * nf / max(1, nf2) for v in icp_sum]   # scaled to the nf frames the caller divides by
    replay_with_events.queue_entries = float(np.mean(entries)) if entries else None
    return rep, ms, nf, icp_ms, float(np.mean(vupd)), p


def roofline_block(n, ms, nf, v_mean, traffic, traffic_info):
    t_int = ms[2] / nf * 1e-3
    alg_bytes = 8.0 * v_mean + 2.0 * W * H
    achieved = alg_bytes / t_int / 1e9
    touched_mib = alg_bytes / 2 / (1 << 20)  # the voxels a frame rewrites, 4 B each
    block = {
        "bound": "hbm",
        "kernel": "integrate stage: k_column_zrange + k_integrate<false> (pass A) + k_integrate_detail2<false> (pass B), one event pair",
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic, "hbm_GBps": None if traffic is None else round(traffic / t_int / 1e9, 1),
        "hbm_frac": None if traffic is None else round(traffic / t_int / 1e9 / HBM_PEAK_GBS, 4),
        "frac_note": ("EFFECTIVE, > 1 BECAUSE DEFERRED: " if achieved / HBM_PEAK_GBS > 1.0 else "") +
                     "achieved / frac: ALGORITHMIC bytes (SURVEY.md 8(d): 8 B x V_upd + 2 B x W x H) over the stage's time -- an effective rate: the "
                     "summaries record deep-free-space weight updates in one byte per 16 voxels (or per 4096: the coarse level) instead of moving "
                     "the voxels, so the bytes that really cross the memory side are `traffic`, and hbm_GBps / hbm_frac = traffic over the same "
                     "time; the volume a read-out sees is bit-equal to the oracle's after the flush (profiles/r05/long_parity_*.txt of this build id)",
        "algorithmic_bytes_per_launch": int(alg_bytes), "v_upd_mean": int(v_mean),
        "pass_b_queue_entries_mean": None if getattr(replay_with_events, "queue_entries", None) is None else int(replay_with_events.queue_entries),
        "avg_launch_us": round(t_int * 1e6, 2), "frames": int(nf),
        "window": "the frames of the timed region, replayed with HIP events between the stages (non-pipelined replay: preprocessing on the "
                  "main stream, the frame's last ICP solve in a launch of its own)",
        "sweep_GBps_upper_bound_bytes_not_algorithmic": round(8.0 * n ** 3 / t_int / 1e9, 1),
        "cache_note": ("%d^3 x 4 B = %d MiB volume; the update rule rewrites %.0f MiB of it per frame, of which the kernels move only the part "
                       "near surfaces through memory (traffic); the Infinity Cache holds 256 MiB, so part of that stays MALL-resident from "
                       "frame to frame: not a pure HBM measurement (roofline_1024's working set is)" % (n, n ** 3 * 4 >> 20, touched_mib))
                      if n ** 3 * 4 <= (1 << 30) else
                      ("%d^3 x 4 B = %d MiB volume; the update rule rewrites %.0f MiB of it per frame (algorithmic), the kernels move `traffic` "
                       "bytes of it through memory: both far beyond the 256 MiB Infinity Cache, so traffic / time is an HBM rate"
                       % (n, n ** 3 * 4 >> 20, touched_mib)),
    }
    if traffic_info is not None:
        block["traffic_detail"] = traffic_info
    return block


def noise_block(args, hsk, torch, n, local_rank, clean_fps):
    """SURVEY.md 8(d)'s noise run on the measured path: the same timed region on the stream with sensor noise
    (sigma = 1.2 mm z^2, 2 % dropout: what a real takeDepthSnapshot frame looks like where the render is exact)"""
    K2, W2 = min(args.steps, 60), min(args.warmup, 10)
    tot = 1 + W2 + K2
    gts, frames = hsk.synth_noisy_frames(tot)
    dev_all = torch.from_numpy(np.stack(frames).view(np.int16)).cuda(local_rank)
    dev = [dev_all[i] for i in range(tot)]
    torch.cuda.synchronize()
    trk, pose, lost, el, _ = timed_single(hsk, torch, n, K2, W2, args.ahead, dev, local_rank)
    trk.close()
    rep, ms, nf, _, v_mean, _ = replay_with_events(hsk, n, K2, W2, frames, dev, local_rank, pose)
    rep.close()
    poses = replay_with_events.poses
    terr = np.array([np.linalg.norm(p[:3, 3] - g[:3, 3]) * 1000.0 for p, g in zip(poses, gts)])
    aerr = np.array([angle_deg(p[:3, :3], g[:3, :3]) for p, g in zip(poses, gts)])
    blk = roofline_block(n, ms, nf, v_mean, None, None)
    fps = K2 / el
    out = {"stream": "scripted synthetic stream + sigma = 1.2 mm x (z / 1 m)^2 per pixel (seed 1234) + 2 % dropout (seed 5678)",
           "frames_per_s": round(fps, 2), "steps": K2, "warmup": W2, "lost_frames": int(lost),
           "vs_clean_render": None if not clean_fps else round(fps / clean_fps, 3),
           "stage_us": {"preprocess": round(ms[0] / nf * 1e3, 1), "icp": round(ms[1] / nf * 1e3, 1), "integrate": round(ms[2] / nf * 1e3, 1),
                        "raycast": round(ms[3] / nf * 1e3, 1)},
           "integrate": {k: blk[k] for k in ("achieved", "frac", "v_upd_mean", "pass_b_queue_entries_mean", "avg_launch_us")},
           "trajectory": {"ate_rmse_mm": round(float(np.sqrt(np.mean(terr ** 2))), 3), "ate_max_mm": round(float(terr.max()), 3),
                          "max_angle_deg": round(float(aerr.max()), 4), "frames": len(poses), "against": "the scripted ground truth"},
           "parity": "tests/test_gpu_parity.py::test_noisy_stream_512_vs_oracle (40 frames, bit-equal) and tools/long_parity.py N FRAMES --noise"}
    return out


def run_single(args, hsk, torch, local_rank):
    K, Wm, n = args.steps, args.warmup, args.volume
    total = 1 + Wm + K
    poses_gt, frames = make_frames(hsk, 0, total)
    dev_all = torch.from_numpy(np.stack(frames).view(np.int16)).cuda(local_rank)  # one upload
    dev_frames = [dev_all[i] for i in range(total)]
    torch.cuda.synchronize()
    trk, pose, lost, elapsed, per = timed_single(hsk, torch, n, K, Wm, args.ahead, dev_frames, local_rank, args.graph, args.sync_api)
    gt = poses_gt[total - 1]
    out = {
        "metric": "frames/sec fused (640x480 into %d^3 TSDF): integrate+ICP+raycast" % n,
        "value": round(K / elapsed, 2), "unit": "frames/s", "n_gpus": 1, "steps": K, "warmup": Wm,
        "ms_per_step": round(1000.0 * elapsed / K, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (int16 fixed-point TSDF storage, f64 ICP sums)", "data": "synthetic",
        "config": {"workload": "configs[2]-shaped: synthetic 640x480 depth @ scripted trajectory into %d^3 TSDF, 3 m cube, "
                               "integrate + 19-iteration ICP + raycast per frame" % n,
                   "volume": n, "image": [W, H], "icp_iters": [10, 5, 4], "parallelism": "1 gpu", "graph": bool(args.graph),
                   "api": "process_frame (sync per frame)" if args.sync_api else "submit/wait (%d frame(s) in flight ahead)" % max(1, min(args.ahead, 2))},
        "tracking": {"lost_frames": int(lost), "final_translation_error_mm": round(float(np.linalg.norm(pose[:3, 3] - gt[:3, 3]) * 1000.0), 3),
                     "final_pose_f32_hex": np.ascontiguousarray(pose[:3, :4], np.float32).tobytes().hex()},
        "frame_ms": {"median": round(float(np.median(per)), 4), "p10": round(float(np.percentile(per, 10)), 4),
                     "p90": round(float(np.percentile(per, 90)), 4),
                     "note": "host clock between consecutive hsk_wait_frame returns over the timed region"},
    }
    # ---- stage times + roofline of the dominant kernel group (integrate) ----
    rep, ms, nf, icp_ms, v_mean, _ = replay_with_events(hsk, n, K, Wm, frames, dev_frames, local_rank, pose)
    traffic, tinfo, ray_pmc = (None, "skipped (--no-traffic)", None)
    if not args.no_traffic:
        traffic, tinfo, ray_pmc = pmc_traffic(n, total, 1 + Wm)
    out["roofline"] = roofline_block(n, ms, nf, v_mean, traffic, tinfo if isinstance(tinfo, dict) else None)
    if traffic is None:
        out["roofline"]["traffic_note"] = str(tinfo)
    if not args.no_traffic:
        out["roofline"]["issue_util"] = issue_util_block(n, total, 1 + Wm)
    pf = replay_with_events.per_frame_us
    out["stage_us"] = {"preprocess": round(ms[0] / nf * 1e3, 1), "icp": round(ms[1] / nf * 1e3, 1), "integrate": round(ms[2] / nf * 1e3, 1),
                       "raycast": round(ms[3] / nf * 1e3, 1),
                       "median": {k: round(float(np.median(pf[:, j])), 1) for j, k in enumerate(("preprocess", "icp", "integrate", "raycast"))},
                       "p10": {k: round(float(np.percentile(pf[:, j], 10)), 1) for j, k in enumerate(("preprocess", "icp", "integrate", "raycast"))},
                       "p90": {k: round(float(np.percentile(pf[:, j], 90)), 1) for j, k in enumerate(("preprocess", "icp", "integrate", "raycast"))},
                       "note": "means (top level), medians and percentiles over the %d frames of the timed region, replayed with HIP events between "
                               "the stages" % nf}
    iters = [10, 5, 4]
    out["icp_us_per_iter"] = {"fine_640x480": round(icp_ms[0] / nf / iters[0] * 1e3, 2), "mid_320x240": round(icp_ms[1] / nf / iters[1] * 1e3, 2),
                              "coarse_160x120": round(icp_ms[2] / nf / iters[2] * 1e3, 2),
                              "note": "latency-bound (19 dependent launches): microseconds per iteration, not a roofline fraction"}
    t_ray = ms[3] / nf * 1e-3
    out["raycast"] = {"rays_per_s": round(W * H / t_ray, 0), "us": round(t_ray * 1e6, 1)}
    if ray_pmc is not None:
        out["raycast"]["traffic"] = ray_pmc
    if not args.no_cpu_baseline:
        b_ray, n_steps, hits = raycast_algorithmic_bytes(n, rep, rep.get_pose())
        out["raycast"].update({"algorithmic_bytes": b_ray, "GBps": round(b_ray / t_ray / 1e9, 1), "march_steps_oracle": n_steps, "hit_rays": hits,
                               "note": "B_ray = steps x 4 B + hits x 64 taps x 4 B + map writes (SURVEY.md 8(d)); steps counted by the oracle on "
                                       "the volume and pose of the last timed frame; gather / latency-bound, reported as rays/s"})
    rep.close()
    # ---- the frames handed over as HOST buffers (PCIe-inclusive), pipelined ----
    if not args.no_host_frames:
        hf = make_frames(hsk, total, min(60, max(8, K)))[1]
        trk.synchronize()
        t1 = time.perf_counter()
        trk.submit_frame(hf[0])
        stamps = []
        for f in hf[1:]:
            trk.submit_frame(f)
            trk.wait_frame()
            stamps.append(time.perf_counter())
        trk.wait_frame()
        trk.synchronize()   # (hsk_wait_frame returns when the POSE is final: the last frame's integrate and raycast may still be running)
        t2 = time.perf_counter()
        per_h = np.diff(np.array([t1] + stamps + [t2]))
        out["pcie_inclusive_pipelined_fps"] = round(len(hf) / (t2 - t1), 2)
        out["pcie_inclusive_note"] = ("hsk_submit_frame / hsk_wait_frame with HOST frames (memcpy into a pinned ring, H2D under the previous "
                                      "frame), %d frames; worst single frame %.2f ms" % (len(hf), float(per_h.max()) * 1e3))
    if not args.no_readout:
        out["readout_ms"] = readout_block(trk, n)
        # the deferred weights' worst case: a host that DOWNLOADS THE VOLUME after every frame pays the flush every frame (the
        # products -- cloud, meshes -- no longer flush: they only ask whether a weight is zero, which no deferred weight is)
        fl = out["readout_ms"]["flush_weights_ms"] * 1e3
        out["roofline"]["integrate_plus_flush_every_frame_us"] = round(out["stage_us"]["integrate"] + fl, 1)
        out["roofline"]["flush_note"] = ("only hsk_download_tsdf flushes (it hands the weights out; %.1f ms by itself at this size); clouds and meshes "
                                         "do not: a host that takes a product per frame pays the product, not the flush" % out["readout_ms"].get("download_tsdf_ms", 0.0))
        out["roofline"]["frac_with_flush_every_frame"] = round(out["roofline"]["algorithmic_bytes_per_launch"] / ((out["stage_us"]["integrate"] + fl) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    trk.close()
    if not args.no_noise:
        out["noise_%d" % n] = noise_block(args, hsk, torch, n, local_rank, out["value"])
    if not args.no_rooms and n <= 512:
        out["concurrent_rooms_one_gpu"] = concurrent_rooms(hsk, n, dev_frames, local_rank)
    # ---- SURVEY.md 8(d) cfg2 / BASELINE configs[1]: the 300-frame scripted stream at 256^3, from a recorded file ----
    if not args.no_trajectory:
        tmpd = tempfile.mkdtemp(prefix="hsk_stream_")
        try:
            path = os.path.join(tmpd, "synthetic_300.hskd")
            record_synthetic(hsk, path, 300)
            out["trajectory_256"], _ = stream_replay(hsk, 256, path, local_rank)
            if not args.no_cpu_baseline:   # the CPU restatement beside configs[1]'s size too (a short sample)
                cb = cpu_baseline(256, hsk, budget_s=4.0)
                out["trajectory_256"]["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample")}
        finally:
            shutil.rmtree(tmpd, ignore_errors=True)
    # ---- 1024^3: the HBM measurement ----
    if n == 512 and not args.no_1024:
        K2, W2 = min(K, 40), min(Wm, 5)
        tot2 = 1 + W2 + K2
        trk2, pose2, lost2, el2, _ = timed_single(hsk, torch, 1024, K2, W2, args.ahead, dev_frames[:tot2], local_rank)
        ro2 = None if args.no_readout else readout_block(trk2, 1024)
        trk2.close()
        rep2, ms2, nf2, _, v2, _ = replay_with_events(hsk, 1024, K2, W2, frames[:tot2], dev_frames[:tot2], local_rank, pose2)
        rep2.close()
        tr2, ti2, _ = (None, "skipped (--no-traffic)", None) if args.no_traffic else pmc_traffic(1024, tot2, 1 + W2, with_raycast=False)
        blk = roofline_block(1024, ms2, nf2, v2, tr2, ti2 if isinstance(ti2, dict) else None)
        if tr2 is None:
            blk["traffic_note"] = str(ti2)
        blk.update({"frames_per_s": round(K2 / el2, 2), "steps": K2, "warmup": W2, "lost_frames": int(lost2),
                    "stage_us": {"preprocess": round(ms2[0] / nf2 * 1e3, 1), "icp": round(ms2[1] / nf2 * 1e3, 1),
                                 "integrate": round(ms2[2] / nf2 * 1e3, 1), "raycast": round(ms2[3] / nf2 * 1e3, 1)}})
        if ro2 is not None:
            blk["readout_ms"] = ro2
            blk["integrate_plus_flush_every_frame_us"] = round(blk["stage_us"]["integrate"] + ro2["flush_weights_ms"] * 1e3, 1)
            blk["frac_with_flush_every_frame"] = round(blk["algorithmic_bytes_per_launch"] / ((blk["stage_us"]["integrate"] + ro2["flush_weights_ms"] * 1e3) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        if not args.no_traffic:
            blk["issue_util"] = issue_util_block(1024, tot2, 1 + W2)
        out["roofline_1024"] = blk
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n, hsk)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# N > 1.  `python bench.py --gpus N` is a LAUNCHER that never touches the GPU itself: for every form of the sharded path it
# starts FRESH worker processes (one per GPU: `--child FORM`), watches them (heartbeat files; a form whose workers exit
# non-zero or stall is killed and recorded as failed, the next form starts on new processes), and assembles ONE line from
# what the workers of rank 0 wrote.  Bare invocation: this process starts all N workers.  Under torch.distributed.run
# (the contract's launch line) every rank process is such a launcher for its own worker only; rank 0's is the director
# (it decides the next form and publishes it as a file the others follow).  The workers of one form meet through a
# torch.distributed FileStore in the launcher's scratch directory (gloo, host side only: the id, the barrier, the max of
# the clocks); the data path's collectives are RCCL calls inside the library, or its one-hop peer exchange.
# ---------------------------------------------------------------------------------------------------------------------
SLAB_FORMS = ("rccl", "rccl_icp_allreduce", "direct")
FORM_TEXT = {
    "rccl": "RCCL: ncclAllReduce(MIN) of the raycast step keys + ncclAllReduce(SUM) of the winners' vertex / normal bits per frame; every slab "
            "runs the whole ICP on the composited maps",
    "rccl_icp_allreduce": "the north_star's literal form: the two RCCL composites per frame AND the ICP row-sharded over the slabs with its 27 sums "
                          "ncclAllReduce'd at each of the 19 iterations (HSK_GROUP_ICP_ALLREDUCE)",
    "direct": "one-hop peer writes over xGMI-mapped memory + stream wait / write-value flags (HSK_GROUP_DIRECT, no RCCL call on the frame path); "
              "every slab runs the whole ICP",
}


def slab_range(i, n, Z):
    base, rem = Z // n, Z % n
    z0 = i * base + min(i, rem)
    return z0, z0 + base + (1 if i < rem else 0)


def slab_halo_planes(n, size_m=3.0, trunc=0.03):
    """planes a slab stores beyond its own on each side (hskinfu_group.hip: slab_halo)"""
    cell = size_m / n
    tau = max(trunc, 2.1 * cell)
    return int(np.ceil(2.0 * 0.8 * tau / cell)) + 3


# single-GPU stage times (ICP, integrate, raycast; us) the multi-GPU prediction is priced from: this round's, and round 3's
# (what DESIGN.md section 6's table was first written with: kept beside it as predicted_us_r03)
STAGE_US_R05 = {512: (112.0, 58.0, 57.0), 1024: (113.0, 205.0, 73.0)}
STAGE_US_R03 = {512: (120.0, 71.0, 59.0), 1024: (124.0, 345.0, 95.0)}


def predicted_us(n, G, stage_us=None):
    """the frame time of G z-slabs priced from the CURRENT single-GPU stage times and xGMI link rates
    (arithmetic, never measured) -- carried in the line so that the first multi-GPU run adjudicates it"""
    base = (stage_us or STAGE_US_R05).get(n)
    if base is None or G < 2:
        return None
    icp, integ, ray = base
    integ_g = integ * (1.0 / G + 2.0 * slab_halo_planes(n) / n)
    ray_g = ray / G + 2.0
    exch = {2: 30.0, 4: 38.0, 8: 45.0}.get(G, 30.0 + 2.5 * (G - 2))
    adopt = 15.0
    frame = icp + integ_g + ray_g + exch + adopt
    return {"icp": icp, "integrate": round(integ_g, 1), "raycast": round(ray_g, 1), "slab_work_us": round(icp + integ_g + ray_g, 1),
            "exchange_us": exch, "adopt_us": adopt, "frame_us": round(frame, 1), "frames_per_s": round(1e6 / frame, 1),
            "single_gpu_frame_us": icp + integ + ray,
            "source": "DESIGN.md section 6 (direct exchange; replicated ICP; from %s single-GPU stage times and ~100 GB/s per xGMI link)"
                      % ("round 3's" if stage_us is STAGE_US_R03 else "round 5's")}


def plane_crcs(vol):
    """crc32 of every z plane of a [nz, Y, X, 2] int16 volume"""
    import zlib
    return [zlib.crc32(memoryview(np.ascontiguousarray(vol[z]))) for z in range(vol.shape[0])]


def poses_digest(poses):
    import hashlib
    return hashlib.sha1(np.ascontiguousarray(np.stack(poses), np.float32).tobytes()).hexdigest()


class Heartbeat:
    """the worker's sign of life: a file whose content is the phase and whose mtime the launcher watches"""

    def __init__(self, path):
        self.path, self.phase = path, "start"

    def __call__(self, phase=None):
        if phase is not None:
            self.phase = phase
        if self.path:
            try:
                with open(self.path, "w") as f:
                    f.write(self.phase)
            except OSError:
                pass


def pipelined_run(first, submit, wait, total, Wm, barrier, hb):
    """frame 0 and the warm-up through `first` (submit + wait), then the timed frames with one frame in flight ahead;
    returns (seconds of the timed region, lost frames, pose of every frame)"""
    poses, lost = [], 0
    for i in range(1 + Wm):
        p, _ = first(i)
        poses.append(p.copy())
    hb("warm")
    barrier()
    t0 = time.perf_counter()
    submit(1 + Wm)
    for i in range(2 + Wm, total):
        submit(i)
        p, ok = wait()
        lost += (not ok)
        poses.append(p.copy())
        if (i & 63) == 0:
            hb()
    p, ok = wait()
    lost += (not ok)
    poses.append(p.copy())
    barrier()
    return time.perf_counter() - t0, lost, poses


def child_main(args):
    """one worker: rank RANK of WORLD_SIZE of form args.child, a fresh process on its own GPU"""
    form, n = args.child, args.volume
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", str(rank)))
    tag = os.path.join(args.child_dir, args.child_tag)
    hb = Heartbeat("%s.hb.%d" % (tag, rank))
    hb("start")
    import torch

    import housescan_amd as hsk
    have = check_build(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    hb("imported")
    K, Wm = args.steps, args.warmup
    total = 1 + Wm + K
    single = form in ("single",)
    dist = None
    if not single:
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method="file://" + tag + ".rdzv", rank=rank, world_size=world)
    hb("rendezvous")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    base, _, sub = form.partition(":")   # "pairs:direct" / "pairs:rccl"
    room = rank // 2 if base == "pairs" else (rank if base == "rooms" else 0)
    poses_gt, frames = make_frames(hsk, 25 * room, total)   # every room its own stretch of the trajectory; slabs: the stream's head
    dev_all = torch.from_numpy(np.stack(frames).view(np.int16)).cuda(local_rank)
    dev_frames = [dev_all[i] for i in range(total)]
    torch.cuda.synchronize()
    out = {"form": form, "volume": n, "world": world, "steps": K, "warmup": Wm, "build_id": have}
    grp = trk = None
    if base in ("single", "rooms"):
        trk = hsk.KinfuTracker(n=n, device_id=local_rank)
        submit = lambda i: trk.submit_frame_dev(dev_frames[i].data_ptr())  # noqa: E731
        first = lambda i: trk.process_frame_dev(dev_frames[i].data_ptr())  # noqa: E731
        wait = trk.wait_frame
    else:
        slab_form = sub if base == "pairs" else base
        flags = hsk.GROUP_PROFILE | {"rccl": 0, "rccl_icp_allreduce": hsk.GROUP_ICP_ALLREDUCE, "direct": hsk.GROUP_DIRECT}[slab_form]
        if base == "pairs":
            pgs = [dist.new_group([2 * p, 2 * p + 1]) for p in range(world // 2)]  # (every rank creates every group)
            g_rank, g_world, g_src, g_pg = rank % 2, 2, 2 * room, pgs[room]
        else:
            g_rank, g_world, g_src, g_pg = rank, world, 0, None
        if g_world == 1 and not (flags & hsk.GROUP_DIRECT):
            flags |= hsk.GROUP_FORCE_RCCL   # a world of one rank still goes through ncclCommInitRank and the two all-reduces
        ids = [(os.urandom(128) if (flags & hsk.GROUP_DIRECT) else hsk.KinfuGroup.unique_id()) if rank == g_src else None]
        dist.broadcast_object_list(ids, src=g_src, group=g_pg)
        try:
            grp, why = hsk.KinfuGroup(hsk.default_config(n, device_id=local_rank), rank=g_rank, world=g_world, comm_id=ids[0], flags=flags), None
        except hsk.KinfuError as e:
            why = str(e)
        ok = torch.tensor([0 if grp is None else 1])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if not bool(ok.item()):   # every rank learns it: nobody is left waiting in a collective
            if rank == 0 or why:
                sys.stderr.write("bench.py worker %d: the group of form %s could not be created: %s\n" % (rank, form, why or "a peer rank failed"))
            raise SystemExit(3)
        out["ranks_seen"] = grp.ranks_seen()
        submit = lambda i: grp.submit_frame_dev([dev_frames[i].data_ptr()])  # noqa: E731
        wait = grp.wait_frame

        def first(i):
            submit(i)
            return wait()
    hb("created")
    elapsed, lost, poses = pipelined_run(first, submit, wait, total, Wm, barrier, hb)
    elapsed = max_over_ranks(elapsed)
    hb("timed")
    rooms = world if base == "rooms" else (world // 2 if base == "pairs" else 1)
    gt = poses_gt[total - 1]
    out.update({"value": round(rooms * K / elapsed, 2), "unit": "frames/s", "ms_per_step": round(1000.0 * elapsed / K, 4), "rooms": rooms,
                "lost_frames": int(lost), "final_translation_error_mm": round(float(np.linalg.norm(poses[-1][:3, 3] - gt[:3, 3]) * 1000.0), 3),
                "final_pose_f32_hex": np.ascontiguousarray(poses[-1][:3, :4], np.float32).tobytes().hex(), "poses_sha1": poses_digest(poses)})
    if grp is not None:
        ms, front, cnt = grp.exchange_ms()
        if cnt:
            out["stage_us"] = {"slab_work_us": round(1e3 * front / cnt, 1), "exchange_us": round(1e3 * ms / cnt, 1), "frames": int(cnt),
                               "note": "rank 0's device, HIP events: slab work = ICP (with its all-reduces in the icp_allreduce form) + integrate + "
                                       "slab-local raycast of a frame; exchange = the two composites, waits for the peers included"}
    # ---- the check against ONE context on the same frames: every pose of the run, and every stored plane of every slab ----
    ref_path = os.path.join(args.child_dir, "single_%d.json" % n)
    if base == "single":
        vol = trk.download_tsdf()
        out["plane_crc"] = plane_crcs(vol)
        del vol
    elif base in SLAB_FORMS:
        sl = grp.slab(0)
        vol = sl.download_tsdf()
        mine = {"rank": rank, "z0": int(sl.stored_z0), "crc": plane_crcs(vol)}
        del vol
        hb("crc")
        got = [None] * world if rank == 0 else None
        dist.gather_object(mine, got, dst=0)
        if rank == 0 and os.path.exists(ref_path):
            ref = json.load(open(ref_path))
            planes_ok = all(g["crc"] == ref["plane_crc"][g["z0"]:g["z0"] + len(g["crc"])] for g in got)
            covered = sum(slab_range(r, world, n)[1] - slab_range(r, world, n)[0] for r in range(world)) == n
            out["matches_single_gpu"] = bool(planes_ok and covered and out["poses_sha1"] == ref["poses_sha1"] and
                                             out["final_pose_f32_hex"] == ref["final_pose_f32_hex"])
            out["matches_detail"] = {"every_pose_of_the_run": out["poses_sha1"] == ref["poses_sha1"],
                                     "every_stored_plane_of_every_slab_crc32": bool(planes_ok), "planes_compared": int(sum(len(g["crc"]) for g in got))}
    elif rank == 0 and os.path.exists(ref_path):   # rooms / pairs: rank 0's room runs the same frames as the single context
        ref = json.load(open(ref_path))
        out["matches_single_gpu"] = bool(out["poses_sha1"] == ref["poses_sha1"])
        out["matches_detail"] = {"every_pose_of_room_0": out["matches_single_gpu"]}
    hb("checked")
    if grp is not None:
        grp.close()
    if trk is not None:
        trk.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        for path in [tag + ".json"] + ([ref_path] if base == "single" else []):   # (the single context's result is also the later forms' reference)
            with open(path + ".tmp", "w") as f:
                json.dump(out, f)
            os.replace(path + ".tmp", path)
    hb("done")


def visible_gpu_count():
    """GPUs this process's workers could open, WITHOUT a HIP call (the launcher must stay a process that has never touched the
    GPU): the kfd topology's nodes with SIMDs, cut down by a *_VISIBLE_DEVICES list.  0 = cannot tell (no kfd here)."""
    import glob
    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(path):
                f = line.split()
                if len(f) == 2 and f[0] == "simd_count" and int(f[1]) > 0:
                    n += 1
        except (OSError, ValueError):
            pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if n and v is not None and v.strip():
            n = min(n, len([x for x in v.split(",") if x.strip()]))
    return n


class Launcher:
    """see the comment block above"""
    # seconds without a sign of life before the workers of a form are killed: while torch / the library are being paged
    # in (a fresh box: minutes), afterwards (HSK_BENCH_STALL_S overrides it), and a follower's wait for the director
    IMPORT_STALL_S, STALL_S, STEP_FILE_S = 420.0, float(os.environ.get("HSK_BENCH_STALL_S", "150")), 1800.0

    def __init__(self, args, argv):
        self.args, self.world = args, args.gpus
        self.torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ
        if self.torchrun:
            self.my_ranks = [int(os.environ["RANK"])]
            # one scratch directory per RUN: the launcher's pid of torch.distributed.run, its port and its run id -- and the
            # director empties it before it publishes anything, so that nothing a crashed earlier run left under the same
            # name (step files, rendezvous files, a single-context reference) can be replayed; followers wait for the nonce
            self.dir = os.path.join(tempfile.gettempdir(), "hskbench_%d_%s_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"),
                                                                                 "".join(c for c in os.environ.get("TORCHELASTIC_RUN_ID", "") if c.isalnum())[:24]))
            if int(os.environ["RANK"]) == 0:
                shutil.rmtree(self.dir, ignore_errors=True)
                os.makedirs(self.dir, exist_ok=True)
                with open(os.path.join(self.dir, "nonce.tmp"), "w") as f:
                    f.write(str(os.getpid()))
                os.replace(os.path.join(self.dir, "nonce.tmp"), os.path.join(self.dir, "nonce"))
            else:
                t_end = time.time() + 600.0
                while not os.path.exists(os.path.join(self.dir, "nonce")) and time.time() < t_end:
                    time.sleep(0.05)
        else:
            self.my_ranks = list(range(self.world))
            self.dir = tempfile.mkdtemp(prefix="hskbench_")
        self.director = 0 in self.my_ranks
        self.step_no = 0
        self.failed = {}
        self.log_tail = {}

    # ---- one step: the workers of one form at one volume ----
    def spawn(self, step):
        a = self.args
        procs = []
        for r in step["ranks"]:
            if r not in self.my_ranks:
                continue
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(len(step["ranks"])), GLOO_SOCKET_IFNAME=os.environ.get("GLOO_SOCKET_IFNAME", "lo"),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                       HSK_FRAME_TIMEOUT_S=os.environ.get("HSK_FRAME_TIMEOUT_S", "30"))
            if not self.torchrun or "LOCAL_RANK" not in os.environ:
                env["LOCAL_RANK"] = str(r)
            for v in ("TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "OMP_NUM_THREADS"):
                env.pop(v, None)
            # (HSK_BENCH_WORKER: the CPU tests of the launcher put a scripted stand-in for the GPU worker here)
            cmd = [sys.executable, os.environ.get("HSK_BENCH_WORKER") or os.path.join(ROOT, "bench.py"), "--child", step["form"], "--child-dir", self.dir, "--child-tag", step["tag"],
                   "--gpus", str(self.world), "--steps", str(step["steps"]), "--warmup", str(step["warmup"]), "--volume", str(step["volume"])]
            if a.share_gpu:
                cmd.append("--share-gpu")
            if a.allow_exp:
                cmd.append("--allow-exp")
            log = open(os.path.join(self.dir, "%s.log.%d" % (step["tag"], r)), "w")
            procs.append((r, subprocess.Popen(cmd, env=env, stdout=log, stderr=subprocess.STDOUT, start_new_session=True), log))
        return procs

    def kill(self, procs):
        import signal
        for _, p, _ in procs:   # exactly the process groups this launcher started
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
        for _, p, _ in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass

    def watch(self, step, procs):
        """wait for the workers; (ok, why)"""
        t_start = time.time()
        first_bad = None
        while True:
            codes = [p.poll() for _, p, _ in procs]
            if all(c is not None for c in codes):
                break
            now = time.time()
            if any(c not in (None, 0) for c in codes):
                first_bad = first_bad or now
                if now - first_bad > 20.0:   # a worker failed: its peers get a moment to notice, then go too
                    self.kill(procs)
                    break
            newest, phases = t_start, []
            for r, _, _ in procs:
                hbp = os.path.join(self.dir, "%s.hb.%d" % (step["tag"], r))
                try:
                    newest = max(newest, os.path.getmtime(hbp))
                    phases.append(open(hbp).read() or "start")
                except OSError:
                    phases.append("not started")
            limit = self.IMPORT_STALL_S if any(ph in ("not started", "start") for ph in phases) else self.STALL_S
            if now - newest > limit:
                self.kill(procs)
                for _, _, log in procs:
                    log.close()
                return False, "stalled for %.0f s in phase %s: workers killed" % (limit, "/".join(sorted(set(phases))))
            time.sleep(0.05)
        for _, _, log in procs:
            log.close()
        codes = [p.returncode for _, p, _ in procs]
        if any(c != 0 for c in codes):
            tails = []
            for r, p, _ in procs:
                if p.returncode != 0:
                    try:
                        lines = open(os.path.join(self.dir, "%s.log.%d" % (step["tag"], r))).read().strip().splitlines()
                    except OSError:
                        lines = []
                    tails.append("rank %d rc %s: %s" % (r, p.returncode, " | ".join(lines[-3:])[-400:]))
            return False, "; ".join(tails)
        return True, None

    def run_step(self, form, volume, ranks=None, steps=None, warmup=None):
        """director: publish the step, run my share of it; returns rank 0's result dict or None"""
        a = self.args
        step = {"form": form, "volume": volume, "ranks": list(range(self.world)) if ranks is None else ranks,
                "steps": a.steps if steps is None else steps, "warmup": a.warmup if warmup is None else warmup,
                "tag": "s%02d_%s_%d" % (self.step_no, form.replace(":", "_"), volume)}
        self.publish(step)
        return self.execute(step)

    def publish(self, step):
        path = os.path.join(self.dir, "step_%03d.json" % self.step_no)
        with open(path + ".tmp", "w") as f:
            json.dump(step, f)
        os.replace(path + ".tmp", path)
        self.step_no += 1

    def execute(self, step):
        procs = self.spawn(step)
        if not procs:
            return None
        ok, why = self.watch(step, procs)
        res_path = os.path.join(self.dir, step["tag"] + ".json")
        if ok and (0 not in [r for r, _, _ in procs] or os.path.exists(res_path)):
            return json.load(open(res_path)) if os.path.exists(res_path) else {}
        self.failed["%s@%d" % (step["form"], step["volume"])] = why or "no result written"
        return None

    def follow(self):
        """a launcher that is not the director (torch.distributed.run, rank != 0): run my worker of every published step"""
        while True:
            path = os.path.join(self.dir, "step_%03d.json" % self.step_no)
            t0 = time.time()
            while not os.path.exists(path):
                # the director removes the scratch directory when it has printed the line: that, too, says "done" (a
                # follower whose last worker exits late must not wait for a file that will never come)
                if not os.path.isdir(self.dir) or time.time() - t0 > self.STEP_FILE_S:
                    return 0   # the director is done or gone; nothing of this rank's is left running
                time.sleep(0.05)
            step = json.load(open(path))
            self.step_no += 1
            if step["form"] == "done":
                return 0
            self.execute(step)

    # ---- the director's plan and the line ----
    def direct(self):
        a, n, G = self.args, self.args.volume, self.world
        t_begin = time.time()
        single = self.run_step("single", n, ranks=[0])
        rooms = self.run_step("rooms", n) if (a.mode in ("slab", "rooms") and not a.no_rooms) or a.mode == "rooms" else None
        forms = {}
        if a.mode == "slab":
            for f in a.forms:
                forms[f] = self.run_step(f, n)
        # (a form counts only when it was CHECKED against the single context and matched: without the reference -- the
        # `single` step failed -- nothing is "good", and the line falls back to the weak-scaling head below)
        good = {f: r for f, r in forms.items() if r and r.get("matches_single_gpu") is True and not r["lost_frames"]}
        best = max(good, key=lambda f: good[f]["value"]) if good else None
        pairs = None
        if a.mode == "pairs" or (a.mode == "slab" and G >= 4 and G % 2 == 0 and not a.no_rooms):
            sub = "direct" if (a.mode == "pairs" and "direct" in a.forms) or (forms.get("direct") and "direct" in good) else "rccl"
            pairs = self.run_step("pairs:" + sub, n)
            if pairs is None and sub == "direct":
                pairs = self.run_step("pairs:rccl", n)
        big = None
        if a.mode == "slab" and n == 512 and not a.no_1024 and best is not None:
            K2, W2 = min(a.steps, 40), min(a.warmup, 5)
            s2 = self.run_step("single", 1024, ranks=[0], steps=K2, warmup=W2)
            order = [best] + [f for f in sorted(good, key=lambda f: -good[f]["value"]) if f != best]
            r2 = f2 = None
            for f in order:
                r2, f2 = self.run_step(f, 1024, steps=K2, warmup=W2), f
                if r2 and r2.get("matches_single_gpu") is True:
                    break
            big = {"workload": "configs[3]: ONE 1024^3 TSDF as %d z-slabs, the same synthetic stream, %d timed frames" % (G, K2),
                   "single_gpu_same_frames": None if s2 is None else {k: s2[k] for k in ("value", "unit", "ms_per_step", "lost_frames")},
                   "form": f2, "slabs": None if r2 is None else {k: r2[k] for k in r2 if k not in ("build_id", "form", "world", "volume")},
                   "speedup_vs_single_gpu": None if not (r2 and s2) else round(r2["value"] / s2["value"], 3),
                   "predicted_us": predicted_us(1024, G), "predicted_us_r03": predicted_us(1024, G, STAGE_US_R03)}
        self.publish({"form": "done"})
        # ---- the line ----
        K, Wm = a.steps, a.warmup
        strip = lambda r: None if r is None else {k: r[k] for k in r if k not in ("build_id", "form", "world", "volume", "steps", "warmup", "plane_crc")}  # noqa: E731
        head = good[best] if best else None
        if head is None and a.mode == "pairs" and pairs:
            head = pairs
        if head is None and rooms:
            head = rooms   # no slab form ran to a checked result: the weak-scaling partition is what this node measured
        if head is None:
            sys.stderr.write("bench.py: no form of the %d-GPU path completed: %s\n" % (G, json.dumps(self.failed)))
            self.dump_logs()
            return None
        slab_head = best is not None
        out = {
            "metric": "frames/sec fused (640x480 into %d^3 TSDF): integrate+ICP+raycast" % n,
            "value": head["value"], "unit": "frames/s", "n_gpus": G, "steps": K, "warmup": Wm, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "strong" if slab_head else "weak", "vs_baseline": None,
            "dtype": "f32 (int16 fixed-point TSDF storage, f64 ICP sums)", "data": "synthetic",
            "config": {"workload": ("configs[3]-shaped: ONE %d^3 TSDF sharded as z-slabs over the GPUs" % n) if slab_head else
                                   ("configs[4]: %d concurrent %d^3 rooms, a GPU pair (two z-slabs) each" % (head["rooms"], n) if head is pairs else
                                    "configs[4]-shaped: one %d^3 room per GPU, no data-path collective" % n),
                       "volume": n, "image": [W, H], "icp_iters": [10, 5, 4],
                       "parallelism": ("slab%d-%s" % (G, best)) if slab_head else ("pairs%d" % head["rooms"] if head is pairs else "rooms%d" % G),
                       "exchange": FORM_TEXT.get(best) if slab_head else None,
                       "api": "hsk_group_submit_frame_dev / hsk_group_wait_frame (C ABI), 1 frame in flight ahead" if (slab_head or head is pairs)
                              else "hsk_submit_frame_dev / hsk_wait_frame"},
            "headline_form": best if slab_head else ("pairs" if head is pairs else "rooms"),
            "headline_note": ("value = the fastest z-slab form whose every pose and every stored TSDF plane equal a single context's on the same "
                              "frames (strong scaling of ONE volume); rooms_weak = the same GPUs with one independent room each") if slab_head else
                             "no z-slab form completed with a checked result on this node (see forms / failed_forms): value is the weak-scaling partition",
            "matches_single_gpu": head.get("matches_single_gpu"),
            "ranks_seen": head.get("ranks_seen", G if not slab_head else None),
            "tracking": {"lost_frames": head["lost_frames"], "final_translation_error_mm": head["final_translation_error_mm"],
                         "final_pose_f32_hex": head["final_pose_f32_hex"]},
            "rooms_weak": None if rooms is None else dict(strip(rooms), scaling="weak",
                                                            workload="one %d^3 room per GPU, hsk_submit_frame_dev / hsk_wait_frame, %d frames each, no data-path collective" % (n, K)),
            "forms": {f: (dict(strip(r), what=FORM_TEXT[f]) if r else {"failed": self.failed.get("%s@%d" % (f, n), "failed")}) for f, r in forms.items()},
            "single_gpu_same_frames": None if single is None else {k: single[k] for k in ("value", "unit", "ms_per_step", "lost_frames", "final_pose_f32_hex")},
            "speedup_vs_single_gpu": None if not (single and slab_head) else round(head["value"] / single["value"], 3),
            "predicted_us": predicted_us(n, G), "predicted_us_r03": predicted_us(n, G, STAGE_US_R03),
        }
        if head.get("stage_us"):
            out["stage_us"] = head["stage_us"]
        if pairs is not None:
            out["pairs_weak"] = dict(strip(pairs), scaling="weak", workload="BASELINE configs[4]: one %d^3 room per GPU pair (two z-slabs, own exchange)" % n)
        if big is not None:
            out["slabs_1024"] = big
        out["launcher"] = {"mode": "torch.distributed.run: every rank process launches its own fresh worker per form" if self.torchrun else
                                   "bare: this process launched all %d workers of every form" % G,
                           "failed_forms": self.failed, "wall_s": round(time.time() - t_begin, 1),
                           "share_gpu_check_only": bool(a.share_gpu)}
        out["build_id"] = head.get("build_id")
        # (build_id.py is loaded by path: importing the package would load libhskinfu.so -- and the HIP runtime -- into the launcher)
        import importlib.util
        spec = importlib.util.spec_from_file_location("hsk_build_id", os.path.join(ROOT, "housescan_amd", "csrc", "build_id.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if out["build_id"] != mod.build_id():
            out["experimental_build"] = True
        return out

    def dump_logs(self):
        for f in sorted(glob.glob(os.path.join(self.dir, "*.log.*"))):
            try:
                txt = open(f).read().strip().splitlines()[-6:]
            except OSError:
                continue
            if txt:
                sys.stderr.write("--- %s\n%s\n" % (os.path.basename(f), "\n".join(txt)))

    def cleanup(self):
        if self.director:
            time.sleep(0.2)
            shutil.rmtree(self.dir, ignore_errors=True)


def check_build(args):
    """The measured library must be the default build of THIS tree: "+exp" marks other compiler flags (timing experiments,
    some of which give wrong results by construction), a different hash a stale .so.  tests/conftest.py refuses both too."""
    from housescan_amd import _lib
    from housescan_amd.csrc import build_id as tree_id
    have, want = _lib.load().hsk_build_id().decode(), tree_id.build_id()
    if have != want and not args.allow_exp:
        raise SystemExit("bench.py: housescan_amd/libhskinfu.so is build %s, the tree is %s -- rebuild with "
                         "`python -c 'import __graft_entry__ as g; g.build()'` (or pass --allow-exp for a timing experiment)" % (have, want))
    return have


def run_multi_torch(args, hsk, torch, world, rank, local_rank):
    """The N > 1 slab flow with the collectives issued from Python through torch.distributed (housescan_amd/sharded.py:
    the harness the group call was checked against).  --backend gloo --share-gpu runs all ranks on device 0: a logic
    check of the flow on a one-GPU box, its numbers mean nothing."""
    import torch.distributed as dist
    from housescan_amd.sharded import ShardedKinfu
    if args.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    K, Wm, n = args.steps, args.warmup, args.volume
    total = 1 + Wm + K
    poses_gt, frames = make_frames(hsk, 0, total)
    dev_all = torch.from_numpy(np.stack(frames).view(np.int16)).cuda(local_rank)
    dev_frames = [dev_all[i] for i in range(total)]
    torch.cuda.synchronize()
    eng = ShardedKinfu(n, rank, world, local_rank, mode="slab", icp=args.icp)

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    lost = 0
    for i in range(1 + Wm):
        eng.process_frame_dev(dev_frames[i])
    barrier()
    t0 = time.perf_counter()
    if args.icp == "replicated":
        nxt = lambda i: dev_frames[i + 1] if i + 1 < total else None  # noqa: E731
        eng.submit_frame_dev(dev_frames[1 + Wm], nxt(1 + Wm))
        for i in range(2 + Wm, total):
            eng.submit_frame_dev(dev_frames[i], nxt(i))
            pose, ok = eng.wait_frame()
            lost += (not ok)
        pose, ok = eng.wait_frame()
        lost += (not ok)
    else:
        for i in range(1 + Wm, total):
            pose, ok = eng.process_frame_dev(dev_frames[i], dev_frames[i + 1] if i + 1 < total else None)
            lost += (not ok)
    barrier()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    gt = poses_gt[total - 1]
    out = {
        "metric": "frames/sec fused (640x480 into %d^3 TSDF): integrate+ICP+raycast" % n,
        "value": round(K / elapsed, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": round(1000.0 * elapsed / K, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32 (int16 fixed-point TSDF storage, f64 ICP sums)", "data": "synthetic",
        "config": {"workload": "configs[3]-shaped: ONE %d^3 TSDF sharded as z-slabs over the GPUs" % n, "volume": n, "image": [W, H],
                   "icp_iters": [10, 5, 4], "parallelism": "slab%d-icp-%s" % (world, args.icp),
                   "api": "housescan_amd/sharded.py over torch.distributed (%s)" % args.backend,
                   **({"check_only": "all ranks share device 0 over %s" % args.backend} if args.share_gpu else {})},
        "tracking": {"lost_frames": int(lost), "final_translation_error_mm": round(float(np.linalg.norm(pose[:3, 3] - gt[:3, 3]) * 1000.0), 3),
                     "final_pose_f32_hex": np.ascontiguousarray(pose[:3, :4], np.float32).tobytes().hex()},
    }
    dist.barrier()
    dist.destroy_process_group()
    return out if rank == 0 else None


def emit(out, have=None, want=None):
    """the JSON line is the LAST thing on stdout: RCCL's version banner sits in the C library's buffer until then"""
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(json.dumps(out))
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--volume", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (and the oracle-counted raycast bytes)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child runs (roofline.traffic = null)")
    ap.add_argument("--no-1024", action="store_true", help="skip the 1024^3 block (N = 1: roofline_1024; N > 1: slabs_1024)")
    ap.add_argument("--no-host-frames", action="store_true", help="skip the PCIe-inclusive (host frame) figure")
    ap.add_argument("--no-readout", action="store_true", help="skip the read-out timings (flush of the deferred weights, cloud, volume download)")
    ap.add_argument("--no-trajectory", action="store_true", help="skip the 300-frame recorded-stream replay at 256^3 (trajectory report)")
    ap.add_argument("--stream", default=None, metavar="FILE.hskd",
                    help="time the replay of a recorded depth stream through hsk_track_stream instead of the synthetic in-HBM frames "
                         "(a missing FILE is first recorded from the 300-frame synthetic stream)")
    ap.add_argument("--no-rooms", action="store_true", help="skip the concurrent-room blocks (N = 1: 2 and 4 rooms at once on the one GPU; N > 1, --mode slab: one room per GPU, a room per GPU pair)")
    ap.add_argument("--no-noise", action="store_true", help="skip the noise run (the timed region again on the stream with sensor noise: noise_<volume>)")
    ap.add_argument("--quick", action="store_true", help="all of the above")
    ap.add_argument("--ahead", type=int, default=1, help="frames submitted ahead of the one being waited for (1 or 2)")
    ap.add_argument("--graph", type=int, default=0, help="synchronous frames replayed from a hipGraph (default: eager, through the ring)")
    ap.add_argument("--sync-api", action="store_true", help="time hsk_process_frame_dev (one host sync per frame) instead of submit/wait")
    ap.add_argument("--mode", choices=["slab", "rooms", "pairs"], default="slab",
                    help="N > 1: z-slabs of ONE volume (strong scaling; the line also carries the rooms / pairs blocks), only one room per GPU, "
                         "or only one room per GPU pair (configs[4])")
    ap.add_argument("--forms", default=None,
                    help="N > 1 slabs: comma-separated forms to time, of " + ", ".join(SLAB_FORMS) + " (default: all; the headline is the fastest "
                         "one that matches a single context)")
    ap.add_argument("--icp", choices=["replicated", "allreduce"], default=None, help="N > 1 (older spelling): with --exchange, names ONE form")
    ap.add_argument("--exchange", choices=["direct", "rccl"], default=None, help="N > 1 (older spelling): time only this exchange")
    ap.add_argument("--engine", choices=["group", "torch"], default="group",
                    help="N > 1 slabs: hsk_group_* (C ABI; the launcher described above) or, under torch.distributed.run only, the Python "
                         "harness over torch.distributed")
    ap.add_argument("--backend", default="nccl", help="--engine torch: torch.distributed backend (nccl = RCCL; gloo for the check below)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="all ranks on device 0 (logic check on a one-GPU box; RCCL refuses two ranks on one device, so only the direct form "
                         "and the rooms run there)")
    ap.add_argument("--allow-exp", action="store_true",
                    help="measure a library built with other than the default flags or from other sources (the A/B scripts under tools/); "
                         "the line then carries \"experimental_build\": true and is not a result")
    ap.add_argument("--child", default=None, help=argparse.SUPPRESS)       # a worker of the N > 1 launcher
    ap.add_argument("--child-dir", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--child-tag", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.quick:
        args.no_cpu_baseline = args.no_traffic = args.no_1024 = args.no_host_frames = args.no_rooms = True
        args.no_readout = args.no_trajectory = args.no_noise = True
    if args.forms:
        args.forms = [f.strip() for f in args.forms.split(",") if f.strip()]
        bad = [f for f in args.forms if f not in SLAB_FORMS]
        if bad:
            raise SystemExit("--forms: unknown form(s) %s (known: %s)" % (bad, ", ".join(SLAB_FORMS)))
    elif args.exchange or args.icp:
        args.forms = ["rccl_icp_allreduce"] if args.icp == "allreduce" else [args.exchange or "direct"]
    else:
        args.forms = list(SLAB_FORMS)

    if args.child:
        return child_main(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    multi = args.gpus > 1 or world > 1 or bool(os.environ.get("HSK_BENCH_FORCE_MULTI"))  # (the variable: the N > 1 flow on one rank, for the tests)
    if multi and not args.stream and not (args.engine == "torch" and args.mode == "slab"):
        # the launcher: no GPU call, no torch import in THIS process (its workers are fresh processes, never an exec)
        if world > 1 and world != args.gpus:
            raise SystemExit("bench.py: --gpus %d under a launch of %d ranks" % (args.gpus, world))
        if args.mode == "pairs" and args.gpus % 2:
            raise SystemExit("--mode pairs needs an even number of GPUs")
        seen = visible_gpu_count()
        if 0 < seen < args.gpus and not args.share_gpu:
            # (said at once, by the launcher: the workers would each find it out after their imports, a minute and a half later)
            sys.stderr.write("bench.py: --gpus %d, but this box shows %d GPU%s; --share-gpu runs the ranks on one device (a correctness "
                             "run of the N > 1 paths, not a measurement)\n" % (args.gpus, seen, "" if seen == 1 else "s"))
            raise SystemExit(2)
        L = Launcher(args, sys.argv)
        if not L.director:
            return L.follow()
        try:
            out = L.direct()
        finally:
            L.cleanup()
        if out is None:
            raise SystemExit(1)
        emit(out)
        return 0

    import torch

    import housescan_amd as hsk
    from housescan_amd.csrc import build_id as tree_id
    have, want = check_build(args), tree_id.build_id()
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if args.stream:
        if not os.path.exists(args.stream):
            record_synthetic(hsk, args.stream, 300)
        rep, _ = stream_replay(hsk, args.volume, args.stream, local_rank)
        out = {"metric": "frames/sec fused (640x480 into %d^3 TSDF): integrate+ICP+raycast" % args.volume,
               "value": rep["frames_per_s_file_and_pcie_inclusive"], "unit": "frames/s", "n_gpus": 1, "steps": rep["frames"], "warmup": 0,
               "ms_per_step": round(1e3 / rep["frames_per_s_file_and_pcie_inclusive"], 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32 (int16 fixed-point TSDF storage, f64 ICP sums)", "data": "recorded stream " + os.path.basename(args.stream),
               "config": {"workload": "configs[2]: %d^3 scan from a RECORDED depth stream (HSKD file; host frames, file read and PCIe inside the "
                                      "timed region -- not the HBM-resident headline figure)" % args.volume, "volume": args.volume, "image": [W, H],
                          "icp_iters": [10, 5, 4], "parallelism": "1 gpu", "api": rep["api"]},
               "stream": rep}
    elif multi:   # --engine torch --mode slab, launched by torch.distributed.run
        if world != args.gpus:
            raise SystemExit("--engine torch must be launched with torch.distributed.run --nproc-per-node N")
        out = run_multi_torch(args, hsk, torch, world, rank, local_rank)
    else:
        out = run_single(args, hsk, torch, local_rank)
    if rank == 0 and out is not None:
        out["build_id"] = have   # which sources the measured library was built from (housescan_amd/csrc/build_id.py)
        if have != want:
            out["experimental_build"] = True
        emit(out)
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
