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
  host_frames_pipelined_fps / sync_process_frame_fps   the same frames through the boundary's own calls with HOST pointers
                  (hsk_submit_frame / hsk_wait_frame; one hsk_process_frame per frame) -- device_frames_fps (= value) beside them
  room_N / holes_N / noise_N   the timed region again on the room scan (camera inside the volume), on a stream with holes
                  as a sensor makes them, and on SURVEY.md 8(d)'s noise run
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

from bench_common import HBM_PEAK_GBS, ROOT, SLAB_FORMS, W, H, check_build, emit, make_frames


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
def pmc_counters(volume, total, passes, timeout_s=240, stream="scripted"):
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
                                                   os.path.join(ROOT, "tools", "replay_frames.py"), str(volume), str(total), stream]
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


def kernel_durations(volume, total, window_first, timeout_s=240, stream="scripted"):
    """mean duration (us) of every kernel over the launches of the timed window: a child run of tools/replay_frames.py under
    rocprofv3 --kernel-trace (the per-dispatch trace, so that the window can be cut out); {name prefix: us} or (None, reason)"""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    tmp = tempfile.mkdtemp(prefix="hsk_kt_")
    try:
        out = os.path.join(tmp, "kt")
        cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.join(ROOT, "tools", "replay_frames.py"),
               str(volume), str(total), stream]
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


def issue_util_block(volume, total, window_first, stream="scripted"):
    """how busy pass A and pass B keep the two issue pipes -- SQ_INSTS_VALU x 2 cycles (a wave64 instruction on a
    SIMD-32: MI355X_MICROARCH.md) over 1024 SIMDs x the kernel's cycles, SQ_INSTS_SALU over 256 scalar units x cycles (2.4 GHz);
    counters and durations from child runs over the frames of the timed window"""
    per, why = pmc_counters(volume, total, ["SQ_INSTS_VALU SQ_INSTS_SALU"], stream=stream)
    if per is None:
        return {"note": why}
    dur, why = kernel_durations(volume, total, window_first, stream=stream)
    if dur is None:
        return {"note": why}
    out = {"note": "SQ_INSTS_VALU x 2 / (1024 SIMDs x cycles), SQ_INSTS_SALU / (256 x cycles), cycles = mean kernel duration x 2.4 GHz; "
                   "chip-wide averages over the launch: the CUs inside the frustum are busier than that"}
    for label, prefix in (("pass_a", "void k_integrate<false"), ("pass_b", "void k_integrate_detail3<false"), ("k_column_zrange", "k_column_zrange")):
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


INTEGRATE_KERNELS = ("k_column_zrange", "void k_integrate<false", "void k_integrate_detail3<false")   # (name prefixes: pass A is k_integrate<false, 2 | 4>)


def pmc_traffic(volume, total, window_first, timeout_s=240, with_raycast=True, stream="scripted"):
    """HBM-side bytes of the integrate stage per frame: FETCH_SIZE and WRITE_SIZE (separate rocprofv3 passes: they do not
    fit one), mean over the launches of the timed window.  gfx950: FETCH_SIZE counts 64 B per 128-B request on wide
    streams, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.  KiB units.  Also returns the
    raycast's counters (its 4-B gathers are an access width the guide calls uncalibrated: raw and doubled both given)."""
    passes = ["FETCH_SIZE", "WRITE_SIZE"] + (["TCC_HIT_sum TCC_MISS_sum"] if with_raycast else [])
    per, why = pmc_counters(volume, total, passes, timeout_s, stream=stream)
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
    tp = time.perf_counter()
    trk.prepare_readout()   # (hsk_prepare_readout: the pinned pair, row tables, cube table and a product buffer made NOW, off the GL thread)
    prepare_ms = round((time.perf_counter() - tp) * 1e3, 3)
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
    # (the tetrahedra form against the cubes form.  It emits four times the triangles, so its floor is its
    # OUTPUT: 36 B per triangle through PCIe into pageable host memory -- what the two figures below say.  The .ply path of
    # upstream's export, and of tools/stitch_rooms_demo.py, is the cubes form; the tetrahedra form is the watertight variant.)
    for name, key in (("extract_mesh_ms", "mesh_triangles"), ("extract_mesh_cubes_ms", "mesh_cubes_triangles")):
        if out.get(name):
            out[name.replace("_ms", "_output_GBps")] = round(out[key] * 36 / (out[name] * 1e-3) / 1e9, 1)
    out["mesh_note"] = ("hsk_extract_mesh_cubes (marching cubes) is the .ply path; hsk_extract_mesh (marching tetrahedra) emits ~4x the triangles and is "
                        "bound by its output through PCIe (extract_mesh_output_GBps against download_GBps_pageable_host)")
    out["first_call_ms"] = first
    out["prepare_readout_ms"] = prepare_ms
    out["first_call_note"] = ("first_call_ms follows hsk_prepare_readout (prepare_readout_ms, once per context, from the worker thread that created "
                              "it): what is left in a first call over the second is the count pass (the second follows a different product too) "
                              "and a product buffer that has to grow")
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


def concurrent_rooms(n, local_rank, counts=(1, 2, 4), frames=126):
    """M independent rooms on ONE GPU at once (BASELINE configs[4] scans four 512^3 rooms concurrently), driven from C THREADS
    in a process without Python: tools/rooms_native.c is compiled and run as a child process per M --
    a POSIX thread and an hsk_ctx per room, 120 timed frames each through hsk_submit_frame / hsk_wait_frame (host pointers,
    one frame ahead), on the room scan (camera inside, the rooms of synth.cpp) and on the open scene of 8(d).  (Rounds 4-5 drove
    the rooms from Python threads of THIS process for 20 frames: that measured the interpreter and whatever streams torch
    holds -- a process's streams share four hardware queues -- as much as the library.)"""
    exe_dir = tempfile.mkdtemp(prefix="hsk_rooms_")
    exe = os.path.join(exe_dir, "rooms_native")
    lib_dir = os.path.join(ROOT, "housescan_amd")
    out = {}
    try:
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "rooms_native.c"),
                               "-L" + lib_dir, "-lhskinfu", "-ldl", "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        env = dict(os.environ)
        if local_rank:
            env["HIP_VISIBLE_DEVICES"] = str(local_rank)
        for stream in ("room", "open"):
            blk = {}
            for M in counts:
                r = subprocess.run([exe, str(M), str(n), str(frames), stream, "1", "host", "0"], env=env, capture_output=True, text=True, timeout=300)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                if r.returncode != 0 or not line:
                    blk["%d_rooms" % M] = {"failed": (r.stderr or r.stdout)[-300:]}
                    continue
                j = json.loads(line[0])
                blk["%d_rooms" % M] = {k: j[k] for k in ("frames_per_s_in_all", "per_room", "lost_frames", "host_us_per_frame_in_submit", "host_us_per_frame_in_wait")}
            out["room_scan_camera_inside" if stream == "room" else "open_scene_8d"] = blk
    except (OSError, subprocess.SubprocessError) as e:
        out["failed"] = "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(exe_dir, ignore_errors=True)
    out["note"] = ("M independent %d^3 rooms at once on ONE GPU, tools/rooms_native.c (C threads, no Python in the process, %d timed frames per room, "
                   "host pointers): the stages of different rooms overlap, those of one room cannot; what limits four rooms is the chip, not the host "
                   "(profiles/r06/rooms_notes.md: one hipGraph per frame, more hardware queues, CU partitions and a process per room all measured)" % (n, frames - 6))
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
def timed_single(hsk, torch, n, K, Wm, ahead, dev_frames, local_rank, graph=0, sync_api=False, init_pose=None, host_frames=None):
    """the timed region at one GPU: warm-up, then K frames through hsk_submit_frame_dev / hsk_wait_frame (frames resident in
    HBM) -- or, host_frames given, the SAME frames as host buffers through hsk_submit_frame / hsk_wait_frame, the call the
    Haskell host binds (include/hskinfu.h) -- or, sync_api, one hsk_process_frame[_dev] per frame"""
    over = {} if init_pose is None else {"init_pose": init_pose}
    trk = hsk.KinfuTracker(n=n, device_id=local_rank, use_graph=graph, **over)
    total = 1 + Wm + K
    lost = 0
    if host_frames is not None:
        first, submit = (lambda i: trk.process_frame(host_frames[i])), (lambda i: trk.submit_frame(host_frames[i]))
    else:
        first, submit = (lambda i: trk.process_frame_dev(dev_frames[i].data_ptr())), (lambda i: trk.submit_frame_dev(dev_frames[i].data_ptr()))
    # warm-up THROUGH THE API THAT IS TIMED: frame 0 (the scan's first frame is always synchronous), then the Wm warm-up frames
    # by the synchronous call, or pipelined like the timed ones (one frame ahead, all collected before the clock starts)
    first(0)
    if sync_api or Wm < 2:
        for i in range(1, 1 + Wm):
            first(i)
    else:
        submit(1)
        for i in range(2, 1 + Wm):
            submit(i)
            trk.wait_frame()
        trk.wait_frame()
    trk.synchronize()
    torch.cuda.synchronize()
    stamps = []
    import gc
    gc_was = gc.isenabled()
    gc.disable()   # (the interpreter's collector out of the timed region: a generation-2 pass over this process's heap is milliseconds)
    t0 = time.perf_counter()
    if sync_api:
        for i in range(1 + Wm, total):
            pose, ok = first(i)
            lost += (not ok)
            stamps.append(time.perf_counter())
    else:
        ahead = max(1, min(ahead, 2, K - 1)) if K > 1 else 1
        for i in range(1 + Wm, min(1 + Wm + ahead, total)):
            submit(i)
        for i in range(1 + Wm + ahead, total):
            submit(i)
            pose, ok = trk.wait_frame()
            lost += (not ok)
            stamps.append(time.perf_counter())
        for _ in range(min(ahead, K)):
            pose, ok = trk.wait_frame()
            lost += (not ok)
            stamps.append(time.perf_counter())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if gc_was:
        gc.enable()
    per = np.diff(np.array([t0] + stamps)) * 1e3
    return trk, pose, lost, elapsed, per


def replay_with_events(hsk, n, K, Wm, frames, dev_frames, local_rank, final_pose, init_pose=None):
    """stage times, ICP level times and V_upd over exactly the timed frames (a second context with events around every
    stage: the timed loop carries none, they would sit in the pipelined stream)"""
    total = 1 + Wm + K
    rep = hsk.KinfuTracker(n=n, device_id=local_rank, use_graph=0, **({} if init_pose is None else {"init_pose": init_pose}))
    rep.set_profiling(True)
    rep_poses = {}
    all_poses = []
    sample = list(range(1 + Wm, total, max(1, K // 10)))
    p = None
    entries, coarse = [], []
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
            coarse.append(rep.integrate_coarse_counts())    # the coarse level's verdicts over the wave-chunks of this frame
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
    replay_with_events.coarse_counts = None if not coarse else dict(zip(("mixed", "settled_whole", "free_but_worked", "quiet_now"), (int(v) for v in np.mean(coarse, axis=0))))
    return rep, ms, nf, icp_ms, float(np.mean(vupd)), p


def roofline_block(n, ms, nf, v_mean, traffic, traffic_info):
    t_int = ms[2] / nf * 1e-3
    alg_bytes = 8.0 * v_mean + 2.0 * W * H
    achieved = alg_bytes / t_int / 1e9
    touched_mib = alg_bytes / 2 / (1 << 20)  # the voxels a frame rewrites, 4 B each
    block = {
        "bound": "hbm",
        "kernel": "integrate stage: k_column_zrange + k_integrate<false> (pass A) + k_integrate_detail3<false> (pass B), one event pair",
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


STREAMS = {
    "noise": ("SURVEY.md 8(d)'s noise run: the scripted stream + sigma = 1.2 mm x (z / 1 m)^2 per pixel (seed 1234) + 2 % INDEPENDENT dropout (seed 5678)",
              "tests/test_gpu_parity.py::test_noisy_stream_512_vs_oracle (40 frames, bit-equal) and tools/long_parity.py N FRAMES --noise"),
    "holes": ("holes as a structured-light sensor makes them (hsk_synth_render_sensor; housescan/HoniHelper.hs:20-36): the scripted stream with no "
              "return from grazing rays (|n.d| < 0.15), a 3-5 px shadow band on the far side of every depth discontinuity, nothing beyond 3.5 m, "
              "nothing from the absorbing block, sigma = 1.2 mm x z^2 on the rest -- contiguous invalid regions",
              "tests/test_gpu_parity.py::test_sensor_holes_stream_512_vs_oracle (40 frames, bit-equal) and tools/long_parity.py N FRAMES --holes"),
    "room0": ("THE ROOM SCAN: the camera stands INSIDE the volume (hsk_synth_room_*: closed room 0 with furniture, the level turn of the 720-frame "
              "three-turn scan, 1.5 deg of yaw per frame) -- what HouseScan's rooms are made with (README.md:12, Main.hs:1738-1762)",
              "tests/test_gpu_parity.py::test_room_scan_512_vs_oracle (three windows, bit-equal) and tools/long_parity.py N FRAMES --room 0"),
}


def stream_block(args, hsk, torch, n, local_rank, clean_fps, stream, with_counters=False, with_raycast_steps=False, max_steps=60):
    """The timed region again on another synthetic stream (tools/replay_frames.py: stream_frames maps the name to the frames):
    frames/s with the frames resident in HBM and as host buffers, stage times, the integrate stage's algorithmic rate and --
    with_counters -- its counter traffic, the coarse level's verdicts, pass B's queue, lost frames, trajectory error"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from replay_frames import stream_frames
    K2, W2 = min(args.steps, max_steps), min(args.warmup, 10)
    tot = 1 + W2 + K2
    gts, frames, init = stream_frames(hsk, stream, tot)
    dev_all = torch.from_numpy(np.stack(frames).view(np.int16)).cuda(local_rank)
    dev = [dev_all[i] for i in range(tot)]
    torch.cuda.synchronize()
    trk, pose, lost, el, _ = timed_single(hsk, torch, n, K2, W2, args.ahead, dev, local_rank, init_pose=init)
    trk.close()
    trk_h, pose_h, lost_h, el_h, _ = timed_single(hsk, torch, n, K2, W2, args.ahead, dev, local_rank, init_pose=init, host_frames=frames)
    trk_h.close()
    assert np.array_equal(pose, pose_h), "host frames and HBM-resident frames must give the same poses"
    rep, ms, nf, _, v_mean, _ = replay_with_events(hsk, n, K2, W2, frames, dev, local_rank, pose, init_pose=init)
    poses = replay_with_events.poses
    terr = np.array([np.linalg.norm(p[:3, 3] - g[:3, 3]) * 1000.0 for p, g in zip(poses, gts)])
    aerr = np.array([angle_deg(p[:3, :3], g[:3, :3]) for p, g in zip(poses, gts)])
    traffic = tinfo = None
    if with_counters and not args.no_traffic:
        traffic, tinfo, _ = pmc_traffic(n, tot, 1 + W2, with_raycast=False, stream=stream)
    blk = roofline_block(n, ms, nf, v_mean, traffic, tinfo if isinstance(tinfo, dict) else None)
    fps = K2 / el
    out = {"stream": STREAMS[stream][0],
           "invalid_pixel_fraction": round(float(np.mean([(f == 0).mean() for f in frames])), 4),
           "frames_per_s": round(fps, 2), "frames_per_s_host_frames": round(K2 / el_h, 2), "steps": K2, "warmup": W2, "lost_frames": int(lost),
           "vs_clean_render": None if not clean_fps else round(fps / clean_fps, 3),
           "stage_us": {"preprocess": round(ms[0] / nf * 1e3, 1), "icp": round(ms[1] / nf * 1e3, 1), "integrate": round(ms[2] / nf * 1e3, 1),
                        "raycast": round(ms[3] / nf * 1e3, 1)},
           "integrate": {k: blk[k] for k in ("achieved", "frac", "traffic", "hbm_GBps", "hbm_frac", "v_upd_mean", "pass_b_queue_entries_mean", "avg_launch_us")},
           "integrate_coarse_counts": replay_with_events.coarse_counts,
           "trajectory": {"ate_rmse_mm": round(float(np.sqrt(np.mean(terr ** 2))), 3), "ate_max_mm": round(float(terr.max()), 3),
                          "max_angle_deg": round(float(aerr.max()), 4), "frames": len(poses), "against": "the scripted ground truth"},
           "parity": STREAMS[stream][1]}
    if traffic is None and with_counters:
        out["integrate"]["traffic_note"] = "skipped (--no-traffic)" if args.no_traffic else str(tinfo)
    if with_counters and not args.no_traffic:
        out["integrate"]["issue_util"] = issue_util_block(n, tot, 1 + W2, stream=stream)
    if with_raycast_steps and not args.no_cpu_baseline:
        t_ray = ms[3] / nf * 1e-3
        b_ray, n_steps, hits = raycast_algorithmic_bytes(n, rep, rep.get_pose())
        out["raycast"] = {"us": round(t_ray * 1e6, 1), "rays_per_s": round(W * H / t_ray, 0), "march_steps_oracle": n_steps, "hit_rays": hits,
                          "algorithmic_bytes": b_ray, "GBps": round(b_ray / t_ray / 1e9, 1)}
    rep.close()
    return out


def run_single(args, hsk, torch, local_rank):
    t_wall0 = time.perf_counter()
    K, Wm, n = args.steps, args.warmup, args.volume
    total = 1 + Wm + K
    poses_gt, frames = make_frames(hsk, 0, total)
    dev_all = torch.from_numpy(np.stack(frames).view(np.int16)).cuda(local_rank)  # one upload
    dev_frames = [dev_all[i] for i in range(total)]
    torch.cuda.synchronize()
    trk, pose, lost, elapsed, per = timed_single(hsk, torch, n, K, Wm, args.ahead, dev_frames, local_rank, args.graph, args.sync_api)
    gt = poses_gt[total - 1]
    # the SAME frames through the boundary's own calls: host pointers, pipelined (hsk_submit_frame /
    # hsk_wait_frame: what INTEGRATION.md section 5 binds) and one hsk_process_frame per frame (SURVEY.md 8(d) metric (1))
    api = {}
    if not args.no_host_frames:
        for key, kw in (("host_frames_pipelined_fps", {}), ("sync_process_frame_fps", {"sync_api": True})):
            t_h, pose_h, lost_h, el_h, per_h = timed_single(hsk, torch, n, K, Wm, args.ahead, dev_frames, local_rank, args.graph, host_frames=frames, **kw)
            t_h.close()
            assert np.array_equal(pose_h, pose), "every API form must end on the same pose, bit for bit"
            api[key] = round(K / el_h, 2)
            api[key.replace("_fps", "_worst_frame_ms")] = round(float(per_h.max()), 3)
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
        "device_frames_fps": round(K / elapsed, 2) if not args.sync_api else None,
        "host_frames_pipelined_fps": api.get("host_frames_pipelined_fps"), "sync_process_frame_fps": api.get("sync_process_frame_fps"),
        "api_note": ("value = device_frames_fps: the task contract defines `value` with the inputs already resident in HBM when the timed region "
                     "starts and says the PCIe-inclusive rate is never `value`; the boundary's OWN call -- hsk_submit_frame / hsk_wait_frame with "
                     "HOST pointers (HoniHelper.hs:45-46; INTEGRATION.md section 5), the same %d frames, memcpy into the pinned ring + H2D under the "
                     "previous frame -- is host_frames_pipelined_fps, and SURVEY.md 8(d) metric (1), one hsk_process_frame(host pointer) per "
                     "frame, is sync_process_frame_fps; all three end on the same pose bit for bit%s" % (K, "" if not api else
                     "; worst single frame %.2f / %.2f ms" % (api["host_frames_pipelined_worst_frame_ms"], api["sync_process_frame_worst_frame_ms"]))),
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
                              "note": "latency-bound (19 dependent launches): microseconds per iteration, not a roofline fraction.  AT THE FLOOR OF ONE "
                                      "LAUNCH PER ITERATION (round 6, closed): of a fine iteration's 6.6 us, 1.25 are the boundary between two dependent "
                                      "launches, and tools/probes/launch_gap_probe.hip finds that boundary the same whether the 19 launches are eager or ONE "
                                      "hipGraph and whether the accumulators live in coarse-grained, fine-grained or uncached memory (profiles/r06/icp_notes.md); "
                                      "forms with fewer boundaries (XCD-local or chip-wide barriers inside a launch, the coarse level fused) were priced in "
                                      "rounds 4-5 and lose"}
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
        # the room scan (camera INSIDE the volume: what HouseScan's rooms are made with), holes as a sensor makes them, and
        # SURVEY.md 8(d)'s noise run -- each the timed region again on that stream
        out["room_%d" % n] = stream_block(args, hsk, torch, n, local_rank, out["value"], "room0", with_counters=True, with_raycast_steps=True)
        out["holes_%d" % n] = stream_block(args, hsk, torch, n, local_rank, out["value"], "holes")
        out["noise_%d" % n] = stream_block(args, hsk, torch, n, local_rank, out["value"], "noise")
    if not args.no_rooms and n <= 512:
        out["concurrent_rooms_one_gpu"] = concurrent_rooms(n, local_rank)
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
        if not args.no_noise:   # the room scan at the north_star's size too
            out["room_1024"] = stream_block(args, hsk, torch, 1024, local_rank, round(K2 / el2, 2), "room0", max_steps=40)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n, hsk)
    out["bench_wall_s"] = round(time.perf_counter() - t_wall0, 1)   # (everything this invocation did after its imports: timed region, replays, counter passes, other streams, rooms, 1024^3, CPU baseline)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--volume", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (and the oracle-counted raycast bytes)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child runs (roofline.traffic = null)")
    ap.add_argument("--no-1024", action="store_true", help="skip the 1024^3 block (N = 1: roofline_1024; N > 1: slabs_1024)")
    ap.add_argument("--no-host-frames", action="store_true", help="skip the host-pointer figures (host_frames_pipelined_fps, sync_process_frame_fps)")
    ap.add_argument("--no-readout", action="store_true", help="skip the read-out timings (flush of the deferred weights, cloud, volume download)")
    ap.add_argument("--no-trajectory", action="store_true", help="skip the 300-frame recorded-stream replay at 256^3 (trajectory report)")
    ap.add_argument("--stream", default=None, metavar="FILE.hskd",
                    help="time the replay of a recorded depth stream through hsk_track_stream instead of the synthetic in-HBM frames "
                         "(a missing FILE is first recorded from the 300-frame synthetic stream)")
    ap.add_argument("--no-rooms", action="store_true", help="skip the concurrent-room blocks (N = 1: 2 and 4 rooms at once on the one GPU; N > 1, --mode slab: one room per GPU, a room per GPU pair)")
    ap.add_argument("--no-noise", action="store_true", help="skip the other-stream blocks (the timed region again on the room scan, the sensor-holes stream and the noise run: room_ / holes_ / noise_<volume>)")
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
        import bench_launcher
        return bench_launcher.child_main(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    multi = args.gpus > 1 or world > 1 or bool(os.environ.get("HSK_BENCH_FORCE_MULTI"))  # (the variable: the N > 1 flow on one rank, for the tests)
    if multi and not args.stream and not (args.engine == "torch" and args.mode == "slab"):
        import bench_launcher   # (no GPU call, no torch import in THIS process: its workers are fresh processes, never an exec)
        return bench_launcher.launch(args)

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
        import bench_launcher
        out = bench_launcher.run_multi_torch(args, hsk, torch, world, rank, local_rank)
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
