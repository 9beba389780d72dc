#!/usr/bin/env python3
"""bench.py -- frames/sec fused (640x480 depth into a 512^3 TSDF): integrate + ICP + raycast per frame.

One "step" = one whole tracker step (`hsk_process_frame_dev`: bilateral/pyramid/maps, 19 ICP iterations,
TSDF integrate, TSDF raycast, model pyramid) on one synthetic 640x480 depth frame that is already resident
in HBM when the timed region starts.  Prints ONE JSON line (see the task contract) with the extra objects
`roofline` (integrate kernel vs the HBM roofline, algorithmic bytes = 8 B x V_upd + 2 B x W x H, SURVEY.md
8(d)) and `cpu_baseline` (the CPU oracle timed on this box's host cores on a bounded sample).

  python bench.py                       # 1 GPU, 512^3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N   # z-slab sharded
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def make_frames(hsk, first, count):
    poses = [hsk.synth_pose(k) for k in range(first, first + count)]
    return poses, [hsk.synth_depth(p) for p in poses]


def cpu_baseline(volume, sample_frames, hsk):
    """Oracle (CPU restatement, kind "port") on a bounded sample of the same workload: frames 0..sample_frames."""
    from oracle import oracle as O
    threads = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    cfg = O.default_config(volume, omp=True)
    trk = O.Tracker(cfg, omp=True)
    _, frames = make_frames(hsk, 0, sample_frames + 1)
    trk.process(frames[0])  # frame 0 is the untracked first frame (integrate only)
    t0 = time.perf_counter()
    done = 0
    for d in frames[1:]:
        trk.process(d)
        done += 1
        if done >= 8 and time.perf_counter() - t0 > 12.0:  # bounded: about 12 s of CPU work, at least 8 frames
            break
    dt = time.perf_counter() - t0
    sample_frames = done
    stages = trk.stage_seconds()
    trk.close()
    return {
        "value": round(sample_frames / dt, 4), "unit": "frames/s", "cores": threads, "kind": "port",
        "sample": f"{sample_frames} tracked frames of the same synthetic stream into a {volume}^3 TSDF "
                  f"(oracle/kinfu_oracle.c, gcc -O2 -fopenmp, {threads} threads)",
        "stage_seconds": {"preprocess": round(stages[0], 3), "icp": round(stages[1], 3),
                          "integrate": round(stages[2], 3), "raycast": round(stages[3], 3)},
    }


def pmc_traffic(volume):
    """HBM bytes per integrate launch from the committed PMC passes (tools/pmc.sh: separate --pmc runs of this
    same command; FETCH_SIZE in KiB doubled per the gfx950 note in MI355X_MICROARCH.md, WRITE_SIZE in KiB)."""
    path = os.path.join(ROOT, "profiles", "latest_integrate_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return t["bytes_per_launch"] if int(t.get("volume", 0)) == int(volume) else None
    except (OSError, ValueError, KeyError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--volume", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ahead", type=int, default=1, help="frames submitted ahead of the one being waited for (1 or 2; the API allows 3 in flight)")
    ap.add_argument("--cpu-frames", type=int, default=48, help="upper bound of the CPU sample (it stops after about 12 s)")
    ap.add_argument("--graph", type=int, default=0, help="synchronous frames replayed from a hipGraph (default: eager, through the ring)")
    ap.add_argument("--slab-graph", type=int, default=0, help="replay the z-slab frame front from a hipGraph (default: eager)")
    ap.add_argument("--sync-api", action="store_true", help="time hsk_process_frame_dev (one host sync per frame) instead of the submit/wait pair")
    ap.add_argument("--mode", choices=["slab", "rooms"], default="slab")
    ap.add_argument("--icp", choices=["replicated", "allreduce"], default="replicated")
    ap.add_argument("--host-frames", action="store_true", help="also time hsk_process_frame with HOST depth buffers (PCIe-inclusive)")
    ap.add_argument("--force-sharded", action="store_true", help="use the z-slab host + collectives even at 1 GPU (plumbing check)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for the check below)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks on device 0 (with --backend gloo: a logic check of the "
                    "N > 1 path on a one-GPU box; its numbers mean nothing)")
    args = ap.parse_args()

    import torch

    import housescan_amd as hsk

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_sharded:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    K, Wm = args.steps, args.warmup
    n = args.volume
    total = 1 + Wm + K  # frame 0 is the untracked first frame
    poses_gt, frames = make_frames(hsk, 0, total)
    host_extra = make_frames(hsk, total, 180)[1] if args.host_frames else []
    dev_all = torch.from_numpy(np.stack(frames).view(np.int16)).cuda(local_rank)  # one upload
    dev_frames = [dev_all[i] for i in range(len(frames))]
    torch.cuda.synchronize()

    if world > 1 or args.force_sharded:
        from housescan_amd.sharded import ShardedKinfu
        eng = ShardedKinfu(n, rank, world, local_rank, mode=args.mode, icp=args.icp, force_collectives=args.force_sharded,
                           use_graph=args.slab_graph)
        step = eng.process_frame_dev
        trk = eng.tracker
    else:
        trk = hsk.KinfuTracker(n=n, device_id=local_rank, use_graph=args.graph)
        step = lambda t: trk.process_frame_dev(t.data_ptr())  # noqa: E731

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # warmup (includes frame 0 and the hipGraph capture)
    lost = 0
    for i in range(1 + Wm):
        _, ok = step(dev_frames[i])
    use_async = world == 1 and not args.force_sharded and not args.sync_api
    barrier()
    t0 = time.perf_counter()
    if use_async:
        # hsk_submit_frame_dev / hsk_wait_frame: frame i+1 is enqueued before the pose of frame i is read back
        ahead = max(1, min(args.ahead, 2, K - 1))
        for i in range(1 + Wm, 1 + Wm + ahead):
            trk.submit_frame_dev(dev_frames[i].data_ptr())
        for i in range(1 + Wm + ahead, total):
            trk.submit_frame_dev(dev_frames[i].data_ptr())
            pose, ok = trk.wait_frame()
            lost += (not ok)
        for _ in range(ahead):
            pose, ok = trk.wait_frame()
            lost += (not ok)
    elif (world > 1 or args.force_sharded) and args.mode == "slab" and args.icp == "replicated" and not args.sync_api:
        # pipelined slab frames: frame i + 1 (with its collectives) is enqueued before the pose of frame i is read;
        # the next frame is named so that its preprocessing overlaps on the second stream
        nxt = lambda i: dev_frames[i + 1] if i + 1 < total else None  # noqa: E731
        eng.submit_frame_dev(dev_frames[1 + Wm], nxt(1 + Wm))
        for i in range(2 + Wm, total):
            eng.submit_frame_dev(dev_frames[i], nxt(i))
            pose, ok = eng.wait_frame()
            lost += (not ok)
        pose, ok = eng.wait_frame()
        lost += (not ok)
    elif world > 1 or args.force_sharded:
        for i in range(1 + Wm, total):
            pose, ok = step(dev_frames[i], dev_frames[i + 1] if i + 1 < total else None)
            lost += (not ok)
    else:
        for i in range(1 + Wm, total):
            pose, ok = step(dev_frames[i])
            lost += (not ok)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    rooms = world if (world > 1 and args.mode == "rooms") else 1
    fps = rooms * K / elapsed
    gt = poses_gt[total - 1]
    err_mm = float(np.linalg.norm(pose[:3, 3] - gt[:3, 3]) * 1000.0)

    out = {
        "metric": "frames/sec fused (640x480 into %d^3 TSDF): integrate+ICP+raycast" % n,
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": round(1000.0 * elapsed / K, 4), "higher_is_better": True,
        "scaling": "weak" if rooms > 1 else ("strong" if world > 1 else "weak"),
        "vs_baseline": None, "dtype": "f32 (int16 fixed-point TSDF storage, f64 ICP sums)", "data": "synthetic",
        "config": {"workload": "configs[2]-shaped: synthetic 640x480 depth @ scripted trajectory into %d^3 TSDF, "
                               "3 m cube, integrate + 19-iteration ICP + raycast per frame" % n,
                   "volume": n, "image": [640, 480], "icp_iters": [10, 5, 4],
                   "parallelism": ("1 gpu" if world == 1 else f"{args.mode}{world}" + (f"-icp-{args.icp}" if args.mode == "slab" else "")),
                   "graph": bool(args.graph), **({"check_only": "all ranks share device 0 over %s" % args.backend} if args.share_gpu else {}),
                   "api": ("submit/wait (%d frame(s) in flight ahead)" % (max(1, min(args.ahead, 2)) if use_async else 1)) if (use_async or ((world > 1 or args.force_sharded) and args.mode == "slab"
                                                                                    and args.icp == "replicated" and not args.sync_api))
                   else "process_frame (sync per frame)"},
        "tracking": {"lost_frames": int(lost), "final_translation_error_mm": round(err_mm, 3),
                     "final_pose_f32_hex": np.ascontiguousarray(pose[:3, :4], np.float32).tobytes().hex()},
    }

    if rank == 0 and world == 1 and not args.force_sharded:
        # ---- roofline of the dominant kernel (integrate), HIP events on the library's own stream ----
        # The timed loop above carries no events (they would sit in the pipelined stream), so the SAME frames are
        # replayed through a second context with an event pair around every stage: the stream is deterministic, so frame
        # i of the replay does exactly the work frame i of the timed region did (same poses, same volume, same weights --
        # the first 128 frames write every voxel they touch, later ones skip the stores of saturated free space).
        rep = hsk.KinfuTracker(n=n, device_id=local_rank, use_graph=0)
        rep.set_profiling(True)
        rep_poses = {}
        sample = list(range(1 + Wm, total, max(1, K // 10)))
        for i in range(total):
            if i == 1 + Wm:
                rep.stage_ms(reset=True)  # warm-up frames are not part of the timed region
            p, ok = rep.process_frame_dev(dev_frames[i].data_ptr())
            if i in sample:
                rep_poses[i] = p.copy()
        ms, nf = rep.stage_ms(reset=True)
        rep.set_profiling(False)
        assert np.array_equal(p, pose), "the replay must reproduce the timed run's final pose bit for bit"
        vupd = [rep.count_updates(frames[i], rep_poses[i]) for i in sample]
        rep.close()
        t_int = ms[2] / nf * 1e-3
        v_mean = float(np.mean(vupd))
        alg_bytes = 8.0 * v_mean + 2.0 * 640 * 480
        achieved = alg_bytes / t_int / 1e9
        sweep = 8.0 * n ** 3 / t_int / 1e9
        out["roofline"] = {
            "bound": "hbm",
            "kernel": "integrate stage: k_column_zrange + k_integrate<false> (pass A) + k_integrate_detail<false> (pass B), one event pair",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(n),
            "algorithmic_bytes_per_launch": int(alg_bytes), "v_upd_mean": int(v_mean),
            "avg_launch_us": round(t_int * 1e6, 2), "frames": int(nf),
            "window": "the frames of the timed region (replayed with events); rocprofv3 --stats averages the whole run, warm-up included",
            "sweep_GBps_upper_bound_bytes_not_algorithmic": round(sweep, 1),
            "cache_note": ("%d^3 x 4 B = %d MiB; > 256 MiB Infinity Cache => HBM measurement" % (n, n ** 3 * 4 >> 20))
            if n ** 3 * 4 > (256 << 20) else "volume fits the 256 MiB Infinity Cache: cache-resident, NOT an HBM measurement",
        }
        out["stage_us"] = {"preprocess": round(ms[0] / nf * 1e3, 1), "icp": round(ms[1] / nf * 1e3, 1),
                           "integrate": round(ms[2] / nf * 1e3, 1), "raycast": round(ms[3] / nf * 1e3, 1),
                           "note": "means over the %d frames of the timed region, replayed with HIP events between the stages" % nf}
        if args.host_frames:
            hf = host_extra[:60]
            trk.synchronize()
            t1 = time.perf_counter()
            for f in hf:
                trk.process_frame(f)
            out["pcie_inclusive_fps"] = round(len(hf) / (time.perf_counter() - t1), 2)
            hf2 = host_extra[60:]
            trk.synchronize()
            lost2, per = 0, []
            t1 = time.perf_counter()
            trk.submit_frame(hf2[0])
            for f in hf2[1:]:
                trk.submit_frame(f)
                lost2 += not trk.wait_frame()[1]
                per.append(time.perf_counter())
            lost2 += not trk.wait_frame()[1]
            t2 = time.perf_counter()
            per = np.diff(np.array([t1] + per + [t2]))
            out["pcie_inclusive_pipelined_fps"] = round(len(hf2) / (t2 - t1), 2)
            # the HIP runtime was seen to hold one H2D copy back for ~40 ms once per process (GPU idle, copy enqueued:
            # profiles/r01/host_frames_note.md); the figure without the single worst frame is the steady state
            out["pcie_inclusive_pipelined_worst_frame_ms"] = round(float(per.max()) * 1e3, 3)
            out["pcie_inclusive_pipelined_fps_excl_worst"] = round((len(hf2) - 1) / (t2 - t1 - float(per.max())), 2)
            out["pcie_inclusive_pipelined_lost"] = int(lost2)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, args.cpu_frames, hsk)
    if (world > 1 or args.force_sharded) and args.mode == "slab":
        # configs[4] beside the z-slab number: every GPU scans its own room (a full volume per rank, no data-path
        # collective), aggregated over the ranks -- weak scaling of the same fused frame
        room = hsk.KinfuTracker(n=n, device_id=local_rank, use_graph=args.graph)
        k2 = min(K, 100)
        for i in range(1 + Wm):
            room.process_frame_dev(dev_frames[i].data_ptr())
        barrier()
        t1 = time.perf_counter()
        room.submit_frame_dev(dev_frames[1 + Wm].data_ptr())
        for i in range(2 + Wm, 1 + Wm + k2):
            room.submit_frame_dev(dev_frames[i].data_ptr())
            room.wait_frame()
        room.wait_frame()
        barrier()
        el2 = time.perf_counter() - t1
        tt = torch.tensor([el2], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        out["independent_rooms"] = {"value": round(world * k2 / float(tt.item()), 2), "unit": "frames/s", "scaling": "weak",
                                    "steps": k2, "note": "one %d^3 volume per GPU, no collective (BASELINE configs[4])" % n}
        room.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL's version banner sits in the C library's buffer until then
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
