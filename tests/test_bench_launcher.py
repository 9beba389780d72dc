"""CPU tests of bench.py's N > 1 launcher: `python bench.py --gpus N` bare and under
torch.distributed.run, with a scripted stand-in for the GPU worker (HSK_BENCH_WORKER) -- the launcher itself makes no GPU
call, so its whole behaviour can be exercised here: fresh worker processes per form, one JSON line as the last line of
stdout, a worker that exits non-zero or hangs is killed and its form recorded as failed while the other forms still run,
the headline is the fastest form that matches the single context, rc != 0 only when nothing completed."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAKE_WORKER = textwrap.dedent('''
    import argparse, json, os, sys, time
    ap = argparse.ArgumentParser()
    for a in ("--child", "--child-dir", "--child-tag"):
        ap.add_argument(a)
    for a in ("--gpus", "--steps", "--warmup", "--volume"):
        ap.add_argument(a, type=int)
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--allow-exp", action="store_true")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert "LOCAL_RANK" in os.environ
    plan = json.loads(os.environ["FAKE_PLAN"])
    tag = os.path.join(args.child_dir, args.child_tag)
    def hb(ph):
        open("%s.hb.%d" % (tag, rank), "w").write(ph)
    hb("imported")
    what = plan.get("%s@%d" % (args.child, args.volume), plan.get(args.child, {}))
    open(os.path.join(os.environ["FAKE_TRACE"], "%s_%d_%d_%d" % (args.child.replace(":", "_"), args.volume, rank, os.getpid())), "w").write(str(world))
    if rank == 0:
        # a real rank 0 cannot finish a step before its peers have started it (they meet in the step's first barrier); the stand-in
        # waits for their trace files, so that a slow start of a peer (a loaded machine) cannot be mistaken for a missing worker
        pre, t0 = "%s_%d_" % (args.child.replace(":", "_"), args.volume), time.time()
        while len([f for f in os.listdir(os.environ["FAKE_TRACE"]) if f.startswith(pre)]) < world and time.time() - t0 < 60:
            time.sleep(0.02)
    if what.get("fail") and rank == what.get("fail_rank", 0):
        sys.stderr.write("scripted failure of %s\\n" % args.child)
        sys.exit(7)
    if what.get("hang"):
        hb("created")
        time.sleep(600)
    base = args.child.partition(":")[0]
    rooms = world if base == "rooms" else (world // 2 if base == "pairs" else 1)
    out = {"form": args.child, "volume": args.volume, "world": world, "steps": args.steps, "warmup": args.warmup, "build_id": "fake",
           "value": what.get("value", 1000.0) * rooms, "unit": "frames/s", "ms_per_step": 1.0, "rooms": rooms, "lost_frames": what.get("lost", 0),
           "final_translation_error_mm": 0.5, "final_pose_f32_hex": "00", "poses_sha1": "ab"}
    if base != "single":
        out["matches_single_gpu"] = what.get("matches", True)
    if base in ("rccl", "rccl_icp_allreduce", "direct", "pairs"):
        out["ranks_seen"] = world if base != "pairs" else 2
        out["stage_us"] = {"slab_work_us": 100.0, "exchange_us": 30.0, "frames": args.steps}
    if rank == 0:
        json.dump(out, open(tag + ".json", "w"))
    hb("done")
''')


def run_bench(tmp_path, plan, extra, torchrun=0, stall="3"):
    worker = tmp_path / "fake_worker.py"
    worker.write_text(FAKE_WORKER)
    trace = tmp_path / "trace"
    trace.mkdir(exist_ok=True)
    env = dict(os.environ, HSK_BENCH_WORKER=str(worker), FAKE_PLAN=json.dumps(plan), FAKE_TRACE=str(trace), HSK_BENCH_STALL_S=stall)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(v, None)
    cmd = [sys.executable]
    if torchrun:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(torchrun), "--master-addr", "127.0.0.1", "--master-port", str(port)]
    r = subprocess.run(cmd + [os.path.join(ROOT, "bench.py")] + extra, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return r, (json.loads(lines[-1]) if lines and lines[-1].startswith("{") else None), sorted(os.listdir(trace))


@pytest.mark.parametrize("torchrun", [0, 2])
def test_launcher_bare_and_under_torchrun(tmp_path, torchrun):
    plan = {"rccl": {"value": 900.0}, "rccl_icp_allreduce": {"value": 700.0}, "direct": {"value": 1200.0}, "single": {"value": 1000.0},
            "rooms": {"value": 990.0}}
    r, out, trace = run_bench(tmp_path, plan, ["--gpus", "2", "--steps", "20", "--warmup", "5"], torchrun)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["headline_form"] == "direct" and out["value"] == 1200.0
    assert out["matches_single_gpu"] is True and out["ranks_seen"] == 2 and out["steps"] == 20 and out["warmup"] == 5
    assert out["rooms_weak"]["value"] == 1980.0 and out["rooms_weak"]["scaling"] == "weak"
    assert out["value_weak"] == 1980.0 and "scaling: weak" in out["value_weak_note"]   # (round 6: the form that scales, at the top level)
    assert out["predicted_us"]["adopt_us"] == 9.0 and out["predicted_us_r03"]["adopt_us"] == 15.0
    assert set(out["forms"]) == {"rccl", "rccl_icp_allreduce", "direct"} and out["forms"]["rccl_icp_allreduce"]["value"] == 700.0
    assert out["launcher"]["failed_forms"] == {} and out["speedup_vs_single_gpu"] == 1.2
    assert out["predicted_us"]["frame_us"] > 0 and out["stage_us"]["exchange_us"] == 30.0
    # the 1024^3 slabs: the single context and the headline form again, on fresh workers
    assert out["slabs_1024"]["form"] == "direct" and out["slabs_1024"]["speedup_vs_single_gpu"] == 1.2
    # every form ran on its own fresh processes, two per form (one for the single context)
    forms = {}
    for t in trace:
        form, vol, rank, pid = t.rsplit("_", 3)
        forms.setdefault((form, vol), set()).add((rank, pid))
    assert len(forms[("single", "512")]) == 1 and len(forms[("direct", "512")]) == 2 and len(forms[("direct", "1024")]) == 2
    pids = [pid for v in forms.values() for _, pid in v]
    assert len(pids) == len(set(pids)) == 1 + 2 * 4 + 1 + 2


def test_launcher_watchdog_failed_and_hung_forms(tmp_path):
    """the direct form hangs (a wait on a peer flag that never comes), one RCCL form exits non-zero on rank 1, another
    gives a result that does not match the single context: the line still comes, with the remaining good form"""
    plan = {"direct": {"hang": True}, "rccl_icp_allreduce": {"fail": True, "fail_rank": 1}, "rccl": {"value": 800.0},
            "rooms": {"value": 990.0}}
    r, out, _ = run_bench(tmp_path, plan, ["--gpus", "2", "--steps", "8", "--warmup", "2", "--no-1024"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["headline_form"] == "rccl" and out["value"] == 800.0 and out["config"]["parallelism"] == "slab2-rccl"
    ff = out["launcher"]["failed_forms"]
    assert "stalled" in ff["direct@512"] and "killed" in ff["direct@512"]
    assert "rank 1 rc 7" in ff["rccl_icp_allreduce@512"] and "scripted failure" in ff["rccl_icp_allreduce@512"]
    assert "failed" in out["forms"]["direct"] and "slabs_1024" not in out
    # a form whose result differs from the single context is never the headline, however fast
    plan = {"direct": {"value": 5000.0, "matches": False}, "rccl": {"value": 800.0}, "rccl_icp_allreduce": {"value": 600.0}}
    r, out, _ = run_bench(tmp_path, plan, ["--gpus", "2", "--steps", "8", "--warmup", "2", "--no-1024", "--no-rooms"])
    assert r.returncode == 0 and out["headline_form"] == "rccl" and out["forms"]["direct"]["matches_single_gpu"] is False
    assert out["rooms_weak"] is None


def test_launcher_falls_back_to_rooms_and_fails_loudly(tmp_path):
    plan = {f: {"fail": True} for f in ("direct", "rccl", "rccl_icp_allreduce")}
    r, out, _ = run_bench(tmp_path, plan, ["--gpus", "2", "--steps", "8", "--warmup", "2"])
    assert r.returncode == 0 and out["headline_form"] == "rooms" and out["scaling"] == "weak" and len(out["launcher"]["failed_forms"]) == 3
    plan["rooms"] = {"fail": True}
    plan["single"] = {"fail": True}
    r, out, _ = run_bench(tmp_path, plan, ["--gpus", "2", "--steps", "8", "--warmup", "2"])
    assert r.returncode != 0 and out is None and "no form of the 2-GPU path completed" in r.stderr


def test_launcher_pairs_and_single_form_selection(tmp_path):
    r, out, trace = run_bench(tmp_path, {}, ["--gpus", "4", "--steps", "8", "--warmup", "2", "--no-1024", "--forms", "direct"])
    assert r.returncode == 0 and set(out["forms"]) == {"direct"} and out["pairs_weak"]["rooms"] == 2 and out["pairs_weak"]["value"] == 2000.0
    assert any(t.startswith("pairs_direct_512_3_") for t in trace)
    r, out, _ = run_bench(tmp_path, {}, ["--gpus", "4", "--steps", "8", "--warmup", "2", "--mode", "pairs"])
    assert r.returncode == 0 and out["headline_form"] == "pairs" and out["scaling"] == "weak" and out["config"]["parallelism"] == "pairs2"
    r, out, _ = run_bench(tmp_path, {}, ["--gpus", "3", "--mode", "pairs"])
    assert r.returncode != 0 and "even number" in r.stderr


def test_visible_gpu_count_reads_the_topology_without_a_gpu_call(tmp_path, monkeypatch):
    """the launcher's device count: kfd topology nodes with SIMDs (CPU nodes have none), cut down by a *_VISIBLE_DEVICES
    list; 0 (cannot tell) where there is no kfd, in which case the launcher lets the workers find out"""
    sys.path.insert(0, ROOT)
    import bench_launcher as bench
    import glob as globmod
    nodes = []
    for i, simds in enumerate((0, 256, 256, 0, 256)):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (96 if simds == 0 else 0, simds))
        nodes.append(str(d / "properties"))
    monkeypatch.setattr(globmod, "glob", lambda pattern: nodes if "kfd" in pattern else [])
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpu_count() == 2
    monkeypatch.setattr(globmod, "glob", lambda pattern: [])
    assert bench.visible_gpu_count() == 0
