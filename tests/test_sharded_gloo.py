"""N > 1 path on CPU: world_size-2 gloo, the production SlabOrchestrator over an oracle-backed engine.
The z-slab sharded tracker must be bit-identical to the single-volume oracle tracker (poses, composited model
maps, and each slab's owned planes of the TSDF) in both ICP modes."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, W, H, FRAMES = 48, 160, 120, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg(O):
    # quarter-size image so that the CPU oracle runs 4 frames x 19 ICP iterations per rank in seconds
    return O.default_config(N, W=W, H=H, fx=131.25, fy=131.25, cx=79.75, cy=59.75)


def _frames(hsk):
    return [hsk.synth_depth(hsk.synth_pose(3 * k), W, H, 131.25, 131.25, 79.75, 59.75) for k in range(FRAMES)]


def _worker(rank, world, port, icp, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import housescan_amd as hsk
    from housescan_amd.sharded import SlabOrchestrator, slab_halo, slab_range
    from oracle import oracle as O
    from oracle_engine import OracleSlabEngine
    cfg = _cfg(O)
    z0, z1 = slab_range(rank, world, N)
    halo = slab_halo(O.tau(cfg), cfg.size[2] / cfg.vol[2])
    eng = OracleSlabEngine(O, cfg, z0, z1, halo)
    orch = SlabOrchestrator(eng, dist, rank, world, icp=icp, icp_iters=tuple(cfg.icp_iters), height=H)
    poses = []
    for d in _frames(hsk):
        p, ok = orch.process_frame(d)
        poses.append(p)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), poses=np.stack(poses), vmod=eng.vmod[0], nmod=eng.nmod[0],
             vmod2=eng.vmod[2], owned=eng.vol[z0 - eng.zs0:z1 - eng.zs0], z0=z0, z1=z1)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("icp", ["replicated", "allreduce"])
def test_two_slabs_match_single_volume(tmp_path, oracle, hsk, icp):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, icp, str(tmp_path)), nprocs=world, join=True)
    cfg = _cfg(oracle)
    ref = oracle.Tracker(cfg)
    ref_poses = [ref.process(d)[0] for d in _frames(hsk)]
    vol = ref.volume()
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(got["poses"].view(np.uint32), np.stack(ref_poses).view(np.uint32)), f"rank {r} poses"
        for name, kind, level in (("vmod", 2, 0), ("nmod", 3, 0), ("vmod2", 2, 2)):
            a, b = got[name], ref.model_map(kind, level)
            assert np.array_equal(np.isnan(a), np.isnan(b)), (r, name)
            assert np.array_equal(np.nan_to_num(a).view(np.uint32), np.nan_to_num(b).view(np.uint32)), (r, name)
        assert np.array_equal(got["owned"], vol[int(got["z0"]):int(got["z1"])]), f"rank {r} owned TSDF planes"
    assert np.isnan(ref.model_map(2, 0)).mean() < 0.7  # the composite actually holds surface


def test_slab_partition_helpers():
    from housescan_amd.sharded import row_range, slab_halo, slab_range
    for Z, G in ((512, 8), (512, 3), (100, 7), (1024, 8)):
        rs = [slab_range(r, G, Z) for r in range(G)]
        assert rs[0][0] == 0 and rs[-1][1] == Z
        assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
        assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1
    assert [row_range(r, 4, 480) for r in range(4)] == [(0, 120), (120, 240), (240, 360), (360, 480)]
    # 512^3 / 3 m: step 24 mm = 4.1 cells -> 7 + 3
    assert slab_halo(0.03, 3.0 / 512) == 12   # ceil(2 * 0.024 / 0.00586) + 3
