"""Multi-GPU host: one process per GPU, z-slab sharding of ONE TSDF volume (BASELINE.json north_star), or one
independent room per GPU (BASELINE.json configs[4]).

Slab mode.  Rank r owns planes [r*Z/G, (r+1)*Z/G) and stores a halo on both sides which it
integrates redundantly (integration is pointwise: identical inputs give identical voxels, so no halo exchange
ever happens).  Per frame the data path has exactly these exchanges, all through torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests):

  1. raycast: every slab marches only the steps it owns and reports, per pixel, the key (step << 1 | type) of
     the first event it found.  all_reduce(MIN) of the int32 key image (1.2 MB) picks the globally first event.
  2. the winning slab contributes the bit patterns of its vertex / normal (others contribute 0):
     all_reduce(SUM) over int32 (7.4 MB) -- exact, keeps NaN payloads and -0.
  3. ICP, either
       "replicated": every rank runs the full 19 iterations on the composited maps (no collective; the
                     iterations are latency-bound, so replicating them is free and the result is
                     bit-identical to one GPU), or
       "allreduce" : rank r accumulates image rows [r*H/G, (r+1)*H/G) and the 27 sums are all_reduced
                     every iteration (216 B; the north_star's formulation).  The sums are exact
                     (products snapped to 2^-26), so this too is bit-identical to one GPU.

The orchestration below only talks to an *engine* (the HIP one here, an oracle-backed one in
tests/test_sharded_gloo.py), so the CPU test drives the very same code path.
"""
import math

import numpy as np

LEVELS = 3


def slab_range(rank, world, Z):
    """Owned planes of `rank`: contiguous, cover [0, Z) exactly, differ by at most one plane."""
    base, rem = divmod(Z, world)
    z0 = rank * base + min(rank, rem)
    return z0, z0 + base + (1 if rank < rem else 0)


def slab_halo(tau, cell_z):
    """Planes a slab must store beyond what it owns: an owned raycast step reads its near sample (one step
    back), the refined vertex within [t - step, t + 2 step] (oracle D3: two steps either side of the far sample's
    plane), +-1 cell for the normal taps and +-1 voxel for the trilinear taps."""
    step = 0.8 * tau
    return int(math.ceil(2.0 * step / cell_z)) + 3


def row_range(rank, world, H):
    base, rem = divmod(H, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


class SlabOrchestrator:
    """Frame loop of the z-slab sharded tracker over an engine + a torch.distributed-like collective object."""

    def __init__(self, engine, dist, rank, world, icp="replicated", icp_iters=(10, 5, 4), height=480,
                 force_collectives=False):
        assert icp in ("replicated", "allreduce")
        self.e, self.dist, self.rank, self.world = engine, dist, rank, world
        self.icp, self.icp_iters, self.H = icp, tuple(icp_iters), height
        # force_collectives: issue the collectives even at world_size 1 (exercises the RCCL plumbing on a 1-GPU box)
        self.coll = world > 1 or force_collectives

    # ---- pipelined form: submit() enqueues a whole frame (collectives included) without waiting for its pose ----
    def submit(self, depth, next_depth=None):
        """Enqueue one frame; poses come back in order from wait().  Needs an engine with the fused frame front and the
        pipelined frame end (the HIP engine); at most 3 frames may be outstanding."""
        e, dist = self.e, self.dist
        if not hasattr(self, "_fifo"):
            self._fifo = []
        if e.restart_pending():
            # first frame of a (re)started scan: no ICP, no composite, finished synchronously; its result is queued
            # behind the frames still in flight so that wait() keeps the order
            e.frame_front(depth)
            if next_depth is not None:
                e.prefetch(next_depth)
            self._fifo.append(e.frame_end(None, None))
            return
        keys = e.frame_front(depth)
        if next_depth is not None:
            e.prefetch(next_depth)
        if self.coll:
            dist.all_reduce(keys, op=dist.ReduceOp.MIN)
        bits = e.raycast_resolve(keys)
        if self.coll:
            dist.all_reduce(bits, op=dist.ReduceOp.SUM)
        e.frame_end_async(keys, bits)
        self._fifo.append(None)

    def wait(self):
        done = self._fifo.pop(0)
        return done if done is not None else self.e.wait_frame()

    def process_frame(self, depth, next_depth=None):
        """One frame.  `next_depth` (optional): the frame that will be passed next -- engines that can, start copying and
        filtering it on a second stream now, under this frame's ICP / integrate / raycast and collectives."""
        e, dist = self.e, self.dist
        first = e.frame_index() == 0
        fused_front = self.icp == "replicated" and hasattr(e, "frame_front")
        if fused_front:
            keys = e.frame_front(depth)  # preprocess (or the prefetched result) + ICP + integrate + local raycast: one call
        else:
            e.frame_begin(depth)
        if next_depth is not None and hasattr(e, "prefetch"):
            e.prefetch(next_depth)
        if first:
            return e.frame_end(None, None)
        if fused_front:
            if self.coll:
                dist.all_reduce(keys, op=dist.ReduceOp.MIN)
            bits = e.raycast_resolve(keys)
            if self.coll:
                dist.all_reduce(bits, op=dist.ReduceOp.SUM)
            return e.frame_end(keys, bits)
        if self.icp == "replicated" and hasattr(e, "icp_replicated"):
            e.icp_replicated()  # the engine runs all 19 iterations itself (fused kernels, no host round trips)
        else:
            for level in range(LEVELS - 1, -1, -1):
                h = self.H >> level
                for _ in range(self.icp_iters[level]):
                    if self.icp == "allreduce" and self.coll:
                        r0, r1 = row_range(self.rank, self.world, h)
                        sums = e.icp_accumulate(level, r0, r1)
                        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
                    else:
                        sums = e.icp_accumulate(level, 0, h)
                    e.icp_update(sums)
        e.integrate()
        keys = e.raycast_local()
        if self.coll:
            dist.all_reduce(keys, op=dist.ReduceOp.MIN)
        bits = e.raycast_resolve(keys)
        if self.coll:
            dist.all_reduce(bits, op=dist.ReduceOp.SUM)
        return e.frame_end(keys, bits)


class HipSlabEngine:
    """Engine over the C ABI's hsk_mgpu_* building blocks; all buffers are torch CUDA tensors and all work is
    enqueued on torch's current stream, so RCCL collectives order against the kernels without host syncs."""

    def __init__(self, tracker, torch, device, stream=None):
        """`stream`: the torch stream every call and collective of this engine will be issued under (default: the
        current one).  A non-default stream lets hsk_mgpu_frame_front replay its frame front from a hipGraph -- the legacy
        default stream cannot be captured."""
        import ctypes as C
        self.C, self.t, self.torch, self.dev = C, tracker, torch, device
        self.lib = tracker.lib
        self.w, self.h = tracker.w, tracker.hgt
        tracker.set_stream((stream or torch.cuda.current_stream(device)).cuda_stream)
        self.sums = torch.zeros(27, dtype=torch.float64, device=device)
        self.keys = torch.empty(self.h * self.w, dtype=torch.int32, device=device)
        self.bits = torch.empty(6 * self.h * self.w, dtype=torch.int32, device=device)

    def _ck(self, rc):
        self.t._ck(rc)

    def frame_index(self):
        return self.lib.hsk_mgpu_frame_index(self.t.h)

    def frame_begin(self, depth_dev):
        self._ck(self.lib.hsk_mgpu_frame_begin(self.t.h, self.C.c_void_p(depth_dev.data_ptr()), self.w, self.h))

    def frame_front(self, depth_dev):
        self._ck(self.lib.hsk_mgpu_frame_front(self.t.h, self.C.c_void_p(depth_dev.data_ptr()), self.w, self.h,
                                               self.C.c_void_p(self.keys.data_ptr())))
        return self.keys

    def restart_pending(self):
        return bool(self.lib.hsk_mgpu_restart_pending(self.t.h))

    def frame_end_async(self, keys_min, bits):
        self._ck(self.lib.hsk_mgpu_frame_end_async(self.t.h, self.C.c_void_p(keys_min.data_ptr()),
                                                   self.C.c_void_p(bits.data_ptr())))

    def wait_frame(self):
        return self.t.wait_frame()

    def prefetch(self, depth_dev):
        self._ck(self.lib.hsk_mgpu_prefetch(self.t.h, self.C.c_void_p(depth_dev.data_ptr()), self.w, self.h))

    def icp_accumulate(self, level, r0, r1):
        self._ck(self.lib.hsk_mgpu_icp_accumulate(self.t.h, level, r0, r1, self.C.c_void_p(self.sums.data_ptr())))
        return self.sums

    def icp_update(self, sums):
        self._ck(self.lib.hsk_mgpu_icp_update(self.t.h, self.C.c_void_p(sums.data_ptr())))

    def icp_replicated(self):
        self._ck(self.lib.hsk_mgpu_icp_replicated(self.t.h))

    def integrate(self):
        self._ck(self.lib.hsk_mgpu_integrate(self.t.h))

    def raycast_local(self):
        self._ck(self.lib.hsk_mgpu_raycast_local(self.t.h, self.C.c_void_p(self.keys.data_ptr())))
        return self.keys

    def raycast_resolve(self, keys_min):
        self._ck(self.lib.hsk_mgpu_raycast_resolve(self.t.h, self.C.c_void_p(keys_min.data_ptr()),
                                                    self.C.c_void_p(self.bits.data_ptr())))
        return self.bits

    def frame_end(self, keys_min, bits):
        pose = np.empty(16, np.float32)
        tracked = self.C.c_int()
        kp = self.C.c_void_p(keys_min.data_ptr()) if keys_min is not None else None
        bp = self.C.c_void_p(bits.data_ptr()) if bits is not None else None
        self._ck(self.lib.hsk_mgpu_frame_end(self.t.h, kp, bp, pose.ctypes.data_as(self.C.POINTER(self.C.c_float)),
                                             self.C.byref(tracked)))
        return pose.reshape(4, 4), bool(tracked.value)


class ShardedKinfu:
    """What bench.py drives for --gpus N > 1."""

    def __init__(self, n, rank, world, local_rank, mode="slab", icp="replicated", force_collectives=False, use_graph=0):
        import torch
        import torch.distributed as dist

        from .kinfu import KinfuTracker, default_config
        self.mode = mode
        dev = torch.device("cuda", local_rank)
        if mode == "rooms":
            self.tracker = KinfuTracker(default_config(n, device_id=local_rank))
            self.process_frame_dev = lambda t, next_depth=None: self.tracker.process_frame_dev(t.data_ptr())
            return
        # use_graph: replay the fused slab frame front (hsk_mgpu_frame_front) from a hipGraph.  Off by default: with the
        # pipelined submit/wait the host runs ahead and eager launches measured 3 % faster (2000 vs 1935 frames/s)
        cfg = default_config(n, device_id=local_rank, use_graph=int(use_graph))
        z0, z1 = slab_range(rank, world, n)
        cell_z = cfg.vol_size_m[2] / cfg.vol_z
        tau = max(cfg.trunc_dist_m, 2.1 * max(cfg.vol_size_m[0] / cfg.vol_x, cfg.vol_size_m[1] / cfg.vol_y, cell_z))
        cfg.own_z0, cfg.own_z1, cfg.halo = z0, z1, slab_halo(tau, cell_z)
        self.tracker = KinfuTracker(cfg)
        self.stream = torch.cuda.Stream(dev)  # kernels, graphs and collectives of the slab frame all order on this stream
        self.engine = HipSlabEngine(self.tracker, torch, dev, self.stream)
        self.orch = SlabOrchestrator(self.engine, dist, rank, world, icp=icp, icp_iters=tuple(cfg.icp_iters),
                                     height=cfg.height, force_collectives=force_collectives)
        self._torch = torch

    def process_frame_dev(self, depth, next_depth=None):
        with self._torch.cuda.stream(self.stream):
            return self.orch.process_frame(depth, next_depth)

    def submit_frame_dev(self, depth, next_depth=None):
        """pipelined: enqueue the frame and its collectives, collect the pose later with wait_frame()"""
        with self._torch.cuda.stream(self.stream):
            self.orch.submit(depth, next_depth)

    def wait_frame(self):
        with self._torch.cuda.stream(self.stream):
            return self.orch.wait()
