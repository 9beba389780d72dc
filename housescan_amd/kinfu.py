"""Host-side mirror of the KinFu tracker interface over the C ABI.

Names follow the upstream KinFu application that HouseScan's README points users at
(/root/reference/README.md:13-14): a tracker object fed one depth frame at a time, returning the camera
pose; `extract_cloud` yields the packed float32 xyz cloud HouseScan stores in `Cloud.cloudPoints`
(/root/reference/housescan/Main.hs:117-121).  Depth frames are numpy uint16 arrays of shape (h, w) in the
row-major layout of HoniHelper.takeDepthSnapshot (/root/reference/housescan/HoniHelper.hs:20-36).
Errors surface as `KinfuError` carrying `hsk_last_error` (the `Left String` of HoniHelper.hs:39-42).
"""
import ctypes as C

import numpy as np

from . import _lib


class KinfuError(RuntimeError):
    pass


def default_config(n=512, **over):
    lib = _lib.load()
    cfg = _lib.HskConfig()
    lib.hsk_default_config(C.byref(cfg), int(n))
    for k, v in over.items():
        if k in ("vol_size_m", "icp_iters", "init_pose"):
            arr = getattr(cfg, k)
            for i, x in enumerate(np.asarray(v).reshape(-1)):
                arr[i] = x
        else:
            setattr(cfg, k, v)
    return cfg


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class KinfuTracker:
    """One TSDF volume + tracker on one MI355X (one `hsk_ctx`)."""

    def __init__(self, cfg=None, **over):
        self.lib = _lib.load()
        self.cfg = cfg if cfg is not None else default_config(**over)
        h = C.c_void_p()
        rc = self.lib.hsk_create(C.byref(self.cfg), C.byref(h))
        if rc != 0:
            raise KinfuError(f"hsk_create failed ({rc}): {self.lib.hsk_last_error(None).decode()}")
        self.h = h
        self.w, self.hgt = self.cfg.width, self.cfg.height
        z0, nz = C.c_int(), C.c_int()
        self.lib.hsk_stored_planes(self.h, C.byref(z0), C.byref(nz))
        self.stored_z0, self.stored_nz = z0.value, nz.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.hsk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise KinfuError(f"hskinfu error {rc}: {self.lib.hsk_last_error(self.h).decode()}")

    @staticmethod
    def _depth(depth):
        d = np.ascontiguousarray(depth, dtype=np.uint16)
        if d.ndim != 2:
            raise KinfuError("depth must be a (h, w) uint16 array")
        return d

    # ---- whole tracker step -------------------------------------------------------------------------
    def process_frame(self, depth):
        d = self._depth(depth)
        pose = np.empty(16, np.float32)
        tracked = C.c_int()
        self._ck(self.lib.hsk_process_frame(self.h, d.ctypes.data, d.shape[1], d.shape[0], _fp(pose), C.byref(tracked)))
        return pose.reshape(4, 4), bool(tracked.value)

    def process_frame_dev(self, depth_dev_ptr):
        pose = np.empty(16, np.float32)
        tracked = C.c_int()
        self._ck(self.lib.hsk_process_frame_dev(self.h, C.c_void_p(depth_dev_ptr), self.w, self.hgt, _fp(pose),
                                                C.byref(tracked)))
        return pose.reshape(4, 4), bool(tracked.value)

    def submit_frame_dev(self, depth_dev_ptr):
        """enqueue a frame (device pointer) without waiting; collect poses in order with wait_frame()"""
        self._ck(self.lib.hsk_submit_frame_dev(self.h, C.c_void_p(depth_dev_ptr), self.w, self.hgt))

    def submit_frame(self, depth):
        """enqueue a frame held in host memory (copied before this returns); collect poses in order with wait_frame()"""
        d = self._depth(depth)
        self._ck(self.lib.hsk_submit_frame(self.h, d.ctypes.data, d.shape[1], d.shape[0]))

    def wait_frame(self):
        pose = np.empty(16, np.float32)
        tracked = C.c_int()
        self._ck(self.lib.hsk_wait_frame(self.h, _fp(pose), C.byref(tracked)))
        return pose.reshape(4, 4), bool(tracked.value)

    def track_stream(self, reader, first=0, count=None):
        """frames [first, first + count) of a recorded stream (products.DepthStreamReader) through the tracker, the
        frame feed running inside the library (hsk_track_stream) -> (poses [count, 4, 4], tracked [count] bool)"""
        if count is None:
            count = len(reader) - first
        poses = np.empty((count, 16), np.float32)
        tracked = np.zeros(count, np.int32)
        self._ck(self.lib.hsk_track_stream(self.h, reader.h, int(first), int(count), _fp(poses), tracked.ctypes.data_as(C.POINTER(C.c_int))))
        return poses.reshape(count, 4, 4), tracked.astype(bool)

    def flush_weights(self):
        """write the deferred free-space weights back into the volume (what a read-out does first); enqueued only"""
        self._ck(self.lib.hsk_flush_weights(self.h))

    def prepare_readout(self, product_bytes=0):
        """allocate now what a read-out would allocate on its first use (pinned staging, row tables, product buffer)"""
        self._ck(self.lib.hsk_prepare_readout(self.h, int(product_bytes)))

    def reset(self):
        self._ck(self.lib.hsk_reset(self.h))

    # ---- stages ------------------------------------------------------------------------------------
    def integrate(self, depth, pose):
        d = self._depth(depth)
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        self._ck(self.lib.hsk_integrate(self.h, d.ctypes.data, d.shape[1], d.shape[0], _fp(p)))

    def count_updates(self, depth, pose):
        d = self._depth(depth)
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        n = C.c_uint64()
        self._ck(self.lib.hsk_count_updates(self.h, d.ctypes.data, d.shape[1], d.shape[0], _fp(p), C.byref(n)))
        return n.value

    def integrate_coarse_counts(self):
        """verdicts of the last integrate's coarse level over the wave-chunks: (mixed, settled as a whole, free but worked by
        pass A, chunks currently quiet)"""
        c = (C.c_uint64 * 4)()
        self._ck(self.lib.hsk_integrate_coarse_counts(self.h, c))
        return tuple(int(x) for x in c)

    def integrate_queue_entries(self):
        """lane-blocks the last integrate's classification pass handed to its per-voxel pass"""
        n = C.c_uint64()
        self._ck(self.lib.hsk_integrate_queue_entries(self.h, C.byref(n)))
        return n.value

    def submit_host_us(self, reset=False):
        """host microseconds the pipelined submissions have spent by phase (staging copy, upload + preprocessing enqueue, wait
        for the preprocessing, main chain enqueue) and their count"""
        us = (C.c_double * 4)()
        n = C.c_uint64()
        self._ck(self.lib.hsk_submit_host_us(self.h, us, C.byref(n), int(reset)))
        return list(us), n.value

    def integrate_light_entries(self):
        """lane-blocks of the last integrate's light class (free space with holes in the depth image under it)"""
        n = C.c_uint64()
        self._ck(self.lib.hsk_integrate_light_entries(self.h, C.byref(n)))
        return n.value

    def raycast(self, pose, want_keys=False):
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        v = np.empty((3, self.hgt, self.w), np.float32)
        n = np.empty((3, self.hgt, self.w), np.float32)
        keys = np.empty((self.hgt, self.w), np.int32) if want_keys else None
        self._ck(self.lib.hsk_raycast(self.h, _fp(p), v.ctypes.data, n.ctypes.data, keys.ctypes.data if want_keys else None))
        return (v, n, keys) if want_keys else (v, n)

    def preprocess(self, depth):
        d = self._depth(depth)
        self._ck(self.lib.hsk_preprocess(self.h, d.ctypes.data, d.shape[1], d.shape[0]))

    def icp_accumulate(self, level, pose_est, row0=0, row1=None):
        p = np.ascontiguousarray(pose_est, np.float32).reshape(16)
        if row1 is None:
            row1 = self.hgt >> level
        out = np.empty(27, np.float64)
        self._ck(self.lib.hsk_icp_accumulate(self.h, level, _fp(p), row0, row1, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def icp_solve(self, sums27):
        s = np.ascontiguousarray(sums27, np.float64)
        x = np.empty(6, np.float32)
        ok = C.c_int()
        self._ck(self.lib.hsk_icp_solve(s.ctypes.data_as(C.POINTER(C.c_double)), _fp(x), C.byref(ok)))
        return x, bool(ok.value)

    def download_tsdf(self, out=None):
        """the stored planes as a row-major [nz, Y, X, 2] int16 array (into `out` when given: no fresh pages to fault in)"""
        if out is None:
            out = np.empty((self.stored_nz, self.cfg.vol_y, self.cfg.vol_x, 2), np.int16)
        want = self.stored_nz * self.cfg.vol_y * self.cfg.vol_x * 2
        if not (isinstance(out, np.ndarray) and out.dtype == np.int16 and out.flags.c_contiguous and out.flags.writeable and out.size == want):
            raise ValueError(f"download_tsdf(out=...): a writeable C-contiguous int16 array of {want} elements is needed (the C side takes a bare pointer)")
        self._ck(self.lib.hsk_download_tsdf(self.h, out.ctypes.data))
        return out

    def upload_tsdf(self, vol):
        v = np.ascontiguousarray(vol, np.int16)
        if v.size != self.stored_nz * self.cfg.vol_y * self.cfg.vol_x * 2:
            raise ValueError(f"upload_tsdf: {self.stored_nz * self.cfg.vol_y * self.cfg.vol_x * 2} int16 elements are needed, got {v.size}")
        self._ck(self.lib.hsk_upload_tsdf(self.h, v.ctypes.data))

    def get_pose(self):
        p = np.empty(16, np.float32)
        self._ck(self.lib.hsk_get_pose(self.h, _fp(p)))
        return p.reshape(4, 4)

    def set_pose(self, pose):
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        self._ck(self.lib.hsk_set_pose(self.h, _fp(p)))

    def download_map(self, kind, level):
        out = np.empty((3, self.hgt >> level, self.w >> level), np.float32)
        self._ck(self.lib.hsk_download_map(self.h, kind, level, out.ctypes.data))
        return out

    def upload_map(self, kind, level, arr):
        a = np.ascontiguousarray(arr, np.float32)
        self._ck(self.lib.hsk_upload_map(self.h, kind, level, a.ctypes.data))

    def download_depth_level(self, level):
        out = np.empty((self.hgt >> level, self.w >> level), np.uint16)
        self._ck(self.lib.hsk_download_depth_level(self.h, level, out.ctypes.data))
        return out

    def download_scaled_depth(self):
        out = np.empty((self.hgt, self.w), np.float32)
        self._ck(self.lib.hsk_download_scaled_depth(self.h, out.ctypes.data))
        return out

    def extract_cloud(self, cap=None):
        n = C.c_size_t()
        self._ck(self.lib.hsk_extract_cloud(self.h, None, 0, C.byref(n)))
        total = n.value
        m = total if cap is None else min(cap, total)
        out = np.empty((m, 3), np.float32)
        if m:
            self._ck(self.lib.hsk_extract_cloud(self.h, out.ctypes.data, m, C.byref(n)))
        return out, total

    def extract_mesh(self, cap=None, cubes=False):
        """TSDF zero level set as triangles [n, 3, 3], voxel order -> (triangles, total): marching tetrahedra, or
        (cubes=True) marching cubes, the form upstream's .ply export has"""
        fn = self.lib.hsk_extract_mesh_cubes if cubes else self.lib.hsk_extract_mesh
        n = C.c_size_t()
        self._ck(fn(self.h, None, 0, C.byref(n)))
        total = n.value
        m = total if cap is None else min(cap, total)
        out = np.empty((m, 3, 3), np.float32)
        if m:
            self._ck(fn(self.h, out.ctypes.data, m, C.byref(n)))
        return out, total

    # ---- streams / profiling -----------------------------------------------------------------------
    def stream(self):
        return self.lib.hsk_stream(self.h)

    def set_stream(self, stream_ptr):
        self._ck(self.lib.hsk_set_stream(self.h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self._ck(self.lib.hsk_synchronize(self.h))

    def set_profiling(self, on):
        self._ck(self.lib.hsk_set_profiling(self.h, int(on)))

    def icp_level_ms(self):
        """profiling: summed time of each ICP level's iterations (index 0 = finest), over the frames of stage_ms()"""
        ms = (C.c_double * _lib.HSK_LEVELS)()
        self._ck(self.lib.hsk_icp_level_ms(self.h, ms))
        return list(ms)

    def stage_ms(self, reset=False):
        ms = (C.c_double * _lib.HSK_NSTAGES)()
        n = C.c_uint64()
        self._ck(self.lib.hsk_stage_ms(self.h, ms, C.byref(n), int(reset)))
        return list(ms), n.value


GROUP_FORCE_RCCL = 1
GROUP_ICP_ALLREDUCE = 2
GROUP_DIRECT = 4      # composites as a one-hop exchange over peer-mapped memory (no RCCL)
GROUP_PROFILE = 8     # events round the exchange (exchange_ms)


class KinfuGroup:
    """One TSDF volume sharded as z-slabs over several GPUs, behind one frame call (`hsk_group_*`): the slab frame loop
    and its RCCL collectives live inside the library.  `device_ids` names the device of every slab (single process), or
    pass rank / world / comm_id for one process per GPU."""

    def __init__(self, cfg=None, device_ids=(0,), flags=0, rank=None, world=None, comm_id=None, **over):
        self.lib = _lib.load()
        self.cfg = cfg if cfg is not None else default_config(**over)
        h = C.c_void_p()
        if rank is None:
            ids = (C.c_int * len(device_ids))(*device_ids)
            rc = self.lib.hsk_group_create(C.byref(self.cfg), len(device_ids), ids, int(flags), C.byref(h))
        else:
            buf = C.create_string_buffer(bytes(comm_id), 128) if comm_id is not None else None
            rc = self.lib.hsk_group_create_rank(C.byref(self.cfg), int(rank), int(world), buf, int(flags), C.byref(h))
        if rc != 0:
            raise KinfuError(f"hsk_group_create failed ({rc}): {self.lib.hsk_group_last_error(None).decode()}")
        self.h = h
        self.w, self.hgt = self.cfg.width, self.cfg.height

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        rc = _lib.load().hsk_group_unique_id(buf)
        if rc != 0:
            raise KinfuError(f"hsk_group_unique_id failed ({rc}): {_lib.load().hsk_group_last_error(None).decode()}")
        return buf.raw

    def _ck(self, rc):
        if rc != 0:
            raise KinfuError(f"hskinfu group error {rc}: {self.lib.hsk_group_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.hsk_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_frame(self, depth):
        d = KinfuTracker._depth(depth)
        pose = np.empty(16, np.float32)
        tracked = C.c_int()
        self._ck(self.lib.hsk_group_process_frame(self.h, d.ctypes.data, d.shape[1], d.shape[0], _fp(pose), C.byref(tracked)))
        return pose.reshape(4, 4), bool(tracked.value)

    def submit_frame(self, depth):
        d = KinfuTracker._depth(depth)
        self._ck(self.lib.hsk_group_submit_frame(self.h, d.ctypes.data, d.shape[1], d.shape[0]))

    def submit_frame_dev(self, device_ptrs):
        arr = (C.c_void_p * len(device_ptrs))(*device_ptrs)
        self._ck(self.lib.hsk_group_submit_frame_dev(self.h, arr, self.w, self.hgt))

    def wait_frame(self):
        pose = np.empty(16, np.float32)
        tracked = C.c_int()
        self._ck(self.lib.hsk_group_wait_frame(self.h, _fp(pose), C.byref(tracked)))
        return pose.reshape(4, 4), bool(tracked.value)

    def reset(self):
        self._ck(self.lib.hsk_group_reset(self.h))

    def n_slabs(self):
        return self.lib.hsk_group_n_slabs(self.h)

    def ranks_seen(self):
        """ranks / devices the group's exchange spans, as the communicator (or the shared flag page) itself counts them"""
        n = C.c_int()
        self._ck(self.lib.hsk_group_ranks_seen(self.h, C.byref(n)))
        return n.value

    def exchange_ms(self):
        """GROUP_PROFILE: (summed ms of the per-frame exchange, summed ms of the slab work before it, frames counted) on
        the first local device"""
        ms, front, n = C.c_double(), C.c_double(), C.c_ulonglong()
        self._ck(self.lib.hsk_group_exchange_ms(self.h, C.byref(ms), C.byref(front), C.byref(n)))
        return ms.value, front.value, n.value

    def slab(self, i):
        """a borrowed KinfuTracker view of slab i (owned by the group: never close it)"""
        t = KinfuTracker.__new__(KinfuTracker)
        t.lib, t.cfg, t.h = self.lib, self.cfg, C.c_void_p(self.lib.hsk_group_slab(self.h, i))
        t.w, t.hgt = self.w, self.hgt
        z0, nz = C.c_int(), C.c_int()
        self.lib.hsk_stored_planes(t.h, C.byref(z0), C.byref(nz))
        t.stored_z0, t.stored_nz = z0.value, nz.value
        t.close = lambda: None
        return t

    def download_tsdf(self, out=None):
        """the planes this process owns, at their place in a full [Z, Y, X, 2] array"""
        if out is None:
            out = np.zeros((self.cfg.vol_z, self.cfg.vol_y, self.cfg.vol_x, 2), np.int16)
        self._ck(self.lib.hsk_group_download_tsdf(self.h, out.ctypes.data))
        return out


def synth_pose(frame):
    lib = _lib.load()
    p = np.empty(16, np.float32)
    lib.hsk_synth_pose(int(frame), _fp(p))
    return p.reshape(4, 4)


def synth_depth(pose, w=640, h=480, fx=525.0, fy=525.0, cx=319.5, cy=239.5):
    lib = _lib.load()
    p = np.ascontiguousarray(pose, np.float32).reshape(16)
    d = np.empty((h, w), np.uint16)
    rc = lib.hsk_synth_render(_fp(p), w, h, fx, fy, cx, cy, d.ctypes.data)
    if rc != 0:
        raise KinfuError(f"hsk_synth_render failed ({rc})")
    return d


def synth_noisy_frames(count, first=0, sigma_mm=1.2, dropout=0.02, seeds=(1234, 5678)):
    """SURVEY.md 8(d)'s noise run: the scripted stream with sensor noise -- sigma = 1.2 mm x (z / 1 m)^2 on every pixel
    (generator seed 1234) and 2 % of the pixels dropped to 0 (seed 5678), what a real takeDepthSnapshot frame looks like
    (/root/reference/housescan/HoniHelper.hs:20-36) where the render is exact.  Frames `first .. first + count - 1`; the
    generators are drawn from frame 0 on, so frame k is the same whatever `first` is.  -> (ground-truth poses, uint16 frames)"""
    rn, rd = np.random.default_rng(seeds[0]), np.random.default_rng(seeds[1])
    poses, frames = [], []
    for k in range(first + count):
        gt = synth_pose(k)
        d = synth_depth(gt).astype(np.float64)
        z = d / 1000.0
        d = d + rn.normal(size=d.shape) * sigma_mm * z * z
        d[rd.random(d.shape) < dropout] = 0
        if k >= first:
            poses.append(gt)
            frames.append(np.clip(np.rint(d), 0, 65535).astype(np.uint16))
    return poses, frames


def synth_sensor_depth(pose, scene=-1, seed=1234, sigma_mm=1.2, range_cut_m=3.5, absorbing=False, w=640, h=480, fx=525.0, fy=525.0, cx=319.5, cy=239.5):
    """one frame with holes as a structured-light sensor makes them (hsk_synth_render_sensor: grazing rays, shadow bands behind
    depth discontinuities, the 3.5 m range cut, sigma_mm x z^2 noise); scene -1 = the open scene of synth_depth, 0..3 = the
    closed rooms -> (uint16 depth, share of pixels without depth)"""
    p = np.ascontiguousarray(pose, np.float32).reshape(16)
    d = np.empty((h, w), np.uint16)
    frac = C.c_double()
    rc = _lib.load().hsk_synth_render_sensor(int(scene), _fp(p), w, h, fx, fy, cx, cy, int(seed), float(sigma_mm), float(range_cut_m), int(bool(absorbing)),
                                             d.ctypes.data, C.byref(frac))
    if rc != 0:
        raise KinfuError(f"hsk_synth_render_sensor failed ({rc})")
    return d, frac.value


def synth_sensor_frames(count, first=0, seed=1234, sigma_mm=1.2, range_cut_m=3.5, absorbing=False, room=None, scan=720):
    """the scripted stream of SURVEY.md 8(d) -- or, room = 0..3, the three-turn scan inside a closed room -- seen by a sensor
    (synth_sensor_depth; frame k's noise is keyed by seed + k) -> (ground-truth poses, uint16 frames)"""
    if room is None:
        poses = [synth_pose(k) for k in range(first, first + count)]
    else:
        poses = [synth_room_pose(room, k, scan) for k in range(first, first + count)]
    return poses, [synth_sensor_depth(p, -1 if room is None else room, seed + first + i, sigma_mm, range_cut_m, absorbing)[0] for i, p in enumerate(poses)]


def synth_room_extents(variant):
    """(x0, x1, y0, y1, z0, z1) of closed synthetic room `variant` in its own scan frame"""
    e = np.empty(6, np.float32)
    _lib.load().hsk_synth_room_extents(int(variant), _fp(e))
    return e


def synth_room_pose(variant, frame, n_frames):
    p = np.empty(16, np.float32)
    rc = _lib.load().hsk_synth_room_pose(int(variant), int(frame), int(n_frames), _fp(p))
    if rc != 0:
        raise KinfuError(f"hsk_synth_room_pose failed ({rc})")
    return p.reshape(4, 4)


def synth_room_depth(variant, pose, w=640, h=480, fx=525.0, fy=525.0, cx=319.5, cy=239.5):
    p = np.ascontiguousarray(pose, np.float32).reshape(16)
    d = np.empty((h, w), np.uint16)
    rc = _lib.load().hsk_synth_room_render(int(variant), _fp(p), w, h, fx, fy, cx, cy, d.ctypes.data)
    if rc != 0:
        raise KinfuError(f"hsk_synth_room_render failed ({rc})")
    return d


def bilateral_tables():
    lib = _lib.load()
    ws = np.empty(169, np.float32)
    wc = np.empty(512, np.float32)
    lib.hsk_bilateral_tables(_fp(ws), _fp(wc))
    return ws, wc
