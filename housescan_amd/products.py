"""Products on the file seam between the KinFu core and HouseScan (host side).

`write_room_dir` produces exactly the directory `loadRoom` reads (/root/reference/housescan/Main.hs:1738-1762):
  cloud_downsampled.pcd   XYZ float PCD                               (Main.hs:1740, :1334-1345)
  planes.txt              "a b c d" per line, PCL form ax+by+cz+d=0   (Main.hs:1379-1389)
  cloud_plane_hull<k>.pcd polygon of plane k, vertices in drawing order (Main.hs:1395-1400)
plus cloud_bin.pcd, the full-resolution cloud HouseScan's printed pcl_transform_point_cloud commands act on
(Main.hs:2311-2313, :2436-2438).
"""
import ctypes as C
import os

import numpy as np

from . import _lib


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc})")


def voxel_downsample(xyz, leaf):
    lib = _lib.load()
    pts = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    out = np.empty_like(pts)
    n = C.c_size_t()
    _ck(lib.hsk_voxel_downsample(pts.ctypes.data, len(pts), C.c_float(leaf), out.ctypes.data, len(pts), C.byref(n)), "hsk_voxel_downsample")
    return out[:n.value].copy()


def write_pcd(path, xyz):
    lib = _lib.load()
    pts = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    _ck(lib.hsk_write_pcd_xyz(os.fsencode(path), pts.ctypes.data, len(pts)), "hsk_write_pcd_xyz")


def write_ply_mesh(path, triangles):
    """triangle soup [n, 3, 3] -> welded binary .ply mesh; returns (vertices, faces) written"""
    lib = _lib.load()
    tri = np.ascontiguousarray(triangles, np.float32).reshape(-1, 9)
    nv, nf = C.c_size_t(), C.c_size_t()
    _ck(lib.hsk_write_ply_mesh(os.fsencode(path), tri.ctypes.data, len(tri), C.byref(nv), C.byref(nf)), "hsk_write_ply_mesh")
    return nv.value, nf.value


def weld_triangles(triangles):
    """triangle soup [n, 3, 3] -> (vertices [v, 3] in order of first appearance, indices [n, 3])"""
    lib = _lib.load()
    tri = np.ascontiguousarray(triangles, np.float32).reshape(-1, 9)
    nv = C.c_size_t()
    verts = np.empty((max(1, 3 * len(tri)), 3), np.float32)
    idx = np.empty((len(tri), 3), np.int32)
    _ck(lib.hsk_weld_triangles(tri.ctypes.data, len(tri), verts.ctypes.data, len(verts), C.byref(nv), idx.ctypes.data), "hsk_weld_triangles")
    return verts[:nv.value].copy(), idx


def detect_planes(xyz, dist_thresh=0.02, min_fraction=0.03, max_planes=12, iterations=300):
    """-> (planes [k,4] as a,b,c,d of ax+by+cz+d=0 with unit normal, labels [n])"""
    lib = _lib.load()
    pts = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    planes = np.zeros((max_planes, 4), np.float32)
    labels = np.empty(len(pts), np.int32)
    k = C.c_int()
    _ck(lib.hsk_detect_planes(pts.ctypes.data, len(pts), C.c_float(dist_thresh), C.c_float(min_fraction), max_planes, iterations,
                              planes.ctypes.data, labels.ctypes.data, C.byref(k)), "hsk_detect_planes")
    return planes[:k.value].copy(), labels


def plane_hull(xyz, labels, plane, abcd):
    lib = _lib.load()
    pts = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    lab = np.ascontiguousarray(labels, np.int32)
    eq = np.ascontiguousarray(abcd, np.float32)
    cap = max(16, int((lab == plane).sum()))
    hull = np.empty((cap, 3), np.float32)
    n = C.c_size_t()
    _ck(lib.hsk_plane_hull(pts.ctypes.data, len(pts), lab.ctypes.data, int(plane), eq.ctypes.data_as(C.POINTER(C.c_float)),
                           hull.ctypes.data, cap, C.byref(n)), "hsk_plane_hull")
    return hull[:n.value].copy()


def write_room_dir(room_dir, cloud_xyz, leaf=0.03, **plane_args):
    """cloud (full resolution, KinFu frame) -> the files HouseScan's loadRoom expects. Returns (planes, n_downsampled)."""
    lib = _lib.load()
    os.makedirs(room_dir, exist_ok=True)
    write_pcd(os.path.join(room_dir, "cloud_bin.pcd"), cloud_xyz)
    down = voxel_downsample(cloud_xyz, leaf)
    write_pcd(os.path.join(room_dir, "cloud_downsampled.pcd"), down)
    planes, labels = detect_planes(down, **plane_args)
    _ck(lib.hsk_write_planes_txt(os.fsencode(os.path.join(room_dir, "planes.txt")), planes.ctypes.data, len(planes)),
        "hsk_write_planes_txt")
    for k, eq in enumerate(planes):
        write_pcd(os.path.join(room_dir, f"cloud_plane_hull{k}.pcd"), plane_hull(down, labels, k, eq))
    return planes, len(down)


def write_xf(path, m):
    lib = _lib.load()
    a = np.ascontiguousarray(m, np.float32).reshape(16)
    _ck(lib.hsk_write_xf(os.fsencode(path), a.ctypes.data_as(C.POINTER(C.c_float))), "hsk_write_xf")


def read_xf(path):
    lib = _lib.load()
    a = np.empty(16, np.float32)
    _ck(lib.hsk_read_xf(os.fsencode(path), a.ctypes.data_as(C.POINTER(C.c_float))), "hsk_read_xf")
    return a.reshape(4, 4)


def transform_cloud(xyz, m):
    lib = _lib.load()
    pts = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    a = np.ascontiguousarray(m, np.float32).reshape(16)
    out = np.empty_like(pts)
    _ck(lib.hsk_transform_cloud(pts.ctypes.data, len(pts), a.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data), "hsk_transform_cloud")
    return out


class DepthStreamWriter:
    """HSKD raw depth recording (frames in the layout of HoniHelper.takeDepthSnapshot)."""

    def __init__(self, path, w=640, h=480, fx=525.0, fy=525.0, cx=319.5, cy=239.5):
        self.lib = _lib.load()
        self.h = self.lib.hsk_stream_create(os.fsencode(path), w, h, fx, fy, cx, cy)
        if not self.h:
            raise RuntimeError(f"cannot create {path}")
        self.shape = (h, w)

    def write(self, depth):
        d = np.ascontiguousarray(depth, np.uint16)
        assert d.shape == self.shape
        _ck(self.lib.hsk_stream_write(self.h, d.ctypes.data), "hsk_stream_write")

    def close(self):
        if self.h:
            _ck(self.lib.hsk_stream_close(self.h), "hsk_stream_close")
            self.h = None


class DepthStreamReader:
    def __init__(self, path):
        self.lib = _lib.load()
        w, h, n = C.c_int(), C.c_int(), C.c_int()
        intr = (C.c_float * 4)()
        self.h = self.lib.hsk_stream_open(os.fsencode(path), C.byref(w), C.byref(h), C.byref(n), intr)
        if not self.h:
            raise RuntimeError(f"{path} is not an HSKD stream")
        self.w, self.hgt, self.n, self.intr = w.value, h.value, n.value, tuple(intr)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        d = np.empty((self.hgt, self.w), np.uint16)
        rc = self.lib.hsk_stream_read(self.h, int(i), d.ctypes.data)
        if rc != 0:
            raise IndexError(i)
        return d

    def close(self):
        if self.h:
            self.lib.hsk_stream_close(self.h)
            self.h = None
