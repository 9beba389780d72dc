"""ctypes wrapper of the CPU oracle (oracle/kinfu_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (housescan_amd) never does.  PARITY UNPINNED -- see kinfu_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
KEY_NONE = 0x7FFFFFFF


class OraConfig(C.Structure):
    _fields_ = [
        ("vol", C.c_int * 3), ("size", C.c_float * 3), ("trunc", C.c_float),
        ("W", C.c_int), ("H", C.c_int),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("icp_iters", C.c_int * 3),
        ("dist_thresh", C.c_float), ("angle_thresh", C.c_float), ("move_thresh", C.c_float),
        ("init_R", C.c_float * 9), ("init_t", C.c_float * 3),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_libs = {}


def build_native():
    """-O3 -march=native -fopenmp build for bench.py's cpu_baseline leg, made ON the machine that
    times it (oracle/_native/, never shipped); returns its path, or None when it cannot be built here"""
    try:
        subprocess.check_call(["make", "-s", "-C", _HERE, "native"])
    except (OSError, subprocess.CalledProcessError):
        return None
    path = os.path.join(_HERE, "_native", "libkinfu_oracle_native.so")
    return path if os.path.exists(path) else None


def build_literal():
    """the Appendix-A-literal form (kinfu_oracle.c built with -DORA_LITERAL and FMA contraction allowed): the yardstick of
    tools/spec_vs_literal.py; built on the machine that runs it"""
    subprocess.check_call(["make", "-s", "-C", _HERE, "literal"])
    return os.path.join(_HERE, "_native", "libkinfu_oracle_literal.so")


def build_variant(name, defs="", contract="off"):
    """one deviation of the literal form at a time (tools/spec_vs_literal.py --attribute); load with lib("var:" + name)"""
    subprocess.check_call(["make", "-s", "-C", _HERE, "variant", "NAME=" + name, "DEFS=" + defs, "CONTRACT=" + contract])


def lib(omp=False):
    name = "libkinfu_oracle_omp.so" if omp else "libkinfu_oracle.so"
    if isinstance(omp, str) and omp.startswith("var:"):
        name = "_native/libkinfu_oracle_var_%s.so" % omp[4:]
    if omp == "native":
        name = "_native/libkinfu_oracle_native.so"
    if omp == "literal":
        name = "_native/libkinfu_oracle_literal.so"
    if name in _libs:
        return _libs[name]
    path = os.path.join(_HERE, name)
    if omp == "literal":
        build_literal()   # (make: rebuilt when kinfu_oracle.c changed)
    elif not os.path.exists(path):
        if omp == "native":
            if build_native() is None:
                return lib(True)
        else:
            build()
    L = C.CDLL(path)
    L.ora_tau.restype = C.c_float
    L.ora_integrate.restype = C.c_uint64
    L.ora_icp_accumulate.restype = C.c_uint64
    L.ora_extract_cloud.restype = C.c_size_t
    L.ora_extract_mesh.restype = C.c_size_t
    L.ora_tracker_create.restype = C.c_void_p
    L.ora_tracker_volume.restype = C.POINTER(C.c_int16)
    L.ora_tracker_model_vmap.restype = C.POINTER(C.c_float)
    L.ora_tracker_model_nmap.restype = C.POINTER(C.c_float)
    L.ora_tracker_last_vupd.restype = C.c_uint64
    _libs[name] = L
    return L


def default_config(n=512, omp=False, **over):
    c = OraConfig()
    lib(omp).ora_default_config(C.byref(c), int(n))
    for k, v in over.items():
        if k in ("vol", "size", "icp_iters", "init_R", "init_t"):
            arr = getattr(c, k)
            for i, x in enumerate(np.asarray(v).reshape(-1)):
                arr[i] = x
        else:
            setattr(c, k, v)
    return c


def tau(cfg):
    return float(lib().ora_tau(C.byref(cfg)))


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _rt(pose):
    p = np.asarray(pose, np.float32).reshape(4, 4)
    return np.ascontiguousarray(p[:3, :3]).reshape(9), np.ascontiguousarray(p[:3, 3])


def _intr(cfg, level=0):
    s = np.float32(1 << level)
    return (C.c_float(np.float32(cfg.fx) / s), C.c_float(np.float32(cfg.fy) / s), C.c_float(np.float32(cfg.cx) / s),
            C.c_float(np.float32(cfg.cy) / s))


def scale_depth(cfg, depth):
    d = np.ascontiguousarray(depth, np.uint16)
    out = np.empty((cfg.H, cfg.W), np.float32)
    lib().ora_scale_depth(_p(d), cfg.W, cfg.H, *_intr(cfg), _f(out))
    return out


def integrate(cfg, vol, scaled, pose, zs0=0, omp=False):
    """vol: int16 [nz, Y, X, 2] modified in place; returns V_upd."""
    R, t = _rt(pose)
    dims = (C.c_int * 3)(*cfg.vol)
    size = (C.c_float * 3)(*cfg.size)
    assert vol.dtype == np.int16 and vol.flags.c_contiguous
    return int(lib(omp).ora_integrate(_p(vol), dims, size, C.c_float(tau(cfg)), zs0, vol.shape[0], _f(scaled),
                                      cfg.W, cfg.H, *_intr(cfg), _f(R), _f(t)))


def bilateral(cfg, depth, omp=False):
    d = np.ascontiguousarray(depth, np.uint16)
    out = np.empty_like(d)
    lib(omp).ora_bilateral(_p(d), d.shape[1], d.shape[0], _p(out))
    return out


def pyrdown(depth):
    d = np.ascontiguousarray(depth, np.uint16)
    out = np.empty((d.shape[0] // 2, d.shape[1] // 2), np.uint16)
    lib().ora_pyrdown(_p(d), d.shape[1], d.shape[0], _p(out))
    return out


def vmap(cfg, depth, level=0):
    d = np.ascontiguousarray(depth, np.uint16)
    out = np.empty((3,) + d.shape, np.float32)
    lib().ora_vmap(_p(d), d.shape[1], d.shape[0], *_intr(cfg, level), _f(out))
    return out


def nmap(vm):
    out = np.empty_like(vm)
    lib().ora_nmap(_f(vm), vm.shape[2], vm.shape[1], _f(out))
    return out


def transform_maps(vm, nm, pose):
    R, t = _rt(pose)
    vo, no = np.empty_like(vm), np.empty_like(nm)
    lib().ora_transform_maps(_f(vm), _f(nm), vm.shape[2], vm.shape[1], _f(R), _f(t), _f(vo), _f(no))
    return vo, no


def resize_vmap(vm):
    out = np.empty((3, vm.shape[1] // 2, vm.shape[2] // 2), np.float32)
    lib().ora_resize_vmap(_f(vm), vm.shape[2], vm.shape[1], _f(out))
    return out


def resize_nmap(nm):
    out = np.empty((3, nm.shape[1] // 2, nm.shape[2] // 2), np.float32)
    lib().ora_resize_nmap(_f(nm), nm.shape[2], nm.shape[1], _f(out))
    return out


def icp_accumulate(cfg, level, vcur, ncur, vprev, nprev, pose, pose_prev, row0=0, row1=None, omp=False):
    R, t = _rt(pose)
    Rp, tp = _rt(pose_prev)
    H, W = vcur.shape[1], vcur.shape[2]
    if row1 is None:
        row1 = H
    out = np.empty(27, np.float64)
    n = lib(omp).ora_icp_accumulate(_f(vcur), _f(ncur), _f(vprev), _f(nprev), W, H, *_intr(cfg, level), _f(R), _f(t), _f(Rp),
                                 _f(tp), C.c_float(cfg.dist_thresh), C.c_float(cfg.angle_thresh), row0, row1,
                                 out.ctypes.data_as(C.POINTER(C.c_double)))
    return out, int(n)


def icp_solve(sums27, omp=False):
    s = np.ascontiguousarray(sums27, np.float64)
    x = np.empty(6, np.float32)
    ok = lib(omp).ora_icp_solve(s.ctypes.data_as(C.POINTER(C.c_double)), _f(x))
    return x, bool(ok)


def pose_update(pose, x6, omp=False):
    R, t = _rt(pose)
    x = np.ascontiguousarray(x6, np.float32)
    lib(omp).ora_pose_update(_f(R), _f(t), _f(x))
    out = np.eye(4, dtype=np.float32)
    out[:3, :3] = R.reshape(3, 3)
    out[:3, 3] = t
    return out


def sincos(x):
    s, c = C.c_double(), C.c_double()
    lib().ora_sincos(C.c_double(x), C.byref(s), C.byref(c))
    return s.value, c.value


def raycast(cfg, vol, pose, zs0=0, zo0=0, zo1=None, omp=False):
    """vol: int16 [nzs, Y, X, 2]; returns vmap, nmap, keys, n_steps."""
    R, t = _rt(pose)
    if zo1 is None:
        zo1 = cfg.vol[2]
    dims = (C.c_int * 3)(*cfg.vol)
    size = (C.c_float * 3)(*cfg.size)
    vm = np.empty((3, cfg.H, cfg.W), np.float32)
    nm = np.empty((3, cfg.H, cfg.W), np.float32)
    keys = np.empty((cfg.H, cfg.W), np.int32)
    ns = C.c_uint64()
    lib(omp).ora_raycast(_p(vol), dims, size, C.c_float(tau(cfg)), zs0, vol.shape[0], zo0, zo1, cfg.W, cfg.H,
                         *_intr(cfg), _f(R), _f(t), _f(vm), _f(nm), _p(keys), C.byref(ns))
    return vm, nm, keys, ns.value


def extract_cloud(cfg, vol, cap=None):
    dims = (C.c_int * 3)(*cfg.vol)
    size = (C.c_float * 3)(*cfg.size)
    n = lib().ora_extract_cloud(_p(vol), dims, size, None, 0)
    m = n if cap is None else min(cap, n)
    out = np.empty((m, 3), np.float32)
    if m:
        lib().ora_extract_cloud(_p(vol), dims, size, _f(out), m)
    return out, int(n)


def extract_mesh(cfg, vol, cap=None, cubes=False):
    """triangle soup of the zero level set: marching tetrahedra (A.8), or marching cubes (A.8b) with cubes=True"""
    dims = (C.c_int * 3)(*cfg.vol)
    size = (C.c_float * 3)(*cfg.size)
    fn = lib().ora_extract_mesh_mc if cubes else lib().ora_extract_mesh
    n = fn(_p(vol), dims, size, None, 0)
    m = n if cap is None else min(cap, n)
    out = np.empty((m, 3, 3), np.float32)
    if m:
        fn(_p(vol), dims, size, _f(out), m)
    return out, int(n)


def mc_table():
    """the oracle's generated marching-cubes table: (ntri[256], codes[256, 5, 3]) -- edge code = low corner | high corner << 4"""
    ntri = np.zeros(256, np.int32)
    codes = np.zeros((256, 5, 3), np.int32)
    lib().ora_mc_table(ntri.ctypes.data_as(C.POINTER(C.c_int)), codes.ctypes.data_as(C.POINTER(C.c_int)))
    return ntri, codes


class Tracker:
    def __init__(self, cfg, omp=False):
        self.L = lib(omp)
        self.cfg = cfg
        self.h = C.c_void_p(self.L.ora_tracker_create(C.byref(cfg)))

    def process(self, depth):
        d = np.ascontiguousarray(depth, np.uint16)
        pose = np.empty(16, np.float32)
        tracked = self.L.ora_tracker_process(self.h, _p(d), _f(pose))
        return pose.reshape(4, 4), bool(tracked)

    def volume(self):
        X, Y, Z = self.cfg.vol
        p = self.L.ora_tracker_volume(self.h)
        return np.ctypeslib.as_array(p, shape=(Z, Y, X, 2))

    def model_map(self, kind, level):
        fn = self.L.ora_tracker_model_vmap if kind == 2 else self.L.ora_tracker_model_nmap
        p = fn(self.h, level)
        return np.ctypeslib.as_array(p, shape=(3, self.cfg.H >> level, self.cfg.W >> level))

    def stage_seconds(self):
        s = (C.c_double * 4)()
        self.L.ora_tracker_stage_seconds(self.h, s)
        return list(s)

    def last_vupd(self):
        return int(self.L.ora_tracker_last_vupd(self.h))

    def close(self):
        if self.h:
            self.L.ora_tracker_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
