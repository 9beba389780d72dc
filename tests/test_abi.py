"""The C-ABI library loads without a GPU and exports every symbol include/hskinfu.h declares; the ctypes
mirror of hsk_config has the C layout; without a HIP device the product fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hskinfu.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hsk_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(hsk):
    from housescan_amd import _lib
    names = declared_functions()
    assert len(names) >= 40
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/hskinfu.h but not exported by libhskinfu.so"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes binding in housescan_amd/_lib.py"
    assert set(_lib.SYMBOLS) <= set(names), set(_lib.SYMBOLS) - set(names)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (hsk_\w+)", out))
    assert set(names) <= exported


def test_house_header_symbols_are_exported_and_bound(hsk):
    """include/hshouse.h (host-side room stitching): same rule as the core header"""
    from housescan_amd import _lib
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "hshouse.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(hsh_[a-z0-9_]+)\s*\(", src)))
    assert len(names) >= 40
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/hshouse.h but not exported"
        assert n in _lib.HOUSE_SYMBOLS, f"{n} has no ctypes binding"
    assert set(_lib.HOUSE_SYMBOLS) <= set(names), set(_lib.HOUSE_SYMBOLS) - set(names)


def test_house_header_is_plain_c(tmp_path):
    src = tmp_path / "c.c"
    src.write_text('#include "hshouse.h"\nint main(void){hsh_house* h = 0; (void)h; return 0;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c",
                           str(src), "-o", str(tmp_path / "c.o")])


def test_config_struct_layout_matches_c(tmp_path, hsk):
    from housescan_amd import _lib
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hskinfu.h"\nint main(){printf("%zu %zu %zu %zu\\n", '
                   'sizeof(hsk_config), offsetof(hsk_config, init_pose), offsetof(hsk_config, own_z0), '
                   'offsetof(hsk_config, use_graph));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    size, off_pose, off_own, off_graph = map(int, subprocess.check_output([str(exe)], text=True).split())
    assert size == C.sizeof(_lib.HskConfig)
    assert off_pose == _lib.HskConfig.init_pose.offset
    assert off_own == _lib.HskConfig.own_z0.offset
    assert off_graph == _lib.HskConfig.use_graph.offset


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "c.c"
    src.write_text('#include "hskinfu.h"\nint main(void){hsk_config c; hsk_ctx* k = 0; (void)c; (void)k; return 0;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c",
                           str(src), "-o", str(tmp_path / "c.o")])


def test_default_config_values(hsk):
    c = hsk.default_config(512)
    assert (c.vol_x, c.vol_y, c.vol_z) == (512, 512, 512)
    assert list(c.vol_size_m) == [3.0, 3.0, 3.0] and abs(c.trunc_dist_m - 0.03) < 1e-9
    assert (c.width, c.height, c.fx, c.fy, c.cx, c.cy) == (640, 480, 525.0, 525.0, 319.5, 239.5)
    assert list(c.icp_iters) == [10, 5, 4]
    assert abs(c.icp_angle_thresh_sin - np.sin(np.radians(20))) < 1e-7
    pose = np.array(c.init_pose, np.float32).reshape(4, 4)
    assert np.allclose(pose[:3, :3], np.eye(3)) and np.allclose(pose[:3, 3], [1.5, 1.5, -0.3], atol=1e-6)
    assert (c.own_z0, c.own_z1, c.halo, c.use_graph) == (0, 512, 0, 0)


def test_no_gpu_fails_loudly(hsk):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(hsk.KinfuError, match="no HIP device"):
        hsk.KinfuTracker(n=64)


def test_group_without_gpu_fails_loudly(hsk):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(hsk.KinfuError, match="no HIP device|device setup failed"):
        hsk.KinfuGroup(hsk.default_config(64), device_ids=[0, 0])


def test_host_solve_mirror_matches_oracle(hsk, oracle):
    """hsk_icp_solve is host code: the device solve's mirror must equal the oracle's bit for bit"""
    import ctypes
    lib = hsk._lib.load()
    rng = np.random.default_rng(11)
    for _ in range(20):
        J = rng.normal(size=(50, 6))
        A, b = J.T @ J, J.T @ rng.normal(size=50)
        s27 = np.array(sum([list(A[i, i:]) + [b[i]] for i in range(6)], []))
        x = np.empty(6, np.float32)
        ok = ctypes.c_int()
        lib.hsk_icp_solve(s27.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                          ctypes.byref(ok))
        xo, oko = oracle.icp_solve(s27)
        assert ok.value == 1 and oko and np.array_equal(x.view(np.uint32), xo.view(np.uint32))


def test_bilateral_tables_exact(hsk):
    ws, wc = hsk.bilateral_tables()
    s2, c2 = np.float32(0.5) / (np.float32(4.5) * np.float32(4.5)), np.float32(0.5) / (np.float32(30) * np.float32(30))
    dy, dx = np.mgrid[-6:7, -6:7]
    ref_s = np.exp(-((dx * dx + dy * dy).astype(np.float32) * s2).astype(np.float64)).astype(np.float32).reshape(-1)
    k = np.arange(512)
    ref_c = np.exp(-((k * k).astype(np.float32) * c2).astype(np.float64)).astype(np.float32)
    assert np.array_equal(ws, ref_s) and np.array_equal(wc, ref_c)


def test_native_rooms_harness_builds_against_the_header(tmp_path, hsk):
    """tools/rooms_native.c (bench.py's concurrent_rooms_one_gpu block compiles and runs it on the GPU box) is plain C11 against
    include/hskinfu.h and links with nothing but the library, libdl and pthreads"""
    from housescan_amd import _lib
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "rooms_native")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-pthread", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "rooms_native.c"), "-L" + lib_dir, "-lhskinfu", "-ldl", "-Wl,-rpath," + lib_dir,
                           "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    r = subprocess.run([exe, "0"], capture_output=True, text=True, timeout=60)   # (0 rooms: the argument check, no GPU touched)
    assert r.returncode == 2

