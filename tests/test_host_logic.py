"""Host-side pieces that need no GPU: synthetic stream, product writers on the file seam HouseScan reads."""
import os

import numpy as np


def test_synth_stream_is_deterministic_and_physical(hsk):
    p0 = hsk.synth_pose(0)
    assert np.allclose(p0, [[1, 0, 0, 1.5], [0, 1, 0, 1.5], [0, 0, 1, -0.3], [0, 0, 0, 1]], atol=1e-7)
    d = hsk.synth_depth(p0)
    assert d.shape == (480, 640) and d.dtype == np.uint16
    assert np.array_equal(d, hsk.synth_depth(p0))
    assert d[239, 319] == 3100 and d.min() == 1500 and (d > 0).all()
    # side wall x = 2.8 seen at the right image edge: z = (2.8 - 1.5) / ((639 - 319.5) / 525)
    assert abs(int(d[239, 639]) - round(1000 * 1.3 / ((639 - 319.5) / 525))) <= 1
    # trajectory: <= 7 mm and <= 0.6 deg per frame, rotation orthonormal
    for k in (1, 37, 150, 299):
        a, b = hsk.synth_pose(k), hsk.synth_pose(k + 1)
        assert np.linalg.norm(a[:3, 3] - b[:3, 3]) < 0.007
        assert np.allclose(a[:3, :3] @ a[:3, :3].T, np.eye(3), atol=1e-6)
        ang = np.degrees(np.arccos(np.clip((np.trace(a[:3, :3].T @ b[:3, :3]) - 1) / 2, -1, 1)))
        assert ang < 0.6
    assert np.allclose(hsk.synth_pose(300), hsk.synth_pose(0), atol=1e-6)   # period 300


def test_pcd_writer_and_downsample(tmp_path, hsk):
    """cloud_downsampled.pcd / cloud_bin.pcd in the layout HouseScan's loader (PCD.loadXyz) expects"""
    import ctypes as C
    lib = hsk._lib.load()
    rng = np.random.default_rng(1)
    pts = rng.uniform(0, 1, size=(5000, 3)).astype(np.float32)
    path = str(tmp_path / "cloud_bin.pcd")
    assert lib.hsk_write_pcd_xyz(path.encode(), pts.ctypes.data, len(pts)) == 0
    raw = open(path, "rb").read()
    head, body = raw.split(b"DATA binary\n", 1)
    h = head.decode()
    assert "FIELDS x y z" in h and "SIZE 4 4 4" in h and "TYPE F F F" in h and f"POINTS {len(pts)}" in h
    assert np.array_equal(np.frombuffer(body, np.float32).reshape(-1, 3), pts)
    out = np.empty((5000, 3), np.float32)
    n = C.c_size_t()
    assert lib.hsk_voxel_downsample(pts.ctypes.data, len(pts), C.c_float(0.25), out.ctypes.data, 5000, C.byref(n)) == 0
    assert n.value == 64     # 4^3 occupied leaves
    cells = np.floor(out[:n.value] / 0.25).astype(int)
    assert len({tuple(c) for c in cells}) == 64
    ref = pts[(np.floor(pts / 0.25).astype(int) == cells[0]).all(axis=1)].astype(np.float64).mean(axis=0)
    assert np.allclose(out[0], ref, atol=1e-6)


def test_oracle_is_not_reachable_from_the_product(hsk):
    """the product package must never import or link the oracle (it is test infrastructure)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "housescan_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dp, f)).read()
                assert "kinfu_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, (dp, f)


def _box_cloud(rng):
    def face(n, axis, val, lo, hi):
        p = rng.uniform(lo, hi, size=(n, 3))
        p[:, axis] = val + rng.normal(0, 0.002, n)
        return p
    return np.concatenate([
        face(4000, 0, 0.2, [0, 0.3, 0], [0, 2.7, 2.8]), face(4000, 0, 2.8, [0, 0.3, 0], [0, 2.7, 2.8]),
        face(5000, 1, 0.3, [0.2, 0, 0], [2.8, 0, 2.8]), face(5000, 1, 2.7, [0.2, 0, 0], [2.8, 0, 2.8]),
        face(6000, 2, 2.8, [0.2, 0.3, 0], [2.8, 2.7, 0]), rng.uniform(0, 3, size=(800, 3))]).astype(np.float32)


def _match_walls(planes, walls, tol=0.02):
    """every expected wall (axis, coordinate) has a detected plane with |n.e| > 0.99 and the right offset"""
    for axis, coord in walls:
        ok = False
        for a, b, c, d in planes:
            n = np.array([a, b, c])
            if abs(n[axis]) > 0.99 and abs(-d / n[axis] - coord) < tol:
                ok = True
        assert ok, (axis, coord, planes)


def test_plane_detection_and_room_dir(tmp_path, hsk):
    """the files HouseScan's loadRoom reads: cloud_downsampled.pcd, planes.txt ("a b c d" per line, PCL sign
    convention -- /root/reference/housescan/Main.hs:1379-1389), cloud_plane_hull<k>.pcd"""
    from housescan_amd import products as P
    pts = _box_cloud(np.random.default_rng(0))
    planes, labels = P.detect_planes(pts)
    assert len(planes) == 5
    assert np.allclose(np.linalg.norm(planes[:, :3], axis=1), 1, atol=1e-5)
    _match_walls(planes, [(0, 0.2), (0, 2.8), (1, 0.3), (1, 2.7), (2, 2.8)], tol=0.005)
    p2, l2 = P.detect_planes(pts)
    assert np.array_equal(planes, p2) and np.array_equal(labels, l2)        # deterministic
    room = tmp_path / "room1"
    planes_w, n_down = P.write_room_dir(str(room), pts, leaf=0.05)
    names = sorted(os.listdir(room))
    assert "cloud_bin.pcd" in names and "cloud_downsampled.pcd" in names and "planes.txt" in names
    # parse planes.txt the way planeEqsFromFile does: lines split on \n, four whitespace-separated decimals
    lines = open(room / "planes.txt").read().split("\n")
    parsed = np.array([[float(t) for t in ln.split()] for ln in lines])
    assert parsed.shape == (len(planes_w), 4) and np.allclose(parsed, planes_w, atol=1e-6)
    for k, eq in enumerate(planes_w):
        raw = open(room / f"cloud_plane_hull{k}.pcd", "rb").read()
        body = raw.split(b"DATA binary\n", 1)[1]
        hull = np.frombuffer(body, np.float32).reshape(-1, 3)
        assert len(hull) >= 4
        assert np.abs(hull @ eq[:3] + eq[3]).max() < 1e-4        # vertices lie on the plane
        # convex polygon in drawing order: all turns have the same orientation about the normal
        e = np.roll(hull, -1, axis=0) - hull
        turns = np.einsum("ij,j->i", np.cross(e, np.roll(e, -1, axis=0)), eq[:3])
        assert (turns > -1e-6).all() or (turns < 1e-6).all()
    assert 0 < n_down < len(pts)


def test_xf_round_trip_and_apply(tmp_path, hsk):
    """row-major left-multiplicative 4x4 in HouseScan's two export layouts (Main.hs:2271-2302)"""
    from housescan_amd import products as P
    th = np.radians(30.0)
    m = np.array([[np.cos(th), 0, np.sin(th), 1.5], [0, 1, 0, -0.25], [-np.sin(th), 0, np.cos(th), 2.0], [0, 0, 0, 1]], np.float32)
    P.write_xf(str(tmp_path / "room.xf"), m)
    txt = open(tmp_path / "room.xf").read().strip().split("\n")
    assert len(txt) == 4 and all(len(r.split()) == 4 for r in txt)          # roomProjectionToXfFormat layout
    assert np.array_equal(P.read_xf(str(tmp_path / "room.xf")), m)
    open(tmp_path / "room.csv", "w").write(",".join(repr(float(x)) for x in m.reshape(-1)))   # roomProjectionToString
    assert np.allclose(P.read_xf(str(tmp_path / "room.csv")), m)
    pts = np.random.default_rng(2).uniform(-1, 1, (1000, 3)).astype(np.float32)
    out = P.transform_cloud(pts, m)
    ref = (m[:3, :3].astype(np.float64) @ pts.T.astype(np.float64)).T + m[:3, 3]
    assert np.allclose(out, ref, atol=1e-5)                                   # p' = M p (left-multiplicative)


def test_depth_stream_container(tmp_path, hsk):
    from housescan_amd import products as P
    path = str(tmp_path / "scan.hskd")
    frames = [hsk.synth_depth(hsk.synth_pose(k), 160, 120, 131.25, 131.25, 79.75, 59.75) for k in range(5)]
    w = P.DepthStreamWriter(path, 160, 120, 131.25, 131.25, 79.75, 59.75)
    for f in frames:
        w.write(f)
    w.close()
    r = P.DepthStreamReader(path)
    assert len(r) == 5 and (r.w, r.hgt) == (160, 120) and r.intr == (131.25, 131.25, 79.75, 59.75)
    for k in (4, 0, 2):
        assert np.array_equal(r[k], frames[k])
    import pytest
    with pytest.raises(IndexError):
        r[5]
    r.close()
    assert os.path.getsize(path) == 36 + 5 * 160 * 120 * 2


def test_exact_division_shortcuts_used_by_the_kernels():
    """hsk_dev.h replaces two correctly-rounded f32 divisions of the specification by binary64 products:
    raw / 32767 (exhaustive over every int16 raw) and x / n for n = weight + 1 <= 129 with a table of correctly rounded
    binary64 reciprocals (the proof is in the header; here: every n, many x, and the reciprocal perturbed by a few ulps to
    show the margin -- v_rcp_f64 itself is far coarser than that and failed the 130-frame GPU parity test)"""
    raw = np.arange(-32768, 32768, dtype=np.int32)
    want = raw.astype(np.float32) / np.float32(32767.0)
    got = (raw.astype(np.float64) * (1.0 / 32767.0)).astype(np.float32)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    rng = np.random.default_rng(11)
    # numerators as the update forms them: Fp * Wp + F with Fp, F in [-1, 1] on the 1/32767 grid and arbitrary floats too
    xs = np.concatenate([
        (rng.integers(-32767, 32768, 200000).astype(np.float32) / np.float32(32767.0)) * rng.integers(1, 129, 200000).astype(np.float32)
        + rng.uniform(-1, 1, 200000).astype(np.float32),
        rng.normal(0, 50, 100000).astype(np.float32),
        np.float32(2.0) ** rng.integers(-20, 20, 2000).astype(np.float32),
        np.arange(1, 4000, dtype=np.float32),
    ])
    for n in range(1, 130):
        want = xs / np.float32(n)
        r = 1.0 / np.float64(n)
        for ulps in (-3, 0, 3):
            rr = r
            for _ in range(abs(ulps)):
                rr = np.nextafter(rr, np.inf if ulps > 0 else -np.inf)
            got = (xs.astype(np.float64) * rr).astype(np.float32)
            assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), (n, ulps)


def test_division_by_a_fixed_cell_size_as_binary64_product():
    """hsk_div_by_const: x / c (binary32) == float32(float64(x) * (1 / float64(c))) for the cell sizes in use and
    adversarial numerators (the argument is in hsk_dev.h)"""
    rng = np.random.default_rng(12)
    cells = [np.float32(3.0) / np.float32(n) for n in (64, 96, 128, 256, 512, 1024)] + [np.float32(3.0) / np.float32(100), np.float32(0.0123)]
    for c in cells:
        rc = 1.0 / np.float64(c)
        xs = np.concatenate([
            rng.uniform(-4, 4, 400000).astype(np.float32),
            (np.arange(-2000, 2000).astype(np.float32) * c),                       # exact multiples
            (np.arange(-2000, 2000).astype(np.float32) + np.float32(0.5)) * c,     # half-cell offsets
            np.nextafter((np.arange(1, 2000).astype(np.float32) * c), np.float32(np.inf)),
            np.nextafter((np.arange(1, 2000).astype(np.float32) * c), np.float32(-np.inf)),
        ])
        want = xs / c
        got = (xs.astype(np.float64) * rc).astype(np.float32)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), float(c)
