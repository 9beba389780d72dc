"""Pins of the CPU oracle (runs without a GPU).  PARITY UNPINNED against PCL KinFu: the reference holds no
KinFu source or golden vectors.  What pins the oracle instead:
  (i)   an independent numpy restatement (tests/np_twin.py) -- bit-exact,
  (ii)  the analytic ground truth of the synthetic scene (known planes, known poses),
  (iii) committed golden vectors generated FROM THE TWIN by tests/golden/make_golden.py (regression / cross-machine pin)."""
import os

import numpy as np
import pytest

import np_twin as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kinfu_golden.npz")
W, H, FX, CX, CY = 160, 120, 131.25, 79.75, 59.75


def small_cfg(O, n=32):
    return O.default_config(n, W=W, H=H, fx=FX, fy=FX, cx=CX, cy=CY)


def small_depth(hsk, k):
    return hsk.synth_depth(hsk.synth_pose(k), W, H, FX, FX, CX, CY)


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.dtype.kind == "f":
        return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a).view(np.uint32),
                                                                          np.nan_to_num(b).view(np.uint32))
    return np.array_equal(a, b)


def test_twin_scale_depth_vmap_nmap_pyrdown(oracle, hsk):
    cfg = small_cfg(oracle)
    d = small_depth(hsk, 5).copy()
    d[10:20, 30:50] = 0
    assert same_bits(oracle.scale_depth(cfg, d), T.scale_depth(d, FX, FX, CX, CY))
    vm = oracle.vmap(cfg, d)
    assert same_bits(vm, T.vmap(d, FX, FX, CX, CY))
    assert same_bits(oracle.nmap(vm), T.nmap(vm))
    assert same_bits(oracle.pyrdown(d), T.pyrdown(d))


@pytest.mark.parametrize("n", [24, 40])
def test_twin_integrate(oracle, hsk, n):
    cfg = small_cfg(oracle, n)
    a = np.zeros((n, n, n, 2), np.int16)
    b = a.copy()
    for k in (0, 9, 18):
        d = small_depth(hsk, k)
        sc = oracle.scale_depth(cfg, d)
        pose = hsk.synth_pose(k)
        na = oracle.integrate(cfg, a, sc, pose)
        nb = T.integrate_full(b, (3.0, 3.0, 3.0), 0.03, sc, FX, FX, CX, CY, pose)
        assert na == nb and na > 0
        assert np.array_equal(a, b)
    # slab form == the same planes of the full volume
    s = np.zeros((10, n, n, 2), np.int16)
    full = np.zeros((n, n, n, 2), np.int16)
    d = small_depth(hsk, 0)
    sc = oracle.scale_depth(cfg, d)
    oracle.integrate(cfg, full, sc, hsk.synth_pose(0))
    oracle.integrate(cfg, s, sc, hsk.synth_pose(0), zs0=7)
    assert np.array_equal(s, full[7:17])


def test_twin_icp_sums(oracle, hsk):
    cfg = small_cfg(oracle)
    p0, p1 = hsk.synth_pose(0), hsk.synth_pose(2)
    d0, d1 = small_depth(hsk, 0), small_depth(hsk, 2)
    vm0 = oracle.vmap(cfg, d0)
    vprev, nprev = oracle.transform_maps(vm0, oracle.nmap(vm0), p0)
    vm1 = oracle.vmap(cfg, d1)
    nm1 = oracle.nmap(vm1)
    est = p0.copy()
    s_o, n_o = oracle.icp_accumulate(cfg, 0, vm1, nm1, vprev, nprev, est, p0)
    s_t, n_t = T.icp_sums(vm1, nm1, vprev, nprev, FX, FX, CX, CY, est, p0, cfg.dist_thresh, cfg.angle_thresh)
    assert n_o == n_t and n_o > 5000
    assert np.array_equal(s_o.view(np.uint64), s_t.view(np.uint64))   # exact: order-independent sums
    # the solve moves the estimate towards the true pose
    x, ok = oracle.icp_solve(s_o)
    assert ok
    new = oracle.pose_update(est, x)
    assert np.linalg.norm(new[:3, 3] - p1[:3, 3]) < np.linalg.norm(est[:3, 3] - p1[:3, 3])


def test_twin_icp_sums_extremes(oracle):
    """ties of the scaled products (k + 1/2: rint must go to even), products of 2^45, sums up to 2^27: oracle and twin agree
    bit for bit"""
    from icp_extremes import extreme_maps
    cfg = small_cfg(oracle)
    W, H = cfg.W, cfg.H
    vcur, ncur, vmod, nmod = extreme_maps(W, H, FX, FX, CX, CY)
    eye = np.eye(4, dtype=np.float32)
    s_o, n_o = oracle.icp_accumulate(cfg, 0, vcur, ncur, vmod, nmod, eye, eye)
    s_t, n_t = T.icp_sums(vcur, ncur, vmod, nmod, FX, FX, CX, CY, eye, eye, cfg.dist_thresh, cfg.angle_thresh)
    total = int((~np.isnan(ncur[0])).sum())
    assert n_o == n_t and n_o == total
    assert np.array_equal(s_o.view(np.uint64), s_t.view(np.uint64))
    assert np.all(np.abs(s_o) < 2.0 ** 27)                      # the exactness bound of the snapped sums


def test_sincos_matches_libm(oracle):
    for x in np.concatenate([np.linspace(-0.2, 0.2, 41), np.linspace(-7, 7, 57), [1e-9, 0.0, 1234.5]]):
        s, c = oracle.sincos(float(x))
        assert abs(s - np.sin(x)) < 5e-16 * max(1, abs(x)) and abs(c - np.cos(x)) < 5e-16 * max(1, abs(x))


def test_solve_against_numpy(oracle):
    rng = np.random.default_rng(3)
    J = rng.normal(size=(200, 6))
    r = rng.normal(size=200)
    A, b = J.T @ J, J.T @ r
    s27 = []
    for i in range(6):
        s27 += list(A[i, i:]) + [b[i]]
    x, ok = oracle.icp_solve(np.array(s27))
    assert ok and np.allclose(x, np.linalg.solve(A, b), rtol=1e-5, atol=1e-6)
    assert not oracle.icp_solve(np.zeros(27))[1]
    assert not oracle.icp_solve(np.full(27, np.nan))[1]


def test_analytic_scene_and_raycast(oracle, hsk):
    """fuse one frame of the analytic scene; the TSDF zero crossing and the raycast land on the true surfaces"""
    n = 96
    cfg = small_cfg(oracle, n)
    pose = hsk.synth_pose(0)
    d = small_depth(hsk, 0)
    assert d[H // 2, W // 2] == 3100          # back wall z = 2.8 seen from z = -0.3
    assert d.min() == 1500                    # block front face z = 1.2
    vol = np.zeros((n, n, n, 2), np.int16)
    oracle.integrate(cfg, vol, oracle.scale_depth(cfg, d), pose)
    cell = 3.0 / n
    col = vol[:, n // 2 + 3, n // 2 - 9, 0].astype(np.int32)   # a column that reaches the back wall
    zc = int(np.where((col[:-1] > 0) & (col[1:] < 0))[0][-1])
    assert abs((zc + 1.0) * cell - 2.8) < 1.5 * cell
    vm, nm, keys, steps = oracle.raycast(cfg, vol, pose)
    zcam = vm[2] - pose[2, 3]
    valid = ~np.isnan(zcam) & (d > 0)
    assert valid.mean() > 0.5
    assert np.median(np.abs(zcam[valid] * 1000 - d[valid])) < 0.5 * cell * 1000
    nvalid = ~np.isnan(nm[0])
    assert np.abs(np.linalg.norm(nm[:, nvalid], axis=0) - 1).max() < 1e-5
    # normals on the back wall point towards the camera (-z)
    back = nvalid & (np.abs(vm[2] - 2.8) < 2 * cell)
    assert back.sum() > 100 and np.median(nm[2][back]) < -0.95


def test_tracker_follows_ground_truth(oracle, hsk):
    cfg = oracle.default_config(96, W=320, H=240, fx=262.5, fy=262.5, cx=159.75, cy=119.75)
    trk = oracle.Tracker(cfg, omp=True)
    for k in range(10):
        gt = hsk.synth_pose(k)
        pose, ok = trk.process(hsk.synth_depth(gt, 320, 240, 262.5, 262.5, 159.75, 119.75))
        assert ok == (k > 0)
        dt = np.linalg.norm(pose[:3, 3] - gt[:3, 3]) * 1000
        ang = np.degrees(np.arccos(np.clip((np.trace(pose[:3, :3].astype(np.float64).T @ gt[:3, :3]) - 1) / 2, -1, 1)))
        assert dt < 5.0 and ang < 0.2, (k, dt, ang)   # stated tolerance: 5 mm / 0.2 deg at 96^3 (31 mm cells)


def test_tracking_lost_resets(oracle, hsk):
    cfg = small_cfg(oracle)
    trk = oracle.Tracker(cfg)
    trk.process(small_depth(hsk, 0))
    pose, ok = trk.process(np.zeros((H, W), np.uint16))
    assert not ok and not trk.volume().any() and np.allclose(pose[:3, 3], [1.5, 1.5, -0.3])


def test_golden_vectors(oracle, hsk):
    """committed fixtures (tests/golden/make_golden.py): regression + cross-machine pin of the oracle"""
    g = np.load(GOLD)
    cfg = small_cfg(oracle, int(g["n"]))
    trk = oracle.Tracker(cfg)
    poses = []
    for k in g["frames"]:
        poses.append(trk.process(small_depth(hsk, int(k)))[0])
    assert np.array_equal(np.stack(poses).view(np.uint32), g["poses"].view(np.uint32))
    assert np.array_equal(trk.volume()[8:24, 8:24, 8:24], g["tsdf_crop"])
    assert same_bits(trk.model_map(2, 0)[:, 40:70, 60:100], g["vmap_crop"])
    assert same_bits(trk.model_map(3, 0)[:, 40:70, 60:100], g["nmap_crop"])
    assert np.array_equal(small_depth(hsk, int(g["frames"][1])), g["depth1"])
    d = small_depth(hsk, 4)
    vm = oracle.vmap(cfg, oracle.bilateral(cfg, d))
    s, _ = oracle.icp_accumulate(cfg, 0, vm, oracle.nmap(vm), trk.model_map(2, 0), trk.model_map(3, 0), poses[-1], poses[-1])
    assert np.array_equal(s.view(np.uint64), g["icp27"].view(np.uint64))
    # vectors added in round 2 (the file is generated by the numpy twin: tests/golden/make_golden.py)
    assert same_bits(trk.model_map(2, 2), g["vmap2"]) and same_bits(trk.model_map(3, 2), g["nmap2"])
    assert np.array_equal(oracle.bilateral(cfg, small_depth(hsk, int(g["frames"][1]))), g["bilateral1"])
    x6, ok = oracle.icp_solve(g["icp27"])
    assert ok and np.array_equal(x6.view(np.uint32), g["solve6"].view(np.uint32))
    assert np.array_equal(oracle.pose_update(poses[-1], x6).view(np.uint32), g["pose_after_solve"].view(np.uint32))
    _, _, keys, _ = oracle.raycast(cfg, trk.volume(), poses[-1])
    assert np.array_equal(keys[40:70, 60:100], g["keys_crop"])
    pts, total = oracle.extract_cloud(cfg, trk.volume())
    assert total == int(g["cloud_count"]) and same_bits(pts[:256], g["cloud_head"])


# ---- round 2: every remaining stage has a second, independently written restatement (tests/np_twin.py) ----
def _libm_acosf():
    import ctypes
    m = ctypes.CDLL("libm.so.6")
    m.acosf.restype = ctypes.c_float
    m.acosf.argtypes = [ctypes.c_float]
    return lambda c: m.acosf(float(c))


def test_twin_bilateral(oracle, hsk):
    cfg = small_cfg(oracle)
    rng = np.random.default_rng(21)
    d = small_depth(hsk, 3).copy()
    d = (d.astype(np.int32) + rng.integers(-12, 13, d.shape)).clip(0, 65535).astype(np.uint16)
    d[rng.random(d.shape) < 0.03] = 0
    d[30:44, 60:90] = 0
    d[0:5, 0:7] = 40000          # beyond the 32767 clamp of the result, at a clipped corner window
    assert same_bits(oracle.bilateral(cfg, d), T.bilateral(d))
    ws, wc = hsk.bilateral_tables()
    tws, twc = T.bilateral_tables()
    assert np.array_equal(ws.reshape(13, 13), tws) and np.array_equal(wc, twc)


def test_twin_resize_and_transform(oracle, hsk):
    cfg = small_cfg(oracle)
    d = small_depth(hsk, 6).copy()
    d[50:60, 20:40] = 0
    vm = oracle.vmap(cfg, d)
    nm = oracle.nmap(vm)
    pose = hsk.synth_pose(6)
    vo, no = oracle.transform_maps(vm, nm, pose)
    tv, tn = T.transform_maps(vm, nm, pose)
    assert same_bits(vo, tv) and same_bits(no, tn)
    assert same_bits(oracle.resize_vmap(vo), T.resize_vmap(vo))
    assert same_bits(oracle.resize_nmap(no), T.resize_nmap(no))
    assert same_bits(oracle.resize_nmap(oracle.resize_nmap(no)), T.resize_nmap(T.resize_nmap(no)))


def test_twin_sincos_solve_pose_update_gate(oracle, hsk):
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.normal(scale=0.02, size=200), rng.uniform(-7, 7, 200), [0.0, -0.0, 1e-12, 1.0e5, -3.0e7, np.nan, 99999.5]])
    for x in xs:
        so, co = oracle.sincos(float(x))
        st, ct = T.sincos(x)
        assert np.float64(so).tobytes() == np.float64(st).tobytes() and np.float64(co).tobytes() == np.float64(ct).tobytes(), x
    for trial in range(60):
        J = rng.normal(size=(40, 6)) * rng.uniform(0.01, 30, size=6)
        if trial % 5 == 0:
            J[:, 5] = J[:, 4] * (1 + 1e-7 * rng.normal(size=40))      # nearly dependent columns
        A, b = J.T @ J, J.T @ rng.normal(size=40)
        s27 = np.array(sum([list(A[i, i:]) + [b[i]] for i in range(6)], []))
        xo, oko = oracle.icp_solve(s27)
        xt, okt = T.icp_solve(s27)
        assert oko == okt and np.array_equal(xo.view(np.uint32), xt.view(np.uint32)), trial
    for bad in (np.zeros(27), np.full(27, np.nan), -np.ones(27)):
        assert not oracle.icp_solve(bad)[1] and not T.icp_solve(bad)[1]
    acosf = _libm_acosf()
    pose = hsk.synth_pose(4)
    for _ in range(40):
        x6 = (rng.normal(size=6) * [0.01, 0.01, 0.01, 0.005, 0.005, 0.005]).astype(np.float32)
        po = oracle.pose_update(pose, x6)
        pt = T.pose_update(pose, x6)
        assert np.array_equal(po.view(np.uint32), pt.view(np.uint32))
        pose = po


def test_solve_and_sincos_against_exact_rational_arithmetic(oracle):
    """an anchor that shares no arithmetic with either restatement: the 6x6 system solved in exact rationals, sin / cos
    from their Taylor series in exact rationals"""
    from fractions import Fraction as Fr
    rng = np.random.default_rng(8)
    for trial in range(25):
        J = rng.normal(size=(30, 6))
        if trial % 4 == 0:
            J[:, 3] = J[:, 2] + 1e-4 * rng.normal(size=30)             # condition number ~1e8
        A, b = J.T @ J, J.T @ rng.normal(size=30)
        s27 = np.array(sum([list(A[i, i:]) + [b[i]] for i in range(6)], []))
        x, ok = oracle.icp_solve(s27)
        assert ok
        M = [[Fr(float(A[min(i, j), max(i, j)])) for j in range(6)] + [Fr(float(b[i]))] for i in range(6)]
        for c in range(6):                                             # Gauss-Jordan in exact arithmetic
            p = next(r for r in range(c, 6) if M[r][c] != 0)
            M[c], M[p] = M[p], M[c]
            M[c] = [v / M[c][c] for v in M[c]]
            for r in range(6):
                if r != c:
                    M[r] = [vr - M[r][c] * vc for vr, vc in zip(M[r], M[c])]
        exact = np.array([float(M[i][6]) for i in range(6)])
        cond = np.linalg.cond(A)
        assert np.abs(x - exact).max() <= (8 * cond * 2.0 ** -53 + 2.0 ** -24) * np.abs(exact).max(), (trial, cond)
    for x in list(np.linspace(-0.3, 0.3, 31)) + [1.0, -2.5, 3.0, 6.0]:
        fx = Fr(float(x))
        s = sum((-1) ** k * fx ** (2 * k + 1) / Fr(int(np.prod(np.arange(1, 2 * k + 2, dtype=object)))) for k in range(30))
        c = sum((-1) ** k * fx ** (2 * k) / Fr(int(np.prod(np.arange(1, 2 * k + 1, dtype=object)))) for k in range(30))
        so, co = oracle.sincos(float(x))
        assert abs(Fr(so) - s) <= Fr(3, 2 ** 54) and abs(Fr(co) - c) <= Fr(3, 2 ** 54), x


def test_twin_gate(oracle, hsk):
    """the gate's own arithmetic (twin vs the oracle tracker's decisions): repeated frames gate off, moving frames on"""
    acosf = _libm_acosf()
    p0, p1 = hsk.synth_pose(0), hsk.synth_pose(1)
    assert T.gate_passes(p1, p0, 0.004, acosf) and not T.gate_passes(p0, p0, 0.004, acosf) and T.gate_passes(p0, p0, 0.0, acosf)
    thr = 0.004
    cfg = small_cfg(oracle, 32)
    cfg.move_thresh = thr
    ot = oracle.Tracker(cfg)
    tt = T.Tracker(32, W, H, FX, FX, CX, CY, move_thresh=thr, acosf=acosf)
    for k in (0, 1, 1, 2, 3, 3):
        d = small_depth(hsk, k)
        po, oko = ot.process(d)
        pt, okt = tt.process(d)
        assert oko == okt and np.array_equal(po.view(np.uint32), pt.view(np.uint32)), k
    assert np.array_equal(ot.volume(), tt.vol)
    assert 1 < int(tt.vol[..., 1].max()) < 6       # six frames: some integrated, at least one repeat gated off


@pytest.mark.parametrize("n,slab", [(32, None), (48, (10, 30, 14, 24))])
def test_twin_raycast(oracle, hsk, n, slab):
    cfg = small_cfg(oracle, n)
    vol = np.zeros((n, n, n, 2), np.int16)
    for k in (0, 4, 8):
        d = small_depth(hsk, k)
        oracle.integrate(cfg, vol, oracle.scale_depth(cfg, d), hsk.synth_pose(k))
    inside = np.array([[0.94, 0, 0.34, 1.2], [0, 1, 0, 1.4], [-0.34, 0, 0.94, 0.6], [0, 0, 0, 1]], np.float32)
    for pose in (hsk.synth_pose(8), hsk.synth_pose(40), inside):
        if slab is None:
            o = oracle.raycast(cfg, vol, pose)
            t = T.raycast(vol, (3.0, 3.0, 3.0), 0.03, W, H, FX, FX, CX, CY, pose)
        else:
            zs0, zs1, zo0, zo1 = slab
            part = np.ascontiguousarray(vol[zs0:zs1])
            o = oracle.raycast(cfg, part, pose, zs0=zs0, zo0=zo0, zo1=zo1)
            t = T.raycast(part, (3.0, 3.0, 3.0), 0.03, W, H, FX, FX, CX, CY, pose, Z=n, zs0=zs0, zo0=zo0, zo1=zo1)
        assert np.array_equal(o[2], t[2]), "step keys"
        assert same_bits(o[0], t[0]) and same_bits(o[1], t[1])
        assert o[3] == t[3] and o[3] > 1000
        assert (o[2] != 0x7FFFFFFF).mean() > (0.2 if slab is None else 0.01)   # a slab ends only the rays that end in it


def test_twin_extract_cloud(oracle, hsk):
    n = 40
    cfg = small_cfg(oracle, n)
    vol = np.zeros((n, n, n, 2), np.int16)
    for k in (0, 5, 10):
        d = small_depth(hsk, k)
        oracle.integrate(cfg, vol, oracle.scale_depth(cfg, d), hsk.synth_pose(k))
    pts, total = oracle.extract_cloud(cfg, vol)
    tp = T.extract_cloud(vol, (3.0, 3.0, 3.0))
    assert total == len(tp) > 500 and same_bits(pts, tp)


def test_twin_tracker_equals_oracle_tracker(oracle, hsk):
    """the whole A.2 state machine, twin vs oracle, poses / TSDF / model maps bit for bit (and a lost frame)"""
    cfg = small_cfg(oracle, 32)
    ot = oracle.Tracker(cfg)
    tt = T.Tracker(32, W, H, FX, FX, CX, CY)
    for k in (0, 2, 4):
        d = small_depth(hsk, k)
        po, oko = ot.process(d)
        pt, okt = tt.process(d)
        assert oko == okt and np.array_equal(po.view(np.uint32), pt.view(np.uint32)), k
    assert np.array_equal(ot.volume(), tt.vol)
    for l in range(3):
        assert same_bits(ot.model_map(2, l), tt.vmod[l]) and same_bits(ot.model_map(3, l), tt.nmod[l])
    po, oko = ot.process(np.zeros((H, W), np.uint16))
    pt, okt = tt.process(np.zeros((H, W), np.uint16))
    assert not oko and not okt and np.array_equal(po, pt) and not tt.vol.any()


def _fuzz_pose(rng, lookat):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    ang = rng.uniform(-np.pi, np.pi)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    P = np.eye(4, dtype=np.float32)
    P[:3, 3] = (1.5 + rng.uniform(-1, 1, 3) * (2.4 if not lookat else 1.2)).astype(np.float32)
    if lookat:
        z = 1.5 + rng.uniform(-0.4, 0.4, 3) - P[:3, 3]
        z /= np.linalg.norm(z) + 1e-9
        x = np.cross(rng.normal(size=3), z)
        x /= np.linalg.norm(x)
        R = np.stack([x, np.cross(z, x), z], axis=1)
    P[:3, :3] = R.astype(np.float32)
    return P


def _fuzz_depth(rng):
    by, bx = int(rng.integers(1, 17)), int(rng.integers(1, 17))
    coarse = rng.integers(300, 5000, size=(H // by + 2, W // bx + 2))
    d = np.kron(coarse, np.ones((by, bx), np.int64))[:H, :W]
    d = d + rng.integers(-20, 21, size=(H, W)) * (rng.random((H, W)) < 0.5)
    d[rng.random((H, W)) < 0.1] = 0
    ex = rng.random((H, W))
    d[ex < 0.002] = 1
    d[ex > 0.998] = 65535
    return np.clip(d, 0, 65535).astype(np.uint16)


@pytest.mark.parametrize("seed", range(3))
def test_twin_fuzz_integrate_and_raycast(oracle, hsk, seed):
    """oracle == twin on inputs nobody tuned either for: blocky random depth (discontinuities, holes, 1 mm and 65535 mm
    pixels) and arbitrary camera poses (inside / outside the volume, any orientation)"""
    rng = np.random.default_rng(3000 + seed)
    n = 32
    cfg = small_cfg(oracle, n)
    a = np.zeros((n, n, n, 2), np.int16)
    b = a.copy()
    total = 0
    for k in range(6):
        pose = _fuzz_pose(rng, k % 2 == 1)
        sc = oracle.scale_depth(cfg, _fuzz_depth(rng))
        na = oracle.integrate(cfg, a, sc, pose)
        nb = T.integrate_full(b, (3.0, 3.0, 3.0), 0.03, sc, FX, FX, CX, CY, pose)
        assert na == nb
        assert np.array_equal(a, b)
        total += na
    assert total > 5000
    hits = 0
    for k in range(2):
        pose = _fuzz_pose(rng, True)
        o = oracle.raycast(cfg, a, pose)
        t = T.raycast(a, (3.0, 3.0, 3.0), 0.03, W, H, FX, FX, CX, CY, pose)
        assert np.array_equal(o[2], t[2]), "step keys"
        assert same_bits(o[0], t[0]) and same_bits(o[1], t[1]) and o[3] == t[3]
        hits += int((o[2] != 0x7FFFFFFF).sum())
    assert hits > 200
