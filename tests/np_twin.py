"""Independent numpy restatement of the oracle's arithmetic ("two independent restatements
agreeing is the only oracle available").  Vectorised float32 numpy: every operator rounds once, no fused
multiply-add, so the results must equal oracle/kinfu_oracle.c bit for bit.  Written from SURVEY.md Appendix A,
not from the C file."""
import numpy as np

f32 = np.float32


def scale_depth(depth, fx, fy, cx, cy):
    H, W = depth.shape
    u = np.arange(W, dtype=f32)[None, :]
    v = np.arange(H, dtype=f32)[:, None]
    xl = (u - f32(cx)) / f32(fx)
    yl = (v - f32(cy)) / f32(fy)
    lam = np.sqrt((xl * xl + yl * yl) + f32(1))
    return (depth.astype(f32) * lam) / f32(1000)


def tau_of(size, dims, trunc):
    cell = [f32(size[i]) / f32(dims[i]) for i in range(3)]
    lo = f32(2.1) * max(cell)
    return max(f32(trunc), lo)


def _integrate(vol, size, trunc, scaled, fx, fy, cx, cy, pose, zs0, Z=None):
    nz, Y, X, _ = vol.shape
    Z = Z or nz
    H, W = scaled.shape
    cell = [f32(size[0]) / f32(X), f32(size[1]) / f32(Y), f32(size[2]) / f32(Z)]
    tau = tau_of(size, (X, Y, Z), trunc)
    tau_inv = f32(1) / tau
    R = pose[:3, :3].astype(f32)
    t = pose[:3, 3].astype(f32)
    Ri = R.T.copy()
    x = np.arange(X, dtype=f32)[None, None, :]
    y = np.arange(Y, dtype=f32)[None, :, None]
    z = (np.arange(nz) + zs0).astype(f32)[:, None, None]
    gx = (x + f32(0.5)) * cell[0] - t[0]
    gy = (y + f32(0.5)) * cell[1] - t[1]
    gz = (z + f32(0.5)) * cell[2] - t[2]
    cam = [(Ri[i, 0] * gx + Ri[i, 1] * gy) + Ri[i, 2] * gz for i in range(3)]
    with np.errstate(all="ignore"):
        front = cam[2] >= f32(1.17549435e-38)       # D6: a denormal camera-space depth counts as not in front
        inv_z = f32(1) / cam[2]
        fu = (cam[0] * f32(fx)) * inv_z + f32(cx)
        fv = (cam[1] * f32(fy)) * inv_z + f32(cy)
        ok = front & (fu > f32(-1e6)) & (fu < f32(1e6)) & (fv > f32(-1e6)) & (fv < f32(1e6))
        u = np.where(ok, np.rint(np.where(ok, fu, 0)), -1).astype(np.int64)
        v = np.where(ok, np.rint(np.where(ok, fv, 0)), -1).astype(np.int64)
    ok &= (u >= 0) & (v >= 0) & (u < W) & (v < H)
    Ds = np.where(ok, scaled[np.clip(v, 0, H - 1), np.clip(u, 0, W - 1)], f32(0))
    dist = np.sqrt(gz * gz + (gx * gx + gy * gy))
    sdf = Ds - dist
    upd = ok & (Ds != 0) & (sdf >= -tau)
    F = np.minimum(sdf * tau_inv, f32(1)).astype(f32)
    tp = vol[..., 0].astype(f32)
    wp = vol[..., 1].astype(f32)
    Fp = tp / f32(32767)
    Fn = (Fp * wp + F) / (wp + f32(1))
    fixed = np.clip(np.trunc(Fn * f32(32767)), -32767, 32767)
    wn = np.minimum(vol[..., 1].astype(np.int32) + 1, 128)
    vol[..., 0] = np.where(upd, fixed, vol[..., 0]).astype(np.int16)
    vol[..., 1] = np.where(upd, wn, vol[..., 1]).astype(np.int16)
    return int(upd.sum())


def integrate_full(vol, size, trunc, scaled, fx, fy, cx, cy, pose):
    return _integrate(vol, size, trunc, scaled, fx, fy, cx, cy, pose, 0, vol.shape[0])


def vmap(depth, fx, fy, cx, cy):
    H, W = depth.shape
    z = depth.astype(f32) / f32(1000)
    u = np.arange(W, dtype=f32)[None, :]
    v = np.arange(H, dtype=f32)[:, None]
    fx_inv, fy_inv = f32(1) / f32(fx), f32(1) / f32(fy)
    X = (z * (u - f32(cx))) * fx_inv
    Y = (z * (v - f32(cy))) * fy_inv
    out = np.stack([X, Y, z]).astype(f32)
    out[:, z == 0] = np.nan
    return out


def nmap(vm):
    _, H, W = vm.shape
    out = np.full_like(vm, np.nan)
    v00, v01, v10 = vm[:, :-1, :-1], vm[:, :-1, 1:], vm[:, 1:, :-1]
    a, b = v01 - v00, v10 - v00
    r = np.stack([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])
    with np.errstate(all="ignore"):
        inv = f32(1) / np.sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2])
        n = (r * inv).astype(f32)
    bad = np.isnan(v00[0]) | np.isnan(v01[0]) | np.isnan(v10[0])
    n[:, bad] = np.nan
    out[:, :-1, :-1] = n
    return out


def pyrdown(src):
    H, W = src.shape
    h2, w2 = H // 2, W // 2
    s = src.astype(np.int64)
    out = np.zeros((h2, w2), np.uint16)
    for y in range(h2):
        for x in range(w2):
            c = s[2 * y, 2 * x]
            win = s[max(2 * y - 2, 0):min(2 * y + 2, H - 2) + 1, max(2 * x - 2, 0):min(2 * x + 2, W - 2) + 1]   # (exclusive upper clip)
            m = np.abs(win - c) < 90
            out[y, x] = win[m].sum() // m.sum()
    return out


def icp_sums(vcur, ncur, vprev, nprev, fx, fy, cx, cy, pose, pose_prev, dist_thresh, angle_thresh):
    _, H, W = vcur.shape
    R, t = pose[:3, :3].astype(f32), pose[:3, 3].astype(f32)
    Rp, tp = pose_prev[:3, :3].astype(f32), pose_prev[:3, 3].astype(f32)
    Ri = Rp.T.copy()

    def rot(M, v):
        return np.stack([(M[i, 0] * v[0] + M[i, 1] * v[1]) + M[i, 2] * v[2] for i in range(3)])

    with np.errstate(all="ignore"):
        vg = rot(R, vcur) + t[:, None, None]
        cp = rot(Ri, vg - tp[:, None, None])
        ok = ~np.isnan(ncur[0]) & (cp[2] > 0)
        fu = (cp[0] * f32(fx)) / cp[2] + f32(cx)
        fv = (cp[1] * f32(fy)) / cp[2] + f32(cy)
        ok &= (fu > f32(-1e6)) & (fu < f32(1e6)) & (fv > f32(-1e6)) & (fv < f32(1e6))
        u = np.where(ok, np.rint(np.where(ok, fu, 0)), -1).astype(np.int64)
        v = np.where(ok, np.rint(np.where(ok, fv, 0)), -1).astype(np.int64)
        ok &= (u >= 0) & (v >= 0) & (u < W) & (v < H)
        uc, vc = np.clip(u, 0, W - 1), np.clip(v, 0, H - 1)
        npg = nprev[:, vc, uc]
        vpg = vprev[:, vc, uc]
        ok &= ~np.isnan(npg[0])
        d = vpg - vg
        dist = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
        ok &= dist <= f32(dist_thresh)
        ng = rot(R, ncur)
        cr = np.stack([ng[1] * npg[2] - ng[2] * npg[1], ng[2] * npg[0] - ng[0] * npg[2], ng[0] * npg[1] - ng[1] * npg[0]])
        sine = np.sqrt((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2])
        ok &= sine < f32(angle_thresh)
        sxn = np.stack([vg[1] * npg[2] - vg[2] * npg[1], vg[2] * npg[0] - vg[0] * npg[2], vg[0] * npg[1] - vg[1] * npg[0]])
        r = (npg[0] * d[0] + npg[1] * d[1]) + npg[2] * d[2]
    row = np.concatenate([sxn, npg, r[None]]).astype(f32)[:, ok].astype(np.float64)
    out = []
    for a in range(6):
        for b in range(a, 7):
            p = row[a] * row[b]
            out.append(float((np.rint(p * 67108864.0) * (1.0 / 67108864.0)).sum()))
    return np.array(out), int(ok.sum())


# ======================================================================================================
# Round 2: the rest of the path, again written from SURVEY.md Appendix A and the deviations D1-D4 of DESIGN.md
# section 4 (the specification this build adopts), not from oracle/kinfu_oracle.c: bilateral (A.3, D1), map resize
# and tranformMaps (A.3, A.2), the 6x6 solve (A.5, D4), sin/cos and the pose update (A.2 step 3), the integration
# gate (A.2 step 5), the raycast (A.6, D3, with the slab ownership and step keys of DESIGN.md section 6), cloud
# extraction (A.7) and the tracker state machine (A.2).  Everything rounds once per operator, float32 unless said.
# ======================================================================================================
KEY_NONE = 0x7FFFFFFF
f64 = np.float64


def bilateral_tables():
    s2 = f32(0.5) / (f32(4.5) * f32(4.5))
    c2 = f32(0.5) / (f32(30) * f32(30))
    dy, dx = np.mgrid[-6:7, -6:7]
    ws = np.exp(-((dx * dx + dy * dy).astype(f32) * s2).astype(f64)).astype(f32)
    k = np.arange(512)
    wc = np.exp(-((k * k).astype(f32) * c2).astype(f64)).astype(f32)
    return ws, wc


def bilateral(src):
    """A.3 + D1: 13x13 window clipped to the image EXCLUSIVE of its last row and column (upstream's loop bounds), weight = ws(dx, dy) * wc(|dd|) (wc = 0 from 512 mm on), a zero
    centre gives 0, the sums run row by row through the window, result = rint(sum1 / sum2) clamped to [0, 32767]"""
    H, W = src.shape
    ws, wc = bilateral_tables()
    s = src.astype(np.int64)
    sum1 = np.zeros((H, W), f32)
    sum2 = np.zeros((H, W), f32)
    yy, xx = np.mgrid[0:H, 0:W]
    for dy in range(-6, 7):
        for dx in range(-6, 7):
            ny, nx = yy + dy, xx + dx
            inside = (ny >= 0) & (ny < H - 1) & (nx >= 0) & (nx < W - 1)   # (upper clip exclusive of the last row / column)
            tmp = s[np.clip(ny, 0, H - 1), np.clip(nx, 0, W - 1)]
            dd = np.abs(s - tmp)
            wcv = np.where(dd < 512, wc[np.minimum(dd, 511)], f32(0)).astype(f32)
            w = ws[dy + 6, dx + 6] * wcv
            sum1 = np.where(inside, sum1 + tmp.astype(f32) * w, sum1).astype(f32)
            sum2 = np.where(inside, sum2 + w, sum2).astype(f32)
    with np.errstate(all="ignore"):
        res = np.where(sum2 > 0, np.rint(sum1 / sum2), 0)
    res = np.clip(np.where(s == 0, 0, res), 0, 32767)
    return np.where(s == 0, 0, res).astype(np.uint16)


def _resize(m, normalize):
    a, b, c, d = m[:, 0::2, 0::2], m[:, 0::2, 1::2], m[:, 1::2, 0::2], m[:, 1::2, 1::2]
    with np.errstate(all="ignore"):
        out = ((((a + b) + c) + d) / f32(4)).astype(f32)
        if normalize:
            inv = f32(1) / np.sqrt((out[0] * out[0] + out[1] * out[1]) + out[2] * out[2])
            out = (out * inv).astype(f32)
    bad = np.isnan(a[0]) | np.isnan(b[0]) | np.isnan(c[0]) | np.isnan(d[0])
    out[:, bad] = np.nan
    return out


def resize_vmap(m):
    return _resize(m, False)


def resize_nmap(m):
    return _resize(m, True)


def _rot(M, v):
    return np.stack([(M[i, 0] * v[0] + M[i, 1] * v[1]) + M[i, 2] * v[2] for i in range(3)]).astype(f32)


def transform_maps(vm, nm, pose):
    R, t = pose[:3, :3].astype(f32), pose[:3, 3].astype(f32)
    with np.errstate(all="ignore"):
        vo = (_rot(R, vm) + t[:, None, None]).astype(f32)
        no = _rot(R, nm)
    vo[:, np.isnan(vm[0])] = np.nan
    no[:, np.isnan(nm[0])] = np.nan
    return vo, no


def _fact(n):
    out = 1
    for i in range(2, n + 1):
        out *= i
    return f64(out)


def sincos(x):
    """A.2 step 3's sines and cosines, specified as: k = rint(x * 2/pi); r = (x - k hi) - k lo with
    pi/2 = hi + lo, hi its first 33 bits; Taylor polynomials to r^15 / r^16 in Horner form; quadrant k mod 4"""
    x = f64(x)
    if not (abs(x) < 1.0e5):
        return f64(0.0), f64(1.0)
    k = np.rint(x * f64(0.63661977236758134308))
    r = (x - k * f64(1.57079632673412561417e+00)) - k * f64(6.07710050650619224932e-11)
    r2 = r * r
    ps = -(f64(1) / _fact(15))
    for n, sign in ((13, 1), (11, -1), (9, 1), (7, -1), (5, 1), (3, -1)):
        ps = ps * r2 + f64(sign) * (f64(1) / _fact(n))
    pc = f64(1) / _fact(16)
    for n, sign in ((14, -1), (12, 1), (10, -1), (8, 1), (6, -1), (4, 1)):
        pc = pc * r2 + f64(sign) * (f64(1) / _fact(n))
    sr = r + (r * r2) * ps
    cr = (f64(1) - f64(0.5) * r2) + (r2 * r2) * pc
    q = int(k) & 3
    return [(sr, cr), (cr, -sr), (-sr, -cr), (-cr, sr)][q]


def icp_solve(s27):
    """A.5 + D4: A = L D L^T with one reciprocal per pivot, binary64; returns (x6 float32, ok)"""
    s = [f64(v) for v in s27]
    A = [[None] * 6 for _ in range(6)]
    b = [None] * 6
    it = iter(s)
    for i in range(6):
        for j in range(i, 7):
            v = next(it)
            if j == 6:
                b[i] = v
            else:
                A[i][j] = A[j][i] = v
    L = [[f64(0)] * 6 for _ in range(6)]
    D, Dinv = [None] * 6, [None] * 6
    det = f64(1)
    zero = np.zeros(6, f32)
    with np.errstate(all="ignore"):
        for j in range(6):
            dj = A[j][j]
            for q in range(j):
                dj = dj - (L[j][q] * L[j][q]) * D[q]
            if not (dj > 0):
                return zero, False
            D[j], Dinv[j] = dj, f64(1) / dj
            det = det * dj
            for i in range(j + 1, 6):
                r = A[i][j]
                for q in range(j):
                    r = r - (L[i][q] * L[j][q]) * D[q]
                L[i][j] = r * Dinv[j]
        if not (det >= 1e-15):
            return zero, False
        y, x = [None] * 6, [None] * 6
        for i in range(6):
            r = b[i]
            for q in range(i):
                r = r - L[i][q] * y[q]
            y[i] = r
        for i in range(5, -1, -1):
            r = y[i] * Dinv[i]
            for q in range(i + 1, 6):
                r = r - L[q][i] * x[q]
            x[i] = r
    for v in x:
        if np.isnan(v) or not (abs(v) < 1e30):
            return zero, False
    return np.array(x, f64).astype(f32), True


def _mm(A, B):
    return np.array([[(A[i, 0] * B[0, j] + A[i, 1] * B[1, j]) + A[i, 2] * B[2, j] for j in range(3)] for i in range(3)], f32)


def pose_update(pose, x6):
    """A.2 step 3: R_inc = Rz(gamma) Ry(beta) Rx(alpha); t <- R_inc t + t_inc; R <- R_inc R (float32)"""
    x6 = np.asarray(x6, f32)
    (sa, ca), (sb, cb), (sg, cg) = [tuple(f32(v) for v in sincos(f64(x6[a]))) for a in range(3)]
    o, z = f32(1), f32(0)
    Rx = np.array([[o, z, z], [z, ca, -sa], [z, sa, ca]], f32)
    Ry = np.array([[cb, z, sb], [z, o, z], [-sb, z, cb]], f32)
    Rz = np.array([[cg, -sg, z], [sg, cg, z], [z, z, o]], f32)
    Rinc = _mm(_mm(Rz, Ry), Rx)
    R, t = pose[:3, :3].astype(f32), pose[:3, 3].astype(f32)
    out = np.eye(4, dtype=f32)
    out[:3, 3] = [((Rinc[i, 0] * t[0] + Rinc[i, 1] * t[1]) + Rinc[i, 2] * t[2]) + x6[3 + i] for i in range(3)]
    out[:3, :3] = _mm(Rinc, R)
    return out


def gate_passes(pose, prev, thr, acosf):
    """A.2 step 5: integrate iff (|rodrigues(R^-1 R_prev)| + |t - t_prev|) / 2 >= thr.  The Rodrigues norm of a rotation
    is its angle acos((trace - 1) / 2); of M = R^T R_prev only the diagonal is formed, M_ii = (R_0i Rp_0i + R_1i Rp_1i) +
    R_2i Rp_2i, trace = (M_00 + M_11) + M_22.  `acosf`: the C library's float arc cosine (passed in by the test)."""
    if not (thr > 0):
        return True
    R, Rp = pose[:3, :3].astype(f32), prev[:3, :3].astype(f32)
    d = [(R[0, i] * Rp[0, i] + R[1, i] * Rp[1, i]) + R[2, i] * Rp[2, i] for i in range(3)]
    c = (((d[0] + d[1]) + d[2]) - f32(1)) / f32(2)
    c = f32(1) if c > 1 else (f32(-1) if c < -1 else c)
    ang = f32(acosf(c))
    e = (pose[:3, 3] - prev[:3, 3]).astype(f32)
    tn = np.sqrt((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2])
    return bool((ang + tn) / f32(2) >= f32(thr))


# ---- A.6 raycast ---------------------------------------------------------------------------------------
def _vox(p, cell):
    """floor(p / cell), -1 for negative or NaN, capped at 10^6"""
    with np.errstate(all="ignore"):
        q = np.floor(p / cell)
    out = np.where(q >= 0, np.minimum(q, f32(1.0e6)), -1)
    return np.where(np.isnan(q), -1, out).astype(np.int64)


class _Grid:
    def __init__(self, vol, size, Z, zs0):
        self.nz, self.Y, self.X, _ = vol.shape
        self.Z, self.zs0 = Z, zs0
        self.raw = vol[..., 0]
        self.cell = [f32(size[0]) / f32(self.X), f32(size[1]) / f32(self.Y), f32(size[2]) / f32(Z)]

    def at(self, x, y, z):
        """stored TSDF integer of voxel (x, y, z) inside the grid; planes this slab does not store read 0"""
        zz = z - self.zs0
        ok = (zz >= 0) & (zz < self.nz)
        return np.where(ok, self.raw[np.clip(zz, 0, self.nz - 1), y, x], 0).astype(np.int64)

    def tsdf(self, x, y, z):
        return self.at(x, y, z).astype(f32) / f32(32767)

    def trilinear(self, p):
        """A.6: NaN on the outer shell; per axis the lower tap is the voxel whose centre lies at or below p"""
        g = [_vox(p[k], self.cell[k]) for k in range(3)]
        dims = (self.X, self.Y, self.Z)
        ok = np.ones(p[0].shape, bool)
        for k in range(3):
            ok &= (g[k] > 0) & (g[k] < dims[k] - 1)
        a = []
        for k in range(3):
            g[k] = np.clip(g[k], 1, dims[k] - 2)
            g[k] = np.where(p[k] < (g[k].astype(f32) + f32(0.5)) * self.cell[k], g[k] - 1, g[k])
            a.append(((p[k] - (g[k].astype(f32) + f32(0.5)) * self.cell[k]) / self.cell[k]).astype(f32))
        A, B, C = a
        x, y, z = g
        one = f32(1)
        res = self.tsdf(x, y, z) * (one - A) * (one - B) * (one - C)
        res = res + self.tsdf(x, y, z + 1) * (one - A) * (one - B) * C
        res = res + self.tsdf(x, y + 1, z) * (one - A) * B * (one - C)
        res = res + self.tsdf(x, y + 1, z + 1) * (one - A) * B * C
        res = res + self.tsdf(x + 1, y, z) * A * (one - B) * (one - C)
        res = res + self.tsdf(x + 1, y, z + 1) * A * (one - B) * C
        res = res + self.tsdf(x + 1, y + 1, z) * A * B * (one - C)
        res = res + self.tsdf(x + 1, y + 1, z + 1) * A * B * C
        return np.where(ok, res, f32(np.nan)).astype(f32)


def raycast(vol, size, trunc, W, H, fx, fy, cx, cy, pose, Z=None, zs0=0, zo0=0, zo1=None):
    """A.6 with D3 and the slab form of DESIGN.md section 6: a march step is taken by the slab that owns the z plane
    of its far sample; keys = (step << 1) | (0 surface hit, 1 abort).  Returns vmap, nmap, keys, steps that read voxels"""
    nz, Y, X, _ = vol.shape
    Z = Z or nz
    zo1 = Z if zo1 is None else zo1
    G = _Grid(vol, size, Z, zs0)
    cell = G.cell
    tau = tau_of(size, (X, Y, Z), trunc)
    R, t = pose[:3, :3].astype(f32), pose[:3, 3].astype(f32)
    sz = [f32(s) for s in size]
    u = np.arange(W, dtype=f32)[None, :] + np.zeros((H, 1), f32)
    v = np.arange(H, dtype=f32)[:, None] + np.zeros((1, W), f32)
    rn = [(u - f32(cx)) / f32(fx), (v - f32(cy)) / f32(fy), np.ones((H, W), f32)]
    d = _rot(R, rn)
    with np.errstate(all="ignore"):
        inv = f32(1) / np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
        d = [(d[k] * inv).astype(f32) for k in range(3)]
        d = [np.where(d[k] == 0, f32(1e-15), d[k]).astype(f32) for k in range(3)]
        tmin = [((np.where(d[k] > 0, f32(0), sz[k]) - t[k]) / d[k]).astype(f32) for k in range(3)]
        tmax = [((np.where(d[k] > 0, sz[k], f32(0)) - t[k]) / d[k]).astype(f32) for k in range(3)]
    t_start = np.maximum(np.maximum(np.maximum(tmin[0], tmin[1]), tmin[2]), f32(0))
    t_exit = np.minimum(np.minimum(tmax[0], tmax[1]), tmax[2])
    step_len = tau * f32(0.8)
    max_time = f32(3) * ((sz[0] + sz[1]) + sz[2])
    vm = np.full((3, H, W), np.nan, f32)
    nm = np.full((3, H, W), np.nan, f32)
    keys = np.full((H, W), KEY_NONE, np.int32)
    alive = t_start < t_exit
    tc = t_start.astype(f32).copy()
    step = np.zeros((H, W), np.int64)
    n_steps = 0
    dims = (X, Y, Z)

    def point(tt, sel=None):
        if sel is None:
            return [(t[k] + d[k] * tt).astype(f32) for k in range(3)]
        return [(t[k] + d[k][sel] * tt).astype(f32) for k in range(3)]

    while True:
        alive &= tc < max_time
        if not alive.any():
            break
        tn = (tc + step_len).astype(f32)
        pn = point(tn)
        g = [_vox(pn[k], cell[k]) for k in range(3)]
        inside = np.ones((H, W), bool)
        for k in range(3):
            inside &= (g[k] >= 0) & (g[k] < dims[k])
        alive &= inside                                   # a far sample outside the grid ends the ray
        owned = alive & (g[2] >= zo0) & (g[2] < zo1)
        if owned.any():
            sel = np.nonzero(owned)
            pc = point(tc[sel], sel)
            pv = [np.clip(_vox(pc[k], cell[k]), 0, dims[k] - 1) for k in range(3)]
            raw_prev = G.at(pv[0], pv[1], pv[2])
            raw_far = G.at(g[0][sel], g[1][sel], g[2][sel])
            n_steps += len(sel[0])
            back = (raw_prev < 0) & (raw_far > 0)
            cross = (raw_prev > 0) & (raw_far < 0)
            stp = step[sel]
            key = np.where(back | cross, (stp << 1) | 1, KEY_NONE).astype(np.int64)
            if cross.any():
                c = np.nonzero(cross)[0]
                csel = (sel[0][c], sel[1][c])
                tcc = tc[csel]
                pnc = point((tcc + step_len).astype(f32), csel)
                pcc = point(tcc, csel)
                Ftdt = G.trilinear(pnc)
                Ft = G.trilinear(pcc)
                with np.errstate(all="ignore"):
                    Ts = (tcc - (step_len * Ft) / (Ftdt - Ft)).astype(f32)
                    good = ~np.isnan(Ftdt) & ~np.isnan(Ft) & (Ts >= tcc - step_len) & (Ts <= tcc + f32(2.0) * step_len)
                if good.any():
                    gi = np.nonzero(good)[0]
                    gsel = (csel[0][gi], csel[1][gi])
                    vtx = point(Ts[gi], gsel)
                    for k in range(3):
                        vm[k][gsel] = vtx[k]
                    key[c[gi]] = stp[c[gi]] << 1
                    q = [_vox(pcc[k][gi], cell[k]) for k in range(3)]
                    deep = np.ones(len(gi), bool)
                    for k in range(3):
                        deep &= (q[k] > 1) & (q[k] < dims[k] - 2)
                    if deep.any():
                        di = np.nonzero(deep)[0]
                        dsel = (gsel[0][di], gsel[1][di])
                        base = [vtx[k][di] for k in range(3)]
                        n = []
                        for k in range(3):
                            hi = [b.copy() for b in base]
                            lo = [b.copy() for b in base]
                            hi[k] = (hi[k] + cell[k]).astype(f32)
                            lo[k] = (lo[k] - cell[k]).astype(f32)
                            n.append((G.trilinear(hi) - G.trilinear(lo)).astype(f32))
                        with np.errstate(all="ignore"):
                            ninv = f32(1) / np.sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2])
                        for k in range(3):
                            nm[k][dsel] = n[k] * ninv
            ended = back | cross
            keys[sel[0][ended], sel[1][ended]] = key[ended].astype(np.int32)
            alive[sel[0][ended], sel[1][ended]] = False
        tc = np.where(alive, tn, tc).astype(f32)
        step = np.where(alive, step + 1, step)
    return vm, nm, keys, n_steps


def extract_cloud(vol, size):
    """A.7: zero crossings between a voxel and its +x, +y, +z neighbour (both observed, neither at +1), linear
    interpolation on |F|, points in voxel order then axis order"""
    Z, Y, X, _ = vol.shape
    cell = [f32(size[0]) / f32(X), f32(size[1]) / f32(Y), f32(size[2]) / f32(Z)]
    fr, w = vol[..., 0].astype(np.int64), vol[..., 1].astype(np.int64)
    usable = (w != 0) & (fr != 32767)
    F = fr.astype(f32) / f32(32767)
    z, y, x = np.mgrid[0:Z, 0:Y, 0:X]
    ctr = [(x.astype(f32) + f32(0.5)) * cell[0], (y.astype(f32) + f32(0.5)) * cell[1], (z.astype(f32) + f32(0.5)) * cell[2]]
    lin = (z * Y + y) * X + x
    pts, order = [], []
    for k, ax in enumerate((2, 1, 0)):       # k = 0: +x neighbour (array axis 2), 1: +y, 2: +z
        a = [slice(None)] * 3
        b = [slice(None)] * 3
        a[ax], b[ax] = slice(0, -1), slice(1, None)
        a, b = tuple(a), tuple(b)
        ok = usable[a] & usable[b] & (((fr[a] > 0) & (fr[b] < 0)) | ((fr[a] < 0) & (fr[b] > 0)))
        Fa, Fb = np.abs(F[a][ok]), np.abs(F[b][ok])
        Vk = ctr[k][a][ok]
        Vn = (Vk + cell[k]).astype(f32)
        with np.errstate(all="ignore"):
            pk = ((Vk * Fb + Vn * Fa) * (f32(1) / (Fa + Fb))).astype(f32)
        p = np.stack([ctr[0][a][ok], ctr[1][a][ok], ctr[2][a][ok]], axis=1)
        p[:, k] = pk
        pts.append(p)
        order.append(lin[a][ok] * 3 + k)
    pts = np.concatenate(pts)
    return pts[np.argsort(np.concatenate(order), kind="stable")].astype(f32)


class Tracker:
    """A.2 state machine over the functions above (small volumes only: numpy speed)"""

    def __init__(self, n, W, H, fx, fy, cx, cy, size=(3.0, 3.0, 3.0), trunc=0.03, iters=(10, 5, 4), dist_thresh=0.10,
                 angle_thresh=0.3420201433256687, move_thresh=0.0, acosf=None):
        self.n, self.W, self.H, self.intr = n, W, H, (f32(fx), f32(fy), f32(cx), f32(cy))
        self.size, self.trunc, self.iters = size, trunc, iters
        self.dth, self.ath, self.move, self.acosf = f32(dist_thresh), f32(angle_thresh), f32(move_thresh), acosf
        self.init = np.eye(4, dtype=f32)
        self.init[:3, 3] = [f32(size[0]) / f32(2), f32(size[1]) / f32(2), f32(size[2]) / f32(2) - f32(1.2) * f32(size[2]) / f32(2)]
        self.reset()

    def reset(self):
        self.vol = np.zeros((self.n, self.n, self.n, 2), np.int16)
        self.pose = self.init.copy()
        self.frame = 0

    def _level(self, l):
        s = f32(1 << l)
        return tuple(v / s for v in self.intr)

    def process(self, depth):
        fx, fy, cx, cy = self.intr
        lv = [bilateral(depth)]
        lv.append(pyrdown(lv[0]))
        lv.append(pyrdown(lv[1]))
        vcur = [vmap(lv[l], *self._level(l)) for l in range(3)]
        ncur = [nmap(v) for v in vcur]
        scaled = scale_depth(depth, fx, fy, cx, cy)
        if self.frame == 0:
            integrate_full(self.vol, self.size, self.trunc, scaled, fx, fy, cx, cy, self.pose)
            tm = [transform_maps(vcur[l], ncur[l], self.pose) for l in range(3)]
            self.vmod, self.nmod = [m[0] for m in tm], [m[1] for m in tm]
            self.frame = 1
            return self.pose.copy(), False
        est = self.pose.copy()
        for l in (2, 1, 0):
            for _ in range(self.iters[l]):
                sums, _ = icp_sums(vcur[l], ncur[l], self.vmod[l], self.nmod[l], *self._level(l), est, self.pose, self.dth, self.ath)
                x6, ok = icp_solve(sums)
                if not ok:
                    self.reset()
                    return self.pose.copy(), False
                est = pose_update(est, x6)
        do_integrate = gate_passes(est, self.pose, self.move, self.acosf)
        self.pose = est
        if do_integrate:
            integrate_full(self.vol, self.size, self.trunc, scaled, fx, fy, cx, cy, self.pose)
        v0, n0, _, _ = raycast(self.vol, self.size, self.trunc, self.W, self.H, fx, fy, cx, cy, self.pose)
        self.vmod, self.nmod = [v0], [n0]
        for l in (1, 2):
            self.vmod.append(resize_vmap(self.vmod[l - 1]))
            self.nmod.append(resize_nmap(self.nmod[l - 1]))
        self.frame += 1
        return self.pose.copy(), True
