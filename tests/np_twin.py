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
        front = cam[2] > 0
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
            win = s[max(2 * y - 2, 0):min(2 * y + 2, H - 1) + 1, max(2 * x - 2, 0):min(2 * x + 2, W - 1) + 1]
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
