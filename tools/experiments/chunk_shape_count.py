"""Which wave footprint makes the most wave-chunks homogeneous (all free / all dead)?  CPU count, same approximations as
chunk_free_count.py.  usage: chunk_shape_count.py N frame"""
import sys
import numpy as np
sys.path.insert(0, ".")
import housescan_amd as hsk

n = int(sys.argv[1]); frame = int(sys.argv[2])
cfg = hsk.default_config(n)
W, H = cfg.width, cfg.height
fx, fy, cx, cy = cfg.fx, cfg.fy, cfg.cx, cfg.cy
cell = np.array(list(cfg.vol_size_m)) / n
tau = cfg.trunc_dist_m
T = 16
def tile_red(img, t, fn):
    th, tw = (H + t - 1) // t, (W + t - 1) // t
    pad = np.zeros((th * t, tw * t)); pad[:H, :W] = img
    return fn(pad.reshape(th, t, tw, t), axis=(1, 3))
pose = hsk.synth_pose(frame).astype(np.float64)
depth = hsk.synth_depth(hsk.synth_pose(frame)).astype(np.float64) / 1000.0
uu, vv = np.meshgrid(np.arange(W), np.arange(H))
scaled = depth * np.sqrt(((uu - cx) / fx) ** 2 + ((vv - cy) / fy) ** 2 + 1.0)
tmin, tmax = tile_red(scaled, T, np.min), tile_red(scaled, T, np.max)
# 2-D range min / max by brute force over tile windows: precompute a sparse table would be nicer; loops are fine here
R, t = pose[:3, :3], pose[:3, 3]
for (sx, sy, sz) in [(64, 4, 8), (32, 8, 8), (16, 16, 8), (16, 16, 16), (64, 4, 16), (32, 8, 16), (16, 8, 16), (8, 8, 8), (64, 16, 8)]:
    bx, by, bz = n // sx, n // sy, n // sz
    ix, iy, iz = np.meshgrid(np.arange(bx), np.arange(by), np.arange(bz), indexing="ij")
    lo = np.stack([(ix * sx + 0.5) * cell[0], (iy * sy + 0.5) * cell[1], (iz * sz + 0.5) * cell[2]], -1) - t
    hi = np.stack([(ix * sx + sx - 0.5) * cell[0], (iy * sy + sy - 0.5) * cell[1], (iz * sz + sz - 0.5) * cell[2]], -1) - t
    zs, us, vs = [], [], []
    for c in range(8):
        p = np.where(np.array([(c >> k) & 1 for k in range(3)], bool), hi, lo)
        cam = p @ R
        zs.append(cam[..., 2]); us.append(cam[..., 0] / cam[..., 2] * fx + cx); vs.append(cam[..., 1] / cam[..., 2] * fy + cy)
    zs = np.stack(zs); us = np.stack(us); vs = np.stack(vs)
    front = zs.min(0) > 0.05
    umin, umax, vmin, vmax = us.min(0) - 1, us.max(0) + 1, vs.min(0) - 1, vs.max(0) + 1
    inimg = front & (umin >= 0) & (vmin >= 0) & (umax <= W - 1) & (vmax <= H - 1)
    live = (zs.max(0) > 0) & ((~front) | ((umax >= -1.5) & (umin <= W + 0.5) & (vmax >= -1.5) & (vmin <= H + 0.5)))
    dmax = np.sqrt(np.maximum(lo ** 2, hi ** 2).sum(-1))
    g0 = np.where((lo <= 0) & (hi >= 0), 0.0, np.minimum(np.abs(lo), np.abs(hi)))
    dmin = np.sqrt((g0 ** 2).sum(-1))
    cand = front & live
    idc = np.argwhere(cand)
    cu0 = np.clip(umin[cand], 0, W - 1).astype(int) // T; cu1 = np.clip(umax[cand], 0, W - 1).astype(int) // T
    cv0 = np.clip(vmin[cand], 0, H - 1).astype(int) // T; cv1 = np.clip(vmax[cand], 0, H - 1).astype(int) // T
    ii = inimg[cand]; dx = dmax[cand]; dn = dmin[cand]
    fr = np.zeros(len(idc), bool); dd = np.zeros(len(idc), bool)
    for q in range(len(idc)):
        wmin = tmin[cv0[q]:cv1[q] + 1, cu0[q]:cu1[q] + 1].min(); wmax = tmax[cv0[q]:cv1[q] + 1, cu0[q]:cu1[q] + 1].max()
        fr[q] = ii[q] and dx[q] + tau * 1.0002 + 1e-4 <= wmin
        dd[q] = dn[q] - wmax > tau * 1.001 + 1e-4
    nl = live.sum(); lb = sx * sy * sz // 16
    print(f"{sx:3d}x{sy:2d}x{sz:2d}: chunks {live.size:7d} live~{nl:6d} free {fr.sum():6d} ({fr.sum()/nl:.2f}) dead {dd.sum():6d} ({dd.sum()/nl:.2f}) "
          f"mixed {nl - fr.sum() - dd.sum():6d} ({1 - (fr.sum() + dd.sum())/nl:.2f}); lane-blocks: free {fr.sum()*lb/1e6:.2f}M dead {dd.sum()*lb/1e6:.2f}M mixed {(nl - fr.sum() - dd.sum())*lb/1e6:.2f}M", flush=True)
