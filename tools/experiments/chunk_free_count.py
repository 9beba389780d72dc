"""Go / no-go for the coarse free-space level of integrate's pass A, on the CPU: over frames of the
bench's scripted stream, the fraction of LIVE wave-chunks (64 x 4 voxels x zchunk planes: what one pass-A wave owns) and
workgroup-chunks (64 x 16 x zchunk) whose every voxel is observed as free space (F == 1) by a conservative box test against a
16-px tile minimum of the scaled depth.  Approximations (a count, not a parity check): raw depth instead of the bilateral
filter's, float64 geometry."""
import sys
import numpy as np
sys.path.insert(0, ".")
import housescan_amd as hsk

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
zchunk = 16 if n >= 1024 else 8
cfg = hsk.default_config(n)
W, H = cfg.width, cfg.height
fx, fy, cx, cy = cfg.fx, cfg.fy, cfg.cx, cfg.cy
size = np.array(list(cfg.vol_size_m)); cell = size / n
tau = cfg.trunc_dist_m if hasattr(cfg, "trunc_dist_m") else 0.03
print("fields", [f[0] for f in cfg._fields_])
print("n", n, "cell", cell, "tau", tau, "W,H", W, H)

def tile_min(scaled, t):
    th, tw = (H + t - 1) // t, (W + t - 1) // t
    pad = np.zeros((th * t, tw * t)); pad[:H, :W] = scaled
    return pad.reshape(th, t, tw, t).min(axis=(1, 3))

for frame in [int(a) for a in sys.argv[2:]] or [10, 25]:
    pose = hsk.synth_pose(frame).astype(np.float64)
    depth = hsk.synth_depth(hsk.synth_pose(frame)).astype(np.float64) / 1000.0
    uu, vv = np.meshgrid(np.arange(W), np.arange(H))
    lam = np.sqrt(((uu - cx) / fx) ** 2 + ((vv - cy) / fy) ** 2 + 1.0)
    scaled = depth * lam
    R, t = pose[:3, :3], pose[:3, 3]
    for name, ny in (("wave-chunk", 4), ("wg-chunk", 16)):
        for T in (16, 8):
            tmin = tile_min(scaled, T)
            bx, by, bz = n // 64, n // ny, n // zchunk
            ix, iy, iz = np.meshgrid(np.arange(bx), np.arange(by), np.arange(bz), indexing="ij")
            lo = np.stack([(ix * 64 + 0.5) * cell[0], (iy * ny + 0.5) * cell[1], (iz * zchunk + 0.5) * cell[2]], -1)
            hi = np.stack([(ix * 64 + 63.5) * cell[0], (iy * ny + ny - 0.5) * cell[1], (iz * zchunk + zchunk - 0.5) * cell[2]], -1)
            us, vs, zs = [], [], []
            dmax2 = 0
            for c in range(8):
                p = np.where(np.array([(c >> k) & 1 for k in range(3)], bool), hi, lo) - t
                cam = p @ R  # R^T p
                zs.append(cam[..., 2]); us.append(cam[..., 0] / cam[..., 2] * fx + cx); vs.append(cam[..., 1] / cam[..., 2] * fy + cy)
            g_lo, g_hi = lo - t, hi - t
            dmax = np.sqrt((np.maximum(g_lo ** 2, g_hi ** 2)).sum(-1))
            zs = np.stack(zs); us = np.stack(us); vs = np.stack(vs)
            zmin = zs.min(0)
            front = zmin > 0.05
            umin, umax, vmin, vmax = us.min(0) - 1, us.max(0) + 1, vs.min(0) - 1, vs.max(0) + 1
            inimg = front & (umin >= 0) & (vmin >= 0) & (umax <= W - 1) & (vmax <= H - 1)
            # live: some corner-ish overlap with the padded frustum (approximation: box of projected corners meets the image, or behind-crossing)
            live = (zs.max(0) > 0) & ((~front) | ((umax >= -1.5) & (umin <= W + 0.5) & (vmax >= -1.5) & (vmin <= H + 0.5)))
            free = np.zeros_like(inimg)
            idx = np.argwhere(inimg)
            tu0 = (umin[inimg].astype(int)) // T; tu1 = (umax[inimg].astype(int)) // T
            tv0 = (vmin[inimg].astype(int)) // T; tv1 = (vmax[inimg].astype(int)) // T
            nt = []
            f = np.zeros(len(idx), bool)
            for q in range(len(idx)):
                m = tmin[tv0[q]:tv1[q] + 1, tu0[q]:tu1[q] + 1].min()
                nt.append((tv1[q] - tv0[q] + 1) * (tu1[q] - tu0[q] + 1))
                f[q] = dmax[tuple(idx[q])] + tau * 1.0002 + 1e-4 <= m
            free[inimg] = f
            # all-dead: every voxel lies farther than the largest depth of its (clamped) pixel box + tau, or has no pixel
            tmax = -tile_min(-scaled, T)
            g0 = np.where((g_lo <= 0) & (g_hi >= 0), 0.0, np.minimum(np.abs(g_lo), np.abs(g_hi)))
            dmin = np.sqrt((g0 ** 2).sum(-1))
            dead = np.zeros_like(inimg)
            cand = front & live
            idc = np.argwhere(cand)
            cu0 = np.clip(umin[cand], 0, W - 1).astype(int) // T; cu1 = np.clip(umax[cand], 0, W - 1).astype(int) // T
            cv0 = np.clip(vmin[cand], 0, H - 1).astype(int) // T; cv1 = np.clip(vmax[cand], 0, H - 1).astype(int) // T
            d = np.zeros(len(idc), bool)
            for q in range(len(idc)):
                m = tmax[cv0[q]:cv1[q] + 1, cu0[q]:cu1[q] + 1].max()
                d[q] = dmin[tuple(idc[q])] - m > tau * 1.001 + 1e-4
            dead[cand] = d
            print(f"   all-dead {dead.sum()} ({dead.sum() / max(live.sum(), 1):.2f} of live)  mixed {live.sum() - dead.sum() - (free & live).sum()}")
            print(f"frame {frame} {name} T={T}: chunks {free.size} live~{live.sum()} in-image {inimg.sum()} all-free {free.sum()} "
                  f"({free.sum() / max(live.sum(), 1):.2f} of live) tiles/query mean {np.mean(nt):.1f} max {np.max(nt)}")
