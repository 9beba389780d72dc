"""Maps for the ICP accumulation's edge cases (identity poses, so the rows are what the maps say): products whose scaled
value is an exact tie (k + 1/2), a few large rows (entries of 600 - 1000: scaled products of 2^45), ordinary pixels in
between -- with the 27 sums below 2^27, where the specification's snapped sums are exact.  Used by the oracle-against-twin
test on the CPU and the GPU parity test."""
import numpy as np


def extreme_maps(W, H, fx, fy, cx, cy, seed=5):
    rng = np.random.default_rng(seed)
    f32 = np.float32
    vcur = np.full((3, H, W), np.nan, f32)
    ncur = np.full((3, H, W), np.nan, f32)
    vmod = np.full((3, H, W), np.nan, f32)
    nmod = np.full((3, H, W), np.nan, f32)
    # mostly ordinary pixels and ties; a handful of large rows (the 27 sums must stay below 2^27 to be exact at all)
    kinds = np.where(rng.random((H, W)) < 0.5, 5, 9)
    flat = rng.permutation(H * W)
    for kind, count in ((1, 4), (3, 3), (0, 500)):
        idx, flat = flat[:count], flat[count:]
        kinds.reshape(-1)[idx] = kind
    for y in range(2, H - 2):
        for x in range(2, W - 2):
            kind = int(kinds[y, x])
            # depth of the current point: ordinary, or far (|s x n| up to 600)
            z = f32({0: 1.5, 1: 1000.0}.get(kind, 1.0 + 2.0 * rng.random()))
            v = np.array([(x - cx) / fx * z, (y - cy) / fy * z, z], f32)
            n = np.array([0.0, 0.0, -1.0], f32)
            s = f32(1.0)
            if kind == 3:
                s = f32(1000.0)          # a "normal" a thousand long (the angle gate looks at its direction only)
            if kind == 5:
                # ties: entries that are odd multiples of 2^-13 and 2^-14 -- their product times 2^26 is an odd number of halves
                n = np.array([0.0, 0.0, -(2 * rng.integers(1, 4000) + 1) * 2.0 ** -13], f32)
                v = np.array([(x - cx) / fx * z, (y - cy) / fy * z, z], f32)
            vcur[:, y, x] = v
            ncur[:, y, x] = n if kind != 5 else np.array([0, 0, -1], f32)
            nmod[:, y, x] = n * s
            d = np.array([(2 * rng.integers(0, 40) + 1) * 2.0 ** -14, 0.0, (2 * rng.integers(0, 40) + 1) * 2.0 ** -14], f32)
            vmod[:, y, x] = v + d  # (within the distance gate; far points round the offset away, which is fine)
    return vcur, ncur, vmod, nmod
