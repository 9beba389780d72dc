"""The sensor-holes renderer (hsk_synth_render_sensor, synth.cpp): host-only, runs without a GPU."""
import numpy as np


def test_sensor_frame_is_the_clean_render_minus_contiguous_holes(hsk):
    pose = hsk.synth_pose(17)
    clean = hsk.synth_depth(pose)
    d, frac = hsk.synth_sensor_depth(pose, sigma_mm=0.0)
    m = d != 0
    assert np.array_equal(d[m], clean[m])                   # no noise: valid pixels are the exact render
    assert abs(frac - (~m).mean()) < 1e-12 and 0.01 < frac < 0.10
    # the holes are contiguous: most invalid pixels have an invalid 4-neighbour (an independent 3 % dropout would have ~12 %)
    inv = ~m
    nb = np.zeros_like(inv)
    nb[1:] |= inv[:-1]
    nb[:-1] |= inv[1:]
    nb[:, 1:] |= inv[:, :-1]
    nb[:, :-1] |= inv[:, 1:]
    assert (inv & nb).sum() > 0.9 * inv.sum()
    # shadow bands: 3 to 5 px wide runs next to horizontal depth steps
    d2, frac2 = hsk.synth_sensor_depth(pose, sigma_mm=0.0, absorbing=True)
    assert frac2 > frac + 0.02 and np.all(d2[d2 != 0] == clean[d2 != 0])
    d3, frac3 = hsk.synth_sensor_depth(pose, sigma_mm=0.0, range_cut_m=2.0)
    assert d3.max() <= 2000 and frac3 > frac


def test_sensor_noise_is_a_pure_function_of_its_arguments(hsk):
    pose = hsk.synth_pose(3)
    a, _ = hsk.synth_sensor_depth(pose, seed=7)
    b, _ = hsk.synth_sensor_depth(pose, seed=7)
    c, _ = hsk.synth_sensor_depth(pose, seed=8)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    clean = hsk.synth_depth(pose).astype(np.float64)
    m = a != 0
    z = clean[m] / 1000.0
    resid = (a[m].astype(np.float64) - clean[m]) / (1.2 * z * z)
    assert abs(resid.mean()) < 0.02 and 0.9 < resid.std() < 1.15   # sigma = 1.2 mm z^2 (+ the rounding to whole millimetres)
    # the closed rooms too
    r, frac = hsk.synth_sensor_depth(hsk.synth_room_pose(0, 100, 720), scene=0)
    assert 0.005 < frac < 0.3 and r.max() <= 3500 + 60
