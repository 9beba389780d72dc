"""Full scripted trajectory (300 frames) on the GPU: worst-case pose error vs ground truth."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import housescan_amd as h
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
trk = h.KinfuTracker(n=n)
worst_t = worst_a = 0.0
lost = 0
for k in range(300):
    gt = h.synth_pose(k)
    p, ok = trk.process_frame(h.synth_depth(gt))
    lost += (k > 0 and not ok)
    dt = np.linalg.norm(p[:3, 3] - gt[:3, 3]) * 1000
    ang = np.degrees(np.arccos(np.clip((np.trace(p[:3, :3].astype(np.float64).T @ gt[:3, :3]) - 1) / 2, -1, 1)))
    worst_t, worst_a = max(worst_t, dt), max(worst_a, ang)
print(f"volume {n}^3: 300 frames, lost {lost}, worst translation error {worst_t:.2f} mm, worst rotation error {worst_a:.3f} deg, final {dt:.2f} mm / {ang:.3f} deg")
