#!/usr/bin/env python3
"""Replays the first <count> frames of a synthetic stream through one tracker (one integrate per frame, frame 0
included) and exits: the child process bench.py runs under `rocprofv3 --pmc` / `--kernel-trace` to read the kernels'
counters and durations over exactly the frames of its timed region.
usage: replay_frames.py <volume> <count> [stream]     stream: scripted (default) | noise | holes | room<V>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import housescan_amd as hsk  # noqa: E402


def stream_frames(hsk, stream, count):
    """(ground-truth poses, frames, init_pose or None) of the named synthetic stream -- the one place that maps a stream's
    name to its frames (bench.py's blocks and their counter passes must see the same frames)"""
    if stream in (None, "", "scripted"):
        gts = [hsk.synth_pose(k) for k in range(count)]
        return gts, [hsk.synth_depth(p) for p in gts], None
    if stream == "noise":      # SURVEY.md 8(d)'s noise run: sigma = 1.2 mm z^2, 2 % independent dropout
        gts, fr = hsk.synth_noisy_frames(count)
        return gts, fr, None
    if stream == "holes":      # holes as a sensor makes them: grazing rays, shadow bands, 3.5 m cut, absorbing furniture, sigma
        gts, fr = hsk.synth_sensor_frames(count, absorbing=True)
        return gts, fr, None
    if stream.startswith("room"):   # the room scan: camera inside the volume, level turn of the 720-frame scan of room V
        v = int(stream[4:] or 0)
        gts = [hsk.synth_room_pose(v, k, 720) for k in range(count)]
        return gts, [hsk.synth_room_depth(v, p) for p in gts], gts[0]
    raise SystemExit("unknown stream " + stream)


if __name__ == "__main__":
    n, count = int(sys.argv[1]), int(sys.argv[2])
    stream = sys.argv[3] if len(sys.argv) > 3 else "scripted"
    gts, frames, init = stream_frames(hsk, stream, count)
    trk = hsk.KinfuTracker(n=n) if init is None else hsk.KinfuTracker(n=n, init_pose=init)
    for d in frames:
        trk.process_frame(d)
    trk.close()
