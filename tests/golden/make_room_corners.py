#!/usr/bin/env python3
"""Extracts the corner-table INPUTS the reference holds for its CPU stitching chain into a data fixture.

The reference records no expected outputs for them; these are inputs only:
  - devSetup's rooms: corners relative to each room's cloud mean (housescan/Main.hs:2346-2413)
  - loadTestRoom1WithCorners: 8 absolute corners (housescan/Main.hs:2531-2540)
  - FitCuboidBFGS's example cuboid: a 2x1x1 box rotated 20 degrees about (1,2,3) (FitCuboidBFGS.hs:29-41)
Runs only where /root/reference exists; the JSON it writes is what the tests read.
"""
import json
import os
import re

REF = "/root/reference/housescan/Main.hs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "room_corners.json")
NUM = r"\(?(-?[0-9.]+(?:e-?[0-9]+)?)\)?"
VEC = re.compile(r"Vec3\s+" + NUM + r"\s+" + NUM + r"\s+" + NUM)


def vecs(lines):
    out = []
    for ln in lines:
        m = VEC.search(ln)
        if m:
            out.append([float(m.group(1)), float(m.group(2)), float(m.group(3))])
    return out


def main():
    src = open(REF).read().split("\n")
    rooms = {}
    name = None
    for ln in src[2345:2414]:
        m = re.search(r'\("([a-z0-9]+)"', ln)
        if m:
            name = m.group(1)
            rooms[name] = []
        elif name and VEC.search(ln):
            rooms[name] += vecs([ln])
    fixture = {
        "source": {"dev_rooms_mean_relative": "housescan/Main.hs:2346-2413", "test_room1_absolute": "housescan/Main.hs:2531-2540",
                   "example_cuboid": "housescan/FitCuboidBFGS.hs:29-41 (2x1x1 box, rotated 20 deg about (1,2,3))"},
        "dev_rooms_mean_relative": rooms,
        "test_room1_absolute": vecs(src[2530:2541]),
        "example_cuboid": {"unrotated": [[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [2, 0, 0], [2, 0, 1], [2, 1, 0], [2, 1, 1]],
                           "axis": [1, 2, 3], "degrees": 20},
    }
    with open(OUT, "w") as f:
        json.dump(fixture, f, indent=1)
    print({k: len(v) for k, v in rooms.items()}, len(fixture["test_room1_absolute"]))


if __name__ == "__main__":
    main()
