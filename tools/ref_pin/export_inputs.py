#!/usr/bin/env python3
"""Writes the inputs of the committed golden fixtures (tests/golden/road_*.npz) and of the four full-size scene variants
as PNG files + cases.txt for ref_pin.cpp.  Needs numpy and zlib only (no OpenCV on this side).
    python tools/ref_pin/export_inputs.py [out dir = tools/ref_pin/inputs]"""
import glob
import os
import struct
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "cart-slam_amd"))


def write_png(path, img):
    """8-bit gray [h,w] or BGR [h,w,3] -> PNG (colour type 0 / 2; BGR is stored as RGB, which cv::imread hands back as BGR)."""
    h, w = img.shape[:2]
    rgb = img if img.ndim == 2 else img[:, :, ::-1]
    raw = b"".join(b"\x00" + np.ascontiguousarray(rgb[y]).tobytes() for y in range(h))
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0 if img.ndim == 2 else 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def export(out, full_size=True):
    """Writes <case>_left.png / <case>_right.png + cases.txt into `out`; -> the cases [(name, min_disp, D, P)].
    full_size=False: the golden fixtures only (what tests/test_ref_pin.py round-trips through the repo's PNG reader)."""
    from cartslam import synth
    os.makedirs(out, exist_ok=True)
    cases = []
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "road_*.npz"))):
        z = np.load(f)
        name = os.path.basename(f)[:-4]
        write_png(os.path.join(out, name + "_left.png"), z["left"]); write_png(os.path.join(out, name + "_right.png"), z["right"])
        cases.append((name, int(z["min_disp"]), int(z["D"]), int(z["P"])))
    for scene in (synth.SCENES if full_size else ()):   # the headline configuration on every scene variant, and the reference's defaults on the road
        l, r, _ = synth.make_pair(1242, 375, 128, 4, scene=scene)
        name = f"full_1242x375_d128_p8_{scene}"
        write_png(os.path.join(out, name + "_left.png"), l); write_png(os.path.join(out, name + "_right.png"), r)
        cases.append((name, 4, 128, 8))
    if full_size:
        l, r, _ = synth.make_pair(1242, 375, 256, 4)
        write_png(os.path.join(out, "full_1242x375_d256_p4_road_left.png"), l); write_png(os.path.join(out, "full_1242x375_d256_p4_road_right.png"), r)
        cases.append(("full_1242x375_d256_p4_road", 4, 256, 4))
    with open(os.path.join(out, "cases.txt"), "w") as f:
        for c in cases:
            f.write("%s %d %d %d\n" % c)
    return cases


if __name__ == "__main__":
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "inputs")
    print(f"{len(export(out_dir))} cases written to {out_dir}")
