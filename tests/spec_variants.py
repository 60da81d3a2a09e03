"""How many output pixels the two open upstream questions of the SGM post stages change (oracle S7 / S8, see
oracle/cart_oracle.h): counts on the committed golden scenes and on the 1242x375 bench scene.  Test infrastructure (it
imports the oracle); run from the repo root:  python tests/spec_variants.py

  S8  the oracle invalidates a left pixel when gray == 0, when it is already 0xFFFF, or on a left-right mismatch.  The
      opencv_contrib port of libSGM's check_consistency_kernel may ALSO test `d <= 0` on the integer disparity (older libSGM
      did; later releases test `org == INVALID_DISP`, the oracle's form): every valid winner at disparity index 0 then
      becomes invalid.
  S7  the oracle's 3x3 medians pass the one-pixel image border through; the alternative is a median over the replicated
      border.  It can only change border pixels and, through the right map, the pixels whose LR partner sits on the border.
"""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "cart-slam_amd")]

import oracle_lib as O  # noqa: E402
from cartslam import synth  # noqa: E402


def median_replicated(a):
    p = np.pad(a, 1, mode="edge")
    h, w = a.shape
    win = np.stack([p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)])
    return np.sort(win, axis=0)[4].astype(np.uint16)


def variants(gl, gr, D, P, md):
    _, S = O.sgm(gl, gr, D, P, md, want_S=True)
    wl, wr = O.wta(S, 12)
    ml, mr = O.median3x3(wl), O.median3x3(wr)
    base = O.lr_check_range(ml, mr, gl, md)
    invalid = (md - 1) * 16
    # S8 variant: valid pixels whose integer disparity (before the min_disp offset) is 0 become invalid too
    s8 = int(((base != invalid) & ((ml >> 4) == 0)).sum())
    # S7 variant: replicated-border medians on both maps
    alt = O.lr_check_range(median_replicated(wl), median_replicated(wr), gl, md)
    s7 = int((alt != base).sum())
    return s8, s7, int((base != invalid).sum()), base.size


def main():
    rows = []
    for f in sorted(glob.glob(os.path.join(HERE, "golden", "road_*.npz"))):
        z = np.load(f)
        l, r = z["left"], z["right"]
        if l.ndim == 3:
            l, r = O.bgr2gray(l), O.bgr2gray(r)
        rows.append((os.path.basename(f),) + variants(l, r, int(z["D"]), int(z["P"]), int(z["min_disp"])))
    l, r, _ = synth.make_pair(1242, 375, 128, 4)
    rows.append(("synth 1242x375 D=128 P=8 (bench frame 0)",) + variants(l, r, 128, 8, 4))
    for scene in ("stripes", "saturated", "pole", "wall", "photometric"):
        l, r, _ = synth.make_pair(1242, 375, 128, 4, scene=scene)
        rows.append((f"synth 1242x375 D=128 P=8 scene={scene}",) + variants(l, r, 128, 8, 4))
    print(f"{'scene':52s} {'S8: d<=0 also invalid':>22s} {'S7: replicated border':>22s} {'valid':>9s} {'pixels':>9s}")
    for name, s8, s7, valid, n in rows:
        print(f"{name:52s} {s8:22d} {s7:22d} {valid:9d} {n:9d}")


if __name__ == "__main__":
    main()
