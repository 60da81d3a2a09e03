"""Generates the committed golden fixtures with the C oracle (oracle/cart_oracle.c).

PARITY UNPINNED: the reference ships no golden vectors for this path and cannot be built here
(SURVEY.md 8c), so these vectors pin the build's own oracle (and, through the GPU tests, the HIP
path) against regressions; they are not outputs of the reference.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "cart-slam_amd")]

import oracle_lib as O  # noqa: E402
from cartslam import synth  # noqa: E402

CASES = [
    # name, w, h, D, P, min_disp, channels, radius, iterations, seed
    ("road_160x96_d64_p4_gray", 160, 96, 64, 4, 4, 1, 2, 1, 11),
    ("road_200x120_d128_p8_bgr", 200, 120, 128, 8, 4, 3, 2, 1, 12),
    ("road_173x67_d64_p8_nosmooth", 173, 67, 64, 8, 0, 1, -1, 5, 13),
    ("road_320x64_d256_p4_r3", 320, 64, 256, 4, 4, 1, 3, 2, 14),
]

for name, w, h, D, P, md, ch, radius, iters, seed in CASES:
    l, r, _ = synth.make_pair(w, h, D, md, seed=seed, channels=ch)
    d = O.disparity_module(l, r, D, P, md, radius=radius, iterations=iters)
    dd, hist = O.plane_derivative(d)
    dir_d, dir_h = O.directional_derivative(d)
    ok, pp = O.histogram_peak_params(hist)
    if not ok:
        pp = (6, 18, -5, 6, 11, 0)
    pl = O.classify(dd, pp)
    ids, n = O.ccl(pl)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), left=l, right=r, D=D, P=P, min_disp=md, radius=radius,
                        iterations=iters, disparity=d, plane_derivative=dd, plane_hist=hist, dir_derivative=dir_d,
                        dir_hist=dir_h, plane_params=np.array(pp, np.int32), plane_params_ok=ok, planes=pl,
                        ccl_ids=ids, ccl_n=n)
    print(name, "valid%.3f" % float((d != -32768).mean()), "planes", np.bincount(pl.ravel(), minlength=3), "ccl", n, "ok", ok)
