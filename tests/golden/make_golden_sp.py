"""Generates the committed superpixel golden fixtures (sp_*.npz) with the C oracle (oracle/cart_oracle_sp.c).

PARITY UNPINNED (see make_golden.py): the reference ships no vectors for this stage; these pin the build's own
oracle and, through the GPU tests, the HIP path.  Run from the repo root:  python tests/golden/make_golden_sp.py
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
    # name, w, h, D, block, iterations per frame, parameter overrides
    ("sp_road_192x96_b8", 192, 96, 64, 8, (6, 2, 2), dict()),
    ("sp_road_230x70_b10_progressive", 230, 70, 64, 10, (5, 3), dict(compactness=0.03, progressive=1.0)),
]

for name, w, h, D, block, iters, kw in CASES:
    p = O.sp_params(**kw)
    labels, mx = O.sp_block_init(w, h, block, block)
    out = dict(block=block, frames=len(iters), p_direct=p.direct_clique_cost, p_diagonal=p.diagonal_clique_cost,
               p_compactness=p.compactness_weight, p_progressive=p.progressive_compactness_cost, p_image=p.image_weight,
               p_disparity=p.disparity_weight, max_label=mx)
    cum = np.zeros(256, np.int32)
    params = None
    for k, it in enumerate(iters):
        l, r, _ = synth.make_pair(w, h, D, 4, seed=500 + w, frame=k, channels=3)
        d = O.disparity_module(l, r, D, 4, 4, radius=2, iterations=1)
        d2, hist = O.directional_derivative(d)
        if params is None:
            ok, pp = O.histogram_peak_params(hist[:, 0].copy())
            params = pp if ok else (6, 18, -5, 6, 11, 0)
        labels, changes = O.sp_relax(p, labels, mx, O.bgr2ycrcb(l), d2, it)
        uns, pl = O.sp_classify(d2, labels, mx, params)
        out.update({f"image{k}": l, f"deriv{k}": d2, f"iters{k}": it, f"labels{k}": labels, f"unsmoothed{k}": uns, f"planes{k}": pl})
        print(name, k, "changes", changes, "planes", np.bincount(pl.ravel(), minlength=3))
    out["plane_params"] = np.array(params, np.int32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
