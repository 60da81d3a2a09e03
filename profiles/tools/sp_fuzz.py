"""Randomised differential campaign of the superpixel stage (cart_superpixels_relax through the Python layer) against the CPU oracle: random sizes, block
sizes 2..16 (small blocks push tiles past their LDS label table), the six parameter sets of tests/test_gpu_superpixels.py plus random weights, short frame
sequences on the persistent label state (sweeps 0..6 per frame).  BUDGET_S seconds (default 150), SEED."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle_lib as O
import test_gpu_superpixels as T
budget, seed = float(os.environ.get("BUDGET_S", 150)), int(os.environ.get("SEED", 7))
rng = np.random.default_rng(seed)
t0 = time.time(); cases = frames = bad = 0
while time.time() - t0 < budget:
    w, h = int(rng.integers(20, 400)), int(rng.integers(16, 200))
    bs = int(rng.integers(2, 17))
    if ((w + bs - 1) // bs) * ((h + bs - 1) // bs) >= 16384:
        continue
    if rng.random() < 0.6:
        kw = dict(T.PARAM_SETS[int(rng.integers(0, len(T.PARAM_SETS)))])
    else:
        kw = dict(compactness=float(rng.choice([0.0, 0.02, 0.1, 0.5])), progressive=float(rng.choice([0.0, 1.0, 2.5])), image=float(rng.choice([0.0, 0.7, 1.5])),
                  disparity=float(rng.choice([0.0, 1.0, 2.0])), direct=float(rng.choice([0.25, 0.5, 1.0])))
    eng = T.geometry_engine(w, h)
    sp = T.make_sp(eng, bs, kw)
    want, mx = O.sp_block_init(w, h, bs, bs)
    p = O.sp_params(**kw)
    use_d2 = kw.get("disparity", 1.0) > 0
    for _ in range(int(rng.integers(1, 4))):
        iters = int(rng.integers(0, 7))
        bgr, d2 = T.random_scene(rng, w, h, coarse=int(rng.integers(3, 9)))
        got = T.labels_np(sp.relax(T.dev(torch, bgr), T.dev(torch, d2) if use_d2 else None, iters))
        want, _ = O.sp_relax(p, want, mx, O.bgr2ycrcb(bgr), d2 if use_d2 else None, iters)
        frames += 1
        nd = int((got != want).sum())
        if nd:
            bad += 1
            print(f"DIFF case {cases}: w={w} h={h} bs={bs} kw={kw} iters={iters}: {nd} labels differ", flush=True)
            break
    sp.close(); eng.close(); cases += 1
print(f"superpixel fuzz: seed {seed}, {cases} cases, {frames} frames, {bad} with differing labels, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
