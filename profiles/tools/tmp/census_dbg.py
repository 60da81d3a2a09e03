import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle_lib as O
from cartslam import Engine, synth
w, h, D = 160, 96, 64
l, r, _ = synth.make_pair(w, h, D, 4, seed=1000 + w + D)
eng = Engine(w, h, num_disparities=D, paths=4, min_disparity=4, max_inflight=4)
eng.compute_disparity(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda())
got = eng.debug_read(2); want = O.census(l)
bad = np.argwhere(got != want)
print(len(bad), "mismatches; first", bad[:10].tolist())
ys, xs = bad[:, 0], bad[:, 1]
print("x mod 64 histogram:", np.bincount(xs % 64, minlength=64).tolist())
print("y mod 16 histogram:", np.bincount(ys % 16, minlength=16).tolist())
y, x = bad[0]
print(hex(got[y, x]), hex(want[y, x]), bin(got[y, x] ^ want[y, x]))
