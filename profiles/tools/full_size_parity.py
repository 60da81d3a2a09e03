"""One-off full oracle comparisons at BASELINE.json's large sizes (too slow for the test suite): 1920x1080 D=256 8 paths
(batch of 4 -> the fused WTA path, and a single pair -> the two-kernel path) and 1242x375 D=256 4 paths (the
reference's default configuration), disparity module output bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle_lib as O
from cartslam import Engine, synth
bad = 0
for w, h, D, P, nb in ((1920, 1080, 256, 8, 4), (1242, 375, 256, 4, 16), (1242, 375, 128, 8, 16)):
    ls, rs = synth.make_batch(2, w, h, D, 4)
    t = time.time(); want = [O.disparity_module(ls[k], rs[k], D, P, 4, radius=2, iterations=1) for k in range(2)]; t_or = time.time() - t
    eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=nb)
    L = torch.from_numpy(np.concatenate([ls] * (nb // 2))).cuda(); R = torch.from_numpy(np.concatenate([rs] * (nb // 2))).cuda()
    got_b = eng.compute_disparity(L, R).cpu().numpy()           # batch (fused WTA where it is the default)
    got_1 = eng.compute_disparity(L[1], R[1]).cpu().numpy()     # single pair (two-kernel WTA)
    ok = all(np.array_equal(got_b[k], want[k % 2]) for k in range(nb)) and np.array_equal(got_1, want[1])
    bad += not ok
    print(f"{w}x{h} D={D} P={P}: batch of {nb} + single pair vs oracle: {'bit-exact' if ok else 'MISMATCH'} (oracle {t_or / 2:.1f} s per pair)")
    eng.close()
sys.exit(1 if bad else 0)
