"""Aggregate / WTA time of the headline configuration against the SIZE of the engine's workspace (max_inflight slots; a 16-frame call uses the
first 16) -- fresh engines in one process, several instances per size.  Product build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import numpy as np, torch
from cartslam import Engine, synth
w, h, D, P, B = 1242, 375, 128, 8, 16
ls, rs = synth.make_batch(4, w, h, D, 4)
L = torch.from_numpy(np.concatenate([ls] * 4)).cuda(); R = torch.from_numpy(np.concatenate([rs] * 4)).cuda()
for inflight in [int(v) for v in (sys.argv[1:] or ["16", "32", "16", "24", "32", "48", "16", "32"])]:
    res = []
    for k in range(4):
        eng = Engine(w, h, num_disparities=D, paths=P, smoothing_radius=2, smoothing_iterations=1, max_inflight=inflight)
        for _ in range(3):
            eng.compute_disparity(L, R)
        torch.cuda.synchronize(); eng.set_timing(True)
        for _ in range(20):
            eng.compute_disparity(L, R)
        torch.cuda.synchronize()
        st, n = eng.collect_timing()
        res.append("%.3f/%.3f" % (st["aggregate"], st["wta"]))
        eng.close()
    print("max_inflight %2d (%5.1f GB of slabs): aggregate/wta per instance  %s" % (inflight, inflight * P * w * h * D / 1e9, "  ".join(res)), flush=True)
