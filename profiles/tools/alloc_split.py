"""(The CART_SLAB_SLOTS hook was removed with the experiment: kept as the record of the command behind profiles/r03_alloc.txt section 3.)
Experiment build (build_variant.sh addr): which allocation's size decides the aggregate's mode?  max_inflight 32 with the slab allocation cut to
16 slots (CART_SLAB_SLOTS=16), against plain 16 and plain 32; 6 fresh engines each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import numpy as np, torch
from cartslam import Engine, synth
w, h, D, P, B = 1242, 375, 128, 8, 16
ls, rs = synth.make_batch(4, w, h, D, 4)
L = torch.from_numpy(np.concatenate([ls] * 4)).cuda(); R = torch.from_numpy(np.concatenate([rs] * 4)).cuda()
for inflight, slots in [tuple(int(x) if x != 'n' else None for x in a.split(':')) for a in (sys.argv[1:] or ['16:n', '32:n', '32:16', '64:16', '32:24', '17:n', '20:n', '16:n', '32:16'])]:
    if slots is None: os.environ.pop("CART_SLAB_SLOTS", None)
    else: os.environ["CART_SLAB_SLOTS"] = str(slots)
    res = []
    for k in range(6):
        eng = Engine(w, h, num_disparities=D, paths=P, smoothing_radius=2, smoothing_iterations=1, max_inflight=inflight)
        for _ in range(3):
            eng.compute_disparity(L, R)
        torch.cuda.synchronize(); eng.set_timing(True)
        for _ in range(16):
            eng.compute_disparity(L, R)
        torch.cuda.synchronize()
        st, n = eng.collect_timing()
        res.append("%.3f/%.3f" % (st["aggregate"], st["wta"]))
        eng.close()
    print("max_inflight %2d, slab slots %s: %s" % (inflight, slots or inflight, "  ".join(res)), flush=True)
