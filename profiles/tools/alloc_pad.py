"""Experiment build (build_variant.sh addr): aggregate / WTA time against the padding between two path slabs (CART_SLAB_PAD bytes added to
slab_bytes; frame stride = P * slab_bytes) and the workspace size; 4 fresh engines per point.  usage: alloc_pad.py [max_inflight ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import numpy as np, torch
from cartslam import Engine, synth
w, h, D, P, B = 1242, 375, 128, 8, 16
ls, rs = synth.make_batch(4, w, h, D, 4)
L = torch.from_numpy(np.concatenate([ls] * 4)).cuda(); R = torch.from_numpy(np.concatenate([rs] * 4)).cuda()
slab = w * h * D
for inflight in [int(v) for v in (sys.argv[1:] or ["16", "32"])]:
    for pad in (0, 256, 4096 - slab % 4096, 65536 - slab % 65536, (1 << 20) - slab % (1 << 20), (2 << 20) - slab % (2 << 20), (2 << 20) - slab % (2 << 20) + 4096,
                (2 << 20) - slab % (2 << 20) + 65536, (64 << 20) - slab % (64 << 20), (64 << 20) - slab % (64 << 20) + (1 << 20)):
        os.environ["CART_SLAB_PAD"] = str(pad)
        res = []
        for k in range(4):
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
        print("inflight %2d  pad %9d (slab stride %% 2 MiB = %8d, %% 4 KiB = %4d): %s" % (inflight, pad, (slab + ((pad + 255) & ~255)) % (2 << 20), (slab + ((pad + 255) & ~255)) % 4096, "  ".join(res)), flush=True)
