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
MiB = 1 << 20
def to_res(r):   # pad that makes the slab stride = r (mod 2 MiB)
    return (r - slab % (2 * MiB)) % (2 * MiB)
PADS = [0, to_res(897024), to_res(MiB), to_res(MiB // 2), to_res(3 * MiB // 2), to_res(MiB + 4096), to_res(0), to_res(MiB) + 2 * MiB, to_res(MiB) + 8 * MiB]
N_INST = int(os.environ.get("N_INST", 8))
for inflight in [int(v) for v in (sys.argv[1:] or ["32"])]:
    for pad in PADS:
        os.environ["CART_SLAB_PAD"] = str(pad)
        res = []
        for k in range(N_INST):
            eng = Engine(w, h, num_disparities=D, paths=P, smoothing_radius=2, smoothing_iterations=1, max_inflight=inflight)
            for _ in range(3):
                eng.compute_disparity(L, R)
            torch.cuda.synchronize(); eng.set_timing(True)
            for _ in range(12):
                eng.compute_disparity(L, R)
            torch.cuda.synchronize()
            st, n = eng.collect_timing()
            res.append((st["aggregate"], st["wta"]))
            eng.close()
        stride = slab + ((pad + 255) & ~255)
        fast = sum(a < 1.47 for a, b in res)
        print("inflight %2d  pad %9d (slab stride %% 2 MiB = %8d; frame stride %% 64 MiB = %9d): fast aggregate %d/%d  mean %.3f/%.3f   %s" %
              (inflight, pad, stride % (2 * MiB), (8 * stride) % (64 * MiB), fast, len(res), sum(a for a, b in res) / len(res), sum(b for a, b in res) / len(res),
               " ".join("%.2f/%.2f" % r for r in res)), flush=True)
