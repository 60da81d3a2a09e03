"""Experiment build (build_variant.sh addr): ONE physical allocation per engine (16 slots + slack, below 8 GiB), the slabs moved inside it
(cart_debug_set_slab_shift): does the level of the aggregation / WTA launch follow the offset inside the same memory?"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import numpy as np, torch
from cartslam import Engine, synth, _lib
w, h, D, P, B = 1242, 375, 128, 8, 16
ls, rs = synth.make_batch(4, w, h, D, 4)
L = torch.from_numpy(np.concatenate([ls] * 4)).cuda(); R = torch.from_numpy(np.concatenate([rs] * 4)).cuda()
os.environ["CART_SLAB_SLACK_MIB"] = "512"
lib = _lib.load(); lib.cart_debug_set_slab_shift.restype = C.c_int; lib.cart_debug_set_slab_shift.argtypes = [C.c_void_p, C.c_size_t]
MiB = 1 << 20
SHIFTS = [0, 256, 4096, 65536, MiB, 2 * MiB, 2 * MiB + 4096, 16 * MiB, 64 * MiB, 64 * MiB + MiB, 128 * MiB, 256 * MiB, 300 * MiB + 256, 0]
ref = None
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    eng = Engine(w, h, num_disparities=D, paths=P, smoothing_radius=2, smoothing_iterations=1, max_inflight=B)
    res = []
    for sh in SHIFTS:
        assert lib.cart_debug_set_slab_shift(eng._h, sh) == 0
        for _ in range(2):
            d = eng.compute_disparity(L, R)
        torch.cuda.synchronize(); eng.set_timing(True)
        for _ in range(10):
            eng.compute_disparity(L, R)
        torch.cuda.synchronize()
        st, n = eng.collect_timing(); eng.set_timing(False)
        if ref is None: ref = d.clone()
        res.append("%.2f/%.2f%s" % (st["aggregate"], st["wta"], "" if bool((d == ref).all()) else "!"))
    print("engine %d: %s" % (k, "  ".join(res)), flush=True)
    eng.close()
print("shifts:", SHIFTS)
