"""Experiment build (build_variant.sh addr): aggregate / WTA time of the headline configuration against how the slab workspace is backed --
max_inflight 32 (15.3 GB of slabs); CART_SLAB_CHUNK_MIB = 0: one hipMalloc; unset: the product rule (one address range, physical allocations
of <= 8 GiB - 64 MiB); other values: that chunk size.  6 fresh engines per point; checks one disparity image against the first engine's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import numpy as np, torch
from cartslam import Engine, synth
w, h, D, P, B = 1242, 375, 128, 8, 16
ls, rs = synth.make_batch(4, w, h, D, 4)
L = torch.from_numpy(np.concatenate([ls] * 4)).cuda(); R = torch.from_numpy(np.concatenate([rs] * 4)).cuda()
ref = None
for arg in (sys.argv[1:] or ["32:0", "32:", "32:4096", "32:2048", "32:1024", "32:0", "32:", "64:", "16:"]):
    inflight, chunk = arg.split(":")
    if chunk == "": os.environ.pop("CART_SLAB_CHUNK_MIB", None)
    else: os.environ["CART_SLAB_CHUNK_MIB"] = chunk
    res = []
    for k in range(6):
        eng = Engine(w, h, num_disparities=D, paths=P, smoothing_radius=2, smoothing_iterations=1, max_inflight=int(inflight))
        for _ in range(3):
            d = eng.compute_disparity(L, R)
        torch.cuda.synchronize(); eng.set_timing(True)
        for _ in range(16):
            eng.compute_disparity(L, R)
        torch.cuda.synchronize()
        st, n = eng.collect_timing()
        if ref is None: ref = d.clone()
        ok = bool((d == ref).all())
        res.append("%.3f/%.3f%s" % (st["aggregate"], st["wta"], "" if ok else " MISMATCH"))
        eng.close()
    print("max_inflight %s, chunk MiB %-7s: %s" % (inflight, chunk if chunk != "" else "product", "  ".join(res)), flush=True)
