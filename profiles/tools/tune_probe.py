"""Product build: Engine.tune_placement on fresh engines (headline configuration, max_inflight 32): launch-pair time before / after, the time
the call took, and the stage times of real calls afterwards; checks that the disparity is unchanged.  usage: tune_probe.py [tries] [engines]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import numpy as np, torch
from cartslam import Engine, synth
tries = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w, h, D, P, B = 1242, 375, 128, 8, 16
ls, rs = synth.make_batch(4, w, h, D, 4)
L = torch.from_numpy(np.concatenate([ls] * 4)).cuda(); R = torch.from_numpy(np.concatenate([rs] * 4)).cuda()
ref = None
def stages(eng):
    for _ in range(3):
        d = eng.compute_disparity(L, R)
    torch.cuda.synchronize(); eng.set_timing(True)
    for _ in range(16):
        eng.compute_disparity(L, R)
    torch.cuda.synchronize()
    st, n = eng.collect_timing(); eng.set_timing(False)
    return d, st
for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    eng = Engine(w, h, num_disparities=D, paths=P, smoothing_radius=2, smoothing_iterations=1, max_inflight=32)
    d0, s0 = stages(eng)
    if ref is None: ref = d0.clone()
    t0 = time.perf_counter(); a, b = eng.tune_placement(B, tries); dt = time.perf_counter() - t0
    d1, s1 = stages(eng)
    print("engine %d: before %.3f/%.3f   tune: %.3f -> %.3f ms in %.0f ms   after %.3f/%.3f   %s" %
          (k, s0["aggregate"], s0["wta"], a, b, dt * 1e3, s1["aggregate"], s1["wta"], "ok" if bool((d0 == ref).all() and (d1 == ref).all()) else "MISMATCH"), flush=True)
    eng.close()
