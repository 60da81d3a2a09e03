"""What the engine's per-stage hipEvents (cart_engine_set_timing: bench.py has them on inside its timed block, the contract asks for the dominant
kernel's duration over the timed region) cost the step: blocks of STEPS steps alternating between timing on and off, same process, same placement."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import numpy as np, torch
from cartslam import Engine, synth
from cartslam.pipeline import StereoPipeline
w, h, B = 1242, 375, 16
D, P = int(os.environ.get("DISP", 128)), int(os.environ.get("PATHS", 8))
steps = int(os.environ.get("STEPS", 50))
eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=2 * B)
eng.tune_placement(B, 32, max_extra_bytes=None)
pipe = StereoPipeline(eng, provider="histogram_peak", with_ccl=True, overlap=True)
ls, rs = synth.make_batch(B, w, h, D, 4)
L = torch.from_numpy(ls).cuda(); R = torch.from_numpy(rs).cuda()
for _ in range(40):
    pipe.process_batch(L, R)
torch.cuda.synchronize()
res = {True: [], False: []}
for rep in range(10):
    on = rep % 2 == 0
    eng.set_timing(on)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        pipe.process_batch(L, R)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if on:
        eng.collect_timing()
    res[on].append(dt / steps * 1e3)
eng.set_timing(False)
for on in (True, False):
    v = sorted(res[on])
    print("D=%d P=%d stage events %-3s: ms per step %s  median %.4f  -> %.0f pairs/s" % (D, P, "on" if on else "off", " ".join("%.4f" % x for x in res[on]), v[len(v) // 2], B / v[len(v) // 2] * 1e3))
