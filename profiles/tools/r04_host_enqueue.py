"""Host time of one pipelined step (StereoPipeline.process_batch: one engine call + the plane stages' launches through ctypes) against the GPU time of the step:
is any configuration bound by the Python host?  STEPS steps are enqueued without synchronisation; `enqueue` = wall time until the last call returns,
`total` = until the GPU has finished."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import numpy as np, torch
from cartslam import Engine, synth
from cartslam.pipeline import StereoPipeline
w, h, B = 1242, 375, 16
for D, P in ((64, 4), (128, 8)):
    eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=2 * B)
    pipe = StereoPipeline(eng, provider="histogram_peak", with_ccl=True, overlap=True)
    ls, rs = synth.make_batch(B, w, h, D, 4)
    L = torch.from_numpy(ls).cuda(); R = torch.from_numpy(rs).cuda()
    for _ in range(40):
        pipe.process_batch(L, R)
    torch.cuda.synchronize()
    for steps in (1, 8, 50, 200):
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.process_batch(L, R)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("D=%d P=%d  %3d steps: enqueue %.3f ms per step, total %.3f ms per step" % (D, P, steps, (t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3), flush=True)
    eng.close()
