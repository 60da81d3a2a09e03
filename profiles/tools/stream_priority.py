"""Does a high-priority main stream keep the side stream's plane stages out of the way of the SGM kernels?
The bench's loop (two-stream pipelining) for D/P from the environment, main stream at priority 0 and -1."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cart-slam_amd"))
import numpy as np, torch
from cartslam import Engine, synth
from cartslam.pipeline import StereoPipeline
w, h, D, P, B = 1242, 375, int(os.environ.get("D", 128)), int(os.environ.get("P", 8)), 16
ls, rs = synth.make_batch(4, w, h, D, 4)
L = torch.from_numpy(np.concatenate([ls] * 4)).cuda(); R = torch.from_numpy(np.concatenate([rs] * 4)).cuda()
for label, prio, overlap in (("one stream", 0, False), ("two streams, main prio 0", 0, True), ("two streams, main prio -1", -1, True)):
    eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=2 * B)
    main = torch.cuda.Stream(priority=prio)
    with torch.cuda.stream(main):
        pipe = StereoPipeline(eng, provider="histogram_peak", with_ccl=True, overlap=overlap)
        for _ in range(5): pipe.process_batch(L, R)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): pipe.process_batch(L, R)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"D={D} P={P} {label:28s} {B * 40 / el:9.1f} pairs/s  {el / 40 * 1e3:.3f} ms per step", flush=True)
    eng.close()
