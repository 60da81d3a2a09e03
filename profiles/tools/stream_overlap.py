"""Does running sub-batches on several streams overlap the VALU-bound aggregation of one with the HBM-bound WTA of another?
Disparity module only (1242x375, D=128, 8 paths), frames resident; per configuration: pairs/s over ~60 batches.
env: PLANS="slabs pairs"  MODES="1x16 2x8 2x16 3x8 4x4"  (streams x frames per call)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cart-slam_amd"))
import numpy as np
import torch
from cartslam import Engine, synth

w, h, D, P = 1242, 375, int(os.environ.get("D", 128)), int(os.environ.get("P", 8))
ls, rs = synth.make_batch(4, w, h, D, 4)
for plan in os.environ.get("PLANS", "slabs pairs").split():
    for mode in os.environ.get("MODES", "1x16 2x8 2x16 3x8 4x4").split():
        ns, nf = (int(v) for v in mode.split("x"))
        eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=2 * ns * nf)
        eng.set_plan(plan)
        left = torch.from_numpy(np.concatenate([ls] * ((nf + 3) // 4))[:nf]).cuda()
        right = torch.from_numpy(np.concatenate([rs] * ((nf + 3) // 4))[:nf]).cuda()
        streams = [torch.cuda.Stream() for _ in range(ns)]
        outs = [torch.empty((nf, h, w), dtype=torch.int16, device="cuda") for _ in range(ns)]
        rounds = max(4, 96 // (ns * nf) * 6)

        def run(n):
            for _ in range(n):
                for s, o in zip(streams, outs):
                    with torch.cuda.stream(s):
                        eng.compute_disparity(left, right, out=o)
        run(3); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(rounds); torch.cuda.synchronize()
        el = time.perf_counter() - t0
        same = all(bool((o == outs[0]).all()) for o in outs)
        print(f"{plan:9s} {mode:5s} {ns * nf * rounds / el:8.1f} pairs/s   {el / rounds / ns / nf * 16e3:6.3f} ms per 16 pairs   streams agree: {same}  status {eng.device_status()}", flush=True)
        eng.close()
