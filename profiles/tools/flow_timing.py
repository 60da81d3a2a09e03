"""Device time of cart_optical_flow at KITTI size for a few search ranges."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import torch
from cartslam import Engine, synth
w, h = 1242, 375
eng = Engine(w, h, num_disparities=0, paths=0, max_inflight=2)
cur = torch.from_numpy(synth.make_pair(w, h, 128, 4, seed=9, frame=1)[0]).cuda()
prev = torch.from_numpy(synth.make_pair(w, h, 128, 4, seed=9, frame=0)[0]).cuda()
for R, B in ((4, 2), (8, 2), (16, 2), (8, 1), (8, 3)):
    for _ in range(3):
        eng.optical_flow(cur, prev, R, B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        eng.optical_flow(cur, prev, R, B)
    e1.record(); torch.cuda.synchronize()
    print(f"radius {R:2d} block {B}: {e0.elapsed_time(e1) / 10:.3f} ms per frame ({(2 * R + 1) ** 2} candidates x {(2 * B + 1) ** 2} window)")
