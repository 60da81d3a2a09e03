"""Diagnostic (GPU): left / right WTA maps of the fused sweep at a D=256 / 4-path case, dumped per engine build (CART_ENGINE_LIB) for comparison."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import numpy as np, torch
from cartslam import Engine, synth
w, h, D, P, md = 400, 26, 256, 4, 0
l, r, _ = synth.make_pair(w, h, D, md, seed=1018, scene="road")
eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=md, smoothing_radius=-1, max_inflight=2)
eng.set_plan("fused_up")
d = eng.compute_disparity(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()).cpu().numpy()
torch.cuda.synchronize()
np.savez(sys.argv[1], disp=d, wl=eng.debug_read(32), wr=eng.debug_read(33))
print("saved", sys.argv[1])
