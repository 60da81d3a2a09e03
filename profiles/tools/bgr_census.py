import os, sys
sys.path[:0] = ["cart-slam_amd"]
import numpy as np, torch
from cartslam import Engine, synth
w, h, D, P, B = 1242, 375, 128, 8, 16
eng = Engine(w, h, num_disparities=D, paths=P, smoothing_radius=2, smoothing_iterations=1, max_inflight=B)
for ch in (1, 3):
    ls, rs = synth.make_batch(4, w, h, D, 4, channels=ch)
    L = torch.from_numpy(np.concatenate([ls] * 4)).cuda(); R = torch.from_numpy(np.concatenate([rs] * 4)).cuda()
    for _ in range(4): eng.compute_disparity(L, R)
    torch.cuda.synchronize(); eng.set_timing(True)
    for _ in range(30): eng.compute_disparity(L, R)
    torch.cuda.synchronize(); st, n = eng.collect_timing(); eng.set_timing(False)
    print("channels", ch, {k: round(v, 4) for k, v in st.items()}, "sum", round(sum(st.values()), 4))
