"""Stage times of cart_compute_disparity_batch alone (no plane stages, one stream): batch 16, 1242x375, D=128, 8 paths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import numpy as np, torch
from cartslam import Engine, synth
w, h, D, P, B = int(os.environ.get("W", 1242)), int(os.environ.get("H", 375)), int(os.environ.get("D", 128)), int(os.environ.get("P", 8)), int(os.environ.get("B", 16))
eng = Engine(w, h, num_disparities=D, paths=P, smoothing_radius=2, smoothing_iterations=1, max_inflight=B)
ls, rs = synth.make_batch(max(1, min(4, B)), w, h, D, 4)
L = torch.from_numpy(np.concatenate([ls] * (B // 4 + 1))[:B]).cuda(); R = torch.from_numpy(np.concatenate([rs] * (B // 4 + 1))[:B]).cuda()
for _ in range(4):
    eng.compute_disparity(L, R)
torch.cuda.synchronize()
eng.set_timing(True)
for _ in range(30):
    eng.compute_disparity(L, R)
torch.cuda.synchronize()
st, n = eng.collect_timing()
tot = sum(st.values())
print(os.environ.get("TAG", ""), {k: round(v, 4) for k, v in st.items()}, "sum", round(tot, 4), "per-frame", round(tot / B, 4))
