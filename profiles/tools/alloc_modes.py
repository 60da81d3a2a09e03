"""Does the WTA / aggregate time depend on where the slab allocation lands?  Engines are created and destroyed in one
process (fresh hipMalloc each time, optionally with a dummy allocation in between to shift the placement)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import numpy as np, torch
from cartslam import Engine, synth
w, h, D, P, B = 1242, 375, 128, 8, 16
ls, rs = synth.make_batch(4, w, h, D, 4)
L = torch.from_numpy(np.concatenate([ls] * 4)).cuda(); R = torch.from_numpy(np.concatenate([rs] * 4)).cuda()
keep = []
for k in range(10):
    if k % 2 == 1:
        keep.append(torch.empty((37 + 11 * k) * 1024 * 1024, dtype=torch.uint8, device="cuda"))   # shifts what the next hipMalloc gets
    eng = Engine(w, h, num_disparities=D, paths=P, smoothing_radius=2, smoothing_iterations=1, max_inflight=B)
    for _ in range(3):
        eng.compute_disparity(L, R)
    torch.cuda.synchronize()
    eng.set_timing(True)
    for _ in range(20):
        eng.compute_disparity(L, R)
    torch.cuda.synchronize()
    st, n = eng.collect_timing()
    print(k, {a: round(b, 4) for a, b in st.items() if a in ("aggregate", "wta")})
    eng.close()
