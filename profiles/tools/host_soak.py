"""Soak run of the C++ frame loop (cart_slam_amd, 32 frames in flight, coalescing on): N frames (default 4000) cycling
through four distinct pairs; the frames still held by the retention ring at the end are dumped and their disparity /
plane images must equal the batched engine's for the same pair (planes with the static provider)."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import numpy as np, torch
from cartslam import Engine, synth
n = int(os.environ.get("N", 4000))
w, h, D, P = 1242, 375, 128, 8
tmp = tempfile.mkdtemp(dir="/tmp")
d = os.path.join(tmp, "ds", "sequences", "00"); os.makedirs(d + "/image_2"); os.makedirs(d + "/image_3"); os.makedirs(tmp + "/dump")
base = [synth.make_pair(w, h, D, 4, frame=f) for f in range(4)]
for f in range(n):
    for cam, k in ((2, 0), (3, 1)):
        path = f"{d}/image_{cam}/{f:06d}.pgm"
        if f >= 4:
            os.link(f"{d}/image_{cam}/{f % 4:06d}.pgm", path)
        else:
            with open(path, "wb") as fh:
                fh.write(b"P5\n%d %d\n255\n" % (w, h)); fh.write(base[f][k].tobytes())
json.dump({"type": "kitti", "path": os.path.join(tmp, "ds"), "sequence": 0}, open(tmp + "/src.json", "w"))
static = {"type": "static", "horizontal_range_min": 2, "horizontal_range_max": 40, "vertical_range_min": -3, "vertical_range_max": 2}
json.dump([{"type": "disparity", "num_disparities": D, "paths": P, "smoothing_radius": 2, "smoothing_iterations": 1},
           {"type": "disparity_planeseg", "parameter_provider": static}], open(tmp + "/mod.json", "w"))
exe = os.path.join(ROOT, "cart-slam_amd", "build", "cart_slam_amd")
r = subprocess.run([exe, tmp + "/src.json", tmp + "/mod.json", "--inflight", "32", "--dump", tmp + "/dump"], capture_output=True, text=True)
print(r.stdout.strip(), r.stderr.strip()[-300:])
assert r.returncode == 0
eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=4)
L = torch.from_numpy(np.stack([b[0] for b in base])).cuda(); R = torch.from_numpy(np.stack([b[1] for b in base])).cuda()
want_d = eng.compute_disparity(L, R)
hist = torch.zeros(256, dtype=torch.int32, device="cuda")
want_p = eng.plane_classify(eng.plane_derivative_hist(want_d, hist), (2, 40, -3, 2, 21, 0))   # centres do not enter the classification
want_d, want_p = want_d.cpu().numpy(), want_p.cpu().numpy()
checked = bad = 0
for fid in range(1, n + 1):
    pd, pp = f"{tmp}/dump/{fid}_disparity.bin", f"{tmp}/dump/{fid}_planes.bin"
    if not os.path.exists(pd):
        continue
    checked += 1
    k = (fid - 1) % 4
    ok = np.array_equal(np.fromfile(pd, np.int16).reshape(h, w), want_d[k]) and np.array_equal(np.fromfile(pp, np.uint8).reshape(h, w), want_p[k])
    bad += not ok
print(f"host soak: {n} frames, {checked} retained frames checked, {bad} differ")
sys.exit(1 if bad or checked < 8 else 0)
