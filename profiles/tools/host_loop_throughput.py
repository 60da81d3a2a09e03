"""Throughput of the C++ frame loop (cart_slam_amd, per-frame module calls, <= 12 frames in flight) at 1242x375.
Writes a 240-frame PGM sequence to /tmp, runs the reference-style config, reads the timing CSV."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from cartslam import synth
tmp = os.environ.get("PREPARE_ONLY") or tempfile.mkdtemp(dir="/tmp")  # PREPARE_ONLY=<dir>: write the data set + the four module lists there and stop
os.makedirs(tmp, exist_ok=True)
d = os.path.join(tmp, "ds", "sequences", "00"); os.makedirs(d + "/image_2"); os.makedirs(d + "/image_3")
n = int(os.environ.get("N", 240))   # frames 4.. are hard links to the four distinct pairs
base = [synth.make_pair(1242, 375, 128, 4, frame=f) for f in range(4)]
SP = [{"type": "superpixels", "initial_iterations": 24, "iterations": 8, "block_size": 12, "reset_iterations": 64},
      {"type": "disparity", "smoothing_radius": 2, "smoothing_iterations": 1}, {"type": "disparity_derivative"}, {"type": "depth"},
      {"type": "superpixel_disparity_planeseg", "parameter_provider": {"type": "histogram_peak"}}]  # config/modules/kitti-planeseg.json minus optflow / GUI
for f in range(n):
    l, r, _ = base[f % 4]
    for cam, img in ((2, l), (3, r)):
        if f >= 4:
            os.link(f"{d}/image_{cam}/{f % 4:06d}.pgm", f"{d}/image_{cam}/{f:06d}.pgm")
            continue
        with open(f"{d}/image_{cam}/{f:06d}.pgm", "wb") as fh:
            fh.write(b"P5\n1242 375\n255\n"); fh.write(img.tobytes())
json.dump({"type": "kitti", "path": os.path.join(tmp, "ds"), "sequence": 0}, open(tmp + "/src.json", "w"))
D128 = {"type": "disparity", "num_disparities": 128, "paths": 8, "smoothing_radius": 2, "smoothing_iterations": 1}
DREF = {"type": "disparity", "smoothing_radius": 2, "smoothing_iterations": 1}
PEAK = {"type": "disparity_planeseg", "parameter_provider": {"type": "histogram_peak"}}
CONFIGS = [("disparity D=128 P=8 + planeseg", [D128, PEAK]),
           ("reference default (D=256, 4 paths) + planeseg", [DREF, PEAK]),
           ("kitti-planeseg.json minus optflow (superpixels 24/8 sweeps, D=256 4 paths, derivative, depth, superpixel planeseg)", SP),
           ("kitti-planeseg.json complete (+ optflow stand-in R=8, temporal smoothing)", SP[:1] + [{"type": "optflow"}] + SP[1:-1] + [dict(SP[-1], use_temporal_smoothing=True)]),
           ("disparity D=128 P=8 alone", [D128])]   # index 4: only with ONLY=4
ci = -1
for name, mods in CONFIGS:
    ci += 1
    if str(ci) not in os.environ.get("ONLY", "0,1,2,3").split(","):   # ONLY=0,1: run these module lists only
        continue
    json.dump(mods, open(tmp + "/mod.json", "w"))
    if os.environ.get("PREPARE_ONLY"):
        json.dump(mods, open(tmp + "/mod_%d.json" % sum(os.path.exists(tmp + "/mod_%d.json" % i) for i in range(8)), "w"))
        continue
    exe = os.path.join(ROOT, "cart-slam_amd", "build", "cart_slam_amd")
    t0 = time.time()
    r = subprocess.run([exe, tmp + "/src.json", tmp + "/mod.json", "--timing", tmp + "/t.csv"] + os.environ.get("EXTRA", "").split(), capture_output=True, text=True)   # EXTRA="--inflight 24"
    wall = time.time() - t0
    rows = [ln.strip().split(";") for ln in open(tmp + "/t.csv")][1:]
    fr = [x for x in rows if x[0] == "Frame"]
    span_ms = max(int(x[4]) for x in fr) - min(int(x[2]) for x in fr)
    disp = [int(x[6]) for x in rows if x[0] == "ImageDisparity"]
    print(f"{name}: {r.stdout.strip()} | {len(fr)} frames in {span_ms} ms -> {len(fr) / max(span_ms, 1) * 1e3:.0f} pairs/s; "
          f"median ImageDisparity module time {sorted(disp)[len(disp) // 2]} us; process wall {wall:.2f} s")
    ends = sorted(int(x[4]) for x in fr)
    if len(ends) > 96:
        print(f"   steady state (frames 48..{len(ends)}): {(len(ends) - 48) / max(ends[-1] - ends[47], 1) * 1e3:.0f} pairs/s")
    fr.sort(key=lambda x: int(x[1]))
    inits = [int(x[2]) for x in fr]
    print("   frame init spacing ms:", [b - a for a, b in zip(inits[:16], inits[1:17])], " frame durations us:", [int(x[6]) for x in fr[:12]])
    for mod in ("PlaneSegmentation", "SuperPixelDetect", "SPPlaneSegmentation", "ImageDisparityDerivative", "Depth", "ImageOpticalFlow"):
        ps = [int(x[6]) for x in rows if x[0] == mod]
        if ps:
            print(f"   median {mod} us:", sorted(ps)[len(ps) // 2])
    ds = [int(x[6]) for x in rows if x[0] == "DataSource"]
    print("   median DataSource (read + decode + upload) us:", sorted(ds)[len(ds) // 2])
