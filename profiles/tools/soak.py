"""Soak run of the pipelined batch path: the same 16 pairs for STEPS steps (default 400), two-stream overlap on.
Checks every step: disparity bit-identical to the first step's; every CCL id is the smallest pixel index of its
component, labels are uniform inside components and no two 4-neighbours of one label carry different ids
(vectorised on the GPU with torch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import numpy as np, torch
from cartslam import Engine, synth
from cartslam.pipeline import StereoPipeline
w, h, D, P, B = 1242, 375, int(os.environ.get("DISP", 128)), int(os.environ.get("PATHS", 8)), 16   # DISP=64 PATHS=4: configs[1] (split horizontal scans)
steps = int(os.environ.get("STEPS", 400))
eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=2 * B)
if os.environ.get("PLAN"):   # PLAN=fused_up | slabs: soak one launch plan (all must reproduce the first step bit for bit)
    eng.set_plan(os.environ["PLAN"])
if os.environ.get("TUNE", "1") != "0":   # the product set-up: a fast placement of the slab workspace (slot groups of at most 8 GiB, each its own hipMalloc)
    print("tune_placement: %.3f -> %.3f ms" % eng.tune_placement(B, 10), flush=True)
pipe = StereoPipeline(eng, provider="histogram_peak", with_ccl=True, overlap=True)
ls, rs = synth.make_batch(4, w, h, D, 4)
L = torch.from_numpy(np.concatenate([ls] * 4)).cuda(); R = torch.from_numpy(np.concatenate([rs] * 4)).cuda()
first = None
idx = torch.arange(h * w, device="cuda", dtype=torch.int32).view(1, h, w)
t0 = time.time(); bad = 0
for s in range(steps):
    out = pipe.process_batch(L, R)
    torch.cuda.current_stream().wait_event(out["done"]) if "done" in out else None
    disp, planes, ids = out["disparity"], out["planes"], out["ids"]
    if first is None:
        first = disp.clone()
    elif not torch.equal(disp, first):
        bad += 1; print("step", s, "disparity differs in", int((disp != first).sum()), "pixels")
    lab = planes <= 1
    ok = torch.equal(ids < 0, ~lab)
    safe = ids.clamp(min=0).long()
    flat_p = planes.view(B, -1); flat_i = ids.view(B, -1)
    root_label = torch.gather(flat_p, 1, safe.view(B, -1)).view(B, h, w)
    ok &= bool(((root_label == planes) | ~lab).all())                    # a component has one label, that of its root
    ok &= bool(((ids <= idx) | ~lab).all())                              # the id is the smallest index: never above the pixel's own
    root_id = torch.gather(flat_i, 1, safe.view(B, -1)).view(B, h, w)
    ok &= bool(((root_id == ids) | ~lab).all())                          # roots point at themselves
    same_r = lab[:, :, 1:] & lab[:, :, :-1] & (planes[:, :, 1:] == planes[:, :, :-1])
    ok &= bool((ids[:, :, 1:][same_r] == ids[:, :, :-1][same_r]).all())  # 4-neighbours of one label share the id
    same_d = lab[:, 1:] & lab[:, :-1] & (planes[:, 1:] == planes[:, :-1])
    ok &= bool((ids[:, 1:][same_d] == ids[:, :-1][same_d]).all())
    if not ok:
        bad += 1; print("step", s, "CCL invariant violated")
torch.cuda.synchronize()
print(f"soak: {steps} steps x {B} pairs in {time.time() - t0:.1f} s, {bad} bad steps, plan {eng.describe_plan(B)['plan']}")
sys.exit(1 if bad else 0)
