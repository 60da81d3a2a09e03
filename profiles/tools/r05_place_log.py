#!/usr/bin/env python3
"""Does the distance between the slots of a slab group (= between streams written in lock-step) decide the placement mode?  One fresh engine at the
headline configuration, 16 real steps, then cart_engine_tune_placement a few times; prints every candidate of unit 0: bytes between slots, launch-pair ms,
the kept set re-timed right after it.  Needs the experiment build of profiles/tools/r05_slot_pad.patch (padded candidates + cart_debug_placement_log).  usage: r05_place_log.py [searches = 3] [tries = 17]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cart-slam_amd"))
import torch
from cartslam import Engine, synth
from cartslam.pipeline import StereoPipeline

searches = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tries = int(sys.argv[2]) if len(sys.argv) > 2 else 17
w, h, D, P, B = 1242, 375, 128, 8, 16
eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=2 * B)
pipe = StereoPipeline(eng, provider="histogram_peak", with_ccl=True)
ls, rs = synth.make_batch(B, w, h, D, 4)
left, right = torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda()
for _ in range(16):
    pipe.process_batch(left, right)
torch.cuda.synchronize()
for s in range(searches):
    rep = eng.tune_placement(B, tries, max_extra_bytes=None, report=True)
    print(f"search {s}: mode {rep['mode']} stopped on {rep['stopped_on']} first {rep['ms_first']:.3f} kept {rep['ms_kept']:.3f} slowest {rep['ms_slowest_seen']:.3f} {rep['seconds']:.2f} s")
    for pad, c, k in eng.debug_placement_log():
        print(f"   pad {pad:9d}  candidate {c:.3f}  kept {k:.3f}  ratio {c / k:.3f}")
    for _ in range(8):
        pipe.process_batch(left, right)
    torch.cuda.synchronize()
eng.set_timing(True)
for _ in range(20):
    pipe.process_batch(left, right)
torch.cuda.synchronize()
st, n = eng.collect_timing()
print("stage ms on the kept placement:", {k: round(v, 4) for k, v in st.items()})
