"""GPU box: engine-only loop of compute_disparity with early = 0 / 1 -- host time per call (no synchronisation) and device
throughput.  usage: early_probe.py [D] [P] [B] [W] [H]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "cart-slam_amd"))
import torch
from cartslam import Engine, synth

D, P, B, W, H = (int(a) for a in (sys.argv[1:] + ["64", "4", "16", "1241", "376"][len(sys.argv) - 1:]))
ls, rs = synth.make_batch(B, W, H, D, 4)
l = torch.from_numpy(ls).cuda(); r = torch.from_numpy(rs).cuda()
eng = Engine(W, H, num_disparities=D, paths=P, min_disparity=4, max_inflight=2 * B)
outs = [torch.empty((B, H, W), dtype=torch.int16, device="cuda") for _ in range(2)]
torch.cuda.synchronize()
ready = torch.cuda.current_stream().record_event()
for early in (0, 1, 0, 1):
    for _ in range(10):
        eng.compute_disparity(l, r, out=outs[0], early=bool(early), inputs_ready=ready)
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for k in range(100):
        h0 = time.perf_counter()
        eng.compute_disparity(l, r, out=outs[k & 1], early=bool(early), inputs_ready=ready)
        host.append(time.perf_counter() - h0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.sort()
    print("early=%d  host/call median %.1f us  max %.1f us  enqueue loop %.2f ms  total %.2f ms  -> %.3f ms/step" %
          (early, host[50] * 1e6, host[-1] * 1e6, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 10), flush=True)
