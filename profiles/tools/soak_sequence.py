"""Soak run of the pipelined batched-sequence mode on one GPU (one RCCL rank): SEQS sequences (default 600) of 16 pairs
submitted back to back with up to three handles outstanding -- scatter, gather and the histogram exchange go through RCCL
on the copy / side streams beside the kernels.  Every gathered disparity and plane image must equal the first sequence's
(static plane parameters, same input)."""
import os, socket, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import numpy as np, torch
import torch.distributed as dist
from cartslam import Engine, synth
from cartslam.pipeline import StereoPipeline
w, h, D, P, B = 1242, 375, 128, 8, 16
seqs = int(os.environ.get("SEQS", 600))
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=2 * B)
pipe = StereoPipeline(eng, provider="static", static_params=(6, 18, -5, 6, 11, 0), with_ccl=True, overlap=True, always_exchange=True)
ls, rs = synth.make_batch(B, w, h, D, 4)
L, R = torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda()
first, bad, pending = None, 0, []
t0 = time.time()
def check(got, i):
    global first, bad
    if first is None:
        first = (got["disparity"].clone(), got["planes"].clone())
    elif not (torch.equal(got["disparity"], first[0]) and torch.equal(got["planes"], first[1])):
        bad += 1; print("sequence", i, "differs from the first")
for i in range(seqs):
    pending.append((i, pipe.submit_sequence(L, R, B)))
    if len(pending) > 3:
        k, hd = pending.pop(0)
        check(hd.result(), k)
for k, hd in pending:
    check(hd.result(), k)
torch.cuda.synchronize()
print(f"soak_sequence: {seqs} pipelined sequences x {B} pairs in {time.time() - t0:.1f} s ({seqs * B / (time.time() - t0):.0f} pairs/s incl. checks), {bad} bad")
dist.destroy_process_group()
sys.exit(1 if bad else 0)
