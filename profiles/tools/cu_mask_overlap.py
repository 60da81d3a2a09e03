"""Plane stages on a side stream that may only use a few CUs (hipExtStreamCreateWithCUMask): do the SGM kernels of the
next batch keep their pace then?  One stream / ordinary side stream / masked side streams, D and P from the environment."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cart-slam_amd"))
import numpy as np, torch
from cartslam import Engine, synth
from cartslam.pipeline import StereoPipeline
w, h, D, P, B = 1242, 375, int(os.environ.get("D", 128)), int(os.environ.get("P", 8)), 16
torch.cuda.init(); torch.zeros(1, device="cuda")
hip = ctypes.CDLL("libamdhip64.so")   # the copy torch has already loaded
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words); s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    if rc != 0: raise RuntimeError(f"hipExtStreamCreateWithCUMask: {rc}")
    return torch.cuda.ExternalStream(s.value)
ls, rs = synth.make_batch(4, w, h, D, 4)
L = torch.from_numpy(np.concatenate([ls] * 4)).cuda(); R = torch.from_numpy(np.concatenate([rs] * 4)).cuda()
MASKS = {"one stream": None, "side stream, all CUs": "all",
         "side: CUs 0-31": [0xffffffff, 0, 0, 0, 0, 0, 0, 0],
         "side: CUs 0-63": [0xffffffff, 0xffffffff, 0, 0, 0, 0, 0, 0],
         "side: every 8th CU (32)": [0x01010101] * 8,
         "side: every 4th CU (64)": [0x11111111] * 8,
         "side: every 16th CU (16)": [0x00010001] * 8}
for label, m in MASKS.items():
    eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=2 * B)
    pipe = StereoPipeline(eng, provider="histogram_peak", with_ccl=True, overlap=m is not None)
    if isinstance(m, list): pipe.side = masked_stream(m)
    first = None
    for _ in range(8): o = pipe.process_batch(L, R)
    torch.cuda.synchronize(); first = (o["disparity"].clone(), o["planes"].clone(), o["ids"].clone())
    t0 = time.perf_counter()
    for _ in range(50): o = pipe.process_batch(L, R)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    same = all(torch.equal(a, b) for a, b in zip(first[:2], (o["disparity"], o["planes"])))
    print(f"D={D} P={P} {label:28s} {B * 50 / el:9.1f} pairs/s  {el / 50 * 1e3:.3f} ms per step  outputs stable: {same}", flush=True)
    eng.close()
