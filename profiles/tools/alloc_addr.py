"""(The CART_SLAB_SHIFT hook this script's second half used was removed with the experiment; the first half -- fresh engines and their base address -- still runs.)
Experiment build only (profiles/tools/build_variant.sh addr; CART_ENGINE_LIB=.../build/ab/addr/libcart_engine.so): aggregate / WTA time of the
headline configuration against WHERE the slab allocation lands -- fresh engines in one process, dummy allocations in between, and the slabs
shifted inside their allocation (CART_SLAB_SHIFT).  Prints the slab base address with the stage times."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "cart-slam_amd")]
import numpy as np, torch
from cartslam import Engine, synth, _lib
w, h, D, P, B = 1242, 375, 128, 8, 16
ls, rs = synth.make_batch(4, w, h, D, 4)
L = torch.from_numpy(np.concatenate([ls] * 4)).cuda(); R = torch.from_numpy(np.concatenate([rs] * 4)).cuda()
lib = _lib.load(); lib.cart_debug_slab_base.restype = C.c_void_p; lib.cart_debug_slab_base.argtypes = [C.c_void_p]
def measure(tag):
    eng = Engine(w, h, num_disparities=D, paths=P, smoothing_radius=2, smoothing_iterations=1, max_inflight=B)
    base = lib.cart_debug_slab_base(eng._h)
    for _ in range(3):
        eng.compute_disparity(L, R)
    torch.cuda.synchronize(); eng.set_timing(True)
    for _ in range(20):
        eng.compute_disparity(L, R)
    torch.cuda.synchronize()
    st, n = eng.collect_timing()
    print("%-28s slabs at 0x%012x (mod 1 GiB: %4d MiB, mod 2 MiB: %7d B)  aggregate %.4f  wta %.4f" % (tag, base, (base >> 20) & 1023, base & ((2 << 20) - 1), st["aggregate"], st["wta"]), flush=True)
    eng.close()
keep = []
for k in range(8):
    if k % 2 == 1:
        keep.append(torch.empty((37 + 11 * k) * 1024 * 1024, dtype=torch.uint8, device="cuda"))
    measure("fresh %d" % k)
for sh in (0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 16 << 20, 64 << 20, 256 << 20, 512 << 20, (1 << 30) - 256):
    os.environ["CART_SLAB_SHIFT"] = str(sh)
    measure("shift %d" % sh)
