"""Throughput of the native batched-sequence mode (cart_shard_amd, host/src/sharder.cpp) on the GPUs of this box:
BASELINE configs[4] -- 64 frames 1242x375, D=128, 8 paths, 16 frames per GPU and call -- frames resident on GPU 0, scatter
/ all-gather / gather over RCCL.  env GPUS (default: all visible), FRAMES (64)."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cart-slam_amd"))
import numpy as np
from cartslam import synth
w, h, D, P = 1242, 375, 128, 8
gpus = int(os.environ.get("GPUS", "0")) or int(subprocess.run(["bash", "-c", "ls /dev/dri/renderD* | wc -l"], capture_output=True, text=True).stdout or 1)
n = int(os.environ.get("FRAMES", 64))
tmp = tempfile.mkdtemp(dir="/tmp")
ls, rs = synth.make_batch(4, w, h, D, 4)
np.concatenate([ls] * (n // 4)).tofile(tmp + "/left.bin"); np.concatenate([rs] * (n // 4)).tofile(tmp + "/right.bin")
exe = os.path.join(ROOT, "cart-slam_amd", "build", "cart_shard_amd")
r = subprocess.run([exe, tmp + "/left.bin", tmp + "/right.bin", str(w), str(h), str(n), str(D), str(P), str(gpus), str(16 * gpus), tmp],
                   capture_output=True, text=True, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
print(r.stdout.strip() or r.stderr[-500:])
