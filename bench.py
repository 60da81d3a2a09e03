#!/usr/bin/env python3
"""bench.py -- stereo-pairs/sec of the dense-stereo hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL: either under `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N` (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or as the plain command above, in which case this
process starts those N ranks itself (before it touches a GPU) and relays rank 0's JSON line and the exit code.

One "step" = one pass of the whole hot path (disparity module: census -> 8-path SGM -> WTA ->
medians/LR/range -> interpolate; plane module: vertical derivative + histogram -> plane parameters
-> classify -> connected components + component table) over one batch of `--batch` synthetic 1242x375 stereo pairs
per GPU, D=128, inputs resident in HBM.  Frames shard by id across ranks (weak scaling); the only
exchange is the all-gather of per-frame 256-bin histograms for the plane-parameter schedule.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cart-slam_amd"))

_REAL_STDOUT = 1   # file descriptor the JSON line is written to (main() moves descriptor 1 out of librccl's way)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (6.3 TB/s achievable)


def alg_bytes_per_pair(w, h, D, P):
    """gray input, materialised u8 slabs."""
    return w * h * (2 + 8 + 8 * P + 2 * P * D + 23 + 11)


def alg_bytes_aggregate(w, h, D, P):
    """Path-aggregation launch only: census re-read per path (8 B) + slab write (D B) per pixel and path."""
    return w * h * (8 * P + P * D)


def alg_bytes_wta(w, h, D, P):
    """Winner-takes-all launch only: slab read (D B per pixel and path) + left disparity u16 + packed right view u16."""
    return w * h * (P * D + 4)


CPU_BASELINE_CHILD = r"""
import json, os, sys, time
root, w, h, D, P, budget, verify_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]), sys.argv[7]
verify = json.loads(sys.argv[8])   # {"frame number of synth.make_pair": [the 6 plane parameters the GPU run classified it with]}
sys.path[:0] = [os.path.join(root, "cart-slam_amd"), os.path.join(root, "tests")]
import numpy as np
import oracle_lib as O      # cpu_baseline leg: allowed importer of oracle/
from cartslam import synth
O.build()
def one(l, r):
    d = O.disparity_module(l, r, D, P, 4, radius=2, iterations=1)
    dd, hist = O.plane_derivative(d)
    ok, pp = O.histogram_peak_params(hist)
    pl = O.classify(dd, pp)
    O.ccl(pl)
    return d, dd
# The frames whose GPU outputs bench.py checks after its timed region (also the warm-up: page faults, thread pool): the
# oracle's disparity, and its plane labels under the parameters the GPU run used for that frame (the parameter schedule
# depends on every frame the run has seen; it has its own tests).
keep = {}
for f, params in verify.items():
    l, r, _ = synth.make_pair(w, h, D, 4, frame=int(f))
    d, dd = one(l, r)
    keep["disp_" + f], keep["planes_" + f] = d, O.classify(dd, tuple(params))
if verify_path:
    np.savez(verify_path, **keep)
if budget < 0:      # verification only (multi-rank runs: the CPU baseline itself is timed at N = 1 only)
    print(json.dumps({"verify_only": True}))
    sys.exit(0)
l, r, _ = synth.make_pair(w, h, D, 4)
one(l, r); one(l, r)
times, t_all = [], time.perf_counter()
while len(times) < 3 or (time.perf_counter() - t_all < budget and len(times) < 40):
    t0 = time.perf_counter(); one(l, r); times.append(time.perf_counter() - t0)
times.sort()
print(json.dumps({"median_s": times[len(times) // 2], "min_s": times[0], "max_s": times[-1], "reps": len(times)}))
"""


def cpu_baseline(w, h, D, P, seconds_budget=12.0, verify_path="", verify=None):
    """Times the CPU oracle (the 'port': oracle/cart_oracle.c, OpenMP) on this box's host cores: a child process with no
    torch / GPU in it, threads pinned (OMP_PROC_BIND=close, OMP_PLACES=cores: both have to be set before libgomp
    loads), median over >= 3 whole-pair repetitions, spread reported.  The same child writes the oracle's disparity and
    plane labels of the frames in `verify` ({frame number of synth.make_pair: plane parameters}) to `verify_path` for
    the bench's self-check."""
    import math
    import subprocess
    visible = os.cpu_count() or 1
    try:
        visible = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = visible
    # a container usually sees every host CPU but may run only a share of them (cgroup CPU quota): more threads than
    # that share only fight over it (256 pinned threads on a 16-CPU share ran 4-10x slower than 16, and 4x apart)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            quota = float(txt[0]) if txt[0] != "max" else -1.0
            period = float(txt[1]) if len(txt) > 1 else float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                cores = max(1, min(visible, int(math.ceil(quota / period))))
            break
        except Exception:
            continue
    pinned = cores == visible   # pin only when the threads have the cores to themselves
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_DYNAMIC="false", OMP_PROC_BIND="close" if pinned else "false")
    if pinned:
        env["OMP_PLACES"] = "cores"
    else:
        env.pop("OMP_PLACES", None)
    r = subprocess.run([sys.executable, "-c", CPU_BASELINE_CHILD, ROOT, str(w), str(h), str(D), str(P), str(seconds_budget), verify_path,
                        json.dumps({str(k): [int(x) for x in v] for k, v in (verify or {}).items()})],
                       env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("cpu_baseline child failed: " + r.stderr[-2000:])
    t = json.loads(r.stdout.strip().splitlines()[-1])
    if t.get("verify_only"):
        return None
    return {"value": round(1.0 / t["median_s"], 4), "unit": "stereo-pairs/sec", "cores": cores, "kind": "port",
            "spread": {"fastest": round(1.0 / t["min_s"], 4), "slowest": round(1.0 / t["max_s"], 4), "repetitions": t["reps"]},
            "sample": f"median of {t['reps']} repetitions of 1 pair {w}x{h} D={D} {P} paths + plane labelling + CCL; OpenMP oracle "
                      f"(census, paths, WTA, medians, LR check, interpolation, 5-tap mean, classify run on all {cores} threads; "
                      "the histogram pass, the peak finder and the CCL union-find are serial), " +
                      ("threads pinned close/cores" if pinned else f"{cores} threads = the container's CPU share of {visible} visible CPUs, not pinned")}


def launch_ranks(args):
    """Plain `python bench.py --gpus N` (no RANK in the environment): start N fresh rank processes through
    torch.distributed.run and relay their output.  This process only counts the devices (on ROCm that opens the HIP
    runtime here, which is harmless: the ranks are fresh child processes, nothing is exec'ed over this one)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and not args.allow_shared_gpu:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, this box has {have} "
                         "(--allow-shared-gpu stacks ranks on the GPUs there are: rehearsals only, never a scaling number)\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="stereo pairs per GPU per step")
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--disparities", type=int, default=128)
    ap.add_argument("--paths", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the informational host-buffer (PCIe-inclusive) measurement")
    ap.add_argument("--no-bgr", action="store_true", help="skip the informational 3-channel-input leg (`value_bgr_input`: what ImageDisparityModule::runInternal is handed, disparity.cu:66-67)")
    ap.add_argument("--latency", action="store_true", help="also time ONE resident pair, host-synchronous (informational `single_pair_latency`; off by default so that "
                    "every aggregation / WTA launch of the default command is a full batch and rocprofv3 --stats averages agree with `roofline.launch_ms`)")
    ap.add_argument("--placement-tries", type=int, default=8, help="physical placements of the slab workspace the engine may try at set-up (1 = keep the first; "
                    "the search ends at the first fast set, or after six tries on a box that has none: include/cart_engine.h, cart_engine_tune_placement)")
    ap.add_argument("--no-overlap", action="store_true", help="run the plane stages on the main stream (no two-stream pipelining of consecutive batches)")
    ap.add_argument("--overlap", action="store_true", help="force the two-stream pipelining (the default)")
    ap.add_argument("--repeats", type=int, default=5, help="the timed block of --steps steps is run this many times; `value` is the median block, `spread` the fastest / slowest")
    ap.add_argument("--collective-timeout", type=float, default=120.0, help="seconds after which a torch.distributed call gives up (a dead peer then ends the job non-zero instead of hanging it)")
    ap.add_argument("--sequence", action="store_true", help="also time the batched-sequence mode (BASELINE configs[4]): frames start on rank 0, "
                    "are scattered frame k -> rank k mod N, outputs are gathered back on rank 0; informational, never `value`")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--allow-shared-gpu", action="store_true", help="rehearsals only: let ranks share a GPU when the box has fewer than --gpus")
    ap.add_argument("--pcie-copy", default="both", choices=["both", "narrow", "blit"],
                    help="how the PCIe-inclusive leg downloads its outputs (both: time the two ways one after the other and report the faster; they trade places from box to box)")
    ap.add_argument("--pcie-wgs", type=int, default=8, help="workgroups of the narrow download kernel")
    ap.add_argument("--timing-every", type=int, default=4, help="the engine records its stage events on every n-th step of the timed blocks (1 = every step)")
    ap.add_argument("--chunk", type=int, default=0, help="frames per launch sequence inside a batch (0 = engine default)")
    ap.add_argument("--plan", default="auto", choices=["auto", "slabs", "fused_up"], help="force a launch plan of the SGM core (all bit-identical)")
    args = ap.parse_args()
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    import datetime
    import torch
    import torch.distributed as dist
    from cartslam.pipeline import CollectiveError

    # rank 0 prints ONE JSON line on stdout and nothing else: librccl writes a version banner straight to file descriptor 1
    # when a communicator is created, so from here on descriptor 1 points at stderr and the line goes to the saved descriptor
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: one rank per GPU, the two must agree")
    ndev = torch.cuda.device_count()
    if ndev == 0:
        sys.exit("bench.py needs a GPU (no CPU fallback)")
    if local_rank >= ndev and not args.allow_shared_gpu:
        sys.exit(f"bench.py: rank {rank} (LOCAL_RANK {local_rank}) has no GPU of its own ({ndev} visible); "
                 "--allow-shared-gpu is for rehearsals only")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    if world > 1 or args.sequence:   # the sequence leg runs its scatter / gather through the backend even with one rank
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        # a collective that a peer never joins fails after this long (RCCL: the watchdog thread aborts the process and logs
        # the rank and the operation; gloo: the call raises) -- either way the job ends non-zero instead of hanging
        timeout = datetime.timedelta(seconds=args.collective_timeout)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index), timeout=timeout)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=timeout)
    try:
        run(args, world, rank, dev_index)
    except CollectiveError as e:
        sys.stderr.write(f"bench.py: {e}\n")
        sys.stderr.flush()
        os._exit(3)   # not sys.exit: destroy_process_group / atexit handlers would wait for the peer that is gone


def run(args, world, rank, dev_index):
    import numpy as np
    import torch
    import torch.distributed as dist
    from cartslam import Engine, synth
    from cartslam.pipeline import StereoPipeline

    w, h, D, P, B = args.width, args.height, args.disparities, args.paths, args.batch
    eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1,
                 max_inflight=2 * B, device_id=dev_index)
    if args.plan != "auto":
        eng.set_plan(args.plan)
    if args.chunk:
        eng.set_chunk_frames(args.chunk)
    plan = eng.describe_plan(B)   # what the engine will launch: frames per launch sequence, plan, slabs materialised
    pipe = StereoPipeline(eng, provider="histogram_peak", with_ccl=True, overlap=False if args.no_overlap else "auto")
    # this rank's B distinct frames of the sequence (scene translates 2 px per frame), generated once, resident in HBM
    first_frame = rank * B
    ls, rs = synth.make_batch(B, w, h, D, 4, first_frame=first_frame)
    left = torch.from_numpy(ls).cuda()
    right = torch.from_numpy(rs).cuda()

    torch.cuda.synchronize()
    resident = torch.cuda.current_stream().record_event()   # the inputs are complete from here on (they live in HBM for the whole run)

    def barrier():
        if world > 1:
            dist.barrier()

    # What a caller gets WITHOUT the placement search (every product path defaults to one try: modules.cpp, FrameSharder): one block of --steps
    # steps on the placement the engine was created with, after its own pre-warm; informational (`placement_tuning.value_untuned`), never `value`.
    placement = None
    if args.placement_tries > 1:
        for _ in range(32 + args.warmup):   # the same pre-warm + warm-up the timed blocks get
            pipe.process_batch(left, right, inputs_ready=resident)
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pipe.process_batch(left, right, inputs_ready=resident)
        torch.cuda.synchronize(); barrier()
        untuned = world * B * args.steps / (time.perf_counter() - t0)   # this rank's clock: informational
        # Untimed set-up, like the allocation itself: the engine times its two slab-bound launches on up to --placement-tries physical
        # placements of the slab workspace and keeps the fastest (include/cart_engine.h, cart_engine_tune_placement; profiles/r03_alloc.txt)
        # a dedicated bench process may hold whatever leaves 4 GiB free while it searches; ranks that share a card (rehearsals) keep to two units
        rep = eng.tune_placement(B, args.placement_tries, max_extra_bytes=None if (world == 1 and not args.allow_shared_gpu) else 0, report=True)
        placement = {"tries_allowed": args.placement_tries, "candidates_timed": rep["candidates"], "units": rep["units"],
                     "mode": rep["mode"], "stopped_on": rep["stopped_on"],
                     "launch_pair_ms_first": round(rep["ms_first"], 4), "launch_pair_ms_kept": round(rep["ms_kept"], 4),
                     "launch_pair_ms_slowest_seen": round(rep["ms_slowest_seen"], 4), "seconds": round(rep["seconds"], 3),
                     "what": "cart_engine_tune_placement: aggregation + WTA launch of one batch on fresh physical placements of the slab workspace, fastest "
                             "kept (set-up, outside every timed region).  mode is RELATIVE to the sets the search saw: fast = the kept set is >= 5.5 % under "
                             "the slowest seen (a set with a launch in its slow mode was met and avoided); mixed = the sets differ by 1.5-5.5 %, the fastest is "
                             "kept; uniform = six sets within 1.5 % of each other (all fast or all slow: stages_ms_per_launch tells which -- the aggregation "
                             "launch of the headline configuration takes ~1.45 ms in its fast mode, ~1.55 in its slow one, the WTA 1.16-1.19 / 1.22-1.27)"}
        placement["value_untuned"] = round(untuned, 2)

    # Untimed pre-warm, before the W warm-up steps the contract asks for: first touch of the workspaces (15 GB of slabs), code
    # object loads, allocator pools and the clock ramp of a GPU that has just been handed over idle.
    PREWARM = 32          # ~0.1 s of GPU work at the headline configuration; reported as "prewarm_steps"
    for _ in range(PREWARM):
        pipe.process_batch(left, right, inputs_ready=resident)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        pipe.process_batch(left, right, inputs_ready=resident)
    torch.cuda.synchronize()
    # hipEvents around each stage, on the stream the kernels are launched on, live inside the timed blocks -- on every 4th step: the five
    # records of a step cost 0.02 ms of its stream time (0.85 % at the headline, 2.3 % at configs[1]: profiles/r04_overlap.txt), and the
    # launches of one step are the launches of the next (roofline.launches_timed says how many were averaged)
    eng.set_timing(True, every=args.timing_every)
    # The timed block -- EXACTLY --steps steps between barrier + synchronize on both sides, max over ranks -- is run
    # --repeats times; `value` is the median block (a 20-step block is 60 ms: single blocks differ by a few per cent)
    blocks, last = [], None
    for _ in range(max(1, args.repeats)):
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = pipe.process_batch(left, right, inputs_ready=resident)
        torch.cuda.synchronize(); barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        blocks.append(dt)
    elapsed = sorted(blocks)[len(blocks) // 2]
    stages, ncalls = eng.collect_timing()
    eng.set_timing(False)
    # what the LAST timed step produced for the first and the last frame of this rank's batch (checked against the oracle
    # below, outside every timed region)
    check_frames = sorted({0, B - 1})
    got = {k: {"disp": last["disparity"][k].cpu().numpy(), "planes": last["planes"][k].cpu().numpy(),
               "params": [int(v) for v in last["params"][k].cpu().numpy()]} for k in check_frames} if rank == 0 else {}
    # Informational, never `value`: the same block of --steps steps once more WITHOUT the engine's stage events (six hipEventRecord per step on
    # the launch stream, ~0.02 ms per step: profiles/r04_overlap.txt) -- what the step does uninstrumented.  Same bracket, max over ranks.
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.process_batch(left, right, inputs_ready=resident)
    torch.cuda.synchronize(); barrier()
    dt_plain = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt_plain], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_plain = float(t.item())
    copy_gbps = None
    if rank == 0:
        # achievable-copy ceiling of THIS box ("quote both fractions"): device-to-device copy of 2 GiB,
        # bytes read + bytes written per second
        try:
            src = torch.empty(1 << 31, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
            src.fill_(1)
            for _ in range(2):
                dst.copy_(src)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(5):
                dst.copy_(src)
            c1.record(); torch.cuda.synchronize()
            copy_gbps = 2 * src.numel() * 5 / (c0.elapsed_time(c1) * 1e-3) / 1e9
            del src, dst
        except Exception:
            copy_gbps = None
    # Informational (never `value`): what ONE pair costs when nothing is batched -- the call a SLAM front end makes at camera
    # rate: disparity + plane labelling + CCL of a single resident pair, host-synchronous, median of 30.
    latency = None
    if rank == 0 and args.latency:
        lat_pipe = StereoPipeline(eng, provider="static", static_params=(6, 18, -5, 6, 11, 0), with_ccl=True, overlap=False)
        l1, r1 = left[:1], right[:1]
        for _ in range(5):
            lat_pipe.process_batch(l1, r1)
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            t1 = time.perf_counter()
            lat_pipe.process_batch(l1, r1)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        ts.sort()
        latency = {"ms": round(ts[len(ts) // 2] * 1e3, 4), "fastest_ms": round(ts[0] * 1e3, 4),
                   "what": "one resident pair through disparity + plane labelling (static parameters) + CCL, enqueue to synchronize, median of 30"}
    # Informational (never `value`): the same step on the input the reference's module is actually handed -- 8UC3 BGR images
    # (src/modules/disparity/disparity.cu:66-67 always converts; CARTSLAM_IMAGE_MAKE_GRAYSCALE is never defined).  The BGR->gray
    # conversion is fused into the census kernel's tile load, so the step reads 6 instead of 2 input bytes per pixel (0.4 % of
    # B_alg).  The frames are the gray frames replicated into three channels: (1868 + 9617 + 4899) g + 8192 >> 14 == g, so the
    # disparities must equal the gray run's bit for bit, which is checked.
    bgr = None
    if world == 1 and not args.no_bgr:   # one rank only: a step of the sharded pipeline is a collective
        l3 = left.unsqueeze(-1).expand(-1, -1, -1, 3).contiguous()
        r3 = right.unsqueeze(-1).expand(-1, -1, -1, 3).contiguous()
        torch.cuda.synchronize()
        ready3 = torch.cuda.current_stream().record_event()
        n_bgr = max(4, min(args.steps, 20))
        for _ in range(8):
            o3 = pipe.process_batch(l3, r3, inputs_ready=ready3)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for _ in range(n_bgr):
            o3 = pipe.process_batch(l3, r3, inputs_ready=ready3)
        torch.cuda.synchronize()
        bgr = {"pairs_per_s": round(B * n_bgr / (time.perf_counter() - tb), 1), "steps": n_bgr, "input": "8UC3 BGR (gray replicated), resident in HBM",
               "disparity_equals_gray_run": bool(torch.equal(o3["disparity"], last["disparity"]))}
        del l3, r3, o3
    pcie = None
    if world == 1 and not args.no_pcie:
        # Informational (never `value`): the same step when the caller hands over HOST buffers -- H2D of the 16 pairs
        # from pinned memory, the hot path, D2H of disparity + planes into pinned memory, copies on their own stream.
        hl, hr = left.cpu().pin_memory(), right.cpu().pin_memory()
        hd = torch.empty((B, h, w), dtype=torch.int16).pin_memory()
        hp = torch.empty((B, h, w), dtype=torch.uint8).pin_memory()
        h2d, d2h = torch.cuda.Stream(), torch.cuda.Stream()
        bufs = [(torch.empty_like(left), torch.empty_like(right)) for _ in range(3)]
        consumed = [None] * 3
        uploaded = {}
        def upload(i):   # pinned host -> HBM on the copy stream (SDMA), one step AHEAD of the batch that consumes it
            dl, dr = bufs[i % 3]
            if consumed[i % 3] is not None:
                h2d.wait_event(consumed[i % 3])  # the batch that last read this input buffer has finished its disparity
            with torch.cuda.stream(h2d):
                dl.copy_(hl, non_blocking=True); dr.copy_(hr, non_blocking=True)
                uploaded[i] = h2d.record_event()
        def pcie_step(i):
            dl, dr = bufs[i % 3]
            if i not in uploaded:
                upload(i)
            upload(i + 1)
            ev = uploaded.pop(i)
            o = pipe.process_batch(dl, dr, inputs_ready=ev)   # the main stream waits for the upload
            consumed[i % 3] = torch.cuda.current_stream().record_event()
            download(o)
        def download(o):
            d2h.wait_event(o["done"]) if "done" in o else d2h.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(d2h):
                if method[0] == "narrow":   # a few workgroups write straight into the pinned (device-mapped) host buffers
                    eng.copy_narrow(hd, o["disparity"], args.pcie_wgs); eng.copy_narrow(hp, o["planes"], args.pcie_wgs)
                else:                            # hipMemcpyAsync: a full-width blit kernel on this system
                    hd.copy_(o["disparity"], non_blocking=True); hp.copy_(o["planes"], non_blocking=True)
                o["disparity"].record_stream(d2h); o["planes"].record_stream(d2h)
        method = ["narrow"]
        names = {"narrow": "cart_copy_narrow (8 workgroups) into pinned host memory", "blit": "hipMemcpyAsync"}
        n_warm, n_pcie, step_no, tried = 24, max(4, min(args.steps, 20)), 0, {}
        for m in (["narrow", "blit"] if args.pcie_copy == "both" else [args.pcie_copy]):
            method[0] = m
            for _ in range(n_warm):   # untimed: the caching allocator settles (outputs now live across two more streams)
                pcie_step(step_no); step_no += 1
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for _ in range(n_pcie):
                pcie_step(step_no); step_no += 1
            torch.cuda.synchronize()
            tried[m] = round(B * n_pcie / (time.perf_counter() - tp), 1)
        best = max(tried, key=tried.get)
        pcie = {"pairs_per_s": tried[best], "steps": n_pcie,
                "moved_per_pair": "2 x gray H2D (pinned), s16 disparity + u8 planes D2H",
                "download": names[best], "pairs_per_s_by_download": tried}
    seq = None
    if args.sequence:
        # Informational (never `value`): BASELINE configs[4], a sequence that starts on rank 0.  64 frames when they fit the
        # ranks' batches, else one batch per rank.  Timed twice: one sequence at a time (scatter -> kernels -> gather, wait),
        # and pipelined (StereoPipeline.submit_sequence: the scatter of sequence i+1 and the gather of sequence i-1 on the copy
        # stream beside the kernels of sequence i).
        n_seq = 64 if -(-64 // world) <= B else world * B
        seq_pipe = StereoPipeline(eng, provider="histogram_peak", with_ccl=True, overlap=False if args.no_overlap else "auto",
                                  always_exchange=True)
        sl = sr = None
        if rank == 0:
            k = -(-n_seq // B)
            sl = torch.from_numpy(np.concatenate([ls] * k)[:n_seq]).cuda()
            sr = torch.from_numpy(np.concatenate([rs] * k)[:n_seq]).cuda()
        n_rounds = 8
        def timed(fn):
            torch.cuda.synchronize(); barrier()
            ts = time.perf_counter()
            fn()
            torch.cuda.synchronize(); barrier()
            t = torch.tensor([time.perf_counter() - ts], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        def serial():
            for _ in range(n_rounds):
                seq_got[0] = seq_pipe.process_sequence(sl, sr, n_seq)
                torch.cuda.synchronize()
        def pipelined():
            hs = [seq_pipe.submit_sequence(sl, sr, n_seq) for _ in range(n_rounds)]
            for hnd in hs:
                seq_got[0] = hnd.result()
        seq_got = [None]
        for _ in range(2):
            seq_pipe.process_sequence(sl, sr, n_seq)
        t_serial = timed(serial)
        t_pipe = timed(pipelined)
        if rank == 0:
            assert tuple(seq_got[0]["disparity"].shape) == (n_seq, h, w) and tuple(seq_got[0]["planes"].shape) == (n_seq, h, w)
            seq = {"frames": n_seq, "rounds": n_rounds,
                   "pairs_per_s": round(n_seq * n_rounds / t_pipe, 1),
                   "pairs_per_s_one_sequence_at_a_time": round(n_seq * n_rounds / t_serial, 1),
                   "moved": "scatter of 2 gray images per frame from rank 0, gather of s16 disparity + u8 planes to rank 0",
                   "pipelining": "scatter of sequence i+1 and gather of sequence i-1 on a copy stream beside the kernels of sequence i"}

    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / elapsed
        fpl = plan["frames_per_launch"]  # the engine runs batches as sub-batches (cart_engine_describe_plan)
        fused = plan["plan"] != "slabs"   # the "up" path is computed inside the WTA sweep
        suffix = "" if not fused else "_" + plan["plan"]
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        try:
            tj = json.load(open(tf))
        except Exception:
            tj = {}
        def measured_traffic(kernel, field="hbm_bytes_per_launch"):   # per launch, from the committed PMC passes of THIS configuration, else None
            return tj.get(f"{kernel}_{w}x{h}_D{D}_P{P}_B{fpl}{suffix}", {}).get(field)
        wta_ms = stages.get("wta", 0.0)
        agg_ms = stages.get("aggregate", 0.0)
        if not fused:
            # dominant kernel = the aggregation launch: census re-read + slab write of all P paths
            roof_kernel = "aggregate_kernel (all paths of all frames in one launch)"
            agg_bytes, roof_ms = alg_bytes_aggregate(w, h, D, P) * fpl, agg_ms
            moved_bytes = agg_bytes
            traffic = measured_traffic("aggregate")
        else:
            # Fused plans move fewer bytes than SURVEY 8d's table assumes (some slabs never exist).  The line is still
            # priced with the table's algorithmic bytes -- all P slabs written and read once -- over the time of the
            # launches that together do that work (aggregation + WTA sweep); `traffic` is what they really moved.
            roof_kernel = f"aggregation launch(es) + WTA sweep, plan {plan['plan']} ({plan['slabs_written']} of {P} slabs materialised)"
            agg_bytes, roof_ms = (alg_bytes_aggregate(w, h, D, P) + alg_bytes_wta(w, h, D, P)) * fpl, agg_ms + wta_ms
            # what this plan has to move: census re-read per path, the materialised slabs written once and read once, WTA maps
            moved_bytes = w * h * (8 * P + 2 * plan["slabs_written"] * D + 4) * fpl
            ta, tw = measured_traffic("aggregate"), measured_traffic("wta")
            traffic = ta + tw if ta and tw else None
        achieved = agg_bytes / (roof_ms * 1e-3) / 1e9 if roof_ms > 0 else 0.0
        moved_gbps = moved_bytes / (roof_ms * 1e-3) / 1e9 if roof_ms > 0 else 0.0
        device_ms_per_pair = sum(stages.values()) / fpl if stages else None
        out = {
            "metric": "stereo-pairs/sec @1242x375xD=128; achieved HBM GB/s vs roofline",
            "value": round(value, 2), "unit": "stereo-pairs/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "prewarm_steps": PREWARM, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "repeats": len(blocks), "spread": {"min": round(pairs / max(blocks), 2), "max": round(pairs / min(blocks), 2),
                                               "note": f"value = median of {len(blocks)} timed blocks of {args.steps} steps each"},
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{w}x{h} gray stereo, D={D}, {P}-direction SGM + interpolate(r=2,it=1) + plane "
                                   f"labelling (histogram_peak) + CCL; " +
                                   {(1242, 375, 128, 8): "BASELINE.json configs[2] (the configuration the metric is quoted on)",
                                    (1242, 375, 64, 4): "BASELINE.json configs[1]",
                                    (1920, 1080, 256, 8): "BASELINE.json configs[3]"}.get((w, h, D, P), "not a BASELINE.json configuration"),
                       "pairs_per_gpu_per_step": B, "global_pairs_per_step": world * B, "sharding": f"frames x{world}",
                       "launch_plan": plan, "two_stream_pipelining": pipe.side is not None, "distinct_frames_per_batch": B,
                       "world_size": dist.get_world_size() if world > 1 else 1,
                       "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if world > 1 else None},
            "roofline": {"bound": "hbm", "kernel": roof_kernel,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": "profiles/traffic.json (stored rocprofv3 PMC passes of this configuration: 2 x FETCH_SIZE + WRITE_SIZE), not a counter of this run; bytes that cross the "
                                           "L2 <-> fabric interface -- reads served by the Infinity Cache included (profiles/r05_census_refetch.txt: at 1920x1080 that is what "
                                           "the 1.25x over the algorithmic bytes consists of)" if traffic else None,
                         "frac_basis": "SURVEY 8d table bytes: all P slabs written and read once",
                         "alg_bytes_per_launch": agg_bytes, "frames_per_launch": fpl, "launch_ms": round(roof_ms, 4),
                         "launches_timed": ncalls,
                         # bytes the launch plan really has to move (equal to the table's for plan slabs); frac_moved and the ratio
                         # to the copy rate are priced with these, so a plan that skips slabs is not credited with bytes it never moved
                         "moved_bytes_per_launch": moved_bytes, "frac_moved": round(moved_gbps / HBM_PEAK_GBS, 4),
                         # device-to-device copy of 2 GiB on this box, bytes read + written per second: a reference rate, not a bound --
                         # half of a copy is reads; the aggregation launch is nine tenths writes and on a good placement of its slabs
                         # (placement_tuning) moves its bytes 4-15 % faster than the copy does (write-only stream: 5.7 TB/s)
                         "copy_rate_GBps": round(copy_gbps, 1) if copy_gbps else None,
                         "ratio_to_copy_rate": round(moved_gbps / copy_gbps, 4) if copy_gbps else None},
            "stages_ms_per_launch": {k: round(v, 4) for k, v in stages.items()},
            "job_alg_GBps": round(alg_bytes_per_pair(w, h, D, P) * value / 1e9, 1),
            "device_ms_per_pair_disparity": round(device_ms_per_pair, 4) if device_ms_per_pair else None,
        }
        vi = measured_traffic("aggregate", "valu_insts_per_launch")
        if vi and agg_ms > 0 and not fused:
            # Informational second roofline of the aggregation launch: its stored instruction count (SQ_INSTS_VALU of the committed PMC pass) x 4 cycles
            # per wave64 instruction over 1024 SIMDs, against this run's launch time.  `frac` uses the chip's 2.4 GHz maximum clock (the run does not
            # measure the clock it held; the PMC passes under profiles/ put it near 2.0 GHz under this load).  What binds the launch is the rate at which the memory system
            # takes its slab writes (profiles/r03_min3.txt: without the stores the launch is 20 % shorter and VALU-bound; with them 11 % fewer
            # instructions change nothing): the VALU floor sits 15-20 % under the write-bound time.
            issue = vi * 4.0 / 1024.0 / (agg_ms * 1e-3)
            out["roofline_valu_issue"] = {"bound": "valu-issue (informational: the floor under the write-bound launch)", "kernel": "aggregate_kernel", "valu_insts_per_launch": vi,
                                          "source": "profiles/traffic.json (stored SQ_INSTS_VALU of this configuration)",
                                          "achieved_GHz_equivalent": round(issue / 1e9, 3), "peak_GHz": 2.4, "frac": round(issue / 2.4e9, 4)}
        if wta_ms > 0 and not fused:
            # the second kernel of the path, same accounting (PD + 4 bytes per pixel), HBM-read bound
            wta_bytes = alg_bytes_wta(w, h, D, P) * fpl
            out["roofline_wta"] = {"bound": "hbm", "kernel": "wta_kernel", "achieved": round(wta_bytes / (wta_ms * 1e-3) / 1e9, 1),
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(wta_bytes / (wta_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "traffic": measured_traffic("wta"), "traffic_source": "profiles/traffic.json" if measured_traffic("wta") else None,
                                   "alg_bytes_per_launch": wta_bytes, "launch_ms": round(wta_ms, 4)}
        out["value_without_stage_events"] = round(world * B * args.steps / dt_plain, 2)   # informational: one untimed-by-events block after the timed ones
        if pcie:
            out["pcie_inclusive"] = pcie
            out["value_pcie_inclusive"] = pcie["pairs_per_s"]   # the same step with the pair uploaded and disparity + planes downloaded
        if bgr:
            out["bgr_input"] = bgr
            out["value_bgr_input"] = bgr["pairs_per_s"]   # the same step on 3-channel inputs, what disparity.cu:66-67 is handed
        if seq:
            out["sequence_mode"] = seq
        if latency:
            out["single_pair_latency"] = latency
        if placement:
            out["placement_tuning"] = placement
        verified = None
        if not args.no_cpu_baseline:
            # the CPU leg: the oracle timed on the host cores (N = 1 only); the same child also hands back the oracle's outputs of
            # the first and the last frame of rank 0's batch, against which the LAST TIMED STEP's outputs are compared (outside
            # every timed region; with more than one rank the child only verifies)
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                vp = os.path.join(td, "verify.npz")
                cb = cpu_baseline(w, h, D, P, seconds_budget=12.0 if world == 1 else -1.0, verify_path=vp,
                                  verify={first_frame + k: got[k]["params"] for k in check_frames})
                if cb is not None:
                    out["cpu_baseline"] = cb
                z = np.load(vp)
                bad = []
                for k in check_frames:
                    f = str(first_frame + k)
                    if not (got[k]["disp"] == z["disp_" + f]).all():
                        bad.append(f"disparity of frame {k}: {int((got[k]['disp'] != z['disp_' + f]).sum())} pixels differ")
                    if not (got[k]["planes"] == z["planes_" + f]).all():
                        bad.append(f"planes of frame {k}: {int((got[k]['planes'] != z['planes_' + f]).sum())} pixels differ")
                verified = not bad
                out["verification"] = {"frames": check_frames, "of": "the last timed step", "against": "oracle (cpu_baseline child): disparity bit-exact, "
                                       "plane labels bit-exact under the parameters the run used", "mismatches": bad}
        out["verified"] = verified   # None: not checked (--no-cpu-baseline)
        sys.stdout.flush()
        os.write(_REAL_STDOUT, (json.dumps(out) + "\n").encode())
        if verified is False:
            sys.stderr.write("bench.py: the timed configuration's outputs differ from the oracle: " + "; ".join(bad) + "\n")
            if dist.is_initialized():
                dist.destroy_process_group()
            sys.exit(4)
    if dist.is_initialized():
        if world > 1:
            dist.barrier()   # rank 0 has run its informational legs meanwhile: every rank leaves together, no communicator is torn down under a peer
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
