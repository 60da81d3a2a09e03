"""Batched-frame driver of the hot path: disparity -> plane derivative/histogram -> plane
parameters -> classify -> connected components, for a batch of frames on one GPU, optionally
frame-sharded over ranks (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm).

It mirrors the per-frame order of the reference's two modules
(ImageDisparityModule::runInternal, src/modules/disparity/disparity.cu:49-80;
DisparityPlaneSegmentationModule::runInternal + updatePlaneParameters,
src/modules/planeseg/planeseg.cu:246-403) with the frame id made explicit, so that a batch gives
exactly what the reference's module would give when fed the same frames one by one in id order.

Sharding: frames are independent up to the plane-parameter refresh.  The
histogram the reference keeps is cumulative over frames and is consulted when id % update_interval
== 1 (planeseg.cu:381-395), so the only exchange step is an all-gather of the per-frame 256-bin
histograms (1 KB per frame); every rank then replays the schedule redundantly.

Batched-sequence mode (BASELINE.json configs[4]): a sequence that lives on one rank is dealt out frame k -> rank k mod
world with grouped point-to-point transfers (rank 0 has a direct xGMI link to every peer: no ring), every rank runs its
share as one batch, and the outputs travel back the same way.  Sequences are double-buffered: the transfers run on a copy
stream of their own, the scatter of sequence i+1 and the gather of sequence i-1 beside the kernels of sequence i
(SequencePipeliner).
"""
import numpy as np

from .engine import PlaneParams, find_plane_params


class CollectiveError(RuntimeError):
    """A torch.distributed call of the sharded path failed or timed out; names the rank and the call so that a launcher's
    log says which peer to look at (bench.py exits non-zero on it)."""

    def __init__(self, rank, what, cause):
        super().__init__(f"rank {rank}: collective '{what}' failed: {type(cause).__name__}: {cause}")
        self.rank, self.what, self.cause = rank, what, cause


def _guarded(what, group, fn):
    """Runs one communication step; any failure (a dead peer's timeout, a backend error) becomes a CollectiveError."""
    import torch.distributed as dist
    try:
        return fn()
    except CollectiveError:
        raise
    except Exception as e:   # noqa: BLE001 -- every backend raises its own types
        try:
            rank = dist.get_rank(group)
        except Exception:   # noqa: BLE001
            rank = -1
        raise CollectiveError(rank, what, e) from e


class PlaneParameterSchedule:
    """Deterministic, in-id-order restatement of the reference's histogram bookkeeping
    (planeseg.cu:271-283 accumulate, :381-395 refresh/reset, :405-458 provider)."""

    def __init__(self, provider="histogram_peak", static_params=None, update_interval=30, reset_interval=10,
                 finder=find_plane_params):
        if provider not in ("histogram_peak", "static"):
            raise ValueError("Unknown parameter provider type.")  # cartconfig.cpp:77
        self.provider = provider
        self.update_interval, self.reset_interval = update_interval, reset_interval
        self.cum = np.zeros(256, np.int64)
        self.params = PlaneParams(*(static_params or (0, 0, 0, 0, 0, 0)))
        self._finder = finder

    def advance(self, first_id, hists):
        """hists: int32 [n,256] of frames first_id .. first_id+n-1 (ids are 1-based like
        SystemRunData::id, cartslam.cpp:194).  Returns the list of PlaneParams each frame is
        classified with."""
        out = []
        for k in range(hists.shape[0]):
            fid = first_id + k
            self.cum += hists[k]
            if fid % self.update_interval == 1:  # planeseg.cu:381
                h32 = self.cum.astype(np.int32)
                if fid % (self.update_interval * self.reset_interval) == 1:  # planeseg.cu:391-394
                    self.cum[:] = 0
                if self.provider == "histogram_peak":
                    _, self.params = self._finder(h32, self.params)
            out.append(PlaneParams(*self.params.as_tuple()))
        return out


def shard_ids(first_id, n_local, rank, world):
    """Global 1-based ids of this rank's frames: frame f -> rank (f - first_id) mod world."""
    return [first_id + k * world + rank for k in range(n_local)]


def share_of(n_total, rank, world):
    """Frames of an n_total-frame sequence that land on `rank` (frame k -> rank k mod world): the first n_total % world
    ranks hold one frame more than the others."""
    return len(range(rank, n_total, world))


def _global_rank(group, r):
    import torch.distributed as dist
    return dist.get_global_rank(group, r) if group is not None else r


def _padded(t, n):
    """t with zero frames appended up to n frames (the collectives move equal shares; short shares travel padded)."""
    import torch
    if t.shape[0] == n:
        return t.contiguous()
    return torch.cat([t, t.new_zeros((n - t.shape[0],) + tuple(t.shape[1:]))])


def scatter_sequence(frames, n_total, like, root=0, group=None):
    """Batched-sequence mode (BASELINE.json configs[4]: a sequence that starts on one rank): frame k of the sequence goes
    to rank k mod world, the interleaving of shard_ids; when n_total is not a multiple of the world size the first ranks
    get one frame more (share_of; the short shares travel padded with one zero frame, which is dropped on arrival).
    `frames` is the [n_total, ...] tensor on `root` (ignored elsewhere); `like` = (per-frame shape, dtype, device) so that
    the other ranks can post their receive.  Returns this rank's [share_of(n_total, rank, world), ...] tensor.  One scatter
    (RCCL: grouped ncclSend / ncclRecv over xGMI, rank 0 has a direct link to every peer) -- not a ring collective."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if n_total < 1:
        raise ValueError(f"sequence length {n_total} must be positive")
    shape, dtype, device = like
    on_host = dist.get_backend(group) == "gloo"   # RCCL moves device tensors; gloo (CPU rehearsals) moves host copies
    wire_dev = "cpu" if on_host else device
    n_max = -(-n_total // world)
    mine = torch.empty((n_max,) + tuple(shape), dtype=dtype, device=wire_dev)
    parts = None
    if rank == root:
        if tuple(frames.shape) != (n_total,) + tuple(shape) or frames.dtype != dtype:
            raise ValueError(f"root holds {tuple(frames.shape)} {frames.dtype}, expected {(n_total,) + tuple(shape)} {dtype}")
        parts = [_padded(frames[r::world].to(wire_dev), n_max) for r in range(world)]
    _guarded("scatter(sequence)", group, lambda: dist.scatter(mine, parts, src=_global_rank(group, root), group=group))
    return mine[:share_of(n_total, rank, world)].to(device)


def gather_sequence(local, n_total=None, root=0, group=None):
    """Inverse of scatter_sequence: on `root` the frames of all ranks back in sequence order ([n_total, ...]), None
    elsewhere.  n_total defaults to world * local.shape[0] (equal shares)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if n_total is None:
        n_total = world * local.shape[0]
    if local.shape[0] != share_of(n_total, rank, world):
        raise ValueError(f"rank {rank} holds {local.shape[0]} frames of a {n_total}-frame sequence, expected {share_of(n_total, rank, world)}")
    device = local.device
    n_max = -(-n_total // world)
    local = _padded(local.cpu() if dist.get_backend(group) == "gloo" else local, n_max)
    wire = local.view(torch.uint8)   # images travel as bytes: neither RCCL nor gloo has a 16-bit integer type
    parts = [torch.empty_like(wire) for _ in range(world)] if rank == root else None
    _guarded("gather(sequence)", group, lambda: dist.gather(wire, parts, dst=_global_rank(group, root), group=group))
    if rank != root:
        return None
    full = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=device)
    for r in range(world):
        n_r = share_of(n_total, r, world)
        if n_r:
            full[r::world] = parts[r].view(local.dtype)[:n_r].to(device)
    return full


class SequenceHandle:
    """Result of SequencePipeliner.submit(): result() returns the gathered outputs ({key: [n_total, ...] tensor} on the
    root rank, {key: None} elsewhere).  Collectives are posted in program order, so every rank must call submit() /
    result() at the same points of its program."""

    def __init__(self, owner, n_total, root, keys):
        self._owner, self.n_total, self.root, self.keys = owner, n_total, root, keys
        self.local = None        # {key: this rank's share}, set when the compute has been enqueued
        self.computed = None     # event: the share's outputs are complete (device backends)
        self.gathered = None     # event: the gathered outputs are complete
        self.value = None

    def result(self, block=True):
        """block=True: returns when the outputs are in place (host-synchronous, like a future).  block=False: makes the
        CURRENT stream wait for them instead and returns at once."""
        self._owner._finish(self, block)
        return self.value


class SequencePipeliner:
    """Double-buffered batched-sequence mode over any per-rank compute function.

    submit(i+1) posts the scatter of sequence i+1, THEN the gather of sequence i (whose kernels are still running), then
    enqueues the kernels of i+1: on the copy stream the scatter of i+1 is not queued behind the wait for sequence i's
    kernels, so it runs beside them, and the gather of i runs beside the kernels of i+1.  `compute(left, right, n_total)`
    -> dict of tensors with a leading frame dimension (+ optionally "done": an event after which they are complete).
    On the gloo backend (CPU rehearsals) every transfer blocks the host, so the pipeline degenerates to the serial order
    -- with the same results, which is what the rehearsal checks."""

    def __init__(self, compute, per_frame_shape, group=None, device="cpu"):
        import torch
        self.compute, self.shape, self.group = compute, tuple(per_frame_shape), group
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.xfer = torch.cuda.Stream(self.device) if self.cuda else None
        self._pending = None     # handle whose kernels are enqueued and whose gather has not been posted yet

    def _on_xfer(self):
        import contextlib
        import torch
        return torch.cuda.stream(self.xfer) if self.cuda else contextlib.nullcontext()

    def submit(self, left, right, n_total, root=0, keys=("disparity", "planes")):
        import torch
        h = SequenceHandle(self, n_total, root, tuple(keys))
        like = (self.shape, torch.uint8, self.device)
        main = torch.cuda.current_stream(self.device) if self.cuda else None
        if self.cuda:
            self.xfer.wait_stream(main)   # the caller wrote left / right on its stream
            for t in (left, right):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(self.xfer)
        with self._on_xfer():
            l = scatter_sequence(left, n_total, like, root, self.group)
            r = scatter_sequence(right, n_total, like, root, self.group)
            scattered = self.xfer.record_event() if self.cuda else None
        prev, self._pending = self._pending, None
        if prev is not None:
            self._post_gather(prev)
        if self.cuda:
            main.wait_event(scattered)
            for t in (l, r):
                t.record_stream(main)
        out = self.compute(l, r, n_total)
        missing = [k for k in keys if k not in out]
        if missing:
            raise KeyError(f"the compute function produced no {missing}")
        h.local = {k: out[k] for k in keys}
        if self.cuda:
            h.computed = out["done"] if "done" in out else main.record_event()
        self._pending = h
        return h

    def _post_gather(self, h):
        import torch
        with self._on_xfer():
            if self.cuda:
                self.xfer.wait_event(h.computed)
            h.value = {}
            for k in h.keys:
                t = h.local[k]
                if not torch.is_tensor(t):
                    raise TypeError(f"output '{k}' is not a tensor and cannot be gathered")
                if self.cuda:
                    t.record_stream(self.xfer)
                h.value[k] = gather_sequence(t, h.n_total, h.root, self.group)
            h.gathered = self.xfer.record_event() if self.cuda else None
        h.local = None

    def _finish(self, h, block):
        import torch
        if h is self._pending:
            self._pending = None
            self._post_gather(h)
        if h.value is None:
            raise RuntimeError("this sequence's gather was never posted (handle of another pipeliner?)")
        if self.cuda and h.gathered is not None:
            if block:
                h.gathered.synchronize()
            else:
                torch.cuda.current_stream(self.device).wait_event(h.gathered)
            for v in h.value.values():
                if v is not None:
                    v.record_stream(torch.cuda.current_stream(self.device))

    def drain(self):
        """Posts the gather of the sequence that is still pending (e.g. before the process group is destroyed)."""
        if self._pending is not None:
            h, self._pending = self._pending, None
            self._post_gather(h)


class StereoPipeline:
    """device_schedule=True (default) replays the plane-parameter bookkeeping on the GPU
    (cart_plane_schedule_advance), so a step has no device->host round trip; False uses the host restatement
    (PlaneParameterSchedule + cart_find_plane_params), which is what the reference's module does per frame."""

    def __init__(self, engine, provider="histogram_peak", static_params=None, update_interval=30, reset_interval=10,
                 with_ccl=True, group=None, device_schedule=True, overlap=False, max_components=4096, keep_hists=False,
                 always_exchange=False):
        import torch
        from .engine import DevicePlaneSchedule
        self.engine = engine
        self.schedule = PlaneParameterSchedule(provider, static_params, update_interval, reset_interval)
        self.dev_schedule = DevicePlaneSchedule(engine, provider, static_params, update_interval, reset_interval) if device_schedule else None
        self.with_ccl = with_ccl
        self.keep_hists = keep_hists           # out["hists"]: a copy of this rank's per-frame 256-bin histograms (tests)
        # always_exchange: run the histogram all-gather and the sequence scatter / gather even in a world of ONE rank, so that
        # a one-GPU box can push the real collectives (backend nccl = RCCL) through the streams the product path uses
        self.always_exchange = always_exchange
        self.max_components = max_components   # rows of the per-frame component table (id, label, area, bbox)
        self.group = group
        self.world = 1
        self.rank = 0
        if group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(group)
            self.rank = torch.distributed.get_rank(group)
        self.next_id = 1
        self._hist = None
        # overlap=True ("auto" = on): the plane stages of batch i run on a side stream while the main stream already computes
        # the disparity of batch i+1 (the plane stages are short, latency-bound launches that leave the GPU mostly idle).
        # The side stream keeps the batches in order, so the plane-parameter schedule still sees the frames in id order.
        # The outputs are then produced on `self.side`: synchronise (or wait for out["done"]) before reading them on
        # another stream.  Needs engine max_inflight >= 2 * batch.  Measured (ms per 16-pair step, one stream / side stream):
        # D=128 P=8 3.20 / 3.02, D=256 P=4 3.07 / 2.98, D=128 P=4 1.80 / 1.74, 1920x1080 D=256 P=8 (4 pairs) 6.13 / 5.99,
        # D=64 P=4 1.28 / 1.24; round 4: D=128 P=8 2.99 / 2.87, D=256 P=4 2.89 / 2.69, D=64 P=4 1.13 / 1.02 (profiles/r04_overlap.txt).
        # (Two further layouts -- plane stages gated behind the next batch's aggregation; census and
        # post stages on streams of their own -- measured level or slower in round 2 and were removed: DESIGN.md appendix.)
        if overlap == "auto":
            overlap = True
        self.side = torch.cuda.Stream() if overlap else None
        self._seq = None   # SequencePipeliner, created by the first sequence call

    def process_batch(self, left, right, inputs_ready=None, n_global=None):
        """inputs_ready: a torch.cuda.Event after which `left` / `right` are complete (the upload's event, or one recorded
        when resident inputs were written).  Without it the inputs are taken to be complete on the current stream only.
        n_global: frames of the global batch when the ranks hold unequal shares (sequence mode); default n * world."""
        import torch
        main = torch.cuda.current_stream()
        if inputs_ready is not None:
            main.wait_event(inputs_ready)
        if self.side is None:
            return self._process_batch(left, right, None, n_global)
        disp = self.engine.compute_disparity(left, right) if left.shape[0] else None
        self.side.wait_stream(main)
        if disp is not None:
            disp.record_stream(self.side)
        with torch.cuda.stream(self.side):
            out = self._process_batch(left, right, disp, n_global)
            out["done"] = self.side.record_event()
        return out

    # ---- batched-sequence mode -------------------------------------------------------------------------------------
    def _sequencer(self, channels):
        import torch
        e = self.engine
        shape = (e.height, e.width) if channels == 1 else (e.height, e.width, 3)
        if self._seq is None or self._seq.shape != shape:
            dev = torch.device("cuda", torch.cuda.current_device())
            self._seq = SequencePipeliner(lambda l, r, n_total: self.process_batch(l, r, n_global=n_total), shape, self.group, dev)
        return self._seq

    def submit_sequence(self, left, right, n_total, root=0, channels=1, keys=("disparity", "planes")):
        """Pipelined batched-sequence mode: a sequence that lives on `root` ([n_total, H, W] gray or, with channels=3,
        [n_total, H, W, 3] BGR; None elsewhere) is dealt out, this rank's share is enqueued as one batch, and a
        SequenceHandle is returned at once; the gather of the PREVIOUS sequence is posted by this call, behind this
        sequence's scatter (SequencePipeliner).  Frame ids continue from the previous call like process_batch's.  Needs
        torch.distributed (any world size, one rank included) -- process_sequence takes the no-exchange shortcut itself."""
        return self._sequencer(channels).submit(left, right, n_total, root, keys)

    def process_sequence(self, left, right, n_total, root=0, channels=1, keys=("disparity", "planes")):
        """Unpipelined form: submit_sequence(...).result().  With world == 1 (and no always_exchange) this is process_batch."""
        import torch
        if self.world == 1 and not self.always_exchange:
            out = self.process_batch(left, right)
            torch.cuda.current_stream().wait_event(out["done"]) if "done" in out else None
            return {k: out[k] for k in keys}
        return self.submit_sequence(left, right, n_total, root, channels, keys).result(block=False)

    def drain(self):
        if self._seq is not None:
            self._seq.drain()

    # ---- one rank's share of a global batch ---------------------------------------------------------------------------
    def _gather_hists(self, n, n_global):
        """All-gather of the per-frame histograms (the path's only exchange step) into frame-id order: [n_global, 256]
        (device tensor; on gloo via host copies).  Ranks with a short share pad theirs with zero rows up to the longest."""
        import torch
        dist = torch.distributed
        n_max = -(-n_global // self.world)
        src = self._hist
        if n < n_max:
            src = torch.cat([src, src.new_zeros((n_max - n, 256))])
        on_host = dist.get_backend(self.group) == "gloo"   # RCCL gathers device tensors; gloo (CPU rehearsals) host copies
        src = src.cpu() if on_host else src.contiguous()
        allh = torch.empty((self.world * n_max, 256), dtype=torch.int32, device=src.device)
        _guarded("all_gather_into_tensor(histograms)", self.group, lambda: dist.all_gather_into_tensor(allh, src, group=self.group))
        # [rank][j] -> id order j*world + rank; padded rows are exactly the ids >= n_global
        return allh.view(self.world, n_max, 256).permute(1, 0, 2).reshape(n_max * self.world, 256)[:n_global].contiguous().to(self._hist.device)

    def _process_batch(self, left, right, disp=None, n_global=None):
        """left/right: uint8 [n,h,w(,3)] on the GPU: this rank's n frames of a global batch of n_global frames (default
        n*world; interleaved ids, frame k of the batch on rank k mod world).  -> dict(disparity, planes, ids, n_components, params)."""
        import torch
        eng = self.engine
        n = left.shape[0]
        if n_global is None:
            n_global = n * self.world
        if n != share_of(n_global, self.rank, self.world):
            raise ValueError(f"rank {self.rank} holds {n} frames of a global batch of {n_global}")
        dev = left.device
        exchange = self.world > 1 or self.always_exchange
        if self._hist is None or self._hist.shape[0] != n:
            self._hist = torch.empty((n, 256), dtype=torch.int32, device=dev)
        self._hist.zero_()
        if n:
            if disp is None:
                disp = eng.compute_disparity(left, right)
            deriv = eng.plane_derivative_hist(disp, self._hist, per_frame_hist=True)
        else:   # fewer frames than ranks: this rank only joins the exchange
            disp = torch.empty((0, eng.height, eng.width), dtype=torch.int16, device=dev)
            deriv = disp
        kept = self._hist.clone() if self.keep_hists else None
        static = self.schedule.provider == "static"
        if self.dev_schedule is not None:
            if exchange and not static:
                hists = self._gather_hists(n, n_global)
            elif exchange:
                hists = self._hist.new_zeros((n_global, 256))
            else:
                hists = self._hist
            allp = self.dev_schedule.advance(self.next_id, hists)
            mine = allp[self.rank::self.world].contiguous() if self.world > 1 else allp
            planes = eng.plane_classify_dev(deriv, mine) if n else torch.empty((0, eng.height, eng.width), dtype=torch.uint8, device=dev)
            params = mine
        else:
            if static:
                per_frame = [self.schedule.params] * n
            else:
                hists = (self._gather_hists(n, n_global) if exchange else self._hist).cpu().numpy()
                allp = self.schedule.advance(self.next_id, hists)
                per_frame = allp[self.rank::self.world]
            planes = eng.plane_classify(deriv, list(per_frame) if n > 1 else per_frame[0]) if n else torch.empty((0, eng.height, eng.width), dtype=torch.uint8, device=dev)
            params = per_frame
        self.next_id += n_global
        out = dict(disparity=disp, planes=planes, params=params)
        if kept is not None:
            out["hists"] = kept
        if self.with_ccl and n:
            out["ids"], out["components"], out["n_components"] = eng.plane_ccl_table(planes, self.max_components)
        elif self.with_ccl:   # a rank without frames still hands every key to the gather
            out["ids"] = torch.empty((0, eng.height, eng.width), dtype=torch.int32, device=dev)
            out["n_components"] = torch.empty((0,), dtype=torch.int32, device=dev)
            out["components"] = torch.empty((0, self.max_components, 7), dtype=torch.int32, device=dev)
        return out
