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
histograms (1 KB per frame); every rank then replays the schedule redundantly on the host.
"""
import numpy as np

from .engine import PlaneParams, find_plane_params


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


def scatter_sequence(frames, n_total, like, root=0, group=None):
    """Batched-sequence mode (BASELINE.json configs[4]: a 64-frame sequence that starts on one rank): frame k of the
    sequence goes to rank k mod world, the interleaving of shard_ids.  `frames` is the [n_total, ...] tensor on `root`
    (ignored elsewhere); `like` = (per-frame shape, dtype, device) so that the other ranks can post their receive.
    Returns this rank's [n_total / world, ...] tensor.  One scatter (RCCL: grouped point-to-point over xGMI, rank 0 has
    a direct link to every peer) -- not a ring collective."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if n_total % world:
        raise ValueError(f"sequence length {n_total} is not a multiple of the world size {world}")
    shape, dtype, device = like
    on_host = dist.get_backend(group) == "gloo"   # RCCL moves device tensors; gloo (CPU rehearsals) moves host copies
    mine = torch.empty((n_total // world,) + tuple(shape), dtype=dtype, device="cpu" if on_host else device)
    parts = None
    if rank == root:
        if tuple(frames.shape) != (n_total,) + tuple(shape) or frames.dtype != dtype:
            raise ValueError(f"root holds {tuple(frames.shape)} {frames.dtype}, expected {(n_total,) + tuple(shape)} {dtype}")
        parts = [(frames[r::world].cpu() if on_host else frames[r::world]).contiguous() for r in range(world)]
    dist.scatter(mine, parts, src=dist.get_global_rank(group, root) if group is not None else root, group=group)
    return mine.to(device)


def gather_sequence(local, root=0, group=None):
    """Inverse of scatter_sequence: on `root` the frames of all ranks back in sequence order ([n_total, ...]), None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    device = local.device
    local = (local.cpu() if dist.get_backend(group) == "gloo" else local).contiguous()
    wire = local.view(torch.uint8)   # images travel as bytes: neither RCCL nor gloo has a 16-bit integer type
    parts = [torch.empty_like(wire) for _ in range(world)] if rank == root else None
    dist.gather(wire, parts, dst=dist.get_global_rank(group, root) if group is not None else root, group=group)
    if rank != root:
        return None
    full = torch.empty((local.shape[0] * world,) + tuple(local.shape[1:]), dtype=local.dtype, device=device)
    for r in range(world):
        full[r::world] = parts[r].view(local.dtype).to(device)
    return full


class StereoPipeline:
    """device_schedule=True (default) replays the plane-parameter bookkeeping on the GPU
    (cart_plane_schedule_advance), so a step has no device->host round trip; False uses the host restatement
    (PlaneParameterSchedule + cart_find_plane_params), which is what the reference's module does per frame."""

    def __init__(self, engine, provider="histogram_peak", static_params=None, update_interval=30, reset_interval=10,
                 with_ccl=True, group=None, device_schedule=True, overlap=False, max_components=4096, keep_hists=False,
                 always_exchange=False, split_stages=False):
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
        # overlap=True: the plane stages of batch i run on a side stream while the main stream already computes the
        # disparity of batch i+1 (the plane stages are short, latency-bound launches that leave the GPU mostly idle).
        # The side stream keeps the batches in order, so the plane-parameter schedule still sees the frames in id
        # order.  The outputs are then produced on `self.side` ("disparity" too when split_stages is on): synchronise (or
        # wait for out["done"]) before reading them on another stream.  Needs engine max_inflight >= 2 * batch.
        # overlap="auto" = on.  Measured (profiles/tools/r02_deferred.sh, 16 pairs per step, ms per step one stream / side stream /
        # deferred, after the residency cap of the aggregation launch): D=128 P=8 3.20 / 3.02 / 3.10, D=256 P=4 3.07 / 2.98 /
        # 3.08, D=128 P=4 1.80 / 1.74 / 1.76, 1920x1080 D=256 P=8 (4 pairs) 6.13 / 5.99 / 6.00, D=64 P=4 1.28 / 1.24 / 1.20.
        # (Before that cap the 4-path engines lost 2-8 % to the side stream: their aggregation launch is bound by the W-step
        # chain of the horizontal scans, which the side stream's kernels slow down.)
        # overlap="deferred": the plane stages of batch i are enqueued by the NEXT process_batch call (or flush()), on the
        # side stream, gated behind the aggregation of batch i+1 (cart_compute_disparity_batch_gated): they then run beside
        # the HBM-bound WTA of batch i+1 instead of beside its aggregation.  process_batch returns the outputs of the
        # PREVIOUS batch (None on the first call); flush() returns those of the last one.  Never chosen by "auto" (it changes
        # what process_batch returns); worth 3 % more than the plain side stream where the aggregation is latency-bound (D=64 / 4 paths: 13.4 k pairs/s).
        if overlap == "auto":
            overlap = True
        self.deferred = overlap == "deferred"
        self._pending = None
        self.side = torch.cuda.Stream() if overlap else None
        # split_stages (with a side stream, not deferred): the main stream carries only aggregation + WTA; the stages after the
        # WTA join the plane stages on the side stream, and -- when the caller says where the inputs are complete
        # (process_batch(..., inputs_ready=event)) -- the census of batch i+1 runs on a third stream beside the WTA of batch i
        # (cart_compute_disparity_batch_streams).  Off by default: measured (profiles/tools/r02_split.sh, three A/B pairs per
        # configuration) it gains 1-4 % at D=64 / 4 paths and loses 1-3 % at D=128 / 8 paths, D=256 / 4 paths and 1920x1080 --
        # the short kernels cost the two long launches more beside them than they cost in front of them.
        self.split_stages = bool(overlap) and not self.deferred and split_stages
        self.pre = torch.cuda.Stream() if self.split_stages else None

    def process_batch(self, left, right, inputs_ready=None):
        """inputs_ready: a torch.cuda.Event after which `left` / `right` are complete (the upload's event, or one recorded
        when resident inputs were written).  Without it the inputs are taken to be complete on the current stream only."""
        if self.side is None:
            return self._process_batch(left, right)
        import torch
        main = torch.cuda.current_stream()
        if self.split_stages:
            census_stream = None
            if inputs_ready is not None:
                self.pre.wait_event(inputs_ready)
                left.record_stream(self.pre); right.record_stream(self.pre)
                census_stream = self.pre
            disp = self.engine.compute_disparity(left, right, census_stream=census_stream, tail_stream=self.side)
            disp.record_stream(self.side)
            with torch.cuda.stream(self.side):   # post + interpolate of this batch are already queued there
                out = self._process_batch(left, right, disp)
                out["done"] = self.side.record_event()
            return out
        if self.deferred:
            disp = self.engine.compute_disparity(left, right, gated_stream=self.side)
            out = self._finish_pending()
            self._pending = (left, right, disp, main.record_event())
            return out
        disp = self.engine.compute_disparity(left, right)
        self.side.wait_stream(main)
        disp.record_stream(self.side)
        with torch.cuda.stream(self.side):
            out = self._process_batch(left, right, disp)
            out["done"] = self.side.record_event()
        return out

    def _finish_pending(self):
        if self._pending is None:
            return None
        left, right, disp, ready = self._pending
        self._pending = None
        self.side.wait_event(ready)   # the batch's disparity (long past when a later batch's gate is already in the queue)
        for t in (left, right, disp):
            t.record_stream(self.side)
        import torch
        with torch.cuda.stream(self.side):
            out = self._process_batch(left, right, disp)
            out["done"] = self.side.record_event()
        return out

    def flush(self):
        """overlap="deferred": enqueue the plane stages of the batch that is still pending and return its outputs
        (None when nothing is pending or in the other modes, whose process_batch has already returned everything)."""
        return self._finish_pending() if self.deferred else None

    def process_sequence(self, left, right, n_total, root=0, channels=1, keys=("disparity", "planes")):
        """A sequence that lives on `root` ([n_total, H, W] gray or, with channels=3, [n_total, H, W, 3] BGR; None elsewhere): scatter the frames, run this rank's
        share as one batch, gather the named outputs back on `root` in sequence order (None on the other ranks).  With
        world == 1 this is process_batch.  Frame ids continue from the previous call like process_batch's."""
        import torch
        if self.deferred and self._pending is not None:
            raise ValueError("process_sequence returns the outputs of the frames it is given: flush() the pending batch first")
        if self.world == 1 and not self.always_exchange:
            out = self._batch_now(left, right)
            torch.cuda.current_stream().wait_event(out["done"]) if "done" in out else None
            return {k: out[k] for k in keys}
        e = self.engine
        dev = torch.device("cuda", torch.cuda.current_device())
        per_frame = (e.height, e.width) if channels == 1 else (e.height, e.width, 3)
        l = scatter_sequence(left, n_total, (per_frame, torch.uint8, dev), root, self.group)
        r = scatter_sequence(right, n_total, (per_frame, torch.uint8, dev), root, self.group)
        out = self._batch_now(l, r)
        if "done" in out:
            torch.cuda.current_stream().wait_event(out["done"])
        return {k: gather_sequence(out[k], root, self.group) for k in keys}

    def _batch_now(self, left, right):
        """process_batch + (deferred mode) flush: the outputs of exactly these frames."""
        out = self.process_batch(left, right)
        return self.flush() if self.deferred else out

    def _process_batch(self, left, right, disp=None):
        """left/right: uint8 [n,h,w(,3)] on the GPU: this rank's n frames of a global batch of
        n*world frames (interleaved ids).  -> dict(disparity, planes, ids, n_components, params)."""
        import torch
        eng = self.engine
        n = left.shape[0]
        if disp is None:
            disp = eng.compute_disparity(left, right)
        if self._hist is None or self._hist.shape[0] != n:
            self._hist = torch.empty((n, 256), dtype=torch.int32, device=left.device)
        self._hist.zero_()
        deriv = eng.plane_derivative_hist(disp, self._hist, per_frame_hist=True)
        kept = self._hist.clone() if self.keep_hists else None
        if self.dev_schedule is not None:
            hists = self._hist
            if (self.world > 1 or self.always_exchange) and self.schedule.provider != "static":
                on_host = torch.distributed.get_backend(self.group) == "gloo"
                src = self._hist.cpu() if on_host else self._hist
                allh = torch.empty((self.world * n, 256), dtype=torch.int32, device=src.device)
                torch.distributed.all_gather_into_tensor(allh, src, group=self.group)
                hists = allh.view(self.world, n, 256).permute(1, 0, 2).reshape(n * self.world, 256).contiguous().to(left.device)
            if self.world > 1 and self.schedule.provider == "static":
                hists = self._hist.new_zeros((n * self.world, 256))
            allp = self.dev_schedule.advance(self.next_id, hists)
            mine = allp[self.rank::self.world].contiguous() if self.world > 1 else allp
            self.next_id += n * self.world
            planes = eng.plane_classify_dev(deriv, mine)
            out = dict(disparity=disp, planes=planes, params=mine)
            if kept is not None:
                out["hists"] = kept
            if self.with_ccl:
                out["ids"], out["n_components"] = eng.plane_ccl(planes)
                out["components"], _ = eng.plane_ccl_stats(planes, out["ids"], self.max_components)
            return out
        if self.schedule.provider == "static":
            per_frame = [self.schedule.params] * n
        else:
            if self.world > 1 or self.always_exchange:
                # RCCL gathers device tensors; the gloo backend (CPU rehearsals) gathers host copies
                on_host = torch.distributed.get_backend(self.group) == "gloo"
                src = self._hist.cpu() if on_host else self._hist
                allh = torch.empty((self.world * n, 256), dtype=torch.int32, device=src.device)
                torch.distributed.all_gather_into_tensor(allh, src, group=self.group)
                # [rank][k] -> id order k*world + rank
                hists = allh.view(self.world, n, 256).permute(1, 0, 2).reshape(n * self.world, 256).cpu().numpy()
            else:
                hists = self._hist.cpu().numpy()
            allp = self.schedule.advance(self.next_id, hists)
            per_frame = allp[self.rank::self.world]
        self.next_id += n * self.world
        planes = eng.plane_classify(deriv, list(per_frame) if n > 1 else per_frame[0])
        out = dict(disparity=disp, planes=planes, params=per_frame)
        if kept is not None:
            out["hists"] = kept
        if self.with_ccl:
            out["ids"], out["n_components"] = eng.plane_ccl(planes)
            out["components"], _ = eng.plane_ccl_stats(planes, out["ids"], self.max_components)
        return out
