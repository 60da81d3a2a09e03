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


class StereoPipeline:
    """device_schedule=True (default) replays the plane-parameter bookkeeping on the GPU
    (cart_plane_schedule_advance), so a step has no device->host round trip; False uses the host restatement
    (PlaneParameterSchedule + cart_find_plane_params), which is what the reference's module does per frame."""

    def __init__(self, engine, provider="histogram_peak", static_params=None, update_interval=30, reset_interval=10,
                 with_ccl=True, group=None, device_schedule=True, overlap=False, max_components=4096):
        import torch
        from .engine import DevicePlaneSchedule
        self.engine = engine
        self.schedule = PlaneParameterSchedule(provider, static_params, update_interval, reset_interval)
        self.dev_schedule = DevicePlaneSchedule(engine, provider, static_params, update_interval, reset_interval) if device_schedule else None
        self.with_ccl = with_ccl
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
        # order.  Outputs other than "disparity" are then produced on `self.side`: synchronise (or wait for
        # out["done"]) before reading them on another stream.  Needs engine max_inflight >= 2 * batch.
        self.side = torch.cuda.Stream() if overlap else None

    def process_batch(self, left, right):
        if self.side is None:
            return self._process_batch(left, right)
        import torch
        main = torch.cuda.current_stream()
        disp = self.engine.compute_disparity(left, right)
        self.side.wait_stream(main)
        disp.record_stream(self.side)
        with torch.cuda.stream(self.side):
            out = self._process_batch(left, right, disp)
            out["done"] = self.side.record_event()
        return out

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
        if self.dev_schedule is not None:
            hists = self._hist
            if self.world > 1 and self.schedule.provider != "static":
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
            if self.with_ccl:
                out["ids"], out["n_components"] = eng.plane_ccl(planes)
                out["components"], _ = eng.plane_ccl_stats(planes, out["ids"], self.max_components)
            return out
        if self.schedule.provider == "static":
            per_frame = [self.schedule.params] * n
        else:
            if self.world > 1:
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
        if self.with_ccl:
            out["ids"], out["n_components"] = eng.plane_ccl(planes)
            out["components"], _ = eng.plane_ccl_stats(planes, out["ids"], self.max_components)
        return out
