"""Thin Python driver over the C ABI (include/cart_engine.h).

torch is used only for device memory and streams; every op below is one C-ABI call on the
current torch stream.  Tensors are [n, h, w(, c)] or [h, w(, c)] CUDA tensors with a contiguous
innermost row; row and frame pitches are taken from the strides (like cv::cuda::GpuMat::step).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import EngineParams, PlaneParams, SuperpixelParams

INVALID = -32768  # CARTSLAM_DISPARITY_INVALID, reference include/modules/disparity.hpp:17


class EngineError(RuntimeError):
    pass


def _stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _geom(t, inner):
    """-> (n_frames, data_ptr, step_bytes, frame_stride_bytes); `inner` = trailing dims that form a row."""
    if not t.is_cuda:
        raise EngineError("tensor must live on the GPU")
    batched = t.dim() == inner + 2
    if t.dim() not in (inner + 1, inner + 2):
        raise EngineError(f"expected {inner + 1} or {inner + 2} dims, got {t.dim()}")
    row_dims = t.shape[-inner:]
    exp = 1
    for k in range(inner):
        if t.stride(-1 - k) != exp:
            raise EngineError("innermost row must be contiguous")
        exp *= row_dims[-1 - k]
    es = t.element_size()
    step = t.stride(-inner - 1) * es
    n = t.shape[0] if batched else 1
    fs = t.stride(0) * es if batched else 0
    return n, C.c_void_p(t.data_ptr()), step, fs


class Engine:
    """One engine = one (width, height, D, paths, ...) configuration on one GPU.
    num_disparities=0, paths=0 gives a geometry-only engine for the post-SGM entry points."""

    def __init__(self, width, height, num_disparities=256, paths=4, min_disparity=4, p1=10, p2=120,
                 uniqueness_ratio=12, smoothing_radius=-1, smoothing_iterations=5, max_inflight=12, device_id=0):
        self._lib = _lib.load()
        p = EngineParams()
        self._lib.cart_engine_default_params(C.byref(p))
        p.device_id, p.width, p.height = device_id, width, height
        p.min_disparity, p.num_disparities, p.paths, p.p1, p.p2 = min_disparity, num_disparities, paths, p1, p2
        p.uniqueness_ratio, p.smoothing_radius, p.smoothing_iterations = uniqueness_ratio, smoothing_radius, smoothing_iterations
        p.max_inflight = max_inflight
        self.params = p
        self._h = C.c_void_p()
        if self._lib.cart_engine_create(C.byref(p), C.byref(self._h)) != 0:
            raise EngineError("cart_engine_create: " + self._lib.cart_last_error(None).decode())
        self.width, self.height, self.D, self.P = width, height, num_disparities, paths

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.cart_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise EngineError(f"{what}: " + self._lib.cart_last_error(self._h).decode())

    # ---- launch plan (every plan gives the same bits; include/cart_engine.h CART_PLAN_*) ----
    def set_plan(self, plan, min_frames=1):
        """plan: "auto" | "slabs" | "fused_up"; a forced plan applies to launches of >= min_frames frames."""
        code = {"auto": _lib.PLAN_AUTO, "slabs": _lib.PLAN_SLABS, "fused_up": _lib.PLAN_FUSED_UP}[plan]
        self._check(self._lib.cart_engine_set_option(self._h, _lib.OPT_PLAN, code), "cart_engine_set_option")
        self._check(self._lib.cart_engine_set_option(self._h, _lib.OPT_PLAN_MIN_FRAMES, int(min_frames)), "cart_engine_set_option")

    def set_spec_variants(self, s8_zero_invalid=False, s7_replicate_border=False, s5_top2=False):
        """The three choices that are open upstream (oracle S8 / S7 / S5 NOTEs); default: the oracle's spec."""
        self._check(self._lib.cart_engine_set_option(self._h, _lib.OPT_SPEC_S5_TOP2, 1 if s5_top2 else 0), "cart_engine_set_option")
        self._check(self._lib.cart_engine_set_option(self._h, _lib.OPT_SPEC_S8_ZERO_INVALID, 1 if s8_zero_invalid else 0), "cart_engine_set_option")
        self._check(self._lib.cart_engine_set_option(self._h, _lib.OPT_SPEC_S7_REPLICATE_BORDER, 1 if s7_replicate_border else 0), "cart_engine_set_option")

    def set_chunk_frames(self, n):
        self._check(self._lib.cart_engine_set_option(self._h, _lib.OPT_CHUNK_FRAMES, int(n)), "cart_engine_set_option")

    def describe_plan(self, n_frames):
        """-> dict(frames_per_launch, plan, slabs_written) of a batched call of n_frames."""
        lp = _lib.LaunchPlan()
        self._check(self._lib.cart_engine_describe_plan(self._h, int(n_frames), C.byref(lp)), "cart_engine_describe_plan")
        return {"frames_per_launch": lp.frames_per_launch, "plan": {0: "slabs", 1: "fused_up"}[lp.plan],
                "slabs_written": lp.slabs_written}

    def copy_narrow(self, dst, src, workgroups=0):
        """dst <- src (same byte size, contiguous; dst may be a PINNED host tensor: its memory is mapped into the device's
        address space) by a copy kernel of a few workgroups on the current stream (cart_copy_narrow)."""
        nbytes = src.numel() * src.element_size()
        if dst.numel() * dst.element_size() != nbytes or not dst.is_contiguous() or not src.is_contiguous():
            raise EngineError("copy_narrow needs two contiguous tensors of the same byte size")
        if not (dst.is_cuda or dst.is_pinned()) or not (src.is_cuda or src.is_pinned()):
            raise EngineError("copy_narrow needs device tensors or pinned host tensors")
        self._check(self._lib.cart_copy_narrow(self._h, C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), nbytes, int(workgroups),
                                               _stream_ptr()), "cart_copy_narrow")
        return dst

    def tune_placement(self, n_frames, max_tries=8, max_extra_bytes=0, report=False):
        """cart_engine_tune_placement (opt-in set-up step): time the slab-bound launches of an n_frames call on up to max_tries physical
        placements of the slot groups behind it and keep the fastest.  max_extra_bytes bounds what the call may hold beyond the
        workspace while it searches (0: two units' worth; None: no cap but 4 GiB left free).  -> (ms before, ms after), or with
        report=True the whole cart_placement_report as a dict (mode: fast / mixed / uniform / unknown, relative to the sets the search saw; candidates timed, why the search
        stopped).  The engine must be idle."""
        r = _lib.PlacementReport()
        cap = C.c_size_t(-1).value if max_extra_bytes is None else int(max_extra_bytes)
        self._check(self._lib.cart_engine_tune_placement(self._h, int(n_frames), int(max_tries), cap, C.byref(r)), "cart_engine_tune_placement")
        if not report:
            return r.ms_first, r.ms_kept
        return {"ms_first": r.ms_first, "ms_kept": r.ms_kept, "ms_fastest_seen": r.ms_fastest_seen, "ms_slowest_seen": r.ms_slowest_seen,
                "seconds": r.seconds, "units": r.units, "candidates": r.candidates, "mode": _lib.PLACE_MODES.get(r.mode, str(r.mode)),
                "stopped_on": _lib.PLACE_STOPS.get(r.stop_reason, str(r.stop_reason))}

    # ---- disparity module (reference src/modules/disparity/disparity.cu:49-80) ----
    def compute_disparity(self, left, right, out=None):
        import torch
        ch = 3 if (left.dim() >= 3 and left.shape[-1] == 3 and left.shape[-2] == self.width
                   and left.shape[-3] == self.height) else 1
        inner = 2 if ch == 3 else 1
        for t in (left, right):
            if tuple(t.shape[-inner - 1:][:2]) != (self.height, self.width):
                raise EngineError(f"image shape {tuple(t.shape)} does not match the engine's {self.height}x{self.width} (bad step/size)")
        n, lp, ls, lfs = _geom(left, inner)
        n2, rp, rs, rfs = _geom(right, inner)
        if n != n2 or left.dtype != torch.uint8 or right.dtype != torch.uint8:
            raise EngineError("left/right must be uint8 tensors of the same batch size")
        shape = (n, self.height, self.width) if left.dim() == inner + 2 else (self.height, self.width)
        if out is None:
            out = torch.empty(shape, dtype=torch.int16, device=left.device)
        _, op, os_, ofs = _geom(out, 1)
        self._check(self._lib.cart_compute_disparity_batch(self._h, n, lp, ls, lfs, rp, rs, rfs, ch, op, os_, ofs,
                                                           _stream_ptr()), "cart_compute_disparity_batch")
        return out

    def compute_disparity_multi(self, lefts, rights, outs=None):
        """Frames in separate allocations (lists of [H,W] or [H,W,3] uint8 tensors with a common row step) in ONE launch
        sequence: what a module adapter uses to coalesce concurrently entered frames."""
        import torch
        n = len(lefts)
        if n == 0 or len(rights) != n:
            raise EngineError("lefts/rights must be non-empty lists of the same length")
        ch = 3 if lefts[0].dim() == 3 else 1
        if outs is None:
            outs = [torch.empty((self.height, self.width), dtype=torch.int16, device=lefts[0].device) for _ in range(n)]
        geo = [[_geom(t, inner) for t in ts] for ts, inner in ((lefts, 2 if ch == 3 else 1), (rights, 2 if ch == 3 else 1), (outs, 1))]
        for ts, g in zip((lefts, rights, outs), geo):
            if any(tuple(t.shape[:2]) != (self.height, self.width) for t in ts) or len({x[2] for x in g}) != 1:
                raise EngineError("every image of one kind must be HxW with the same row step")
        tables = [(C.c_void_p * n)(*[x[1].value for x in g]) for g in geo]
        self._check(self._lib.cart_compute_disparity_multi(self._h, n, tables[0], geo[0][0][2], tables[1], geo[1][0][2], ch,
                                                           tables[2], geo[2][0][2], _stream_ptr()), "cart_compute_disparity_multi")
        return outs

    def interpolate(self, disp, radius, iterations, min_disp16, max_disp):
        n, p, s, fs = _geom(disp, 1)
        self._check(self._lib.cart_interpolate(self._h, n, p, s, fs, radius, iterations, min_disp16, max_disp,
                                               _stream_ptr()), "cart_interpolate")
        return disp

    # ---- derivative module (reference src/modules/disparity/derivative.cu:151-184) ----
    def disparity_derivative(self, disp):
        import torch
        n, p, s, fs = _geom(disp, 1)
        out = torch.empty(tuple(disp.shape) + (2,), dtype=torch.int16, device=disp.device)
        hist = torch.empty((n, 256, 2), dtype=torch.int32, device=disp.device)
        _, op, os_, ofs = _geom(out, 2)
        self._check(self._lib.cart_disparity_derivative(self._h, n, p, s, fs, op, os_, ofs, C.c_void_p(hist.data_ptr()),
                                                        _stream_ptr()), "cart_disparity_derivative")
        return out, (hist if disp.dim() == 3 else hist[0])

    # ---- plane label module (reference src/modules/planeseg/planeseg.cu:246-377) ----
    def plane_derivative_hist(self, disp, hist, per_frame_hist=False):
        """hist: int32 [256] (persistent, added to) or [n,256] when per_frame_hist."""
        import torch
        n, p, s, fs = _geom(disp, 1)
        out = torch.empty_like(disp)
        _, op, os_, ofs = _geom(out, 1)
        if hist.dtype != torch.int32 or not hist.is_contiguous():
            raise EngineError("hist must be a contiguous int32 tensor")
        self._check(self._lib.cart_plane_derivative_hist(self._h, n, p, s, fs, op, os_, ofs, C.c_void_p(hist.data_ptr()),
                                                         256 if per_frame_hist else 0, _stream_ptr()),
                    "cart_plane_derivative_hist")
        return out

    def plane_classify(self, deriv, params):
        """params: one PlaneParams / 6-tuple, or a list with one per frame."""
        import torch
        n, p, s, fs = _geom(deriv, 1)
        per_frame = isinstance(params, (list,)) and len(params) == n and n > 1
        plist = params if isinstance(params, list) else [params]
        arr = (PlaneParams * len(plist))()
        for i, q in enumerate(plist):
            arr[i] = q if isinstance(q, PlaneParams) else PlaneParams(*q)
        planes = torch.empty(deriv.shape, dtype=torch.uint8, device=deriv.device)
        _, pp, ps, pfs = _geom(planes, 1)
        self._check(self._lib.cart_plane_classify(self._h, n, p, s, fs, arr, 1 if per_frame else 0, pp, ps, pfs,
                                                  _stream_ptr()), "cart_plane_classify")
        return planes

    def plane_label_multi(self, disps, hist, params):
        """Frames in separate allocations (a list of [H,W] int16 disparity tensors with a common row step): plane
        derivative + cumulative histogram (`hist`: int32 [256], added to) and classification with `params` (one
        PlaneParams / 6-tuple for all frames, or a list with one per frame), one launch per stage.
        Returns (list of derivative images, list of plane images)."""
        import torch
        n = len(disps)
        if n == 0:
            raise EngineError("disps must be a non-empty list")
        geo = [_geom(t, 1) for t in disps]
        if any(tuple(t.shape) != (self.height, self.width) or t.dtype != torch.int16 for t in disps) or len({g[2] for g in geo}) != 1:
            raise EngineError("every disparity image must be int16 HxW with the same row step")
        if hist.dtype != torch.int32 or not hist.is_contiguous():
            raise EngineError("hist must be a contiguous int32 tensor")
        derivs = [torch.empty((self.height, self.width), dtype=torch.int16, device=disps[0].device) for _ in range(n)]
        planes = [torch.empty((self.height, self.width), dtype=torch.uint8, device=disps[0].device) for _ in range(n)]
        table = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        self._check(self._lib.cart_plane_derivative_hist_multi(self._h, n, table(disps), geo[0][2], table(derivs), self.width * 2,
                                                               C.c_void_p(hist.data_ptr()), 0, _stream_ptr()), "cart_plane_derivative_hist_multi")
        plist = params if isinstance(params, list) else [params]
        arr = (PlaneParams * len(plist))()
        for i, q in enumerate(plist):
            arr[i] = q if isinstance(q, PlaneParams) else PlaneParams(*q)
        self._check(self._lib.cart_plane_classify_multi(self._h, n, table(derivs), self.width * 2, arr, 1 if len(plist) == n and n > 1 else 0,
                                                        table(planes), self.width, _stream_ptr()), "cart_plane_classify_multi")
        return derivs, planes

    def plane_classify_dev(self, deriv, params_dev):
        """params_dev: int32 CUDA tensor [n,6] (one cart_plane_params per frame) or [6] (shared)."""
        import torch
        n, p, s, fs = _geom(deriv, 1)
        if params_dev.dtype != torch.int32 or not params_dev.is_contiguous() or not params_dev.is_cuda:
            raise EngineError("params_dev must be a contiguous int32 CUDA tensor")
        per_frame = params_dev.dim() == 2
        if per_frame and params_dev.shape[0] != n:
            raise EngineError("one parameter row per frame expected")
        planes = torch.empty(deriv.shape, dtype=torch.uint8, device=deriv.device)
        _, pp, ps, pfs = _geom(planes, 1)
        self._check(self._lib.cart_plane_classify_dev(self._h, n, p, s, fs, C.c_void_p(params_dev.data_ptr()), 1 if per_frame else 0,
                                                      pp, ps, pfs, _stream_ptr()), "cart_plane_classify_dev")
        return planes

    def plane_temporal_vote(self, planes, prev_planes, flows):
        """planes: uint8 [h,w]; prev_planes: list of uint8 [h,w]; flows: list of int16 [h,w,2] (S10.5) -- planeseg.cu:199-240."""
        import torch
        n = len(prev_planes)
        if len(flows) != n:
            raise EngineError("one flow per previous plane image")
        _, p, s, _ = _geom(planes, 1)
        out = torch.empty_like(planes)
        _, op, os_, _ = _geom(out, 1)
        P = (C.c_void_p * max(n, 1))(); PS = (C.c_size_t * max(n, 1))(); F = (C.c_void_p * max(n, 1))(); FS = (C.c_size_t * max(n, 1))()
        for k in range(n):
            _, pp, ps, _ = _geom(prev_planes[k], 1)
            _, fp, fs, _ = _geom(flows[k], 2)
            P[k], PS[k], F[k], FS[k] = pp.value, ps, fp.value, fs
        self._check(self._lib.cart_plane_temporal_vote(self._h, p, s, n, P, PS, F, FS, op, os_, _stream_ptr()), "cart_plane_temporal_vote")
        return out

    # ---- superpixel plane labelling (reference src/modules/planeseg/sp_planeseg.cu:27-178) ----
    def superpixel_plane_classify(self, deriv2, labels, max_label, params, prev_planes=(), flows=()):
        """deriv2: int16 [h,w,2]; labels: uint16 [h,w] (torch has no uint16 arithmetic: pass an int16 view);
        -> (planes_unsmoothed, planes) uint8 [h,w]."""
        import torch
        n = len(prev_planes)
        if len(flows) != n:
            raise EngineError("one flow per previous plane image")
        _, dp, ds, _ = _geom(deriv2, 2)
        _, lp, ls, _ = _geom(labels, 1)
        if labels.element_size() != 2:
            raise EngineError("labels must be a 16-bit tensor")
        pp = params if isinstance(params, PlaneParams) else PlaneParams(*params)
        uns = torch.empty(labels.shape, dtype=torch.uint8, device=labels.device)
        out = torch.empty_like(uns)
        _, up, us, _ = _geom(uns, 1)
        _, op, os_, _ = _geom(out, 1)
        P = (C.c_void_p * max(n, 1))(); PS = (C.c_size_t * max(n, 1))(); F = (C.c_void_p * max(n, 1))(); FS = (C.c_size_t * max(n, 1))()
        for k in range(n):
            _, q, qs, _ = _geom(prev_planes[k], 1)
            _, fp, fs, _ = _geom(flows[k], 2)
            P[k], PS[k], F[k], FS[k] = q.value, qs, fp.value, fs
        self._check(self._lib.cart_superpixel_plane_classify(self._h, dp, ds, lp, ls, int(max_label), C.byref(pp), n, P, PS, F, FS,
                                                             up, us, op, os_, _stream_ptr()), "cart_superpixel_plane_classify")
        return uns, out

    # ---- optical flow (stand-in for src/modules/optflow.cpp; oracle S15) ----
    def optical_flow(self, cur, prev, radius=8, block=2):
        """cur/prev: uint8 [h,w] or [h,w,3] -> int16 [h,w,2] S10.5 flow (previous position = p - (flow >> 5))."""
        import torch
        ch = 3 if cur.dim() == 3 else 1
        _, cp, cs, _ = _geom(cur, 2 if ch == 3 else 1)
        _, pp, ps, _ = _geom(prev, 2 if ch == 3 else 1)
        if tuple(cur.shape[:2]) != (self.height, self.width) or cur.shape != prev.shape:
            raise EngineError("image shape does not match the engine")
        out = torch.empty((self.height, self.width, 2), dtype=torch.int16, device=cur.device)
        _, op, os_, _ = _geom(out, 2)
        self._check(self._lib.cart_optical_flow(self._h, cp, cs, pp, ps, ch, int(radius), int(block), op, os_, _stream_ptr()),
                    "cart_optical_flow")
        return out

    # ---- depth module (reference src/modules/depth.cpp:9-25) ----
    def reproject_depth(self, disp, Q):
        import torch
        n, p, s, fs = _geom(disp, 1)
        q = (C.c_float * 16)(*[float(v) for v in np.asarray(Q, np.float32).reshape(16)])
        out = torch.empty(tuple(disp.shape) + (3,), dtype=torch.float32, device=disp.device)
        _, op, os_, ofs = _geom(out, 2)
        self._check(self._lib.cart_reproject_depth(self._h, n, p, s, fs, q, op, os_, ofs, _stream_ptr()), "cart_reproject_depth")
        return out

    def plane_ccl(self, planes):
        import torch
        n, p, s, fs = _geom(planes, 1)
        ids = torch.empty(planes.shape, dtype=torch.int32, device=planes.device)
        ncomp = torch.empty((n,), dtype=torch.int32, device=planes.device)
        _, ip, is_, ifs = _geom(ids, 1)
        self._check(self._lib.cart_plane_ccl(self._h, n, p, s, fs, ip, is_, ifs, C.c_void_p(ncomp.data_ptr()),
                                             _stream_ptr()), "cart_plane_ccl")
        return ids, ncomp

    def plane_ccl_stats(self, planes, ids, max_components=4096):
        """-> (table int32 [n, max_components, 7] rows {id, label, area, x0, y0, x1, y1} in ascending id order,
        n_components int32 [n]) -- frames with more components keep their first max_components rows."""
        import torch
        n, p, s, fs = _geom(planes, 1)
        _, ip, is_, ifs = _geom(ids, 1)
        table = torch.empty((n, max_components, 7), dtype=torch.int32, device=planes.device)  # rows >= n_components stay undefined
        ncomp = torch.empty((n,), dtype=torch.int32, device=planes.device)
        self._check(self._lib.cart_plane_ccl_stats(self._h, n, p, s, fs, ip, is_, ifs, C.c_void_p(table.data_ptr()), int(max_components),
                                                   C.c_void_p(ncomp.data_ptr()), _stream_ptr()), "cart_plane_ccl_stats")
        return table, ncomp

    def plane_ccl_table(self, planes, max_components=4096):
        """plane_ccl + plane_ccl_stats in one call (cart_plane_ccl_table: four launches) -> (ids, table, n_components)."""
        import torch
        n, p, s, fs = _geom(planes, 1)
        ids = torch.empty(planes.shape, dtype=torch.int32, device=planes.device)
        _, ip, is_, ifs = _geom(ids, 1)
        table = torch.empty((n, max_components, 7), dtype=torch.int32, device=planes.device)  # rows >= n_components stay undefined
        ncomp = torch.empty((n,), dtype=torch.int32, device=planes.device)
        self._check(self._lib.cart_plane_ccl_table(self._h, n, p, s, fs, ip, is_, ifs, C.c_void_p(table.data_ptr()), int(max_components),
                                                   C.c_void_p(ncomp.data_ptr()), _stream_ptr()), "cart_plane_ccl_table")
        return ids, table, ncomp

    # ---- diagnostics ----
    def debug_ccl_scratch_nonzero(self):
        """Non-zero words of the component-table scratch (must be 0 between calls)."""
        n = C.c_size_t(0)
        self._check(self._lib.cart_debug_ccl_scratch_nonzero(self._h, C.byref(n)), "cart_debug_ccl_scratch_nonzero")
        return n.value

    def debug_read(self, what, frame_slot=0):
        lib = self._lib
        npx = self.width * self.height
        if what in (0, 1):
            buf = np.empty((self.height, self.width), np.uint8)
        elif what in (2, 3):
            buf = np.empty((self.height, self.width), np.uint32)
        elif 16 <= what < 16 + self.P:
            buf = np.empty((self.height, self.width, self.D), np.uint8)
        elif what in (32, 33):
            buf = np.empty((self.height, self.width), np.uint16)
        else:
            raise EngineError("unknown debug selector")
        assert buf.size >= npx
        self._check(lib.cart_debug_read(self._h, frame_slot, what, buf.ctypes.data_as(C.c_void_p), buf.nbytes),
                    "cart_debug_read")
        return buf

    def slab_layout(self):
        """cart_debug_slab_layout -> dict(group_slots, groups, slot_bytes, group_bytes): how the cost-slab workspace is cut into device allocations."""
        gs, ng, sb, gb = C.c_int(0), C.c_int(0), C.c_size_t(0), C.c_size_t(0)
        self._check(self._lib.cart_debug_slab_layout(self._h, C.byref(gs), C.byref(ng), C.byref(sb), C.byref(gb)), "cart_debug_slab_layout")
        return {"group_slots": gs.value, "groups": ng.value, "slot_bytes": sb.value, "group_bytes": gb.value}

    def set_timing(self, enabled=True, every=1):
        """Stage events on every `every`-th compute call (every=1: all)."""
        self._check(self._lib.cart_engine_set_timing(self._h, max(1, int(every)) if enabled else 0), "cart_engine_set_timing")

    def collect_timing(self):
        """-> ({stage: mean ms per call}, n_calls) over the calls recorded since set_timing(True)."""
        names = (C.c_char_p * 8)()
        ms = (C.c_float * 8)()
        calls = C.c_int(0)
        n = self._lib.cart_engine_collect_timing(self._h, names, ms, 8, C.byref(calls))
        if n < 0:
            raise EngineError("cart_engine_collect_timing: " + self._lib.cart_last_error(self._h).decode())
        return {names[i].decode(): float(ms[i]) for i in range(n)}, calls.value


class DevicePlaneSchedule:
    """Device-side replay of the reference's plane-parameter bookkeeping (cart_plane_schedule_* in the C ABI)."""

    def __init__(self, engine, provider="histogram_peak", static_params=None, update_interval=30, reset_interval=10):
        if provider not in ("histogram_peak", "static"):
            raise ValueError("Unknown parameter provider type.")  # cartconfig.cpp:77
        self._eng = engine
        self._lib = engine._lib
        init = PlaneParams(*(static_params or (0,) * 6))
        self._h = C.c_void_p()
        rc = self._lib.cart_plane_schedule_create(engine._h, 1 if provider == "histogram_peak" else 0, C.byref(init), update_interval,
                                                  reset_interval, C.byref(self._h))
        if rc != 0:
            raise EngineError("cart_plane_schedule_create: " + self._lib.cart_last_error(engine._h).decode())

    def advance(self, first_id, hists):
        """hists: int32 CUDA [n,256] in frame-id order -> int32 CUDA [n,6] parameters per frame."""
        import torch
        if hists.dtype != torch.int32 or not hists.is_contiguous() or hists.dim() != 2 or hists.shape[1] != 256:
            raise EngineError("hists must be a contiguous int32 [n,256] CUDA tensor")
        out = torch.empty((hists.shape[0], 6), dtype=torch.int32, device=hists.device)
        rc = self._lib.cart_plane_schedule_advance(self._h, first_id, hists.shape[0], C.c_void_p(hists.data_ptr()),
                                                   C.c_void_p(out.data_ptr()), _stream_ptr())
        if rc != 0:
            raise EngineError("cart_plane_schedule_advance: " + self._lib.cart_last_error(self._eng._h).decode())
        return out

    def read(self):
        p = PlaneParams()
        cum = (C.c_int32 * 256)()
        if self._lib.cart_plane_schedule_read(self._h, C.byref(p), cum) != 0:
            raise EngineError("cart_plane_schedule_read: " + self._lib.cart_last_error(self._eng._h).decode())
        return p, np.array(cum, dtype=np.int32)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cart_plane_schedule_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Superpixels:
    """The reference's ContourRelaxation object (persistent label image + relax), cart_superpixels_* in the C ABI.
    Label images are uint16; torch tensors carry them as int16 (same bits)."""

    def __init__(self, engine, block_size=12, direct_clique_cost=0.5, diagonal_clique_cost=None, compactness_weight=0.1,
                 progressive_compactness_cost=0.0, image_weight=1.5, disparity_weight=1.0, block_h=None):
        self._eng = engine
        self._lib = engine._lib
        p = SuperpixelParams(direct_clique_cost, direct_clique_cost / np.sqrt(2.0) if diagonal_clique_cost is None else diagonal_clique_cost,
                             compactness_weight, progressive_compactness_cost, image_weight, disparity_weight)
        self.params = p
        self._h = C.c_void_p()
        rc = self._lib.cart_superpixels_create(engine._h, C.byref(p), int(block_size), int(block_h or block_size), C.byref(self._h))
        if rc != 0:
            raise EngineError("cart_superpixels_create: " + self._lib.cart_last_error(engine._h).decode())

    def _check(self, rc, what):
        if rc != 0:
            raise EngineError(f"{what}: " + self._lib.cart_last_error(self._eng._h).decode())

    @property
    def max_label(self):
        return self._lib.cart_superpixels_max_label(self._h)

    def reset(self):
        self._check(self._lib.cart_superpixels_reset(self._h, _stream_ptr()), "cart_superpixels_reset")

    def set_labels(self, labels, max_label_id):
        _, p, s, _ = _geom(labels, 1)
        if labels.element_size() != 2:
            raise EngineError("labels must be a 16-bit tensor")
        self._check(self._lib.cart_superpixels_set_labels(self._h, p, s, int(max_label_id), _stream_ptr()), "cart_superpixels_set_labels")

    def relax(self, image, deriv2, iterations):
        """image: uint8 [h,w,3] BGR or [h,w] gray; deriv2: int16 [h,w,2] or None -> labels int16-viewed uint16 [h,w]."""
        import torch
        ch = 3 if image.dim() == 3 else 1
        _, ip, is_, _ = _geom(image, 2 if ch == 3 else 1)
        if tuple(image.shape[:2]) != (self._eng.height, self._eng.width):
            raise EngineError("image shape does not match the engine")
        dp, ds = C.c_void_p(None), 0
        if deriv2 is not None:
            _, dp, ds, _ = _geom(deriv2, 2)
        out = torch.empty((self._eng.height, self._eng.width), dtype=torch.int16, device=image.device)
        _, op, os_, _ = _geom(out, 1)
        self._check(self._lib.cart_superpixels_relax(self._h, ip, is_, ch, dp, ds, int(iterations), op, os_, _stream_ptr()),
                    "cart_superpixels_relax")
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cart_superpixels_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def resize_linear(img, dst_width, dst_height):
    """cv::cuda::resize(..., INTER_LINEAR) of the KITTI source (cart_resize_linear): uint8 CUDA [h,w] or [h,w,3] -> [dh,dw(,3)]."""
    import torch
    lib = _lib.load()
    ch = 3 if img.dim() == 3 else 1
    _, sp, ss, _ = _geom(img, 2 if ch == 3 else 1)
    sh, sw = img.shape[:2]
    out = torch.empty((dst_height, dst_width, 3) if ch == 3 else (dst_height, dst_width), dtype=torch.uint8, device=img.device)
    _, dp, ds, _ = _geom(out, 2 if ch == 3 else 1)
    rc = lib.cart_resize_linear(img.device.index or 0, sp, ss, sw, sh, ch, dp, ds, dst_width, dst_height, _stream_ptr())
    if rc != 0:
        raise EngineError("cart_resize_linear: " + lib.cart_last_error(None).decode())
    return out


def uniq_table(uniqueness_ratio, engine=None):
    """Integer uniqueness thresholds T(best) for best = 0..2047 (cart_debug_uniq_table): from the GPU when an engine is
    given, from the host-compiled copy of the same function otherwise."""
    lib = _lib.load()
    out = np.empty(2048, np.uint16)
    rc = lib.cart_debug_uniq_table(engine._h if engine is not None else None, int(uniqueness_ratio), out.ctypes.data_as(C.POINTER(C.c_uint16)))
    if rc != 0:
        raise EngineError("cart_debug_uniq_table: " + lib.cart_last_error(None).decode())
    return out


def find_plane_params(hist256, params=None):
    """HOST: reference HistogramPeakPlaneParameterProvider::updatePlaneParameters (planeseg.cu:405-458).
    -> (updated: bool, PlaneParams)."""
    lib = _lib.load()
    h = np.ascontiguousarray(np.asarray(hist256, dtype=np.int32).reshape(256))
    p = PlaneParams(*(params.as_tuple() if isinstance(params, PlaneParams) else (params or (0,) * 6)))
    rc = lib.cart_find_plane_params(h.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(p))
    if rc < 0:
        raise EngineError("cart_find_plane_params: " + lib.cart_last_error(None).decode())
    return bool(rc), p


def find_peaks(data):
    """HOST: reference util::findPeaks (src/utils/peaks.cpp:12-72) -> list of (born, died, left, right)."""
    lib = _lib.load()
    d = np.ascontiguousarray(np.asarray(data, dtype=np.int32).ravel())
    n = d.size
    arrs = [(C.c_int * n)() for _ in range(4)]
    k = lib.cart_find_peaks(d.ctypes.data_as(C.POINTER(C.c_int32)), n, *arrs)
    if k < 0:
        raise EngineError("cart_find_peaks: " + lib.cart_last_error(None).decode())
    return [tuple(a[i] for a in arrs) for i in range(k)]
