"""ctypes binding of the C ABI declared in include/cart_engine.h.

The shared library is built in-tree by `make -C cart-slam_amd` (or __graft_entry__.build()).
There is NO fallback: if the library is missing, loading raises and every op fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CART_ENGINE_LIB") or os.path.join(os.path.dirname(_HERE), "build", "libcart_engine.so")  # override: timing experiments


class EngineParams(C.Structure):
    # mirrors cart_engine_params (include/cart_engine.h)
    _fields_ = [(n, C.c_int) for n in (
        "device_id", "width", "height", "min_disparity", "num_disparities", "paths", "p1", "p2",
        "uniqueness_ratio", "smoothing_radius", "smoothing_iterations", "max_inflight")]


class PlaneParams(C.Structure):
    # mirrors cart_plane_params (include/cart_engine.h; reference include/modules/planeseg.hpp:25-34)
    _fields_ = [(n, C.c_int) for n in (
        "horizontal_min", "horizontal_max", "vertical_min", "vertical_max",
        "horizontal_center", "vertical_center")]

    def as_tuple(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


class LaunchPlan(C.Structure):
    # mirrors cart_launch_plan (include/cart_engine.h)
    _fields_ = [("frames_per_launch", C.c_int), ("plan", C.c_int), ("slabs_written", C.c_int)]


PLAN_AUTO, PLAN_SLABS, PLAN_FUSED_UP = -1, 0, 1      # CART_PLAN_*
OPT_PLAN, OPT_PLAN_MIN_FRAMES, OPT_CHUNK_FRAMES = 0, 1, 2           # CART_OPT_*
OPT_SPEC_S8_ZERO_INVALID, OPT_SPEC_S7_REPLICATE_BORDER, OPT_SPEC_S5_TOP2 = 3, 4, 5   # CART_OPT_SPEC_*: upstream variants of oracle S8 / S7 / S5


class PlacementReport(C.Structure):
    # mirrors cart_placement_report (include/cart_engine.h)
    _fields_ = [("ms_first", C.c_float), ("ms_kept", C.c_float), ("ms_fastest_seen", C.c_float), ("ms_slowest_seen", C.c_float),
                ("seconds", C.c_float), ("units", C.c_int), ("candidates", C.c_int), ("mode", C.c_int), ("stop_reason", C.c_int)]


PLACE_MODES = {0: "unknown", 1: "fast", 2: "mixed", 3: "uniform"}                                       # CART_PLACE_MODE_*
PLACE_STOPS = {0: "nothing to do", 1: "fast set found", 2: "uniform", 3: "tries", 4: "time", 5: "memory"}   # CART_PLACE_STOP_*


class SuperpixelParams(C.Structure):
    # mirrors cart_superpixel_params (include/cart_engine.h; reference cartconfig.cpp:121-133)
    _fields_ = [(n, C.c_double) for n in (
        "direct_clique_cost", "diagonal_clique_cost", "compactness_weight", "progressive_compactness_cost",
        "image_weight", "disparity_weight")]


# every symbol include/cart_engine.h declares, with its prototype
_vp, _sz, _i = C.c_void_p, C.c_size_t, C.c_int
PROTOTYPES = {
    "cart_engine_default_params": (None, [C.POINTER(EngineParams)]),
    "cart_engine_create": (_i, [C.POINTER(EngineParams), C.POINTER(_vp)]),
    "cart_engine_destroy": (None, [_vp]),
    "cart_last_error": (C.c_char_p, [_vp]),
    "cart_engine_set_option": (_i, [_vp, _i, _i]),
    "cart_engine_get_option": (_i, [_vp, _i, C.POINTER(_i)]),
    "cart_engine_describe_plan": (_i, [_vp, _i, C.POINTER(LaunchPlan)]),
    "cart_engine_tune_placement": (_i, [_vp, _i, _i, _sz, C.POINTER(PlacementReport)]),
    "cart_compute_disparity": (_i, [_vp, _vp, _sz, _vp, _sz, _i, _vp, _sz, _vp]),
    "cart_compute_disparity_batch": (_i, [_vp, _i, _vp, _sz, _sz, _vp, _sz, _sz, _i, _vp, _sz, _sz, _vp]),
    "cart_compute_disparity_multi": (_i, [_vp, _i, _vp, _sz, _vp, _sz, _i, _vp, _sz, _vp]),
    "cart_interpolate": (_i, [_vp, _i, _vp, _sz, _sz, _i, _i, _i, _i, _vp]),
    "cart_disparity_derivative": (_i, [_vp, _i, _vp, _sz, _sz, _vp, _sz, _sz, _vp, _vp]),
    "cart_plane_derivative_hist": (_i, [_vp, _i, _vp, _sz, _sz, _vp, _sz, _sz, _vp, _sz, _vp]),
    "cart_plane_classify": (_i, [_vp, _i, _vp, _sz, _sz, C.POINTER(PlaneParams), _i, _vp, _sz, _sz, _vp]),
    "cart_plane_derivative_hist_multi": (_i, [_vp, _i, _vp, _sz, _vp, _sz, _vp, _sz, _vp]),
    "cart_plane_classify_multi": (_i, [_vp, _i, _vp, _sz, C.POINTER(PlaneParams), _i, _vp, _sz, _vp]),
    "cart_plane_ccl": (_i, [_vp, _i, _vp, _sz, _sz, _vp, _sz, _sz, _vp, _vp]),
    "cart_plane_ccl_stats": (_i, [_vp, _i, _vp, _sz, _sz, _vp, _sz, _sz, _vp, _i, _vp, _vp]),
    "cart_plane_ccl_table": (_i, [_vp, _i, _vp, _sz, _sz, _vp, _sz, _sz, _vp, _i, _vp, _vp]),
    "cart_plane_schedule_create": (_i, [_vp, _i, C.POINTER(PlaneParams), _i, _i, C.POINTER(_vp)]),
    "cart_plane_schedule_destroy": (None, [_vp]),
    "cart_plane_schedule_advance": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "cart_plane_schedule_read": (_i, [_vp, C.POINTER(PlaneParams), C.POINTER(C.c_int32)]),
    "cart_plane_classify_dev": (_i, [_vp, _i, _vp, _sz, _sz, _vp, _i, _vp, _sz, _sz, _vp]),
    "cart_plane_temporal_vote": (_i, [_vp, _vp, _sz, _i, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_vp), C.POINTER(_sz), _vp, _sz, _vp]),
    "cart_reproject_depth": (_i, [_vp, _i, _vp, _sz, _sz, C.POINTER(C.c_float), _vp, _sz, _sz, _vp]),
    "cart_superpixel_default_params": (None, [C.POINTER(SuperpixelParams)]),
    "cart_superpixels_create": (_i, [_vp, C.POINTER(SuperpixelParams), _i, _i, C.POINTER(_vp)]),
    "cart_superpixels_destroy": (None, [_vp]),
    "cart_superpixels_reset": (_i, [_vp, _vp]),
    "cart_superpixels_set_labels": (_i, [_vp, _vp, _sz, _i, _vp]),
    "cart_superpixels_relax": (_i, [_vp, _vp, _sz, _i, _vp, _sz, _i, _vp, _sz, _vp]),
    "cart_superpixels_max_label": (_i, [_vp]),
    "cart_superpixel_plane_classify": (_i, [_vp, _vp, _sz, _vp, _sz, _i, C.POINTER(PlaneParams), _i, C.POINTER(_vp), C.POINTER(_sz),
                                           C.POINTER(_vp), C.POINTER(_sz), _vp, _sz, _vp, _sz, _vp]),
    "cart_optical_flow": (_i, [_vp, _vp, _sz, _vp, _sz, _i, _i, _i, _vp, _sz, _vp]),
    "cart_resize_linear": (_i, [_i, _vp, _sz, _i, _i, _i, _vp, _sz, _i, _i, _vp]),
    "cart_copy_narrow": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "cart_find_plane_params": (_i, [C.POINTER(C.c_int32), C.POINTER(PlaneParams)]),
    "cart_find_peaks": (_i, [C.POINTER(C.c_int32), _i] + [C.POINTER(C.c_int)] * 4),
    "cart_debug_read": (_i, [_vp, _i, _i, _vp, _sz]),
    "cart_debug_uniq_table": (_i, [_vp, _i, C.POINTER(C.c_uint16)]),
    "cart_debug_ccl_scratch_nonzero": (_i, [_vp, C.POINTER(C.c_size_t)]),
    "cart_debug_slab_layout": (_i, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "cart_engine_set_timing": (_i, [_vp, _i]),
    "cart_engine_collect_timing": (_i, [_vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), _i, C.POINTER(_i)]),
    "cart_engine_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Loads libcart_engine.so once; raises (loudly) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP engine is not built (run `make -C cart-slam_amd` or "
                "__graft_entry__.build()). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib
