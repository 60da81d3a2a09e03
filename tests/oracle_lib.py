"""ctypes wrapper around oracle/_build/libcart_oracle.so -- TEST INFRASTRUCTURE ONLY.

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.environ.get("CART_ORACLE_LIB") or os.path.join(ORACLE_DIR, "_build", "libcart_oracle.so")   # override: sanitizer builds (make -C oracle sanitize)


class SgmParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("width", "height", "min_disparity", "num_disparities", "paths", "p1", "p2",
                                       "uniqueness_ratio")]


class PlaneParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("horizontal_min", "horizontal_max", "vertical_min", "vertical_max",
                                       "horizontal_center", "vertical_center")]

    def as_tuple(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


_lib = None


def build():
    if os.environ.get("CART_ORACLE_LIB"):
        return
    src_time = max(os.path.getmtime(os.path.join(ORACLE_DIR, f)) for f in ("cart_oracle.c", "cart_oracle_sp.c", "cart_oracle.h"))
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < src_time:
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.cart_oracle_sgm.restype = C.c_int
        _lib.cart_oracle_sgm_ex.restype = C.c_int
        _lib.cart_oracle_find_peaks.restype = C.c_int
        _lib.cart_oracle_histogram_peak_params.restype = C.c_int
        _lib.cart_oracle_ccl.restype = C.c_int
        _lib.cart_oracle_sp_block_init.restype = C.c_int
        _lib.cart_oracle_sp_relax.restype = C.c_long
        _lib.cart_oracle_log.restype = C.c_double
        _lib.cart_oracle_log.argtypes = [C.c_double]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def bgr2gray(bgr):
    h, w, _ = bgr.shape
    bgr = np.ascontiguousarray(bgr, np.uint8)
    out = np.empty((h, w), np.uint8)
    lib().cart_oracle_bgr2gray(_p(bgr), C.c_size_t(w * 3), w, h, _p(out))
    return out


def census(gray):
    h, w = gray.shape
    gray = np.ascontiguousarray(gray, np.uint8)
    out = np.empty((h, w), np.uint32)
    lib().cart_oracle_census9x7(_p(gray), w, h, _p(out))
    return out


def path_dir(i):
    dx, dy = C.c_int(), C.c_int()
    lib().cart_oracle_path_dir(i, C.byref(dx), C.byref(dy))
    return dx.value, dy.value


def aggregate_path(cl, cr, D, min_disp, p1, p2, dx, dy):
    h, w = cl.shape
    L = np.empty((h, w, D), np.uint8)
    lib().cart_oracle_aggregate_path(_p(np.ascontiguousarray(cl)), _p(np.ascontiguousarray(cr)), w, h, D, min_disp, p1, p2,
                                     dx, dy, _p(L))
    return L


def wta(S, uniqueness_ratio, variants=0):
    h, w, D = S.shape
    S = np.ascontiguousarray(S, np.uint16)
    l = np.empty((h, w), np.uint16); r = np.empty((h, w), np.uint16)
    lib().cart_oracle_wta_ex(_p(S), w, h, D, uniqueness_ratio, _p(l), _p(r), int(variants))
    return l, r


VARIANT_S8_ZERO_INVALID, VARIANT_S7_REPLICATE_BORDER, VARIANT_S5_TOP2 = 1, 2, 4   # CART_ORACLE_VARIANT_* (the choices that are open upstream)


def median3x3(a, variants=0):
    h, w = a.shape
    a = np.ascontiguousarray(a, np.uint16)
    o = np.empty_like(a)
    lib().cart_oracle_median3x3_u16_ex(_p(a), w, h, _p(o), int(variants))
    return o


def lr_check_range(lm, rm, gray, min_disp, variants=0):
    h, w = lm.shape
    o = np.empty((h, w), np.int16)
    lib().cart_oracle_lr_check_range_ex(_p(np.ascontiguousarray(lm, np.uint16)), _p(np.ascontiguousarray(rm, np.uint16)),
                                        _p(np.ascontiguousarray(gray, np.uint8)), w, h, min_disp, _p(o), int(variants))
    return o


def sgm(gl, gr, D, paths, min_disp=4, p1=10, p2=120, uniq=12, want_S=False, variants=0):
    h, w = gl.shape
    p = SgmParams(w, h, min_disp, D, paths, p1, p2, uniq)
    disp = np.empty((h, w), np.int16)
    S = np.empty((h, w, D), np.uint16) if want_S else None
    rc = lib().cart_oracle_sgm_ex(C.byref(p), _p(np.ascontiguousarray(gl, np.uint8)), _p(np.ascontiguousarray(gr, np.uint8)),
                                  _p(disp), _p(S) if want_S else None, int(variants))
    assert rc == 0, "cart_oracle_sgm failed"
    return (disp, S) if want_S else disp


def interpolate(disp, radius, iterations, min_disp16, max_disp):
    h, w = disp.shape
    o = np.empty((h, w), np.int16)
    lib().cart_oracle_interpolate(_p(np.ascontiguousarray(disp, np.int16)), w, h, radius, iterations, min_disp16, max_disp, _p(o))
    return o


def disparity_module(left, right, D, paths, min_disp=4, p1=10, p2=120, uniq=12, radius=-1, iterations=5, variants=0):
    """Whole ImageDisparityModule::runInternal (disparity.cu:49-80) on host arrays."""
    if left.ndim == 3:
        left, right = bgr2gray(left), bgr2gray(right)
    d = sgm(left, right, D, paths, min_disp, p1, p2, uniq, variants=variants)
    if radius > 0 and iterations > 0:
        d = interpolate(d, radius, iterations, min_disp * 16, left.shape[1])
    return d


def directional_derivative(disp):
    h, w = disp.shape
    o = np.empty((h, w, 2), np.int16); hist = np.empty((256, 2), np.int32)
    lib().cart_oracle_directional_derivative(_p(np.ascontiguousarray(disp, np.int16)), w, h, _p(o), _p(hist))
    return o, hist


def plane_derivative(disp, hist=None):
    h, w = disp.shape
    o = np.empty((h, w), np.int16)
    if hist is None:
        hist = np.zeros(256, np.int32)
    assert hist.dtype == np.int32 and hist.flags.c_contiguous
    lib().cart_oracle_plane_derivative(_p(np.ascontiguousarray(disp, np.int16)), w, h, _p(o), _p(hist))
    return o, hist


def find_peaks(data):
    data = np.ascontiguousarray(data, np.int32)
    n = data.size
    arrs = [np.empty(n, np.int32) for _ in range(4)]
    k = lib().cart_oracle_find_peaks(_p(data), n, *[_p(a) for a in arrs])
    return [tuple(int(a[i]) for a in arrs) for i in range(k)]


def histogram_peak_params(hist, params=None):
    hist = np.ascontiguousarray(hist, np.int32)
    p = PlaneParams(*(params or (0,) * 6))
    rc = lib().cart_oracle_histogram_peak_params(_p(hist), C.byref(p))
    return bool(rc), p.as_tuple()


def classify(deriv, params):
    h, w = deriv.shape
    p = PlaneParams(*params)
    o = np.empty((h, w), np.uint8)
    lib().cart_oracle_classify(_p(np.ascontiguousarray(deriv, np.int16)), w, h, C.byref(p), _p(o))
    return o


def ccl(planes):
    h, w = planes.shape
    ids = np.empty((h, w), np.int32)
    n = lib().cart_oracle_ccl(_p(np.ascontiguousarray(planes, np.uint8)), w, h, _p(ids))
    return ids, n


def ccl_stats(planes, ids, max_components=None):
    """-> int32 [n,7] rows {id, label, area, x0, y0, x1, y1} in ascending id order (S12)."""
    h, w = planes.shape
    cap = int(max_components if max_components is not None else h * w)
    table = np.zeros((cap, 7), np.int32)
    lib().cart_oracle_ccl_stats.restype = C.c_int
    n = lib().cart_oracle_ccl_stats(_p(np.ascontiguousarray(planes, np.uint8)), _p(np.ascontiguousarray(ids, np.int32)), w, h, _p(table), cap)
    return table[:min(n, cap)], n


def reproject_depth(disp, Q):
    h, w = disp.shape
    Q = np.ascontiguousarray(Q, np.float32).reshape(16)
    o = np.empty((h, w, 3), np.float32)
    lib().cart_oracle_reproject_depth(_p(np.ascontiguousarray(disp, np.int16)), w, h, _p(Q), _p(o))
    return o


def resize_linear(img, dw, dh):
    """S16: u8 [h,w] or [h,w,3] -> [dh,dw(,3)] (cv::cuda::resize INTER_LINEAR as restated in cart_oracle.h)."""
    img = np.ascontiguousarray(img, np.uint8)
    sh, sw = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    o = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, ch), np.uint8)
    lib().cart_oracle_resize_linear(_p(img), sw, sh, ch, _p(o), dw, dh)
    return o


def kitti_q_matrix(p2, p3, scale_width=1.0, scale_height=1.0):
    """Q as KITTIDataSource builds it (src/sources/kitti.cpp:28-86, :137-148) from the 12 numbers of the P2 / P3 rows;
    scale_* = configured image size / file image size (float32, :137-138)."""
    fx, cx, fubx, cy = np.float32(p2[0]), np.float32(p2[2]), np.float32(p2[3]), np.float32(p2[6])
    cxr = np.float32(p3[2])
    sw, sh = np.float32(scale_width), np.float32(scale_height)
    baseline = np.float32(-fubx / fx)
    Q = np.eye(4, dtype=np.float32)
    Q[0, 3] = -cx * sw; Q[1, 3] = -cy * sh; Q[2, 2] = 0; Q[2, 3] = fx * sw
    Q[3, 2] = np.float32(-1.0 / baseline)
    Q[3, 3] = np.float32((cx - cxr) * sw / baseline)
    return Q


def temporal_vote(planes, prev_planes, flows):
    h, w = planes.shape
    n = len(prev_planes)
    pp = [np.ascontiguousarray(a, np.uint8) for a in prev_planes]
    ff = [np.ascontiguousarray(a, np.int16) for a in flows]
    P = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in pp])
    F = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in ff])
    o = np.empty((h, w), np.uint8)
    lib().cart_oracle_temporal_vote(_p(np.ascontiguousarray(planes, np.uint8)), w, h, n, P, F, _p(o))
    return o


# ---- superpixels (S13/S14) -------------------------------------------------------------------------------------
class SpParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("direct_clique_cost", "diagonal_clique_cost", "compactness_weight",
                                          "progressive_compactness_cost", "image_weight", "disparity_weight")]


def sp_params(direct=0.5, diagonal=None, compactness=0.1, progressive=0.0, image=1.5, disparity=1.0):
    """Defaults of the reference's JSON factory (cartconfig.cpp:121-133)."""
    return SpParams(direct, direct / np.sqrt(2.0) if diagonal is None else diagonal, compactness, progressive, image, disparity)


def bgr2ycrcb(bgr):
    h, w, _ = bgr.shape
    bgr = np.ascontiguousarray(bgr, np.uint8)
    out = np.empty((h, w, 3), np.uint8)
    lib().cart_oracle_bgr2ycrcb(_p(bgr), C.c_size_t(w * 3), w, h, _p(out))
    return out


def sp_block_init(w, h, bw, bh):
    labels = np.empty((h, w), np.uint16)
    mx = lib().cart_oracle_sp_block_init(w, h, bw, bh, _p(labels))
    return labels, mx


def log(x):
    return lib().cart_oracle_log(float(x))


def sp_relax(params, labels, max_label_id, ycrcb, deriv2, iterations):
    """-> (new labels, number of label changes)."""
    h, w = labels.shape
    out = np.ascontiguousarray(labels, np.uint16).copy()
    yc = None if ycrcb is None else np.ascontiguousarray(ycrcb, np.uint8)
    d2 = None if deriv2 is None else np.ascontiguousarray(deriv2, np.int16)
    n = lib().cart_oracle_sp_relax(C.byref(params), _p(out), w, h, int(max_label_id), _p(yc) if yc is not None else None,
                                   _p(d2) if d2 is not None else None, int(iterations))
    if n < 0:
        raise ValueError("cart_oracle_sp_relax: bad arguments")
    return out, n


def sp_classify(deriv2, labels, max_label, params, prev_planes=(), flows=()):
    h, w = labels.shape
    d2 = np.ascontiguousarray(deriv2, np.int16)
    lb = np.ascontiguousarray(labels, np.uint16)
    pp = params if isinstance(params, PlaneParams) else PlaneParams(*params)
    n = len(prev_planes)
    prevs = [np.ascontiguousarray(a, np.uint8) for a in prev_planes]
    fl = [np.ascontiguousarray(a, np.int16) for a in flows]
    pa = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in prevs])
    fa = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in fl])
    uns = np.empty((h, w), np.uint8)
    out = np.empty((h, w), np.uint8)
    lib().cart_oracle_sp_classify(_p(d2), _p(lb), w, h, int(max_label), C.byref(pp), n, pa, fa, _p(uns), _p(out))
    return uns, out


def block_flow(gray_cur, gray_prev, radius=8, block=2):
    """S15: -> int16 [h,w,2] S10.5 flow (previous position = p - (flow >> 5))."""
    h, w = gray_cur.shape
    cc = np.ascontiguousarray(census(gray_cur)); cp = np.ascontiguousarray(census(gray_prev))
    out = np.empty((h, w, 2), np.int16)
    lib().cart_oracle_block_flow(_p(cc), _p(cp), w, h, int(radius), int(block), _p(out))
    return out
