"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/cart_engine.h
declares (no compute calls -- there is no GPU here), and its HOST entry points agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cart_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cart_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from cartslam import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 15
    for name in syms:
        assert hasattr(lib, name), f"{name} declared in include/cart_engine.h but not exported"
    assert set(syms) == set(_lib.PROTOTYPES), "python prototypes out of sync with the header"
    assert lib.cart_engine_version().decode().startswith("cart_engine gfx950")


def test_struct_layouts_match_header():
    from cartslam import _lib
    assert C.sizeof(_lib.EngineParams) == 12 * 4 and C.sizeof(_lib.PlaneParams) == 6 * 4
    assert C.sizeof(_lib.SuperpixelParams) == 6 * 8
    assert C.sizeof(_lib.PlacementReport) == 5 * 4 + 4 * 4 and C.sizeof(_lib.LaunchPlan) == 3 * 4   # cart_placement_report, cart_launch_plan
    assert len(_lib.PLACE_MODES) == 4 and len(_lib.PLACE_STOPS) == 6                                   # CART_PLACE_MODE_* / CART_PLACE_STOP_*
    sp = _lib.SuperpixelParams()
    _lib.load().cart_superpixel_default_params(C.byref(sp))
    # cartconfig.cpp:128-133
    assert (sp.direct_clique_cost, sp.compactness_weight, sp.progressive_compactness_cost, sp.image_weight, sp.disparity_weight) == (0.5, 0.1, 0.0, 1.5, 1.0)
    assert sp.diagonal_clique_cost == 0.5 / np.sqrt(2.0)
    p = _lib.EngineParams()
    _lib.load().cart_engine_default_params(C.byref(p))
    # reference defaults: cartconfig.cpp:144-152, disparity.hpp:32, cartslam.hpp:4
    assert (p.min_disparity, p.num_disparities, p.paths, p.p1, p.p2) == (4, 256, 4, 10, 120)
    assert (p.uniqueness_ratio, p.smoothing_radius, p.smoothing_iterations, p.max_inflight) == (12, -1, 5, 12)


def test_create_fails_loudly_without_gpu_or_with_bad_params():
    import torch
    from cartslam import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(100, 50, num_disparities=100)  # validated before any device call
    if not torch.cuda.is_available():
        with pytest.raises(EngineError):
            Engine(100, 50, num_disparities=64)  # no device: must raise, never fall back to a CPU path


def test_host_peak_finder_matches_oracle():
    from cartslam import find_peaks, find_plane_params
    rng = np.random.default_rng(9)
    for k in range(60):
        if k % 3 == 0:
            hh = rng.integers(0, 40, 256)
        elif k % 3 == 1:
            hh = rng.integers(0, 5, 256) * rng.integers(0, 3000, 256)
        else:  # smooth two-peak shapes like real derivative histograms
            x = np.arange(256)
            hh = 5000 * np.exp(-0.5 * ((x - 128) / rng.uniform(1, 4)) ** 2) + 2000 * np.exp(-0.5 * ((x - rng.integers(132, 150)) / rng.uniform(1, 5)) ** 2)
            hh = hh + rng.integers(0, 30, 256)
        hh = hh.astype(np.int32)
        assert find_peaks(hh) == O.find_peaks(hh)
        prev = tuple(int(v) for v in rng.integers(-20, 20, 6))
        ok, p = find_plane_params(hh, prev)
        eok, ep = O.histogram_peak_params(hh, prev)
        assert ok == eok and p.as_tuple() == ep


def test_every_entry_point_rejects_null_arguments():
    """No entry point dereferences a NULL handle / image / table: all-zero arguments give a non-zero status and a message
    (the reference's GPU failures exit() the process, include/utils/cuda.cuh:193-201; the C ABI never does).  Runs
    without a GPU: validation comes before any device call."""
    from cartslam import _lib
    lib = _lib.load()
    skipped = {"cart_engine_create", "cart_find_plane_params", "cart_find_peaks"}   # covered by their own tests
    checked = 0
    for name, (res, args) in _lib.PROTOTYPES.items():
        if res is not C.c_int or not args or name in skipped:
            continue
        zeros = [a(0) if a in (C.c_int, C.c_size_t, C.c_float, C.c_double) else None for a in args]
        assert getattr(lib, name)(*zeros) != 0, name
        assert lib.cart_last_error(None), name
        checked += 1
    assert checked >= 25


def _uniq_threshold_reference(ratio):
    """T(best) = min{ s : (float)s * u >= (float)best } by brute force over s = 0..4095 in float32 -- the float compare
    of oracle S5 (oracle/cart_oracle.c, cart_oracle_wta) solved for s; 4095 where no s passes."""
    u = np.float32(100 - ratio) / np.float32(100.0)
    s = np.arange(4096, dtype=np.float32) * u                   # float32 products, like the kernels' (float)s * u
    best = np.arange(2048, dtype=np.float32)
    ok = s[None, :] >= best[:, None]
    return np.where(ok.any(axis=1), ok.argmax(axis=1), 4095).astype(np.uint16)


def _check_uniq_table(got, ratio):
    """got[best] for best = 0..2047.  Reachable sums are <= 8 * 255 = 2040, so (a) wherever the true threshold is
    reachable the table must hold exactly it (above, the kernels clamp to 4095 = "no S passes"), and (b) -- the property
    the kernels use -- S >= T(best) must equal the float compare for EVERY best <= 2040 and S <= 2040."""
    exp = _uniq_threshold_reference(ratio)
    reach = np.arange(2048) <= 2040
    exact = (got == exp) | ((exp > 2040) & (got > 2040))
    assert exact[reach].all(), f"ratio {ratio}: first mismatch at best = {int(np.argmax(~exact & reach))}"
    u = np.float32(100 - ratio) / np.float32(100.0)
    S = np.arange(2041, dtype=np.int64)
    lhs = S[None, :] >= got.astype(np.int64)[:2041, None]
    rhs = (S.astype(np.float32) * u)[None, :] >= np.arange(2041, dtype=np.float32)[:, None]
    assert (lhs == rhs).all(), f"ratio {ratio}: integer test and float compare disagree"


def test_integer_uniqueness_threshold_is_the_float_compare_exhaustively():
    """DESIGN section 4 claims the WTA kernels' integer threshold equals the float uniqueness compare: enumerated here for
    every best cost 0..2040, every S 0..2040 and every uniqueness ratio 0..100 (host-compiled copy of the device function;
    the GPU twin of this test is tests/test_gpu_parity.py::test_integer_uniqueness_threshold_on_device)."""
    from cartslam.engine import uniq_table
    for ratio in range(101):
        _check_uniq_table(uniq_table(ratio), ratio)
