"""CPU tests: the C oracle against (a) the independent numpy restatement, (b) hand-computed
known answers, (c) the committed golden fixtures.  PARITY UNPINNED: the reference holds no
fixtures for this path; the golden files are produced by tests/golden/make_golden.py."""
import glob
import os

import numpy as np
import pytest

import np_ref as N
import oracle_lib as O
from cartslam import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def rand_disp(rng, h, w, lo=64, hi=1200, p_invalid=0.2):
    d = rng.integers(lo, hi, (h, w)).astype(np.int16)
    d[rng.random((h, w)) < p_invalid] = -32768
    return d


def test_gray_known_answers():
    bgr = np.array([[[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 77]]], np.uint8)
    g = O.bgr2gray(bgr)[0]
    # (1868*B + 9617*G + 4899*R + 8192) >> 14
    assert list(g) == [0, 255, 29, 150, 76, (1868 * 10 + 9617 * 200 + 4899 * 77 + 8192) >> 14]
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (9, 13, 3)).astype(np.uint8)
    assert (O.bgr2gray(x) == N.bgr2gray(x)).all()


def test_census_known_answers():
    # horizontal ramp: every pair with dx<0 compares darker-left > brighter-right = 0, dx>0 -> 1
    g = np.tile(np.arange(20, dtype=np.uint8) * 5, (12, 1))
    c = O.census(g)
    assert (c[:3] == 0).all() and (c[-3:] == 0).all() and (c[:, :4] == 0).all() and (c[:, -4:] == 0).all()
    bits = []
    for dy in (-3, -2, -1):
        bits += [1 if dx > 0 else 0 for dx in range(-4, 5)]
    bits += [0] * 4
    expect = int("".join(map(str, bits)), 2)
    assert (c[3:-3, 4:-4] == expect).all()
    assert (O.census(np.full((10, 12), 7, np.uint8)) == 0).all()


@pytest.mark.parametrize("w,h,D,P,md", [(72, 40, 64, 8, 4), (80, 24, 64, 4, 0), (150, 20, 128, 8, 7)])
def test_sgm_matches_numpy_restatement(w, h, D, P, md):
    l, r, _ = synth.make_pair(w, h, D, md, seed=123 + w)
    ref = N.sgm(l, r, D, P, md)
    cl, cr = O.census(l), O.census(r)
    assert (cl == ref["census_l"]).all() and (cr == ref["census_r"]).all()
    for i in range(P):
        dx, dy = O.path_dir(i)
        assert (dx, dy) == N.DIRS[i]
        assert (O.aggregate_path(cl, cr, D, md, 10, 120, dx, dy) == ref["paths"][i]).all(), f"path {i}"
    d, S = O.sgm(l, r, D, P, md, want_S=True)
    assert (S == ref["S"]).all()
    wl, wr = O.wta(S, 12)
    assert (wl == ref["wta_l"]).all() and (wr == ref["wta_r"]).all()
    assert (d == ref["disp"]).all()


def test_path_properties():
    l, r, _ = synth.make_pair(96, 48, 64, 4, seed=5)
    cl, cr = O.census(l), O.census(r)
    C = N.cost_volume(cl, cr, 64, 4)
    for i in range(8):
        dx, dy = O.path_dir(i)
        L = O.aggregate_path(cl, cr, 64, 4, 10, 120, dx, dy).astype(np.int32)
        # path start == matching cost; everywhere C <= L <= C + P2
        if dy > 0: assert (L[0] == C[0]).all()
        if dy < 0: assert (L[-1] == C[-1]).all()
        if dx > 0: assert (L[:, 0] == C[:, 0]).all()
        if dx < 0: assert (L[:, -1] == C[:, -1]).all()
        assert (L >= C).all() and (L <= C + 120).all() and L.max() <= 151
    # P1 = P2 = 0 makes every path the plain cost plus nothing: L == C
    L0 = O.aggregate_path(cl, cr, 64, 4, 0, 0, 1, 1)
    assert (L0 == C).all()


def test_wta_known_answers():
    D = 64
    S = np.full((1, 70, D), 500, np.uint16)
    S[0, :, 10] = 100; S[0, :, 9] = 300; S[0, :, 11] = 200           # clean minimum, sub-pixel towards 11
    S[0, 5, 40] = 105                                                 # a second, non-adjacent near-minimum -> not unique
    S[0, 6, 11] = 100                                                 # adjacent tie -> unique, lowest d wins
    S[0, 7, :] = 77                                                   # flat -> everything equal: 77*0.88 < 77 -> invalid
    wl, wr = O.wta(S, 12)
    num, den = 300 - 200, 300 - 200 + 200
    assert wl[0, 0] == 10 * 16 + (num * 16 + den) // (2 * den)
    assert wl[0, 5] == 0xFFFF and wl[0, 7] == 0xFFFF
    assert wl[0, 6] == 10 * 16 + ((300 - 100) * 16 + (300 - 200 + 100)) // (2 * (300 - 200 + 100))
    # right view: argmin_d S(p+d, d); column 10 is the minimum everywhere it exists
    assert wr[0, 20] == 10 and wr[0, 59] == 10 and wr[0, 69] == 0
    assert wr[0, 0] == 7  # S(7,7) = 77 on the flat pixel beats S(10,10) = 100
    # uniqueness 0 -> u = 1.0: every candidate passes S*1.0 >= best, even the flat pixel (-> d = 0)
    wl0, _ = O.wta(S, 0)
    assert wl0[0, 5] != 0xFFFF and wl0[0, 7] == 0


def test_median_and_lr_check():
    rng = np.random.default_rng(3)
    a = rng.integers(0, 2000, (17, 23)).astype(np.uint16)
    a[rng.random(a.shape) < 0.3] = 0xFFFF
    assert (O.median3x3(a) == N.median3x3(a)).all()
    lm = (rng.integers(0, 30, (9, 40)) * 16).astype(np.uint16); lm[2, 3] = 0xFFFF
    rm = rng.integers(0, 30, (9, 40)).astype(np.uint16)
    g = rng.integers(0, 3, (9, 40)).astype(np.uint8)
    assert (O.lr_check_range(lm, rm, g, 4) == N.lr_check_range(lm, rm, g, 4)).all()
    out = O.lr_check_range(lm, rm, np.zeros_like(g), 4)
    assert (out == 48).all()  # gray==0 masks everything: (min_disp-1)*16


@pytest.mark.parametrize("r,it", [(1, 1), (2, 1), (3, 2), (2, 5)])
def test_interpolate(r, it):
    rng = np.random.default_rng(r * 10 + it)
    d = rand_disp(rng, 21, 37, 40, 400, 0.3)
    assert (O.interpolate(d, r, it, 64, 300) == N.interpolate(d, r, it, 64, 300)).all()


def test_derivatives_classify_ccl():
    rng = np.random.default_rng(7)
    d = rand_disp(rng, 33, 41, 64, 180, 0.15)
    d[5, 5] = 32767; d[3, 5] = -32767  # forces s16 wrap in both derivative kernels
    a, h = O.directional_derivative(d); a2, h2 = N.directional_derivative(d)
    assert (a == a2).all() and (h == h2).all()
    b, hb = O.plane_derivative(d); b2, hb2 = N.plane_derivative(d)
    assert (b == b2).all() and (hb == hb2).all()
    _, hb3 = O.plane_derivative(d, hb.copy())
    assert (hb3 == 2 * hb).all()  # cumulative, planeseg.cu:157
    params = (6, 18, -5, 6, 11, 0)
    pl = O.classify(b, params)
    assert (pl == N.classify(b, params)).all()
    pl2 = rng.integers(0, 3, (25, 31)).astype(np.uint8)
    ids, n = O.ccl(pl2); ids2, n2 = N.ccl(pl2)
    assert (ids == ids2).all() and n == n2
    assert (ids[pl2 == 2] == -1).all()


def test_find_peaks_and_params():
    rng = np.random.default_rng(11)
    for k in range(30):
        hh = rng.integers(0, 40, 256) if k % 2 else rng.integers(0, 5, 256) * rng.integers(0, 3000, 256)
        assert O.find_peaks(hh.astype(np.int32)) == N.find_peaks(hh)
    # two clean triangular peaks: at bin 128 (zero derivative) and bin 139
    h = np.zeros(256, np.int32)
    for i in range(-4, 5):
        h[128 + i] = 1000 - 200 * abs(i)
        h[139 + i] = max(h[139 + i], 600 - 120 * abs(i))
    ok, p = O.histogram_peak_params(h)
    pk = O.find_peaks(h)
    assert pk[0][0] == 128 and pk[0][1] == -1 and pk[1][0] == 139
    assert ok and p[5] == 0 and p[4] == 11
    # valley = first strict minimum scanning 128..138
    valley = 128 + int(np.argmin(h[128:139]))
    assert p[3] == valley - 127 and p[0] == valley - 127
    # early-outs keep the previous parameters
    ok2, p2 = O.histogram_peak_params(np.zeros(256, np.int32), (1, 2, 3, 4, 5, 6))
    assert not ok2 and p2 == (1, 2, 3, 4, 5, 6)


def test_golden_fixtures():
    files = sorted(glob.glob(os.path.join(HERE, "golden", "road_*.npz")))
    assert files, "no golden fixtures committed"
    for f in files:
        z = np.load(f)
        cfg = {k: int(z[k]) for k in ("D", "P", "min_disp", "radius", "iterations")}
        d = O.disparity_module(z["left"], z["right"], cfg["D"], cfg["P"], cfg["min_disp"], radius=cfg["radius"],
                               iterations=cfg["iterations"])
        assert (d == z["disparity"]).all(), f
        dd, hist = O.plane_derivative(d)
        assert (dd == z["plane_derivative"]).all() and (hist == z["plane_hist"]).all(), f
        dir_d, dir_h = O.directional_derivative(d)
        assert (dir_d == z["dir_derivative"]).all() and (dir_h == z["dir_hist"]).all(), f
        ok, pp = O.histogram_peak_params(hist)
        assert tuple(z["plane_params"]) == pp and bool(z["plane_params_ok"]) == ok, f
        pl = O.classify(dd, pp)
        assert (pl == z["planes"]).all(), f
        ids, n = O.ccl(pl)
        assert (ids == z["ccl_ids"]).all() and n == int(z["ccl_n"]), f


# ---- superpixels (S13 / S14) ----------------------------------------------------------------------------------------
def test_ycrcb_and_block_init_known_answers():
    # OpenCV's documented 8-bit BGR2YCrCb values for the primaries (S14)
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [90, 90, 90]]], np.uint8)  # B, G, R, white, black, gray
    want = np.array([[[29, 107, 255], [150, 21, 43], [76, 255, 85], [255, 128, 128], [0, 128, 128], [90, 128, 128]]], np.uint8)
    assert (O.bgr2ycrcb(px) == want).all() and (N.bgr2ycrcb(px) == want).all()
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (13, 17, 3)).astype(np.uint8)
    assert (O.bgr2ycrcb(img) == N.bgr2ycrcb(img)).all()
    # initialization.cu:13-58: 10x7 image, 4x3 blocks -> 3 x 3 blocks, the last ones smaller
    lab, mx = O.sp_block_init(10, 7, 4, 3)
    assert mx == 9 and lab[0, 0] == 0 and lab[0, 9] == 2 and lab[6, 0] == 6 and lab[6, 9] == 8 and lab[2, 3] == 0 and lab[3, 4] == 4


def test_spec_log():
    import math
    assert O.log(1.0) == 0.0
    rng = np.random.default_rng(3)
    for x in np.concatenate([rng.uniform(0.5, 4.0, 300), 10.0 ** rng.uniform(-1, 12, 300), [2.0 * math.pi / 12.0, 2.0, 0.5, math.sqrt(2.0)]]):
        a = O.log(float(x))
        assert a == N.spec_log(float(x))                       # the two restatements agree bit for bit
        assert abs(a - math.log(x)) <= 4e-16 * max(1.0, abs(math.log(x)))   # and the sequence is a ~1 ulp log


@pytest.mark.parametrize("w,h,bs,it,kw", [
    (40, 30, 6, 3, {}),
    (37, 29, 5, 4, dict(progressive=1.0, compactness=0.03)),
    (33, 21, 8, 3, dict(disparity=0.0)),
    (30, 20, 7, 2, dict(image=0.0, compactness=0.0)),
    (26, 18, 4, 3, dict(image=0.0, compactness=0.0, disparity=0.0)),
])
def test_superpixel_relax_matches_python_restatement(w, h, bs, it, kw):
    rng = np.random.default_rng(5 + w)
    base = rng.integers(0, 256, (h // 4 + 2, w // 4 + 2, 3)).astype(np.uint8)
    bgr = np.kron(base, np.ones((4, 4, 1), np.uint8))[:h, :w]
    bgr = np.clip(bgr.astype(int) + rng.integers(-6, 7, bgr.shape), 0, 255).astype(np.uint8)
    yc = O.bgr2ycrcb(bgr)
    d2 = rng.integers(-40, 40, (h, w, 2)).astype(np.int16)
    d2[rng.random((h, w, 2)) < 0.1] = -32768
    lab, mx = O.sp_block_init(w, h, bs, bs)
    a, n = O.sp_relax(O.sp_params(**kw), lab, mx, yc, d2, it)
    b = N.sp_relax(lab, yc, d2, it, **kw)
    assert (a == b).all()
    if kw.get("image", 1) or kw.get("disparity", 1):
        assert n > 0
    # sums over labels are conserved and no label id is invented
    assert a.max() < mx and set(np.unique(a)) <= set(np.unique(lab))


def test_superpixel_relax_properties_and_errors():
    w, h = 48, 36
    lab, mx = O.sp_block_init(w, h, 6, 6)
    yc = np.full((h, w, 3), 77, np.uint8)
    d2 = np.zeros((h, w, 2), np.int16)
    same, n = O.sp_relax(O.sp_params(), lab, mx, yc, d2, 0)
    assert n == 0 and (same == lab).all()
    # a featureless frame: the regular block grid is a fixed point of the sweep (every move only adds clique cost)
    flat, n = O.sp_relax(O.sp_params(), lab, mx, yc, d2, 3)
    assert n == 0 and (flat == lab).all()
    # a pixel of a foreign label in the middle of a block is absorbed by the first sweep
    lab2 = lab.copy(); lab2[3, 3] = lab[3, 20]
    fixed, n = O.sp_relax(O.sp_params(), lab2, mx, yc, d2, 1)
    assert fixed[3, 3] == lab[3, 3]
    with pytest.raises(ValueError):
        O.sp_relax(O.sp_params(), lab, 3, yc, d2, 1)          # labels >= max_label_id
    with pytest.raises(ValueError):
        O.sp_relax(O.sp_params(), lab, mx, yc, None, 1)        # disparity feature without its image
    with pytest.raises(ValueError):
        O.sp_relax(O.sp_params(compactness=-1.0), lab, mx, yc, d2, 1)


def test_superpixel_plane_classify_matches_numpy():
    rng = np.random.default_rng(14)
    w, h = 61, 37
    for n_prev in (0, 2):
        labels, mx = O.sp_block_init(w, h, 7, 5)
        d2 = rng.integers(-12, 30, (h, w, 2)).astype(np.int16)
        d2[rng.random((h, w, 2)) < 0.2] = -32768
        prev = [rng.integers(0, 3, (h, w)).astype(np.uint8) for _ in range(n_prev)]
        flows = [(rng.integers(-40, 40, (h, w, 2)) * rng.integers(1, 64, (h, w, 2))).astype(np.int16) for _ in range(n_prev)]
        params = (6, 22, -4, 6, 14, 1)
        a = O.sp_classify(d2, labels, mx, params, prev, flows)
        b = N.sp_classify(d2, labels, mx, params, prev, flows)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    # known answers (sp_planeseg.cu:145-160): UNKNOWN keeps ties, VERTICAL needs > UNKNOWN, HORIZONTAL needs > max(U, V)
    labels = np.zeros((1, 6), np.uint16)
    for vals, want in (((10, 10, 0, 0, -32768, -32768), 2), ((10, 10, 10, 0, 0, -32768), 0), ((10, 10, 0, 0, 0, -32768), 1),
                       ((10, 10, 10, 0, 0, 0), 1), ((10, 0, -32768, -32768, -32768, -32768), 2)):
        d2 = np.zeros((1, 6, 2), np.int16); d2[0, :, 0] = vals
        _, pl = O.sp_classify(d2, labels, 1, (6, 22, -4, 6, 14, 1))
        assert (pl == want).all(), (vals, want)


def test_golden_superpixel_fixtures():
    files = sorted(glob.glob(os.path.join(HERE, "golden", "sp_*.npz")))
    assert files, "no superpixel golden fixtures committed"
    for f in files:
        z = np.load(f)
        p = O.SpParams(*(float(z["p_" + k]) for k in ("direct", "diagonal", "compactness", "progressive", "image", "disparity")))
        h, w = z["labels0"].shape
        labels, mx = O.sp_block_init(w, h, int(z["block"]), int(z["block"]))
        assert mx == int(z["max_label"])
        for k in range(int(z["frames"])):
            labels, _ = O.sp_relax(p, labels, mx, O.bgr2ycrcb(z[f"image{k}"]), z[f"deriv{k}"], int(z[f"iters{k}"]))
            assert (labels == z[f"labels{k}"]).all(), (f, k)
            uns, pl = O.sp_classify(z[f"deriv{k}"], labels, mx, tuple(int(v) for v in z["plane_params"]))
            assert (uns == z[f"unsmoothed{k}"]).all() and (pl == z[f"planes{k}"]).all(), (f, k)


# ---- optical flow (S15) ---------------------------------------------------------------------------------------------
def test_block_flow_matches_numpy_and_recovers_shifts():
    rng = np.random.default_rng(6)
    w, h = 70, 44
    cur = rng.integers(0, 256, (h, w)).astype(np.uint8)
    prev = rng.integers(0, 256, (h, w)).astype(np.uint8)
    prev[5:30, 8:50] = cur[7:32, 5:47]       # a patch that moved by (u, v) = (-3, +2): prev(p - (u,v)) = cur(p)
    for R, B in ((3, 1), (4, 2)):
        a = O.block_flow(cur, prev, R, B)
        b = N.block_flow(O.census(cur), O.census(prev), R, B)
        assert (a == b).all(), (R, B)
    inner = a[14:24, 16:40]
    assert (inner[..., 0] == -3 * 32).all() and (inner[..., 1] == 2 * 32).all()
    # identical frames -> zero flow everywhere (ties keep (0,0)); a flat image too
    assert (O.block_flow(cur, cur, 3, 2) == 0).all()
    flat = np.full((h, w), 90, np.uint8)
    assert (O.block_flow(flat, flat, 3, 2) == 0).all()


def test_ccl_component_table_known_answers():
    pl = np.full((6, 8), 2, np.uint8)
    pl[1:3, 1:4] = 0          # a 2x3 horizontal-plane block
    pl[4, 2:7] = 1            # a 1x5 vertical-plane strip
    pl[0, 7] = 1              # a single pixel
    ids, n = O.ccl(pl)
    table, n2 = O.ccl_stats(pl, ids)
    assert n == n2 == 3
    assert table.tolist() == [[7, 1, 1, 7, 0, 7, 0], [9, 0, 6, 1, 1, 3, 2], [34, 1, 5, 2, 4, 6, 4]]
    t2, n3 = O.ccl_stats(pl, ids, max_components=2)
    assert n3 == 3 and t2.tolist() == table[:2].tolist()


def test_resize_linear_known_answers():
    """S16 by hand: identity at equal size, exact 2x down-sampling picks the even samples (weights 1 / 0), a 2-pixel ramp
    stretched to 4 gives the quarter-step blend with the last column clamped, and .5 results round to even."""
    img = np.arange(48, dtype=np.uint8).reshape(6, 8)
    assert (O.resize_linear(img, 8, 6) == img).all()
    assert (O.resize_linear(img, 4, 3) == img[::2, ::2]).all()
    row = np.array([[10, 20]], np.uint8)
    assert O.resize_linear(row, 4, 1).tolist() == [[10, 15, 20, 20]]          # src_x = 0, .5, 1, 1.5: (10+20)/2 = 15, then x2 clamps
    assert O.resize_linear(np.array([[1, 2]], np.uint8), 4, 1).tolist() == [[1, 2, 2, 2]]   # 1.5 rounds to even (2)
    assert O.resize_linear(np.array([[2, 3]], np.uint8), 4, 1).tolist() == [[2, 2, 3, 3]]   # 2.5 rounds to even (2)
    rgb = np.stack([img, img + 1, img + 2], axis=-1).astype(np.uint8)
    assert (O.resize_linear(rgb, 4, 3) == rgb[::2, ::2]).all()
