"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.

Bar: bit-exact on every integer image / label / histogram.  Small and medium sizes are compared
stage by stage (census, every path slab, WTA maps, disparity, plane stages); BASELINE.json's full
sizes are covered by one full oracle comparison plus size-independent properties."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O
from cartslam import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_engine(w, h, D, P, md=4, radius=-1, iters=5, inflight=4, plan=None, plan_min_frames=1, **kw):
    from cartslam import Engine
    eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=md, smoothing_radius=radius,
                 smoothing_iterations=iters, max_inflight=inflight, **kw)
    if plan is not None:   # force a launch plan (every plan must give the same bits)
        eng.set_plan(plan, plan_min_frames)
    return eng


CASES = [
    # w, h, D, P, min_disp
    (160, 96, 64, 4, 4),
    (173, 67, 64, 8, 0),      # ragged: nothing is a multiple of a tile
    (200, 120, 128, 8, 4),
    (330, 50, 256, 8, 9),
    (64, 16, 64, 8, 4),       # image narrower than D: every right feature partly out of range
    (97, 131, 128, 4, 64),    # tall, max min_disparity
]


@pytest.mark.parametrize("w,h,D,P,md", CASES)
def test_sgm_stage_by_stage(torch_cuda, w, h, D, P, md):
    torch = torch_cuda
    l, r, _ = synth.make_pair(w, h, D, md, seed=1000 + w + D)
    eng = make_engine(w, h, D, P, md)
    disp = eng.compute_disparity(dev(torch, l), dev(torch, r)).cpu().numpy()
    cl, cr = O.census(l), O.census(r)
    assert (eng.debug_read(0) == l).all() and (eng.debug_read(1) == r).all()
    assert (eng.debug_read(2) == cl).all(), "census left"
    assert (eng.debug_read(3) == cr).all(), "census right"
    S = np.zeros((h, w, D), np.uint16)
    for i in range(P):
        dx, dy = O.path_dir(i)
        L = O.aggregate_path(cl, cr, D, md, 10, 120, dx, dy)
        got = eng.debug_read(16 + i)
        assert (got == L).all(), f"path {i} ({dx},{dy}): {int((got != L).sum())} cells differ"
        S += L
    wl, wr = O.wta(S, 12)
    assert (eng.debug_read(32) == wl).all(), "wta left"
    assert (eng.debug_read(33) == wr).all(), "wta right"
    exp = O.lr_check_range(O.median3x3(wl), O.median3x3(wr), l, md)
    assert (disp == exp).all(), "disparity"
    eng.close()


@pytest.mark.parametrize("w,h,D,P", [(173, 67, 128, 8), (333, 35, 64, 4), (211, 45, 64, 8), (97, 29, 256, 4), (1242, 375, 64, 4),
                                     # heights h with h % (2 x rows per pair) in {1, rows per pair + 1}: the last workgroup's pair(s) hold ONE valid row, every other
                                     # lane group clones it -- two waves store the same bytes to the same cells (sgm_kernels.hip, split-scan row cloning)
                                     (173, 65, 128, 8), (333, 49, 64, 4), (97, 25, 256, 4)])
def test_split_horizontal_scans_equal_plain_ones(torch_cuda, w, h, D, P):
    """The aggregation launch runs its horizontal scans as producer / consumer wave pairs when it would otherwise wait for their W-step
    chains (agg_hsplit: few frames, or few directions), and as plain one-wave scans otherwise.  The same frame alone (split) and inside a
    16-frame batch (plain, for 8 paths at D >= 128; D=64 always splits -- its consumer is the one that stores whole 128-byte lines, two steps per
    store pair -- and both calls are then compared with the oracle only) must leave the
    same horizontal slabs, cell for cell, and the oracle's -- on widths that are no multiple of 16 (the producer's tail steps) and heights
    that leave the last workgroup a partial pair and an idle pair (rows cloned, every wave at every barrier)."""
    torch = torch_cuda
    ls, rs = synth.make_batch(16, w, h, D, 4, scene="stripes")
    eng = make_engine(w, h, D, P, 4, inflight=16, plan="slabs")
    cl, cr = O.census(ls[5]), O.census(rs[5])
    want = {i: O.aggregate_path(cl, cr, D, 4, 10, 120, *O.path_dir(i)) for i in (2, 3)}   # right, left
    eng.compute_disparity(dev(torch, ls[5:6]), dev(torch, rs[5:6]))
    alone = {i: eng.debug_read(16 + i, frame_slot=0) for i in (2, 3)}
    eng.compute_disparity(dev(torch, ls), dev(torch, rs))
    batched = {i: eng.debug_read(16 + i, frame_slot=5) for i in (2, 3)}
    for i in (2, 3):
        assert (alone[i] == want[i]).all(), f"path {i}, one frame: {int((alone[i] != want[i]).sum())} cells differ"
        assert (batched[i] == want[i]).all(), f"path {i}, 16 frames: {int((batched[i] != want[i]).sum())} cells differ"
    eng.close()


def test_bgr_pitched_batched(torch_cuda):
    """BGR input (fused gray conversion), non-tight pitches, a batch of frames, smoothing on."""
    torch = torch_cuda
    w, h, D, P, n = 190, 70, 64, 8, 3
    eng = make_engine(w, h, D, P, 4, radius=2, iters=2, inflight=4)
    ls, rs = synth.make_batch(n, w, h, D, 4, seed=77, channels=3)
    lbuf = torch.zeros((n, h + 3, w + 11, 3), dtype=torch.uint8, device="cuda")
    rbuf = torch.zeros((n, h + 1, w + 5, 3), dtype=torch.uint8, device="cuda")
    lv, rv = lbuf[:, :h, :w], rbuf[:, :h, :w]
    lv.copy_(dev(torch, ls)); rv.copy_(dev(torch, rs))
    obuf = torch.full((n, h + 2, w + 6), 12345, dtype=torch.int16, device="cuda")
    ov = obuf[:, :h, :w]
    eng.compute_disparity(lv, rv, out=ov)
    got = obuf.cpu().numpy()
    for f in range(n):
        exp = O.disparity_module(ls[f], rs[f], D, P, 4, radius=2, iterations=2)
        assert (got[f, :h, :w] == exp).all(), f"frame {f}"
    assert (got[:, h:, :] == 12345).all() and (got[:, :, w:] == 12345).all(), "wrote outside the image"
    eng.close()


def test_engine_reuse_and_params(torch_cuda):
    """Same engine called repeatedly gives identical output (workspace reuse); other P1/P2/uniqueness."""
    torch = torch_cuda
    w, h, D, P = 128, 48, 64, 4
    l, r, _ = synth.make_pair(w, h, D, 4, seed=5)
    eng = make_engine(w, h, D, P, 4, inflight=2, p1=7, p2=90, uniqueness_ratio=5)
    a = eng.compute_disparity(dev(torch, l), dev(torch, r)).cpu().numpy()
    l2, r2, _ = synth.make_pair(w, h, D, 4, seed=6)
    eng.compute_disparity(dev(torch, l2), dev(torch, r2))
    b = eng.compute_disparity(dev(torch, l), dev(torch, r)).cpu().numpy()
    assert (a == b).all()
    assert (a == O.disparity_module(l, r, D, P, 4, p1=7, p2=90, uniq=5)).all()
    eng.close()


def test_edge_inputs(torch_cuda):
    """All-black left image (LR mask everywhere), constant images, extreme contrast."""
    torch = torch_cuda
    w, h, D, P = 96, 40, 64, 8
    eng = make_engine(w, h, D, P, 4)
    rng = np.random.default_rng(0)
    cases = [
        (np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8)),
        (np.full((h, w), 200, np.uint8), np.full((h, w), 200, np.uint8)),
        (rng.integers(0, 2, (h, w)).astype(np.uint8) * 255, rng.integers(0, 2, (h, w)).astype(np.uint8) * 255),
        (rng.integers(0, 256, (h, w)).astype(np.uint8), rng.integers(0, 256, (h, w)).astype(np.uint8)),
    ]
    for i, (l, r) in enumerate(cases):
        got = eng.compute_disparity(dev(torch, l), dev(torch, r)).cpu().numpy()
        assert (got == O.disparity_module(l, r, D, P, 4)).all(), f"case {i}"
    eng.close()


@pytest.mark.parametrize("w,h,D,P", [(16384, 8, 64, 8), (16, 2000, 64, 4), (16, 8, 256, 8), (4099, 11, 128, 8)])
def test_extreme_shapes(torch_cuda, w, h, D, P):
    """The smallest and the most lopsided images the engine accepts (16 x 8 is its minimum, 16384 its widest): one W-step scan
    per row group, 2000-step vertical scans on two lane groups, an image narrower than a sixteenth of its disparity range, and a
    prime width -- single frame and a batch of three, every launch plan, through plane labelling and components."""
    torch = torch_cuda
    rng = np.random.default_rng(w * 31 + h)
    base = rng.integers(0, 256, (h, w + 40)).astype(np.uint8)
    l, r = np.ascontiguousarray(base[:, 20:20 + w]), np.ascontiguousarray(base[:, 27:27 + w])   # a 7-pixel shift: real matches where the width allows
    exp = O.disparity_module(l, r, D, P, 4, radius=2, iterations=1)
    eng = make_engine(w, h, D, P, 4, radius=2, iters=1, inflight=3)
    for plan in ("slabs", "fused_up"):
        eng.set_plan(plan)
        got = eng.compute_disparity(dev(torch, np.stack([l, l, l])), dev(torch, np.stack([r, r, r]))).cpu().numpy()
        assert all((got[k] == exp).all() for k in range(3)), f"{plan}: {int((got[0] != exp).sum())} pixels differ"
        assert (eng.compute_disparity(dev(torch, l), dev(torch, r)).cpu().numpy() == exp).all(), plan
    hist = torch.zeros(256, dtype=torch.int32, device="cuda")
    d = eng.compute_disparity(dev(torch, l), dev(torch, r))
    pd = eng.plane_derivative_hist(d, hist)
    eb, eh = O.plane_derivative(exp)
    assert (pd.cpu().numpy() == eb).all() and (hist.cpu().numpy() == eh).all()
    pp = (6, 18, -5, 6, 11, 0)
    planes = eng.plane_classify(pd, pp)
    ep = O.classify(eb, pp)
    assert (planes.cpu().numpy() == ep).all()
    ids, n = eng.plane_ccl(planes)
    eids, en = O.ccl(ep)
    assert (ids.cpu().numpy() == eids).all() and int(n.item()) == en
    eng.close()


def test_bad_arguments_fail_loudly(torch_cuda):
    torch = torch_cuda
    from cartslam import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(100, 50, num_disparities=100)
    with pytest.raises(EngineError):
        Engine(100, 50, num_disparities=64, paths=5)
    with pytest.raises(EngineError):
        Engine(100, 50, num_disparities=64, p2=230)
    eng = make_engine(100, 50, 64, 4, inflight=2)
    l = torch.zeros((3, 50, 100), dtype=torch.uint8, device="cuda")
    with pytest.raises(EngineError):
        eng.compute_disparity(l, l)  # 3 frames > max_inflight 2
    eng.close()


@pytest.mark.parametrize("r,it", [(1, 1), (2, 1), (3, 2), (2, 5)])
def test_interpolate(torch_cuda, r, it):
    torch = torch_cuda
    rng = np.random.default_rng(r * 7 + it)
    w, h = 211, 77
    d = rng.integers(40, 1400, (2, h, w)).astype(np.int16)
    d[rng.random(d.shape) < 0.3] = -32768
    eng = make_engine(w, h, 64, 4, inflight=2)
    t = dev(torch, d)
    eng.interpolate(t, r, it, 64, w)
    got = t.cpu().numpy()
    for f in range(2):
        assert (got[f] == O.interpolate(d[f], r, it, 64, w)).all()
    eng.close()


def test_plane_stages(torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(21)
    w, h, n = 203, 91, 3
    d = rng.integers(64, 200, (n, h, w)).astype(np.int16)
    d[rng.random(d.shape) < 0.15] = -32768
    d[0, 5, 5] = 32767; d[0, 3, 5] = -32767; d[0, 9, 7] = 32767; d[0, 9, 3] = -32767  # s16 wrap
    eng = make_engine(w, h, 64, 4, inflight=4)
    t = dev(torch, d)
    dd, dh = eng.disparity_derivative(t)
    dd, dh = dd.cpu().numpy(), dh.cpu().numpy()
    hist = torch.zeros(256, dtype=torch.int32, device="cuda")
    hist_pf = torch.zeros((n, 256), dtype=torch.int32, device="cuda")
    pd = eng.plane_derivative_hist(t, hist).cpu().numpy()
    eng.plane_derivative_hist(t, hist_pf, per_frame_hist=True)
    eng.plane_derivative_hist(t, hist)  # cumulative: second call doubles it
    total = np.zeros(256, np.int64)
    params = [(6, 18, -5, 6, 11, 0), (-3, 2, 2, 40, 0, 20), (0, 0, 0, 0, 0, 0)]
    planes_pf = eng.plane_classify(dev(torch, pd), params).cpu().numpy()
    planes_one = eng.plane_classify(dev(torch, pd), params[0]).cpu().numpy()
    ids, ncomp = eng.plane_ccl(dev(torch, planes_pf))
    ids, ncomp = ids.cpu().numpy(), ncomp.cpu().numpy()
    for f in range(n):
        a, h2 = O.directional_derivative(d[f])
        assert (dd[f] == a).all() and (dh[f] == h2).all(), f"dir derivative frame {f}"
        b, hb = O.plane_derivative(d[f])
        assert (pd[f] == b).all(), f"plane derivative frame {f}"
        assert (hist_pf[f].cpu().numpy() == hb).all()
        total += hb
        assert (planes_pf[f] == O.classify(b, params[f])).all()
        assert (planes_one[f] == O.classify(b, params[0])).all()
        eids, en = O.ccl(planes_pf[f])
        assert (ids[f] == eids).all() and ncomp[f] == en, f"ccl frame {f}"
    assert (hist.cpu().numpy() == 2 * total).all()
    eng.close()


def test_geometry_only_engine(torch_cuda):
    """num_disparities = paths = 0: post-stage entry points work, the SGM entry point refuses loudly."""
    torch = torch_cuda
    from cartslam import Engine, EngineError
    w, h = 150, 40
    eng = Engine(w, h, num_disparities=0, paths=0, max_inflight=2)
    rng = np.random.default_rng(4)
    d = rng.integers(64, 300, (h, w)).astype(np.int16)
    hist = torch.zeros(256, dtype=torch.int32, device="cuda")
    pd = eng.plane_derivative_hist(dev(torch, d), hist)
    eb, eh = O.plane_derivative(d)
    assert (pd.cpu().numpy() == eb).all() and (hist.cpu().numpy() == eh).all()
    ids, n = eng.plane_ccl(eng.plane_classify(pd, (6, 18, -5, 6, 12, 0)))
    eids, en = O.ccl(O.classify(eb, (6, 18, -5, 6, 12, 0)))
    assert (ids.cpu().numpy() == eids).all() and int(n.item()) == en
    img = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    with pytest.raises(EngineError):
        eng.compute_disparity(img, img)
    eng.close()


def test_ccl_hard_shapes(torch_cuda):
    """Spirals / combs / checkerboards: long union-find chains and many tiny components."""
    torch = torch_cuda
    w, h = 131, 67
    eng = make_engine(w, h, 64, 4, inflight=4)
    yy, xx = np.mgrid[0:h, 0:w]
    shapes = [
        ((xx + yy) % 2).astype(np.uint8),                       # checkerboard: every pixel its own component
        np.where(yy % 2 == 0, 0, np.where(xx == (yy // 2 % 2) * (w - 1), 0, 1)).astype(np.uint8),  # serpentine
        np.where(xx % 3 == 0, 2, (yy // 5 % 2)).astype(np.uint8),
        np.zeros((h, w), np.uint8),
    ]
    p = dev(torch, np.stack(shapes))
    ids, n = eng.plane_ccl(p)
    ids, n = ids.cpu().numpy(), n.cpu().numpy()
    for f, s in enumerate(shapes):
        e, en = O.ccl(s)
        assert (ids[f] == e).all() and n[f] == en, f"shape {f}"
    eng.close()




def test_golden_fixtures(torch_cuda):
    torch = torch_cuda
    from cartslam import find_plane_params
    files = sorted(glob.glob(os.path.join(HERE, "golden", "road_*.npz")))
    assert files
    for f in files:
        z = np.load(f)
        D, P, md, radius, iters = (int(z[k]) for k in ("D", "P", "min_disp", "radius", "iterations"))
        l, r = z["left"], z["right"]
        h, w = l.shape[:2]
        eng = make_engine(w, h, D, P, md, radius=radius, iters=iters)
        d = eng.compute_disparity(dev(torch, l), dev(torch, r))
        assert (d.cpu().numpy() == z["disparity"]).all(), f
        hist = torch.zeros(256, dtype=torch.int32, device="cuda")
        pd = eng.plane_derivative_hist(d, hist)
        assert (pd.cpu().numpy() == z["plane_derivative"]).all() and (hist.cpu().numpy() == z["plane_hist"]).all(), f
        dd, dh = eng.disparity_derivative(d)
        assert (dd.cpu().numpy() == z["dir_derivative"]).all() and (dh.cpu().numpy() == z["dir_hist"]).all(), f
        ok, pp = find_plane_params(hist.cpu().numpy())
        assert ok == bool(z["plane_params_ok"]) and pp.as_tuple() == tuple(int(v) for v in z["plane_params"]), f
        planes = eng.plane_classify(pd, pp)
        assert (planes.cpu().numpy() == z["planes"]).all(), f
        ids, n = eng.plane_ccl(planes)
        assert (ids.cpu().numpy() == z["ccl_ids"]).all() and int(n.item()) == int(z["ccl_n"]), f
        eng.close()


@pytest.mark.parametrize("w,h,D,P", [(1242, 375, 64, 4), (1242, 375, 128, 8)])
def test_full_size_against_oracle(torch_cuda, w, h, D, P):
    """BASELINE.json configs 2 and 3 at full size: whole disparity module + plane labelling, bit-exact."""
    torch = torch_cuda
    l, r, gt = synth.make_pair(w, h, D, 4)
    eng = make_engine(w, h, D, P, 4, radius=2, iters=1, inflight=2)
    d = eng.compute_disparity(dev(torch, l), dev(torch, r))
    exp = O.disparity_module(l, r, D, P, 4, radius=2, iterations=1)
    got = d.cpu().numpy()
    assert (got == exp).all(), f"{int((got != exp).sum())} pixels differ"
    valid = got != -32768
    assert valid.mean() > 0.8 and np.median(np.abs(got[valid] / 16.0 - gt[valid])) < 0.5  # it is a disparity map
    hist = torch.zeros(256, dtype=torch.int32, device="cuda")
    pd = eng.plane_derivative_hist(d, hist)
    eb, eh = O.plane_derivative(exp)
    assert (pd.cpu().numpy() == eb).all() and (hist.cpu().numpy() == eh).all()
    assert int(hist.sum()) <= w * h
    # ... and on through the plane-parameter provider, classification, connected components and the component table
    ok, pp = O.histogram_peak_params(eh)
    assert ok, "the synthetic scene must give two histogram peaks"
    planes = eng.plane_classify(pd, pp)
    ep = O.classify(eb, pp)
    assert (planes.cpu().numpy() == ep).all(), "planes"
    assert {0, 1, 2} <= set(np.unique(ep).tolist()), "the label map should hold all three classes"
    ids, ncomp = eng.plane_ccl(planes)
    eids, en = O.ccl(ep)
    assert (ids.cpu().numpy() == eids).all() and int(ncomp.item()) == en, "component ids / count"
    cap = 1 << 15
    table, n2 = eng.plane_ccl_stats(planes, ids, max_components=cap)
    et, _ = O.ccl_stats(ep, eids, max_components=cap)
    assert int(n2.item()) == en and en <= cap and (table.cpu().numpy().reshape(-1, 7)[:len(et)] == et).all(), "component table"
    eng.close()


@pytest.mark.parametrize("scene", ["stripes", "saturated", "pole", "wall", "photometric"])
def test_full_size_scene_content_against_oracle(torch_cuda, scene):
    """The headline configuration (1242x375, D=128, 8 paths) on scenes with the content street images have and value noise
    does not -- an exactly periodic striped facade (equal-cost candidates every 8 / 16 / 24 px), a saturated and a BLACK
    patch (gray == 0 is the LR check's mask, oracle S8), 1-px / 3-px poles, a large textureless wall, a right camera with its
    own gain, offset and noise -- whole frame through
    classification, components and the component table, every launch plan, bit-exact.  (No KITTI data is available offline;
    this is the nearest substitute.)"""
    torch = torch_cuda
    w, h, D, P = 1242, 375, 128, 8
    l, r, _ = synth.make_pair(w, h, D, 4, scene=scene)
    if scene == "saturated":
        assert (l == 0).sum() > 20000 and (l == 255).sum() > 10000
    exp = O.disparity_module(l, r, D, P, 4, radius=2, iterations=1)
    raw = O.sgm(l, r, D, P, 4)
    regions = synth._scene_regions(w, h, scene, 0)
    inv = raw == 3 * 16
    if scene == "saturated":   # the black patch is masked out completely, whatever the matching says
        y0, y1, x0, x1 = regions[1]
        assert inv[y0:y1, x0:x1].all()
    if scene == "wall":        # a textureless wall is mostly rejected (uniqueness / LR check), its edges are not
        y0, y1, x0, x1 = regions[0]
        assert 0.5 < inv[y0:y1, x0:x1].mean() < 1.0
    eng = make_engine(w, h, D, P, 4, radius=2, iters=1, inflight=4)
    L, R = dev(torch, np.stack([l, l])), dev(torch, np.stack([r, r]))
    for plan in ("slabs", "fused_up"):
        eng.set_plan(plan)
        got = eng.compute_disparity(L, R).cpu().numpy()
        assert (got[0] == exp).all() and (got[1] == exp).all(), f"{plan}: {int((got[0] != exp).sum())} pixels differ"
    eng.set_plan("auto")
    d = eng.compute_disparity(dev(torch, l), dev(torch, r))
    hist = torch.zeros(256, dtype=torch.int32, device="cuda")
    pd = eng.plane_derivative_hist(d, hist)
    eb, eh = O.plane_derivative(exp)
    assert (pd.cpu().numpy() == eb).all() and (hist.cpu().numpy() == eh).all()
    ok, pp = O.histogram_peak_params(eh)
    if not ok:
        pp = (6, 18, -5, 6, 11, 0)
    planes = eng.plane_classify(pd, pp)
    ep = O.classify(eb, pp)
    assert (planes.cpu().numpy() == ep).all(), "planes"
    ids, ncomp = eng.plane_ccl(planes)
    eids, en = O.ccl(ep)
    assert (ids.cpu().numpy() == eids).all() and int(ncomp.item()) == en, "component ids / count"
    cap = 1 << 15
    table, n2 = eng.plane_ccl_stats(planes, ids, max_components=cap)
    et, _ = O.ccl_stats(ep, eids, max_components=cap)
    assert int(n2.item()) == en and en <= cap and (table.cpu().numpy().reshape(-1, 7)[:len(et)] == et).all(), "component table"
    eng.close()


@pytest.mark.parametrize("w,h", [(1241, 376), (1226, 370), (1224, 370)])
def test_native_kitti_frame_sizes(torch_cuda, w, h):
    """The other frame sizes KITTI sequences come in (odometry 00-02: 1241x376, 03: 1242x375, 04-10: 1226x370; raw drives
    1224x370): the reference resizes to its configured size, but an engine configured for the native size must be just as
    exact -- D=128, 8 paths, every launch plan, a batch of two (the widths are odd or leave ragged tiles everywhere)."""
    torch = torch_cuda
    D, P = 128, 8
    l, r, _ = synth.make_pair(w, h, D, 4, seed=w * 7 + h, scene="pole")
    exp = O.disparity_module(l, r, D, P, 4, radius=2, iterations=1)
    eng = make_engine(w, h, D, P, 4, radius=2, iters=1, inflight=2)
    for plan in ("slabs", "fused_up"):
        eng.set_plan(plan)
        got = eng.compute_disparity(dev(torch, np.stack([l, l])), dev(torch, np.stack([r, r]))).cpu().numpy()
        assert (got[0] == exp).all() and (got[1] == exp).all(), f"{plan}: {int((got[0] != exp).sum())} pixels differ"
    eng.close()


def test_full_size_properties_1080p_d256(torch_cuda):
    """BASELINE config 4 (1920x1080, D=256, 8 paths): too slow for a full oracle run in a test, so
    size-independent properties: slab bounds, path starts equal the matching cost, batch == single,
    determinism, and an oracle check of a full-width strip of rows for the horizontal paths."""
    torch = torch_cuda
    w, h, D, P, md = 1920, 1080, 256, 8, 4
    l, r, _ = synth.make_pair(w, h, D, md)
    eng = make_engine(w, h, D, P, md, inflight=2)
    tl, tr = dev(torch, l), dev(torch, r)
    a = eng.compute_disparity(tl, tr).cpu().numpy()
    cl, cr = eng.debug_read(2), eng.debug_read(3)
    assert (cl == O.census(l)).all() and (cr == O.census(r)).all()
    # horizontal paths are row-local: oracle on a strip of rows must equal the same rows of the slabs
    ys = slice(500, 508)
    for i in (2, 3):
        dx, dy = O.path_dir(i)
        L = O.aggregate_path(cl[ys], cr[ys], D, md, 10, 120, dx, dy)
        assert (eng.debug_read(16 + i)[ys] == L).all(), f"path {i}"
    # vertical/diagonal: first row of a down path == matching cost == first row of any path started there
    down = eng.debug_read(16 + 0)
    C0 = O.aggregate_path(cl[:1], cr[:1], D, md, 0, 0, 0, 1)  # P1=P2=0 on one row -> plain cost
    assert (down[0] == C0[0]).all()
    assert down.max() <= 151
    del down
    both = eng.compute_disparity(torch.stack([tl, tl]), torch.stack([tr, tr])).cpu().numpy()
    assert (both[0] == a).all() and (both[1] == a).all()
    assert ((a >= (md - 1) * 16) & (a < (md + D) * 16)).all()
    eng.close()


def test_batched_pipeline_matches_frame_by_frame_oracle(torch_cuda):
    """Whole hot path through the batched driver (disparity -> plane derivative + histogram -> plane-parameter
    schedule -> classify -> CCL) against the oracle fed the same frames one at a time in id order, i.e. what the
    reference's two modules would produce (disparity.cu:49-80, planeseg.cu:246-403)."""
    torch = torch_cuda
    from cartslam.pipeline import StereoPipeline
    w, h, D, P, n = 256, 96, 64, 8, 7
    ui, ri = 3, 2
    eng = make_engine(w, h, D, P, 4, radius=2, iters=1, inflight=8)
    ls, rs = synth.make_batch(n, w, h, D, 4, seed=99)
    for device_schedule in (True, False):
        _pipeline_case(torch, eng, ls, rs, w, h, D, P, n, ui, ri, device_schedule)
    # two-stream pipelining of consecutive batches (plane stages of batch i on a side stream under the disparity of i+1)
    _pipeline_case(torch, eng, ls, rs, w, h, D, P, n, ui, ri, True, overlap=True)
    eng.close()


def _pipeline_case(torch, eng, ls, rs, w, h, D, P, n, ui, ri, device_schedule, overlap=False):
    from cartslam.pipeline import StereoPipeline
    pipe = StereoPipeline(eng, provider="histogram_peak", update_interval=ui, reset_interval=ri, with_ccl=True,
                          device_schedule=device_schedule, overlap=overlap)
    inputs = [(dev(torch, ls[a:a + 4]), dev(torch, rs[a:a + 4])) for a in range(0, n, 4)]  # batches of 4, then 3
    raw = [pipe.process_batch(l, r) for l, r in inputs]  # enqueued back to back, nothing read in between
    torch.cuda.synchronize()
    outs = [{k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in o.items()} for o in raw]
    disp = np.concatenate([o["disparity"] for o in outs]); planes = np.concatenate([o["planes"] for o in outs])
    ids = np.concatenate([o["ids"] for o in outs]); ncomp = np.concatenate([o["n_components"] for o in outs])
    comps = np.concatenate([o["components"] for o in outs])
    cum = np.zeros(256, np.int64)
    params = (0, 0, 0, 0, 0, 0)
    for f in range(n):
        fid = f + 1
        ed = O.disparity_module(ls[f], rs[f], D, P, 4, radius=2, iterations=1)
        assert (disp[f] == ed).all(), f"disparity frame {fid}"
        dd, hist = O.plane_derivative(ed)
        cum += hist
        if fid % ui == 1:
            h32 = cum.astype(np.int32)
            if fid % (ui * ri) == 1:
                cum[:] = 0
            _, params = O.histogram_peak_params(h32, params)
        ep = O.classify(dd, params)
        assert (planes[f] == ep).all(), f"planes frame {fid}"
        eids, en = O.ccl(ep)
        assert (ids[f] == eids).all() and ncomp[f] == en, f"ccl frame {fid}"
        et, _ = O.ccl_stats(ep, eids, max_components=4096)
        assert (comps[f][:len(et)] == et).all(), f"component table frame {fid}"


def test_device_plane_schedule_matches_host_restatement(torch_cuda):
    """cart_plane_schedule_advance (one small kernel, no host round trip) against the host restatement that is itself
    checked against the sequential reference order in tests/test_distributed.py."""
    torch = torch_cuda
    from cartslam import DevicePlaneSchedule
    from cartslam.pipeline import PlaneParameterSchedule
    from test_distributed import make_hists
    eng = make_engine(64, 32, 64, 4)
    rng = np.random.default_rng(17)
    extra = [rng.integers(0, 40, 256), rng.integers(0, 5, 256) * rng.integers(0, 3000, 256), np.zeros(256), np.full(256, 7)]
    hists = np.concatenate([make_hists(70), np.stack(extra).astype(np.int32)])
    for ui, ri in ((30, 10), (7, 2), (3, 3)):
        host = PlaneParameterSchedule("histogram_peak", update_interval=ui, reset_interval=ri)
        devs = DevicePlaneSchedule(eng, "histogram_peak", None, ui, ri)
        a = 0
        for n in (16, 5, 1, 30, 22):
            exp = [p.as_tuple() for p in host.advance(a + 1, hists[a:a + n])]
            got = devs.advance(a + 1, dev(torch, hists[a:a + n])).cpu().numpy()
            assert [tuple(int(v) for v in row) for row in got] == exp, (ui, ri, a)
            a += n
        p, cum = devs.read()
        assert p.as_tuple() == host.params.as_tuple() and (cum == host.cum.astype(np.int32)).all()
        devs.close()
    st = DevicePlaneSchedule(eng, "static", (6, 18, -5, 6, 12, 0))
    assert (st.advance(1, dev(torch, hists[:3])).cpu().numpy() == np.array([(6, 18, -5, 6, 12, 0)] * 3)).all()
    eng.close()


def test_reproject_depth_batched(torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(5)
    w, h, n = 157, 61, 3
    d = rng.integers(64, 1200, (n, h, w)).astype(np.int16)
    d[rng.random(d.shape) < 0.1] = -32768
    Q = O.kitti_q_matrix([718.856, 0, 607.1928, 45.38225, 0, 718.856, 185.2157, -0.113, 0, 0, 1, 0.0037],
                         [718.856, 0, 607.1928, -337.2877, 0, 718.856, 185.2157, 2.369, 0, 0, 1, 0.0049])
    eng = make_engine(w, h, 64, 4, inflight=4)
    got = eng.reproject_depth(dev(torch, d), Q).cpu().numpy()
    for f in range(n):
        exp = O.reproject_depth(d[f], Q)
        assert np.allclose(got[f], exp, rtol=1e-4, atol=1e-4)  # tolerance of the float path (north_star: 1e-4)
    eng.close()


@pytest.mark.parametrize("plan", ["slabs", "fused_up"])
def test_randomized_configurations(torch_cuda, plan):
    """Seeded sweep over sizes / D / paths / min_disparity / P1 / P2 / uniqueness / smoothing, incl. the extremes
    (uniqueness 0 and 100, min_disparity 0 and 64, width < D, 1-pixel-ragged tiles); every output bit-exact -- once
    per launch plan, each forced for every call."""
    torch = torch_cuda
    rng = np.random.default_rng(20260101)
    cases = []
    for i in range(28):
        D = int(rng.choice([64, 128, 256]))
        P = int(rng.choice([4, 8]))
        w = int(rng.integers(16, 420)); h = int(rng.integers(8, 130))
        md = int(rng.choice([0, 1, 4, 17, 64]))
        p1 = int(rng.integers(0, 30)); p2 = int(rng.integers(p1, 224))
        uniq = int(rng.choice([0, 5, 12, 50, 99, 100]))
        radius = int(rng.choice([-1, 1, 2, 3])); iters = int(rng.choice([1, 2, 5]))
        ch = int(rng.choice([1, 3]))
        cases.append((w, h, D, P, md, p1, p2, uniq, radius, iters, ch))
    cases += [(16, 8, 64, 8, 0, 10, 120, 12, 2, 1, 1), (65, 17, 128, 8, 4, 10, 120, 100, -1, 1, 1), (257, 9, 256, 4, 64, 0, 0, 0, 3, 2, 3)]
    for k, (w, h, D, P, md, p1, p2, uniq, radius, iters, ch) in enumerate(cases):
        if k % 3 == 2:  # pure noise images: no structure, many ties / invalid pixels
            l = rng.integers(0, 256, (h, w) if ch == 1 else (h, w, 3)).astype(np.uint8)
            r = rng.integers(0, 256, l.shape).astype(np.uint8)
        else:   # the road scene and, in turn, its variants (stripes, saturated / black patches, poles, a textureless wall)
            l, r, _ = synth.make_pair(w, h, D, md, seed=1000 + k, channels=ch, scene=synth.SCENES[(k // 3) % len(synth.SCENES)])
        eng = make_engine(w, h, D, P, md, radius=radius, iters=iters, inflight=2, plan=plan, p1=p1, p2=p2, uniqueness_ratio=uniq)
        got = eng.compute_disparity(dev(torch, l), dev(torch, r)).cpu().numpy()
        exp = O.disparity_module(l, r, D, P, md, p1=p1, p2=p2, uniq=uniq, radius=radius, iterations=iters)
        assert (got == exp).all(), f"case {k}: {(w, h, D, P, md, p1, p2, uniq, radius, iters, ch)}: {int((got != exp).sum())} pixels differ"
        eng.close()


def test_temporal_vote(torch_cuda):
    """classifyPlanes' temporal branch (planeseg.cu:199-240) with random planes / S10.5 flows, 0..3 previous frames,
    flows large enough to leave the image and negative flows (arithmetic >> 5)."""
    torch = torch_cuda
    rng = np.random.default_rng(33)
    w, h = 139, 53
    eng = make_engine(w, h, 64, 4)
    for n_prev in (0, 1, 2, 3):
        planes = rng.integers(0, 3, (h, w)).astype(np.uint8)
        prev = [rng.integers(0, 3, (h, w)).astype(np.uint8) for _ in range(n_prev)]
        flows = [(rng.integers(-40, 40, (h, w, 2)) * rng.integers(1, 64, (h, w, 2))).astype(np.int16) for _ in range(n_prev)]
        got = eng.plane_temporal_vote(dev(torch, planes), [dev(torch, p) for p in prev], [dev(torch, f) for f in flows]).cpu().numpy()
        assert (got == O.temporal_vote(planes, prev, flows)).all(), n_prev
    eng.close()


def test_engine_lifecycle_and_bad_arguments(torch_cuda):
    """Create/destroy must not leak device memory; malformed pitches / pointers are refused with a message."""
    torch = torch_cuda
    from cartslam import EngineError
    l = torch.zeros((96, 320), dtype=torch.uint8, device="cuda")

    def cycle(n):
        for _ in range(n):
            eng = make_engine(320, 96, 128, 8, inflight=3)
            eng.compute_disparity(l, l)
            torch.cuda.synchronize()
            eng.close()

    cycle(2)  # first use loads code objects and grows the runtime's own pools
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    cycle(8)
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 16 << 20, f"leak: {free0 - free1} bytes over 8 create/destroy cycles"
    eng = make_engine(320, 96, 64, 4, inflight=2)
    with pytest.raises(EngineError, match="does not match"):
        eng.compute_disparity(l[:, :300], l[:, :300])              # image narrower than the engine's width
    with pytest.raises(EngineError, match="contiguous"):
        eng.compute_disparity(l, l, out=torch.zeros((96, 640), dtype=torch.int16, device="cuda")[:, ::2])  # non-contiguous rows
    with pytest.raises(EngineError):
        eng.compute_disparity(l.cpu(), l.cpu())                     # host tensors are not device memory
    with pytest.raises(EngineError):
        eng.interpolate(torch.zeros((96, 320), dtype=torch.int16, device="cuda"), 9, 1, 64, 320)  # radius > 8
    eng.close()


def test_concurrent_host_threads_one_engine(torch_cuda):
    """The reference enters one module object for up to 12 frames at once (include/cartslam.hpp:4-5): 8 host threads,
    each on its own stream, share one engine with 4 workspace slots; every result must be bit-exact."""
    torch = torch_cuda
    import threading
    w, h, D, P = 256, 80, 64, 8
    eng = make_engine(w, h, D, P, 4, radius=2, iters=1, inflight=4)
    pairs = [synth.make_pair(w, h, D, 4, seed=300 + i)[:2] for i in range(8)]
    expected = [O.disparity_module(l, r, D, P, 4, radius=2, iterations=1) for l, r in pairs]
    results, errors = [None] * 8, []

    def work(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(3):
                    out = eng.compute_disparity(dev(torch, pairs[i][0]), dev(torch, pairs[i][1]))
                s.synchronize()
            results[i] = out.cpu().numpy()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(8):
        assert (results[i] == expected[i]).all(), f"thread {i}"
    eng.close()


FUSED_CASES = [
    # w, h, D, P, min_disp, frames
    (160, 96, 64, 4, 4, 2),
    (173, 67, 64, 8, 0, 3),       # ragged: the last block holds invalid columns, odd height
    (200, 120, 128, 8, 4, 2),
    (330, 50, 256, 8, 9, 2),
    (64, 16, 64, 8, 4, 1),        # narrower than D, one block
    (97, 131, 128, 4, 64, 2),
    (257, 33, 128, 8, 4, 9),      # a wave of entirely invalid columns (257 = 8*32 + 1), batch above the default threshold
    (1700, 12, 256, 8, 4, 2),     # >= 1600 wide at D=256: the sweep's blocks are 8 waves = 32 columns (last block: 4 valid columns)
    (1605, 9, 256, 4, 11, 1),     # the same with 4 paths
]


@pytest.mark.parametrize("w,h,D,P,md,n", FUSED_CASES)
def test_fused_wta_path(torch_cuda, w, h, D, P, md, n):
    """Batches take the WTA kernel that computes the "up" direction on the fly (its slab is never written): the
    remaining slabs, both WTA maps and the disparity must still equal the oracle's, for every frame of the batch."""
    torch = torch_cuda
    eng = make_engine(w, h, D, P, md, inflight=max(n, 2), plan="fused_up")
    ls, rs = synth.make_batch(n, w, h, D, md, seed=4000 + w)
    disp = eng.compute_disparity(dev(torch, ls), dev(torch, rs)).cpu().numpy()
    for f in range(n):
        cl, cr = O.census(ls[f]), O.census(rs[f])
        S = np.zeros((h, w, D), np.uint16)
        for i in range(P):
            dx, dy = O.path_dir(i)
            L = O.aggregate_path(cl, cr, D, md, 10, 120, dx, dy)
            if (dx, dy) != (0, -1):
                assert (eng.debug_read(16 + i, frame_slot=f) == L).all(), f"frame {f} path {i}"
            S += L
        wl, wr = O.wta(S, 12)
        assert (eng.debug_read(32, frame_slot=f) == wl).all(), f"frame {f} wta left: {int((eng.debug_read(32, frame_slot=f) != wl).sum())} differ"
        assert (eng.debug_read(33, frame_slot=f) == wr).all(), f"frame {f} wta right"
        exp = O.lr_check_range(O.median3x3(wl), O.median3x3(wr), ls[f], md)
        assert (disp[f] == exp).all(), f"frame {f} disparity"
    eng.close()


def test_launch_plans_agree_at_full_size(torch_cuda):
    torch = torch_cuda
    w, h, D, P, n = 1242, 375, 128, 8, 8
    ls, rs = synth.make_batch(2, w, h, D, 4, seed=31)
    L = dev(torch, np.concatenate([ls] * 4)); R = dev(torch, np.concatenate([rs] * 4))
    eng = make_engine(w, h, D, P, 4, radius=2, iters=1, inflight=n, plan="slabs")
    assert eng.describe_plan(n) == {"frames_per_launch": n, "plan": "slabs", "slabs_written": 8}
    a = eng.compute_disparity(L, R).cpu().numpy()
    for plan, slabs in (("fused_up", 7),):
        eng.set_plan(plan)
        assert eng.describe_plan(n) == {"frames_per_launch": n, "plan": plan, "slabs_written": slabs}
        b = eng.compute_disparity(L, R).cpu().numpy()
        assert (a == b).all(), plan
    eng.close()
    assert (a[0] == a[2]).all() and (a[0] != a[1]).any()
    assert (a[0] == O.disparity_module(ls[0], rs[0], D, P, 4, radius=2, iterations=1)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,D,P,n", [(1242, 375, 64, 4, 16), (1242, 375, 256, 4, 16), (1920, 1080, 256, 8, 4)])
def test_launch_plans_agree_at_the_other_baseline_sizes(torch_cuda, w, h, D, P, n):
    """configs[1], the reference's default configuration and configs[3] at the bench's batch sizes: the plan AUTO picks, the
    forced SLABS and the forced FUSED_UP give the same bits (16 frames per launch exercise the residency cap of the aggregation
    launch and the u16 right-view rows of the sweep at full width); frame 0 equals the oracle where it finishes in seconds."""
    torch = torch_cuda
    ls, rs = synth.make_batch(2, w, h, D, 4, seed=77)
    L = dev(torch, np.concatenate([ls] * (n // 2))); R = dev(torch, np.concatenate([rs] * (n // 2)))
    eng = make_engine(w, h, D, P, 4, radius=2, iters=1, inflight=n)
    auto = eng.describe_plan(n)["plan"]
    assert auto == ("fused_up" if D == 256 else "slabs")
    a = eng.compute_disparity(L, R).cpu().numpy()
    for plan in ("slabs", "fused_up"):
        eng.set_plan(plan)
        b = eng.compute_disparity(L, R).cpu().numpy()
        assert (a == b).all(), plan
    eng.close()
    assert (a[0] == a[2]).all() and (a[0] != a[1]).any()
    if w * h * D <= 1242 * 375 * 256:
        assert (a[0] == O.disparity_module(ls[0], rs[0], D, P, 4, radius=2, iterations=1)).all()


def test_tile_loaders_on_the_reference_ramp(torch_cuda):
    """The only "test vector" in the reference tree is the 1280x720 coordinate ramp v = y*1280 + x that its debug kernel
    feeds to copyToShared (src/utils/sanity_check.cu:57-65): every element identifies its own position, so a
    mis-addressed tile or halo load shows up.  Every kernel of the build that stages tiles or reads neighbourhoods runs
    on ramp-derived inputs of that size against the oracle."""
    torch = torch_cuda
    w, h = 1280, 720
    yy, xx = np.mgrid[0:h, 0:w]
    ramp = (yy * 1280 + xx).astype(np.int64)
    # disparity-like s16 image: position-unique inside every 1100-wide window, valid for the post filter (64 < v < 1280)
    disp = (ramp % 1100 + 80).astype(np.int16)
    disp[(ramp % 97) == 0] = -32768
    eng = make_engine(w, h, 64, 4, 4, inflight=1)
    d = dev(torch, disp)
    for r, it in ((2, 1), (4, 2)):
        got = eng.interpolate(d.clone(), r, it, 64, 1280).cpu().numpy()
        assert (got == O.interpolate(disp, r, it, 64, 1280)).all(), (r, it)
    hist = torch.zeros(256, dtype=torch.int32, device="cuda")
    pd = eng.plane_derivative_hist(d, hist)
    epd, eh = O.plane_derivative(disp)
    assert (pd.cpu().numpy() == epd).all() and (hist.cpu().numpy() == eh).all()
    dd, dh = eng.disparity_derivative(d)
    edd, edh = O.directional_derivative(disp)
    assert (dd.cpu().numpy() == edd).all() and (dh.cpu().numpy() == edh).all()
    # census tile + halo: gray = a position hash (the ramp itself is monotone along x: every comparison would be equal)
    gray = (((ramp * 2654435761) >> 13) & 255).astype(np.uint8)
    eng.compute_disparity(dev(torch, gray), dev(torch, np.roll(gray, -7, axis=1)))
    assert (eng.debug_read(2) == O.census(gray)).all()
    assert (eng.debug_read(3) == O.census(np.roll(gray, -7, axis=1))).all()
    eng.close()
    # superpixel label tile + halo
    from cartslam import Superpixels
    geo = make_engine(w, h, 0, 0)
    sp = Superpixels(geo, block_size=16)
    bgr = np.stack([gray, np.roll(gray, 3, axis=0), np.roll(gray, 5, axis=1)], axis=-1)
    got = sp.relax(dev(torch, bgr), dev(torch, edd), 2).cpu().numpy().view(np.uint16)
    lab, mx = O.sp_block_init(w, h, 16, 16)
    want, changes = O.sp_relax(O.sp_params(), lab, mx, O.bgr2ycrcb(bgr), edd, 2)
    assert changes > 0 and (got == want).all()
    sp.close(); geo.close()


def test_optical_flow_block_matching(torch_cuda):
    """cart_optical_flow (stand-in for the reference's NVIDIA hardware flow, oracle S15): census block matching, every
    flow vector bit-exact, incl. borders, BGR input, ragged sizes and frames without texture."""
    torch = torch_cuda
    for (w, h, R, B, ch) in ((200, 80, 6, 2, 1), (131, 53, 4, 1, 3), (64, 16, 8, 3, 1), (333, 41, 16, 2, 1)):
        eng = make_engine(w, h, 0, 0)
        cur, _, _ = synth.make_pair(w, h, 64, 4, seed=50 + w, frame=1, channels=ch)
        prev, _, _ = synth.make_pair(w, h, 64, 4, seed=50 + w, frame=0, channels=ch)   # scene translated by 2 px per frame
        got = eng.optical_flow(dev(torch, cur), dev(torch, prev), R, B).cpu().numpy()
        gc = cur if ch == 1 else O.bgr2gray(cur)
        gp = prev if ch == 1 else O.bgr2gray(prev)
        exp = O.block_flow(gc, gp, R, B)
        assert (got == exp).all(), (w, h, R, B, ch, int((got != exp).any(axis=-1).sum()))
        assert (exp != 0).any()
        flat = np.full(cur.shape, 77, np.uint8)
        assert (eng.optical_flow(dev(torch, flat), dev(torch, flat), R, B).cpu().numpy() == 0).all()
        eng.close()
    # full KITTI size against the oracle + recovery of a known global shift in the interior
    w, h, R, B = 1242, 375, 8, 2
    eng = make_engine(w, h, 0, 0)
    cur, _, _ = synth.make_pair(w, h, 128, 4, seed=9)
    prev = np.roll(cur, (-3, 5), axis=(0, 1))          # prev(p - (u,v)) = cur(p) with (u,v) = (-5, 3)
    got = eng.optical_flow(dev(torch, cur), dev(torch, prev), R, B).cpu().numpy()
    exp = O.block_flow(cur, prev, R, B)
    assert (got == exp).all()
    inner = got[20:-20, 20:-20]
    assert ((inner[..., 0] == -5 * 32) & (inner[..., 1] == 3 * 32)).mean() > 0.999
    with pytest.raises(Exception):
        eng.optical_flow(dev(torch, cur), dev(torch, prev), 17, 2)
    eng.close()


def test_ccl_component_table(torch_cuda):
    """cart_plane_ccl_stats (per-component label, area, bounding box) against the oracle: random maps,
    hard shapes, a batch, runs longer than a wave, and truncation at max_components."""
    torch = torch_cuda
    rng = np.random.default_rng(77)
    w, h = 300, 70
    eng = make_engine(w, h, 0, 0, inflight=3)
    maps = [rng.integers(0, 3, (h, w)).astype(np.uint8),
            np.kron(rng.integers(0, 3, (h // 7, w // 10)), np.ones((7, 10), int)).astype(np.uint8),
            np.zeros((h, w), np.uint8)]
    maps[2][::2, :] = 1                      # full-width runs (many 64-pixel pieces per run)
    maps[2][:, 150] = 0                      # ... joined by one column
    pl = dev(torch, np.stack(maps))
    ids, n = eng.plane_ccl(pl)
    table, n2 = eng.plane_ccl_stats(pl, ids, max_components=8192)
    assert (n.cpu().numpy() == n2.cpu().numpy()).all()
    for f in range(3):
        eids, en = O.ccl(maps[f])
        et, en2 = O.ccl_stats(maps[f], eids)
        assert en == en2 == int(n2[f])
        got = table[f, :en].cpu().numpy()
        assert (got == et).all(), f"frame {f}: {int((got != et).any(axis=1).sum())} rows differ"
        assert got[:, 2].sum() == (maps[f] < 2).sum()
    # truncation keeps the first rows and still reports the true count
    table_s, n3 = eng.plane_ccl_stats(pl[:1], ids[:1], max_components=5)
    et, en = O.ccl_stats(maps[0], O.ccl(maps[0])[0], max_components=5)
    assert int(n3[0]) == en and (table_s[0].cpu().numpy() == et).all()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("channels,n", [(1, 5), (3, 19)])
def test_multi_equals_batch(channels, n):
    """cart_compute_disparity_multi (frames in separate allocations, here with a padded row step and shuffled order in
    memory) gives the batch entry point's result bit for bit; 19 frames cross the 16-frame launch sequence."""
    import torch
    from cartslam import Engine, EngineError
    w, h, D = 320, 96, 64
    eng = Engine(w, h, num_disparities=D, paths=8, smoothing_radius=2, smoothing_iterations=1, max_inflight=n)
    ls, rs = synth.make_batch(n, w, h, D, 4)
    if channels == 3:
        ls = np.stack([ls, ls // 2 + 3, 255 - ls], -1); rs = np.stack([rs, rs // 2 + 3, 255 - rs], -1)
    L, R = torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda()
    want = eng.compute_disparity(L, R).cpu().numpy()
    pad = (0, 0, 0, 24) if channels == 3 else (0, 24)   # widen the row step
    order = np.random.default_rng(5).permutation(n)
    store = {int(f): (torch.nn.functional.pad(L[f], pad), torch.nn.functional.pad(R[f], pad), torch.full((h, w + 8), -7, dtype=torch.int16, device="cuda")) for f in order}
    lefts = [store[f][0][:, :w] for f in range(n)]; rights = [store[f][1][:, :w] for f in range(n)]; outs = [store[f][2][:, :w] for f in range(n)]
    eng.compute_disparity_multi(lefts, rights, outs)
    torch.cuda.synchronize()
    for f in range(n):
        assert np.array_equal(outs[f].cpu().numpy(), want[f]), f
        assert (store[f][2][:, w:] == -7).all()   # nothing written past the row
    with pytest.raises(EngineError):
        eng.compute_disparity_multi(lefts, rights[:-1])


@pytest.mark.gpu
def test_plane_multi_equals_batch():
    """cart_plane_derivative_hist_multi + cart_plane_classify_multi on 21 separately allocated frames (two launch
    sequences) = the strided entry points on the same frames: derivative images, cumulative histogram, labels."""
    import torch
    from cartslam import Engine
    w, h, n = 333, 77, 21
    eng = Engine(w, h, num_disparities=64, paths=4, max_inflight=n)
    rng = np.random.default_rng(11)
    disp = rng.integers(40, 1200, (n, h, w)).astype(np.int16)
    disp[rng.random((n, h, w)) < 0.05] = -32768
    disp = np.sort(disp, axis=1)[:, ::-1].copy()   # mostly monotone columns: derivatives land inside the histogram range
    D = torch.from_numpy(disp).cuda()
    params = [(2 + f % 3, 40, -30, 2 + f % 3, 20, -10) for f in range(n)]
    hist_a = torch.zeros(256, dtype=torch.int32, device="cuda")
    want_d = eng.plane_derivative_hist(D, hist_a)
    want_p = eng.plane_classify(want_d, params)
    hist_b = torch.zeros(256, dtype=torch.int32, device="cuda")
    padded = [torch.nn.functional.pad(D[f], (0, 9)) for f in range(n)]   # a wider row step than the image
    got_d, got_p = eng.plane_label_multi([t[:, :w] for t in padded], hist_b, params)
    torch.cuda.synchronize()
    assert torch.equal(hist_a, hist_b) and int(hist_a.sum()) > 0
    for f in range(n):
        assert torch.equal(got_d[f], want_d[f]) and torch.equal(got_p[f], want_p[f]), f
    one = eng.plane_label_multi([D[3]], torch.zeros(256, dtype=torch.int32, device="cuda"), params[3])
    assert torch.equal(one[1][0], want_p[3])


@pytest.mark.gpu
def test_ccl_tile_borders(torch_cuda):
    """The labelling works on 64 x 32 tiles in LDS and unites across tile borders afterwards: sizes around the tile
    multiples, blobs at several scales (components that span many tiles and noise that does not), and U shapes whose two
    arms lie in one tile and meet only in the neighbouring tile (two tile roots of ONE tile end up in one component)."""
    torch = torch_cuda
    rng = np.random.default_rng(4242)
    for w, h in [(64, 32), (65, 33), (63, 31), (128, 64), (129, 65), (200, 100), (16, 8), (320, 37), (70, 160)]:
        eng = make_engine(w, h, 0, 0, inflight=6)
        yy, xx = np.mgrid[0:h, 0:w]
        maps = [rng.integers(0, 3, (h, w)).astype(np.uint8)]
        for scale in (3, 9, 27):
            maps.append(np.kron(rng.integers(0, 3, (h // scale + 1, w // scale + 1)), np.ones((scale, scale), int))[:h, :w].astype(np.uint8))
        u = np.full((h, w), 2, np.uint8)   # U across the first vertical tile border: arms on rows 2 and 6 left of x = 64, joined at x >= 64
        if w > 70 and h > 8:
            u[2, 40:70] = 0; u[6, 40:70] = 0; u[2:7, 69] = 0
            u[10:12, :] = 1
        maps.append(u)
        v = np.full((h, w), 2, np.uint8)   # the same across the first horizontal border (y = 32), plus a frame around the image
        if h > 40 and w > 8:
            v[20:40, 3] = 1; v[20:40, 7] = 1; v[39, 3:8] = 1
        v[0, :] = 0; v[-1, :] = 0; v[:, 0] = 0; v[:, -1] = 0
        maps.append(v)
        ids, n = eng.plane_ccl(dev(torch, np.stack(maps)))
        ids, n = ids.cpu().numpy(), n.cpu().numpy()
        for f, m in enumerate(maps):
            e, en = O.ccl(m)
            assert (ids[f] == e).all() and n[f] == en, f"{w}x{h} map {f}"
        eng.close()


@pytest.mark.gpu
def test_stage_timing_records_every_kth_call(torch_cuda):
    """cart_engine_set_timing(k): HIP events around the stages of every k-th compute call only (bench.py samples its timed steps so that the
    events' cost is not on every step); collect_timing averages what was recorded; the results never depend on it."""
    torch = torch_cuda
    w, h, D, P = 200, 64, 64, 8
    ls, rs = synth.make_batch(2, w, h, D, 4, seed=3)
    eng = make_engine(w, h, D, P, 4, radius=2, iters=1, inflight=2)
    L, R = dev(torch, ls), dev(torch, rs)
    ref = eng.compute_disparity(L, R).cpu().numpy()
    for every, calls, want in ((1, 6, 6), (4, 10, 3), (3, 3, 1)):
        eng.set_timing(True, every=every)
        for _ in range(calls):
            got = eng.compute_disparity(L, R)
        stages, n = eng.collect_timing()
        assert n == want, (every, calls, n)
        assert set(stages) == {"census", "aggregate", "wta", "post"} and all(v > 0 for v in stages.values())   # post carries the fused first interpolation pass
        assert (got.cpu().numpy() == ref).all()
    eng.set_timing(False)
    eng.compute_disparity(L, R)
    assert eng.collect_timing()[1] == 0
    eng.close()


@pytest.mark.gpu
def test_ccl_table_in_one_call(torch_cuda):
    """cart_plane_ccl_table: ids, count and component table from ONE call (the pass that writes the final ids feeds a statistics scratch keyed
    by root pixel; ccl_table_kernel ranks the roots from per-segment counts and empties the scratch again).  Against the oracle on the tile-border
    maps, hard shapes and noise; the two-call path (cart_plane_ccl + cart_plane_ccl_stats) must agree; the scratch must be all zeros after
    every call -- also after a truncated table and after an id map that is not one (foreign ids are ignored, not accumulated for ever)."""
    torch = torch_cuda
    rng = np.random.default_rng(515)
    # (the last three: more than 64 tile columns per row -- the table kernel walks a row's segment counts in chunks of 64 --, a single tile column, a prime width)
    for w, h in [(64, 32), (65, 33), (200, 100), (16, 8), (320, 37), (70, 160), (1242, 375), (16384, 8), (4160, 40), (16, 2000), (4099, 11)]:
        eng = make_engine(w, h, 0, 0, inflight=6)
        yy, xx = np.mgrid[0:h, 0:w]
        maps = [rng.integers(0, 3, (h, w)).astype(np.uint8), ((xx + yy) % 2).astype(np.uint8), np.zeros((h, w), np.uint8)]
        for scale in (3, 9, 27):
            maps.append(np.kron(rng.integers(0, 3, (h // scale + 1, w // scale + 1)), np.ones((scale, scale), int))[:h, :w].astype(np.uint8))
        cap = w * h + 1
        pl = dev(torch, np.stack(maps))
        for rep in range(2):   # the second call runs on the scratch the first one left behind
            ids, table, n = eng.plane_ccl_table(pl, max_components=cap)
            assert eng.debug_ccl_scratch_nonzero() == 0, f"{w}x{h}: scratch not returned to zero"
            for f, m in enumerate(maps):
                eids, en = O.ccl(m)
                et, _ = O.ccl_stats(m, eids)
                assert (ids[f].cpu().numpy() == eids).all() and int(n[f]) == en, f"{w}x{h} map {f} ids / count"
                got = table[f, :en].cpu().numpy()
                assert (got == et).all(), f"{w}x{h} map {f}: {int((got != et).any(axis=1).sum())} table rows differ"
        ids2, n2 = eng.plane_ccl(pl)
        table2, n3 = eng.plane_ccl_stats(pl, ids2, max_components=cap)
        assert torch.equal(ids2, ids) and torch.equal(n2, n) and torch.equal(n3, n)
        for f in range(len(maps)):
            assert torch.equal(table2[f, :int(n[f])], table[f, :int(n[f])])
        # truncation: the first rows, the true count, a clean scratch
        _, ts, ns = eng.plane_ccl_table(pl[:1], max_components=3)
        et, en = O.ccl_stats(maps[0], O.ccl(maps[0])[0], max_components=3)
        assert int(ns[0]) == en and (ts[0].cpu().numpy() == et).all() and eng.debug_ccl_scratch_nonzero() == 0
        # an "id map" whose values are not roots of itself: ignored, nothing left behind, and the next call is right again
        bogus = torch.full_like(ids2[:1], 5)
        bogus[0, 0, :8] = -1
        eng.plane_ccl_stats(pl[:1], bogus, max_components=8)
        assert eng.debug_ccl_scratch_nonzero() == 0
        _, t4, n4 = eng.plane_ccl_table(pl[:1], max_components=cap)
        assert torch.equal(t4[0, :int(n4[0])], table[0, :int(n[0])])
        eng.close()


@pytest.mark.gpu
def test_full_size_oracle_1080p_d256(torch_cuda):
    """BASELINE configs[3] against the oracle on the full image: 1920x1080, D=256, 8 paths, a batch of 4 (fused WTA, the
    default there) and a single pair (two-kernel WTA), bit for bit; ~2 s of oracle time per pair on the GPU box's cores."""
    torch = torch_cuda
    w, h, D, P = 1920, 1080, 256, 8
    ls, rs = synth.make_batch(2, w, h, D, 4)
    want = [O.disparity_module(ls[k], rs[k], D, P, 4, radius=2, iterations=1) for k in range(2)]
    eng = make_engine(w, h, D, P, radius=2, iters=1, inflight=4)
    L, R = dev(torch, np.concatenate([ls, ls])), dev(torch, np.concatenate([rs, rs]))
    got = eng.compute_disparity(L, R).cpu().numpy()
    for k in range(4):
        assert (got[k] == want[k % 2]).all(), f"batch frame {k}: {int((got[k] != want[k % 2]).sum())} pixels differ"
    assert (eng.compute_disparity(L[1], R[1]).cpu().numpy() == want[1]).all(), "single pair"
    eng.close()


@pytest.mark.gpu
def test_integer_uniqueness_threshold_on_device(torch_cuda):
    """The device function behind the WTA kernels' uniqueness test, enumerated over its whole domain (best cost 0..2047 x
    ratio 0..100) against the float compare it replaces."""
    from cartslam.engine import uniq_table
    from test_cabi import _check_uniq_table
    eng = make_engine(64, 32, 64, 4)
    for ratio in range(101):
        got = uniq_table(ratio, eng)
        _check_uniq_table(got, ratio)
        assert (got == uniq_table(ratio)).all(), f"device and host copies differ at ratio {ratio}"
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sw,sh,dw,dh,ch", [(300, 100, 256, 96, 3), (1242, 375, 621, 188, 1), (97, 61, 200, 130, 3), (64, 32, 64, 32, 1), (333, 77, 100, 231, 1)])
def test_resize_linear(torch_cuda, sw, sh, dw, dh, ch):
    """cart_resize_linear (KITTI source, kitti.cpp:169-172) against oracle S16: down- and up-scaling, both axes ragged, a
    pitched source, identity size."""
    torch = torch_cuda
    from cartslam.engine import resize_linear
    rng = np.random.default_rng(sw * 7 + dh)
    img = rng.integers(0, 256, (sh, sw, 3) if ch == 3 else (sh, sw)).astype(np.uint8)
    buf = torch.zeros((sh + 2, sw + 13, 3) if ch == 3 else (sh + 2, sw + 13), dtype=torch.uint8, device="cuda")
    view = buf[:sh, :sw]
    view.copy_(dev(torch, img))
    got = resize_linear(view, dw, dh).cpu().numpy()
    exp = O.resize_linear(img, dw, dh)
    assert got.shape == exp.shape and (got == exp).all(), f"{int((got != exp).sum())} samples differ, max |diff| {int(np.abs(got.astype(int) - exp.astype(int)).max())}"
    if (sw, sh) == (dw, dh):
        assert (got == img).all()


@pytest.mark.gpu
@pytest.mark.parametrize("P", [4, 8])
def test_side_stream_overlap_gives_the_same_outputs(torch_cuda, P):
    """StereoPipeline(overlap=True) -- the plane stages of batch i on a side stream beside the disparity kernels of batch
    i+1, with and without the caller's inputs-ready event -- returns exactly what the one-stream pipeline returns: five
    batches whose scenes differ, a parameter refresh in between (ids 1 and 31)."""
    torch = torch_cuda
    from cartslam.pipeline import StereoPipeline
    w, h, D, B = 330, 120, 64, 8
    batches = []
    for k in range(5):
        ls, rs = synth.make_batch(B, w, h, D if k % 2 == 0 else 40, 4, first_frame=B * k)
        batches.append((dev(torch, ls), dev(torch, rs)))
    results = {}
    torch.cuda.synchronize()
    ready = torch.cuda.current_stream().record_event()   # the inputs are complete from here on
    for mode in ("one_stream", "side", "side_ready"):
        eng = make_engine(w, h, D, P, 4, radius=2, iters=1, inflight=2 * B)
        pipe = StereoPipeline(eng, provider="histogram_peak", with_ccl=True, overlap=mode != "one_stream")
        assert (pipe.side is not None) == (mode != "one_stream")
        outs = [pipe.process_batch(l, r, inputs_ready=ready if mode == "side_ready" else None) for l, r in batches]
        torch.cuda.synchronize()
        results[mode] = [{k: o[k].cpu().numpy() for k in ("disparity", "planes", "ids", "n_components", "components", "params")} for o in outs]
        eng.close()
    for mode in ("side", "side_ready"):
        for a, b in zip(results["one_stream"], results[mode]):
            for k in a:
                if k == "components":   # rows past a frame's component count are not written
                    for f in range(B):
                        nc = min(int(a["n_components"][f]), a[k].shape[1])
                        assert np.array_equal(a[k][f, :nc], b[k][f, :nc]), (mode, k, f)
                else:
                    assert np.array_equal(a[k], b[k]), (mode, k)
    # the frames really differ from batch to batch (otherwise a one-batch shift would go unnoticed)
    assert not np.array_equal(results["one_stream"][0]["disparity"], results["one_stream"][1]["disparity"])


@pytest.mark.gpu
@pytest.mark.parametrize("variants", [1, 2, 3, 4, 7])
def test_spec_variants(torch_cuda, variants):
    """The three choices that are open upstream -- S8: the LR check also invalidates integer disparity 0; S7: medians over
    a replicated border; S5: uniqueness from the second-best cost only (the top-2 wording of SURVEY 8a-4(4)) -- as engine
    options (CART_OPT_SPEC_*), singly and together against the oracle's variants, on images that hit
    them (ragged sizes, a border in every tile shape, min_disparity 0 so that disparity 0 wins often), with interpolation on top;
    and back to the default spec afterwards."""
    torch = torch_cuda
    for (w, h, D, P, md, scene) in [(173, 67, 64, 8, 0, "road"), (330, 50, 128, 4, 4, "saturated"), (340, 60, 256, 8, 1, "wall"), (1242, 375, 128, 8, 4, "pole")]:
        l, r, _ = synth.make_pair(w, h, D, md, seed=77, scene=scene)
        base = O.disparity_module(l, r, D, P, md, radius=2, iterations=1)
        exp = O.disparity_module(l, r, D, P, md, radius=2, iterations=1, variants=variants)
        assert (exp != base).any(), "the variant changes nothing on this image: the test would not see a missing switch"
        eng = make_engine(w, h, D, P, md, radius=2, iters=1, inflight=2)
        eng.set_spec_variants(s8_zero_invalid=bool(variants & 1), s7_replicate_border=bool(variants & 2), s5_top2=bool(variants & 4))
        if variants & 4:   # the S5 variant lives in the two-kernel WTA: such an engine takes plan SLABS whatever was asked for
            eng.set_plan("fused_up")
            assert eng.describe_plan(2)["plan"] == "slabs"
        got = eng.compute_disparity(dev(torch, np.stack([l, l])), dev(torch, np.stack([r, r]))).cpu().numpy()
        assert (got[0] == exp).all() and (got[1] == exp).all(), f"{(w, h, D, P, md, scene)}: {int((got[0] != exp).sum())} pixels differ"
        eng.set_spec_variants()
        assert (eng.compute_disparity(dev(torch, l), dev(torch, r)).cpu().numpy() == base).all()
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,D,P,n", [(173, 67, 64, 4, 16), (201, 45, 128, 8, 8), (330, 50, 256, 8, 16), (131, 90, 256, 4, 24)])
def test_xcd_placed_launches_cover_every_frame(torch_cuda, w, h, D, P, n):
    """Launches whose frame count is a multiple of 8 decode their grid per XCD (frames x, x + 8, ... on XCD x; aggregation launch
    and fused sweep, sgm_kernels.hip xcd_placement): n DISTINCT frames, every launch plan, EVERY frame against the oracle -- a
    decode that skipped or doubled a (direction, frame, line group) could not hide behind repeated frames."""
    torch = torch_cuda
    ls, rs = synth.make_batch(n, w, h, D, 4, seed=5150 + D)
    exp = [O.disparity_module(ls[k], rs[k], D, P, 4, radius=2, iterations=1) for k in range(n)]
    assert any((exp[0] != exp[k]).any() for k in range(1, n))
    eng = make_engine(w, h, D, P, 4, radius=2, iters=1, inflight=n)
    eng.set_chunk_frames(n)   # one launch sequence of n frames (24 > the default 16)
    for plan in ("slabs", "fused_up"):
        eng.set_plan(plan)
        assert eng.describe_plan(n)["frames_per_launch"] == n
        got = eng.compute_disparity(dev(torch, ls), dev(torch, rs)).cpu().numpy()
        for k in range(n):
            assert (got[k] == exp[k]).all(), f"{plan}, frame {k}: {int((got[k] != exp[k]).sum())} pixels differ"
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("D,P,inflight,cap", [(128, 8, 40, None), (256, 4, 6, 0), (64, 4, 8, 1 << 20)])
def test_placement_tuning_keeps_the_bits(torch_cuda, D, P, inflight, cap):
    """cart_engine_tune_placement (slot groups of the slab workspace re-allocated a few times, the fastest set kept; the workspace is one
    plain hipMalloc per group of at most 8 GiB) changes no result: a call over EVERY slot -- launch sequences in every group, some of
    them across a group boundary -- gives the same disparities before and after, and the oracle's; the reported times are positive
    and the kept one is not slower than the first.  cap = max_extra_bytes: None no cap, 0 the default (two units), 1 MiB = too small
    for any candidate (the call then only measures)."""
    torch = torch_cuda
    w, h = 1242, 375
    eng = make_engine(w, h, D, P, 4, radius=2, iters=1, inflight=inflight)   # D=128 P=8 x 40 slots = 19 GB: groups of 16 + 16 + 8 slots
    lsb, rsb = synth.make_batch(inflight, w, h, D, 4, scene="stripes")
    l, r = dev(torch, lsb), dev(torch, rsb)
    def every_slot():
        # a lease takes the lowest free range last used on its stream (engine acquire()), so calls of a few frames on one stream
        # never leave the first slots: one call of max_inflight frames is what reaches every slot group
        outs = []
        for plan in ("slabs", "fused_up", "auto"):
            eng.set_plan(plan)
            outs.append(eng.compute_disparity(l, r).cpu().numpy())
        return outs
    before = every_slot()
    first, kept = eng.tune_placement(min(16, inflight), 3, max_extra_bytes=cap)
    assert first > 0 and 0 < kept <= first
    if cap == 1 << 20:
        assert kept == first   # nothing could be tried under a 1 MiB cap
    after = every_slot()
    for a, b in zip(before, after):
        assert np.array_equal(a, b)
    for f in sorted({0, inflight // 2, 17 % inflight, inflight - 1}):
        assert np.array_equal(after[2][f], O.disparity_module(lsb[f], rsb[f], D, P, 4, radius=2, iterations=1)), f
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,D,P,inflight,want_slots,want_groups", [
    (1242, 375, 128, 8, 40, 16, 3),     # 477 MB per slot: 16 slots = 7.6 GB per group, 16 + 16 + 8
    (1242, 375, 64, 4, 32, 32, 1),      # 119 MB per slot: everything in one allocation of 3.8 GB
    (1920, 1080, 256, 8, 3, 2, 2),      # 4.25 GB per slot: two fit (7.91 GiB), three would not
    (3840, 1200, 256, 8, 2, 1, 2),      # 9.4 GB per slot: larger than the limit by itself, one allocation per slot
    (1242, 375, 256, 4, 12, 12, 1),     # the reference's defaults with its 12 frames in flight: one allocation of 5.7 GB
])
def test_slab_workspace_is_cut_into_groups_of_at_most_8_gib(torch_cuda, w, h, D, P, inflight, want_slots, want_groups):
    """no device allocation behind the cost slabs exceeds 8 GiB - 64 MiB (the aggregation launch is always in its slow
    mode above 8 GiB, profiles/r03_alloc.txt), groups hold a multiple of 16 slots from 16 up, and a slot that is larger than the limit by
    itself still gets one allocation."""
    eng = make_engine(w, h, D, P, 4, inflight=inflight)
    lay = eng.slab_layout()
    assert lay["slot_bytes"] == w * h * D * P
    assert (lay["group_slots"], lay["groups"]) == (want_slots, want_groups)
    assert lay["group_bytes"] == lay["group_slots"] * lay["slot_bytes"]
    assert lay["group_bytes"] <= (8 << 30) - (64 << 20) or lay["group_slots"] == 1
    eng.close()


@pytest.mark.gpu
def test_two_large_engines_tuned_side_by_side(torch_cuda):
    """Round-3 hazard, made unreachable: the slab workspace used to sit behind HIP virtual-memory-management calls, and ranges that were
    freed and re-reserved while others were live ended in GPU memory access faults (profiles/r04_vmm_faults.txt).  The workspace is plain
    hipMalloc groups now.  The sequence that was never covered: two engines above 8 GiB alive at once, both tuned, the first tuned a
    second time, one destroyed and a third created and tuned in its place -- every engine's call over all of its slots must give the
    oracle's disparities at every step."""
    torch = torch_cuda
    w, h = 1242, 375
    cfgs = {"a": (128, 8, 20), "b": (256, 4, 18)}   # 9.5 GB (groups of 16 + 4 slots) and 8.6 GB (16 + 2)
    engs, data = {}, {}
    def check(k):
        D, P, n = cfgs[k]
        ls, rs = data[k]
        got = engs[k].compute_disparity(dev(torch, ls), dev(torch, rs)).cpu().numpy()
        for f in (0, 15, 16, n - 1):   # both groups, both sides of the boundary
            assert np.array_equal(got[f], O.disparity_module(ls[f], rs[f], D, P, 4, radius=2, iterations=1)), (k, f)
    for k, (D, P, n) in cfgs.items():
        engs[k] = make_engine(w, h, D, P, 4, radius=2, iters=1, inflight=n)
        data[k] = synth.make_batch(n, w, h, D, 4, first_frame=3)
    check("a"); check("b")
    for k in ("a", "b"):
        first, kept = engs[k].tune_placement(16, 3)
        assert first > 0 and 0 < kept <= first
    check("a"); check("b")
    engs["a"].tune_placement(16, 4, max_extra_bytes=None)   # second search on an engine whose first search freed its losers
    check("a"); check("b")
    engs["b"].close()                                       # its groups go back to the allocator while "a" is live ...
    engs["b"] = make_engine(w, h, 256, 4, 4, radius=2, iters=1, inflight=18)   # ... and come back for a new engine
    engs["b"].tune_placement(16, 3)
    check("b"); check("a")
    for e in engs.values():
        e.close()
