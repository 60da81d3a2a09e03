"""GPU parity tests (-m gpu) of the superpixel stage and the superpixel plane labelling, through the
C ABI (cart_superpixels_*, cart_superpixel_plane_classify) against the CPU oracle (spec S13/S14).  Labels are
indices: the bar is bit-exact, including the double-precision cost comparisons that pick them."""
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


def geometry_engine(w, h, inflight=4):
    from cartslam import Engine
    return Engine(w, h, num_disparities=0, paths=0, max_inflight=inflight)


def labels_np(t):
    return t.cpu().numpy().view(np.uint16)


def random_scene(rng, w, h, coarse=5):
    """blocky colour image + noise, and a 2-channel derivative image with INVALID holes"""
    base = rng.integers(0, 256, (h // coarse + 2, w // coarse + 2, 3)).astype(np.uint8)
    bgr = np.kron(base, np.ones((coarse, coarse, 1), np.uint8))[:h, :w]
    bgr = np.clip(bgr.astype(int) + rng.integers(-9, 10, bgr.shape), 0, 255).astype(np.uint8)
    d2 = (np.kron(rng.integers(-30, 30, (h // 7 + 2, w // 7 + 2, 2)), np.ones((7, 7, 1), int))[:h, :w] + rng.integers(-3, 4, (h, w, 2))).astype(np.int16)
    d2[rng.random((h, w, 2)) < 0.08] = -32768
    return bgr, d2


PARAM_SETS = [
    dict(),                                                      # JSON factory defaults (cartconfig.cpp:121-133)
    dict(compactness=0.03, progressive=1.0, image=1.5, disparity=1.0),   # config/modules/kitti-superpixels.json
    dict(compactness=0.05, image=1.0, disparity=1.25),           # constructor defaults (superpixels.hpp:18-27)
    dict(disparity=0.0),                                         # no disparity feature: deriv2 may be NULL
    dict(image=0.0, compactness=0.0, disparity=2.0),             # disparity feature + cliques only
    dict(image=0.0, compactness=0.0, disparity=0.0),             # cliques only
]


def make_sp(eng, bs, kw, block_h=None):
    from cartslam import Superpixels
    return Superpixels(eng, block_size=bs, block_h=block_h, direct_clique_cost=kw.get("direct", 0.5),
                       diagonal_clique_cost=kw.get("diagonal"), compactness_weight=kw.get("compactness", 0.1),
                       progressive_compactness_cost=kw.get("progressive", 0.0), image_weight=kw.get("image", 1.5),
                       disparity_weight=kw.get("disparity", 1.0))


@pytest.mark.parametrize("w,h,bs,case", [(96, 64, 8, 0), (131, 77, 12, 1), (64, 16, 5, 2), (200, 45, 10, 3), (77, 130, 7, 4), (90, 70, 9, 5),
                                          (70, 41, 3, 1), (66, 50, 2, 0), (100, 37, 4, 2)])   # blocks of 2-4 pixels: more labels in a tile than its LDS table holds
def test_relax_matches_oracle(torch_cuda, w, h, bs, case):
    torch = torch_cuda
    kw = PARAM_SETS[case]
    rng = np.random.default_rng(100 + case)
    eng = geometry_engine(w, h)
    sp = make_sp(eng, bs, kw)
    want, mx = O.sp_block_init(w, h, bs, bs)
    assert sp.max_label == mx
    p = O.sp_params(**kw)
    # a short frame sequence on the persistent state: 3 sweeps, 2 sweeps on a new frame, 0 sweeps, reset, 4 sweeps
    for step, iters in enumerate((3, 2, 0, -1, 4)):
        if iters < 0:
            sp.reset()
            want, _ = O.sp_block_init(w, h, bs, bs)
            continue
        bgr, d2 = random_scene(rng, w, h)
        use_d2 = kw.get("disparity", 1.0) > 0
        got = labels_np(sp.relax(dev(torch, bgr), dev(torch, d2) if use_d2 else None, iters))
        want, changes = O.sp_relax(p, want, mx, O.bgr2ycrcb(bgr), d2 if use_d2 else None, iters)
        assert (got == want).all(), (step, int((got != want).sum()))
        if iters and kw != PARAM_SETS[5]:
            assert changes > 0  # the case does exercise label moves
    sp.close(); eng.close()


def test_gray_input_and_rectangular_blocks(torch_cuda):
    torch = torch_cuda
    w, h = 150, 60
    rng = np.random.default_rng(7)
    eng = geometry_engine(w, h)
    sp = make_sp(eng, 16, {}, block_h=6)
    want, mx = O.sp_block_init(w, h, 16, 6)
    gray = np.kron(rng.integers(0, 256, (h // 6 + 1, w // 6 + 1)).astype(np.uint8), np.ones((6, 6), np.uint8))[:h, :w]
    _, d2 = random_scene(rng, w, h)
    got = labels_np(sp.relax(dev(torch, gray), dev(torch, d2), 3))
    yc = O.bgr2ycrcb(np.repeat(gray[..., None], 3, axis=2))  # processImage replicates gray into BGR (datasource.cpp:11-13)
    assert (yc[..., 0] == gray).all() and (yc[..., 1:] == 128).all()
    want, _ = O.sp_relax(O.sp_params(), want, mx, yc, d2, 3)
    assert (got == want).all()
    sp.close(); eng.close()


def test_set_labels_and_pitched_images(torch_cuda):
    """setLabelImage with arbitrary (non-block, non-contiguous-id) labels, and pitched BGR / derivative / output rows."""
    torch = torch_cuda
    w, h = 101, 58
    rng = np.random.default_rng(8)
    eng = geometry_engine(w, h)
    sp = make_sp(eng, 8, {})
    # vertical stripes with wobbling borders, labels 3, 40, 41, 900 (most ids unused: vanished labels have count 0)
    ids = np.array([3, 40, 41, 900], np.uint16)
    edges = (np.array([25, 50, 75])[None, :] + rng.integers(-3, 4, (h, 3))).astype(int)
    lab = np.zeros((h, w), np.uint16)
    xs = np.arange(w)[None, :]
    lab[:] = ids[(xs >= edges[:, :1]).astype(int) + (xs >= edges[:, 1:2]) + (xs >= edges[:, 2:3])]
    with pytest.raises(Exception):
        sp.set_labels(dev(torch, lab.view(np.int16)), 900)  # label 900 is not < 900
    sp.set_labels(dev(torch, lab.view(np.int16)), 1000)
    assert sp.max_label == 1000
    bgr, d2 = random_scene(rng, w, h)
    big_img = torch.zeros((h, w + 13, 3), dtype=torch.uint8, device="cuda"); big_img[:, :w] = dev(torch, bgr)
    big_d2 = torch.zeros((h, w + 5, 2), dtype=torch.int16, device="cuda"); big_d2[:, :w] = dev(torch, d2)
    got = labels_np(sp.relax(big_img[:, :w], big_d2[:, :w], 5))
    want, changes = O.sp_relax(O.sp_params(), lab, 1000, O.bgr2ycrcb(bgr), d2, 5)
    assert changes > 0 and (got == want).all()
    sp.close(); eng.close()


def test_bad_arguments_fail_loudly(torch_cuda):
    torch = torch_cuda
    from cartslam import EngineError, Superpixels
    eng = geometry_engine(64, 32)
    with pytest.raises(EngineError):
        Superpixels(eng, block_size=0)
    with pytest.raises(EngineError):
        Superpixels(eng, block_size=8, compactness_weight=-1.0)
    with pytest.raises(EngineError):
        Superpixels(eng, block_size=100)  # image smaller than a block (initialization.cu:42)
    big = geometry_engine(1242, 375)
    with pytest.raises(EngineError):
        Superpixels(big, block_size=4)    # 311*94 blocks >= 16384
    big.close()
    sp = Superpixels(eng, block_size=8)
    img = torch.zeros((32, 64, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(EngineError):
        sp.relax(img, None, 1)            # disparity feature on, no derivative image
    with pytest.raises(EngineError):
        sp.relax(img[:16], None, 1)
    sp.close(); eng.close()


def test_superpixel_plane_classify(torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(21)
    w, h = 139, 53
    eng = geometry_engine(w, h)
    for n_prev in (0, 1, 3):
        labels, mx = O.sp_block_init(w, h, 9, 9)
        labels = np.roll(labels, rng.integers(0, 4), axis=1)  # not aligned with the strip walk
        d2 = rng.integers(-12, 30, (h, w, 2)).astype(np.int16)
        d2[rng.random((h, w, 2)) < 0.15] = -32768
        params = (6, 22, -4, 6, 14, 1)
        prev = [rng.integers(0, 3, (h, w)).astype(np.uint8) for _ in range(n_prev)]
        flows = [(rng.integers(-40, 40, (h, w, 2)) * rng.integers(1, 64, (h, w, 2))).astype(np.int16) for _ in range(n_prev)]
        uns, pl = eng.superpixel_plane_classify(dev(torch, d2), dev(torch, labels.view(np.int16)), mx, params,
                                                [dev(torch, p) for p in prev], [dev(torch, f) for f in flows])
        wu, wp = O.sp_classify(d2, labels, mx, params, prev, flows)
        assert (uns.cpu().numpy() == wu).all() and (pl.cpu().numpy() == wp).all(), n_prev
        assert len(np.unique(wp)) >= 2
    # one superpixel covering > 65535 pixels: the reference's u16 vote counters wrap (sp_planeseg.cu:37,122-127)
    eng.close()
    w, h = 400, 200
    eng = geometry_engine(w, h)
    labels = np.zeros((h, w), np.uint16); labels[:, 390:] = 1
    d2 = np.zeros((h, w, 2), np.int16); d2[..., 0] = 10            # horizontal everywhere ...
    d2[:20, :, 0] = 0                                               # ... but 20 rows vertical
    params = (6, 22, -4, 6, 14, 1)
    uns, pl = eng.superpixel_plane_classify(dev(torch, d2), dev(torch, labels.view(np.int16)), 2, params)
    wu, wp = O.sp_classify(d2, labels, 2, params)
    assert (uns.cpu().numpy() == wu).all() and (pl.cpu().numpy() == wp).all()
    assert wp[100, 0] == 1  # label 0: 70200 horizontal votes wrap to 4664 < 7800 vertical votes -> VERTICAL
    eng.close()


def test_full_size_kitti_geometry(torch_cuda):
    """1242x375, block 12 (config/modules/kitti-planeseg.json): 8 sweeps against the oracle, then properties."""
    torch = torch_cuda
    w, h, bs = 1242, 375, 12
    l, r, _ = synth.make_pair(w, h, 128, 4, seed=77, channels=3)
    d = O.disparity_module(l, r, 64, 4, 4, radius=2, iterations=1)
    d2, hist = O.directional_derivative(d)
    eng = geometry_engine(w, h)
    sp = make_sp(eng, bs, {})
    want, mx = O.sp_block_init(w, h, bs, bs)
    assert mx == 104 * 32
    got = labels_np(sp.relax(dev(torch, l), dev(torch, d2), 8))
    want, changes = O.sp_relax(O.sp_params(), want, mx, O.bgr2ycrcb(l), d2, 8)
    assert changes > 10000 and (got == want).all(), int((got != want).sum())
    assert got.max() < mx
    # zero sweeps return the state unchanged; the state persists across calls
    again = labels_np(sp.relax(dev(torch, l), dev(torch, d2), 0))
    assert (again == got).all()
    ok, pp = O.histogram_peak_params(hist[:, 0].copy())
    params = pp if ok else (6, 18, -5, 6, 11, 0)
    uns, pl = eng.superpixel_plane_classify(dev(torch, d2), dev(torch, got.view(np.int16)), mx, params)
    wu, wp = O.sp_classify(d2, got, mx, params)
    assert (uns.cpu().numpy() == wu).all() and (pl.cpu().numpy() == wp).all()
    # every superpixel carries exactly one plane label
    pln = pl.cpu().numpy()
    per_label = np.zeros((mx, 3), np.int64)
    np.add.at(per_label, (got.astype(np.int64).ravel(), pln.astype(np.int64).ravel()), 1)
    assert ((per_label > 0).sum(axis=1) <= 1).all()
    sp.close(); eng.close()


def test_golden_superpixel_fixtures(torch_cuda):
    torch = torch_cuda
    files = sorted(glob.glob(os.path.join(HERE, "golden", "sp_*.npz")))
    assert files, "no superpixel golden fixtures committed"
    for f in files:
        z = np.load(f)
        h, w = z["labels0"].shape
        kw = {k: float(z["p_" + k]) for k in ("direct", "diagonal", "compactness", "progressive", "image", "disparity")}
        eng = geometry_engine(w, h)
        sp = make_sp(eng, int(z["block"]), kw)
        for k in range(int(z["frames"])):
            got = labels_np(sp.relax(dev(torch, z[f"image{k}"]), dev(torch, z[f"deriv{k}"]), int(z[f"iters{k}"])))
            assert (got == z[f"labels{k}"]).all(), (f, k)
            uns, pl = eng.superpixel_plane_classify(dev(torch, z[f"deriv{k}"]), dev(torch, got.view(np.int16)), sp.max_label,
                                                    tuple(int(v) for v in z["plane_params"]))
            assert (uns.cpu().numpy() == z[f"unsmoothed{k}"]).all() and (pl.cpu().numpy() == z[f"planes{k}"]).all(), (f, k)
        sp.close(); eng.close()


def test_streams_threads_and_lifecycle(torch_cuda):
    """One object driven alternately from two streams without host synchronisation (calls are ordered by the object's
    event), two objects driven concurrently from two host threads, and create/destroy cycles that must not leak."""
    import threading
    torch = torch_cuda
    w, h, bs = 160, 90, 9
    rng = np.random.default_rng(44)
    eng = geometry_engine(w, h)
    scenes = [random_scene(rng, w, h) for _ in range(4)]
    dscenes = [(dev(torch, b), dev(torch, d)) for b, d in scenes]
    want, mx = O.sp_block_init(w, h, bs, bs)
    exp = []
    for b, d in scenes:
        want, _ = O.sp_relax(O.sp_params(), want, mx, O.bgr2ycrcb(b), d, 2)
        exp.append(want)
    # (1) alternate streams, no sync in between
    sp = make_sp(eng, bs, {})
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    outs = []
    for k, (b, d) in enumerate(dscenes):
        with torch.cuda.stream(streams[k % 2]):
            outs.append(sp.relax(b, d, 2))
    torch.cuda.synchronize()
    for k in range(4):
        assert (labels_np(outs[k]) == exp[k]).all(), k
    sp.close()
    # (2) two objects, two host threads
    results, errors = {}, []

    def work(i):
        try:
            s = torch.cuda.Stream()
            obj = make_sp(eng, bs, {})
            with torch.cuda.stream(s):
                got = [obj.relax(b, d, 2) for b, d in dscenes]
            s.synchronize()
            results[i] = [labels_np(g) for g in got]
            obj.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for i in range(2):
        for k in range(4):
            assert (results[i][k] == exp[k]).all(), (i, k)
    # (3) lifecycle
    def cycle(n):
        for _ in range(n):
            o = make_sp(eng, bs, {})
            o.relax(dscenes[0][0], dscenes[0][1], 1)
            o.close()
        torch.cuda.synchronize()
    cycle(3)
    free0 = torch.cuda.mem_get_info()[0]
    cycle(20)
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, f"leak: {(free0 - free1) >> 20} MiB over 20 create/destroy cycles"
    eng.close()
