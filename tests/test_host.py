"""Tests of the C++ host mirror (cart-slam_amd/host): the reference's plugin API shape driven by the reference's own
JSON config format.  CPU part: config errors read like the reference's and there is no CPU fallback.  GPU part: the
frame loop's blackboard outputs equal the oracle's, frame by frame."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from cartslam import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "cart-slam_amd", "build", "cart_slam_amd")

# Same content as the reference's config/modules/kitti-naive-segmentation.json:1-17 and kitti-disparity.json:1-10
# (re-typed here as data), plus the derivative module type (cartconfig.cpp:161-163).
MODULES_NAIVE_SEG = [
    {"type": "disparity", "smoothing_radius": 2, "smoothing_iterations": 1},
    {"type": "disparity_planeseg", "parameter_provider": {"type": "histogram_peak"}},
    {"type": "disparity_planeseg_visualization", "show_histogram": True},
]


def write_pnm(path, img):
    with open(path, "wb") as f:
        if img.ndim == 2:
            f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0])); f.write(img.tobytes())
        else:
            f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0])); f.write(img[..., ::-1].tobytes())  # BGR -> RGB on disk


def make_dataset(tmp, n, w, h, channels=1, seq=0):
    d = os.path.join(tmp, "dataset", "sequences", "%02d" % seq)
    os.makedirs(os.path.join(d, "image_2")); os.makedirs(os.path.join(d, "image_3"))
    frames = []
    for f in range(n):
        l, r, _ = synth.make_pair(w, h, 128, 4, seed=4242, frame=f, channels=channels)
        ext = "pgm" if channels == 1 else "ppm"
        write_pnm(os.path.join(d, "image_2", "%06d.%s" % (f, ext)), l)
        write_pnm(os.path.join(d, "image_3", "%06d.%s" % (f, ext)), r)
        frames.append((l, r))
    src = os.path.join(tmp, "source.json")
    json.dump({"type": "kitti", "path": os.path.join(tmp, "dataset"), "sequence": seq}, open(src, "w"))
    return src, frames


def run_exe(src, modules, tmp, extra=(), env=None):
    mod = os.path.join(tmp, "modules.json")
    json.dump(modules, open(mod, "w"))
    return subprocess.run([EXE, src, mod, *extra], capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))


def test_exe_exists():
    assert os.path.exists(EXE), "host executable not built (make -C cart-slam_amd)"


def test_runtime_without_gpu():
    """System / worker pool / blackboard / cross-frame dependencies / retention ring with dummy modules (host/tests/runtime_test.cpp):
    400 frames, 12 in flight, consumers listed before their providers, one module that throws on one frame."""
    exe = os.path.join(os.path.dirname(EXE), "runtime_test")
    assert os.path.exists(exe), "runtime_test not built (make -C cart-slam_amd)"
    for _ in range(5):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def _write_png(path, img, color_type, filters):
    """Own PNG writer (zlib only): img uint8 [h, w, c]; one filter type per row from `filters`, cycled (0 None, 1 Sub, 2 Up,
    3 Average, 4 Paeth), so that every branch of the reader's un-filtering is exercised -- PIL picks filters adaptively."""
    import struct
    import zlib
    h, w, c = img.shape
    raw, prev = bytearray(), np.zeros(w * c, np.int32)
    for y in range(h):
        cur = img[y].reshape(-1).astype(np.int32)
        ft = filters[y % len(filters)]
        left = np.concatenate([np.zeros(c, np.int32), cur[:-c]])
        upleft = np.concatenate([np.zeros(c, np.int32), prev[:-c]])
        if ft == 0: f = cur
        elif ft == 1: f = cur - left
        elif ft == 2: f = cur - prev
        elif ft == 3: f = cur - ((left + prev) >> 1)
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            f = cur - pred
        raw.append(ft); raw += (f & 255).astype(np.uint8).tobytes()
        prev = cur
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    comp = zlib.compress(bytes(raw), 6)
    idat = b"".join(chunk(b"IDAT", comp[i:i + 1000]) for i in range(0, len(comp), 1000))   # several IDAT chunks
    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0)) +
                 chunk(b"tEXt", b"Comment\0ancillary chunks are skipped") + idat + chunk(b"IEND", b""))


@pytest.mark.parametrize("color_type,channels", [(0, 1), (2, 3), (4, 2), (6, 4)])
def test_png_reader(tmp_path, color_type, channels):
    """The host's PNG reader (host/src/png.cpp; cv::imread of src/sources/kitti.cpp:131,152-153): gray / RGB / gray+alpha /
    RGBA, every row filter, several IDAT chunks, odd sizes -> BGR bytes like cv::imread's default; malformed files throw."""
    exe = os.path.join(os.path.dirname(EXE), "runtime_test")
    rng = np.random.default_rng(color_type)
    for (w, h), filters in (((37, 11), [0, 1, 2, 3, 4]), ((1, 1), [4]), ((64, 9), [3, 4, 1]), ((5, 40), [2, 4])):
        img = rng.integers(0, 256, (h, w, channels)).astype(np.uint8)
        img[:, : w // 2] //= 16   # smooth regions as well: small differences exercise the Paeth tie rules
        path = str(tmp_path / f"t_{color_type}_{w}x{h}.png")
        _write_png(path, img, color_type, filters)
        r = subprocess.run([exe, "--png", path], capture_output=True, timeout=60)
        assert r.returncode == 0, r.stdout[:200]
        head, _, body = r.stdout.partition(b"\n")
        gw, gh, gc = (int(v) for v in head.split())
        # cv::imread(path) = IMREAD_COLOR: always 3-channel BGR, gray replicated, alpha dropped
        exp = np.repeat(img[..., :1], 3, axis=2) if channels <= 2 else img[..., 2::-1]
        assert (gw, gh, gc) == (w, h, 3)
        got = np.frombuffer(body, np.uint8).reshape(exp.shape)
        assert (got == exp).all(), f"{w}x{h} filters {filters}"
    # truncated data and a wrong signature fail loudly; a missing file is reported as such
    data = open(path, "rb").read()
    bad = str(tmp_path / "bad.png"); open(bad, "wb").write(data[: len(data) // 2])
    assert subprocess.run([exe, "--png", bad], capture_output=True, timeout=60).returncode == 3
    open(bad, "wb").write(b"JFIF" + data[4:])
    assert subprocess.run([exe, "--png", bad], capture_output=True, timeout=60).returncode == 3
    assert subprocess.run([exe, "--png", str(tmp_path / "none.png")], capture_output=True, timeout=60).returncode == 2


def test_config_errors_read_like_the_reference(tmp_path):
    tmp = str(tmp_path)
    src, _ = make_dataset(tmp, 1, 64, 32)
    r = run_exe(src, [{"type": "features"}], tmp)  # a module type outside the hot path
    assert r.returncode != 0 and "Unknown module type features." in r.stderr  # cartconfig.cpp:226
    r = run_exe(src, [{"type": "superpixels", "block_size": 0}], tmp)
    assert r.returncode != 0 and "blockSize must be more than 1" in r.stderr  # superpixels.cu:37-39
    r = run_exe(src, [{"type": "superpixels", "image_weight": -1.0}], tmp)
    assert r.returncode != 0 and "weight must be non-negative" in r.stderr  # superpixels.cu:45-47
    r = run_exe(src, {"type": "disparity"}, tmp)
    assert r.returncode != 0 and "Modules configuration is not an array." in r.stderr  # cartconfig.cpp:107-109
    r = run_exe(src, [{"type": "disparity_planeseg", "parameter_provider": {"type": "static", "horizontal_range_min": 1}}], tmp)
    assert r.returncode != 0 and "Key horizontal_range_max not found." in r.stderr  # cartconfig.cpp:48-53
    r = run_exe(src, [{"type": "disparity_planeseg", "parameter_provider": {"type": "magic"}}], tmp)
    assert r.returncode != 0 and "Unknown parameter provider type." in r.stderr  # cartconfig.cpp:77
    bad = os.path.join(tmp, "bad_source.json")
    json.dump({"type": "lidar", "path": "/x"}, open(bad, "w"))
    r = run_exe(bad, [], tmp)
    assert r.returncode != 0 and "Unknown data source type." in r.stderr  # cartconfig.cpp:100


def test_no_cpu_fallback(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("only meaningful on a box without a GPU")
    tmp = str(tmp_path)
    src, _ = make_dataset(tmp, 1, 64, 32)
    r = run_exe(src, MODULES_NAIVE_SEG, tmp)
    assert r.returncode != 0 and "cart_engine_create" in r.stderr  # constructing the module must fail loudly


def load(tmp, fid, key, dtype, shape):
    return np.fromfile(os.path.join(tmp, "dump", f"{fid}_{key}.bin"), dtype=dtype).reshape(shape)


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [1, 3])
def test_frame_loop_matches_oracle(tmp_path, channels):
    """The reference's default SGM configuration (num_disparities 256, MODE_HH4, min_disparity 4) through
    System -> DataSource -> modules, exactly as `./cart_slam <source.json> <modules.json>` would run it."""
    tmp = str(tmp_path)
    w, h, n = 400, 112, 4
    src, frames = make_dataset(tmp, n, w, h, channels)
    os.makedirs(os.path.join(tmp, "dump"))
    modules = MODULES_NAIVE_SEG[:2] + [{"type": "disparity_derivative"}] + MODULES_NAIVE_SEG[2:]
    modules[1] = dict(modules[1], update_interval=2, reset_interval=2, label_components=True)
    r = run_exe(src, modules, tmp, ("--dump", os.path.join(tmp, "dump"), "--sequential", "1"))
    assert r.returncode == 0, r.stderr
    assert "skipping GUI module type disparity_planeseg_visualization" in r.stderr
    cum = np.zeros(256, np.int64)
    params = (0, 0, 0, 0, 0, 0)
    for f, (l, rr) in enumerate(frames):
        fid = f + 1
        ed = O.disparity_module(l, rr, 256, 4, 4, radius=2, iterations=1)
        assert (load(tmp, fid, "disparity", np.int16, (h, w)) == ed).all(), f"disparity frame {fid}"
        dd, dh = O.directional_derivative(ed)
        assert (load(tmp, fid, "disparity_derivative", np.int16, (h, w, 2)) == dd).all()
        assert (load(tmp, fid, "disparity_derivative_histogram", np.int32, (256, 2)) == dh).all()
        pd, hist = O.plane_derivative(ed)
        cum += hist
        if fid % 2 == 1:
            h32 = cum.astype(np.int32)
            if fid % 4 == 1:
                cum[:] = 0
            _, params = O.histogram_peak_params(h32, params)
        ep = O.classify(pd, params)
        assert (load(tmp, fid, "planes", np.uint8, (h, w)) == ep).all(), f"planes frame {fid}"
        eids, en = O.ccl(ep)
        assert (load(tmp, fid, "plane_components", np.int32, (h, w)) == eids).all()
        et, _ = O.ccl_stats(ep, eids, max_components=4096)
        assert int(np.fromfile(os.path.join(tmp, "dump", f"{fid}_plane_component_count.bin"), np.int32)[0]) == en
        got = np.fromfile(os.path.join(tmp, "dump", f"{fid}_plane_component_table.bin"), np.int32).reshape(4096, 7)
        assert (got[:len(et)] == et).all(), f"component table frame {fid}"


@pytest.mark.gpu
@pytest.mark.parametrize("coalesce", ["2", "1", "0"])
def test_concurrent_frames_disparity(tmp_path, coalesce):
    """Up to 12 frames in flight through one module object (include/cartslam.hpp:4-5): disparity is frame-local, so
    every frame must still be bit-exact whatever the interleaving -- whether the module hands the frames that wait
    together to the engine as one launch sequence (CARTSLAM_COALESCE = groups on the GPU at once) or one by one (0)."""
    tmp = str(tmp_path)
    w, h, n, distinct = 320, 96, 30, 5
    src, frames = make_dataset(tmp, distinct, w, h)
    seq_dir = os.path.join(tmp, "dataset", "sequences", "00")
    for f in range(distinct, n):   # frames 5..29 repeat the first five: the oracle runs once per distinct frame
        for cam in ("image_2", "image_3"):
            os.link(os.path.join(seq_dir, cam, "%06d.pgm" % (f % distinct)), os.path.join(seq_dir, cam, "%06d.pgm" % f))
    os.makedirs(os.path.join(tmp, "dump"))
    tcsv = os.path.join(tmp, "timing.csv")
    r = run_exe(src, [{"type": "disparity", "num_disparities": 64, "paths": 8}], tmp, ("--dump", os.path.join(tmp, "dump"), "--timing", tcsv),
                env={"CARTSLAM_COALESCE": coalesce})
    assert r.returncode == 0, r.stderr
    per_launch = float(r.stdout.split("frames_per_launch")[1].split()[0])
    assert per_launch == 1.0 if coalesce == "0" else 1.0 <= per_launch <= 12.0, r.stdout
    rows = [ln.strip().split(";") for ln in open(tcsv)]
    assert rows[0][:6] == ["name", "run_id", "time_init", "time_start", "time_end", "duration_ms"]  # include/timing.hpp:59
    assert sorted(int(x[1]) for x in rows[1:] if x[0] == "ImageDisparity") == list(range(1, n + 1))
    assert sum(1 for x in rows[1:] if x[0] == "Frame") == n
    want = [O.disparity_module(l, rr, 64, 8, 4) for l, rr in frames]
    for f in range(n):
        assert (load(tmp, f + 1, "disparity", np.int16, (h, w)) == want[f % distinct]).all(), f"frame {f + 1}"


# ---------------------------------------------------------------- KITTI source (PNG + calib.txt) and depth module
P_ROWS = {  # KITTI odometry calib.txt layout: "P<i>: 12 numbers"
    0: [718.856, 0.0, 607.1928, 0.0, 0.0, 718.856, 185.2157, 0.0, 0.0, 0.0, 1.0, 0.0],
    1: [718.856, 0.0, 607.1928, -386.1448, 0.0, 718.856, 185.2157, 0.0, 0.0, 0.0, 1.0, 0.0],
    2: [718.856, 0.0, 607.1928, 45.38225, 0.0, 718.856, 185.2157, -0.1130887, 0.0, 0.0, 1.0, 0.003779761],
    3: [718.856, 0.0, 607.1928, -337.2877, 0.0, 718.856, 185.2157, 2.369057, 0.0, 0.0, 1.0, 0.004915215],
}


def make_kitti_png_dataset(tmp, n, w, h, color, seq=3, calib=True):
    from PIL import Image
    d = os.path.join(tmp, "kitti", "sequences", "%02d" % seq)
    os.makedirs(os.path.join(d, "image_2")); os.makedirs(os.path.join(d, "image_3"))
    frames = []
    for f in range(n):
        l, r, _ = synth.make_pair(w, h, 128, 4, seed=777, frame=f, channels=3 if color else 1)
        for cam, img in ((2, l), (3, r)):
            pil = Image.fromarray(img[..., ::-1].copy(), "RGB") if color else Image.fromarray(img, "L")
            pil.save(os.path.join(d, "image_%d" % cam, "%06d.png" % f))
        frames.append((l, r))
    if calib:
        with open(os.path.join(d, "calib.txt"), "w") as fh:
            for i, row in P_ROWS.items():
                fh.write("P%d: %s\n" % (i, " ".join(repr(v) for v in row)))
    src = os.path.join(tmp, "kitti_source.json")
    json.dump({"type": "kitti", "path": os.path.join(tmp, "kitti"), "sequence": seq}, open(src, "w"))
    return src, frames


def test_kitti_calibration_errors(tmp_path):
    tmp = str(tmp_path)
    src, _ = make_kitti_png_dataset(tmp, 1, 64, 32, color=False, calib=False)
    r = run_exe(src, [], tmp)
    assert r.returncode != 0 and "Failed to open calibration file" in r.stderr  # kitti.cpp:100-103
    with open(os.path.join(tmp, "kitti", "sequences", "03", "calib.txt"), "w") as fh:
        fh.write("P2: 1 2 3\nTr: 0 0 0\n")  # rows without exactly 12 numbers are ignored (kitti.cpp:77-79)
    r = run_exe(src, [], tmp)
    assert r.returncode != 0 and "Failed to read calibration file" in r.stderr  # kitti.cpp:126-128


@pytest.mark.gpu
@pytest.mark.parametrize("color", [True, False])
def test_kitti_png_source_and_depth(tmp_path, color):
    """image_2/image_3 PNGs + calib.txt -> Q (src/sources/kitti.cpp:89-149) -> disparity -> depth (src/modules/depth.cpp)."""
    tmp = str(tmp_path)
    w, h, n = 300, 100, 11 if color else 2   # 11 frames: more than the read-ahead depth
    src, frames = make_kitti_png_dataset(tmp, n, w, h, color)
    os.makedirs(os.path.join(tmp, "dump"))
    modules = [{"type": "disparity", "num_disparities": 128, "smoothing_radius": 2, "smoothing_iterations": 1}, {"type": "depth"},
               {"type": "depth_visualization"}]
    # colour files through the read-ahead decoder pool (default), gray files read inside getNext like the reference
    r = run_exe(src, modules, tmp, ("--dump", os.path.join(tmp, "dump")), env={"CARTSLAM_READAHEAD": "4" if color else "0"})
    assert r.returncode == 0, r.stderr
    Q = np.fromfile(os.path.join(tmp, "dump", "Q.bin"), np.float32).reshape(4, 4)
    assert np.array_equal(Q, O.kitti_q_matrix(P_ROWS[2], P_ROWS[3])), Q
    for f, (l, rr) in enumerate(frames):
        # cv::imread gives 3-channel BGR even for gray files; BGR2GRAY of a replicated gray is the identity
        ed = O.disparity_module(l, rr, 128, 4, 4, radius=2, iterations=1)
        assert (load(tmp, f + 1, "disparity", np.int16, (h, w)) == ed).all(), f"disparity frame {f + 1}"
        got = load(tmp, f + 1, "depth", np.float32, (h, w, 3))
        exp = O.reproject_depth(ed, Q)
        assert np.allclose(got, exp, rtol=1e-4, atol=1e-4), float(np.abs(got - exp).max())  # float path: 1e-4 (north_star)
        z = got[..., 2][ed > 64]
        assert np.isfinite(z).all() and (z > 0).all()


@pytest.mark.gpu
def test_kitti_source_resizes_to_the_configured_size(tmp_path):
    """KITTIDataSource with an image size other than the files' (kitti.hpp:11; kitti.cpp:137-148 scales Q, :169-172 resizes both
    images with cv::cuda::resize INTER_LINEAR, oracle S16).  The size comes from the `image_width` / `image_height` keys,
    an extension of the source JSON (the reference's factory never passes its ctor's imageSize)."""
    tmp = str(tmp_path)
    w, h, n, dw, dh = 300, 100, 3, 256, 96
    src, frames = make_kitti_png_dataset(tmp, n, w, h, color=True)
    cfg = json.load(open(src)); cfg.update(image_width=dw, image_height=dh); json.dump(cfg, open(src, "w"))
    os.makedirs(os.path.join(tmp, "dump"))
    r = run_exe(src, [{"type": "disparity", "num_disparities": 64, "smoothing_radius": 2, "smoothing_iterations": 1}], tmp,
                ("--dump", os.path.join(tmp, "dump")))
    assert r.returncode == 0, r.stderr
    Q = np.fromfile(os.path.join(tmp, "dump", "Q.bin"), np.float32).reshape(4, 4)
    assert np.array_equal(Q, O.kitti_q_matrix(P_ROWS[2], P_ROWS[3], np.float32(dw) / np.float32(w), np.float32(dh) / np.float32(h))), Q
    for f, (l, rr) in enumerate(frames):
        ed = O.disparity_module(O.resize_linear(l, dw, dh), O.resize_linear(rr, dw, dh), 64, 4, 4, radius=2, iterations=1)
        assert (load(tmp, f + 1, "disparity", np.int16, (dh, dw)) == ed).all(), f"disparity of resized frame {f + 1}"


@pytest.mark.gpu
def test_temporal_smoothing_frame_loop(tmp_path):
    """use_temporal_smoothing through the module API: cross-frame dependencies (planes_unsmoothed of runs -1..-3, optflow
    of runs 0..-2; include/modules/planeseg.hpp:127-143) and the tables of planeseg.cu:303-347, flow replayed from files."""
    tmp = str(tmp_path)
    w, h, n, dist = 200, 72, 6, 3
    src, frames = make_dataset(tmp, n, w, h)
    fdir = os.path.join(tmp, "dataset", "sequences", "00", "flow")
    os.makedirs(fdir)
    rng = np.random.default_rng(8)
    flows = []
    for f in range(n):
        fl = (rng.integers(-6, 7, (h, w, 2)) * 32 + rng.integers(0, 32, (h, w, 2))).astype(np.int16)
        fl.tofile(os.path.join(fdir, "%06d.bin" % f)); flows.append(fl)
    os.makedirs(os.path.join(tmp, "dump"))
    static = {"type": "static", "horizontal_range_min": 6, "horizontal_range_max": 18, "vertical_range_min": -5, "vertical_range_max": 6}
    modules = [{"type": "disparity", "num_disparities": 64, "paths": 8, "smoothing_radius": 2, "smoothing_iterations": 1},
               {"type": "optflow_file"},
               {"type": "disparity_planeseg", "parameter_provider": static, "use_temporal_smoothing": True, "temporal_smoothing_distance": dist}]
    r = run_exe(src, modules, tmp, ("--dump", os.path.join(tmp, "dump"), "--sequential", "1"))
    assert r.returncode == 0, r.stderr
    params = (6, 18, -5, 6, 12, 0)
    unsm = []
    for f, (l, rr) in enumerate(frames):
        fid = f + 1
        ed = O.disparity_module(l, rr, 64, 8, 4, radius=2, iterations=1)
        pd, _ = O.plane_derivative(ed)
        cur = O.classify(pd, params)
        unsm.append(cur)
        assert (load(tmp, fid, "planes_unsmoothed", np.uint8, (h, w)) == cur).all(), f"unsmoothed frame {fid}"
        if fid == 1:
            exp = cur
        else:
            k = min(dist, fid - 1)
            prev = [unsm[f - i] for i in range(1, k + 1)]          # frames id-1, id-2, ...
            fl = [flows[f - i] for i in range(0, k)]               # flow of frames id, id-1, ...
            exp = O.temporal_vote(cur, prev, fl)
        assert (load(tmp, fid, "planes", np.uint8, (h, w)) == exp).all(), f"smoothed frame {fid}"
    # bad parameters of the native "optflow" stand-in fail loudly
    r = run_exe(src, [{"type": "optflow", "search_radius": 40}], tmp)
    assert r.returncode != 0 and "search_radius must be in [1, 16]" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("temporal", [False, "file", "native"])
def test_superpixel_planeseg_frame_loop(tmp_path, temporal):
    """The reference's plane-segmentation configuration (config/modules/kitti-planeseg.json: superpixels, disparity,
    disparity_derivative, superpixel_disparity_planeseg with the histogram_peak provider; "optflow" replaced by the file
    replay) through the frame loop with up to 12 frames in flight: the stateful modules take frames in id order."""
    tmp = str(tmp_path)
    w, h, n, bs, reset = 320, 96, 7, 8, 4
    src, frames = make_dataset(tmp, n, w, h, channels=3)
    flows = []
    if temporal == "native":  # the "optflow" module type itself (census block matching, oracle S15)
        flows = [None] + [O.block_flow(O.bgr2gray(frames[f][0]), O.bgr2gray(frames[f - 1][0]), 4, 2) for f in range(1, n)]
    elif temporal:
        fdir = os.path.join(tmp, "dataset", "sequences", "00", "flow")
        os.makedirs(fdir)
        rng = np.random.default_rng(18)
        for f in range(n):
            fl = (rng.integers(-6, 7, (h, w, 2)) * 32 + rng.integers(0, 32, (h, w, 2))).astype(np.int16)
            fl.tofile(os.path.join(fdir, "%06d.bin" % f)); flows.append(fl)
    os.makedirs(os.path.join(tmp, "dump"))
    modules = [{"type": "superpixels", "initial_iterations": 5, "iterations": 2, "block_size": bs, "reset_iterations": reset},
               {"type": "disparity", "num_disparities": 64, "smoothing_radius": 2, "smoothing_iterations": 1},
               {"type": "disparity_derivative"},
               {"type": "superpixel_disparity_planeseg", "parameter_provider": {"type": "histogram_peak"}, "update_interval": 2, "reset_interval": 2,
                "use_temporal_smoothing": bool(temporal)},
               {"type": "disparity_planeseg_visualization", "show_histogram": False}]
    if temporal:
        modules.insert(1, {"type": "optflow", "search_radius": 4} if temporal == "native" else {"type": "optflow_file"})
    r = run_exe(src, modules, tmp, ("--dump", os.path.join(tmp, "dump")))
    assert r.returncode == 0, r.stderr
    sp = O.sp_params()  # JSON factory defaults (cartconfig.cpp:128-133)
    labels, mx = O.sp_block_init(w, h, bs, bs)
    running, params, unsm = None, (0, 0, 0, 0, 0, 0), []
    for f, (l, rr) in enumerate(frames):
        fid = f + 1
        ed = O.disparity_module(l, rr, 64, 4, 4, radius=2, iterations=1)
        d2, dh = O.directional_derivative(ed)
        # SuperPixelModule::runInternal (superpixels.cu:92-115)
        iters = 5 if (fid == 1 or fid % reset == 0) else 2
        if fid % reset == 0:
            labels, mx = O.sp_block_init(w, h, bs, bs)
        labels, _ = O.sp_relax(sp, labels, mx, O.bgr2ycrcb(l), d2, iters)
        assert (load(tmp, fid, "superpixels", np.uint16, (h, w)) == labels).all(), f"superpixels frame {fid}"
        assert int(np.fromfile(os.path.join(tmp, "dump", f"{fid}_superpixels_max_label.bin"), np.uint16)[0]) == mx
        # SuperPixelDisparityPlaneSegmentationModule::updatePlaneParameters (sp_planeseg.cu:349-387)
        hist = dh[:, 0].astype(np.int64)
        if running is None:
            running = np.zeros(256, np.int64)
        else:
            running += hist
            hist = running.copy()
        if fid % 4 == 1:
            running[:] = 0
        if fid % 2 == 1:
            _, params = O.histogram_peak_params(hist.astype(np.int32), params)
        prev, fl = [], []
        if temporal and fid > 1:
            k = min(3, fid - 1)
            prev = [unsm[f - i] for i in range(1, k + 1)]
            fl = [flows[f - i] for i in range(0, k)]
        if temporal == "native" and fid > 1:
            assert (load(tmp, fid, "optflow", np.int16, (h, w, 2)) == flows[f]).all(), f"optflow frame {fid}"
        eu, ep = O.sp_classify(d2, labels, mx, params, prev, fl)
        unsm.append(eu)
        assert (load(tmp, fid, "planes_unsmoothed", np.uint8, (h, w)) == eu).all(), f"unsmoothed frame {fid}"
        assert (load(tmp, fid, "planes", np.uint8, (h, w)) == ep).all(), f"planes frame {fid}"
    assert len({tuple(np.unique(load(tmp, i + 1, "planes", np.uint8, (h, w)))) for i in range(n)} - {(2,)}) >= 1  # not all UNKNOWN
