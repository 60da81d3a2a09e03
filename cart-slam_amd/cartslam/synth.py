"""Seeded synthetic road-scene stereo pairs: no KITTI data is available offline.

Left image = 4 octaves of value noise (+ small gaussian-like noise); ground-truth disparity = sky
above the horizon, a ground plane below it (constant non-zero vertical disparity derivative ->
"horizontal plane" histogram peak) and fronto-parallel boxes (zero derivative -> "vertical
plane" peak); the right image is the left one forward-warped by the rounded disparity, nearer
surfaces winning, holes filled from the left neighbour.  Pure numpy, deterministic in (seed, frame).

`scene` adds content that street scenes have and value noise does not (the nearest substitute for KITTI
frames, which are not available offline):
  "stripes"    a facade of exactly periodic vertical stripes, periods 8 / 16 / 24 px: repetitive texture, uniqueness rejections;
  "saturated"  a saturated (255) and a black (0) patch in the left image: gray == 0 is the LR check's mask (oracle S8);
  "pole"       a 1-px-wide and a 3-px-wide near pole in front of the ground plane: thin structures;
  "wall"       a large textureless fronto-parallel wall;
  "photometric" the right camera sees the scene 12 % brighter with an offset and sensor noise of its own (clipped at 255):
               what two real cameras do to each other, and what the census transform is there to absorb.
"""
import numpy as np

DEFAULT_SEED = 0x5EEDCA2751A40001
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _hash01(ix, iy, salt):
    """uniform [0,1) from integer lattice coordinates."""
    with np.errstate(over="ignore"):
        k = (ix.astype(np.uint64) * np.uint64(0x9E3779B1) + iy.astype(np.uint64) * np.uint64(0x85EBCA77)
             + np.uint64(salt & 0xFFFFFFFFFFFFFFFF))
        h = _splitmix64(k)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _value_noise(w, h, cell, salt, xoff=0):
    xs = (np.arange(w, dtype=np.float64) + xoff) / cell
    ys = np.arange(h, dtype=np.float64) / cell
    x0 = np.floor(xs).astype(np.int64); y0 = np.floor(ys).astype(np.int64)
    fx = xs - x0; fy = ys - y0
    fx = fx * fx * (3 - 2 * fx); fy = fy * fy * (3 - 2 * fy)
    X0, Y0 = np.meshgrid(x0 + (1 << 20), y0 + (1 << 20))
    FX, FY = np.meshgrid(fx, fy)
    v00 = _hash01(X0, Y0, salt); v10 = _hash01(X0 + 1, Y0, salt)
    v01 = _hash01(X0, Y0 + 1, salt); v11 = _hash01(X0 + 1, Y0 + 1, salt)
    return (v00 * (1 - FX) + v10 * FX) * (1 - FY) + (v01 * (1 - FX) + v11 * FX) * FY


def _texture(w, h, salt, xoff):
    img = np.full((h, w), 128.0)
    for o, (cell, amp) in enumerate(((48.0, 64.0), (16.0, 32.0), (6.0, 16.0), (2.5, 8.0))):
        img += amp * (2.0 * _value_noise(w, h, cell, salt + 7919 * (o + 1), xoff) - 1.0)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.int64), np.arange(w, dtype=np.int64), indexing="ij")
    n = sum(_hash01(xx + int(xoff), yy, salt + 104729 * (k + 1)) for k in range(4)) - 2.0  # ~N(0, 0.58)
    img += n * (2.0 / 0.577)
    return np.clip(np.rint(img), 1, 255).astype(np.uint8)  # never 0: gray==0 is the reference's LR-check mask


SCENES = ("road", "stripes", "saturated", "pole", "wall", "photometric")


def _scene_regions(w, h, scene, frame):
    """(y0, y1, x0, x1) boxes of a scene's extra content; they move with the boxes of the road scene (2 px per frame)."""
    s = 2 * frame
    if scene == "stripes":
        return [(int(0.30 * h), int(0.80 * h), max(0, int(0.08 * w) - s), max(0, int(0.46 * w) - s))]
    if scene == "saturated":
        return [(int(0.50 * h), int(0.70 * h), max(0, int(0.15 * w) - s), max(0, int(0.30 * w) - s)),
                (int(0.62 * h), int(0.92 * h), max(0, int(0.55 * w) - s), max(0, int(0.75 * w) - s))]
    if scene == "pole":
        x1, x3 = max(0, int(0.37 * w) - s), max(0, int(0.61 * w) - s)
        return [(int(0.25 * h), int(0.85 * h), x1, x1 + 1), (int(0.30 * h), int(0.90 * h), x3, x3 + 3)]
    if scene == "wall":
        return [(int(0.20 * h), int(0.75 * h), max(0, int(0.50 * w) - s), max(0, int(0.92 * w) - s))]
    return []


def ground_truth_disparity(w, h, D, min_disp=4, seed=DEFAULT_SEED, frame=0, scene="road"):
    """float disparity map of the LEFT view (pixels)."""
    if scene not in SCENES:
        raise ValueError(f"unknown scene {scene!r}: one of {SCENES}")
    dmax = min(min_disp + D - 2, 76)
    horizon = 0.45 * h
    y = np.arange(h, dtype=np.float64)[:, None]
    d = np.where(y < horizon, min_disp + 1.0, min_disp + 1.0 + (y - horizon) * (dmax - min_disp - 1.0) / (0.55 * h))
    d = np.repeat(d, w, axis=1)
    rng_salt = seed ^ 0xB0B0
    for b in range(6):
        u = _hash01(np.array([b] * 5), np.arange(5), rng_salt)
        bw = int(w * (0.06 + 0.10 * u[0])); bh = int(h * (0.15 + 0.25 * u[1]))
        cx = int(u[2] * (w - bw)) - 2 * frame
        by1 = int(horizon + u[3] * (h - horizon) * 0.8); by0 = max(0, by1 - bh)
        # a box standing on the ground plane has the ground's disparity at its foot
        db = d[min(by1, h - 1), 0]
        x0, x1 = max(0, cx), min(w, cx + bw)
        if x1 > x0 and by1 > by0:
            d[by0:by1, x0:x1] = np.maximum(d[by0:by1, x0:x1], db)
    if scene in ("stripes", "pole", "wall"):   # fronto-parallel surfaces standing on the ground: the ground's disparity at their foot
        for y0, y1, x0, x1 in _scene_regions(w, h, scene, frame):
            if x1 > x0:
                d[y0:y1, x0:x1] = np.maximum(d[y0:y1, x0:x1], d[min(y1, h - 1), 0])
    return d


def _paint_scene(tex, scene, w, h, frame):
    """The scene's extra content painted into the LEFT texture (the right image is warped from it afterwards)."""
    regions = _scene_regions(w, h, scene, frame)
    if scene == "stripes":
        (y0, y1, x0, x1), = regions
        third = max(1, (y1 - y0) // 3)
        xs = np.arange(x0, x1) + 2 * frame   # stripes fixed to the facade, which moves with the scene
        for k, period in enumerate((8, 16, 24)):
            ya, yb = y0 + k * third, (y1 if k == 2 else y0 + (k + 1) * third)
            # exactly periodic, no noise: the census of a stripe pixel repeats every `period` columns, so the matching cost has
            # one minimum per period and only the aggregation's smoothness term (or nothing) picks among them
            tex[ya:yb, x0:x1] = np.where((xs // (period // 2)) % 2 == 0, 70, 185).astype(np.uint8)[None, :]
    elif scene == "saturated":
        (ya, yb, xa, xb), (yc, yd, xc, xd) = regions
        tex[ya:yb, xa:xb] = 255
        tex[yc:yd, xc:xd] = 0      # gray == 0: the reference's LR-check mask (oracle S8)
    elif scene == "pole":
        for y0, y1, x0, x1 in regions:
            tex[y0:y1, x0:x1] = 245
    elif scene == "wall":
        (y0, y1, x0, x1), = regions
        tex[y0:y1, x0:x1] = 141
    return tex


def make_pair(w, h, D, min_disp=4, seed=DEFAULT_SEED, frame=0, channels=1, scene="road"):
    """-> (left, right, gt_disp): uint8 [h,w] (channels=1) or [h,w,3] BGR; gt float64 [h,w]."""
    gt = ground_truth_disparity(w, h, D, min_disp, seed, frame, scene)
    di = np.rint(gt).astype(np.int64)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.int64), np.arange(w, dtype=np.int64), indexing="ij")
    xr = xx - di
    ok = xr >= 0
    key = (di << 20) | xx  # nearer surface (larger d) wins; ties -> larger x
    best = np.full((h, w), -1, dtype=np.int64)
    np.maximum.at(best, (yy[ok], xr[ok]), key[ok])
    src = best & ((1 << 20) - 1)
    hole = best < 0
    idx = np.where(hole, 0, np.arange(w)[None, :])
    idx = np.maximum.accumulate(idx, axis=1)  # fill holes from the left neighbour
    src = np.take_along_axis(np.where(hole, 0, src), idx, axis=1)
    lefts, rights = [], []
    for c in range(channels):
        tex = _paint_scene(_texture(w, h, seed + 1000003 * c, 2 * frame), scene, w, h, frame)
        lefts.append(tex)
        right = np.take_along_axis(tex, src, axis=1)
        if scene == "photometric":   # gain, offset and independent noise in the right camera
            yy2, xx2 = np.meshgrid(np.arange(h, dtype=np.int64), np.arange(w, dtype=np.int64), indexing="ij")
            noise = sum(_hash01(xx2 + 7 * frame, yy2, seed + 15485863 * (k + 1) + c) for k in range(4)) - 2.0   # ~N(0, 0.58)
            right = np.clip(np.rint(right.astype(np.float64) * 1.12 + 6.0 + noise * (3.0 / 0.577)), 0, 255).astype(np.uint8)
        rights.append(right)
    if channels == 1:
        return lefts[0], rights[0], gt
    return np.stack(lefts, axis=-1), np.stack(rights, axis=-1), gt


def make_batch(n, w, h, D, min_disp=4, seed=DEFAULT_SEED, channels=1, first_frame=0, scene="road"):
    ls, rs = [], []
    for f in range(n):
        l, r, _ = make_pair(w, h, D, min_disp, seed, first_frame + f, channels, scene)
        ls.append(l); rs.append(r)
    return np.stack(ls), np.stack(rs)
