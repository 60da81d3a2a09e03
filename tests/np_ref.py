"""Second, independent restatement of the hot path in numpy / pure Python (small images only).

Used only to cross-check the C oracle (oracle/cart_oracle.c): it is written from the spec in
oracle/cart_oracle.h (S1..S12), not from the C code -- array shifts and per-pixel numpy ops
instead of scan-line walkers -- so a transcription slip in either shows up as a mismatch.
"""
import numpy as np

INVALID = -32768


def bgr2gray(bgr):
    b, g, r = (bgr[..., i].astype(np.uint32) for i in range(3))
    return ((1868 * b + 9617 * g + 4899 * r + 8192) >> 14).astype(np.uint8)


def census(gray):
    h, w = gray.shape
    g = gray.astype(np.int32)
    pad = np.zeros((h + 6, w + 8), np.int32)
    pad[3:-3, 4:-4] = g

    def sh(dy, dx):
        return pad[3 + dy:3 + dy + h, 4 + dx:4 + dx + w]

    f = np.zeros((h, w), np.uint32)
    pairs = [(dy, dx) for dy in (-3, -2, -1) for dx in range(-4, 5)] + [(0, dx) for dx in range(-4, 0)]
    assert len(pairs) == 31
    for dy, dx in pairs:
        f = (f << np.uint32(1)) | (sh(dy, dx) > sh(-dy, -dx)).astype(np.uint32)
    inner = np.zeros((h, w), bool)
    inner[3:h - 3, 4:w - 4] = True
    return np.where(inner, f, 0).astype(np.uint32)


def cost_volume(cl, cr, D, min_disp):
    h, w = cl.shape
    C = np.empty((h, w, D), np.int32)
    xs = np.arange(w)
    for d in range(D):
        xr = xs - d - min_disp
        ok = (xr >= 0) & (xr < w)
        fr = np.where(ok[None, :], cr[:, np.clip(xr, 0, w - 1)], 0).astype(np.uint32)
        x = cl ^ fr
        C[:, :, d] = np.array([bin(int(v)).count("1") for v in x.ravel()], np.int32).reshape(h, w)
    return C


def aggregate(C, dx, dy, p1, p2):
    h, w, D = C.shape
    L = np.zeros((h, w, D), np.int32)
    ys = range(h) if dy >= 0 else range(h - 1, -1, -1)
    xs = range(w) if dx >= 0 else range(w - 1, -1, -1)
    big = 1 << 20
    for y in ys:
        for x in xs:
            py, px = y - dy, x - dx
            if 0 <= py < h and 0 <= px < w:
                prev = L[py, px]
            else:
                prev = np.zeros(D, np.int32)
            m = prev.min()
            lo = np.concatenate(([big], prev[:-1])) + p1
            hi = np.concatenate((prev[1:], [big])) + p1
            best = np.minimum(np.minimum(prev, lo), np.minimum(hi, m + p2))
            L[y, x] = C[y, x] + best - m
    return L


DIRS = [(0, 1), (0, -1), (1, 0), (-1, 0), (1, 1), (-1, 1), (-1, -1), (1, -1)]


def wta(S, uniqueness_ratio):
    h, w, D = S.shape
    u = np.float32(100 - uniqueness_ratio) / np.float32(100.0)
    left = np.empty((h, w), np.uint16)
    right = np.empty((h, w), np.uint16)
    for y in range(h):
        for x in range(w):
            s = S[y, x].astype(np.int64)
            bd = int(np.argmin(s))  # first minimum
            bc = int(s[bd])
            lhs = s.astype(np.float32) * u
            ok = (lhs >= np.float32(bc)) | (np.abs(np.arange(D) - bd) <= 1)
            if not ok.all():
                left[y, x] = 0xFFFF
                continue
            sub = bd * 16
            if 0 < bd < D - 1:
                num = int(s[bd - 1] - s[bd + 1]); den = int(s[bd - 1] - 2 * bc + s[bd + 1])
                if den != 0:
                    q = num * 16 + den
                    sub += int(abs(q) // (2 * den)) * (1 if q >= 0 else -1)  # C truncation (den > 0)
            left[y, x] = sub & 0xFFFF
        for p in range(w):
            n = min(D, w - p)
            diag = np.array([S[y, p + d, d] for d in range(n)], np.int64)
            right[y, p] = int(np.argmin(diag))
    return left, right


def median3x3(a):
    h, w = a.shape
    out = a.copy()
    if h >= 3 and w >= 3:
        st = np.stack([a[1 + dy:h - 1 + dy, 1 + dx:w - 1 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1)])
        out[1:-1, 1:-1] = np.sort(st, axis=0)[4]
    return out


def lr_check_range(lm, rm, gray, min_disp):
    h, w = lm.shape
    out = np.empty((h, w), np.int16)
    for y in range(h):
        for x in range(w):
            org = int(lm[y, x])
            bad = gray[y, x] == 0 or org == 0xFFFF
            if not bad:
                d = org >> 4
                k = x - d
                if 0 <= k < w and abs(int(rm[y, k]) - d) > 1:
                    bad = True
            out[y, x] = (min_disp - 1) * 16 if bad else org + min_disp * 16
    return out


def sgm(gl, gr, D, paths, min_disp=4, p1=10, p2=120, uniq=12):
    cl, cr = census(gl), census(gr)
    C = cost_volume(cl, cr, D, min_disp)
    S = np.zeros(C.shape, np.int64)
    Ls = []
    for dx, dy in DIRS[:paths]:
        L = aggregate(C, dx, dy, p1, p2)
        Ls.append(L)
        S += L
    wl, wr = wta(S, uniq)
    disp = lr_check_range(median3x3(wl), median3x3(wr), gl, min_disp)
    return dict(census_l=cl, census_r=cr, paths=Ls, S=S, wta_l=wl, wta_r=wr, disp=disp)


def interpolate(disp, radius, iterations, min16, maxd):
    h, w = disp.shape
    a = disp.astype(np.int64)
    for _ in range(iterations):
        b = np.empty_like(a)
        for y in range(h):
            for x in range(w):
                win = a[max(0, y - radius + 1):min(h, y + radius), max(0, x - radius + 1):min(w, x + radius)]
                v = win[(win > min16) & (win < maxd)]
                b[y, x] = int(v.sum()) // v.size if v.size > radius * radius + 1 else INVALID
        a = b
    return a.astype(np.int16)


def _wrap16(v):
    return ((np.asarray(v, np.int64) + 32768) % 65536 - 32768)


def directional_derivative(disp):
    h, w = disp.shape
    d = disp.astype(np.int64)
    out = np.full((h, w, 2), INVALID, np.int64)
    hist = np.zeros((256, 2), np.int64)
    for y in range(h):
        for x in range(w):
            if 2 <= y < h - 2 and d[y + 2, x] != INVALID and d[y - 2, x] != INVALID:
                v = int(_wrap16(d[y + 2, x] - d[y - 2, x])); out[y, x, 0] = v
                if -128 <= v <= 127: hist[v + 128, 0] += 1
            if 2 <= x < w - 2 and d[y, x + 2] != INVALID and d[y, x - 2] != INVALID:
                v = int(_wrap16(d[y, x + 2] - d[y, x - 2])); out[y, x, 1] = v
                if -128 <= v <= 127: hist[v + 128, 1] += 1
    return out.astype(np.int16), hist.astype(np.int32)


def plane_derivative(disp):
    h, w = disp.shape
    d = disp.astype(np.int64)
    lp = np.full((h, w), INVALID, np.int64)
    for y in range(h):
        for x in range(w):
            col = d[max(0, y - 2):min(h, y + 3), x]
            v = col[col != INVALID]
            if v.size:
                s = int(_wrap16(v.sum()))  # sequential s16 wrap == wrap of the total
                q = abs(s) // v.size
                lp[y, x] = q if s >= 0 else -q
    out = np.full((h, w), INVALID, np.int64)
    hist = np.zeros(256, np.int64)
    for y in range(1, h - 1):
        for x in range(w):
            if lp[y - 1, x] != INVALID and lp[y, x] != INVALID and lp[y + 1, x] != INVALID:
                v = int(_wrap16(lp[y + 1, x] - lp[y - 1, x])); out[y, x] = v
                if -128 <= v <= 127: hist[v + 128] += 1
    return out.astype(np.int16), hist.astype(np.int32)


def classify(deriv, params):
    hmin, hmax, vmin, vmax = params[:4]
    d = deriv.astype(np.int64)
    ok = d != INVALID
    out = np.full(d.shape, 2, np.uint8)
    isv = ok & (d >= vmin) & (d < vmax)
    ish = ok & (d >= hmin) & (d < hmax)
    out[isv] = 1
    out[ish] = 0  # horizontal test comes first in the reference (planeseg.cu:191-195)
    return out


def ccl(planes):
    """flood fill; id = min linear index."""
    h, w = planes.shape
    ids = np.full((h, w), -1, np.int32)
    seen = np.zeros((h, w), bool)
    n = 0
    for y in range(h):
        for x in range(w):
            if planes[y, x] > 1 or seen[y, x]:
                continue
            n += 1
            c = planes[y, x]; root = y * w + x
            stack = [(y, x)]; seen[y, x] = True
            while stack:
                cy, cx = stack.pop()
                ids[cy, cx] = root
                for ny, nx in ((cy - 1, cx), (cy + 1, cx), (cy, cx - 1), (cy, cx + 1)):
                    if 0 <= ny < h and 0 <= nx < w and not seen[ny, nx] and planes[ny, nx] == c:
                        seen[ny, nx] = True; stack.append((ny, nx))
    return ids, n


def find_peaks(data):
    """0-dim persistent homology of a 1-D signal (watershed from the top), spec S11."""
    data = [int(v) for v in data]
    n = len(data)
    order = sorted(range(n), key=lambda i: (-data[i], i))
    comp = [-1] * n
    peaks = []  # dicts
    for idx in order:
        l = comp[idx - 1] if idx > 0 else -1
        r = comp[idx + 1] if idx < n - 1 else -1
        if l < 0 and r < 0:
            peaks.append(dict(born=idx, left=idx, right=idx, died=-1)); comp[idx] = len(peaks) - 1
        elif l >= 0 and r < 0:
            peaks[l]["right"] += 1; comp[idx] = l
        elif l < 0 and r >= 0:
            peaks[r]["left"] -= 1; comp[idx] = r
        else:
            if data[peaks[l]["born"]] > data[peaks[r]["born"]]:
                peaks[r]["died"] = idx; peaks[l]["right"] = peaks[r]["right"]
                comp[peaks[l]["right"]] = comp[idx] = l
            else:
                peaks[l]["died"] = idx; peaks[r]["left"] = peaks[l]["left"]
                comp[peaks[r]["left"]] = comp[idx] = r
    def pers(p):
        return (1 << 31) - 1 if p["died"] < 0 else data[p["born"]] - data[p["died"]]
    peaks.sort(key=lambda p: -pers(p))  # python's sort is stable
    return [(p["born"], p["died"], p["left"], p["right"]) for p in peaks]


# ---- superpixels (S13 / S14): pure Python, tiny images only --------------------------------------------------------
def bgr2ycrcb(bgr):
    b, g, r = (bgr[..., i].astype(np.int64) for i in range(3))
    y = (1868 * b + 9617 * g + 4899 * r + 8192) >> 14
    cr = ((r - y) * 11682 + (128 << 14) + 8192) >> 14
    cb = ((b - y) * 9241 + (128 << 14) + 8192) >> 14
    return np.stack([np.clip(c, 0, 255) for c in (y, cr, cb)], axis=-1).astype(np.uint8)


def spec_log(x):
    """S13 log, written from the header text (Python floats are IEEE doubles, no contraction)."""
    import struct
    bits = struct.unpack("<Q", struct.pack("<d", x))[0]
    e = (bits >> 52) - 1023
    m = struct.unpack("<d", struct.pack("<Q", (bits & ((1 << 52) - 1)) | (1023 << 52)))[0]
    if m > float.fromhex("0x1.6a09e667f3bcdp+0"):
        m, e = m * 0.5, e + 1
    s = (m - 1.0) / (m + 1.0)
    z = s * s
    p = 1.0 / 23.0
    for k in range(10, -1, -1):
        p = p * z + 1.0 / float(2 * k + 1)
    r = (2.0 * s) * p
    return float(e) * float.fromhex("0x1.62e42fee00000p-1") + (r + float(e) * float.fromhex("0x1.a39ef35793c76p-33"))


def _gauss(n, s, q):
    if n == 0:
        return 0.0
    a, b = float(q) / float(n), float(s) / float(n)
    var = a - b * b
    if not var >= 1.0 / 12.0:
        var = 1.0 / 12.0
    return (float(n) / 2 * spec_log(float.fromhex("0x1.921fb54442d18p+2") * var)) + (float(n) / 2)


def _compact(n, s, q):
    if n == 0:
        return 0.0
    return float(q) - (float(s) * float(s)) / float(n)


def sp_relax(labels, ycrcb, deriv2, iterations, direct=0.5, diagonal=0.5 / np.sqrt(2.0), compactness=0.1, progressive=0.0,
             image=1.5, disparity=1.0):
    """Statistics are rebuilt from scratch before every iteration (the oracle updates them incrementally)."""
    lab = labels.astype(np.int64).copy()
    h, w = lab.shape
    groups = []  # (weight, divisor, channel value getters, cost fn, is_compactness)
    if compactness > 0:
        groups.append((compactness, None, [lambda x, y: x, lambda x, y: y], _compact, True))
    if disparity > 0:
        groups.append((disparity, 2.0, [lambda x, y, c=c: int(deriv2[y, x, c]) for c in range(2)], _gauss, False))
    if image > 0:
        groups.append((image, 3.0, [lambda x, y, c=c: int(ycrcb[y, x, c]) for c in range(3)], _gauss, False))
    for _ in range(iterations):
        cnt = {}
        sums = [[{} for _ in g[2]] for g in groups]
        sqs = [[{} for _ in g[2]] for g in groups]
        for y in range(h):
            for x in range(w):
                L = int(lab[y, x])
                cnt[L] = cnt.get(L, 0) + 1
                for gi, g in enumerate(groups):
                    for ci, get in enumerate(g[2]):
                        v = get(x, y)
                        sums[gi][ci][L] = sums[gi][ci].get(L, 0) + v
                        sqs[gi][ci][L] = sqs[gi][ci].get(L, 0) + v * v
        new = lab.copy()
        for y in range(h):
            for x in range(w):
                nb = {}
                order = []
                for dx in (-1, 0, 1):
                    for dy in (-1, 0, 1):
                        xx, yy = x + dx, y + dy
                        if 0 <= xx < w and 0 <= yy < h:
                            L = int(lab[yy, xx])
                            nb[(dx, dy)] = L
                            if L not in order:
                                order.append(L)
                O = int(lab[y, x])
                if len(order) < 2:
                    continue
                best, best_cost = O, None
                for P in order:
                    nd = sum(1 for k in ((-1, 0), (1, 0), (0, -1), (0, 1)) if k in nb and nb[k] != P)
                    ng = sum(1 for k in ((-1, -1), (-1, 1), (1, -1), (1, 1)) if k in nb and nb[k] != P)
                    cost = nd * direct + ng * diagonal
                    for gi, (wgt, div, getters, fn, is_c) in enumerate(groups):
                        f = 0.0
                        for L in order:
                            n = cnt[L]
                            delta = 0
                            if O != P and L == O:
                                delta = -1
                            elif O != P and L == P:
                                delta = 1
                            n += delta
                            if n == 0:
                                continue
                            ks = []
                            for ci, get in enumerate(getters):
                                v = get(x, y)
                                ks.append(fn(n, sums[gi][ci][L] + delta * v, sqs[gi][ci][L] + delta * v * v))
                            if is_c:
                                f += ks[0] + ks[1]
                            else:
                                for k in ks:
                                    f += k
                        if is_c:
                            if progressive > 0.0:
                                f *= 1.0 + progressive * (float(h) - float(y)) / float(h)
                            cost += wgt * f
                        else:
                            cost += wgt * (f / div)
                    if best_cost is None or cost < best_cost:
                        best, best_cost = P, cost
                new[y, x] = best
        lab = new
    return lab.astype(np.uint16)


def sp_classify(deriv2, labels, max_label, params, prev_planes=(), flows=()):
    hmin, hmax, vmin, vmax = params[:4]
    h, w = labels.shape
    d = deriv2[..., 0].astype(np.int64)
    uns = np.full((h, w), 2, np.uint8)
    uns[(d != INVALID) & (d >= vmin) & (d < vmax)] = 1
    uns[(d != INVALID) & (d >= hmin) & (d < hmax)] = 0
    voted = uns.copy()
    if len(prev_planes):
        for y in range(h):
            for x in range(w):
                v = [0, 0, 0]
                v[uns[y, x]] += 2
                px, py = x, y
                for k in range(len(prev_planes)):
                    px -= int(flows[k][y, x, 0]) >> 5
                    py -= int(flows[k][y, x, 1]) >> 5
                    if 0 <= px < w and 0 <= py < h:
                        v[prev_planes[k][py, px]] += 1
                p = 0 if v[0] > v[1] else 1
                voted[y, x] = 2 if v[p] < v[2] else p
    votes = np.zeros((max_label, 3), np.int64)
    np.add.at(votes, (labels.astype(np.int64).ravel(), voted.astype(np.int64).ravel()), 1)
    votes &= 0xFFFF
    assign = np.full(max_label, 2, np.uint8)
    mx = votes[:, 2].copy()
    sel = votes[:, 1] > mx
    assign[sel] = 1
    mx[sel] = votes[sel, 1]
    assign[votes[:, 0] > mx] = 0
    return uns, assign[labels.astype(np.int64)]


# ---- S15 optical flow: array-shift formulation with a summed-area table ----------------------------------------------
def block_flow(cen_cur, cen_prev, radius, block):
    h, w = cen_cur.shape
    R, B = radius, block
    padp = np.zeros((h + 2 * R, w + 2 * R), np.uint32)
    padp[R:R + h, R:R + w] = cen_prev

    def cost(u, v):
        shifted = padp[R - v:R - v + h, R - u:R - u + w]          # cenP(q - (u, v)), zero outside the image
        x = cen_cur ^ shifted
        ham = np.zeros((h, w), np.int64)
        for k in range(32):
            ham += (x >> np.uint32(k)) & np.uint32(1)
        sat = np.zeros((h + 1, w + 1), np.int64)
        sat[1:, 1:] = ham.cumsum(0).cumsum(1)
        y0 = np.clip(np.arange(h) - B, 0, h); y1 = np.clip(np.arange(h) + B + 1, 0, h)
        x0 = np.clip(np.arange(w) - B, 0, w); x1 = np.clip(np.arange(w) + B + 1, 0, w)
        return sat[y1][:, x1] - sat[y0][:, x1] - sat[y1][:, x0] + sat[y0][:, x0]

    best = cost(0, 0)
    bu = np.zeros((h, w), np.int64); bv = np.zeros((h, w), np.int64)
    for v in range(-R, R + 1):
        for u in range(-R, R + 1):
            c = cost(u, v)
            upd = c < best
            best = np.where(upd, c, best); bu = np.where(upd, u, bu); bv = np.where(upd, v, bv)
    return np.stack([bu * 32, bv * 32], axis=-1).astype(np.int16)
