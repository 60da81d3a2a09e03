#!/usr/bin/env python3
"""Round 5 A/B builds of the aggregation launch (does the number of direction x frame row fronts that are resident at a
time -- the launch's instantaneous write window -- bind it?).  The product source is not touched: every variant is a patched COPY of
cart-slam_amd/csrc/sgm_kernels.hip, compiled into cart-slam_amd/build/ab/<name>/libcart_engine.so (select with CART_ENGINE_LIB=<path>).

  res<k>   at most k aggregation workgroups resident per CU at the headline (the product leaves 7-8 direction launches of 16 frames uncapped)
  fm       vertical / diagonal directions in FRAME-major order behind the horizontal scans ([frame][direction][block]); XCD placement as in the product
  fmnox    the same without the per-XCD frame placement
  g4nox    no XCD placement, frames in groups of 4, direction-major inside a group ([group][direction][frame][block])
  privcen   every XCD reads its own private copy of the census planes: cart_engine.hip patched as well

usage: r05_variants.py <name> [<name> ...]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "cart-slam_amd")
HIPCC = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wall", "-Wno-unused-result",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc")]


def sub(src, old, new, count=1):
    assert src.count(old) >= 1, "anchor not found: " + old[:70]
    return src.replace(old, new, count)


DECODE_OLD = """    int di = 0;
    for (int i = 1; i < a.ndirs; ++i)
        if (bid >= a.dirs[i].blk0 * nfr) di = i;
"""
FRAME_OLD = """    const int rb = bid - a.dirs[di].blk0 * nfr;
    const int frame = frame0 + fstep * (rb / nblk);
    const int bl = rb - (rb / nblk) * nblk;   // block inside the frame's share of this direction
"""


def order_variant(src, group):
    """group = 0: whole launch frame-major behind the horizontal scans; group = g: frames in groups of g, direction-major inside a group"""
    src = sub(src, DECODE_OLD, """    int di = 0, ab_frame = -1, ab_bl = 0;
    {
        int nh = 0;
        while (nh < a.ndirs && a.dirs[nh].dy == 0) ++nh;
        const int bh = nh < a.ndirs ? a.dirs[nh].blk0 : a.blocks_per_frame, bv = a.blocks_per_frame - bh;
        if (bid < bh * nfr || bv == 0) {
            for (int i = 1; i < nh; ++i)
                if (bid >= a.dirs[i].blk0 * nfr) di = i;
        } else {
            const int rem = bid - bh * nfr, G = %d > 0 ? min(%d, nfr) : 1;
            const int grp = rem / (bv * G), ing = rem - grp * (bv * G);           // frame group, block inside the group
            // inside a group: direction-major over its G frames
            di = nh;
            for (int i = nh + 1; i < a.ndirs; ++i)
                if (ing >= (a.dirs[i].blk0 - bh) * G) di = i;
            const int nb = (i_nblk(a, di));
            const int r2 = ing - (a.dirs[di].blk0 - bh) * G;
            ab_frame = grp * G + r2 / nb; ab_bl = r2 - (r2 / nb) * nb;
        }
    }
""" % (group, group))
    src = sub(src, FRAME_OLD, """    const int rb = bid - a.dirs[di].blk0 * nfr;
    const int frame = frame0 + fstep * (ab_frame >= 0 ? ab_frame : rb / nblk);
    const int bl = ab_frame >= 0 ? ab_bl : rb - (rb / nblk) * nblk;   // block inside the frame's share of this direction
""")
    # helper: blocks of a non-horizontal direction (full-size workgroups)
    src = sub(src, "constexpr int kAggWaves = 4;", "constexpr int kAggWaves = 4;\n#define i_nblk(a, di) (((a).dirs[di].nlines + LINES_PER_BLOCK - 1) / LINES_PER_BLOCK)")
    return src


def build(name):
    src = open(os.path.join(PKG, "csrc", "sgm_kernels.hip")).read()
    eng = None
    extra_hdr = None
    if name.startswith("res"):
        k = int(name[3:])
        src = sub(src, "    return (D >= 256 || n_frames < 16) ? 4 : 0;", "    return (D >= 256 || n_frames < 16) ? 4 : %d;" % k)
    elif name == "fm":
        src = order_variant(src, 0)
    elif name == "fmnox":
        src = order_variant(src, 0)
        src = sub(src, "    return n_frames > 0 && (n_frames & 7) == 0 && g.census_elems * 8 <= (size_t)(8u << 20);", "    return false;")
    elif name == "g4nox":
        src = order_variant(src, 4)
        src = sub(src, "    return n_frames > 0 && (n_frames & 7) == 0 && g.census_elems * 8 <= (size_t)(8u << 20);", "    return false;")
    elif name in ("wtaz", "wtazx"):
        # WTA launch with the FRAME as the fastest-varying block index (wtaz: [row][tile][frame]; wtazx: [row][frame][tile]): the resident blocks then read
        # from all 16 slots at once instead of sweeping one slot after the other -- does the launch's slow placement mode go away when its reads are spread?
        if name == "wtaz":
            src = sub(src, "    const int x0 = blockIdx.x * kWtaTileX, y = blockIdx.y, frame = blockIdx.z;", "    const int x0 = blockIdx.y * kWtaTileX, y = blockIdx.z, frame = blockIdx.x;")
            src = sub(src, "    dim3 grid((g.w + kWtaTileX - 1) / kWtaTileX, g.h, n_frames), block(256);", "    dim3 grid(n_frames, (g.w + kWtaTileX - 1) / kWtaTileX, g.h), block(256);")
        else:
            src = sub(src, "    const int x0 = blockIdx.x * kWtaTileX, y = blockIdx.y, frame = blockIdx.z;", "    const int x0 = blockIdx.x * kWtaTileX, y = blockIdx.z, frame = blockIdx.y;")
            src = sub(src, "    dim3 grid((g.w + kWtaTileX - 1) / kWtaTileX, g.h, n_frames), block(256);", "    dim3 grid((g.w + kWtaTileX - 1) / kWtaTileX, n_frames, g.h), block(256);")
    elif name in ("vd4", "w8", "w2", "ntoff", "prio0", "lb7", "lb5", "hspf4", "hspf16"):
        # the aggregation launch's tuning knobs once more, now that the launch is measured in its FAST placement mode (rounds 2-3 tuned them on workspaces above
        # 8 GiB, i.e. in the slow mode, where the fabric's write stalls bind and the knobs could not show)
        if name == "vd4":
            src = sub(src, "template <int LPP> constexpr int v_depth() { return LPP == 4 ? 4 : 2; }", "template <int LPP> constexpr int v_depth() { return 4; }")
        elif name in ("w8", "w2"):
            src = sub(src, "constexpr int kAggWaves = 4;", "constexpr int kAggWaves = %d;" % (8 if name == "w8" else 2))
        elif name == "ntoff":
            src = sub(src, "        __builtin_nontemporal_store(q, (CART_GLOBAL v4u *)po);", "        *(CART_GLOBAL v4u *)po = q;")
        elif name == "prio0":
            src = sub(src, "        __builtin_amdgcn_s_setprio(3);\n        if constexpr (HS) {", "        if constexpr (HS) {")
        elif name in ("lb7", "lb5"):
            src = sub(src, "(LPP >= 8 && !HS) ? 6 : 4) void aggregate_kernel", "(LPP >= 8 && !HS) ? %s : 4) void aggregate_kernel" % name[2])
        else:
            src = sub(src, "constexpr int HS_PF = 8;", "constexpr int HS_PF = %s;" % name[4:])
    elif name in ("uc", "fg"):
        # the slab groups from hipExtMallocWithFlags: uc = hipDeviceMallocUncached, fg = hipDeviceMallocFinegrained -- does another kind of device memory
        # show the two placement modes?  (engine patch only; the kernels are the product's)
        eng = open(os.path.join(PKG, "csrc", "cart_engine.hip")).read()
        flag = "hipDeviceMallocUncached" if name == "uc" else "hipDeviceMallocFinegrained"
        eng = sub(eng, "        if (dev_alloc(&b, sp.bytes_of(gi))) { slab_pool_free(sp); return -1; }",
                  "        if (hipExtMallocWithFlags(reinterpret_cast<void **>(&b), sp.bytes_of(gi), %s) != hipSuccess) { slab_pool_free(sp); return fail(\"slab alloc\"); }" % flag)
        eng = sub(eng, "                if (dev_alloc(&cand[(size_t)gi], sp.bytes_of(gi))) { ok = false; break; }",
                  "                if (hipExtMallocWithFlags(reinterpret_cast<void **>(&cand[(size_t)gi]), sp.bytes_of(gi), %s) != hipSuccess) { ok = false; break; }" % flag)
    elif name in ("fs_noagg", "fs_nocensus"):
        # timing builds of the fused sweep (results wrong by design): fs_noagg = no "up" recurrence at all (slab loads + sums + WTA only: what the sweep's
        # memory access alone costs), fs_nocensus = recurrence on constant features (no census loads / LDS window staging)
        if name == "fs_noagg":
            src = sub(src, "        agg_step<LPP, false>(st, mm, xr, sel_lo, sel_hi, p1p1, p2p2, nullptr);\n    };", "        (void)xr;\n    };")
        else:
            src = sub(src, "        win_read<LPP>(wbuf, rbase, c.r);\n        uint32_t xr[16];\n        agg_xor(c, xr);\n        load_census_row(max(y - 1, 0));",
                      "        uint32_t xr[16];\n#pragma unroll\n        for (int k = 0; k < 16; ++k) xr[k] = (uint32_t)(k * 0x01010101) ^ (uint32_t)y;")
    elif name == "ns2":
        # the fused sweep with two slab rows in flight for every variant (the product until the 4-path sweeps got four)
        src = sub(src, "template <int NP> constexpr int fused_rows_in_flight() { return NP <= 4 ? 4 : 2; }", "template <int NP> constexpr int fused_rows_in_flight() { return 2; }")
    elif name == "nox":
        src = sub(src, "    return n_frames > 0 && (n_frames & 7) == 0 && g.census_elems * 8 <= (size_t)(8u << 20);", "    return false;")
    elif name in ("privcen", "privcen0"):
        # VERDICT r4 item 3 -- are the census re-fetches of the 1080p aggregation launch (every XCD's L2 pulls every census row: 8x the planes'
        # bytes leave the L2s) served by HBM or by the Infinity Cache?  privcen: every XCD reads its OWN copy of the census planes (copy x lives
        # x * copy_stride elements further on; the patched engine makes the copies after the census kernel), so the eight fetches of a row can no
        # longer meet in the Infinity Cache.  privcen0: the same build, every XCD reads copy 0 (control: same code, same extra copies made).
        # XCC_ID: hwreg 20, bits 3:0.  Only the kernel that launch uses is patched (aggregate_kernel, every instantiation).
        hdr = open(os.path.join(PKG, "csrc", "engine_internal.h")).read()
        hdr = sub(hdr, "    int hsplit;", "    int hsplit;\n    size_t ab_copy_stride;   // A/B build: elements between the per-XCD copies of the census planes")
        extra_hdr = hdr
        sel = "(size_t)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 7)" if name == "privcen" else "(size_t)0"
        src = sub(src, "    const Geometry &g = a.g;\n    // 1-D grid, direction-major: [dir][frame][line group].",
                  "    const ptrdiff_t ab_off = (ptrdiff_t)(" + sel + " * a.ab_copy_stride);\n"
                  "    const Geometry &g = a.g;\n    // 1-D grid, direction-major: [dir][frame][line group].")
        k0 = src.index("void aggregate_kernel(AggArgs a) {"); k1 = src.index("int agg_lines_per_block(int D)")
        body = src[k0:k1]
        assert body.count("a.cen_l + uniform(") == 3 and body.count("a.cen_r + uniform(") == 3, (body.count("a.cen_l + uniform("), body.count("a.cen_r + uniform("))
        body = body.replace("a.cen_l + uniform(", "a.cen_l + uniform(ab_off + ").replace("a.cen_r + uniform(", "a.cen_r + uniform(ab_off + ")
        src = src[:k0] + body + src[k1:]
        eng = open(os.path.join(PKG, "csrc", "cart_engine.hip")).read()
        eng = sub(eng, "    rc |= dev_alloc(&e->cen_l_alloc, n * g.census_elems + 2 * e->cen_slack);\n    rc |= dev_alloc(&e->cen_r_alloc, n * g.census_elems + 2 * e->cen_slack);",
                  "    rc |= dev_alloc(&e->cen_l_alloc, 8 * (n * g.census_elems + 2 * e->cen_slack));\n    rc |= dev_alloc(&e->cen_r_alloc, 8 * (n * g.census_elems + 2 * e->cen_slack));")
        eng = sub(eng, "    if (hipMemset(e->cen_l_alloc, 0, (n * g.census_elems + 2 * e->cen_slack) * 4) != hipSuccess ||\n        hipMemset(e->cen_r_alloc, 0, (n * g.census_elems + 2 * e->cen_slack) * 4) != hipSuccess) {",
                  "    if (hipMemset(e->cen_l_alloc, 0, 8 * (n * g.census_elems + 2 * e->cen_slack) * 4) != hipSuccess ||\n        hipMemset(e->cen_r_alloc, 0, 8 * (n * g.census_elems + 2 * e->cen_slack) * 4) != hipSuccess) {")
        # after the census kernel: copies 1..7 of this launch's planes (untimed by the aggregate stage: the STAGE marker comes after them)
        eng = sub(eng, "        const int launch_plan = plan_for(e, opt, n);\n",
                  "        {\n            const size_t cs = e->slots.size() * g.census_elems + 2 * e->cen_slack;\n            for (int k = 1; k < 8; ++k) {\n"
                  "                (void)hipMemcpyAsync(cl + k * cs, cl, (size_t)n * g.census_elems * 4, hipMemcpyDeviceToDevice, st);\n"
                  "                (void)hipMemcpyAsync(cr + k * cs, cr, (size_t)n * g.census_elems * 4, hipMemcpyDeviceToDevice, st);\n            }\n        }\n"
                  "        const int launch_plan = plan_for(e, opt, n);\n")
        eng = sub(eng, "        a.cen_l = cl; a.cen_r = cr; a.slabs = slabs;\n        launch_aggregate(a, n, st);",
                  "        a.cen_l = cl; a.cen_r = cr; a.slabs = slabs;\n        a.ab_copy_stride = e->slots.size() * g.census_elems + 2 * e->cen_slack;\n        launch_aggregate(a, n, st);")
        eng = sub(eng, "        a.cen_l = cl; a.cen_r = cr; a.slabs = slabs;\n        launch_aggregate(a, n, nullptr);",
                  "        a.cen_l = cl; a.cen_r = cr; a.slabs = slabs;\n        a.ab_copy_stride = e->slots.size() * g.census_elems + 2 * e->cen_slack;\n        launch_aggregate(a, n, nullptr);")
    else:
        raise SystemExit("unknown variant " + name)
    out = os.path.join(PKG, "build", "ab", name)
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "sgm_kernels.hip")
        open(f, "w").write(src)
        inc = []
        if extra_hdr is not None:   # the patched header shadows the product's (the temp dir comes first on the include path)
            open(os.path.join(td, "engine_internal.h"), "w").write(extra_hdr)
            inc = ["-I" + td]
        objs = [os.path.join(td, "sgm_kernels.o")]
        subprocess.run(HIPCC[:8] + inc + HIPCC[8:] + ["-c", f, "-o", objs[0]], check=True)
        reuse = ["cart_engine", "post_kernels", "flow_kernels", "superpixel_kernels"]
        if eng is not None:
            fe = os.path.join(td, "cart_engine.hip")
            open(fe, "w").write(eng)
            objs.append(os.path.join(td, "cart_engine.o"))
            subprocess.run(HIPCC[:8] + inc + HIPCC[8:] + ["-c", fe, "-o", objs[-1]], check=True)
            reuse.remove("cart_engine")
        for o in reuse:
            objs.append(os.path.join(PKG, "build", o + ".o"))
        subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", os.path.join(out, "libcart_engine.so")] + objs, check=True)
    print("built", os.path.join(out, "libcart_engine.so"))


if __name__ == "__main__":
    for n in sys.argv[1:]:
        build(n)
