// sgm_kernels.hip -- gfx950 kernels for the SGM core of the disparity module.
//
// Replaces what cv::cuda::StereoSGM::compute does for the reference
// (src/modules/disparity/disparity.cu:71): census 9x7, per-direction path
// aggregation into u8 cost slabs, winner-takes-all with uniqueness / sub-pixel / right view,
// 3x3 medians, left-right check and range fix.  Written for wave64: one scan line is owned by
// one 16-lane DPP row (D/16 disparities per lane), neighbour exchange and the min over D are
// DPP row_shr/row_shl/row_ror ops, no LDS and no barriers in the recurrence.
#include "engine_internal.h"

namespace cart_amd {

// ------------------------------------------------------------------ DPP helpers
constexpr int DPP_ROW_SHL1 = 0x101;
constexpr int DPP_ROW_SHR1 = 0x111;
constexpr int DPP_ROW_ROR1 = 0x121, DPP_ROW_ROR2 = 0x122, DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128;

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_keep(uint32_t old, uint32_t v) {
    // lanes without a valid source inside their 16-lane row keep `old`
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t row_allmin(uint32_t v) {
    v = min(v, dpp_keep<DPP_ROW_ROR8>(v, v));
    v = min(v, dpp_keep<DPP_ROW_ROR4>(v, v));
    v = min(v, dpp_keep<DPP_ROW_ROR2>(v, v));
    v = min(v, dpp_keep<DPP_ROW_ROR1>(v, v));
    return v;
}

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

template <int N>
__device__ __forceinline__ void load_u32s(const uint32_t *p, uint32_t (&r)[N]) {
    static_assert(N % 4 == 0, "N must be a multiple of 4");
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(p + 4 * i);
        r[4 * i + 0] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
    }
}

__device__ __forceinline__ uint32_t pack4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return a | (b << 8) | (c << 16) | (d << 24);
}

template <int N>
__device__ __forceinline__ void store_u8s(uint8_t *p, const uint32_t (&v)[N]) {
    if constexpr (N == 4) {
        *reinterpret_cast<uint32_t *>(p) = pack4(v[0], v[1], v[2], v[3]);
    } else if constexpr (N == 8) {
        uint2 o = make_uint2(pack4(v[0], v[1], v[2], v[3]), pack4(v[4], v[5], v[6], v[7]));
        *reinterpret_cast<uint2 *>(p) = o;
    } else {
        uint4 o = make_uint4(pack4(v[0], v[1], v[2], v[3]), pack4(v[4], v[5], v[6], v[7]),
                             pack4(v[8], v[9], v[10], v[11]), pack4(v[12], v[13], v[14], v[15]));
        *reinterpret_cast<uint4 *>(p) = o;
    }
}

template <int N>
__device__ __forceinline__ void load_u8s_add(const uint8_t *p, uint32_t (&acc)[N]) {
    if constexpr (N == 4) {
        uint32_t v = *reinterpret_cast<const uint32_t *>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += (v >> (8 * i)) & 0xffu;
    } else if constexpr (N == 8) {
        uint2 v = *reinterpret_cast<const uint2 *>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i] += (v.x >> (8 * i)) & 0xffu; acc[4 + i] += (v.y >> (8 * i)) & 0xffu; }
    } else {
        uint4 v = *reinterpret_cast<const uint4 *>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] += (v.x >> (8 * i)) & 0xffu; acc[4 + i] += (v.y >> (8 * i)) & 0xffu;
            acc[8 + i] += (v.z >> (8 * i)) & 0xffu; acc[12 + i] += (v.w >> (8 * i)) & 0xffu;
        }
    }
}

// ------------------------------------------------------------------ gray + census
// One block = 64x16 output pixels of one image; LDS tile with a 4-column / 3-row halo.
// BGR->gray (oracle S1) is fused into the tile load; the gray plane is written out because the
// left-right check masks on gray_left == 0.  Also resets the packed right-view minima.
constexpr int CT_W = 64, CT_H = 16, CT_LW = CT_W + 8, CT_LH = CT_H + 6;

__global__ __launch_bounds__(256) void census_kernel(ImageBatch left, ImageBatch right, int channels,
                                                     uint8_t *gray_l, uint8_t *gray_r, uint32_t *cen_l,
                                                     uint32_t *cen_r, uint32_t *right_pk, Geometry g) {
    __shared__ uint8_t tile[CT_LH][CT_LW + 8];
    const int frame = blockIdx.z >> 1, side = blockIdx.z & 1;
    const ImageBatch img = side ? right : left;
    const uint8_t *src = img.ptr + (size_t)frame * img.frame_stride;
    uint8_t *gray = (side ? gray_r : gray_l) + (size_t)frame * g.npx;
    uint32_t *cen = (side ? cen_r : cen_l) + (size_t)frame * g.census_elems;
    const int x0 = blockIdx.x * CT_W, y0 = blockIdx.y * CT_H;
    const int tid = threadIdx.y * 64 + threadIdx.x;

    for (int i = tid; i < CT_LH * CT_LW; i += 256) {
        const int ty = i / CT_LW, tx = i - ty * CT_LW;
        const int gx = x0 - 4 + tx, gy = y0 - 3 + ty;
        uint32_t v = 0;
        if (gx >= 0 && gx < g.w && gy >= 0 && gy < g.h) {
            const uint8_t *row = src + (size_t)gy * img.step;
            if (channels == 3) {
                const uint32_t b = row[3 * gx], gg = row[3 * gx + 1], r = row[3 * gx + 2];
                v = (1868u * b + 9617u * gg + 4899u * r + 8192u) >> 14;
            } else {
                v = row[gx];
            }
            if (tx >= 4 && tx < 4 + CT_W && ty >= 3 && ty < 3 + CT_H) gray[(size_t)gy * g.w + gx] = (uint8_t)v;
        }
        tile[ty][tx] = (uint8_t)v;
    }
    __syncthreads();

    const int x = x0 + threadIdx.x;
#pragma unroll
    for (int r = 0; r < CT_H / 4; ++r) {
        const int ly = threadIdx.y + 4 * r, y = y0 + ly;
        if (x >= g.w || y >= g.h) continue;
        uint32_t f = 0;
        if (x >= 4 && x < g.w - 4 && y >= 3 && y < g.h - 3) {
            const int cx = threadIdx.x + 4, cy = ly + 3;
#pragma unroll
            for (int dy = -3; dy < 0; ++dy)
#pragma unroll
                for (int dx = -4; dx <= 4; ++dx)
                    f = (f << 1) | (uint32_t)(tile[cy + dy][cx + dx] > tile[cy - dy][cx - dx]);
#pragma unroll
            for (int dx = -4; dx < 0; ++dx) f = (f << 1) | (uint32_t)(tile[cy][cx + dx] > tile[cy][cx - dx]);
        }
        cen[(size_t)y * g.cpitch + g.cpadl + x] = f;
        if (side == 0) right_pk[(size_t)frame * g.npx + (size_t)y * g.w + x] = 0xffffffffu;
    }
}

void launch_census(const ImageBatch &left, const ImageBatch &right, int channels, int n_frames, uint8_t *gray_l,
                   uint8_t *gray_r, uint32_t *cen_l, uint32_t *cen_r, uint32_t *right_pk, const Geometry &g,
                   hipStream_t s) {
    dim3 grid((g.w + CT_W - 1) / CT_W, (g.h + CT_H - 1) / CT_H, n_frames * 2), block(64, 4);
    hipLaunchKernelGGL(census_kernel, grid, block, 0, s, left, right, channels, gray_l, gray_r, cen_l, cen_r,
                       right_pk, g);
}

// ------------------------------------------------------------------ path aggregation
// All directions of all frames in ONE launch (blockIdx.x -> direction + 16 scan lines,
// blockIdx.y -> frame).  Every direction is a set of independent 1-D lines: vertical and
// diagonal lines are indexed by their (skewed) entry column so no state ever crosses lanes
// other than the +-1 disparity neighbours inside a 16-lane row.
template <int DPL>
__global__ __launch_bounds__(256) void aggregate_kernel(AggArgs a) {
    const Geometry &g = a.g;
    const int frame = blockIdx.y;
    const int b = blockIdx.x;
    int di = 0;
    for (int i = 1; i < a.ndirs; ++i)
        if (b >= a.dirs[i].blk0) di = i;
    const int dx = a.dirs[di].dx, dy = a.dirs[di].dy;
    const int lane16 = threadIdx.x & 15;
    const int line = (b - a.dirs[di].blk0) * kLinesPerBlock + (threadIdx.x >> 4);
    if (line >= a.dirs[di].nlines) return;  // whole 16-lane rows leave together
    const int j = a.dirs[di].jmin + line;

    int xs, ys, t0, t1;
    if (dy != 0) {
        ys = dy > 0 ? 0 : g.h - 1;
        xs = j;
        if (dx > 0) { t0 = max(0, -j); t1 = min(g.h, g.w - j); }
        else if (dx < 0) { t0 = max(0, j - g.w + 1); t1 = min(g.h, j + 1); }
        else { t0 = 0; t1 = g.h; }
    } else {
        ys = j; xs = dx > 0 ? 0 : g.w - 1; t0 = 0; t1 = g.w;
    }
    if (t0 >= t1) return;
    const int x = xs + dx * t0, y = ys + dy * t0;
    const int d0 = lane16 * DPL;

    const uint32_t *pl = a.cen_l + (size_t)frame * g.census_elems + (size_t)y * g.cpitch + g.cpadl + x;
    const uint32_t *pr = a.cen_r + (size_t)frame * g.census_elems + (size_t)y * g.cpitch + g.cpadl + x -
                         g.min_disp - d0 - (DPL - 1);
    const ptrdiff_t cstride = (ptrdiff_t)dy * g.cpitch + dx;
    uint8_t *po = a.slabs + ((size_t)(frame * g.P + a.dirs[di].path) * g.npx + (size_t)y * g.w + x) * g.D + d0;
    const ptrdiff_t ostride = ((ptrdiff_t)dy * g.w + dx) * g.D;

    const uint32_t p1 = (uint32_t)g.p1, p2 = (uint32_t)g.p2;
    constexpr uint32_t INF = 0x7fffu;
    uint32_t dp[DPL];
#pragma unroll
    for (int k = 0; k < DPL; ++k) dp[k] = 0;
    uint32_t last_min = 0;

    uint32_t fl_n = *pl;
    uint32_t r_n[DPL];
    load_u32s<DPL>(pr, r_n);

    for (int t = t0; t < t1; ++t) {
        const uint32_t fl = fl_n;
        uint32_t cost[DPL];
#pragma unroll
        for (int k = 0; k < DPL; ++k) cost[k] = (uint32_t)__builtin_popcount(fl ^ r_n[DPL - 1 - k]);
        pl += cstride; pr += cstride;
        if (t + 1 < t1) {  // prefetch the next pixel's features while this one is reduced
            fl_n = *pl;
            load_u32s<DPL>(pr, r_n);
        }
        // oracle S4: L(d) = C(d) + min(Lp(d), Lp(d-1)+P1, Lp(d+1)+P1, m+P2) - m
        const uint32_t prev_hi = dpp_keep<DPP_ROW_SHR1>(INF, dp[DPL - 1]);  // lane-1's top disparity
        const uint32_t next_lo = dpp_keep<DPP_ROW_SHL1>(INF, dp[0]);        // lane+1's bottom disparity
        const uint32_t mp2 = last_min + p2;
        uint32_t nd[DPL];
        uint32_t lmin = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const uint32_t lo = k == 0 ? prev_hi : dp[k - 1];
            const uint32_t hi = k == DPL - 1 ? next_lo : dp[k + 1];
            uint32_t tt = min(lo, hi) + p1;
            tt = min(tt, dp[k]);
            tt = min(tt, mp2);
            nd[k] = tt - last_min + cost[k];
            lmin = min(lmin, nd[k]);
        }
#pragma unroll
        for (int k = 0; k < DPL; ++k) dp[k] = nd[k];
        last_min = row_allmin(lmin);
        store_u8s<DPL>(po, nd);
        po += ostride;
    }
}

void launch_aggregate(const AggArgs &a, int n_frames, hipStream_t s) {
    dim3 grid(a.blocks_per_frame, n_frames), block(256);
    switch (a.g.D) {
        case 64: hipLaunchKernelGGL(aggregate_kernel<4>, grid, block, 0, s, a); break;
        case 128: hipLaunchKernelGGL(aggregate_kernel<8>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(aggregate_kernel<16>, grid, block, 0, s, a); break;
    }
}

// ------------------------------------------------------------------ winner takes all
// Block = 64 pixels of one row, 16-lane row per pixel (4 passes of 16 pixels).  The summed
// costs of the tile live in LDS as u16 [64][D] so that (a) the sub-pixel neighbours and
// (b) the right-view diagonal minima S(p+d, d) come from LDS; per-tile right minima are
// merged across tiles with one packed atomicMin per right pixel and tile.
template <int DPL>
__global__ __launch_bounds__(256) void wta_kernel(const uint8_t *slabs, uint16_t *wta_l, uint32_t *right_pk,
                                                  Geometry g, float uniq) {
    extern __shared__ __attribute__((aligned(16))) uint16_t s_lds[];  // [kWtaTileX][D]
    constexpr int D = DPL * 16;
    const int x0 = blockIdx.x * kWtaTileX, y = blockIdx.y, frame = blockIdx.z;
    const int grp = threadIdx.x >> 4, lane16 = threadIdx.x & 15, d0 = lane16 * DPL;
    const int row_in_wave = (threadIdx.x & 63) >> 4;

    uint32_t pk_res[4];
    bool uniq_res[4];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int xl = pass * 16 + grp;
        const int xc = min(x0 + xl, g.w - 1);
        uint32_t S[DPL];
#pragma unroll
        for (int k = 0; k < DPL; ++k) S[k] = 0;
        const uint8_t *p = slabs + ((size_t)frame * g.P * g.npx + (size_t)y * g.w + xc) * D + d0;
        for (int r = 0; r < g.P; ++r) load_u8s_add<DPL>(p + (size_t)r * g.slab_bytes, S);
        uint16_t *dst = s_lds + xl * D + d0;
#pragma unroll
        for (int k = 0; k < DPL; k += 2)
            *reinterpret_cast<uint32_t *>(dst + k) = S[k] | (S[k + 1] << 16);
        uint32_t pk = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < DPL; ++k) pk = min(pk, (S[k] << 16) | (uint32_t)(d0 + k));
        pk = row_allmin(pk);
        const uint32_t bc = pk >> 16;
        const int bd = (int)(pk & 0xffffu);
        const float bcf = (float)bc;
        bool fail = false;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const float lhs = (float)S[k] * uniq;
            const bool u1 = lhs >= bcf;
            const bool u2 = abs(d0 + k - bd) <= 1;
            fail |= !(u1 || u2);
        }
        const unsigned long long bal = __ballot(fail);
        uniq_res[pass] = ((bal >> (16 * row_in_wave)) & 0xffffull) == 0ull;
        pk_res[pass] = pk;
    }
    __syncthreads();

    if (lane16 == 0) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int xl = pass * 16 + grp, x = x0 + xl;
            if (x >= g.w) continue;
            uint32_t out = kWtaInvalid;
            if (uniq_res[pass]) {
                const int bd = (int)(pk_res[pass] & 0xffffu), bc = (int)(pk_res[pass] >> 16);
                int subp = bd * 16;
                if (bd > 0 && bd < D - 1) {
                    const int l = s_lds[xl * D + bd - 1], r = s_lds[xl * D + bd + 1];
                    const int num = l - r, den = l - 2 * bc + r;
                    if (den != 0) subp += (num * 16 + den) / (2 * den);
                }
                out = (uint32_t)subp & 0xffffu;
            }
            wta_l[(size_t)frame * g.npx + (size_t)y * g.w + x] = (uint16_t)out;
        }
    }

    // right view (oracle S6): partial minima over this tile's pixels, merged by atomicMin
    for (int pi = threadIdx.x; pi < kWtaTileX + D - 1; pi += 256) {
        const int p = x0 - (D - 1) + pi;
        if (p < 0 || p >= g.w) continue;
        const int xa = max(x0, p), xb = min(min(x0 + kWtaTileX, p + D), g.w);
        uint32_t best = 0xffffffffu;
        for (int x = xa; x < xb; ++x) {
            const int d = x - p;
            best = min(best, ((uint32_t)s_lds[(x - x0) * D + d] << 16) | (uint32_t)d);
        }
        if (xa < xb) atomicMin(&right_pk[(size_t)frame * g.npx + (size_t)y * g.w + p], best);
    }
}

void launch_wta(const uint8_t *slabs, uint16_t *wta_l, uint32_t *right_pk, const Geometry &g, float uniq,
                int n_frames, hipStream_t s) {
    dim3 grid((g.w + kWtaTileX - 1) / kWtaTileX, g.h, n_frames), block(256);
    const size_t lds = (size_t)kWtaTileX * g.D * sizeof(uint16_t);
    switch (g.D) {
        case 64: hipLaunchKernelGGL(wta_kernel<4>, grid, block, lds, s, slabs, wta_l, right_pk, g, uniq); break;
        case 128: hipLaunchKernelGGL(wta_kernel<8>, grid, block, lds, s, slabs, wta_l, right_pk, g, uniq); break;
        default: hipLaunchKernelGGL(wta_kernel<16>, grid, block, lds, s, slabs, wta_l, right_pk, g, uniq); break;
    }
}

// ------------------------------------------------------------------ median x2 + LR check + range fix
#define CART_SORT2(a, b) { const uint32_t _t = min(a, b); b = max(a, b); a = _t; }
__device__ __forceinline__ uint32_t median9(uint32_t (&p)[9]) {
    CART_SORT2(p[1], p[2]) CART_SORT2(p[4], p[5]) CART_SORT2(p[7], p[8])
    CART_SORT2(p[0], p[1]) CART_SORT2(p[3], p[4]) CART_SORT2(p[6], p[7])
    CART_SORT2(p[1], p[2]) CART_SORT2(p[4], p[5]) CART_SORT2(p[7], p[8])
    CART_SORT2(p[0], p[3]) CART_SORT2(p[5], p[8]) CART_SORT2(p[4], p[7])
    CART_SORT2(p[3], p[6]) CART_SORT2(p[1], p[4]) CART_SORT2(p[2], p[5])
    CART_SORT2(p[4], p[7]) CART_SORT2(p[4], p[2]) CART_SORT2(p[6], p[4])
    CART_SORT2(p[4], p[2])
    return p[4];
}

template <typename T>
__device__ __forceinline__ uint32_t median_at(const T *img, int x, int y, int w, int h) {
    if (x < 1 || x >= w - 1 || y < 1 || y >= h - 1) return (uint32_t)img[(size_t)y * w + x] & 0xffffu;  // S7 border
    uint32_t v[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = (uint32_t)img[(size_t)(y - 1 + i / 3) * w + (x - 1 + i % 3)] & 0xffffu;
    return median9(v);
}

__global__ __launch_bounds__(256) void post_kernel(const uint16_t *wta_l, const uint32_t *right_pk,
                                                   const uint8_t *gray_l, int16_t *out, size_t out_step,
                                                   size_t out_frame_stride, Geometry g) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, frame = blockIdx.z;
    if (x >= g.w || y >= g.h) return;
    const uint16_t *wl = wta_l + (size_t)frame * g.npx;
    const uint32_t *rp = right_pk + (size_t)frame * g.npx;
    const uint32_t ml = median_at(wl, x, y, g.w, g.h);
    bool invalid = gray_l[(size_t)frame * g.npx + (size_t)y * g.w + x] == 0 || ml == kWtaInvalid;
    if (!invalid) {
        const int d = (int)(ml >> 4);
        const int k = x - d;
        if (k >= 0 && k < g.w) {
            const int mr = (int)median_at(rp, k, y, g.w, g.h);
            if (abs(mr - d) > 1) invalid = true;
        }
    }
    const int v = invalid ? (g.min_disp - 1) * 16 : (int)ml + g.min_disp * 16;
    int16_t *orow = reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(out) + (size_t)frame * out_frame_stride +
                                                (size_t)y * out_step);
    orow[x] = (int16_t)v;
}

void launch_post(const uint16_t *wta_l, const uint32_t *right_pk, const uint8_t *gray_l, int16_t *out, size_t out_step,
                 size_t out_frame_stride, const Geometry &g, int n_frames, hipStream_t s) {
    dim3 grid((g.w + 63) / 64, (g.h + 3) / 4, n_frames), block(64, 4);
    hipLaunchKernelGGL(post_kernel, grid, block, 0, s, wta_l, right_pk, gray_l, out, out_step, out_frame_stride, g);
}

}  // namespace cart_amd
