/*
 * cart_oracle.c -- CPU restatement (see cart_oracle.h for the spec S1..S12 and
 * the "TEST INFRASTRUCTURE ONLY / PARITY UNPINNED" notice).
 *
 * Plain C99 (+ optional OpenMP), no dependencies.  Every function cites the
 * reference file:line it follows; the SGM core follows the algorithm of the
 * third-party cv::cuda::StereoSGM the reference calls (disparity.cu:71).
 */
#include "cart_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define INVALID16 ((int16_t)CART_ORACLE_INVALID)

/* ------------------------------------------------------------------ S1 gray */
/* reference call site: src/modules/disparity/disparity.cu:66-67 (cv::cuda::cvtColor) */
void cart_oracle_bgr2gray(const uint8_t *bgr, size_t src_step, int w, int h, uint8_t *gray) {
    for (int y = 0; y < h; y++) {
        const uint8_t *row = bgr + (size_t)y * src_step;
        for (int x = 0; x < w; x++) {
            unsigned b = row[3 * x + 0], g = row[3 * x + 1], r = row[3 * x + 2];
            gray[(size_t)y * w + x] = (uint8_t)((1868u * b + 9617u * g + 4899u * r + 8192u) >> 14);
        }
    }
}

/* ---------------------------------------------------------------- S2 census */
void cart_oracle_census9x7(const uint8_t *gray, int w, int h, uint32_t *census) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            uint32_t f = 0;
            if (x >= 4 && x < w - 4 && y >= 3 && y < h - 3) {
                for (int dy = -3; dy < 0; dy++)
                    for (int dx = -4; dx <= 4; dx++) {
                        uint8_t a = gray[(size_t)(y + dy) * w + (x + dx)];
                        uint8_t b = gray[(size_t)(y - dy) * w + (x - dx)];
                        f = (f << 1) | (uint32_t)(a > b);
                    }
                for (int dx = -4; dx < 0; dx++) {
                    uint8_t a = gray[(size_t)y * w + (x + dx)];
                    uint8_t b = gray[(size_t)y * w + (x - dx)];
                    f = (f << 1) | (uint32_t)(a > b);
                }
            }
            census[(size_t)y * w + x] = f;
        }
    }
}

/* ------------------------------------------------------------ S3 + S4 paths */
static inline int popc32(uint32_t v) { return __builtin_popcount(v); }

/* one step of the recurrence: prev[D] (state of previous pixel on the path, or
 * all-zero at a path start), cost[D] -> cur[D]; returns nothing. */
static inline void dp_step(const int *prev, const int *cost, int D, int p1, int p2, int *cur) {
    int m = prev[0];
    for (int d = 1; d < D; d++)
        if (prev[d] < m) m = prev[d];
    for (int d = 0; d < D; d++) {
        int best = prev[d];
        if (d > 0 && prev[d - 1] + p1 < best) best = prev[d - 1] + p1;
        if (d + 1 < D && prev[d + 1] + p1 < best) best = prev[d + 1] + p1;
        if (m + p2 < best) best = m + p2;
        cur[d] = cost[d] + best - m;
    }
}

static void walk_line(const uint32_t *cen_l, const uint32_t *cen_r, int w, int h, int D, int min_disp,
                      int p1, int p2, int dx, int dy, int x, int y, uint8_t *L, int *prev, int *cur, int *cost) {
    memset(prev, 0, sizeof(int) * (size_t)D);
    while (x >= 0 && x < w && y >= 0 && y < h) {
        uint32_t fl = cen_l[(size_t)y * w + x];
        for (int d = 0; d < D; d++) {
            int xr = x - d - min_disp;
            uint32_t fr = (xr >= 0 && xr < w) ? cen_r[(size_t)y * w + xr] : 0u;
            cost[d] = popc32(fl ^ fr);
        }
        dp_step(prev, cost, D, p1, p2, cur);
        uint8_t *out = L + ((size_t)y * w + x) * D;
        for (int d = 0; d < D; d++) out[d] = (uint8_t)cur[d];
        int *t = prev; prev = cur; cur = t;
        x += dx; y += dy;
    }
}

void cart_oracle_aggregate_path(const uint32_t *cen_l, const uint32_t *cen_r, int w, int h, int D,
                                int min_disp, int p1, int p2, int dx, int dy, uint8_t *L) {
    /* enumerate every pixel whose predecessor (x-dx,y-dy) is outside the image: path starts */
    int nstart = 0;
    int *sx = (int *)malloc(sizeof(int) * (size_t)(w + h) * 2);
    int *sy = (int *)malloc(sizeof(int) * (size_t)(w + h) * 2);
    if (dy != 0) {
        int y0 = dy > 0 ? 0 : h - 1;
        for (int x = 0; x < w; x++) { sx[nstart] = x; sy[nstart] = y0; nstart++; }
    }
    if (dx != 0) {
        int x0 = dx > 0 ? 0 : w - 1;
        for (int y = 0; y < h; y++) {
            if (dy != 0 && y == (dy > 0 ? 0 : h - 1)) continue; /* corner already listed */
            sx[nstart] = x0; sy[nstart] = y; nstart++;
        }
    }
#pragma omp parallel
    {
        int *buf = (int *)malloc(sizeof(int) * (size_t)D * 3);
#pragma omp for schedule(dynamic, 8)
        for (int i = 0; i < nstart; i++)
            walk_line(cen_l, cen_r, w, h, D, min_disp, p1, p2, dx, dy, sx[i], sy[i], L, buf, buf + D, buf + 2 * D);
        free(buf);
    }
    free(sx); free(sy);
}

void cart_oracle_path_dir(int index, int *dx, int *dy) {
    /* 0..3 = MODE_HH4 {down, up, right, left}; 4..7 = diagonals added by MODE_HH */
    static const int DX[8] = {0, 0, 1, -1, 1, -1, -1, 1};
    static const int DY[8] = {1, -1, 0, 0, 1, 1, -1, -1};
    *dx = DX[index & 7]; *dy = DY[index & 7];
}

/* ----------------------------------------------------------- S5 + S6 WTA */
void cart_oracle_wta(const uint16_t *S, int w, int h, int D, int uniqueness_ratio,
                     uint16_t *left, uint16_t *right) {
    cart_oracle_wta_ex(S, w, h, D, uniqueness_ratio, left, right, 0);
}
void cart_oracle_wta_ex(const uint16_t *S, int w, int h, int D, int uniqueness_ratio,
                        uint16_t *left, uint16_t *right, int variants) {
    const int top2 = (variants & CART_ORACLE_VARIANT_S5_TOP2) != 0;
    const float u = (float)(100 - uniqueness_ratio) / 100.0f;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        const uint16_t *row = S + (size_t)y * w * D;
        for (int x = 0; x < w; x++) {
            const uint16_t *s = row + (size_t)x * D;
            uint32_t best = 0xffffffffu;
            for (int d = 0; d < D; d++) {
                uint32_t pk = ((uint32_t)s[d] << 16) | (uint32_t)d;
                if (pk < best) best = pk;
            }
            int bd = (int)(best & 0xffffu);
            uint32_t bc = best >> 16;
            int uniq = 1;
            if (top2) {
                /* S5 variant, the wording of SURVEY.md 8a-4(4): ONE pass over d that tracks the best and the second-best
                 * (cost, d), both replaced on a strictly smaller cost only; only the second-best is tested */
                uint32_t c1 = 0xffffffffu, c2 = 0xffffffffu;
                int d1 = -1, d2 = -1;
                for (int d = 0; d < D; d++) {
                    if (s[d] < c1) { c2 = c1; d2 = d1; c1 = s[d]; d1 = d; }
                    else if (s[d] < c2) { c2 = s[d]; d2 = d; }
                }
                if (d2 >= 0) {
                    int dd = d2 - d1; if (dd < 0) dd = -dd;
                    uniq = ((float)c2 * u >= (float)c1) || dd <= 1;
                }
            } else
            for (int d = 0; d < D; d++) {
                float lhs = (float)s[d] * u; /* one rounded f32 multiply (-ffp-contract=off) */
                int u1 = lhs >= (float)bc;
                int dd = d - bd; if (dd < 0) dd = -dd;
                if (!(u1 || dd <= 1)) { uniq = 0; break; }
            }
            uint16_t out = CART_ORACLE_WTA_INVALID;
            if (uniq) {
                int subp = bd * 16;
                if (bd > 0 && bd < D - 1) {
                    int l = s[bd - 1], r = s[bd + 1];
                    int num = l - r, den = l - 2 * (int)bc + r;
                    if (den != 0) subp += (num * 16 + den) / (2 * den);
                }
                out = (uint16_t)subp;
            }
            left[(size_t)y * w + x] = out;
        }
        for (int p = 0; p < w; p++) {
            uint32_t best = 0xffffffffu;
            for (int d = 0; d < D && p + d < w; d++) {
                uint32_t pk = ((uint32_t)row[(size_t)(p + d) * D + d] << 16) | (uint32_t)d;
                if (pk < best) best = pk;
            }
            right[(size_t)y * w + p] = (uint16_t)(best & 0xffffu);
        }
    }
}

/* ------------------------------------------------------------- S7 median */
static int cmp_u16(const void *a, const void *b) {
    uint16_t x = *(const uint16_t *)a, y = *(const uint16_t *)b;
    return (x > y) - (x < y);
}

void cart_oracle_median3x3_u16_ex(const uint16_t *src, int w, int h, uint16_t *dst, int variants) {
    const int replicate = (variants & CART_ORACLE_VARIANT_S7_REPLICATE_BORDER) != 0;
#pragma omp parallel for schedule(static)   /* rows are independent: same values in any order */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            if (!replicate && (x < 1 || x >= w - 1 || y < 1 || y >= h - 1)) {
                dst[(size_t)y * w + x] = src[(size_t)y * w + x];
                continue;
            }
            uint16_t buf[9];
            for (int i = 0; i < 9; i++) {
                int yy = y - 1 + i / 3, xx = x - 1 + i % 3;   /* S7 variant: the window reads the replicated border */
                yy = yy < 0 ? 0 : yy >= h ? h - 1 : yy;
                xx = xx < 0 ? 0 : xx >= w ? w - 1 : xx;
                buf[i] = src[(size_t)yy * w + xx];
            }
            qsort(buf, 9, sizeof(uint16_t), cmp_u16);
            dst[(size_t)y * w + x] = buf[4];
        }
}
void cart_oracle_median3x3_u16(const uint16_t *src, int w, int h, uint16_t *dst) { cart_oracle_median3x3_u16_ex(src, w, h, dst, 0); }

/* -------------------------------------------------------- S8 + S9 LR/range */
void cart_oracle_lr_check_range(const uint16_t *left_med, const uint16_t *right_med, const uint8_t *gray_left,
                                int w, int h, int min_disp, int16_t *out) {
    cart_oracle_lr_check_range_ex(left_med, right_med, gray_left, w, h, min_disp, out, 0);
}
void cart_oracle_lr_check_range_ex(const uint16_t *left_med, const uint16_t *right_med, const uint8_t *gray_left,
                                   int w, int h, int min_disp, int16_t *out, int variants) {
    const int zero_invalid = (variants & CART_ORACLE_VARIANT_S8_ZERO_INVALID) != 0;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            uint16_t org = left_med[i];
            int invalid = 0;
            if (gray_left[i] == 0 || org == CART_ORACLE_WTA_INVALID) invalid = 1;
            else if (zero_invalid && (org >> 4) == 0) invalid = 1;   /* S8 variant: `d <= 0` on the integer disparity */
            else {
                int d = (int)org >> 4;
                int k = x - d;
                if (k >= 0 && k < w) {
                    int diff = (int)right_med[(size_t)y * w + k] - d;
                    if (diff < 0) diff = -diff;
                    if (diff > 1) invalid = 1;
                }
            }
            out[i] = invalid ? (int16_t)((min_disp - 1) * 16) : (int16_t)((int)org + min_disp * 16);
        }
}

/* --------------------------------------------------------------- full SGM */
int cart_oracle_sgm(const cart_oracle_sgm_params *p, const uint8_t *gray_l, const uint8_t *gray_r,
                    int16_t *disp, uint16_t *S_out) {
    return cart_oracle_sgm_ex(p, gray_l, gray_r, disp, S_out, 0);
}
int cart_oracle_sgm_ex(const cart_oracle_sgm_params *p, const uint8_t *gray_l, const uint8_t *gray_r,
                       int16_t *disp, uint16_t *S_out, int variants) {
    const int w = p->width, h = p->height, D = p->num_disparities;
    if (!(D == 64 || D == 128 || D == 256) || !(p->paths == 4 || p->paths == 8) || w <= 0 || h <= 0) return -1;
    if (p->p2 + 31 > 255) return -1;
    size_t npx = (size_t)w * h;
    uint32_t *cl = (uint32_t *)malloc(npx * 4), *cr = (uint32_t *)malloc(npx * 4);
    uint8_t *L = (uint8_t *)malloc(npx * D);
    uint16_t *S = S_out ? S_out : (uint16_t *)malloc(npx * D * 2);
    uint16_t *wl = (uint16_t *)malloc(npx * 2), *wr = (uint16_t *)malloc(npx * 2);
    uint16_t *ml = (uint16_t *)malloc(npx * 2), *mr = (uint16_t *)malloc(npx * 2);
    if (!cl || !cr || !L || !S || !wl || !wr || !ml || !mr) return -1;
    cart_oracle_census9x7(gray_l, w, h, cl);
    cart_oracle_census9x7(gray_r, w, h, cr);
    memset(S, 0, npx * D * 2);
    for (int r = 0; r < p->paths; r++) {
        int dx, dy;
        cart_oracle_path_dir(r, &dx, &dy);
        cart_oracle_aggregate_path(cl, cr, w, h, D, p->min_disparity, p->p1, p->p2, dx, dy, L);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)(npx * D); i++) S[i] = (uint16_t)(S[i] + L[i]);
    }
    cart_oracle_wta_ex(S, w, h, D, p->uniqueness_ratio, wl, wr, variants);
    cart_oracle_median3x3_u16_ex(wl, w, h, ml, variants);
    cart_oracle_median3x3_u16_ex(wr, w, h, mr, variants);
    cart_oracle_lr_check_range_ex(ml, mr, gray_l, w, h, p->min_disparity, disp, variants);
    free(cl); free(cr); free(L); if (!S_out) free(S);
    free(wl); free(wr); free(ml); free(mr);
    return 0;
}

/* ------------------------------------------------------- a-5 interpolate */
/* src/modules/disparity/interpolation.cu:17-82: window (2r-1)^2, value counted iff
 * min < v < max (:36-39), result sum/count iff count > r*r+1 (:33,:43) else INVALID. */
void cart_oracle_interpolate(const int16_t *in, int w, int h, int radius, int iterations,
                             int min_disp16, int max_disp, int16_t *out) {
    size_t npx = (size_t)w * h;
    int16_t *a = (int16_t *)malloc(npx * 2), *b = (int16_t *)malloc(npx * 2);
    memcpy(a, in, npx * 2);
    const unsigned min_count = (unsigned)(radius * radius + 1);
    for (int it = 0; it < iterations; it++) {
#pragma omp parallel for schedule(static)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int sum = 0, count = 0;
                for (int l = -radius + 1; l < radius; l++) {
                    int yy = y + l;
                    if (yy < 0 || yy >= h) continue;
                    for (int k = -radius + 1; k < radius; k++) {
                        int xx = x + k;
                        if (xx < 0 || xx >= w) continue;
                        int v = a[(size_t)yy * w + xx];
                        if (v > min_disp16 && v < max_disp) { sum += v; count++; }
                    }
                }
                b[(size_t)y * w + x] = ((unsigned)count > min_count) ? (int16_t)(sum / count) : INVALID16;
            }
        int16_t *t = a; a = b; b = t;
    }
    memcpy(out, a, npx * 2);
    free(a); free(b);
}

/* ------------------------------------------- a-7 directional derivative */
/* src/modules/disparity/derivative.cu:55-85 (DERIV_OFFSET 2), :99-116 merge */
void cart_oracle_directional_derivative(const int16_t *disp, int w, int h, int16_t *out, int32_t *hist) {
    memset(hist, 0, sizeof(int32_t) * 512);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int16_t dv = INVALID16, dh = INVALID16;
            if (y - 2 >= 0 && y + 2 < h) {
                int16_t a = disp[(size_t)(y + 2) * w + x], b = disp[(size_t)(y - 2) * w + x];
                if (a != INVALID16 && b != INVALID16) dv = (int16_t)(a - b);
                if (a != INVALID16 && b != INVALID16 && dv >= -128 && dv <= 127) hist[2 * (dv + 128)]++;
            }
            if (x - 2 >= 0 && x + 2 < w) {
                int16_t a = disp[(size_t)y * w + x + 2], b = disp[(size_t)y * w + x - 2];
                if (a != INVALID16 && b != INVALID16) dh = (int16_t)(a - b);
                if (a != INVALID16 && b != INVALID16 && dh >= -128 && dh <= 127) hist[2 * (dh + 128) + 1]++;
            }
            out[((size_t)y * w + x) * 2 + 0] = dv;
            out[((size_t)y * w + x) * 2 + 1] = dh;
        }
}

/* ------------------------------------------------ a-8 plane derivative */
/* src/modules/planeseg/planeseg.cu:60-104 (5-tap vertical mean, s16 sum),
 * :109-129 (1-px vertical difference + histogram), :144-158 (cumulative add). */
void cart_oracle_plane_derivative(const int16_t *disp, int w, int h, int16_t *out, int32_t *hist256) {
    size_t npx = (size_t)w * h;
    int16_t *lp = (int16_t *)malloc(npx * 2);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int16_t sum = 0; /* derivative_t accumulator: wraps like the reference's (:62) */
            int count = 0;
            for (int k = -2; k <= 2; k++) {
                int yy = y + k;
                if (yy < 0 || yy >= h) continue;
                int16_t v = disp[(size_t)yy * w + x];
                if (v != INVALID16) { sum = (int16_t)(sum + v); count++; }
            }
            lp[(size_t)y * w + x] = count == 0 ? INVALID16 : (int16_t)((int)sum / count);
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int16_t o = INVALID16;
            if (y - 1 >= 0 && y + 1 < h) {
                int16_t a = lp[(size_t)(y + 1) * w + x], c = lp[(size_t)y * w + x], b = lp[(size_t)(y - 1) * w + x];
                if (a != INVALID16 && b != INVALID16 && c != INVALID16) {
                    o = (int16_t)(a - b);
                    if (o >= -128 && o <= 127) hist256[o + 128]++;
                    /* a wrapped difference equal to -32768 is indistinguishable from INVALID, as in the reference */
                }
            }
            out[(size_t)y * w + x] = o;
        }
    free(lp);
}

/* ------------------------------------------------------- a-9 findPeaks */
/* src/utils/peaks.cpp:12-72 */
typedef struct { int idx; int32_t val; } idxval;
static int cmp_desc(const void *a, const void *b) {
    const idxval *x = (const idxval *)a, *y = (const idxval *)b;
    if (x->val != y->val) return (x->val < y->val) - (x->val > y->val);
    return (x->idx > y->idx) - (x->idx < y->idx);
}

int cart_oracle_find_peaks(const int32_t *data, int n, int *born, int *died, int *left, int *right) {
    int *idxtopeak = (int *)malloc(sizeof(int) * (size_t)n);
    idxval *order = (idxval *)malloc(sizeof(idxval) * (size_t)n);
    for (int i = 0; i < n; i++) { idxtopeak[i] = -1; order[i].idx = i; order[i].val = data[i]; }
    qsort(order, (size_t)n, sizeof(idxval), cmp_desc);
    int np = 0;
    for (int o = 0; o < n; o++) {
        int idx = order[o].idx;
        int lftdone = idx > 0 && idxtopeak[idx - 1] != -1;
        int rgtdone = idx < n - 1 && idxtopeak[idx + 1] != -1;
        int il = lftdone ? idxtopeak[idx - 1] : -1;
        int ir = rgtdone ? idxtopeak[idx + 1] : -1;
        if (!lftdone && !rgtdone) {
            born[np] = left[np] = right[np] = idx; died[np] = -1;
            idxtopeak[idx] = np++;
        } else if (lftdone && !rgtdone) {
            right[il] += 1; idxtopeak[idx] = il;
        } else if (!lftdone && rgtdone) {
            left[ir] -= 1; idxtopeak[idx] = ir;
        } else {
            if (data[born[il]] > data[born[ir]]) {
                died[ir] = idx; right[il] = right[ir];
                idxtopeak[right[il]] = idxtopeak[idx] = il;
            } else {
                died[il] = idx; left[ir] = left[il];
                idxtopeak[left[ir]] = idxtopeak[idx] = ir;
            }
        }
    }
    /* stable insertion sort by descending persistence (died==-1 -> INT_MAX), peaks.cpp:4-10,70 */
    for (int i = 1; i < np; i++) {
        int b = born[i], d = died[i], l = left[i], r = right[i];
        long pi = d == -1 ? (long)INT_MAX : (long)data[b] - (long)data[d];
        int j = i - 1;
        while (j >= 0) {
            long pj = died[j] == -1 ? (long)INT_MAX : (long)data[born[j]] - (long)data[died[j]];
            if (pj >= pi) break;
            born[j + 1] = born[j]; died[j + 1] = died[j]; left[j + 1] = left[j]; right[j + 1] = right[j];
            j--;
        }
        born[j + 1] = b; died[j + 1] = d; left[j + 1] = l; right[j + 1] = r;
    }
    free(idxtopeak); free(order);
    return np;
}

/* src/modules/planeseg/planeseg.cu:405-458 */
int cart_oracle_histogram_peak_params(const int32_t *hist, cart_oracle_plane_params *params) {
    int born[256], died[256], left[256], right[256];
    int np = cart_oracle_find_peaks(hist, 256, born, died, left, right);
    if (np < 2) return 0;
    int p0 = born[0], p1 = born[1];
    if (abs(p0 - 128) > abs(p1 - 128)) { int t = p0; p0 = p1; p1 = t; }
    /* the reference assigns the centres before its early-outs (:418-419) */
    params->vertical_center = p0 - 128;
    params->horizontal_center = p1 - 128;
    int lo = p0 < p1 ? p0 : p1, hi = p0 < p1 ? p1 : p0;
    int min_index = lo;
    for (int i = lo; i < hi; i++)
        if (hist[i] < hist[min_index]) min_index = i;
    int vdist = abs(min_index - p0), hdist = abs(min_index - p1);
    if (vdist == 0 || hdist == 0) return 0;
    int vder = (hist[p0] - hist[min_index]) / vdist;
    int hder = (hist[p1] - hist[min_index]) / hdist;
    if (vder == 0 || hder == 0) return 0;
    int vwidth = hist[p0] / vder, hwidth = hist[p1] / hder;
    params->vertical_min = p0 - vwidth - 128;
    params->vertical_max = min_index - 127;
    params->horizontal_min = min_index - 127;
    params->horizontal_max = p1 + hwidth - 127;
    return 1;
}

/* ---------------------------------------------------------- a-10 classify */
/* src/modules/planeseg/planeseg.cu:188-197 */
void cart_oracle_classify(const int16_t *deriv, int w, int h, const cart_oracle_plane_params *p, uint8_t *planes) {
    size_t npx = (size_t)w * h;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < npx; i++) {
        int d = deriv[i];
        uint8_t plane = 2;
        if (d != CART_ORACLE_INVALID && d >= p->horizontal_min && d < p->horizontal_max) plane = 0;
        else if (d != CART_ORACLE_INVALID && d >= p->vertical_min && d < p->vertical_max) plane = 1;
        planes[i] = plane;
    }
}

/* ------------------------------------------------- a-10 temporal voting */
void cart_oracle_temporal_vote(const uint8_t *planes, int w, int h, int n_prev, const uint8_t *const *prev_planes,
                               const int16_t *const *flows, uint8_t *smoothed) {
    for (int py = 0; py < h; py++)
        for (int px = 0; px < w; px++) {
            int votes[3] = {0, 0, 0};
            votes[planes[(size_t)py * w + px]]++;
            int x = px, y = py;
            for (int k = 0; k < n_prev; k++) {
                int16_t fx = flows[k][((size_t)py * w + px) * 2 + 0], fy = flows[k][((size_t)py * w + px) * 2 + 1];
                fx = (int16_t)(fx >> 5); fy = (int16_t)(fy >> 5);
                x -= fx; y -= fy;
                if (x < 0 || y < 0 || x >= w || y >= h) continue;
                votes[prev_planes[k][(size_t)y * w + x]]++;
            }
            int best = votes[0] > votes[1] ? 0 : 1;
            if (votes[best] == 0) best = 2;
            smoothed[(size_t)py * w + px] = (uint8_t)best;
        }
}

/* ------------------------------------------------------------ 8f-2 depth */
/* src/modules/depth.cpp:18-19 (convertTo 1/16, reprojectImageTo3D); operation order of OpenCV's CUDA kernel */
void cart_oracle_reproject_depth(const int16_t *disp, int w, int h, const float Q[16], float *xyz) {
    for (int y = 0; y < h; y++) {
        const float qx = Q[1] * y + Q[3], qy = Q[5] * y + Q[7], qz = Q[9] * y + Q[11], qw = Q[13] * y + Q[15];
        for (int x = 0; x < w; x++) {
            const float d = (float)disp[(size_t)y * w + x] * 0.0625f;
            const float iW = 1.f / (qw + Q[12] * x + Q[14] * d);
            float *o = xyz + ((size_t)y * w + x) * 3;
            o[0] = (qx + Q[0] * x + Q[2] * d) * iW;
            o[1] = (qy + Q[4] * x + Q[6] * d) * iW;
            o[2] = (qz + Q[8] * x + Q[10] * d) * iW;
        }
    }
}

/* --------------------------------------------------------------- a-11 CCL */
/* ------------------------------------------------------------ S16 resize */
void cart_oracle_resize_linear(const uint8_t *src, int sw, int sh, int channels, uint8_t *dst, int dw, int dh) {
    const float fx = (float)((double)sw / (double)dw), fy = (float)((double)sh / (double)dh);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            const float src_x = (float)x * fx, src_y = (float)y * fy;
            const int x1 = (int)floorf(src_x), y1 = (int)floorf(src_y), x2 = x1 + 1, y2 = y1 + 1;
            const int x2r = x2 < sw - 1 ? x2 : sw - 1, y2r = y2 < sh - 1 ? y2 : sh - 1;
            const float wx1 = (float)x2 - src_x, wx2 = src_x - (float)x1, wy1 = (float)y2 - src_y, wy2 = src_y - (float)y1;
            for (int c = 0; c < channels; c++) {
                float out = 0.f;
                out = out + (float)src[((size_t)y1 * sw + x1) * channels + c] * (wx1 * wy1);
                out = out + (float)src[((size_t)y1 * sw + x2r) * channels + c] * (wx2 * wy1);
                out = out + (float)src[((size_t)y2r * sw + x1) * channels + c] * (wx1 * wy2);
                out = out + (float)src[((size_t)y2r * sw + x2r) * channels + c] * (wx2 * wy2);
                float r = rintf(out);   /* round half to even (default rounding mode), like __float2int_rn */
                dst[((size_t)y * dw + x) * channels + c] = (uint8_t)(r < 0.f ? 0.f : r > 255.f ? 255.f : r);
            }
        }
}

static int uf_find(int32_t *parent, int i) {
    while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; }
    return i;
}

int cart_oracle_ccl(const uint8_t *planes, int w, int h, int32_t *ids) {
    size_t npx = (size_t)w * h;
    for (size_t i = 0; i < npx; i++) ids[i] = (int32_t)i;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int i = y * w + x;
            uint8_t c = planes[i];
            if (c > 1) continue;
            if (x > 0 && planes[i - 1] == c) {
                int a = uf_find(ids, i), b = uf_find(ids, i - 1);
                if (a < b) ids[b] = a; else ids[a] = b;
            }
            if (y > 0 && planes[i - w] == c) {
                int a = uf_find(ids, i), b = uf_find(ids, i - w);
                if (a < b) ids[b] = a; else ids[a] = b;
            }
        }
    int ncomp = 0;
    for (size_t i = 0; i < npx; i++) {
        if (planes[i] > 1) { ids[i] = -1; continue; }
        int r = uf_find(ids, (int)i);
        if (r == (int)i) ncomp++;
    }
    /* second pass: flatten (roots are minimal indices because unions always point to the smaller root) */
    for (size_t i = 0; i < npx; i++)
        if (planes[i] <= 1) ids[i] = uf_find(ids, (int)i);
    return ncomp;
}

/* a-11 (S12): the component table.  Roots are the pixels whose id equals their own linear index, so a raster scan meets
 * the components in ascending id order. */
int cart_oracle_ccl_stats(const uint8_t *planes, const int32_t *ids, int w, int h, int32_t *table, int max_components) {
    const size_t npx = (size_t)w * h;
    int32_t *slot = (int32_t *)malloc(npx * sizeof(int32_t));  /* root index -> table row */
    int n = 0;
    for (size_t i = 0; i < npx; ++i) {
        slot[i] = -1;
        if (ids[i] == (int32_t)i) {
            if (n < max_components) {
                int32_t *e = table + (size_t)n * 7;
                e[0] = (int32_t)i; e[1] = planes[i]; e[2] = 0; e[3] = w; e[4] = h; e[5] = -1; e[6] = -1;
                slot[i] = n;
            }
            ++n;
        }
    }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int32_t id = ids[(size_t)y * w + x];
            if (id < 0 || slot[id] < 0) continue;
            int32_t *e = table + (size_t)slot[id] * 7;
            e[2] += 1;
            if (x < e[3]) e[3] = x;
            if (y < e[4]) e[4] = y;
            if (x > e[5]) e[5] = x;
            if (y > e[6]) e[6] = y;
        }
    free(slot);
    return n;
}
