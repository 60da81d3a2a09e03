/*
 * cart_oracle_sp.c -- CPU restatement of the superpixel stage (contour relaxation) and the superpixel plane labelling.
 * TEST INFRASTRUCTURE ONLY / PARITY UNPINNED: see cart_oracle.h (spec items S13, S14).
 *
 * Reference files followed (read as text, nothing copied):
 *   src/modules/superpixels.cu:71-118                       module flow (YCrCb, iteration count, reset)
 *   src/modules/superpixels/contourrelaxation/initialization.cu:13-58
 *   src/modules/superpixels/contourrelaxation/contourrelaxation.cu:79-150,248-322,350-447
 *   src/modules/superpixels/contourrelaxation/features/gaussian.cu:34-210
 *   src/modules/superpixels/contourrelaxation/features/compactness.cu:30-215
 *   src/modules/planeseg/sp_planeseg.cu:27-178
 */
#include <float.h>
#include <stdlib.h>
#include <string.h>

#include "cart_oracle.h"

/* ---- S14 ----------------------------------------------------------------------------------------------------- */
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

void cart_oracle_bgr2ycrcb(const uint8_t *bgr, size_t src_step, int w, int h, uint8_t *ycrcb) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const uint8_t *p = bgr + (size_t)y * src_step + (size_t)x * 3;
            const int b = p[0], g = p[1], r = p[2];
            const int Y = (b * 1868 + g * 9617 + r * 4899 + 8192) >> 14;
            const int cr = ((r - Y) * 11682 + (128 << 14) + 8192) >> 14;
            const int cb = ((b - Y) * 9241 + (128 << 14) + 8192) >> 14;
            uint8_t *o = ycrcb + ((size_t)y * w + x) * 3;
            o[0] = sat_u8(Y); o[1] = sat_u8(cr); o[2] = sat_u8(cb);
        }
}

/* ---- block initialisation ------------------------------------------------------------------------------------ */
int cart_oracle_sp_block_init(int w, int h, int block_w, int block_h, uint16_t *labels) {
    const int nbx = (w + block_w - 1) / block_w, nby = (h + block_h - 1) / block_h;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) labels[(size_t)y * w + x] = (uint16_t)((y / block_h) * nbx + x / block_w);
    return nbx * nby;
}

/* ---- S13 log -------------------------------------------------------------------------------------------------- */
double cart_oracle_log(double x) {
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int e = (int)(bits >> 52) - 1023;
    bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m;
    memcpy(&m, &bits, 8);
    if (m > 0x1.6a09e667f3bcdp+0) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 23.0;
    for (int k = 10; k >= 0; --k) p = p * z + 1.0 / (double)(2 * k + 1);
    const double r = (2.0 * s) * p;
    const double de = (double)e;
    return de * 0x1.62e42fee00000p-1 + (r + de * 0x1.a39ef35793c76p-33);
}

/* ---- label statistics ----------------------------------------------------------------------------------------- */
/* channels: 0 x, 1 y (compactness) | 2 derivative ch0, 3 derivative ch1 (disparity feature) | 4 Y, 5 Cr, 6 Cb (colour) */
#define NCH 7
typedef struct {
    int64_t n;
    int64_t s[NCH], q[NCH]; /* exact integer sums / sums of squares (the reference keeps them in doubles) */
    double cost[NCH];
} label_stats;

typedef struct {
    const cart_oracle_sp_params *p;
    int w, h;
    const uint8_t *ycrcb;
    const int16_t *deriv2;
    label_stats *st;
    int ch_on[NCH];
} relax_ctx;

static inline int64_t pixel_value(const relax_ctx *c, int ch, int x, int y) {
    switch (ch) {
        case 0: return x;
        case 1: return y;
        case 2: case 3: return c->deriv2[((size_t)y * c->w + x) * 2 + (ch - 2)];
        default: return c->ycrcb[((size_t)y * c->w + x) * 3 + (ch - 4)];
    }
}

/* gaussian.cu:34-46 */
static double gauss_cost(int64_t n, int64_t s, int64_t q) {
    if (n == 0) return 0.0;
    const double dn = (double)n;
    const double a = (double)q / dn, b = (double)s / dn;
    double var = a - b * b;
    if (!(var >= 1.0 / 12.0)) var = 1.0 / 12.0; /* constants.hpp:33 */
    return (dn / 2 * cart_oracle_log(0x1.921fb54442d18p+2 * var)) + (dn / 2);
}

/* compactness.cu:30-37 */
static double compact_cost(int64_t n, int64_t s, int64_t q) {
    if (n == 0) return 0.0;
    const double ds = (double)s;
    return (double)q - (ds * ds) / (double)n;
}

static inline double channel_cost(int ch, int64_t n, int64_t s, int64_t q) {
    return ch < 2 ? compact_cost(n, s, q) : gauss_cost(n, s, q);
}

static void refresh_costs(relax_ctx *c, int max_label_id) {
    for (int l = 0; l <= max_label_id; ++l)
        for (int ch = 0; ch < NCH; ++ch)
            c->st[l].cost[ch] = c->ch_on[ch] ? channel_cost(ch, c->st[l].n, c->st[l].s[ch], c->st[l].q[ch]) : 0.0;
}

/* cost of giving pixel (x,y) (current label O) the label P; N = unique neighbour labels (contourrelaxation.cu:126-143) */
static double candidate_cost(const relax_ctx *c, int x, int y, const int *neigh, int O, int P, const int *N, int nN) {
    const cart_oracle_sp_params *p = c->p;
#define DIFF(dx, dy) (neigh[((dx) + 1) + ((dy) + 1) * 3] >= 0 && neigh[((dx) + 1) + ((dy) + 1) * 3] != P)
    const int nd = DIFF(-1, 0) + DIFF(1, 0) + DIFF(0, -1) + DIFF(0, 1);
    const int ng = DIFF(-1, -1) + DIFF(-1, 1) + DIFF(1, -1) + DIFF(1, 1);
#undef DIFF
    double cost = nd * p->direct_clique_cost + ng * p->diagonal_clique_cost;

    /* statistics of O and P as they would be after the move (only these two change) */
    const label_stats *so = &c->st[O], *sp = &c->st[P];
    int64_t on = so->n, pn = sp->n;
    double oc[NCH], pc[NCH];
    for (int ch = 0; ch < NCH; ++ch) { oc[ch] = so->cost[ch]; pc[ch] = sp->cost[ch]; }
    if (O != P) {
        on -= 1; pn += 1;
        for (int ch = 0; ch < NCH; ++ch) {
            if (!c->ch_on[ch]) continue;
            const int64_t v = pixel_value(c, ch, x, y);
            oc[ch] = channel_cost(ch, on, so->s[ch] - v, so->q[ch] - v * v);
            pc[ch] = channel_cost(ch, pn, sp->s[ch] + v, sp->q[ch] + v * v);
        }
    }
    /* features in the order superpixels.cu:61-68 adds them: compactness, disparity, colour */
    if (p->compactness_weight > 0) {
        double f = 0;
        for (int i = 0; i < nN; ++i) {
            const int L = N[i];
            const int64_t n = L == O ? on : L == P ? pn : c->st[L].n;
            const double *k = L == O ? oc : L == P ? pc : c->st[L].cost;
            if (n == 0) continue;
            f += k[0] + k[1];
        }
        if (p->progressive_compactness_cost > 0.0)
            f *= 1.0 + p->progressive_compactness_cost * ((double)c->h - (double)y) / (double)c->h;
        cost += p->compactness_weight * f;
    }
    if (p->disparity_weight > 0) {
        double f = 0;
        for (int i = 0; i < nN; ++i) {
            const int L = N[i];
            const int64_t n = L == O ? on : L == P ? pn : c->st[L].n;
            const double *k = L == O ? oc : L == P ? pc : c->st[L].cost;
            for (int ch = 2; ch < 4; ++ch) {
                if (n == 0) continue;
                f += k[ch];
            }
        }
        cost += p->disparity_weight * (f / 2.0);
    }
    if (p->image_weight > 0) {
        double f = 0;
        for (int i = 0; i < nN; ++i) {
            const int L = N[i];
            const int64_t n = L == O ? on : L == P ? pn : c->st[L].n;
            const double *k = L == O ? oc : L == P ? pc : c->st[L].cost;
            for (int ch = 4; ch < 7; ++ch) {
                if (n == 0) continue;
                f += k[ch];
            }
        }
        cost += p->image_weight * (f / 3.0);
    }
    return cost;
}

long cart_oracle_sp_relax(const cart_oracle_sp_params *p, uint16_t *labels, int w, int h, int max_label_id,
                          const uint8_t *ycrcb, const int16_t *deriv2, int iterations) {
    if (!p || !labels || w <= 0 || h <= 0 || max_label_id <= 0 || max_label_id >= 16384 || iterations < 0) return -1;
    if (p->direct_clique_cost < 0 || p->compactness_weight < 0 || p->image_weight < 0 || p->disparity_weight < 0) return -1;
    if ((p->image_weight > 0 && !ycrcb) || (p->disparity_weight > 0 && !deriv2)) return -1;
    const size_t px = (size_t)w * h;
    for (size_t i = 0; i < px; ++i)
        if (labels[i] >= max_label_id) return -1;

    relax_ctx c = {p, w, h, ycrcb, deriv2, NULL, {0}};
    c.ch_on[0] = c.ch_on[1] = p->compactness_weight > 0;
    c.ch_on[2] = c.ch_on[3] = p->disparity_weight > 0;
    c.ch_on[4] = c.ch_on[5] = c.ch_on[6] = p->image_weight > 0;
    c.st = (label_stats *)calloc((size_t)max_label_id + 1, sizeof(label_stats));
    uint16_t *next = (uint16_t *)malloc(px * sizeof(uint16_t));
    if (!c.st || !next) { free(c.st); free(next); return -1; }

    /* statistics of the current labelling over the WHOLE image (S13) */
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            label_stats *s = &c.st[labels[(size_t)y * w + x]];
            s->n += 1;
            for (int ch = 0; ch < NCH; ++ch) {
                if (!c.ch_on[ch]) continue;
                const int64_t v = pixel_value(&c, ch, x, y);
                s->s[ch] += v; s->q[ch] += v * v;
            }
        }
    refresh_costs(&c, max_label_id);

    long changes = 0;
    for (int it = 0; it < iterations; ++it) {
#pragma omp parallel for schedule(dynamic, 4)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                int neigh[9], N[9], nN = 0;
                for (int dx = -1; dx <= 1; ++dx)
                    for (int dy = -1; dy <= 1; ++dy) {
                        const int xx = x + dx, yy = y + dy;
                        neigh[(dx + 1) + (dy + 1) * 3] = (xx < 0 || yy < 0 || xx >= w || yy >= h) ? -1 : labels[(size_t)yy * w + xx];
                    }
                for (int dx = -1; dx <= 1; ++dx)      /* contourrelaxation.cu:82-108: dx outer, dy inner */
                    for (int dy = -1; dy <= 1; ++dy) {
                        const int L = neigh[(dx + 1) + (dy + 1) * 3];
                        if (L < 0) continue;
                        int found = 0;
                        for (int k = 0; k < nN; ++k) found |= N[k] == L;
                        if (!found) N[nN++] = L;
                    }
                const int cur = neigh[4];
                int best = cur;
                if (nN > 1) {
                    double min_cost = DBL_MAX;
                    for (int i = 0; i < nN; ++i) {
                        const double k = candidate_cost(&c, x, y, neigh, cur, N[i], N, nN);
                        if (k < min_cost) { min_cost = k; best = N[i]; }
                    }
                }
                next[(size_t)y * w + x] = (uint16_t)best;
            }
        /* contourrelaxation.cu:296-322: apply, update statistics */
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const size_t i = (size_t)y * w + x;
                if (next[i] == labels[i]) continue;
                label_stats *so = &c.st[labels[i]], *sn = &c.st[next[i]];
                so->n -= 1; sn->n += 1;
                for (int ch = 0; ch < NCH; ++ch) {
                    if (!c.ch_on[ch]) continue;
                    const int64_t v = pixel_value(&c, ch, x, y);
                    so->s[ch] -= v; so->q[ch] -= v * v;
                    sn->s[ch] += v; sn->q[ch] += v * v;
                }
                labels[i] = next[i];
                ++changes;
            }
        refresh_costs(&c, max_label_id);
    }
    free(c.st); free(next);
    return changes;
}

/* ---- superpixel plane labelling -------------------------------------------------------------------------------- */
void cart_oracle_sp_classify(const int16_t *deriv2, const uint16_t *labels, int w, int h, int max_label,
                             const cart_oracle_plane_params *p, int n_prev, const uint8_t *const *prev_planes,
                             const int16_t *const *flows, uint8_t *planes_unsmoothed, uint8_t *planes) {
    uint16_t *votes = (uint16_t *)calloc((size_t)(max_label > 0 ? max_label : 1) * 3, sizeof(uint16_t));
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            const int d = deriv2[i * 2];
            int plane = 2;
            if (d != CART_ORACLE_INVALID && d >= p->horizontal_min && d < p->horizontal_max) plane = 0;
            else if (d != CART_ORACLE_INVALID && d >= p->vertical_min && d < p->vertical_max) plane = 1;
            planes_unsmoothed[i] = (uint8_t)plane;
            if (n_prev > 0) { /* sp_planeseg.cu:78-114 */
                int v[3] = {0, 0, 0};
                v[plane] += 2;
                int px = x, py = y;
                for (int k = 0; k < n_prev; ++k) {
                    px -= flows[k][i * 2] >> 5;
                    py -= flows[k][i * 2 + 1] >> 5;
                    if (px < 0 || py < 0 || px >= w || py >= h) continue;
                    v[prev_planes[k][(size_t)py * w + px]]++;
                }
                plane = v[0] > v[1] ? 0 : 1;
                if (v[plane] < v[2]) plane = 2;
            }
            const int L = labels[i];
            if (L < max_label) votes[L * 3 + plane]++; /* u16 wrap-around like the reference's counters */
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            const int L = labels[i];
            int best = 2;
            if (L < max_label) { /* sp_planeseg.cu:145-160 */
                int mx = votes[L * 3 + 2];
                if (votes[L * 3 + 1] > mx) { mx = votes[L * 3 + 1]; best = 1; }
                if (votes[L * 3 + 0] > mx) best = 0;
            }
            planes[i] = (uint8_t)best;
        }
    free(votes);
}

/* ---- S15 optical flow (census block matching) ------------------------------------------------------------------- */
void cart_oracle_block_flow(const uint32_t *cen_cur, const uint32_t *cen_prev, int w, int h, int radius, int block,
                            int16_t *flow) {
#pragma omp parallel for schedule(dynamic, 2)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            long best = -1;
            int bu = 0, bv = 0;
            for (int pass = 0; pass < 2; ++pass) {            /* pass 0: (0,0); pass 1: the scan */
                for (int v = pass ? -radius : 0; v <= (pass ? radius : 0); ++v)
                    for (int u = pass ? -radius : 0; u <= (pass ? radius : 0); ++u) {
                        if (pass && u == 0 && v == 0) continue;
                        long c = 0;
                        for (int dy = -block; dy <= block; ++dy)
                            for (int dx = -block; dx <= block; ++dx) {
                                const int qx = x + dx, qy = y + dy;
                                if (qx < 0 || qy < 0 || qx >= w || qy >= h) continue;
                                const int px = qx - u, py = qy - v;
                                const uint32_t fp = (px < 0 || py < 0 || px >= w || py >= h) ? 0u : cen_prev[(size_t)py * w + px];
                                c += __builtin_popcount(cen_cur[(size_t)qy * w + qx] ^ fp);
                            }
                        if (best < 0 || c < best) { best = c; bu = u; bv = v; }
                    }
            }
            flow[((size_t)y * w + x) * 2] = (int16_t)(bu * 32);
            flow[((size_t)y * w + x) * 2 + 1] = (int16_t)(bv * 32);
        }
}
