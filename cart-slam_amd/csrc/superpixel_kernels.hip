// superpixel_kernels.hip -- contour-relaxation superpixels and the superpixel plane labelling for gfx950.
//
// What the reference does (read as text, nothing copied):
//   createBlockInitialization / performBlockIntialization   src/modules/superpixels/contourrelaxation/initialization.cu:13-58
//   ContourRelaxation::relax                                 .../contourrelaxation.cu:350-447
//     findBorderPixels -> host sync -> performRelaxation -> updateLabels, per iteration (:409-426)
//   CUDAGaussianFeature / CUDACompactnessFeature             .../features/gaussian.cu:34-210, compactness.cu:30-215
//   performSuperPixelClassifications, classifyPlanes         src/modules/planeseg/sp_planeseg.cu:27-178
//
// MI355X design: the reference compacts border pixels into a global list and reads its length back to the host in
// every iteration (one device->host round trip per iteration) and keeps per-label statistics behind device-side
// virtual calls.  Here one iteration is two launches and no host round trip:
//   sp_relax_kernel   one 32x16 tile per workgroup; labels + halo in LDS; the tile's ~20 distinct labels get slots in an
//                     LDS table that holds their statistics rows and feature costs (fetched once per workgroup); border
//                     pixels are compacted into an LDS list, ordered by their number of candidate labels, so that all
//                     256 lanes work on border pixels with like trip counts; every border pixel evaluates its candidate
//                     labels against the label statistics of the iteration start (Jacobi, oracle S13), writes the
//                     next label image and accumulates the statistics DELTA of its own move with 64-bit integer
//                     atomics (sums of integers: exact, order independent);
//   sp_fold_kernel    statistics += delta, delta = 0, per-label feature costs refreshed (one thread per label; not after
//                     the last sweep of a call: the next call rebuilds the statistics from the image and the labels).
//   sp_stats_kernel   the statistics of a call's first sweep: per tile into the same kind of LDS table, one global
//                     atomic per label and row.
// Statistics are structure-of-arrays [row][label] so that a wave's candidate look-ups hit the same few cache lines.
// All cost arithmetic is IEEE double in the oracle's operation order, no FMA contraction, log() = the S13 sequence.
#include "engine_internal.h"

#pragma clang fp contract(off)

namespace cart_amd {

namespace {

#ifndef CART_SP_TILE_W
#define CART_SP_TILE_W 32
#define CART_SP_TILE_H 16
#endif
constexpr int kTileW = CART_SP_TILE_W, kTileH = CART_SP_TILE_H;   // 32x16: ~1000 workgroups at 1242x375 (64x16 left half the CUs with one workgroup)
static_assert(kTileW * kTileH % 256 == 0 && 256 % kTileW == 0 && kTileW <= 256 && kTileH <= 255, "tile must be a multiple of the 256-thread block");
constexpr uint16_t kOob = 0xFFFFu;

__device__ __forceinline__ double sp_log(double x) {  // oracle S13
    unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    int e = (int)(bits >> 52) - 1023;
    double m = __longlong_as_double((long long)((bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
    if (m > 0x1.6a09e667f3bcdp+0) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 23.0;
#pragma unroll
    for (int k = 10; k >= 0; --k) p = p * z + 1.0 / (double)(2 * k + 1);
    const double r = (2.0 * s) * p;
    const double de = (double)e;
    return de * 0x1.62e42fee00000p-1 + (r + de * 0x1.a39ef35793c76p-33);
}

__device__ __forceinline__ double gauss_cost(long long n, long long s, long long q) {  // gaussian.cu:34-46
    if (n == 0) return 0.0;
    const double dn = (double)n;
    const double a = (double)q / dn, b = (double)s / dn;
    double var = a - b * b;
    if (!(var >= 1.0 / 12.0)) var = 1.0 / 12.0;
    return (dn / 2 * sp_log(0x1.921fb54442d18p+2 * var)) + (dn / 2);
}

__device__ __forceinline__ double compact_cost(long long n, long long s, long long q) {  // compactness.cu:30-37
    if (n == 0) return 0.0;
    const double ds = (double)s;
    return (double)q - (ds * ds) / (double)n;
}

__device__ __forceinline__ double channel_cost(int ch, long long n, long long s, long long q) {
    return ch < 2 ? compact_cost(n, s, q) : gauss_cost(n, s, q);
}

// channel values of one pixel: 0 x, 1 y | 2,3 derivative ch0,ch1 | 4,5,6 Y,Cr,Cb
__device__ __forceinline__ void pixel_values(const SpRelaxArgs &a, int x, int y, long long v[kSpChannels]) {
    v[0] = x; v[1] = y;
    v[2] = v[3] = 0;
    if (a.ch_mask & 0x0cu) {
        const int d = *reinterpret_cast<const int *>(reinterpret_cast<const uint8_t *>(a.deriv) + (size_t)y * a.deriv_step + (size_t)x * 4);
        v[2] = (short)(d & 0xffff); v[3] = (short)(d >> 16);
    }
    const uint32_t c = a.ycc[(size_t)y * a.w + x];
    v[4] = c & 0xff; v[5] = (c >> 8) & 0xff; v[6] = (c >> 16) & 0xff;
}

__device__ __forceinline__ void atomic_add_i64(long long *p, long long v) {
    atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v);
}

}  // namespace

// ------------------------------------------------------------------ block initialisation (initialization.cu:13-58)
__global__ __launch_bounds__(256) void sp_block_init_kernel(uint16_t *labels, int w, int h, int bw, int bh, int nbx) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < w && y < h) labels[(size_t)y * w + x] = (uint16_t)((y / bh) * nbx + x / bw);
}

// ------------------------------------------------------------------ BGR -> packed YCrCb (oracle S14, superpixels.cu:81)
__global__ __launch_bounds__(256) void sp_ycrcb_kernel(const uint8_t *img, size_t step, int channels, uint32_t *ycc, int w, int h) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const uint8_t *p = img + (size_t)y * step + (size_t)x * channels;
    const int b = p[0], g = channels == 3 ? p[1] : b, r = channels == 3 ? p[2] : b;
    const int Y = (b * 1868 + g * 9617 + r * 4899 + 8192) >> 14;
    const int cr = ((r - Y) * 11682 + (128 << 14) + 8192) >> 14;
    const int cb = ((b - Y) * 9241 + (128 << 14) + 8192) >> 14;
    const uint32_t yy = (uint32_t)min(max(Y, 0), 255), c1 = (uint32_t)min(max(cr, 0), 255), c2 = (uint32_t)min(max(cb, 0), 255);
    ycc[(size_t)y * w + x] = yy | (c1 << 8) | (c2 << 16);
}

// ------------------------------------------------------------------ statistics of the current labelling
// One workgroup per 32x16 tile: the tile's ~20 labels get slots in an LDS table (as in sp_relax_kernel below), every thread adds its two pixels to the
// slot's 15 accumulators with LDS atomics, and the tile leaves ONE global atomic per label and row (~300 per tile instead of ~900 from per-strip flushes:
// 35.5 -> 21.5 us per frame at 1242x375).  Sums of integers: exact and order independent.  A tile with more labels than slots adds straight to global memory.
constexpr int kSpSlots = 64;
constexpr unsigned kSpEmpty = 0xFFFFFFFFu;
__device__ __forceinline__ unsigned sp_hash(unsigned L) { return (L ^ (L >> 6) ^ (L >> 11)) & (kSpSlots - 1); }
// inserts L (first of a run of equal labels) into the open-addressing table; false when the table is full
__device__ __forceinline__ bool sp_table_insert(unsigned *hkey, unsigned L) {
    unsigned h = sp_hash(L);
    for (int probes = 0; probes < kSpSlots; ++probes) {
        const unsigned old = atomicCAS(&hkey[h], kSpEmpty, L);
        if (old == kSpEmpty || old == L) return true;
        h = (h + 1) & (kSpSlots - 1);
    }
    return false;
}
__device__ __forceinline__ unsigned sp_table_find(const unsigned *hkey, unsigned L) {   // L is in the table
    unsigned h = sp_hash(L);
    while (hkey[h] != L) h = (h + 1) & (kSpSlots - 1);
    return h;
}

__global__ __launch_bounds__(256) void sp_stats_kernel(SpRelaxArgs a) {
    __shared__ uint16_t tile[kTileH][kTileW];
    __shared__ unsigned hkey[kSpSlots];
    __shared__ unsigned long long acc[kSpStatRows][kSpSlots];
    __shared__ int overflow;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    if (tid == 0) overflow = 0;
    if (tid < kSpSlots) hkey[tid] = kSpEmpty;
    for (int i = tid; i < kSpStatRows * kSpSlots; i += 256) (&acc[0][0])[i] = 0ull;
    for (int i = tid; i < kTileW * kTileH; i += 256) {
        const int lx = i % kTileW, ly = i / kTileW;
        const int x = x0 + lx, y = y0 + ly;
        tile[ly][lx] = (x < a.w && y < a.h) ? a.cur[(size_t)y * a.w + x] : kOob;
    }
    __syncthreads();
    for (int i = tid; i < kTileW * kTileH; i += 256) {
        const int lx = i % kTileW, ly = i / kTileW;
        const unsigned L = tile[ly][lx];
        if (L == kOob || (lx > 0 && tile[ly][lx - 1] == L)) continue;
        if (!sp_table_insert(hkey, L)) overflow = 1;
    }
    __syncthreads();
    const bool direct = overflow != 0;
    for (int i = tid; i < kTileW * kTileH; i += 256) {
        const int lx = i % kTileW, ly = i / kTileW;
        const unsigned L = tile[ly][lx];
        if (L == kOob) continue;
        long long v[kSpChannels];
        pixel_values(a, x0 + lx, y0 + ly, v);
        if (direct) {
            atomic_add_i64(&a.stats[L], 1);
#pragma unroll
            for (int ch = 0; ch < kSpChannels; ++ch)
                if ((a.ch_mask >> ch) & 1u) {
                    atomic_add_i64(&a.stats[(size_t)(1 + ch) * a.ld + L], v[ch]);
                    atomic_add_i64(&a.stats[(size_t)(8 + ch) * a.ld + L], v[ch] * v[ch]);
                }
        } else {
            const unsigned sl = sp_table_find(hkey, L);
            atomicAdd(&acc[0][sl], 1ull);
#pragma unroll
            for (int ch = 0; ch < kSpChannels; ++ch)
                if ((a.ch_mask >> ch) & 1u) {
                    atomicAdd(&acc[1 + ch][sl], (unsigned long long)v[ch]);
                    atomicAdd(&acc[8 + ch][sl], (unsigned long long)(v[ch] * v[ch]));
                }
        }
    }
    if (direct) return;
    __syncthreads();
    for (int i = tid; i < kSpStatRows * kSpSlots; i += 256) {
        const int sl = i & (kSpSlots - 1), row = i / kSpSlots;
        const unsigned L = hkey[sl];
        const unsigned long long s = acc[row][sl];
        if (L != kSpEmpty && s != 0ull) atomicAdd(reinterpret_cast<unsigned long long *>(&a.stats[(size_t)row * a.ld + L]), s);
    }
}

// statistics += delta; delta = 0; per-label feature costs (gaussian.cu:34-46, compactness.cu:30-37).  Eight lanes per label: lanes 0..6 take one channel each
// (its two statistics rows and its cost: three divisions and a logarithm series at most -- one thread per label walked seven of them in sequence, 7.4 us per
// launch between every two sweeps), lane 7 the pixel count, which it hands to the others before anything is written.
__global__ __launch_bounds__(256) void sp_fold_kernel(long long *stats, long long *delta, double *costs, int ld, unsigned ch_mask) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int l = t >> 3, p = t & 7;
    const bool live = l < ld;
    long long n = 0;
    if (live && p == 7) {
        n = stats[l] + delta[l];
        stats[l] = n;
        delta[l] = 0;
    }
    n = __shfl(n, (int)(threadIdx.x & 63u) | 7);   // every lane of the wave takes part (ld need not fill the last wave)
    if (!live || p == 7) return;
    const size_t is = (size_t)(1 + p) * ld + l, iq = (size_t)(8 + p) * ld + l;
    const long long s = stats[is] + delta[is], q = stats[iq] + delta[iq];
    stats[is] = s; stats[iq] = q;
    delta[is] = 0; delta[iq] = 0;
    costs[(size_t)p * ld + l] = ((ch_mask >> p) & 1u) ? channel_cost(p, n, s, q) : 0.0;
}

// ------------------------------------------------------------------ one relaxation sweep (contourrelaxation.cu:248-322)
// Where a border pixel's candidate labels get their statistics from.  A tile and its halo hold ~20 distinct labels; their 15 statistics rows and 7 feature
// costs are fetched ONCE per workgroup into an LDS table (SpLdsAcc: the per-pixel loops then address labels by their table slot), instead of once per pixel,
// candidate and neighbour label from global memory, one dependent load after the other (the sweep is bound by exactly those latencies: 43 us for 6 us worth of
// arithmetic, profiles/r04_superpixels.txt).  A tile with more than kSpSlots distinct labels takes the global-memory path (SpGlobalAcc: ids are the labels).
struct SpGlobalAcc {
    const long long *stats; const double *costs; int ld;
    __device__ __forceinline__ long long stat(int row, int id) const { return stats[(size_t)row * ld + id]; }
    __device__ __forceinline__ double cost(int ch, int id) const { return costs[(size_t)ch * ld + id]; }
    __device__ __forceinline__ int label(int id) const { return id; }
};
struct SpLdsAcc {
    const long long (*st)[kSpSlots]; const double (*co)[kSpSlots]; const unsigned *key;
    __device__ __forceinline__ long long stat(int row, int id) const { return st[row][id]; }
    __device__ __forceinline__ double cost(int ch, int id) const { return co[ch][id]; }
    __device__ __forceinline__ int label(int id) const { return (int)key[id]; }
};

// The border pixels of a tile: candidates, costs, choice, statistics delta.  `ids` = the tile + halo as label ids of `acc` (kOob outside the image).
template <class Acc>
__device__ __forceinline__ void sp_relax_border(const SpRelaxArgs &a, const Acc &acc, const uint16_t (*ids)[kTileW + 2], const uint16_t *list, int n_border,
                                                uint16_t (*cand)[256], int x0, int y0, int tid) {
    const int ld = a.ld;
    for (int i = tid; i < n_border; i += 256) {
        const int lx = list[i] & 0xff, ly = list[i] >> 8;
        const int x = x0 + lx, y = y0 + ly;
        int nb[9];
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) nb[(dx + 1) + (dy + 1) * 3] = ids[ly + 1 + dy][lx + 1 + dx];
        // unique neighbour labels, dx outer / dy inner like getNeighbourLabels (contourrelaxation.cu:79-108)
        int nN = 0;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int L = nb[(dx + 1) + (dy + 1) * 3];
                if (L != kOob) {
                    bool found = false;
                    for (int k = 0; k < nN; ++k) found |= cand[k][tid] == L;
                    if (!found) cand[nN++][tid] = (uint16_t)L;
                }
            }
        const int O = nb[4];
        long long v[kSpChannels];
        pixel_values(a, x, y, v);
        // O after losing this pixel (the same for every candidate != O)
        const long long on_less = acc.stat(0, O) - 1;
        double oc[kSpChannels];
#pragma unroll
        for (int ch = 0; ch < kSpChannels; ++ch)
            oc[ch] = ((a.ch_mask >> ch) & 1u) ? channel_cost(ch, on_less, acc.stat(1 + ch, O) - v[ch], acc.stat(8 + ch, O) - v[ch] * v[ch]) : 0.0;
        int best = O;
        double min_cost = 0x1.fffffffffffffp+1023;  // DBL_MAX
        for (int ci = 0; ci < nN; ++ci) {
            const int P = cand[ci][tid];
#define CART_DIFF(k) ((nb[k] != kOob) & (nb[k] != P))
            const int nd = CART_DIFF(3) + CART_DIFF(5) + CART_DIFF(1) + CART_DIFF(7);
            const int ng = CART_DIFF(0) + CART_DIFF(6) + CART_DIFF(2) + CART_DIFF(8);
#undef CART_DIFF
            double cost = nd * a.direct + ng * a.diagonal;
            const bool move = P != O;
            long long pn_more = 0;
            double pc[kSpChannels];
            if (move) {
                pn_more = acc.stat(0, P) + 1;
#pragma unroll
                for (int ch = 0; ch < kSpChannels; ++ch)
                    pc[ch] = ((a.ch_mask >> ch) & 1u) ? channel_cost(ch, pn_more, acc.stat(1 + ch, P) + v[ch], acc.stat(8 + ch, P) + v[ch] * v[ch]) : 0.0;
            } else {
#pragma unroll
                for (int ch = 0; ch < kSpChannels; ++ch) pc[ch] = 0.0;
            }
            double fc = 0, fd = 0, fi = 0;
            for (int j = 0; j < nN; ++j) {
                const int L = cand[j][tid];
                long long n;
                double k[kSpChannels];
                if (move && L == O) {
                    n = on_less;
#pragma unroll
                    for (int ch = 0; ch < kSpChannels; ++ch) k[ch] = oc[ch];
                } else if (move && L == P) {
                    n = pn_more;
#pragma unroll
                    for (int ch = 0; ch < kSpChannels; ++ch) k[ch] = pc[ch];
                } else {
                    n = acc.stat(0, L);
#pragma unroll
                    for (int ch = 0; ch < kSpChannels; ++ch) k[ch] = acc.cost(ch, L);
                }
                if (n == 0) continue;
                fc += k[0] + k[1];
                fd += k[2]; fd += k[3];
                fi += k[4]; fi += k[5]; fi += k[6];
            }
            if (a.w_comp > 0) {
                if (a.prog > 0.0) fc *= 1.0 + a.prog * ((double)a.h - (double)y) / (double)a.h;
                cost += a.w_comp * fc;
            }
            if (a.w_disp > 0) cost += a.w_disp * (fd / 2.0);
            if (a.w_img > 0) cost += a.w_img * (fi / 3.0);
            if (cost < min_cost) { min_cost = cost; best = P; }
        }
        const int lbest = acc.label(best), lO = acc.label(O);
        a.next[(size_t)y * a.w + x] = (uint16_t)lbest;
        if (best != O) {  // updateLabels (contourrelaxation.cu:296-322), as a delta folded in by sp_fold_kernel
            atomic_add_i64(&a.delta[lO], -1);
            atomic_add_i64(&a.delta[lbest], 1);
#pragma unroll
            for (int ch = 0; ch < kSpChannels; ++ch)
                if ((a.ch_mask >> ch) & 1u) {
                    atomic_add_i64(&a.delta[(size_t)(1 + ch) * ld + lO], -v[ch]);
                    atomic_add_i64(&a.delta[(size_t)(8 + ch) * ld + lO], -v[ch] * v[ch]);
                    atomic_add_i64(&a.delta[(size_t)(1 + ch) * ld + lbest], v[ch]);
                    atomic_add_i64(&a.delta[(size_t)(8 + ch) * ld + lbest], v[ch] * v[ch]);
                }
        }
    }
}

__global__ __launch_bounds__(256) void sp_relax_kernel(SpRelaxArgs a) {
    __shared__ uint16_t tile[kTileH + 2][kTileW + 2];    // labels
    __shared__ uint16_t tslot[kTileH + 2][kTileW + 2];   // the same entries as slots of the label table
    __shared__ uint16_t list[kTileW * kTileH], sorted[kTileW * kTileH];
    __shared__ uint16_t cand[9][256];
    __shared__ int bucket[10];
    __shared__ unsigned hkey[kSpSlots];
    __shared__ long long c_stats[kSpStatRows][kSpSlots];
    __shared__ double c_costs[kSpChannels][kSpSlots];
    __shared__ int count, overflow;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    if (tid == 0) { count = 0; overflow = 0; }
    if (tid < kSpSlots) hkey[tid] = kSpEmpty;
    if (tid < 10) bucket[tid] = 0;
    for (int i = tid; i < (kTileH + 2) * (kTileW + 2); i += 256) {
        const int ly = i / (kTileW + 2), lx = i % (kTileW + 2);
        const int x = x0 + lx - 1, y = y0 + ly - 1;
        tile[ly][lx] = (x >= 0 && x < a.w && y >= 0 && y < a.h) ? a.cur[(size_t)y * a.w + x] : kOob;
    }
    __syncthreads();
    // label table: open addressing, one insertion per run of equal labels along a row (~50 insertions instead of 612)
    for (int i = tid; i < (kTileH + 2) * (kTileW + 2); i += 256) {
        const int ly = i / (kTileW + 2), lx = i % (kTileW + 2);
        const unsigned L = tile[ly][lx];
        if (L == kOob || (lx > 0 && tile[ly][lx - 1] == L)) continue;
        if (!sp_table_insert(hkey, L)) overflow = 1;
    }
    // pixels whose in-image neighbourhood holds a single label keep it (they would have one candidate only)
#pragma unroll
    for (int k = 0; k < kTileW * kTileH / 256; ++k) {
        const int lx = tid % kTileW, ly = tid / kTileW + (256 / kTileW) * k;
        const int x = x0 + lx, y = y0 + ly;
        if (x < a.w && y < a.h) {
            const uint16_t c = tile[ly + 1][lx + 1];
            bool border = false;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const uint16_t n = tile[ly + 1 + dy][lx + 1 + dx];
                    border |= (n != kOob) & (n != c);
                }
            if (border) list[atomicAdd(&count, 1)] = (uint16_t)(lx | (ly << 8));
            else a.next[(size_t)y * a.w + x] = c;
        }
    }
    __syncthreads();
    const int n_border = count;
    if (n_border == 0) return;
    // The border pixels ordered by their number of candidate labels: a wave walks its candidate loops as often as its pixel with the MOST candidates needs
    // (each walk ~2 500 issue cycles of double-precision divisions and logarithms), and in list order nearly every wave holds a 3- or 4-label pixel among
    // its 2-label ones.  Counting sort over the nine possible counts; the order inside a count does not matter (every pixel reads the sweep's start state).
    {
        int nn[kTileW * kTileH / 256];
#pragma unroll
        for (int k = 0; k < kTileW * kTileH / 256; ++k) {
            const int i = tid + 256 * k;
            nn[k] = 0;
            if (i < n_border) {
                const int lx = list[i] & 0xff, ly = list[i] >> 8;
                uint16_t nb[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) nb[q] = tile[ly + q / 3][lx + q % 3];
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    bool fresh = nb[q] != kOob;
#pragma unroll
                    for (int r = 0; r < q; ++r) fresh &= nb[r] != nb[q];
                    nn[k] += fresh;
                }
                atomicAdd(&bucket[nn[k]], 1);
            }
        }
        __syncthreads();
        if (tid == 0) {
            int base = 0;
            for (int q = 0; q < 10; ++q) { const int c = bucket[q]; bucket[q] = base; base += c; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kTileW * kTileH / 256; ++k) {
            const int i = tid + 256 * k;
            if (i < n_border) sorted[atomicAdd(&bucket[nn[k]], 1)] = list[i];
        }
        __syncthreads();
    }
    if (overflow) {   // more distinct labels than the table holds: every look-up from global memory
        sp_relax_border(a, SpGlobalAcc{a.stats, a.costs, a.ld}, tile, sorted, n_border, cand, x0, y0, tid);
        return;
    }
    for (int i = tid; i < (kTileH + 2) * (kTileW + 2); i += 256) {
        const int ly = i / (kTileW + 2), lx = i % (kTileW + 2);
        const unsigned L = tile[ly][lx];
        tslot[ly][lx] = (uint16_t)(L != kOob ? sp_table_find(hkey, L) : (unsigned)kOob);
    }
    for (int i = tid; i < (kSpStatRows + kSpChannels) * kSpSlots; i += 256) {
        const int slot = i & (kSpSlots - 1), row = i / kSpSlots;
        const unsigned L = hkey[slot];
        if (L == kSpEmpty) continue;
        if (row < kSpStatRows) c_stats[row][slot] = a.stats[(size_t)row * a.ld + L];
        else c_costs[row - kSpStatRows][slot] = a.costs[(size_t)(row - kSpStatRows) * a.ld + L];
    }
    __syncthreads();
    sp_relax_border(a, SpLdsAcc{c_stats, c_costs, hkey}, tslot, sorted, n_border, cand, x0, y0, tid);
}

// ------------------------------------------------------------------ label image copies (tight <-> pitched) + range check
__global__ __launch_bounds__(256) void sp_copy_kernel(const uint16_t *src, size_t src_step, uint16_t *dst, size_t dst_step, int w, int h, int *max_seen) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const uint16_t v = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(src) + (size_t)y * src_step + (size_t)x * 2);
    *reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(dst) + (size_t)y * dst_step + (size_t)x * 2) = v;
    if (max_seen) atomicMax(max_seen, (int)v);
}

// ------------------------------------------------------------------ superpixel plane labelling (sp_planeseg.cu:27-178)
__global__ __launch_bounds__(256) void sp_vote_kernel(SpClassifyArgs a) {
    const int strips = (a.w + 15) / 16;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= strips * a.h) return;
    const int y = id / strips, xs = (id % strips) * 16, xe = min(xs + 16, a.w);
    const uint16_t *lrow = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(a.labels) + (size_t)y * a.labels_step);
    const uint8_t *drow = reinterpret_cast<const uint8_t *>(a.deriv) + (size_t)y * a.deriv_step;
    unsigned acc[3] = {0, 0, 0};
    int cur = lrow[xs];
    for (int x = xs; x <= xe; ++x) {
        const int L = x < xe ? lrow[x] : -1;
        if (L != cur) {
            if (cur < a.max_label) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    if (acc[p]) atomicAdd(&a.votes[cur * 3 + p], acc[p]);
            }
            acc[0] = acc[1] = acc[2] = 0;
            cur = L;
        }
        if (x == xe) break;
        const int d = *reinterpret_cast<const short *>(drow + (size_t)x * 4);  // channel 0 = vertical derivative
        int plane = 2;
        if (d != -32768 && d >= a.p.horizontal_min && d < a.p.horizontal_max) plane = 0;
        else if (d != -32768 && d >= a.p.vertical_min && d < a.p.vertical_max) plane = 1;
        a.unsmoothed[(size_t)y * a.unsmoothed_step + x] = (uint8_t)plane;  // sp_planeseg.cu:75 stores the pre-vote class
        if (a.t.n_prev > 0) {  // :78-114, current frame counts twice, UNKNOWN wins when it out-votes the winner
            int v0 = plane == 0 ? 2 : 0, v1 = plane == 1 ? 2 : 0, v2 = plane == 2 ? 2 : 0;
            int px = x, py = y;
            for (int k = 0; k < a.t.n_prev; ++k) {
                const int f = *reinterpret_cast<const int *>(reinterpret_cast<const uint8_t *>(a.t.flow[k]) + (size_t)y * a.t.flow_step[k] + (size_t)x * 4);
                px -= ((int)(short)(f & 0xffff)) >> 5;
                py -= (f >> 16) >> 5;
                if (px < 0 || py < 0 || px >= a.w || py >= a.h) continue;
                const int pv = a.t.prev[k][(size_t)py * a.t.prev_step[k] + px];
                v0 += pv == 0; v1 += pv == 1; v2 += pv == 2;
            }
            plane = v0 > v1 ? 0 : 1;
            if ((plane == 0 ? v0 : v1) < v2) plane = 2;
        }
        acc[plane] += 1;
    }
}

__global__ __launch_bounds__(256) void sp_assign_kernel(SpClassifyArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.w || y >= a.h) return;
    const int L = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(a.labels) + (size_t)y * a.labels_step + (size_t)x * 2);
    int best = 2;
    if (L < a.max_label) {  // u16 counters in the reference: votes wrap modulo 65536
        const int h = a.votes[L * 3 + 0] & 0xffff, v = a.votes[L * 3 + 1] & 0xffff, u = a.votes[L * 3 + 2] & 0xffff;
        int mx = u;
        if (v > mx) { mx = v; best = 1; }
        if (h > mx) best = 0;
    }
    a.planes[(size_t)y * a.planes_step + x] = (uint8_t)best;
}

// ------------------------------------------------------------------ launchers
static dim3 px_grid(int w, int h) { return dim3((w + 63) / 64, (h + 3) / 4); }

void launch_sp_block_init(uint16_t *labels, int w, int h, int bw, int bh, hipStream_t s) {
    hipLaunchKernelGGL(sp_block_init_kernel, px_grid(w, h), dim3(256), 0, s, labels, w, h, bw, bh, (w + bw - 1) / bw);
}
void launch_sp_ycrcb(const uint8_t *img, size_t step, int channels, uint32_t *ycc, int w, int h, hipStream_t s) {
    hipLaunchKernelGGL(sp_ycrcb_kernel, px_grid(w, h), dim3(256), 0, s, img, step, channels, ycc, w, h);
}
void launch_sp_stats(const SpRelaxArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(sp_stats_kernel, dim3((a.w + kTileW - 1) / kTileW, (a.h + kTileH - 1) / kTileH), dim3(256), 0, s, a);
}
void launch_sp_fold(long long *stats, long long *delta, double *costs, int ld, unsigned ch_mask, hipStream_t s) {
    hipLaunchKernelGGL(sp_fold_kernel, dim3((ld * 8 + 255) / 256), dim3(256), 0, s, stats, delta, costs, ld, ch_mask);
}
void launch_sp_relax(const SpRelaxArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(sp_relax_kernel, dim3((a.w + kTileW - 1) / kTileW, (a.h + kTileH - 1) / kTileH), dim3(256), 0, s, a);
}
void launch_sp_copy(const uint16_t *src, size_t src_step, uint16_t *dst, size_t dst_step, int w, int h, int *max_seen, hipStream_t s) {
    hipLaunchKernelGGL(sp_copy_kernel, px_grid(w, h), dim3(256), 0, s, src, src_step, dst, dst_step, w, h, max_seen);
}
void launch_sp_classify(const SpClassifyArgs &a, hipStream_t s) {
    const int threads = ((a.w + 15) / 16) * a.h;
    hipLaunchKernelGGL(sp_vote_kernel, dim3((threads + 255) / 256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(sp_assign_kernel, px_grid(a.w, a.h), dim3(256), 0, s, a);
}

}  // namespace cart_amd
