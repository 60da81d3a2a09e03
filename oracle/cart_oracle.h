/*
 * cart_oracle.h -- CPU restatement of CART-SLAM's dense-stereo hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing outside tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may include, link or call this library; the
 * product path (cart-slam_amd/) never routes through it.
 *
 * PARITY UNPINNED (tools/ref_pin/ is the kit that pins it on a machine with OpenCV-CUDA; tests/test_ref_pin.py
 * consumes its outputs).  The reference (LorgeN/CART-SLAM) ships no tests, golden
 * vectors or fixtures for this path, cannot be built here (CUDA + OpenCV-CUDA
 * + Boost + log4cxx, none installed) and delegates the SGM core to the
 * un-vendored, un-pinned third-party module opencv_contrib `cudastereo`
 * (cv::cuda::StereoSGM; call sites include/modules/disparity.hpp:31-33 and
 * src/modules/disparity/disparity.cu:66-71, `find_package(OpenCV REQUIRED)`
 * CMakeLists.txt:19).  This file therefore restates
 *   (a) the published SGM / libSGM algorithm as used by that module, with every
 *       choice the upstream text leaves open written down below as THE spec,
 *   (b) the reference's own post-SGM kernels under clean "intent" semantics
 *       (no tile row-shift, no in-place races).
 *
 * SPEC DECISIONS (each is a build-owned definition):
 *  S1 gray      = (1868*B + 9617*G + 4899*R + 8192) >> 14      (OpenCV 8-bit BGR2GRAY)
 *  S2 census    = 9x7 symmetric, 31 bits: for dy=-3..-1, dx=-4..4 then dy=0,
 *                 dx=-4..-1 append bit I(y+dy,x+dx) > I(y-dy,x-dx) (MSB first).
 *                 Pixels within 4 columns / 3 rows of the border get feature 0.
 *  S3 cost      C(x,y,d) = popcount(cenL(x,y) ^ cenR(x-d-min_disp,y)); the right
 *                 feature is 0 when x-d-min_disp is outside [0,W).
 *  S4 path      L_r(p,d) = C(p,d) + min(L_r(q,d), L_r(q,d-1)+P1, L_r(q,d+1)+P1,
 *                 m+P2) - m, q = p-r, m = min_k L_r(q,k); a path starts at the
 *                 first in-image pixel with the recurrence state all-zero
 *                 (=> L = C there).  d-1 / d+1 outside [0,D) do not take part.
 *                 Stored as u8 (values <= 31+P2; P2 <= 224 required).
 *                 4 paths = {down, up, right, left}; 8 paths add the 4 diagonals.
 *  S5 WTA left  S = sum_r L_r; best = min over d of (S<<16 | d)  (ties -> lowest d);
 *                 unique iff for every d: (float)S[d]*u >= (float)S[best] or
 *                 |d-best| <= 1, u = (float)(100-uniqueness_ratio)/100.0f;
 *                 not unique -> 0xFFFF.  Sub-pixel (x16): if 0<best<D-1,
 *                 num = S[b-1]-S[b+1], den = S[b-1]-2S[b]+S[b+1],
 *                 disp = 16*b + (den ? (16*num+den)/(2*den) : 0)  (C truncation).
 *                 NOTE -- departs from the wording of SURVEY.md 8a-4(4), which describes the
 *                 test from upstream memory as "top-2 tracking: keep iff cost2*u >= cost1 or
 *                 |d2-d1| <= 1" (only the SECOND-best cost is examined).  The two differ
 *                 when the second-best cost sits next to the best (|d2-d1| <= 1, so top-2
 *                 keeps the pixel) while a third, non-adjacent disparity also fails the
 *                 ratio: S5 then rejects the pixel, top-2 keeps it.  S5 is the form of the
 *                 published libSGM winner-takes-all kernel that cv::cuda::StereoSGM is
 *                 built from (every lane evaluates "cost*u >= best || |d-best| <= 1" and
 *                 the results are AND-reduced over the whole disparity range); the survey
 *                 text is a paraphrase, marked [EXTERNAL-UNVERIFIED] there.  Neither form
 *                 can be checked against the reference here (opencv_contrib is absent, no
 *                 fixtures): parity unpinned.  Should a reference-side fixture ever show
 *                 the top-2 form: it exists as CART_ORACLE_VARIANT_S5_TOP2 here and as
 *                 CART_OPT_SPEC_S5_TOP2 on the engine (one pass over d tracking best and second-best
 *                 (cost, d), replaced on a strictly smaller cost only = the two smallest (cost<<16 | d) keys).
 *  S6 WTA right R(p) = argmin_d S(p+d, d) over d with p+d < W, ties -> lowest d,
 *                 integer disparity (not x16), never invalid.
 *  S7 median    3x3 on both maps as u16 (0xFFFF sorts highest); the one-pixel
 *                 image border passes through unfiltered.
 *                 NOTE -- a build-owned choice [EXTERNAL-UNVERIFIED]: what the upstream median kernel leaves in the
 *                 border row / column (pass-through, a replicated-border median, or untouched memory) cannot be
 *                 checked here.  A replicated-border median changes 66-100 output pixels on the committed goldens and
 *                 836-844 of 465 750 at 1242x375 (tests/spec_variants.py).  One place to change: the first `if` of
 *                 cart_oracle_median3x3_u16 (and left_median_at / right_median_at in sgm_kernels.hip).
 *  S8 LR check  left pixel -> 0xFFFF if gray_left==0, or already 0xFFFF, or
 *                 k = x-(dL>>4) in [0,W) and |R(k)-(dL>>4)| > 1.
 *                 NOTE -- probable point of departure [EXTERNAL-UNVERIFIED]: the check_consistency kernel of older libSGM
 *                 releases, from which opencv_contrib's cudastereo was ported, tests
 *                 `mask == 0 || d <= 0 || (k in range && |R(k) - d| > 1)` on the INTEGER disparity d = dL >> 4, which also
 *                 invalidates every valid winner at disparity index 0; later libSGM releases test
 *                 `org == INVALID_DISP` instead, which is the form above.  Which of the two cv::cuda::StereoSGM runs depends
 *                 on the OpenCV version the reference was built with (un-versioned: CMakeLists.txt:19).  The `d <= 0` form
 *                 would invalidate 558 / 789 / 1187 / 592 more pixels on the four committed goldens and 2225-2686 of
 *                 465 750 on the 1242x375 scenes (tests/spec_variants.py).  One line to change in each place:
 *                 cart_oracle_lr_check_range (`gray_left[i] == 0 || org == CART_ORACLE_WTA_INVALID` -> add
 *                 `|| (org >> 4) == 0`) and post_kernel's `bool invalid =` line in sgm_kernels.hip.  tools/ref_pin produces the
 *                 reference outputs that decide it; tests/test_ref_pin.py names this variant when they disagree.  Both forms exist:
 *                 CART_ORACLE_VARIANT_S8_ZERO_INVALID here, CART_OPT_SPEC_S8_ZERO_INVALID on the engine (likewise S7).
 *  S9 range     0xFFFF -> (min_disp-1)*16, else += min_disp*16; stored s16.
 *  S10 post stages: Jacobi reads of the unmodified input, out-of-image samples
 *                 skipped by window means and making a difference INVALID.
 *  S11 findPeaks: index sort is by descending value, ties by ascending index
 *                 (the reference's std::sort leaves tie order unspecified);
 *                 peaks sorted by descending persistence, ties by birth order.
 *  S12 CCL      (no reference counterpart) 4-connected components over the plane
 *                 label map for labels {0,1}; component id = smallest linear index
 *                 y*W+x in the component; label-2 (UNKNOWN) pixels get -1.
 *                 Component table: one entry {id, label, area, x0, y0, x1, y1} (inclusive bounding
 *                 box) per component, ordered by ascending id.
 *  S13 superpixels (contour relaxation; reference: src/modules/superpixels.cu and superpixels/contourrelaxation/): the reference's
 *                 result depends on racy featureCost refreshes, nvcc FMA contraction and CUDA's log(); the spec is the
 *                 race-free intent: per iteration EVERY pixel picks, from the unique labels of its in-image 3x3
 *                 neighbourhood (visited dx-major: (-1,-1),(-1,0),(-1,1),(0,-1),...,(1,1)), the first one of minimal cost,
 *                 all costs evaluated on the label image and label statistics of the iteration start (Jacobi); then all
 *                 changes are applied and the statistics (exact integer sums) and per-label feature costs refreshed.
 *                 Statistics cover the whole image (the reference's floor()ed init grid, contourrelaxation.cu:371, drops
 *                 the right/bottom strips).  All arithmetic is IEEE double, no FMA contraction, operations in the
 *                 reference's source order; log() is cart_oracle_log below (a fixed sequence of IEEE operations, so that
 *                 CPU and GPU agree bit for bit).
 *  S14 YCrCb    = OpenCV 8-bit BGR2YCrCb: Y as S1; Cr = ((R-Y)*11682 + (128<<14) + 8192) >> 14;
 *                 Cb = ((B-Y)*9241 + (128<<14) + 8192) >> 14, arithmetic shift, saturated to u8; stored (Y,Cr,Cb).
 *  S16 resize     (KITTIDataSource with an image size other than the files', src/sources/kitti.cpp:169-172:
 *                 cv::cuda::resize(..., INTER_LINEAR) [EXTERNAL-UNVERIFIED: restated from the published
 *                 opencv cudawarping resize_linear kernel]): no half-pixel offset, src = dst * (src_size /
 *                 dst_size) in float (the ratio rounded once from double), x1 = floor, x2 = x1 + 1 (read
 *                 clamped to the last column / row), out = sum over the four taps in the order (y1,x1), (y1,x2),
 *                 (y2,x1), (y2,x2) of tap * (wx * wy) accumulated in float without contraction, rounded to
 *                 nearest even and saturated to u8, per channel.
 *  S15 optical flow (stand-in provider of "optflow"; the reference's is NVIDIA fixed-function hardware,
 *                 src/modules/optflow.cpp:57-70, so there is nothing to restate): census block matching, integer pixels.
 *                 For the current pixel p and a displacement (u,v), |u|,|v| <= R:
 *                   cost(p,u,v) = sum over q in the (2B+1)x(2B+1) window around p, q inside the image, of
 *                                 popcount(cenC(q) ^ cenP(q - (u,v))),  cenP = 0 outside the image (like S3);
 *                 cenC / cenP = S2 features of the current / previous gray image.  The winner starts as (0,0) and is
 *                 replaced only by a strictly smaller cost, candidates visited v = -R..R outer, u = -R..R inner.
 *                 flow = (32u, 32v) as S10.5 (include/modules/optflow.hpp:16): the previous position of p is
 *                 p - (flow >> 5), which is how planeseg.cu:212-219 consumes it.
 */
#ifndef CART_ORACLE_H
#define CART_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CART_ORACLE_INVALID (-32768)      /* include/modules/disparity.hpp:17 */
#define CART_ORACLE_WTA_INVALID 0xFFFFu

typedef struct {
    int width, height;
    int min_disparity;      /* cartconfig.cpp:147 default 4 */
    int num_disparities;    /* 64 | 128 | 256 */
    int paths;              /* 4 | 8 */
    int p1, p2;             /* 10, 120 */
    int uniqueness_ratio;   /* disparity.hpp:32 -> 12 */
} cart_oracle_sgm_params;

/* include/modules/planeseg.hpp:25-34 */
typedef struct {
    int horizontal_min, horizontal_max;
    int vertical_min, vertical_max;
    int horizontal_center, vertical_center;
} cart_oracle_plane_params;

/* S1; src pitched BGR (src_step bytes/row), dst tight w*h. disparity.cu:66-67 */
void cart_oracle_bgr2gray(const uint8_t *bgr, size_t src_step, int w, int h, uint8_t *gray);

/* S2; gray tight, census tight u32. */
void cart_oracle_census9x7(const uint8_t *gray, int w, int h, uint32_t *census);

/* S3+S4 for one direction (dx,dy in {-1,0,1}); L is [h][w][D] u8. */
void cart_oracle_aggregate_path(const uint32_t *cen_l, const uint32_t *cen_r, int w, int h, int D,
                                int min_disp, int p1, int p2, int dx, int dy, uint8_t *L);

/* direction table used for `paths` = 4 or 8: index -> (dx,dy). */
void cart_oracle_path_dir(int index, int *dx, int *dy);

/* S5+S6 on a summed volume S [h][w][D] u16. */
void cart_oracle_wta(const uint16_t *S, int w, int h, int D, int uniqueness_ratio,
                     uint16_t *left, uint16_t *right);
void cart_oracle_wta_ex(const uint16_t *S, int w, int h, int D, int uniqueness_ratio,
                        uint16_t *left, uint16_t *right, int variants);   /* variants: CART_ORACLE_VARIANT_S5_TOP2 */

/* The three choices that are open upstream (NOTEs at S5 / S7 / S8 above), selectable so that tests can hold BOTH forms
 * against the engine (cart_engine_set_option CART_OPT_SPEC_*) until tools/ref_pin decides; 0 = the spec as written. */
#define CART_ORACLE_VARIANT_S8_ZERO_INVALID 1      /* LR check also invalidates integer disparity 0 (`d <= 0`) */
#define CART_ORACLE_VARIANT_S7_REPLICATE_BORDER 2  /* medians read a replicated border instead of passing it through */
#define CART_ORACLE_VARIANT_S5_TOP2 4              /* uniqueness tests the second-best cost only (the wording of SURVEY.md 8a-4(4)) */

/* S7 */
void cart_oracle_median3x3_u16(const uint16_t *src, int w, int h, uint16_t *dst);
void cart_oracle_median3x3_u16_ex(const uint16_t *src, int w, int h, uint16_t *dst, int variants);

/* S8+S9: left_med/right_med are the median-filtered WTA maps. */
void cart_oracle_lr_check_range(const uint16_t *left_med, const uint16_t *right_med, const uint8_t *gray_left,
                                int w, int h, int min_disp, int16_t *out);
void cart_oracle_lr_check_range_ex(const uint16_t *left_med, const uint16_t *right_med, const uint8_t *gray_left,
                                   int w, int h, int min_disp, int16_t *out, int variants);

/* Whole SGM core (a-4): gray L/R tight -> s16 disparity x16.  If S_out is
 * non-NULL it receives the summed volume [h][w][D] u16. Returns 0 / -1. */
int cart_oracle_sgm(const cart_oracle_sgm_params *p, const uint8_t *gray_l, const uint8_t *gray_r,
                    int16_t *disp, uint16_t *S_out);
int cart_oracle_sgm_ex(const cart_oracle_sgm_params *p, const uint8_t *gray_l, const uint8_t *gray_r,
                       int16_t *disp, uint16_t *S_out, int variants);

/* a-5 interpolation.cu:17-82 under S10; min_disp16 = cfg*16, max_disp = image width
 * (disparity.hpp:27-28 quirk). in/out tight s16; out may not alias in. */
void cart_oracle_interpolate(const int16_t *in, int w, int h, int radius, int iterations,
                             int min_disp16, int max_disp, int16_t *out);

/* a-7 derivative.cu:27-116: out is [h][w][2] (ch0 vertical, ch1 horizontal),
 * hist is [256][2] (interleaved like CV_32SC2 1x256), overwritten. */
void cart_oracle_directional_derivative(const int16_t *disp, int w, int h, int16_t *out, int32_t *hist);

/* a-8 planeseg.cu:31-158: out tight s16, hist256 is ADDED to (persistent). */
void cart_oracle_plane_derivative(const int16_t *disp, int w, int h, int16_t *out, int32_t *hist256);

/* a-9 peaks.cpp:12-72 (S11): returns number of peaks; born/died/left/right
 * arrays must hold n entries each, sorted by persistence. */
int cart_oracle_find_peaks(const int32_t *data, int n, int *born, int *died, int *left, int *right);

/* a-9 planeseg.cu:405-458: updates *params in place; returns 1 if updated, 0 on
 * the reference's early-outs (previous parameters kept). */
int cart_oracle_histogram_peak_params(const int32_t *hist256, cart_oracle_plane_params *params);

/* a-10 planeseg.cu:160-198 (non-temporal). */
void cart_oracle_classify(const int16_t *deriv, int w, int h, const cart_oracle_plane_params *params, uint8_t *planes);

/* a-10 temporal branch, planeseg.cu:199-240: votes[plane]++; walk back k = 0..n_prev-1: the flow (S10.5, 2 x s16
 * interleaved, flows[k]) is read at the ORIGINAL pixel (as the reference does, :212-213), >>5, subtracted from the
 * running position; inside the image -> votes[prev_planes[k][y][x]]++ (outside: skipped, position kept).
 * out = votes[H] > votes[V] ? H : V, UNKNOWN if that count is 0 (:235-238). */
void cart_oracle_temporal_vote(const uint8_t *planes, int w, int h, int n_prev, const uint8_t *const *prev_planes,
                               const int16_t *const *flows, uint8_t *smoothed);

/* SURVEY 8f-2, src/modules/depth.cpp:9-25: disp/16 -> float, cv::cuda::reprojectImageTo3D(Q) (no missing-value
 * handling): [X Y Z W]^T = Q [x y d 1]^T, out = (X/W, Y/W, Z/W) as float [h][w][3].  Float, compare within 1e-4. */
void cart_oracle_reproject_depth(const int16_t *disp, int w, int h, const float Q[16], float *xyz);

/* S16: tight u8 [sh][sw][channels] -> tight [dh][dw][channels] */
void cart_oracle_resize_linear(const uint8_t *src, int sw, int sh, int channels, uint8_t *dst, int dw, int dh);

/* a-11 (S12). Returns the number of components. */
int cart_oracle_ccl(const uint8_t *planes, int w, int h, int32_t *ids);

/* a-11 (S12) component table from the label map and its ids: 7 int32 per component {id, label, area, x0, y0, x1, y1},
 * ascending id; writes at most max_components entries, returns the number of components. */
int cart_oracle_ccl_stats(const uint8_t *planes, const int32_t *ids, int w, int h, int32_t *table, int max_components);

/* ---- superpixels + superpixel plane labelling ------------------------------------------------- */

/* cartconfig.cpp:121-133 / superpixels.hpp:16-27 */
typedef struct {
    double direct_clique_cost, diagonal_clique_cost;
    double compactness_weight, progressive_compactness_cost;
    double image_weight, disparity_weight;
} cart_oracle_sp_params;

/* S14; bgr pitched (3 B/px), out tight [h][w][3] = (Y,Cr,Cb). superpixels.cu:81 */
void cart_oracle_bgr2ycrcb(const uint8_t *bgr, size_t src_step, int w, int h, uint8_t *ycrcb);

/* initialization.cu:13-58: label = (y/bh)*ceil(w/bw) + x/bw; returns maxLabelId = ceil(w/bw)*ceil(h/bh). */
int cart_oracle_sp_block_init(int w, int h, int block_w, int block_h, uint16_t *labels);

/* S13 log for x > 0, finite, normal: x = m*2^e, m in (sqrt(.5), sqrt(2)] (m > 0x1.6a09e667f3bcdp+0 is halved),
 * s = (m-1)/(m+1), z = s*s, p = Horner over z of c_k = 1.0/(2k+1), k = 11..0, r = (2*s)*p,
 * log = e*0x1.62e42fee00000p-1 + (r + e*0x1.a39ef35793c76p-33). */
double cart_oracle_log(double x);

/* S13: contourrelaxation.cu:248-294 (performRelaxation), :296-322 (updateLabels), features/gaussian.cu,
 * features/compactness.cu.  labels [h][w] u16 in/out (all < max_label_id < 16384); ycrcb tight [h][w][3];
 * deriv2 tight [h][w][2] s16 (ch0 vertical, ch1 horizontal; may be NULL iff disparity_weight <= 0).
 * Returns the total number of label changes, or -1 on bad arguments. */
long cart_oracle_sp_relax(const cart_oracle_sp_params *p, uint16_t *labels, int w, int h, int max_label_id,
                          const uint8_t *ycrcb, const int16_t *deriv2, int iterations);

/* sp_planeseg.cu:27-128 (per-pixel classification of deriv ch0, optional temporal vote with weight 2 for the current
 * frame, u16 per-label vote counts) + :130-178 (per-label majority, UNKNOWN wins ties, HORIZONTAL needs > max(U,V)).
 * planes_unsmoothed = the per-pixel classification BEFORE the temporal vote (what the kernel stores, :75);
 * planes = the per-superpixel result. */
void cart_oracle_sp_classify(const int16_t *deriv2, const uint16_t *labels, int w, int h, int max_label,
                             const cart_oracle_plane_params *params, int n_prev, const uint8_t *const *prev_planes,
                             const int16_t *const *flows, uint8_t *planes_unsmoothed, uint8_t *planes);

/* S15: census planes tight [h][w] u32 of the current and the previous frame -> flow tight [h][w][2] s16 (S10.5). */
void cart_oracle_block_flow(const uint32_t *cen_cur, const uint32_t *cen_prev, int w, int h, int radius, int block,
                            int16_t *flow);

#ifdef __cplusplus
}
#endif
#endif
