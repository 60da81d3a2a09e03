// engine_internal.h -- shared declarations between the C-ABI host code and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "cart_engine.h"

namespace cart_amd {

constexpr int kMaxPaths = 8;
constexpr int kWtaTileX = 64;        // pixels of one row handled by one WTA block
constexpr int kMaxBatchArgs = 128;   // frames per classify launch (params travel as kernel args)
constexpr uint32_t kWtaInvalid = 0xFFFFu;
// Byte order of the 16 disparities of one lane chunk inside a cost slab: byte k of the chunk holds disparity
// chunk_base + kSlabChunkOrder[k] (the aggregation kernel's split-halves register order; the WTA consumes it as is).
constexpr int kSlabChunkOrder[16] = {0, 8, 1, 9, 2, 10, 3, 11, 4, 12, 5, 13, 6, 14, 7, 15};

// Geometry of one engine instance; all buffers below are per workspace slot (= frame).
struct Geometry {
    int w, h, D, P;
    int min_disp, p1, p2;
    int cpitch;          // census row pitch in u32 elements (zero padded left/right)
    int cpadl;           // index of image column 0 inside a census row
    size_t npx;          // w*h
    size_t census_elems; // h*cpitch
    size_t slab_bytes;   // npx*D (one path)
};

// One scan direction inside the fused path-aggregation launch.
struct DirDesc {
    int dx, dy;
    int nlines;   // number of scan lines
    int jmin;     // line index of line 0 (skewed start column, or row for horizontal paths)
    int blk0;     // first block of this direction
    int path;     // slab index (oracle order: down, up, right, left, diagonals)
};

// Where the cost slabs of the frames of ONE launch live: frame f's P path slabs are frame[f] + path * slab_bytes, each [h][w][D].
// The slab workspace is a set of separate device allocations of at most 8 GiB (cart_engine.hip, SlabPool), so the frames of a
// launch need not be one address range; every SGM kernel takes this table by value (wave-uniform index: one scalar load).
constexpr int kMaxLaunchFrames = 64;   // upper bound of CART_OPT_CHUNK_FRAMES
struct SlabTable { uint8_t *frame[kMaxLaunchFrames]; };

struct AggArgs {
    const uint32_t *cen_l, *cen_r;   // slot 0 of the lease
    SlabTable slabs;                 // per frame of the launch: [path][h][w][D]
    Geometry g;
    int ndirs;
    int blocks_per_frame;
    int n_frames;        // filled by launch_aggregate
    int xcd_frames;      // filled by launch_aggregate: decode the grid per XCD (frames x, x + 8, ... on XCD x)
    int hsplit;          // filled by launch_aggregate: horizontal scans run as producer / consumer wave pairs (2 P rows per workgroup)
    DirDesc dirs[kMaxPaths];
};

constexpr int kLaunchFrames = 16;   // frames per launch sequence (cart_engine::chunk_frames) = size of the per-launch frame tables

// Pitched caller image(s) of one launch: frame f starts at ptr + f * frame_stride, or -- for frames that live in separate
// allocations (cart_compute_disparity_multi) -- at frames[f].
struct ImageBatch {
    const uint8_t *ptr;
    size_t step, frame_stride;
    const uint8_t *frames[kLaunchFrames];
    int scattered;
};
inline ImageBatch strided_images(const uint8_t *ptr, size_t step, size_t frame_stride) {
    ImageBatch b{}; b.ptr = ptr; b.step = step; b.frame_stride = frame_stride; return b;
}

struct OutBatch {   // the same for the s16 disparity images a launch writes
    int16_t *ptr;
    size_t step, frame_stride;
    int16_t *frames[kLaunchFrames];
    int scattered;
};
inline OutBatch strided_out(int16_t *ptr, size_t step, size_t frame_stride) {
    OutBatch b{}; b.ptr = ptr; b.step = step; b.frame_stride = frame_stride; return b;
}

// ---- launchers (sgm_kernels.hip) ----
void launch_census(const ImageBatch &left, const ImageBatch &right, int channels, int n_frames,
                   uint8_t *gray_l, uint8_t *gray_r, uint32_t *cen_l, uint32_t *cen_r, uint32_t *right_pk,
                   const Geometry &g, hipStream_t s);
int agg_lines_per_block(int D);  // scan lines per 256-thread block (a pixel is owned by D/16 lanes)
int agg_residency_cap(int ndirs, int D, int n_frames, bool hsplit = false);  // 4-wave aggregation workgroups allowed per CU at a time, 0 = uncapped (measured table at its definition)
bool agg_hsplit(const Geometry &g, int ndirs, int n_frames);  // the launch runs its horizontal scans as producer / consumer wave pairs
void launch_aggregate(const AggArgs &a, int n_frames, hipStream_t s);
// thr = device table of the integer uniqueness threshold for every best cost 0..2047 (launch_uniq_table, built once per engine)
void launch_wta(const SlabTable &slabs, uint16_t *wta_l, uint32_t *right_pk, const Geometry &g, const uint16_t *thr,
                int n_frames, hipStream_t s, bool top2 = false);   // top2: the S5 variant (second-best only), two-kernel WTA only
// WTA fused with the "up" direction (slab kFusedUpPath is never read: the aggregate launch may skip that direction)
constexpr int kFusedUpPath = 1;
size_t wta_fused_partial_elems(const Geometry &g);  // u32 elements of the per-frame right-view partial buffer
void launch_wta_fused(const uint32_t *cen_l, const uint32_t *cen_r, const SlabTable &slabs, uint16_t *wta_l, uint32_t *right_pk,
                      uint32_t *partial, const Geometry &g, const uint16_t *thr, int n_frames, hipStream_t s);
void launch_uniq_table(float u, uint16_t *out_dev, hipStream_t s);   // test access to the integer uniqueness threshold
void uniq_table_host(float u, uint16_t *out);
void launch_post(const uint16_t *wta_l, const uint32_t *right_pk, const uint8_t *gray_l, const OutBatch &out, const Geometry &g, int n_frames, hipStream_t s,
                 int spec = 0);
// post stage + first Jacobi pass of the radius-2 interpolation in one launch (sgm_kernels.hip, post_interp_kernel)
bool post_interp_fusable(int radius, int min_disp16, int max_disp);
void launch_post_interp(const uint16_t *wta_l, const uint32_t *right_pk, const uint8_t *gray_l, const OutBatch &out, const Geometry &g, int n_frames,
                        hipStream_t s, int spec, int min_disp16, int max_disp);   // spec: CART_OPT_SPEC_* bits (1 = S8 zero-disparity-invalid, 2 = S7 replicated border)

// ---- launchers (post_kernels.hip) ----
void launch_interpolate(const int16_t *src, size_t src_step, size_t src_fs, const OutBatch &dst, int w, int h, int radius,
                        int min_disp16, int max_disp, int n_frames, hipStream_t s);
void launch_dir_derivative(const int16_t *disp, size_t step, size_t fs, int16_t *out, size_t ostep, size_t ofs,
                           int32_t *hist512, int w, int h, int n_frames, hipStream_t s);
// Optional per-launch frame table of the plane kernels: with `scattered` the image of frame f is p[f] instead of
// base + f * frame_stride (the *_multi entry points: frames in separate allocations).
struct FrameTable { const void *p[kLaunchFrames]; int scattered; };
void launch_plane_derivative(const int16_t *disp, size_t step, size_t fs, int16_t *out, size_t ostep, size_t ofs,
                             int32_t *hist256, size_t hist_fs, int w, int h, int n_frames, hipStream_t s,
                             const FrameTable *disp_table = nullptr, const FrameTable *out_table = nullptr);
struct ClassifyParams { cart_plane_params p[kMaxBatchArgs]; };
void launch_classify(const int16_t *deriv, size_t step, size_t fs, const ClassifyParams &params, int per_frame,
                     uint8_t *planes, size_t pstep, size_t pfs, int w, int h, int n_frames, hipStream_t s,
                     const FrameTable *deriv_table = nullptr, const FrameTable *planes_table = nullptr);
// stat / seg / table non-null: ids + count + component table in four launches (stat = [n_frames][npx][5] scratch, all zero between calls;
// seg = [n_frames][h][tile columns] roots per row segment); null: ids + count in three
void launch_ccl(const uint8_t *planes, size_t pstep, size_t pfs, int32_t *work, int32_t *ids, size_t istep, size_t ifs,
                int32_t *ncomp, int w, int h, int n_frames, hipStream_t s, int32_t *stat = nullptr, int32_t *seg = nullptr,
                cart_component *table = nullptr, int max_components = 0);
// component table (S12) of a given id map
void launch_ccl_stats(const uint8_t *planes, size_t pstep, size_t pfs, const int32_t *ids, size_t istep, size_t ifs, int32_t *stat, int32_t *seg,
                      cart_component *table, int max_components, int32_t *ncomp, int w, int h, int n_frames, hipStream_t s);
size_t ccl_stats_ws_ints(int w, int h);   // int32 elements of the table workspace per slot: scratch + segment counts

void launch_classify_dev(const int16_t *deriv, size_t step, size_t fs, const cart_plane_params *params_dev, int params_stride,
                         uint8_t *planes, size_t pstep, size_t pfs, int w, int h, int n_frames, hipStream_t s);
struct ScheduleState {   // device resident
    int32_t cum[256];
    cart_plane_params params;
};
void launch_plane_schedule(ScheduleState *state, int provider, int first_id, int n_frames, int update_interval, int reset_interval,
                           const int32_t *hists, cart_plane_params *params_out, hipStream_t s);

struct TemporalArgs {
    int n_prev;
    const uint8_t *prev[CART_MAX_TEMPORAL];
    size_t prev_step[CART_MAX_TEMPORAL];
    const int16_t *flow[CART_MAX_TEMPORAL];
    size_t flow_step[CART_MAX_TEMPORAL];
};
void launch_temporal_vote(const uint8_t *planes, size_t pstep, const TemporalArgs &t, uint8_t *smoothed, size_t sstep, int w, int h, hipStream_t s);
struct QMatrix { float q[16]; };
void launch_reproject(const int16_t *disp, size_t step, size_t fs, const QMatrix &Q, float *xyz, size_t ostep, size_t ofs, int w, int h,
                      int n_frames, hipStream_t s);

// ---- superpixels (superpixel_kernels.hip) ----
constexpr int kSpChannels = 7;    // 0 x, 1 y | 2,3 disparity-derivative ch0,ch1 | 4,5,6 Y,Cr,Cb
constexpr int kSpStatRows = 15;   // 0 pixel count | 1..7 channel sums | 8..14 channel sums of squares
constexpr int kSpMaxLabels = 16384;  // the reference reserves 1 << 14 as its out-of-image marker (contourrelaxation.cu:21)
struct SpRelaxArgs {
    const uint16_t *cur;     // tight [h][w]
    uint16_t *next;
    const uint32_t *ycc;     // tight [h][w], Y | Cr << 8 | Cb << 16
    const int16_t *deriv;    // caller's 2-channel derivative image (NULL when the disparity feature is off)
    size_t deriv_step;       // bytes
    long long *stats;        // [kSpStatRows][ld]
    const double *costs;     // [kSpChannels][ld]
    long long *delta;        // [kSpStatRows][ld]
    int ld;                  // max_label_id + 1
    int w, h;
    unsigned ch_mask;        // bit ch = channel takes part
    double direct, diagonal, w_comp, prog, w_img, w_disp;
};
void launch_sp_block_init(uint16_t *labels, int w, int h, int bw, int bh, hipStream_t s);
void launch_sp_ycrcb(const uint8_t *img, size_t step, int channels, uint32_t *ycc, int w, int h, hipStream_t s);
void launch_sp_stats(const SpRelaxArgs &a, hipStream_t s);
void launch_sp_fold(long long *stats, long long *delta, double *costs, int ld, unsigned ch_mask, hipStream_t s);
void launch_sp_relax(const SpRelaxArgs &a, hipStream_t s);
void launch_sp_copy(const uint16_t *src, size_t src_step, uint16_t *dst, size_t dst_step, int w, int h, int *max_seen, hipStream_t s);
struct SpClassifyArgs {
    const int16_t *deriv; size_t deriv_step;      // 2-channel, bytes
    const uint16_t *labels; size_t labels_step;   // bytes
    int w, h, max_label;
    cart_plane_params p;
    TemporalArgs t;
    uint8_t *unsmoothed; size_t unsmoothed_step;
    uint8_t *planes; size_t planes_step;
    unsigned *votes;                              // [max_label][3], zeroed by the caller
};
void launch_sp_classify(const SpClassifyArgs &a, hipStream_t s);

// ---- optical flow (flow_kernels.hip) ----
void launch_block_flow(const uint32_t *cen_cur, const uint32_t *cen_prev, const Geometry &g, int radius, int block, int16_t *flow,
                       size_t flow_step, hipStream_t s);

void launch_resize_linear(const uint8_t *src, size_t sstep, int sw, int sh, int channels, uint8_t *dst, size_t dstep, int dw, int dh, hipStream_t s);
void launch_narrow_copy(const void *src, void *dst, size_t bytes, int blocks, hipStream_t s);
int kernel_count();

}  // namespace cart_amd
