// post_kernels.hip -- gfx950 kernels for the stages after the SGM core.
//
// Each kernel restates one of the reference's own CUDA kernels under the clean "intent"
// semantics fixed in SURVEY.md 8c (Jacobi reads of the unmodified input, no tile row-shift,
// out-of-image samples skipped by window means / making a difference INVALID):
//   interpolate_kernel        <- interpolateKernel        src/modules/disparity/interpolation.cu:17-82
//   dir_derivative_kernel     <- calculateDirectionalDerivatives + mergeDerivativeHistograms
//                                                           src/modules/disparity/derivative.cu:27-116
//   plane_derivative_kernel   <- calculateDerivatives + mergeHistogram   src/modules/planeseg/planeseg.cu:31-158
//   classify_kernel           <- classifyPlanes (non-temporal)           src/modules/planeseg/planeseg.cu:160-198
//   ccl_*                     <- new stage (no reference counterpart), spec S12 in oracle/cart_oracle.h
// Images are small (<= 4 MB) and L2 resident; the kernels are one-pass, coalesced along x, with
// block-local LDS histograms flushed by one global atomicAdd per bin and block.
#include <type_traits>

#include "engine_internal.h"

namespace cart_amd {

constexpr int INVALID = CART_DISPARITY_INVALID;

template <typename T>
__device__ __forceinline__ T *row_ptr(T *base, size_t frame_stride, size_t step, int frame, int y) {
    typedef typename std::conditional<std::is_const<T>::value, const uint8_t, uint8_t>::type B;
    return reinterpret_cast<T *>(reinterpret_cast<B *>(base) + (size_t)frame * frame_stride + (size_t)y * step);
}

__device__ __forceinline__ int16_t *out_row(const OutBatch &o, int frame, int y) {
    uint8_t *base = o.scattered ? reinterpret_cast<uint8_t *>(o.frames[frame]) : reinterpret_cast<uint8_t *>(o.ptr) + (size_t)frame * o.frame_stride;
    return reinterpret_cast<int16_t *>(base + (size_t)y * o.step);
}

// ------------------------------------------------------------------ interpolate (one Jacobi pass)
__global__ __launch_bounds__(256) void interpolate_kernel(const int16_t *src, size_t src_step, size_t src_fs, OutBatch dst, int w, int h,
                                                          int radius, int min_disp16, int max_disp) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, frame = blockIdx.z;
    if (x >= w || y >= h) return;
    int sum = 0, count = 0;
    for (int l = -radius + 1; l < radius; ++l) {
        const int yy = y + l;
        if (yy < 0 || yy >= h) continue;
        const int16_t *row = row_ptr(src, src_fs, src_step, frame, yy);
        for (int k = -radius + 1; k < radius; ++k) {
            const int xx = x + k;
            if (xx < 0 || xx >= w) continue;
            const int v = row[xx];
            if (v > min_disp16 && v < max_disp) { sum += v; ++count; }
        }
    }
    const unsigned min_count = (unsigned)(radius * radius + 1);  // interpolation.cu:33
    out_row(dst, frame, y)[x] = ((unsigned)count > min_count) ? (int16_t)(sum / count) : (int16_t)INVALID;
}

// radius 2 (3x3 window, the configured smoothing of the reference's KITTI setups): four adjacent pixels per thread share
// their six window columns -- per column the sum and count of the valid values of the three rows, then three columns per
// output.  sum / count (count <= 9, |sum| < 2^24) is a float division truncated, which is exact here.
__global__ __launch_bounds__(256) void interpolate_r2_kernel(const int16_t *src, size_t src_step, size_t src_fs, OutBatch dst, int w, int h,
                                                             int min_disp16, int max_disp) {
    const int xb = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y, frame = blockIdx.z;
    if (xb >= w || y >= h) return;
    // vector accesses need 4-byte aligned source rows / 8-byte aligned destination rows (the engine's own buffers are)
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | src_step | src_fs) & 3) == 0;
    const bool vec_out = ((reinterpret_cast<uintptr_t>(out_row(dst, frame, 0)) | dst.step) & 7) == 0;
    int csum[6], ccnt[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { csum[c] = 0; ccnt[c] = 0; }
#pragma unroll
    for (int l = -1; l <= 1; ++l) {
        const int yy = y + l;
        if (yy < 0 || yy >= h) continue;
        const int16_t *row = row_ptr(src, src_fs, src_step, frame, yy);
        int v[6];
        if (vec && xb >= 2 && xb + 6 <= w) {
            // pixels xb-2 .. xb+5 as one 16-byte load (4-byte aligned: rows and xb are), the window is its middle six
            typedef uint32_t v4u_a4 __attribute__((ext_vector_type(4), aligned(4)));
            const v4u_a4 q = *reinterpret_cast<const v4u_a4 *>(row + xb - 2);
            v[0] = (int)q.x >> 16; v[1] = (int16_t)q.y; v[2] = (int)q.y >> 16; v[3] = (int16_t)q.z; v[4] = (int)q.z >> 16; v[5] = (int16_t)q.w;
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int xx = xb - 1 + c;
                v[c] = (xx >= 0 && xx < w) ? (int)row[xx] : (int)0x80000000;   // never passes the range test below (min_disp16 > -2^20)
            }
        }
#pragma unroll
        for (int c = 0; c < 6; ++c)
            if (v[c] > min_disp16 && v[c] < max_disp) { csum[c] += v[c]; ++ccnt[c]; }
    }
    int16_t *orow = out_row(dst, frame, y);
    int16_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int sum = csum[i] + csum[i + 1] + csum[i + 2], count = ccnt[i] + ccnt[i + 1] + ccnt[i + 2];
        // interpolation.cu:33: count > r*r + 1 = 5
        o[i] = count > 5 ? (int16_t)(int)((float)sum / (float)count) : (int16_t)INVALID;
    }
    if (vec_out && xb + 4 <= w) {
        *reinterpret_cast<uint2 *>(orow + xb) = make_uint2((uint16_t)o[0] | ((uint32_t)(uint16_t)o[1] << 16), (uint16_t)o[2] | ((uint32_t)(uint16_t)o[3] << 16));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (xb + i < w) orow[xb + i] = o[i];
    }
}

void launch_interpolate(const int16_t *src, size_t src_step, size_t src_fs, const OutBatch &dst, int w, int h, int radius,
                        int min_disp16, int max_disp, int n_frames, hipStream_t s) {
    if (radius == 2 && min_disp16 > -(1 << 20) && max_disp < (1 << 20)) {
        dim3 grid((w + 255) / 256, (h + 3) / 4, n_frames), block(64, 4);
        hipLaunchKernelGGL(interpolate_r2_kernel, grid, block, 0, s, src, src_step, src_fs, dst, w, h, min_disp16, max_disp);
        return;
    }
    dim3 grid((w + 63) / 64, (h + 3) / 4, n_frames), block(64, 4);
    hipLaunchKernelGGL(interpolate_kernel, grid, block, 0, s, src, src_step, src_fs, dst, w, h, radius, min_disp16, max_disp);
}

// ------------------------------------------------------------------ directional derivatives + 2x256 histogram
__global__ __launch_bounds__(256) void dir_derivative_kernel(const int16_t *disp, size_t step, size_t fs, int16_t *out,
                                                             size_t ostep, size_t ofs, int32_t *hist512, int w, int h) {
    __shared__ int lh[512];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    lh[tid] = 0; lh[tid + 256] = 0;
    __syncthreads();
    const int x = blockIdx.x * 64 + threadIdx.x, frame = blockIdx.z;
    for (int r = 0; r < 4; ++r) {
        const int y = blockIdx.y * 16 + threadIdx.y * 4 + r;
        if (x >= w || y >= h) continue;
        int dv = INVALID, dh = INVALID;
        if (y - 2 >= 0 && y + 2 < h) {
            const int a = row_ptr(disp, fs, step, frame, y + 2)[x], b = row_ptr(disp, fs, step, frame, y - 2)[x];
            if (a != INVALID && b != INVALID) {
                dv = (int16_t)(a - b);
                if (dv >= -128 && dv <= 127) atomicAdd(&lh[2 * (dv + 128)], 1);
            }
        }
        if (x - 2 >= 0 && x + 2 < w) {
            const int16_t *row = row_ptr(disp, fs, step, frame, y);
            const int a = row[x + 2], b = row[x - 2];
            if (a != INVALID && b != INVALID) {
                dh = (int16_t)(a - b);
                if (dh >= -128 && dh <= 127) atomicAdd(&lh[2 * (dh + 128) + 1], 1);
            }
        }
        int16_t *orow = row_ptr(out, ofs, ostep, frame, y);
        *reinterpret_cast<short2 *>(orow + 2 * x) = make_short2((short)dv, (short)dh);
    }
    __syncthreads();
    int32_t *gh = hist512 + (size_t)frame * 512;
    if (lh[tid]) atomicAdd(&gh[tid], lh[tid]);
    if (lh[tid + 256]) atomicAdd(&gh[tid + 256], lh[tid + 256]);
}

void launch_dir_derivative(const int16_t *disp, size_t step, size_t fs, int16_t *out, size_t ostep, size_t ofs,
                           int32_t *hist512, int w, int h, int n_frames, hipStream_t s) {
    (void)hipMemsetAsync(hist512, 0, sizeof(int32_t) * 512 * (size_t)n_frames, s);  // fresh histogram per frame (derivative.cu:169)
    dim3 grid((w + 63) / 64, (h + 15) / 16, n_frames), block(64, 4);
    hipLaunchKernelGGL(dir_derivative_kernel, grid, block, 0, s, disp, step, fs, out, ostep, ofs, hist512, w, h);
}

// ------------------------------------------------------------------ plane derivative (5-tap vertical mean, 1-px diff)
// A thread owns one column of PD_ROWS output rows: it loads the PD_ROWS + 6 disparities its outputs depend on once and slides
// the 5-tap sums over them.  The low-pass accumulator of the reference is the 16-bit derivative_t (planeseg.cu:62): summing
// in int and truncating once gives the same value because wrapping addition is associative.
// The block histogram is kept in 16 copies, one per lane%16 and skewed by a bank each: disparity slopes cluster on two or
// three values per wave, which on a single copy serialises the LDS atomics 64 deep.
constexpr int PD_ROWS = 8, PD_COPIES = 16, PD_PITCH = 257;

__global__ __launch_bounds__(256) void plane_derivative_kernel(const int16_t *disp, size_t step, size_t fs, int16_t *out,
                                                               size_t ostep, size_t ofs, int32_t *hist256, size_t hist_fs,
                                                               int w, int h, FrameTable dtab, FrameTable otab) {
    __shared__ int lh[PD_COPIES * PD_PITCH];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    for (int i = tid; i < PD_COPIES * PD_PITCH; i += 256) lh[i] = 0;
    __syncthreads();
    const int x = blockIdx.x * 64 + threadIdx.x, frame = blockIdx.z;
    const int ybase = blockIdx.y * (4 * PD_ROWS) + threadIdx.y * PD_ROWS;
    if (dtab.scattered) { disp = static_cast<const int16_t *>(dtab.p[frame]); fs = 0; }
    if (otab.scattered) { out = static_cast<int16_t *>(const_cast<void *>(otab.p[frame])); ofs = 0; }
    if (x < w && ybase < h) {
        int v[PD_ROWS + 6];   // rows ybase-3 .. ybase+PD_ROWS+2; INVALID where outside the image
#pragma unroll
        for (int j = 0; j < PD_ROWS + 6; ++j) {
            const int yy = ybase - 3 + j;
            v[j] = (yy >= 0 && yy < h) ? (int)row_ptr(disp, fs, step, frame, yy)[x] : INVALID;
        }
        int lp[PD_ROWS + 2];  // low-pass of rows ybase-1 .. ybase+PD_ROWS
#pragma unroll
        for (int j = 0; j < PD_ROWS + 2; ++j) {
            const int y = ybase - 1 + j;
            int sum = 0, count = 0;
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (v[j + k] != INVALID) { sum += v[j + k]; ++count; }
            lp[j] = (y < 0 || y >= h || count == 0) ? INVALID : (int)(int16_t)((int)(int16_t)sum / count);
        }
        int *mine = lh + (threadIdx.x & (PD_COPIES - 1)) * PD_PITCH;
#pragma unroll
        for (int r = 0; r < PD_ROWS; ++r) {
            const int y = ybase + r;
            if (y >= h) break;
            int o = INVALID;
            if (lp[r] != INVALID && lp[r + 1] != INVALID && lp[r + 2] != INVALID) {
                o = (int16_t)(lp[r + 2] - lp[r]);
                if (o >= -128 && o <= 127) atomicAdd(&mine[o + 128], 1);
            }
            row_ptr(out, ofs, ostep, frame, y)[x] = (int16_t)o;
        }
    }
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int c = 0; c < PD_COPIES; ++c) total += lh[c * PD_PITCH + tid];
    if (total) atomicAdd(&hist256[(size_t)frame * hist_fs + tid], total);  // cumulative, planeseg.cu:157
}

void launch_plane_derivative(const int16_t *disp, size_t step, size_t fs, int16_t *out, size_t ostep, size_t ofs,
                             int32_t *hist256, size_t hist_fs, int w, int h, int n_frames, hipStream_t s,
                             const FrameTable *disp_table, const FrameTable *out_table) {
    dim3 grid((w + 63) / 64, (h + 4 * PD_ROWS - 1) / (4 * PD_ROWS), n_frames), block(64, 4);
    const FrameTable none{};
    hipLaunchKernelGGL(plane_derivative_kernel, grid, block, 0, s, disp, step, fs, out, ostep, ofs, hist256, hist_fs, w, h,
                       disp_table ? *disp_table : none, out_table ? *out_table : none);
}

// ------------------------------------------------------------------ classify
__global__ __launch_bounds__(256) void classify_kernel(const int16_t *deriv, size_t step, size_t fs, ClassifyParams params,
                                                       int per_frame, uint8_t *planes, size_t pstep, size_t pfs, int w, int h,
                                                       FrameTable dtab, FrameTable ptab) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, frame = blockIdx.z;
    if (x >= w || y >= h) return;
    if (dtab.scattered) { deriv = static_cast<const int16_t *>(dtab.p[frame]); fs = 0; }
    if (ptab.scattered) { planes = static_cast<uint8_t *>(const_cast<void *>(ptab.p[frame])); pfs = 0; }
    const cart_plane_params pp = params.p[per_frame ? frame : 0];
    const int d = row_ptr(deriv, fs, step, frame, y)[x];
    int plane = CART_PLANE_UNKNOWN;
    if (d != INVALID && d >= pp.horizontal_min && d < pp.horizontal_max) plane = CART_PLANE_HORIZONTAL;
    else if (d != INVALID && d >= pp.vertical_min && d < pp.vertical_max) plane = CART_PLANE_VERTICAL;
    row_ptr(planes, pfs, pstep, frame, y)[x] = (uint8_t)plane;
}

void launch_classify(const int16_t *deriv, size_t step, size_t fs, const ClassifyParams &params, int per_frame,
                     uint8_t *planes, size_t pstep, size_t pfs, int w, int h, int n_frames, hipStream_t s,
                     const FrameTable *deriv_table, const FrameTable *planes_table) {
    dim3 grid((w + 63) / 64, (h + 3) / 4, n_frames), block(64, 4);
    const FrameTable none{};
    hipLaunchKernelGGL(classify_kernel, grid, block, 0, s, deriv, step, fs, params, per_frame, planes, pstep, pfs, w, h,
                       deriv_table ? *deriv_table : none, planes_table ? *planes_table : none);
}

__global__ __launch_bounds__(256) void classify_dev_kernel(const int16_t *deriv, size_t step, size_t fs, const cart_plane_params *params,
                                                           int params_stride, uint8_t *planes, size_t pstep, size_t pfs, int w, int h) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, frame = blockIdx.z;
    if (x >= w || y >= h) return;
    const cart_plane_params pp = params[(size_t)frame * params_stride];
    const int d = row_ptr(deriv, fs, step, frame, y)[x];
    int plane = CART_PLANE_UNKNOWN;
    if (d != INVALID && d >= pp.horizontal_min && d < pp.horizontal_max) plane = CART_PLANE_HORIZONTAL;
    else if (d != INVALID && d >= pp.vertical_min && d < pp.vertical_max) plane = CART_PLANE_VERTICAL;
    row_ptr(planes, pfs, pstep, frame, y)[x] = (uint8_t)plane;
}

void launch_classify_dev(const int16_t *deriv, size_t step, size_t fs, const cart_plane_params *params_dev, int params_stride,
                         uint8_t *planes, size_t pstep, size_t pfs, int w, int h, int n_frames, hipStream_t s) {
    dim3 grid((w + 63) / 64, (h + 3) / 4, n_frames), block(64, 4);
    hipLaunchKernelGGL(classify_dev_kernel, grid, block, 0, s, deriv, step, fs, params_dev, params_stride, planes, pstep, pfs, w, h);
}

// ------------------------------------------------------------------ temporal voting (planeseg.cu:199-240)
__global__ __launch_bounds__(256) void temporal_vote_kernel(const uint8_t *planes, size_t pstep, TemporalArgs t, uint8_t *smoothed, size_t sstep,
                                                            int w, int h) {
    const int px = blockIdx.x * 64 + threadIdx.x, py = blockIdx.y * 4 + threadIdx.y;
    if (px >= w || py >= h) return;
    int votes[3] = {0, 0, 0};
    votes[planes[(size_t)py * pstep + px]]++;
    int x = px, y = py;
    for (int k = 0; k < t.n_prev; ++k) {
        // the reference reads the flow at the ORIGINAL pixel, not at the tracked position (:212-213)
        const short2 f = *reinterpret_cast<const short2 *>(reinterpret_cast<const uint8_t *>(t.flow[k]) + (size_t)py * t.flow_step[k] + (size_t)px * 4);
        x -= (int16_t)(f.x >> 5);  // S10.5 -> whole pixels (:216-217)
        y -= (int16_t)(f.y >> 5);
        if (x < 0 || y < 0 || x >= w || y >= h) continue;
        votes[t.prev[k][(size_t)y * t.prev_step[k] + x]]++;
    }
    int best = votes[CART_PLANE_HORIZONTAL] > votes[CART_PLANE_VERTICAL] ? CART_PLANE_HORIZONTAL : CART_PLANE_VERTICAL;
    if (votes[best] == 0) best = CART_PLANE_UNKNOWN;
    smoothed[(size_t)py * sstep + px] = (uint8_t)best;
}

void launch_temporal_vote(const uint8_t *planes, size_t pstep, const TemporalArgs &t, uint8_t *smoothed, size_t sstep, int w, int h, hipStream_t s) {
    dim3 grid((w + 63) / 64, (h + 3) / 4), block(64, 4);
    hipLaunchKernelGGL(temporal_vote_kernel, grid, block, 0, s, planes, pstep, t, smoothed, sstep, w, h);
}

// ------------------------------------------------------------------ depth reprojection
// depth.cpp:18-19: convertTo(CV_32F, 1/16) + cv::cuda::reprojectImageTo3D(Q): 12 B written per 2 B read, one pass.
__global__ __launch_bounds__(256) void reproject_kernel(const int16_t *disp, size_t step, size_t fs, QMatrix Q, float *xyz, size_t ostep,
                                                        size_t ofs, int w, int h) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, frame = blockIdx.z;
    if (x >= w || y >= h) return;
    const float *q = Q.q;
    const float qx = q[1] * y + q[3], qy = q[5] * y + q[7], qz = q[9] * y + q[11], qw = q[13] * y + q[15];
    const float d = (float)row_ptr(disp, fs, step, frame, y)[x] * 0.0625f;
    const float iW = 1.f / (qw + q[12] * x + q[14] * d);
    float *o = row_ptr(xyz, ofs, ostep, frame, y) + 3 * x;
    o[0] = (qx + q[0] * x + q[2] * d) * iW;
    o[1] = (qy + q[4] * x + q[6] * d) * iW;
    o[2] = (qz + q[8] * x + q[10] * d) * iW;
}

void launch_reproject(const int16_t *disp, size_t step, size_t fs, const QMatrix &Q, float *xyz, size_t ostep, size_t ofs, int w, int h,
                      int n_frames, hipStream_t s) {
    dim3 grid((w + 63) / 64, (h + 3) / 4, n_frames), block(64, 4);
    hipLaunchKernelGGL(reproject_kernel, grid, block, 0, s, disp, step, fs, Q, xyz, ostep, ofs, w, h);
}

// ------------------------------------------------------------------ bilinear resize (KITTI source, oracle S16)
// kitti.cpp:169-172: cv::cuda::resize(..., INTER_LINEAR) when the configured image size differs from the files'.
__global__ __launch_bounds__(256) void resize_linear_kernel(const uint8_t *src, size_t sstep, int sw, int sh, int channels, uint8_t *dst, size_t dstep,
                                                            int dw, int dh, float fx, float fy) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const float src_x = (float)x * fx, src_y = (float)y * fy;
    const int x1 = (int)floorf(src_x), y1 = (int)floorf(src_y), x2 = x1 + 1, y2 = y1 + 1;
    const int x2r = min(x2, sw - 1), y2r = min(y2, sh - 1);
    const float wx1 = (float)x2 - src_x, wx2 = src_x - (float)x1, wy1 = (float)y2 - src_y, wy2 = src_y - (float)y1;
    const uint8_t *r1 = src + (size_t)y1 * sstep, *r2 = src + (size_t)y2r * sstep;
    for (int c = 0; c < channels; ++c) {
        float out = 0.f;   // -ffp-contract=off: four multiplies and four adds in this order, like the oracle
        out = out + (float)r1[x1 * channels + c] * (wx1 * wy1);
        out = out + (float)r1[x2r * channels + c] * (wx2 * wy1);
        out = out + (float)r2[x1 * channels + c] * (wx1 * wy2);
        out = out + (float)r2[x2r * channels + c] * (wx2 * wy2);
        const float r = rintf(out);
        dst[(size_t)y * dstep + x * channels + c] = (uint8_t)fminf(fmaxf(r, 0.f), 255.f);
    }
}
void launch_resize_linear(const uint8_t *src, size_t sstep, int sw, int sh, int channels, uint8_t *dst, size_t dstep, int dw, int dh, hipStream_t s) {
    const float fx = (float)((double)sw / (double)dw), fy = (float)((double)sh / (double)dh);
    hipLaunchKernelGGL(resize_linear_kernel, dim3((dw + 63) / 64, (dh + 3) / 4), dim3(64, 4), 0, s, src, sstep, sw, sh, channels, dst, dstep, dw, dh, fx, fy);
}

// ------------------------------------------------------------------ plane-parameter schedule (device replay)
// One block replays the frames of a batch in id order (planeseg.cu:379-403).  At a refresh frame the 256 bins are
// ranked in parallel (descending value, ties by ascending index: oracle S11), then thread 0 runs the persistence
// scan of util::findPeaks (peaks.cpp:32-67), picks the two most persistent peaks (stable order) and derives the
// ranges exactly like HistogramPeakPlaneParameterProvider::updatePlaneParameters (planeseg.cu:408-453).
__global__ __launch_bounds__(256) void plane_schedule_kernel(ScheduleState *state, int provider, int first_id, int n_frames,
                                                             int update_interval, int reset_interval, const int32_t *hists,
                                                             cart_plane_params *params_out) {
    __shared__ int32_t hsnap[256];
    __shared__ int order[256], owner[256];
    __shared__ int born[130], died[130], lft[130], rgt[130];
    __shared__ cart_plane_params cur;
    const int t = threadIdx.x;
    int32_t cumr = state->cum[t];   // this thread's bin of the cumulative histogram
    if (t == 0) cur = state->params;
    __syncthreads();
    // the frames' histogram bins of this thread, fetched 16 frames at a time BEFORE the serial replay: one load per frame
    // inside the loop put a global-memory round trip (~2 us) on every frame's critical path, 30 us per 16-frame launch
    int32_t hv[16];
    for (int k = 0; k < n_frames; ++k) {
        const int fid = first_id + k;
        if ((k & 15) == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) hv[j] = k + j < n_frames ? hists[(size_t)(k + j) * 256 + t] : 0;
        }
        int32_t add = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) add = (k & 15) == j ? hv[j] : add;   // register select: no dynamic indexing (scratch)
        cumr += add;
        const bool refresh = fid % update_interval == 1;                          // planeseg.cu:381
        if (refresh) {   // block-uniform; frames in between touch nothing shared and need no barrier
            hsnap[t] = cumr;                                                      // "download" (:389)
            if (fid % (update_interval * reset_interval) == 1) cumr = 0;          // reset after download (:391-394)
            __syncthreads();
        }
        if (refresh && provider == 1) {
            const int v = hsnap[t];
            int rank = 0;
            for (int j = 0; j < 256; ++j) {
                const int u = hsnap[j];
                rank += (u > v) || (u == v && j < t);
            }
            order[rank] = t;
            owner[t] = -1;
            __syncthreads();
            if (t == 0) {
                int np = 0;
                for (int o = 0; o < 256; ++o) {
                    const int idx = order[o];
                    const int il = idx > 0 ? owner[idx - 1] : -1, ir = idx < 255 ? owner[idx + 1] : -1;
                    if (il < 0 && ir < 0) { born[np] = lft[np] = rgt[np] = idx; died[np] = -1; owner[idx] = np++; }
                    else if (il >= 0 && ir < 0) { rgt[il] += 1; owner[idx] = il; }
                    else if (il < 0 && ir >= 0) { lft[ir] -= 1; owner[idx] = ir; }
                    else if (hsnap[born[il]] > hsnap[born[ir]]) { died[ir] = idx; rgt[il] = rgt[ir]; owner[rgt[il]] = owner[idx] = il; }
                    else { died[il] = idx; lft[ir] = lft[il]; owner[lft[ir]] = owner[idx] = ir; }
                }
                if (np >= 2) {
                    // two most persistent peaks, first-born wins ties (stable sort of peaks.cpp:70)
                    int b0 = -1, b1 = -1;
                    long long p0 = -1, p1 = -1;
                    for (int i = 0; i < np; ++i) {
                        const long long pers = died[i] < 0 ? 2147483647LL : (long long)hsnap[born[i]] - hsnap[died[i]];
                        if (pers > p0) { b1 = b0; p1 = p0; b0 = i; p0 = pers; }
                        else if (pers > p1) { b1 = i; p1 = pers; }
                    }
                    int pv = born[b0], ph = born[b1];
                    if (abs(pv - 128) > abs(ph - 128)) { const int tmp = pv; pv = ph; ph = tmp; }
                    cur.vertical_center = pv - 128;
                    cur.horizontal_center = ph - 128;
                    int valley = min(pv, ph);
                    for (int i = valley; i < max(pv, ph); ++i)
                        if (hsnap[i] < hsnap[valley]) valley = i;
                    const int vdist = abs(valley - pv), hdist = abs(valley - ph);
                    if (vdist != 0 && hdist != 0) {
                        const int vslope = (hsnap[pv] - hsnap[valley]) / vdist, hslope = (hsnap[ph] - hsnap[valley]) / hdist;
                        if (vslope != 0 && hslope != 0) {
                            const int vwidth = hsnap[pv] / vslope, hwidth = hsnap[ph] / hslope;
                            cur.vertical_min = pv - vwidth - 128; cur.vertical_max = valley - 127;
                            cur.horizontal_min = valley - 127; cur.horizontal_max = ph + hwidth - 127;
                        }
                    }
                }
            }
            __syncthreads();
        }
        if (t == 0) params_out[k] = cur;   // `cur` is written and read by thread 0 only
    }
    state->cum[t] = cumr;
    if (t == 0) state->params = cur;
}

void launch_plane_schedule(ScheduleState *state, int provider, int first_id, int n_frames, int update_interval, int reset_interval,
                           const int32_t *hists, cart_plane_params *params_out, hipStream_t s) {
    hipLaunchKernelGGL(plane_schedule_kernel, dim3(1), dim3(256), 0, s, state, provider, first_id, n_frames, update_interval, reset_interval,
                       hists, params_out);
}

// ------------------------------------------------------------------ connected components (tile-local + border union-find)
// 1. ccl_tile:   a workgroup labels a 64 x 32 tile in LDS (runs by ballot, LDS union-find) -> every pixel links to its
//                tile root, by global index;
// 2. ccl_border: one union per run that touches a tile border, atomicMin-based on the global link array, so roots are
//                the minimal linear index of the component (oracle S12);
// 3. ccl_final:  tile roots resolve their root once (LDS), every other pixel follows its in-tile link.
// Global parent links only ever decrease (atomicMin) and always point inside the component, so a stale
// read (per-XCD L2s are not coherent inside a launch) can only lengthen a walk, never break it;
// later passes run in later launches and therefore see every link.
// find with path halving: every visited node is re-linked to its grandparent.  A link only ever moves to a smaller index
// of the same component, so a stale read elsewhere merely walks the old (still valid) path.
__device__ __forceinline__ int ccl_find(int32_t *L, int i) {
    int p = __hip_atomic_load(&L[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != i) {
        const int gp = __hip_atomic_load(&L[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp != p) atomicMin(&L[i], gp);
        i = p; p = gp;
    }
    return i;
}

__device__ __forceinline__ void ccl_union(int32_t *L, int a, int b) {
    for (int guard = 0; guard < (1 << 24); ++guard) {  // bounded: every retry strictly lowers a root
        a = ccl_find(L, a); b = ccl_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}

// ---- tile-local labelling in LDS ------------------------------------------------------------------------------------
// One workgroup labels a 64 x 32 tile completely in LDS: a wave owns 8 rows, a row's horizontal runs come from one
// ballot (run head = nearest set bit at or below the lane), vertically touching runs are united with an atomicMin
// union-find on an LDS parent array (only where a run starts above or below), and every pixel
// leaves with the GLOBAL linear index of its tile-local root.  Local indices r * 64 + x order like global ones inside the
// tile, so the root is the component's smallest pixel of the tile.  What is left for global memory are the unions across
// tile borders (ccl_border_kernel): ~1/25 of the unions of the run-based version, on trees one level deep.
constexpr int CT_TW = 64, CT_TH = 32;

__device__ __forceinline__ int lds_find(const int *P, int i) {
    int p = P[i];
    while (p != i) { i = p; p = P[i]; }
    return i;
}

__global__ __launch_bounds__(256) void ccl_tile_kernel(const uint8_t *planes, size_t pstep, size_t pfs, int32_t *work, int32_t *ncomp_zero, int w, int h,
                                                       size_t npx) {
    __shared__ uint8_t cls[CT_TH][CT_TW];
    __shared__ int parent[CT_TH * CT_TW];
    const int x0 = blockIdx.x * CT_TW, y0 = blockIdx.y * CT_TH, frame = blockIdx.z;
    if (ncomp_zero && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ncomp_zero[frame] = 0;   // ccl_final_kernel<false> counts into it two launches later
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int x = x0 + lane;
    // 1. classes + runs of the wave's rows
#pragma unroll
    for (int k = 0; k < CT_TH / 4; ++k) {
        const int r = wid * (CT_TH / 4) + k, y = y0 + r;
        const int c = (x < w && y < h) ? row_ptr(planes, pfs, pstep, frame, y)[x] : 255;
        cls[r][lane] = (uint8_t)c;
        const int left = __shfl_up(c, 1);
        const bool valid = c <= 1;
        const bool head = valid && (lane == 0 || left != c);
        const unsigned long long heads = __ballot(head);
        const int s = 63 - __clzll((long long)(heads & (~0ull >> (63 - lane))));   // nearest head at or below the lane (exists when valid)
        parent[r * CT_TW + lane] = valid ? r * CT_TW + s : -1;
    }
    __syncthreads();
    // 2. unions of vertically touching runs, where a run starts in either row
#pragma unroll
    for (int k = 0; k < CT_TH / 4; ++k) {
        const int r = wid * (CT_TH / 4) + k;
        if (r == 0) continue;
        const int c = cls[r][lane];
        if (c > 1 || cls[r - 1][lane] != c) continue;
        const bool head = lane == 0 || cls[r][lane - 1] != c, up_head = lane == 0 || cls[r - 1][lane - 1] != c;
        if (!head && !up_head) continue;
        int a = parent[r * CT_TW + lane], b = parent[(r - 1) * CT_TW + lane];   // the two run heads
        for (int guard = 0; guard < CT_TW * CT_TH; ++guard) {   // every retry strictly lowers a root
            a = lds_find(parent, a); b = lds_find(parent, b);
            if (a == b) break;
            if (a < b) { const int t = a; a = b; b = t; }
            const int old = atomicMin(&parent[a], b);
            if (old == a) break;
            a = old;
        }
    }
    __syncthreads();
    // 3. pixel -> run head -> tile root, as a global index
    int32_t *L = work + (size_t)frame * npx;
#pragma unroll
    for (int k = 0; k < CT_TH / 4; ++k) {
        const int r = wid * (CT_TH / 4) + k, y = y0 + r;
        if (x >= w || y >= h) continue;
        const int p = parent[r * CT_TW + lane];
        int out = -1;
        if (p >= 0) {
            const int root = lds_find(parent, p);
            out = (y0 + root / CT_TW) * w + x0 + (root % CT_TW);
        }
        L[y * w + x] = out;
    }
}

// Unions across tile borders on the global parent array (a pixel points at its tile root, tile roots at themselves).
// blockIdx.y < nby: the pixels of the rows y = 32 k against the row above; else the pixels of the columns x = 64 k against
// the column to their left.  A pair is skipped when the pair before it along the border holds the same class on both
// sides (that pair, plus the in-tile adjacency, already connects it) -- except on a tile's first row, where the "in-tile
// adjacency" of a column pair would itself be a border pair that defers back to this one.
__global__ __launch_bounds__(256) void ccl_border_kernel(const uint8_t *planes, size_t pstep, size_t pfs, int32_t *work, int w, int h, size_t npx,
                                                         int nrows, int ncols) {
    const int frame = blockIdx.z;
    int32_t *L = work + (size_t)frame * npx;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if ((int)blockIdx.y < nrows) {   // horizontal border below row y - 1
        const int y = ((int)blockIdx.y + 1) * CT_TH, x = t;
        if (x >= w) return;
        const uint8_t *row = row_ptr(planes, pfs, pstep, frame, y), *up = row_ptr(planes, pfs, pstep, frame, y - 1);
        const uint8_t c = row[x];
        if (c > 1 || up[x] != c) return;
        if (x > 0 && row[x - 1] == c && up[x - 1] == c) return;
        ccl_union(L, L[y * w + x], L[(y - 1) * w + x]);
    } else {                          // vertical border left of column x
        const int x = ((int)blockIdx.y - nrows + 1) * CT_TW, y = t;
        if (y >= h) return;
        const uint8_t *row = row_ptr(planes, pfs, pstep, frame, y);
        const uint8_t c = row[x];
        if (c > 1 || row[x - 1] != c) return;
        if (y % CT_TH != 0) {
            const uint8_t *up = row_ptr(planes, pfs, pstep, frame, y - 1);
            if (up[x] == c && up[x - 1] == c) return;
        }
        ccl_union(L, L[y * w + x], L[y * w + x - 1]);
    }
    (void)ncols;
}

// ---- component statistics, accumulated where the ids are made -------------------------------------------------------------
// The component table (oracle S12: id, label, area, bbox, ascending id) used to take four more launches over the finished id
// map (roots per row, row scan, ordered rank of every root, statistics).  Now the tile that writes a piece of a component adds the
// piece to that component's SCRATCH entry, keyed by the component's root pixel (a per-tile LDS hash first merges the pieces of a
// tile: a road-sized component otherwise receives thousands of same-address global atomics per frame), and notes how many roots
// each of its rows holds (seg[y][tile column]).  ccl_table_kernel then ranks the roots in raster order from those counts alone
// -- no pass over the id map: it only opens the 64-pixel segments that hold a root -- moves every root's scratch entry into its
// table row and ZEROES the entry again.  The scratch ([slot][npx] x 5 ints, zero-neutral encoding: area, w - x0, h - y0, x1 + 1,
// y1 + 1, every field grown by atomicAdd / atomicMax from 0) is therefore all zeros between calls: no memset, no initialised table.
constexpr int kCclStatInts = 5;
constexpr int kCclHash = 256;   // entries of a tile's component hash (a tile with more components than fit adds their pieces to the scratch directly)
struct CclHash { int key[kCclHash], area[kCclHash], x0[kCclHash], y0[kCclHash], x1[kCclHash], y1[kCclHash]; };

__device__ __forceinline__ void ccl_hash_clear(CclHash &hsh, int tid, int w, int h) {
    for (int i = tid; i < kCclHash; i += 256) { hsh.key[i] = -1; hsh.area[i] = 0; hsh.x0[i] = w; hsh.y0[i] = h; hsh.x1[i] = -1; hsh.y1[i] = -1; }
}
__device__ __forceinline__ void ccl_stat_add(int32_t *stat, int root, int area, int x0, int y0, int x1, int y1, int w, int h) {
    int32_t *e = stat + (size_t)root * kCclStatInts;
    atomicAdd(&e[0], area);
    atomicMax(&e[1], w - x0); atomicMax(&e[2], h - y0);
    atomicMax(&e[3], x1 + 1); atomicMax(&e[4], y1 + 1);
}
// one row of a tile, one wave: `id` = the lane's final id (-1: no component) at (x, y).  Counts the row's roots into *seg_out and adds the row's
// pieces (maximal runs of one id inside the wave) to the tile's hash, or straight to the scratch when the hash is full.
__device__ __forceinline__ void ccl_row_stats(CclHash &hsh, int32_t *stat, int32_t *seg_out, int id, int x, int y, int lane, int w, int h) {
    const unsigned long long rootm = __ballot(id >= 0 && id == y * w + x);
    if (lane == 0) *seg_out = __popcll(rootm);
    const int idp = __shfl_up(id, 1);
    const bool in = id >= 0;
    const bool head = in && (lane == 0 || idp != id);
    const unsigned long long heads = __ballot(head), ins = __ballot(in);
    if (!head) return;
    const unsigned long long later = lane == 63 ? 0ull : (~0ull << (lane + 1));
    const unsigned long long stop = (heads | ~ins) & later;   // next head or first non-member lane
    const int end_lane = stop ? __ffsll((long long)stop) - 2 : 63;
    const int len = end_lane - lane + 1;
    int hpos = (int)(((unsigned)id * 2654435761u) >> 24);   // 8 bits
    for (int probe = 0; probe < 8; ++probe, hpos = (hpos + 1) & (kCclHash - 1)) {
        const int old = atomicCAS(&hsh.key[hpos], -1, id);
        if (old == -1 || old == id) {
            atomicAdd(&hsh.area[hpos], len);
            atomicMin(&hsh.x0[hpos], x); atomicMax(&hsh.x1[hpos], x + len - 1);
            atomicMin(&hsh.y0[hpos], y); atomicMax(&hsh.y1[hpos], y);
            return;
        }
    }
    ccl_stat_add(stat, id, len, x, y, x + len - 1, y, w, h);
}
__device__ __forceinline__ void ccl_hash_flush(const CclHash &hsh, int32_t *stat, int tid, int w, int h) {
    for (int i = tid; i < kCclHash; i += 256)
        if (hsh.key[i] >= 0) ccl_stat_add(stat, hsh.key[i], hsh.area[i], hsh.x0[i], hsh.y0[i], hsh.x1[i], hsh.y1[i], w, h);
}

// Final ids, again one workgroup per 64 x 32 tile.  After the border unions a pixel's link is either inside its tile (an
// ordinary pixel pointing at its tile root, or a tile root that a union hung under another root of the same tile) or it
// is a tile root: a link to itself or to a root outside the tile.  Tile roots walk the global forest once and park the
// result in LDS; everybody else follows its in-tile links through an LDS copy of the tile's links until it meets a
// resolved entry (one or two hops).  No per-pixel gathers from global memory, no separate compression pass.
// STATS: the tile also feeds the component table (see above); the count then comes from ccl_table_kernel, not from here.
template <bool STATS>
__global__ __launch_bounds__(256) void ccl_final_kernel(const int32_t *work, int32_t *ids, size_t istep, size_t ifs, int32_t *ncomp, int32_t *stat_all,
                                                        int32_t *seg_all, int w, int h, size_t npx) {
    __shared__ int link[CT_TH * CT_TW];      // >= 0: local index of the pixel's link target; -1: unlabelled; <= -2: resolved, global root = -(value + 2)
    __shared__ int roots;
    __shared__ typename std::conditional<STATS, CclHash, int>::type hsh_mem;   // the tile's component hash exists only with STATS
    CclHash &hsh = reinterpret_cast<CclHash &>(hsh_mem);
    const int x0 = blockIdx.x * CT_TW, y0 = blockIdx.y * CT_TH, frame = blockIdx.z;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int x = x0 + lane;
    const int32_t *L = work + (size_t)frame * npx;
    if (threadIdx.x == 0) roots = 0;
    if constexpr (STATS) ccl_hash_clear(hsh, threadIdx.x, w, h);
#pragma unroll
    for (int k = 0; k < CT_TH / 4; ++k) {
        const int r = wid * (CT_TH / 4) + k, y = y0 + r, li = r * CT_TW + lane;
        int lk = -1;
        if (x < w && y < h) {
            const int gi = y * w + x;
            const int t = L[gi];
            if (t >= 0) {
                const int ty = t / w, tx = t - ty * w;
                const bool inside = ty >= y0 && ty < y0 + CT_TH && tx >= x0 && tx < x0 + CT_TW;
                if (inside && t != gi) {
                    lk = (ty - y0) * CT_TW + (tx - x0);
                } else {   // a tile root: resolve it in the global forest
                    int q = t, n = L[q];
                    while (n != q) { q = n; n = L[q]; }
                    lk = -(q + 2);
                }
            }
        }
        link[li] = lk;
    }
    __syncthreads();
    int found = 0;
    const int ntx = gridDim.x;
#pragma unroll
    for (int k = 0; k < CT_TH / 4; ++k) {
        const int r = wid * (CT_TH / 4) + k, y = y0 + r, li = r * CT_TW + lane;   // y is wave-uniform
        if (y >= h) continue;
        int id = -1;
        if (x < w) {
            int lk = link[li];
            if (lk != -1) {
                while (lk >= 0) lk = link[lk];   // in-tile links only ever lead to smaller indices: terminates at a resolved entry
                id = -(lk + 2);
            }
            row_ptr(ids, ifs, istep, frame, y)[x] = id;
            found += id == y * w + x;
        }
        if constexpr (STATS)
            ccl_row_stats(hsh, stat_all + (size_t)frame * npx * kCclStatInts, seg_all + ((size_t)frame * h + y) * ntx + blockIdx.x, id, x, y, lane, w, h);
    }
    if constexpr (STATS) {
        __syncthreads();
        ccl_hash_flush(hsh, stat_all + (size_t)frame * npx * kCclStatInts, threadIdx.x, w, h);
    } else if (ncomp) {   // components = pixels that are their own root; one global atomic per workgroup (noise scenes have ~10^4 roots per
                          // frame, and that many same-address atomics took longer than the labelling); ccl_tile_kernel zeroed the counter
        if (found) atomicAdd(&roots, found);
        __syncthreads();
        if (threadIdx.x == 0 && roots) atomicAdd(&ncomp[frame], roots);
    }
}

// The statistics half alone, for a caller that hands in an id map (cart_plane_ccl_stats): same tiles, ids read instead of resolved.
// An id that is not a root of the map it comes from (ids[id] != id: not an id map of cart_plane_ccl) is skipped -- its scratch entry
// would never be collected, and the scratch has to return to zero.
__global__ __launch_bounds__(256) void ccl_stats_tile_kernel(const int32_t *ids, size_t istep, size_t ifs, int32_t *stat_all, int32_t *seg_all, int w,
                                                             int h, size_t npx) {
    __shared__ CclHash hsh;
    const int x0 = blockIdx.x * CT_TW, y0 = blockIdx.y * CT_TH, frame = blockIdx.z;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int x = x0 + lane, ntx = gridDim.x;
    ccl_hash_clear(hsh, threadIdx.x, w, h);
    __syncthreads();
    int32_t *stat = stat_all + (size_t)frame * npx * kCclStatInts;
#pragma unroll
    for (int k = 0; k < CT_TH / 4; ++k) {
        const int y = y0 + wid * (CT_TH / 4) + k;
        if (y >= h) continue;
        int id = x < w ? row_ptr(ids, ifs, istep, frame, y)[x] : -1;
        if (id >= 0) {
            const bool ok = (size_t)id < npx && row_ptr(ids, ifs, istep, frame, id / w)[id - (id / w) * w] == id;
            if (!ok) id = -1;
        }
        ccl_row_stats(hsh, stat, seg_all + ((size_t)frame * h + y) * ntx + blockIdx.x, id, x, y, lane, w, h);
    }
    __syncthreads();
    ccl_hash_flush(hsh, stat, threadIdx.x, w, h);
}

// Roots are the pixels whose id is their own linear index, so raster order = ascending id = table order.  One workgroup per band of 32 rows
// (a tile row) and frame: roots before the band and per row of the band from seg[][] (a few thousand ints per frame), then each wave walks its
// rows segment by segment, opens the segments that hold a root, ranks the roots by ballot, writes their table rows from the scratch entries and
// zeroes those.  The band that ends the image writes the frame's component count.
constexpr int kTableWaves = 16;   // waves per band of 32 rows: a row's root-bearing segments are walked one after the other by ONE wave, so the rows go to as many waves as a workgroup holds
__global__ __launch_bounds__(64 * kTableWaves) void ccl_table_kernel(const uint8_t *planes, size_t pstep, size_t pfs, const int32_t *ids, size_t istep, size_t ifs,
                                                        int32_t *stat_all, const int32_t *seg_all, cart_component *table, int max_components,
                                                        int32_t *ncomp, int w, int h, int ntx, size_t npx) {
    constexpr int NT = 64 * kTableWaves;
    __shared__ int red[kTableWaves];
    __shared__ int rowbase[CT_TH + 1];
    const int band = blockIdx.x, frame = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int y0 = band * CT_TH, rows = min(CT_TH, h - y0);
    const int32_t *seg = seg_all + (size_t)frame * h * ntx;
    int before = 0;
    {   // roots in the bands above: up to a few thousand counts per block, eight independent loads in flight per thread (one load per trip ran the last band's
        // block at 28 dependent memory latencies: 18 of the kernel's 18 us)
        const int nb = y0 * ntx;
        int i = tid, part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (; i + 7 * NT < nb; i += 8 * NT) {
#pragma unroll
            for (int u = 0; u < 8; ++u) part[u] += seg[i + u * NT];
        }
        for (; i < nb; i += NT) before += seg[i];
#pragma unroll
        for (int u = 0; u < 8; ++u) before += part[u];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_down(before, o);
    if (lane == 0) red[wid] = before;
    // roots per row of the band: eight lanes per row, each a strided share of the row's segment counts (all loads of the block in flight together)
    if (tid < 8 * CT_TH) {
        const int r = tid >> 3, part = tid & 7;
        int sum = 0;
        if (r < rows)
            for (int t = part; t < ntx; t += 8) sum += seg[(size_t)(y0 + r) * ntx + t];
        sum += __shfl_down(sum, 4, 8); sum += __shfl_down(sum, 2, 8); sum += __shfl_down(sum, 1, 8);
        if (part == 0) rowbase[r + 1] = sum;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int k = 0; k < kTableWaves; ++k) run += red[k];
        rowbase[0] = run;
        for (int r = 0; r < CT_TH; ++r) { run += rowbase[r + 1]; rowbase[r + 1] = run; }   // rowbase[r] = roots before row y0 + r
        if (ncomp && y0 + rows >= h) ncomp[frame] = run;
    }
    __syncthreads();
    int32_t *stat = stat_all + (size_t)frame * npx * kCclStatInts;
    cart_component *tab = table + (size_t)frame * max_components;
    // A wave takes rows wid, wid + kTableWaves, ...: lane t holds the count of segment c0 + t (all of a row's counts in ONE load; the rows' loads are independent
    // and in flight together), the segments that hold a root are then walked from a ballot -- a typical label map has a few hundred roots per frame, so most
    // rows open no segment at all.
    for (int c0 = 0; c0 < ntx; c0 += 64) {
        static_assert(CT_TH % kTableWaves == 0, "rows of a band are dealt evenly over the waves");
        int cnt[CT_TH / kTableWaves];
#pragma unroll
        for (int k = 0; k < CT_TH / kTableWaves; ++k) {
            const int r = wid + kTableWaves * k;
            cnt[k] = (r < rows && c0 + lane < ntx) ? seg[(size_t)(y0 + r) * ntx + c0 + lane] : 0;
        }
#pragma unroll
        for (int k = 0; k < CT_TH / kTableWaves; ++k) {
            const int r = wid + kTableWaves * k;
            if (r >= rows) continue;   // wave-uniform
            const int y = y0 + r;
            unsigned long long todo = __ballot(cnt[k] > 0);
            if (!todo && c0 + 64 >= ntx) continue;
            // roots of this row in the segments before lane t of this chunk (inclusive scan over the lanes, minus the lane's own count)
            int incl = cnt[k];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            const int chunk_total = __shfl(incl, 63);
            const int32_t *irow = row_ptr(ids, ifs, istep, frame, y);
            const uint8_t *prow = row_ptr(planes, pfs, pstep, frame, y);
            while (todo) {
                const int t = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int run = rowbase[r] + __shfl(incl, t) - __shfl(cnt[k], t);   // roots of the image before segment c0 + t of this row
                const int x = (c0 + t) * CT_TW + lane;
                const bool root = x < w && irow[x] == y * w + x;
                const unsigned long long m = __ballot(root);
                if (root) {
                    const int rk = run + __popcll(m & ((1ull << lane) - 1ull));
                    int32_t *e = stat + (size_t)(y * w + x) * kCclStatInts;
                    const int area = e[0], ex0 = e[1], ey0 = e[2], ex1 = e[3], ey1 = e[4];
                    e[0] = 0; e[1] = 0; e[2] = 0; e[3] = 0; e[4] = 0;
                    if (rk < max_components) {
                        cart_component c;
                        c.id = y * w + x; c.label = prow[x]; c.area = area; c.x0 = w - ex0; c.y0 = h - ey0; c.x1 = ex1 - 1; c.y1 = ey1 - 1;
                        tab[rk] = c;
                    }
                }
            }
            if (lane == 0) rowbase[r] += chunk_total;   // the next chunk of 64 segments of this row continues from here (rows of a wave are its own)
        }
    }
}

// ids + count: three launches (the tile kernel zeroes the counter).  With `table`: the final kernel also feeds the component scratch and
// ccl_table_kernel writes table and count -- four launches for ids, count and table, none of them a memset.
void launch_ccl(const uint8_t *planes, size_t pstep, size_t pfs, int32_t *work, int32_t *ids, size_t istep, size_t ifs,
                int32_t *ncomp, int w, int h, int n_frames, hipStream_t s, int32_t *stat, int32_t *seg, cart_component *table, int max_components) {
    const size_t npx = (size_t)w * h;
    const int ntx = (w + CT_TW - 1) / CT_TW, nty = (h + CT_TH - 1) / CT_TH;
    hipLaunchKernelGGL(ccl_tile_kernel, dim3(ntx, nty, n_frames), dim3(256), 0, s, planes, pstep, pfs, work, table ? nullptr : ncomp, w, h, npx);
    const int nrows = nty - 1, ncols = ntx - 1;   // inner tile borders
    if (nrows + ncols > 0) {
        const int span = std::max(nrows > 0 ? w : 0, ncols > 0 ? h : 0);
        hipLaunchKernelGGL(ccl_border_kernel, dim3((span + 255) / 256, nrows + ncols, n_frames), dim3(256), 0, s, planes, pstep, pfs, work, w, h, npx, nrows, ncols);
    }
    if (table) {
        hipLaunchKernelGGL(ccl_final_kernel<true>, dim3(ntx, nty, n_frames), dim3(256), 0, s, (const int32_t *)work, ids, istep, ifs, (int32_t *)nullptr, stat, seg, w, h, npx);
        hipLaunchKernelGGL(ccl_table_kernel, dim3(nty, n_frames), dim3(64 * kTableWaves), 0, s, planes, pstep, pfs, (const int32_t *)ids, istep, ifs, stat, (const int32_t *)seg, table,
                           max_components, ncomp, w, h, ntx, npx);
    } else {
        hipLaunchKernelGGL(ccl_final_kernel<false>, dim3(ntx, nty, n_frames), dim3(256), 0, s, (const int32_t *)work, ids, istep, ifs, ncomp, (int32_t *)nullptr,
                           (int32_t *)nullptr, w, h, npx);
    }
}

// component table of a given id map: two launches
void launch_ccl_stats(const uint8_t *planes, size_t pstep, size_t pfs, const int32_t *ids, size_t istep, size_t ifs, int32_t *stat, int32_t *seg,
                      cart_component *table, int max_components, int32_t *ncomp, int w, int h, int n_frames, hipStream_t s) {
    const size_t npx = (size_t)w * h;
    const int ntx = (w + CT_TW - 1) / CT_TW, nty = (h + CT_TH - 1) / CT_TH;
    hipLaunchKernelGGL(ccl_stats_tile_kernel, dim3(ntx, nty, n_frames), dim3(256), 0, s, ids, istep, ifs, stat, seg, w, h, npx);
    hipLaunchKernelGGL(ccl_table_kernel, dim3(nty, n_frames), dim3(64 * kTableWaves), 0, s, planes, pstep, pfs, ids, istep, ifs, stat, (const int32_t *)seg, table, max_components,
                       ncomp, w, h, ntx, npx);
}
size_t ccl_stats_ws_ints(int w, int h) { return (size_t)w * h * kCclStatInts + (size_t)h * ((w + CT_TW - 1) / CT_TW); }

// ------------------------------------------------------------------ narrow copy (downloads over PCIe)
// A device -> host-mapped copy for the module outputs a caller wants in host memory.  hipMemcpyAsync(D2H) runs here as
// a full-width blit kernel whose waves sit on PCIe-latency stores all over the chip: with two of them per step next to the
// compute kernels, the WTA launch went from 1.27 to 2.2 ms.  This one uses `blocks` workgroups only (PCIe Gen5 x16 needs
// ~100 KB in flight; 8 x 256 lanes x 16 B x 4 deep = 128 KB) and leaves the other CUs alone.
__global__ __launch_bounds__(256) void narrow_copy_kernel(const uint8_t *src, uint8_t *dst, size_t bytes) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const size_t n16 = bytes / 16, stride = (size_t)gridDim.x * 256;
    const v4u *s16 = reinterpret_cast<const v4u *>(src);
    v4u *d16 = reinterpret_cast<v4u *>(dst);
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {   // four independent 16-byte moves in flight per lane
        const v4u a = __builtin_nontemporal_load(s16 + i), b = __builtin_nontemporal_load(s16 + i + stride);
        const v4u c = __builtin_nontemporal_load(s16 + i + 2 * stride), d = __builtin_nontemporal_load(s16 + i + 3 * stride);
        d16[i] = a; d16[i + stride] = b; d16[i + 2 * stride] = c; d16[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) d16[i] = __builtin_nontemporal_load(s16 + i);
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) dst[n16 * 16 + threadIdx.x] = src[n16 * 16 + threadIdx.x];
}
void launch_narrow_copy(const void *src, void *dst, size_t bytes, int blocks, hipStream_t s) {
    hipLaunchKernelGGL(narrow_copy_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst), bytes);
}

int kernel_count() { return 58; }  // device kernels in the library (counted from the generated ISA): sgm_kernels 30 (census, aggregate x6, wta x6, wta_fused x8, rv_merge x6, post, post_interp, uniq_table) + post_kernels 17 + superpixel_kernels 8 + flow 3

}  // namespace cart_amd
