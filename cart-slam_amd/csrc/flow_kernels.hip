// flow_kernels.hip -- dense optical flow by census block matching (oracle S15), the stand-in provider of the "optflow"
// blackboard key.  The reference's provider is NVIDIA's fixed-function optical-flow engine behind
// cv::cuda::NvidiaOpticalFlow_2_0 (src/modules/optflow.cpp:57-70, grid size 1, S10.5 output); there is no algorithm to
// restate, only the output format and the way planeseg.cu:212-219 / sp_planeseg.cu:88-99 consume it
// (previous position = p - (flow >> 5)).
//
// One block = 32x8 output pixels.  The previous frame's census features of the tile + block halo + search range sit in
// LDS (zero outside the image, like S3), the 25 current-frame features of a pixel's 5x5 window in registers; every
// candidate (u,v) costs 25 x (ds_read + v_xor + v_bcnt-accumulate) with the window addressed by ONE per-candidate
// offset + instruction immediates.  Pixels whose window leaves the image take a masked slow path.
#include "engine_internal.h"

namespace cart_amd {

namespace {
constexpr int FT_W = 32, FT_H = 8;
}

template <int B>
__global__ __launch_bounds__(256) void block_flow_kernel(const uint32_t *cen_cur, const uint32_t *cen_prev, int cpitch, int cpadl,
                                                         int w, int h, int radius, int16_t *flow, size_t flow_step) {
    extern __shared__ uint32_t s_prev[];   // [(FT_H + 2B + 2R)][pitch], pitch = FT_W + 2B + 2R
    constexpr int WIN = 2 * B + 1;
    const int R = radius;
    const int pitch = FT_W + 2 * B + 2 * R, rows = FT_H + 2 * B + 2 * R;
    const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H;
    for (int i = threadIdx.x; i < pitch * rows; i += 256) {
        const int ty = i / pitch, tx = i - ty * pitch;
        const int gx = x0 - B - R + tx, gy = y0 - B - R + ty;
        s_prev[i] = (gx >= 0 && gx < w && gy >= 0 && gy < h) ? cen_prev[(size_t)gy * cpitch + cpadl + gx] : 0u;
    }
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int x = x0 + lx, y = y0 + ly;
    // the pixel's window of current-frame features; out-of-image window positions are masked out of every cost
    uint32_t cc[WIN][WIN];
    bool interior = true;
#pragma unroll
    for (int dy = 0; dy < WIN; ++dy)
#pragma unroll
        for (int dx = 0; dx < WIN; ++dx) {
            const int qx = x + dx - B, qy = y + dy - B;
            const bool in = qx >= 0 && qx < w && qy >= 0 && qy < h;
            interior &= in;
            cc[dy][dx] = in ? cen_cur[(size_t)qy * cpitch + cpadl + qx] : 0u;
        }
    __syncthreads();
    if (x >= w || y >= h) return;
    // window origin for (u,v) = (0,0): tile position of q = p - (B,B)
    const int origin = (ly + R) * pitch + (lx + R);
    unsigned best;
    int bu = 0, bv = 0;
    if (interior) {
        auto cost = [&](int off) {
            unsigned c = 0;
#pragma unroll
            for (int dy = 0; dy < WIN; ++dy)
#pragma unroll
                for (int dx = 0; dx < WIN; ++dx) c += (unsigned)__builtin_popcount(cc[dy][dx] ^ s_prev[off + dy * pitch + dx]);
            return c;
        };
        best = cost(origin);
        for (int v = -R; v <= R; ++v)
            for (int u = -R; u <= R; ++u) {
                const unsigned c = cost(origin - v * pitch - u);
                if (c < best) { best = c; bu = u; bv = v; }
            }
    } else {
        auto cost = [&](int off) {
            unsigned c = 0;
#pragma unroll
            for (int dy = 0; dy < WIN; ++dy)
#pragma unroll
                for (int dx = 0; dx < WIN; ++dx) {
                    const int qx = x + dx - B, qy = y + dy - B;
                    if (qx >= 0 && qx < w && qy >= 0 && qy < h) c += (unsigned)__builtin_popcount(cc[dy][dx] ^ s_prev[off + dy * pitch + dx]);
                }
            return c;
        };
        best = cost(origin);
        for (int v = -R; v <= R; ++v)
            for (int u = -R; u <= R; ++u) {
                const unsigned c = cost(origin - v * pitch - u);
                if (c < best) { best = c; bu = u; bv = v; }
            }
    }
    int16_t *row = reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(flow) + (size_t)y * flow_step);
    *reinterpret_cast<short2 *>(row + 2 * x) = make_short2((short)(bu * 32), (short)(bv * 32));
}

void launch_block_flow(const uint32_t *cen_cur, const uint32_t *cen_prev, const Geometry &g, int radius, int block, int16_t *flow,
                       size_t flow_step, hipStream_t s) {
    dim3 grid((g.w + FT_W - 1) / FT_W, (g.h + FT_H - 1) / FT_H), threads(256);
    const size_t lds = (size_t)(FT_W + 2 * block + 2 * radius) * (FT_H + 2 * block + 2 * radius) * sizeof(uint32_t);
    switch (block) {
        case 1: hipLaunchKernelGGL(block_flow_kernel<1>, grid, threads, lds, s, cen_cur, cen_prev, g.cpitch, g.cpadl, g.w, g.h, radius, flow, flow_step); break;
        case 2: hipLaunchKernelGGL(block_flow_kernel<2>, grid, threads, lds, s, cen_cur, cen_prev, g.cpitch, g.cpadl, g.w, g.h, radius, flow, flow_step); break;
        default: hipLaunchKernelGGL(block_flow_kernel<3>, grid, threads, lds, s, cen_cur, cen_prev, g.cpitch, g.cpadl, g.w, g.h, radius, flow, flow_step); break;
    }
}

}  // namespace cart_amd
