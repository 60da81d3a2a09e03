// flow_kernels.hip -- dense optical flow by census block matching (oracle S15), the stand-in provider of the "optflow"
// blackboard key.  The reference's provider is NVIDIA's fixed-function optical-flow engine behind
// cv::cuda::NvidiaOpticalFlow_2_0 (src/modules/optflow.cpp:57-70, grid size 1, S10.5 output); there is no algorithm to
// restate, only the output format and the way planeseg.cu:212-219 / sp_planeseg.cu:88-99 consume it
// (previous position = p - (flow >> 5)).
//
// A wave owns 64 adjacent columns of one image row (the outer B on each side are halo: 64 - 2B outputs).  Per candidate
// (u,v) a lane computes only the COLUMN sum of its own column -- 2B+1 x (ds_read + v_xor + v_and + v_bcnt-accumulate)
// against the previous frame's features in LDS -- and the (2B+1)^2 window cost is the sum of the neighbouring lanes'
// column sums, fetched with wave-wide DPP shifts (v_add_u32 ... wave_shr:1 / wave_shl:1): 27 instead of 78 operations
// per candidate and pixel at B = 2.  Window positions outside the image are masked out of the column sums (S15).
#include "engine_internal.h"

namespace cart_amd {

namespace {
constexpr int FT_ROWS = 8;             // output rows per block: 4 waves x 2 rows
constexpr int DPP_WAVE_SHL1 = 0x130;   // lane i <- lane i+1
constexpr int DPP_WAVE_SHR1 = 0x138;   // lane i <- lane i-1

template <int CTRL>
__device__ __forceinline__ unsigned wave_shift(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);   // 0 shifted in at the wave edge
}
}  // namespace

template <int B>
__global__ __launch_bounds__(256) void block_flow_kernel(const uint32_t *cen_cur, const uint32_t *cen_prev, int cpitch, int cpadl,
                                                         int w, int h, int radius, int16_t *flow, size_t flow_step) {
    extern __shared__ uint32_t s_prev[];   // [FT_ROWS + 2B + 2R][64 + 2R]: tile position (ty, tx) = image (y0-B-R+ty, x0-B-R+tx)
    constexpr int WIN = 2 * B + 1, OUT_W = 64 - 2 * B;
    const int R = radius;
    const int pitch = 64 + 2 * R, rows = FT_ROWS + 2 * B + 2 * R;
    const int x0 = blockIdx.x * OUT_W, y0 = blockIdx.y * FT_ROWS;   // first output pixel of the block
    for (int i = threadIdx.x; i < pitch * rows; i += 256) {
        const int ty = i / pitch, tx = i - ty * pitch;
        const int gx = x0 - B - R + tx, gy = y0 - B - R + ty;
        s_prev[i] = (gx >= 0 && gx < w && gy >= 0 && gy < h) ? cen_prev[(size_t)gy * cpitch + cpadl + gx] : 0u;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int xc = x0 - B + lane;                       // this lane's column (halo lanes included)
    const bool col_in = xc >= 0 && xc < w;
    for (int half = 0; half < 2; ++half) {
        const int ly = wid + 4 * half, y = y0 + ly;     // wave-uniform
        if (y >= h) break;
        // the lane's column of current features over the window rows; mask = 0 where the position is outside the image
        uint32_t cc[WIN], mk[WIN];
#pragma unroll
        for (int dy = 0; dy < WIN; ++dy) {
            const int qy = y + dy - B;
            const bool in = col_in && qy >= 0 && qy < h;
            cc[dy] = in ? cen_cur[(size_t)qy * cpitch + cpadl + xc] : 0u;
            mk[dy] = in ? 0xffffffffu : 0u;
        }
        // tile position of (column xc, window row 0) for (u,v) = (0,0)
        const int origin = (ly + R) * pitch + (lane + R);
        auto cost = [&](int off) {
            unsigned c = 0;
#pragma unroll
            for (int dy = 0; dy < WIN; ++dy) c += (unsigned)__builtin_popcount((cc[dy] ^ s_prev[off + dy * pitch]) & mk[dy]);
            unsigned sum = c, l = c, r = c;
#pragma unroll
            for (int k = 0; k < B; ++k) {
                l = wave_shift<DPP_WAVE_SHR1>(l);   // column sums of x-1, x-2, ...
                r = wave_shift<DPP_WAVE_SHL1>(r);   // x+1, x+2, ...
                sum += l + r;
            }
            return sum;
        };
        unsigned best = cost(origin);
        int bu = 0, bv = 0;
        for (int v = -R; v <= R; ++v)
            for (int u = -R; u <= R; ++u) {
                const unsigned c = cost(origin - v * pitch - u);
                if (c < best) { best = c; bu = u; bv = v; }
            }
        if (lane >= B && lane < 64 - B && xc < w) {
            int16_t *row = reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(flow) + (size_t)y * flow_step);
            *reinterpret_cast<short2 *>(row + 2 * xc) = make_short2((short)(bu * 32), (short)(bv * 32));
        }
    }
}

void launch_block_flow(const uint32_t *cen_cur, const uint32_t *cen_prev, const Geometry &g, int radius, int block, int16_t *flow,
                       size_t flow_step, hipStream_t s) {
    const int out_w = 64 - 2 * block;
    dim3 grid((g.w + out_w - 1) / out_w, (g.h + FT_ROWS - 1) / FT_ROWS), threads(256);
    const size_t lds = (size_t)(64 + 2 * radius) * (FT_ROWS + 2 * block + 2 * radius) * sizeof(uint32_t);
    switch (block) {
        case 1: hipLaunchKernelGGL(block_flow_kernel<1>, grid, threads, lds, s, cen_cur, cen_prev, g.cpitch, g.cpadl, g.w, g.h, radius, flow, flow_step); break;
        case 2: hipLaunchKernelGGL(block_flow_kernel<2>, grid, threads, lds, s, cen_cur, cen_prev, g.cpitch, g.cpadl, g.w, g.h, radius, flow, flow_step); break;
        default: hipLaunchKernelGGL(block_flow_kernel<3>, grid, threads, lds, s, cen_cur, cen_prev, g.cpitch, g.cpadl, g.w, g.h, radius, flow, flow_step); break;
    }
}

}  // namespace cart_amd
