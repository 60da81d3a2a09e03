// hsweep_bw.hip -- feasibility model of a WTA sweep that runs ALONG the rows (a horizontal path computed inline instead of read):
// frames*H/8 waves (750 at 16 x 375), each walking W = 1242 steps; per step a wave reads NS = 7 slabs x (8 rows x 128 B) -- lane l:
// row l/8, 16 B at d = 16*(l%8) -- through a register ring of K steps, and executes a chain of NV dependent packed VALU
// instructions standing for the recurrence + sums + argmin (~250 in the real kernel).  Reports the read rate.
// hipcc --offload-arch=gfx950 -O3 hsweep_bw.hip -o hsweep_bw && ./hsweep_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
constexpr int W = 1242, H = 375, D = 128, F = 16, NS = 7;

template <int K, int NV, int WPB>
__global__ __launch_bounds__(64 * WPB) void hsweep(const uint8_t *base, uint32_t *sink) {
    const int wave = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    constexpr int waves_per_frame = (H + 7) / 8;
    const int frame = wave / waves_per_frame, y0 = (wave % waves_per_frame) * 8;
    if (frame >= F) return;
    const int y = min(y0 + lane / 8, H - 1);
    const size_t slab = (size_t)W * H * D;
    const uint8_t *p = base + (size_t)frame * 8 * slab + ((size_t)y * W) * D + (lane % 8) * 16;
    v4u ring[K][NS];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int s = 0; s < NS; ++s) ring[k][s] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p + (size_t)s * slab + (size_t)k * D));
    uint32_t a0 = lane, a1 = lane * 3, acc = 0;
    for (int x = 0; x < W; x += K) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            v4u cur[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) cur[s] = ring[k][s];
            const int xn = min(x + k + K, W - 1);
#pragma unroll
            for (int s = 0; s < NS; ++s) ring[k][s] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p + (size_t)s * slab + (size_t)xn * D));
            uint32_t t = 0;
#pragma unroll
            for (int s = 0; s < NS; ++s) t += cur[s].x ^ cur[s].y ^ cur[s].z ^ cur[s].w;
            // two interleaved dependent chains of packed ops (the real step has ~2-way ILP between recurrence and WTA)
#pragma unroll
            for (int i = 0; i < NV / 2; ++i) {
                asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a0) : "v"(t));
                asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a1) : "v"(t));
            }
            acc += a0 ^ a1;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// the same model in plan FUSED_UP's shape: frames * W/8 = 2496 waves walking H = 375 rows, 7 x 1 KB contiguous per step
template <int K, int NV, int WPB>
__global__ __launch_bounds__(64 * WPB) void vsweep(const uint8_t *base, uint32_t *sink) {
    const int wave = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    constexpr int waves_per_frame = (W + 7) / 8;
    const int frame = wave / waves_per_frame, x0 = (wave % waves_per_frame) * 8;
    if (frame >= F) return;
    const int x = min(x0 + lane / 8, W - 1);
    const size_t slab = (size_t)W * H * D, row = (size_t)W * D;
    const uint8_t *p = base + (size_t)frame * 8 * slab + (size_t)x * D + (lane % 8) * 16;
    v4u ring[K][NS];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int s = 0; s < NS; ++s) ring[k][s] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p + (size_t)s * slab + (size_t)k * row));
    uint32_t a0 = lane, a1 = lane * 3, acc = 0;
    for (int y = 0; y < H; y += K) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            v4u cur[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) cur[s] = ring[k][s];
            const int yn = min(y + k + K, H - 1);
#pragma unroll
            for (int s = 0; s < NS; ++s) ring[k][s] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p + (size_t)s * slab + (size_t)yn * row));
            uint32_t t = 0;
#pragma unroll
            for (int s = 0; s < NS; ++s) t += cur[s].x ^ cur[s].y ^ cur[s].z ^ cur[s].w;
#pragma unroll
            for (int i = 0; i < NV / 2; ++i) {
                asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a0) : "v"(t));
                asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a1) : "v"(t));
            }
            acc += a0 ^ a1;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int K, int NV, int WPB>
void runv(const uint8_t *buf, uint32_t *sink) {
    const int waves = F * ((W + 7) / 8), blocks = (waves + WPB - 1) / WPB;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((vsweep<K, NV, WPB>), dim3(blocks), dim3(64 * WPB), 0, 0, buf, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)F * NS * W * H * D;
    printf("column sweep: ring %d rows  %3d VALU/step  %d waves/block (%4d blocks): %.3f ms  %.2f TB/s\n", K, NV, WPB, blocks, best, bytes / best / 1e9);
}

template <int K, int NV, int WPB>
void run(const uint8_t *buf, uint32_t *sink) {
    const int waves = F * ((H + 7) / 8), blocks = (waves + WPB - 1) / WPB;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((hsweep<K, NV, WPB>), dim3(blocks), dim3(64 * WPB), 0, 0, buf, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)F * NS * W * H * D;
    printf("ring %d steps  %3d VALU/step  %d waves/block (%4d blocks): %.3f ms  %.2f TB/s\n", K, NV, WPB, blocks, best, bytes / best / 1e9);
}

int main() {
    const size_t bytes = (size_t)F * 8 * W * H * D;
    uint8_t *buf; uint32_t *sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4); hipMemset(buf, 1, bytes);
    run<4, 100, 1>(buf, sink); run<6, 100, 1>(buf, sink); run<8, 100, 1>(buf, sink);
    run<4, 250, 1>(buf, sink); run<6, 250, 1>(buf, sink); run<8, 250, 1>(buf, sink); run<12, 250, 1>(buf, sink);
    run<6, 250, 2>(buf, sink); run<8, 250, 4>(buf, sink);
    runv<2, 100, 2>(buf, sink); runv<2, 250, 2>(buf, sink); runv<3, 250, 2>(buf, sink); runv<2, 250, 4>(buf, sink); runv<2, 350, 2>(buf, sink); runv<2, 180, 2>(buf, sink);
    return 0;
}
