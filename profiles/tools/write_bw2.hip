// write_bw2.hip -- why does hipMemset write at 6.3 TB/s when a plain 16 B/lane store stream reaches 5.7?  Variants of a linear fill:
// grid size, stores per thread, store width, workgroup size; and hipMemsetAsync / hipMemsetD32Async for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
// each thread writes K consecutive 16-byte pieces per grid-stride trip (K = 1: a wave writes 1 KB per instruction, contiguous)
template <int K, int NT>
__global__ __launch_bounds__(NT) void fill16(v4u *p, size_t n) {   // n in 16-byte units
    const size_t stride = (size_t)gridDim.x * NT * K;
    const v4u v = {1u, 2u, 3u, 4u};
    for (size_t i = ((size_t)blockIdx.x * NT + threadIdx.x) * K; i + K <= n; i += stride)
#pragma unroll
        for (int k = 0; k < K; ++k) p[i + k] = v;
}
// K pieces per thread, each piece wave-contiguous (piece k of a trip sits K-th of the trip's span apart): every instruction writes 1 KB contiguous
template <int K, int NT>
__global__ __launch_bounds__(NT) void fill16w(v4u *p, size_t n) {
    const size_t span = (size_t)gridDim.x * NT;
    const v4u v = {1u, 2u, 3u, 4u};
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i + (K - 1) * span < n; i += span * K)
#pragma unroll
        for (int k = 0; k < K; ++k) p[i + k * span] = v;
}
template <int NT>
__global__ __launch_bounds__(NT) void fill8(v2u *p, size_t n) {
    const v2u v = {1u, 2u};
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) p[i] = v;
}
template <int NT>
__global__ __launch_bounds__(NT) void fill4(uint32_t *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) p[i] = 7u;
}
// one workgroup owns one contiguous region (block-contiguous instead of grid-strided)
template <int NT>
__global__ __launch_bounds__(NT) void fill16_blocked(v4u *p, size_t n) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x, b = (size_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    const v4u v = {1u, 2u, 3u, 4u};
    for (size_t i = b + threadIdx.x; i < e; i += NT) p[i] = v;
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}
int main() {
    const size_t bytes = (size_t)8 << 30;
    uint8_t *p; if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const size_t n16 = bytes / 16;
    float t;
    for (int g : {256, 512, 1024, 2048, 4096, 8192, 32768}) {
        t = timeit([&] { hipLaunchKernelGGL((fill16<1, 256>), dim3(g), dim3(256), 0, 0, (v4u *)p, n16); });
        printf("fill16 K=1 wg256 grid %6d      %.3f ms  %.2f TB/s\n", g, t, bytes / t / 1e9);
    }
    for (int g : {256, 1024, 4096}) {
        t = timeit([&] { hipLaunchKernelGGL((fill16<4, 256>), dim3(g), dim3(256), 0, 0, (v4u *)p, n16); });
        printf("fill16 K=4 (64 B per lane) grid %6d  %.3f ms  %.2f TB/s\n", g, t, bytes / t / 1e9);
        t = timeit([&] { hipLaunchKernelGGL((fill16w<4, 256>), dim3(g), dim3(256), 0, 0, (v4u *)p, n16); });
        printf("fill16w K=4 (4 x 1 KB per wave) grid %6d  %.3f ms  %.2f TB/s\n", g, t, bytes / t / 1e9);
        t = timeit([&] { hipLaunchKernelGGL((fill16w<8, 256>), dim3(g), dim3(256), 0, 0, (v4u *)p, n16); });
        printf("fill16w K=8 grid %6d  %.3f ms  %.2f TB/s\n", g, t, bytes / t / 1e9);
        t = timeit([&] { hipLaunchKernelGGL((fill16_blocked<256>), dim3(g), dim3(256), 0, 0, (v4u *)p, n16); });
        printf("fill16 block-contiguous grid %6d  %.3f ms  %.2f TB/s\n", g, t, bytes / t / 1e9);
    }
    for (int g : {1024, 4096}) {
        t = timeit([&] { hipLaunchKernelGGL((fill16<1, 1024>), dim3(g), dim3(1024), 0, 0, (v4u *)p, n16); });
        printf("fill16 K=1 wg1024 grid %6d     %.3f ms  %.2f TB/s\n", g, t, bytes / t / 1e9);
        t = timeit([&] { hipLaunchKernelGGL((fill16<1, 64>), dim3(g * 4), dim3(64), 0, 0, (v4u *)p, n16); });
        printf("fill16 K=1 wg64 grid %6d       %.3f ms  %.2f TB/s\n", g * 4, t, bytes / t / 1e9);
        t = timeit([&] { hipLaunchKernelGGL((fill8<256>), dim3(g), dim3(256), 0, 0, (v2u *)p, bytes / 8); });
        printf("fill8 grid %6d                 %.3f ms  %.2f TB/s\n", g, t, bytes / t / 1e9);
        t = timeit([&] { hipLaunchKernelGGL((fill4<256>), dim3(g), dim3(256), 0, 0, (uint32_t *)p, bytes / 4); });
        printf("fill4 grid %6d                 %.3f ms  %.2f TB/s\n", g, t, bytes / t / 1e9);
    }
    t = timeit([&] { hipMemsetAsync(p, 1, bytes, 0); });
    printf("hipMemsetAsync                    %.3f ms  %.2f TB/s\n", t, bytes / t / 1e9);
    t = timeit([&] { hipMemsetD32Async((hipDeviceptr_t)p, 0x01020304, bytes / 4, 0); });
    printf("hipMemsetD32Async                 %.3f ms  %.2f TB/s\n", t, bytes / t / 1e9);
    return 0;
}
