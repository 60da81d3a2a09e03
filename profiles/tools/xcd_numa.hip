// xcd_numa.hip -- is device memory uniform from every XCD?  A 48 GiB buffer in 1 GiB granules; for every granule and every XCD x (the workgroups
// whose index = x mod 8; the others return at once) the rate at which that XCD alone reads / writes the granule (16 B per lane, non-temporal).
// Prints two tables (GB/s): rows = granule, columns = XCD.  hipcc --offload-arch=gfx950 -O3 xcd_numa.hip -o xcd_numa
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <int WRITE>
__global__ __launch_bounds__(256) void one_xcd(v4u *p, size_t n, int xcd, uint32_t *sink) {
    if ((int)(blockIdx.x & 7) != xcd) return;
    const size_t nb = gridDim.x >> 3, b = blockIdx.x >> 3;
    uint32_t acc = 0;
    for (size_t i = b * 256 + threadIdx.x; i < n; i += nb * 256) {
        if (WRITE) { v4u v = {(uint32_t)i, 1u, 2u, 3u}; __builtin_nontemporal_store(v, p + i); }
        else { v4u v = __builtin_nontemporal_load(p + i); acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (!WRITE && acc == 0x12345678u) sink[0] = acc;
}
int main(int argc, char **argv) {
    const int gib = argc > 1 ? atoi(argv[1]) : 48;
    const size_t G = (size_t)1 << 30;
    uint8_t *buf; uint32_t *sink;
    if (hipMalloc(&buf, gib * G) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMalloc(&sink, 4); hipMemset(buf, 1, gib * G);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wr = 0; wr < 2; ++wr) {
        printf("%s, GB/s per (granule, XCD):\n", wr ? "WRITE" : "READ");
        for (int r = 0; r < gib; ++r) {
            printf("granule %2d:", r);
            for (int x = 0; x < 8; ++x) {
                float best = 1e9f;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (wr) hipLaunchKernelGGL(one_xcd<1>, dim3(8 * 256), dim3(256), 0, 0, (v4u *)(buf + r * G), G / 16, x, sink);
                    else hipLaunchKernelGGL(one_xcd<0>, dim3(8 * 256), dim3(256), 0, 0, (v4u *)(buf + r * G), G / 16, x, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf(" %5.0f", (double)G / best / 1e6);
            }
            printf("\n");
        }
    }
    return 0;
}
