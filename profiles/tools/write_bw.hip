// write_bw.hip -- HBM write throughput of (a) a linear 16 B/lane stream, (b) the row-major SGM slab pattern
// (each wave writes 1 KB per step, steps W*D = 159 KB apart, 375 steps), (c) the same bytes with each
// wave's steps contiguous (column-tiled slab layout).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int W = 1242, H = 375, D = 128, NF = 16, NP = 8;
__global__ __launch_bounds__(256) void linear(uint4 *p, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint4 v = make_uint4(i, i, i, i);
    for (; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
// mode 0: row-major [slab][y][x][D]; mode 1: column-tiled [slab][x/8][y][8][D]
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <int FL>
__device__ __forceinline__ void st16(uint8_t *a, uint4 v) {
    v4u q = {v.x, v.y, v.z, v.w};
    if (FL == 0) *reinterpret_cast<v4u *>(a) = q;
    if (FL == 1) __builtin_nontemporal_store(q, reinterpret_cast<v4u *>(a));
    if (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(a), "v"(q) : "memory");
    if (FL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(a), "v"(q) : "memory");
    if (FL == 4) asm volatile("global_store_dwordx4 %0, %1, off nt sc1" ::"v"(a), "v"(q) : "memory");
}
template <int MODE, int FL = 0>
__global__ __launch_bounds__(256) void slabs(uint8_t *p) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int tiles = (W + 7) / 8;                  // waves per slab
    const int gw = blockIdx.x * 4 + wid;            // global wave
    const int slab = gw / tiles, tile = gw % tiles;
    if (slab >= NF * NP) return;
    const int x = tile * 8 + lane / 8;
    if (x >= W) return;
    uint8_t *base = p + (size_t)slab * W * H * D;
    uint4 v = make_uint4(lane, gw, 3, 4);
    for (int y = 0; y < H; ++y) {
        size_t off = MODE == 0 ? ((size_t)y * W + x) * D + (lane % 8) * 16
                               : (((size_t)tile * H + y) * 8 + lane / 8) * D + (lane % 8) * 16;
        st16<FL>(base + off, v);
        v.x += 1;
    }
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    const size_t bytes = (size_t)NF * NP * ((W + 7) / 8 * 8) * H * D;
    uint8_t *p; hipMalloc(&p, bytes);
    float t;
    t = timeit([&] { hipLaunchKernelGGL(linear, dim3(2048 * 4), dim3(256), 0, 0, (uint4 *)p, bytes / 16); });
    printf("linear stream      %.3f ms  %.2f TB/s\n", t, bytes / t / 1e9);
    const int nwaves = NF * NP * ((W + 7) / 8);
    t = timeit([&] { hipLaunchKernelGGL(slabs<0>, dim3((nwaves + 3) / 4), dim3(256), 0, 0, p); });
    printf("row-major slabs    %.3f ms  %.2f TB/s\n", t, (double)NF * NP * W * H * D / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(slabs<1>, dim3((nwaves + 3) / 4), dim3(256), 0, 0, p); });
    printf("column-tiled slabs %.3f ms  %.2f TB/s\n", t, (double)NF * NP * W * H * D / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((slabs<0, 1>), dim3((nwaves + 3) / 4), dim3(256), 0, 0, p); });
    printf("row-major nt       %.3f ms  %.2f TB/s\n", t, (double)NF * NP * W * H * D / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((slabs<0, 2>), dim3((nwaves + 3) / 4), dim3(256), 0, 0, p); });
    printf("row-major sc0 sc1  %.3f ms  %.2f TB/s\n", t, (double)NF * NP * W * H * D / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((slabs<0, 3>), dim3((nwaves + 3) / 4), dim3(256), 0, 0, p); });
    printf("row-major sc1      %.3f ms  %.2f TB/s\n", t, (double)NF * NP * W * H * D / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((slabs<0, 4>), dim3((nwaves + 3) / 4), dim3(256), 0, 0, p); });
    printf("row-major nt sc1   %.3f ms  %.2f TB/s\n", t, (double)NF * NP * W * H * D / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((slabs<1, 1>), dim3((nwaves + 3) / 4), dim3(256), 0, 0, p); });
    printf("col-tiled nt       %.3f ms  %.2f TB/s\n", t, (double)NF * NP * W * H * D / t / 1e9);
    t = timeit([&] { hipMemsetAsync(p, 1, bytes, 0); });
    printf("hipMemset          %.3f ms  %.2f TB/s\n", t, bytes / t / 1e9);
    return 0;
}
