// mall_probe.hip -- does a buffer that was just written (non-temporal 16-B stores, as the slabs are) read back faster than HBM when it
// fits the 256 MB Infinity Cache?  For each size: write pass, then read pass, each timed by events; read-after-read for comparison.
// hipcc --offload-arch=gfx950 -O3 mall_probe.hip -o mall_probe && ./mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(256) void wr(v4u *p, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * 256) {
        v4u v = {(uint32_t)i, 1u, 2u, 3u};
        if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}
template <int NT>
__global__ __launch_bounds__(256) void rd(const v4u *p, size_t n, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * 256) {
        v4u v = NT ? __builtin_nontemporal_load(p + i) : p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t maxb = (size_t)4 << 30;
    v4u *p; uint32_t *o; hipMalloc(&p, maxb); hipMalloc(&o, 4);
    v4u *scrub; hipMalloc(&scrub, (size_t)1 << 30);
    hipEvent_t e[4]; for (auto &x : e) hipEventCreate(&x);
    for (size_t mb : {32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048, 4096}) {
        const size_t n = (mb << 20) / 16;
        for (int nt = 0; nt < 2; ++nt) {
            float tw = 0, tr = 0, tr2 = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipLaunchKernelGGL(wr<0>, dim3(4096), dim3(256), 0, 0, scrub, ((size_t)1 << 30) / 16);   // push earlier contents out of the caches
                hipEventRecord(e[0]);
                if (nt) hipLaunchKernelGGL(wr<1>, dim3(4096), dim3(256), 0, 0, p, n); else hipLaunchKernelGGL(wr<0>, dim3(4096), dim3(256), 0, 0, p, n);
                hipEventRecord(e[1]);
                if (nt) hipLaunchKernelGGL(rd<1>, dim3(4096), dim3(256), 0, 0, p, n, o); else hipLaunchKernelGGL(rd<0>, dim3(4096), dim3(256), 0, 0, p, n, o);
                hipEventRecord(e[2]);
                if (nt) hipLaunchKernelGGL(rd<1>, dim3(4096), dim3(256), 0, 0, p, n, o); else hipLaunchKernelGGL(rd<0>, dim3(4096), dim3(256), 0, 0, p, n, o);
                hipEventRecord(e[3]);
                hipEventSynchronize(e[3]);
                float a, b, c; hipEventElapsedTime(&a, e[0], e[1]); hipEventElapsedTime(&b, e[1], e[2]); hipEventElapsedTime(&c, e[2], e[3]);
                if (rep) { tw += a; tr += b; tr2 += c; }
            }
            const double gb = (double)(mb << 20) / 1e9;
            printf("%5zu MB %s  write %.2f TB/s   read-after-write %.2f TB/s   read-after-read %.2f TB/s\n", mb, nt ? "nt    " : "normal",
                   gb / (tw / 4) , gb / (tr / 4), gb / (tr2 / 4));
        }
    }
    return 0;
}
