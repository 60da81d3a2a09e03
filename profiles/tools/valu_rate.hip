// valu_rate.hip -- measures VALU issue cost (cycles per wave64 instruction per SIMD) of the integer ops
// the SGM kernels are built from.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define N_ITER 2000
#define UNROLL 32

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 0x01010101u;
    uint32_t b = seed ^ 0x00070007u, c = 0x05040100u;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            uint32_t &x = a[u & 7];
            if (OP == 0) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 1) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 2) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 4) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 6) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 8) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(x) : "v"(b));
            if (OP == 9) asm volatile("v_min_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x) : "v"(b));
            if (OP == 10) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(x) : "v"(b));
            if (OP == 11) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 12) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(x) : "v"(b));
            if (OP == 13) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(*(double *)&a[(u & 3) * 2]) : "v"(*(double *)&a[0]));
            if (OP == 14) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(x) : "v"(b));
            if (OP == 15) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(x) : "v"(b));
            if (OP == 16) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 17) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 18) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 19) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 20) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 21) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 22) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 23) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 24) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
            if (OP == 25) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x));
            if (OP == 26) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 27) asm volatile("v_min_u16 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 28) asm volatile("v_max_u32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 29) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 30) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 31) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b));
            if (OP == 32) asm volatile("v_pk_add_u16 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(x) : "v"(b));
            if (OP == 33) asm volatile("v_min_i32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 34) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 35) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 36) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 37) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 38) asm volatile("v_minimum3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 39) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 40) asm volatile("v_min3_u16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 41) asm volatile("v_ashr_pk_u8_i32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r + (uint32_t)(t1 - t0);
}

template <int OP>
void run(const char *name, uint32_t *d, int waves_per_simd) {
    int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)N_ITER * UNROLL * waves_per_simd;
    printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (x2.4GHz = %.2f clk)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

// v_pk_minimum3_f16 on u16 bit patterns below 0x7C00 must equal the unsigned minimum (denormals preserved)
__global__ void min3_check(uint32_t *bad) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;   // 2^20 threads: a = t & 0x3ff.., walk b, c
    uint32_t nbad = 0;
    for (uint32_t k = 0; k < 1024; ++k) {
        const uint32_t a = (t * 2654435761u >> 7) % 0x7C00u, b = (k * 40503u + t) % 0x7C00u, c = (k * k + 3 * t) % 0x7C00u;
        const uint32_t pa = a | (b << 16), pb = c | (a << 16), pc = b | (c << 16);
        uint32_t r;
        asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(pa), "v"(pb), "v"(pc));
        const uint32_t m = min(a, min(b, c));
        nbad += r != (m | (m << 16));
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    {
        uint32_t *bad, h = 0;
        hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(min3_check, dim3(4096), dim3(256), 0, 0, bad);
        hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("v_pk_minimum3_f16 vs unsigned min over 2^30 triples of patterns < 0x7C00: %u mismatches\n", h);
    }
    uint32_t *d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {4}) {
        run<0>("v_min_u32", d, w); run<1>("v_pk_min_u16", d, w); run<2>("v_pk_add_u16", d, w); run<3>("v_perm_b32", d, w);
        run<4>("v_bcnt_u32_b32", d, w); run<5>("v_xor_b32", d, w); run<6>("v_min3_u32", d, w); run<7>("v_add_u32", d, w);
        run<8>("v_alignbit_b32", d, w); run<9>("v_min_u32_dpp", d, w); run<10>("v_pk_sub_u16 clamp", d, w);
        run<11>("v_pk_mad_u16", d, w); run<12>("v_lshl_or_b32", d, w); run<13>("v_pk_fma_f32", d, w); run<14>("v_fma_f32", d, w);
        run<15>("v_add_u32_sdwa", d, w); run<16>("v_sad_u8", d, w); run<17>("v_dot4_u32_u8", d, w);
        run<18>("v_pk_min_f16", d, w); run<19>("v_pk_min_i16", d, w); run<20>("v_pk_add_f16", d, w); run<21>("v_sub_u32", d, w);
        run<22>("v_and_b32", d, w); run<23>("v_or_b32", d, w); run<24>("v_mov_b32", d, w); run<25>("v_lshlrev_b32", d, w);
        run<26>("v_min_f32", d, w); run<27>("v_min_u16", d, w); run<28>("v_max_u32", d, w); run<29>("v_add3_u32", d, w);
        run<30>("v_and_or_b32", d, w); run<31>("v_cndmask_b32", d, w); run<32>("v_pk_add_u16 opsel", d, w); run<33>("v_min_i32", d, w);
        run<34>("v_xad_u32", d, w); run<35>("v_pk_mul_lo_u16", d, w);
        run<36>("v_pk_minimum3_f16", d, w); run<37>("v_pk_maximum3_f16", d, w); run<38>("v_minimum3_f32", d, w); run<39>("v_bitop3_b32", d, w);
        run<40>("v_min3_u16", d, w); run<41>("v_ashr_pk_u8_i32", d, w);
    }
    return 0;
}
