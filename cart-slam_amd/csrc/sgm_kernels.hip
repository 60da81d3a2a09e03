// sgm_kernels.hip -- gfx950 kernels for the SGM core of the disparity module.
//
// Replaces what cv::cuda::StereoSGM::compute does for the reference
// (src/modules/disparity/disparity.cu:71): census 9x7, per-direction path
// aggregation into u8 cost slabs, winner-takes-all with uniqueness / sub-pixel / right view,
// 3x3 medians, left-right check and range fix.  Written for wave64: a pixel is owned by D/16
// adjacent lanes (16 disparities per lane, packed u16 pairs), neighbour exchange and the min over
// D are DPP ops inside a 16-lane row; see the comments at each kernel.
#include <type_traits>

#include <algorithm>
#include <cstdlib>
#include "engine_internal.h"

namespace cart_amd {

// ------------------------------------------------------------------ DPP helpers
constexpr int DPP_ROW_SHL1 = 0x101;
constexpr int DPP_ROW_SHR1 = 0x111;

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

template <int N>
__device__ __forceinline__ void load_u32s(const uint32_t *p, uint32_t (&r)[N]) {
    static_assert(N % 4 == 0, "N must be a multiple of 4");
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(p + 4 * i);
        r[4 * i + 0] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
    }
}

// ------------------------------------------------------------------ gray + census
// One block = 64x16 output pixels of one image; LDS tile with a 4-column / 3-row halo.  BGR->gray (oracle S1)
// is fused into the tile load; the gray plane is written out because the left-right check masks on
// gray_left == 0.  Each thread produces 4 horizontally adjacent features from 7 x 3 aligned dword reads of the
// tile (its 12-byte-wide window), so the 31 comparisons per pixel run on register bytes.  The kernel also
// resets the packed right-view minima.
constexpr int CT_W = 64, CT_H = 16, CT_LW = CT_W + 8, CT_LH = CT_H + 6, CT_PITCH = 76;  // bytes; 19 dwords per row

// One census bit: f = 2 f + (byte SA of a > byte SB of b).  The byte selects ride on the compare (SDWA) and the bit enters
// through the carry of v_addc, so a comparison costs two VALU instructions instead of two extracts, a compare and a
// shift-or (the kernel is VALU-bound: 124 comparisons per thread).
#define CART_CENSUS_BIT(SA, SB)                                                                                        \
    asm("v_cmp_gt_u32_sdwa vcc, %1, %2 src0_sel:BYTE_" #SA " src1_sel:BYTE_" #SB "\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" \
        : "+v"(f) : "v"(a), "v"(b) : "vcc")
// Stores at less than natural alignment (HSA runs the memory pipeline in unaligned mode; with the alignment spelled out
// the compiler emits one global_store_dword / _dwordx4 instead of splitting into bytes / dwords).
struct __attribute__((packed, aligned(1))) U32A1 { uint32_t v; };
struct __attribute__((packed, aligned(4))) U128A4 { uint32_t x, y, z, w; };

__device__ __forceinline__ void census_bit(uint32_t &f, uint32_t a, int sa, uint32_t b, int sb) {
    switch (sa * 4 + sb) {   // constant after unrolling
        case 0: CART_CENSUS_BIT(0, 0); break;   case 1: CART_CENSUS_BIT(0, 1); break;
        case 2: CART_CENSUS_BIT(0, 2); break;   case 3: CART_CENSUS_BIT(0, 3); break;
        case 4: CART_CENSUS_BIT(1, 0); break;   case 5: CART_CENSUS_BIT(1, 1); break;
        case 6: CART_CENSUS_BIT(1, 2); break;   case 7: CART_CENSUS_BIT(1, 3); break;
        case 8: CART_CENSUS_BIT(2, 0); break;   case 9: CART_CENSUS_BIT(2, 1); break;
        case 10: CART_CENSUS_BIT(2, 2); break;  case 11: CART_CENSUS_BIT(2, 3); break;
        case 12: CART_CENSUS_BIT(3, 0); break;  case 13: CART_CENSUS_BIT(3, 1); break;
        case 14: CART_CENSUS_BIT(3, 2); break;  default: CART_CENSUS_BIT(3, 3); break;
    }
}
#undef CART_CENSUS_BIT
// bit = I(window column ca of row registers wa) > I(window column cb of wb)
__device__ __forceinline__ void census_cmp(uint32_t &f, const uint32_t (&wa)[3], int ca, const uint32_t (&wb)[3], int cb) {
    census_bit(f, wa[ca >> 2], ca & 3, wb[cb >> 2], cb & 3);
}

__global__ __launch_bounds__(256) void census_kernel(ImageBatch left, ImageBatch right, int channels,
                                                     uint8_t *gray_l, uint8_t *gray_r, uint32_t *cen_l,
                                                     uint32_t *cen_r, uint32_t *right_pk, Geometry g) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[CT_LH * CT_PITCH];
    const int frame = blockIdx.z >> 1, side = blockIdx.z & 1;
    // the two sides are read in place: a by-value copy of the selected batch would put its frame table in scratch
    const uint8_t *src;
    size_t img_step;
    if (side) { src = right.scattered ? right.frames[frame] : right.ptr + (size_t)frame * right.frame_stride; img_step = right.step; }
    else { src = left.scattered ? left.frames[frame] : left.ptr + (size_t)frame * left.frame_stride; img_step = left.step; }
    uint8_t *gray = (side ? gray_r : gray_l) + (size_t)frame * g.npx;
    uint32_t *cen = (side ? cen_r : cen_l) + (size_t)frame * g.census_elems;
    const int x0 = blockIdx.x * CT_W, y0 = blockIdx.y * CT_H;
    const int tid = threadIdx.x;

    // gray input: the tile is 18 dwords x 22 rows, fetched as the two aligned dwords that cover each element's four bytes
    // (rows of a 1242-wide image start at 2 mod 4) and funnel-shifted, then one LDS dword store -- two rounds of the block
    // instead of seven rounds of byte loads and byte stores.  Elements within 8 bytes of a row end are assembled bytewise:
    // the aligned pair never reaches outside the row it belongs to.  Interior elements also carry the gray copy the post
    // stage reads.
    const bool dwords = channels == 1;
    if (dwords) {
        for (int i = tid; i < CT_LH * (CT_LW / 4); i += 256) {
            const int ty = i / (CT_LW / 4), k = i - ty * (CT_LW / 4);
            const int gx = x0 - 4 + 4 * k, gy = y0 - 3 + ty;
            uint32_t v = 0;
            if (gy >= 0 && gy < g.h && gx + 3 >= 0 && gx < g.w) {
                const uint8_t *row = src + (size_t)gy * img_step;
                if (gx >= 4 && gx + 8 <= g.w) {
                    const uint32_t a = (uint32_t)reinterpret_cast<uintptr_t>(row + gx) & 3u;   // pointer arithmetic keeps the global address space
                    const uint32_t *ap = reinterpret_cast<const uint32_t *>(row + gx - a);
                    v = __builtin_amdgcn_alignbyte(ap[1], ap[0], a);
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (gx + b >= 0 && gx + b < g.w) v |= (uint32_t)row[gx + b] << (8 * b);
                }
                if (k >= 1 && k < 1 + CT_W / 4 && ty >= 3 && ty < 3 + CT_H) {
                    uint8_t *gp = gray + (size_t)gy * g.w + gx;
                    if (gx + 3 < g.w) {
                        reinterpret_cast<U32A1 *>(gp)->v = v;
                    } else {
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            if (gx + b < g.w) gp[b] = (uint8_t)(v >> (8 * b));
                    }
                }
            }
            *reinterpret_cast<uint32_t *>(tile + ty * CT_PITCH + 4 * k) = v;
        }
    }
    for (int i = tid; i < (dwords ? 0 : CT_LH * CT_LW); i += 256) {
        const int ty = i / CT_LW, tx = i - ty * CT_LW;
        const int gx = x0 - 4 + tx, gy = y0 - 3 + ty;
        uint32_t v = 0;
        if (gx >= 0 && gx < g.w && gy >= 0 && gy < g.h) {
            const uint8_t *row = src + (size_t)gy * img_step;
            if (channels == 3) {
                const uint32_t b = row[3 * gx], gg = row[3 * gx + 1], r = row[3 * gx + 2];
                v = (1868u * b + 9617u * gg + 4899u * r + 8192u) >> 14;
            } else {
                v = row[gx];
            }
            if (tx >= 4 && tx < 4 + CT_W && ty >= 3 && ty < 3 + CT_H) gray[(size_t)gy * g.w + gx] = (uint8_t)v;
        }
        tile[ty * CT_PITCH + tx] = (uint8_t)v;
    }
    __syncthreads();

    // thread -> 4 pixels: tile columns 4*tq+4 .. 4*tq+7 of row ly (window = tile columns 4*tq .. 4*tq+11)
    const int tq = tid & 15, ly = tid >> 4;
    const int y = y0 + ly, xb = x0 + 4 * tq;
    if (y >= g.h || xb >= g.w) return;
    uint32_t w[7][3];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(tile + (ly + r) * CT_PITCH) + tq;
        w[r][0] = rp[0]; w[r][1] = rp[1]; w[r][2] = rp[2];
    }
    uint32_t f[4] = {0, 0, 0, 0};
#pragma unroll
    for (int dy = -3; dy < 0; ++dy)
#pragma unroll
        for (int dx = -4; dx <= 4; ++dx)
#pragma unroll
            for (int i = 0; i < 4; ++i) census_cmp(f[i], w[3 + dy], 4 + i + dx, w[3 - dy], 4 + i - dx);
#pragma unroll
    for (int dx = -4; dx < 0; ++dx)
#pragma unroll
        for (int i = 0; i < 4; ++i) census_cmp(f[i], w[3], 4 + i + dx, w[3], 4 + i - dx);
    const bool yin = y >= 3 && y < g.h - 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] &= (yin && xb + i >= 4 && xb + i < g.w - 4) ? 0xffffffffu : 0u;  // oracle S2: border features are 0
    uint32_t *crow = cen + (size_t)y * g.cpitch + g.cpadl + xb;    // cpitch, cpadl, xb are multiples of 4: 16-byte aligned
    if (xb + 3 < g.w) {
        *reinterpret_cast<uint4 *>(crow) = make_uint4(f[0], f[1], f[2], f[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (xb + i < g.w) crow[i] = f[i];
    }
    if (side == 0) {
        uint32_t *rp = right_pk + (size_t)frame * g.npx + (size_t)y * g.w + xb;
        if (xb + 3 < g.w) {
            *reinterpret_cast<U128A4 *>(rp) = U128A4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (xb + i < g.w) rp[i] = 0xffffffffu;
        }
    }
}

void launch_census(const ImageBatch &left, const ImageBatch &right, int channels, int n_frames, uint8_t *gray_l,
                   uint8_t *gray_r, uint32_t *cen_l, uint32_t *cen_r, uint32_t *right_pk, const Geometry &g,
                   hipStream_t s) {
    dim3 grid((g.w + CT_W - 1) / CT_W, (g.h + CT_H - 1) / CT_H, n_frames * 2), block(256);
    hipLaunchKernelGGL(census_kernel, grid, block, 0, s, left, right, channels, gray_l, gray_r, cen_l, cen_r,
                       right_pk, g);
}

// ------------------------------------------------------------------ path aggregation
// All directions of all frames in ONE launch (blockIdx.x -> direction + a group of scan lines,
// blockIdx.y -> frame).  Every direction is a set of independent 1-D lines: vertical and
// diagonal lines are indexed by their (skewed) entry column so no state ever crosses pixels.
//
// The kernel is VALU-issue bound (rocprofv3: SQ_ACTIVE_INST_VALU ~ 93 % of SIMD time in the first,
// 32-bit version), so the recurrence runs on PACKED u16 pairs (v_pk_min_u16 / v_pk_add_u16: two
// disparities per instruction).  A pixel is owned by LPP = D/16 adjacent lanes, 16 disparities per
// lane held in 8 registers with a split-halves layout  reg i = (L[d0+i], L[d0+i+8]) :
//   * the d-1 / d+1 neighbour vectors of reg i are simply reg i-1 / reg i+1 (register renaming);
//     only reg 0 / reg 7 need one v_perm_b32 that stitches in the neighbouring lane's value
//     (DPP row_shr/row_shl), and that same v_perm writes 0xFFFF (= never chosen) at the ends of the
//     disparity range through a per-lane selector,
//   * the matching cost is popcount(xor) with the "- min" of the recurrence folded into
//     v_bcnt_u32_b32's accumulate operand, packed by one v_perm_b32 per pair,
//   * the u8 slab bytes are produced by v_perm_b32 byte gathers (8 per 16 cells),
//   * min over D = packed min tree + DPP (quad_perm / row_half_mirror / row_mirror) on the
//     replicated pair, which doubles as the packed (m,m) operand of the next step.
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
// min of three packed u16 pairs in ONE instruction: gfx950's v_pk_minimum3_f16 applied to the bit patterns.  Valid where every operand
// half is a cost below 0x7C00 (no inf / NaN pattern; non-negative, so IEEE order = unsigned order) -- the recurrence's operands are
// <= 255 + P2 < 1024, i.e. f16 denormals, which the wave's mode register must preserve (keep_f16_denormals below).  Issue cost as
// v_pk_min_u16 (profiles/tools/valu_rate.hip), so each use saves one of ~100 instructions of the VALU-bound step.
// (A/B against two v_pk_min_u16: profiles/r03_min3.txt.)  The instruction exists on gfx950 only -- the one target this file is written
// for; the host pass of the compiler and any other --offload-arch get the two-instruction form.
__device__ __forceinline__ uint32_t pk_min3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__gfx950__)
    uint32_t r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return pk_min(pk_min(a, b), c);
#endif
}
// MODE.FP_DENORM[3:2] (f16 / f64) = 3: denormals in and out.  That is the code object's default mode, and the default is what pk_min3
// relies on (the asm above carries no dependency on this s_setreg, so the compiler may order the two freely; every kernel that uses
// pk_min3 still states the mode once at its entry, so that a changed default cannot go unnoticed).  hwreg(HW_REG_MODE = 1, offset 6, width 2)
__device__ __forceinline__ void keep_f16_denormals() { __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 3); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b));
}
__device__ __forceinline__ uint32_t perm(uint32_t hi_src, uint32_t lo_src, uint32_t sel) {
    return __builtin_amdgcn_perm(hi_src, lo_src, sel);  // selector bytes: 0-3 = lo_src, 4-7 = hi_src, 0x0c = 0x00, 0x0d = 0xFF
}

constexpr int DPP_QUAD_XOR1 = 0xB1;   // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;   // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// all-reduce (min) over the LPP lanes that own one pixel; v is an unsigned key
template <int LPP>
__device__ __forceinline__ uint32_t group_allmin(uint32_t v) {
    v = min(v, dpp_mov<DPP_QUAD_XOR1>(v));
    v = min(v, dpp_mov<DPP_QUAD_XOR2>(v));
    if constexpr (LPP >= 8) v = min(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
    if constexpr (LPP >= 16) v = min(v, dpp_mov<DPP_ROW_MIRROR>(v));
    return v;
}

template <int LPP>
__device__ __forceinline__ uint32_t group_allsum(uint32_t v) {
    v += dpp_mov<DPP_QUAD_XOR1>(v);
    v += dpp_mov<DPP_QUAD_XOR2>(v);
    if constexpr (LPP >= 8) v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
    if constexpr (LPP >= 16) v += dpp_mov<DPP_ROW_MIRROR>(v);
    return v;
}

// wave-uniform values kept in SGPRs: per-lane addresses become "scalar base + 32-bit lane offset" (saddr form), and
// the per-step pointer increments run on the scalar unit instead of 64-bit VALU adds
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ ptrdiff_t uniform(ptrdiff_t v) {  // element offsets from kernel-argument bases (pointer provenance kept)
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (ptrdiff_t)(((uint64_t)hi << 32) | lo);
}

// Pins a wave-uniform pointer into an SGPR pair so that "pointer + zero-extended 32-bit lane byte offset" selects the
// scalar-base addressing form (global_load ... v_off, s[base:base+1]) instead of a 64-bit VALU add per access.  The
// result is typed as a GLOBAL-address-space pointer: the asm hides the kernel-argument provenance the compiler would
// otherwise use to pick global_* over flat_* instructions.
#define CART_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ CART_GLOBAL T *sgpr(T *p) {
    asm volatile("" : "+s"(p));
    return (CART_GLOBAL T *)p;
}
// keeps the zero-extension of a lane offset next to its use: hoisted out of the loop as a 64-bit value it would no longer
// match the scalar-base addressing pattern
__device__ __forceinline__ unsigned pin_v(unsigned &off) {  // in place: no register copy
    asm volatile("" : "+v"(off));
    return off;
}
__device__ __forceinline__ uint32_t ld_u32(const uint32_t *ubase, unsigned &byte_off) {
    return *(const CART_GLOBAL uint32_t *)((const CART_GLOBAL char *)sgpr(ubase) + pin_v(byte_off));
}
typedef uint32_t u32x4_g4 __attribute__((ext_vector_type(4), aligned(4)));
// 16 consecutive features at a 4-byte aligned address (4 x dwordx4)
__device__ __forceinline__ void ld_u32x16(const uint32_t *ubase, unsigned &byte_off, uint32_t (&r)[16]) {
    const CART_GLOBAL char *b = (const CART_GLOBAL char *)sgpr(ubase) + pin_v(byte_off);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32x4_g4 v = *(const CART_GLOBAL u32x4_g4 *)(b + 16 * i);
        r[4 * i + 0] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
    }
}

struct CensusRegs {
    uint32_t fl;
    uint32_t r[16];
};

// left feature + the lane's 16 right features; pl / pr are wave-uniform, the offsets per lane (bytes)
__device__ __forceinline__ void load_census(const uint32_t *pl, unsigned &off_l, const uint32_t *pr, unsigned &off_r, CensusRegs &c) {
    c.fl = ld_u32(pl, off_l);
    ld_u32x16(pr, off_r, c.r);
}

// x[k] = left feature ^ right feature k: consumes the loaded registers right away so that the next
// prefetch can land in them while the rest of the step runs
__device__ __forceinline__ void agg_xor(const CensusRegs &c, uint32_t (&xr)[16]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) xr[k] = c.fl ^ c.r[k];
}

template <int LPP, bool STORE = true>
__device__ __forceinline__ void agg_step(uint32_t (&a)[8], uint32_t &mm, const uint32_t (&xr)[16], uint32_t sel_lo,
                                         uint32_t sel_hi, uint32_t p1p1, uint32_t p2p2, CART_GLOBAL uint8_t *po) {
    // Issue cost on gfx950 (profiles/tools/valu_rate.hip): v_add/v_sub/v_xor ~2.7 clk, packed ops / v_perm / v_bcnt /
    // shifts ~4.5 clk.  Wherever a packed op cannot carry or borrow between the halves, the plain 32-bit one is used.
    const uint32_t mp2 = mm + p2p2;  // halves stay < 2^15
    // neighbour vectors at the two ends: (prev lane's L[d0-1], own L[d0+7]) and (own L[d0+8], next lane's L[d0+16])
    const uint32_t lo0 = perm(a[7], dpp_mov<DPP_ROW_SHR1>(a[7]), sel_lo);
    const uint32_t hi7 = perm(dpp_mov<DPP_ROW_SHL1>(a[0]), a[0], sel_hi);
    uint32_t n[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t lo = i == 0 ? lo0 : a[i - 1];
        const uint32_t hi = i == 7 ? hi7 : a[i + 1];
        uint32_t t = pk_min(lo, hi) + p1p1;  // no carry between halves (min(lo,hi) is a real cost < 2^15)
        t = pk_min3(t, a[i], mp2);
        // oracle S4: L = C + (min(...) - m).  Every candidate of the min is >= m in both halves, so "- m" is a plain
        // 32-bit subtract; the low-half cost rides on v_bcnt's accumulate operand, the high-half one is shifted in.
        uint32_t u = t - mm;
        asm volatile("" : "+v"(u));  // keep the three adds apart: v_sub (fast), v_bcnt with accumulate, v_lshl_add
        uint32_t lo_sum = (uint32_t)__builtin_popcount(xr[15 - i]) + u;
        asm volatile("" : "+v"(lo_sum));
        n[i] = ((uint32_t)__builtin_popcount(xr[7 - i]) << 16) + lo_sum;
    }
    // u8 slab bytes of the lane's 16 disparities in the kernel's native order (one v_perm per register pair): dword q
    // holds d0 + {2q, 2q+8, 2q+1, 2q+9}; the WTA widens byte pairs straight back into the same split-halves registers
    // (slab byte layout: see kSlabChunkOrder in engine_internal.h)
    if constexpr (STORE) {
        uint4 o;
        o.x = perm(n[1], n[0], 0x06040200u); o.y = perm(n[3], n[2], 0x06040200u);
        o.z = perm(n[5], n[4], 0x06040200u); o.w = perm(n[7], n[6], 0x06040200u);
        // write-once streaming data: non-temporal so the slabs do not evict the census planes from L2
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u q = {o.x, o.y, o.z, o.w};
        __builtin_nontemporal_store(q, (CART_GLOBAL v4u *)po);
    } else {
        (void)po;  // the fused WTA consumes the new costs from the registers
    }
    // min over the pixel's D disparities, replicated into both halves
    uint32_t x = pk_min(pk_min3(n[0], n[1], n[2]), pk_min3(n[3], n[4], pk_min3(n[5], n[6], n[7])));
    x = pk_min(x, __builtin_amdgcn_alignbit(x, x, 16));
    mm = group_allmin<LPP>(x);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = n[i];
}

// ---- LDS staging of the right-census window (vertical + diagonal directions) ----
// The P = 64/LPP pixels a wave works on at one step are adjacent columns of one image row, so the
// P windows of D right features overlap in all but P-1 entries.  Per-lane window loads cost 4 B per
// DP cell through the L1->VGPR path (rocprofv3: TA_BUSY 76 %, TD_BUSY 78 %, 3x line-access inflation
// from the 4-byte-aligned dwordx4 loads) and bound the first two versions of this kernel; instead the
// wave loads the window once, coalesced, writes it to a wave-private LDS buffer and every lane reads
// its 16 features from there.
// Layout: one REGION per disparity chunk g (= lane gl of a pixel): the RL = 16 + P-1 window dwords the P pixels'
// chunk-g lanes read, contiguous, so that a lane's 16 features sit at region base + pg + k -- one address register
// and immediate offsets (the first layout padded every 16 dwords and needed P-1 extra address registers per lane,
// 15 of them at D = 64, which cost that variant an occupancy step and 47 % of its LDS cycles in bank conflicts).
// Regions start RS dwords apart with RS = P/2 (mod 32): the 32 lanes one ds_read_b32 cycle serves are P/2 pixels x
// LPP chunks = LPP runs of P/2 consecutive dwords, which then tile the 32 banks exactly, for every k.
template <int LPP>
struct Win {
    static constexpr int P = 64 / LPP;              // pixels (scan lines) per wave
    static constexpr int D = 16 * LPP;
    static constexpr int RL = 16 + P - 1;           // dwords per region
    static constexpr int RS = LPP == 4 ? 40 : LPP == 8 ? 36 : 34;   // region stride: >= RL, = P/2 mod 32
    static constexpr int NE = LPP * RL;             // staged dwords per step (window dwords shared by two regions are staged twice)
    static constexpr int NLD = (NE + 63) / 64;      // cooperative dword loads per lane and step
    static constexpr int BUF = LPP * RS;            // dwords per LDS buffer
    static_assert(RS >= RL && RS % 32 == (P / 2) % 32, "region stride");
};

// per-lane constants of the staging: where the lane's i-th cooperative load comes from / goes to, and where it reads
template <int LPP>
struct WinLane {
    unsigned goff[Win<LPP>::NLD];   // byte offset from the window's first dword (window dword 0 = disparity D-1 of the wave's first pixel)
    int lslot[Win<LPP>::NLD];       // LDS dword index inside the buffer
    int rbase;                      // LDS dword index of this lane's feature 0
    __device__ __forceinline__ void init(int lane) {
        using WN = Win<LPP>;
#pragma unroll
        for (int i = 0; i < WN::NLD; ++i) {
            const int e = min(64 * i + lane, WN::NE - 1);   // the last round's excess lanes repeat the last element
            const int g = e / WN::RL, o = e - g * WN::RL;
            goff[i] = (unsigned)(WN::D - 16 - 16 * g + o) * 4u;
            lslot[i] = g * WN::RS + o;
        }
        rbase = (lane % LPP) * WN::RS + lane / LPP;
    }
};

template <int LPP>
__device__ __forceinline__ void win_read(const uint32_t *lds_buf, int rbase, uint32_t (&r)[16]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = lds_buf[rbase + k];
}

// 6 waves per SIMD (<= 80 VGPRs) fit without spills for D >= 128; the D = 64 variant carries 15 lane offsets more
// ---- horizontal scans with a sliding right-feature window --------------------------------------------------------
// Along a row the lane's 16 right features move by ONE element per step, so the window lives in 16 registers that are
// renamed instead of reloaded (a 16-step group is unrolled; logical slot k of sub-step j is register (k + j*DX) & 15)
// and a step loads two dwords -- the left feature and the entering right feature -- instead of 17.  Those two come from
// a FIFO filled HS_PF steps ahead: a horizontal wave is alone on its SIMD for most of its 1242 steps, nothing else hides
// the load latency, and with the two-step prefetch of the reloading loop every step waited for memory (0.75 us per step
// against 0.2 us of issue time).  The first group is peeled so that the loop header merges two identical VMEM
// histories (counted s_waitcnt, see the NOTE in aggregate_kernel).
constexpr int HS_PF = 8;
template <int LPP, int DX>
__device__ __forceinline__ void hscan_sliding(uint32_t (&st)[8], uint32_t &mm, const uint32_t *&pl, unsigned &lo_l, const uint32_t *&pr,
                                              unsigned &lo_r, uint8_t *&po, unsigned &lo_o, ptrdiff_t ostride, int groups, uint32_t sel_lo,
                                              uint32_t sel_hi, uint32_t p1p1, uint32_t p2p2) {
    uint32_t win[16], ffl[HS_PF], fnw[HS_PF], xr[16];
    ld_u32x16(pr, lo_r, win);                          // window of step 0
    unsigned lo_n = lo_r + (DX > 0 ? 15u * 4u : 0u);   // the element that enters the window: slot 15 going right, slot 0 going left
#pragma unroll
    for (int q = 0; q < HS_PF; ++q) {                  // FIFO entry q: left feature of step q, entering element of step q + 1
        ffl[q] = ld_u32(pl + q * DX, lo_l);
        fnw[q] = ld_u32(pr + (q + 1) * DX, lo_n);
    }
    auto group = [&]() {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            constexpr int M = 15;
            const int slot = j % HS_PF;
#pragma unroll
            for (int k = 0; k < 16; ++k) xr[k] = ffl[slot] ^ win[(DX > 0 ? k + j : k - j + 16) & M];
            win[(DX > 0 ? j : 15 - j) & M] = fnw[slot];
            __builtin_amdgcn_sched_barrier(0);
            ffl[slot] = ld_u32(pl + (j + HS_PF) * DX, lo_l);          // step j + HS_PF (reads row padding past the end)
            fnw[slot] = ld_u32(pr + (j + HS_PF + 1) * DX, lo_n);
            __builtin_amdgcn_sched_barrier(0);
            agg_step<LPP>(st, mm, xr, sel_lo, sel_hi, p1p1, p2p2, sgpr(po + j * ostride) + pin_v(lo_o));
            __builtin_amdgcn_sched_barrier(0);
        }
        pl += 16 * DX; pr += 16 * DX; po += 16 * ostride;
    };
    group();
    for (int gi = 1; gi < groups; ++gi) group();
}

// Block barrier that orders LDS traffic only.  __syncthreads() carries a workgroup fence, i.e. s_waitcnt vmcnt(0): at the
// end of a burst every wave would sit out the acknowledgement of its global stores (~10 us under this read load), 23
// times per sweep (0.3 ms per 16-frame launch).  The bursts only exchange data through LDS.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- horizontal scans split over a wave PAIR (AggArgs::hsplit) ----------------------------------------------------------
// A horizontal scan is a chain of `width` steps on which one wave issues ~102 instructions per step and nothing can be done
// in parallel along the row: alone on its SIMD it already uses every issue slot (1242 steps x ~445 clocks = 0.27 ms whatever
// the launch holds), and launches with few frames -- or D = 64, where the other directions are short -- wait for these chains.
// 40 of the 102 instructions do not depend on the recurrence at all: the matching cost (xor, popcount, pack).  In split mode a
// PRODUCER wave keeps the sliding census window and writes the packed costs of step t+1 into LDS while the CONSUMER wave of
// the pair runs the recurrence of step t on the costs it reads back: ~60 instructions per step on the chain instead of 102.
// The two sit on different SIMDs of the CU (waves of a workgroup are dealt over its four SIMDs); one s_barrier per step keeps
// them one step apart (double-buffered costs).  A 4-wave workgroup holds two pairs, i.e. 2 P rows instead of 4 P.
constexpr int kHsCostDwords = 2 * 2 * 64 * 4;   // per pair: [buffer][half][lane][4 dwords]

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <int LPP>
__device__ __forceinline__ v4u agg_step_c(uint32_t (&a)[8], uint32_t &mm, const uint32_t (&c)[8], uint32_t sel_lo, uint32_t sel_hi,
                                          uint32_t p1p1, uint32_t p2p2) {
    // agg_step with the matching costs handed in: c[i] = (C[d0+8+i] << 16) + C[d0+i], the pair agg_step builds from its popcounts;
    // returns the lane's 16 slab bytes (the caller stores them)
    const uint32_t mp2 = mm + p2p2;
    const uint32_t lo0 = perm(a[7], dpp_mov<DPP_ROW_SHR1>(a[7]), sel_lo);
    const uint32_t hi7 = perm(dpp_mov<DPP_ROW_SHL1>(a[0]), a[0], sel_hi);
    uint32_t n[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t lo = i == 0 ? lo0 : a[i - 1];
        const uint32_t hi = i == 7 ? hi7 : a[i + 1];
        uint32_t t = pk_min(lo, hi) + p1p1;
        t = pk_min3(t, a[i], mp2);
        n[i] = (t - mm) + c[i];   // both halves of t are >= m (see agg_step)
    }
    const v4u q = {perm(n[1], n[0], 0x06040200u), perm(n[3], n[2], 0x06040200u), perm(n[5], n[4], 0x06040200u), perm(n[7], n[6], 0x06040200u)};
    uint32_t x = pk_min(pk_min3(n[0], n[1], n[2]), pk_min3(n[3], n[4], pk_min3(n[5], n[6], n[7])));
    x = pk_min(x, __builtin_amdgcn_alignbit(x, x, 16));
    mm = group_allmin<LPP>(x);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = n[i];
    return q;
}

// producer of a pair: costs of every step t = 0 .. w-1 into buffer t & 1, one barrier after each, one more at the end (the consumer's last step)
template <int LPP, int DX>
__device__ __forceinline__ void hsplit_producer(const uint32_t *pl, unsigned lo_l, const uint32_t *pr, unsigned lo_r, int w, uint32_t *cost, int lane) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    auto emit = [&](int parity, const uint32_t (&c)[8]) {
        v4u *dst = reinterpret_cast<v4u *>(cost) + parity * 128 + lane;
        dst[0] = v4u{c[0], c[1], c[2], c[3]};
        dst[64] = v4u{c[4], c[5], c[6], c[7]};
        lds_barrier();
    };
    const int groups = w / 16;
    if (groups > 0) {
        uint32_t win[16], ffl[HS_PF], fnw[HS_PF];
        ld_u32x16(pr, lo_r, win);                          // window of step 0
        unsigned lo_n = lo_r + (DX > 0 ? 15u * 4u : 0u);   // the element that enters the window (see hscan_sliding)
#pragma unroll
        for (int q = 0; q < HS_PF; ++q) {
            ffl[q] = ld_u32(pl + q * DX, lo_l);
            fnw[q] = ld_u32(pr + (q + 1) * DX, lo_n);
        }
        auto group = [&]() {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int slot = j % HS_PF;
                const uint32_t f = ffl[slot];
                uint32_t c[8];   // c[i] = (C[d0+8+i] << 16) + C[d0+i]: logical window slot k of this sub-step is register (k +- j) & 15
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    c[i] = ((uint32_t)__builtin_popcount(f ^ win[(DX > 0 ? 7 - i + j : 7 - i - j + 16) & 15]) << 16) +
                           (uint32_t)__builtin_popcount(f ^ win[(DX > 0 ? 15 - i + j : 15 - i - j + 16) & 15]);
                win[(DX > 0 ? j : 15 - j) & 15] = fnw[slot];
                ffl[slot] = ld_u32(pl + (j + HS_PF) * DX, lo_l);          // step j + HS_PF (reads row padding past the end)
                fnw[slot] = ld_u32(pr + (j + HS_PF + 1) * DX, lo_n);
                emit(j & 1, c);
            }
            pl += 16 * DX; pr += 16 * DX;
        };
        group();
        for (int gi = 1; gi < groups; ++gi) group();
    }
    for (int t = groups * 16; t < w; ++t) {   // the last w % 16 steps: plain loads
        CensusRegs cr;
        load_census(pl, lo_l, pr, lo_r, cr);
        uint32_t c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            c[i] = ((uint32_t)__builtin_popcount(cr.fl ^ cr.r[7 - i]) << 16) + (uint32_t)__builtin_popcount(cr.fl ^ cr.r[15 - i]);
        emit(t & 1, c);
        pl += DX; pr += DX;
    }
    lds_barrier();
}

// lanes 32..63 of `a` <-> lanes 0..31 of `b` (gfx950's v_permlane32_swap): a = {a.lo, b.lo}, b = {a.hi, b.hi}
// (gfx950 only, like pk_min3: any other --offload-arch stops here instead of failing in the assembler)
__device__ __forceinline__ void swap_halves(uint32_t &a, uint32_t &b) {
#if defined(__gfx950__)
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
#elif defined(__HIP_DEVICE_COMPILE__)
#error "sgm_kernels.hip is written for gfx950 (v_permlane32_swap_b32)"
#endif
}

// consumer of a pair.  D = 64 (LPP = 4): a pixel is 64 bytes, half a 128-byte line, and a step's store would write 16 rows x 64 B; the stores of TWO
// steps are regrouped instead (four lane-half swaps) so that each instruction writes whole lines: rows 0-7 of both steps, then rows 8-15 (lo_a / lo_b:
// this lane's byte offsets in the two stores, from the LOWER-x pixel of the step pair).  Measured with a timing build that wrote whole KB per store:
// aggregate 0.545 -> 0.499 ms at 1242x375 D=64 P=4, and this form reaches it (0.503); D >= 128 pixels are whole lines already and gain nothing
// (profiles/r04_hsplit.txt).
template <int LPP, int DX>
__device__ __forceinline__ void hsplit_consumer(uint32_t (&st)[8], uint32_t &mm, uint8_t *po, unsigned lo_o, unsigned lo_a, unsigned lo_b, int D, int w,
                                                const uint32_t *cost, int lane, uint32_t sel_lo, uint32_t sel_hi, uint32_t p1p1, uint32_t p2p2) {
    const ptrdiff_t ostride = (ptrdiff_t)DX * D;
    auto step = [&](int parity) {
        const v4u *src = reinterpret_cast<const v4u *>(cost) + parity * 128 + lane;
        const v4u c0 = src[0], c1 = src[64];
        const uint32_t c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const v4u q = agg_step_c<LPP>(st, mm, c, sel_lo, sel_hi, p1p1, p2p2);
        lds_barrier();
        return q;
    };
    lds_barrier();   // the costs of step 0 are in buffer 0
    int t = 0;
    for (; t + 1 < w; t += 2) {
        if constexpr (LPP == 4) {
            const v4u s0 = step(0), s1 = step(1);
            uint32_t u0[4] = {s0.x, s0.y, s0.z, s0.w}, u1[4] = {s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) swap_halves(u0[i], u1[i]);
            const v4u q0 = {u0[0], u0[1], u0[2], u0[3]}, q1 = {u1[0], u1[1], u1[2], u1[3]};
            uint8_t *pp = DX > 0 ? po : po + ostride;   // the pair's lower-x pixel
            __builtin_nontemporal_store(q0, (CART_GLOBAL v4u *)(sgpr(pp) + pin_v(lo_a)));
            __builtin_nontemporal_store(q1, (CART_GLOBAL v4u *)(sgpr(pp) + pin_v(lo_b)));
        } else {
            const v4u q0 = step(0);
            __builtin_nontemporal_store(q0, (CART_GLOBAL v4u *)(sgpr(po) + pin_v(lo_o)));
            const v4u q1 = step(1);
            __builtin_nontemporal_store(q1, (CART_GLOBAL v4u *)(sgpr(po + ostride) + pin_v(lo_o)));
        }
        po += 2 * ostride;
    }
    if (t < w) {
        const v4u q = step(0);
        __builtin_nontemporal_store(q, (CART_GLOBAL v4u *)(sgpr(po) + pin_v(lo_o)));
    }
}

// prefetch depth of the vertical / diagonal scans in steps (see the note at the loop): 4 at D = 64 (-10 %), 2 elsewhere (flat)
template <int LPP> constexpr int v_depth() { return LPP == 4 ? 4 : 2; }

constexpr int kAggWaves = 4;   // waves per workgroup: nothing in the kernel is shared between waves (1-2: 1.85 instead of 1.58 ms at the headline; 8: slower but at D=64)
// HS: the launch runs its horizontal scans as producer / consumer wave pairs (hsplit_*); a separate instantiation, so that the plain launch keeps its
// 70 VGPRs (7 waves per SIMD) and the split one gets the registers its producer needs without spilling
template <int LPP, bool HS = false>
__global__ __launch_bounds__(64 * kAggWaves, (LPP >= 8 && !HS) ? 6 : 4) void aggregate_kernel(AggArgs a) {
    using WN = Win<LPP>;
    constexpr int P = WN::P;
    constexpr int LINES_PER_BLOCK = kAggWaves * P;
    __shared__ uint32_t s_win[kAggWaves][2][WN::BUF];
    __shared__ __attribute__((aligned(16))) uint32_t s_cost[HS ? kAggWaves / 2 : 1][HS ? kHsCostDwords : 4];   // split horizontal scans: the pairs' cost buffers
    const Geometry &g = a.g;
    // 1-D grid, direction-major: [dir][frame][line group].  The horizontal directions come first so that
    // their W-step serial scans of EVERY frame start at once; the H-step scans fill in behind them.
    // XCD placement (speed only, never correctness): workgroups are dealt round-robin over the 8 XCDs, each with an L2 of its
    // own, and every direction re-reads its frame's census planes (4.2 MB per frame).  With n_frames a multiple of 8 the
    // grid is decoded per XCD (xcd_placement() below): XCD x works on frames x, x + 8, ... in the same direction-major order,
    // so that a frame's planes are fetched into ONE L2 instead of all eight.
    keep_f16_denormals();
    int bid = (int)blockIdx.x, nfr = a.n_frames, frame0 = 0, fstep = 1;
    if (a.xcd_frames) { frame0 = bid & 7; bid >>= 3; nfr = a.n_frames >> 3; fstep = 8; }
    int di = 0;
    for (int i = 1; i < a.ndirs; ++i)
        if (bid >= a.dirs[i].blk0 * nfr) di = i;
    const int dx = a.dirs[di].dx, dy = a.dirs[di].dy;
    const bool hsplit = HS && dy == 0;   // this workgroup runs two producer / consumer pairs on 2 P rows (launch_aggregate counted its blocks that way)
    const int lpb = hsplit ? 2 * P : LINES_PER_BLOCK;
    const int nblk = (a.dirs[di].nlines + lpb - 1) / lpb;
    const int rb = bid - a.dirs[di].blk0 * nfr;
    const int frame = frame0 + fstep * (rb / nblk);
    const int bl = rb - (rb / nblk) * nblk;   // block inside the frame's share of this direction
    const int lane = threadIdx.x & 63, wid = uniform((int)(threadIdx.x >> 6));
    const int gl = lane % LPP, pg = lane / LPP;  // lane inside the pixel's lane group, pixel group inside the wave
    const int line0 = bl * lpb + (hsplit ? wid >> 1 : wid) * P;  // wave-uniform
    const int line = line0 + pg;
    const int nlines = a.dirs[di].nlines;
    if (line0 >= nlines && !hsplit) return;  // whole wave idle (a split-scan workgroup keeps all four waves: they meet at a barrier every step)
    const int d0 = gl * 16;
    const uint32_t p1p1 = (uint32_t)g.p1 * 0x10001u, p2p2 = (uint32_t)g.p2 * 0x10001u;
    // selectors of the two stitching v_perm: 0x0d bytes inject 0xFFFF where d-1 / d+1 leave [0, D)
    const uint32_t sel_lo = gl == 0 ? 0x05040d0du : 0x05040302u;
    const uint32_t sel_hi = gl == LPP - 1 ? 0x0d0d0302u : 0x05040302u;

    uint32_t st[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = 0;
    uint32_t mm = 0;
    CensusRegs ca, cb;

    // NOTE on the loop shapes below.  vmcnt retires in issue order and counts stores too, so a load issued
    // AFTER a slab store cannot be consumed before that store has been written back (>1 us under write
    // pressure).  Each step therefore issues the loads of step t+2 BEFORE its own store, consumes the loads
    // of step t+1 after it, and keeps every VMEM instruction of the main loops unconditional (prefetches past
    // the last step read valid padding / slack) so that the compiler can use exact counted vmcnt waits.
    if (dy == 0) {
        // ---- horizontal scans: the wave's pixels sit on P different rows, nothing to share; per-lane loads.
        // These waves carry the longest dependency chain of the launch: let them win VALU arbitration.
        __builtin_amdgcn_s_setprio(3);
        if constexpr (HS) {
            // rows past the image (the last wave pair of a direction) clone the pair's last valid row: same reads, same bytes to the same cells,
            // and every wave that entered reaches every barrier.  Two waves then store to the same slab cells without ordering: benign only because
            // both compute the same state from the same census rows -- held by tests/test_gpu_parity.py::test_split_horizontal_scans_equal_plain_ones,
            // whose heights leave the last workgroup a partial pair and an idle pair (every slab row, the cloned ones included, against the oracle)
            const int l0 = min(line0, nlines - 1), pgv = min(pg, nlines - l0 - 1);   // (a pair wholly past the image clones the last row)
            const int y0s = a.dirs[di].jmin + l0, xs = dx > 0 ? 0 : g.w - 1;
            const uint32_t *pls = a.cen_l + uniform((ptrdiff_t)frame * (ptrdiff_t)g.census_elems + (ptrdiff_t)y0s * g.cpitch + g.cpadl + xs);
            const uint32_t *prs = a.cen_r + uniform((ptrdiff_t)frame * (ptrdiff_t)g.census_elems + (ptrdiff_t)y0s * g.cpitch + g.cpadl + xs - g.min_disp - (WN::D - 1));
            uint8_t *pos = a.slabs.frame[frame] + uniform((ptrdiff_t)a.dirs[di].path * (ptrdiff_t)g.slab_bytes + ((ptrdiff_t)y0s * g.w + xs) * g.D);
            const unsigned so_l = (unsigned)pgv * g.cpitch * 4u, so_r = so_l + (unsigned)(WN::D - 16 - d0) * 4u;
            const unsigned so_o = (unsigned)pgv * g.w * g.D + d0;
            // paired-step stores of the D = 64 consumer (hsplit_consumer): lanes 0-31 store the pair's first step, lanes 32-63 its second, of rows
            // pg & 7 (first store) and 8 + (pg & 7) (second); offsets count from the pair's lower-x pixel
            const int rlim = nlines - l0 - 1, r8 = (lane & 31) / LPP;
            const unsigned xo = ((dx > 0) == (lane >= 32)) ? (unsigned)g.D : 0u;
            const unsigned so_a = (unsigned)min(r8, rlim) * g.w * g.D + d0 + xo, so_b = (unsigned)min(r8 + 8, rlim) * g.w * g.D + d0 + xo;
            uint32_t *cost = &s_cost[wid >> 1][0];
            // D = 64: the producer has ~30 % slack per step; below the consumers' priority it stops taking issue slots from the consumer of ANOTHER pair
            // on its SIMD (aggregate 0.505 -> 0.498 ms at 16 frames, 0.305 -> 0.296 at 8).  At D = 256 the same costs 7 % (0.491 -> 0.524, 6 frames): left at 3.
            if constexpr (LPP == 4) { if ((wid & 1) == 0) __builtin_amdgcn_s_setprio(1); }
            if ((wid & 1) == 0) {
                if (dx > 0) hsplit_producer<LPP, 1>(pls, so_l, prs, so_r, g.w, cost, lane);
                else hsplit_producer<LPP, -1>(pls, so_l, prs, so_r, g.w, cost, lane);
            } else {
                if (dx > 0) hsplit_consumer<LPP, 1>(st, mm, pos, so_o, so_a, so_b, g.D, g.w, cost, lane, sel_lo, sel_hi, p1p1, p2p2);
                else hsplit_consumer<LPP, -1>(st, mm, pos, so_o, so_a, so_b, g.D, g.w, cost, lane, sel_lo, sel_hi, p1p1, p2p2);
            }
            return;
        }
        if (line >= nlines) return;
        const int y0r = a.dirs[di].jmin + line0;          // row of the wave's first line (uniform)
        const int x = dx > 0 ? 0 : g.w - 1, t1 = g.w;
        // uniform bases + non-negative per-lane element offsets
        const uint32_t *pl_u = a.cen_l + uniform((ptrdiff_t)frame * (ptrdiff_t)g.census_elems + (ptrdiff_t)y0r * g.cpitch + g.cpadl + x);
        const uint32_t *pr_u = a.cen_r + uniform((ptrdiff_t)frame * (ptrdiff_t)g.census_elems + (ptrdiff_t)y0r * g.cpitch + g.cpadl + x - g.min_disp - (WN::D - 1));
        uint8_t *po_u = a.slabs.frame[frame] + uniform((ptrdiff_t)a.dirs[di].path * (ptrdiff_t)g.slab_bytes + ((ptrdiff_t)y0r * g.w + x) * g.D);
        unsigned lo_l = (unsigned)pg * g.cpitch * 4u, lo_r = lo_l + (unsigned)(WN::D - 16 - d0) * 4u;  // bytes
        unsigned lo_o = (unsigned)pg * g.w * g.D + d0;
        const ptrdiff_t cstride = dx, ostride = (ptrdiff_t)dx * g.D;
        const uint32_t *pl = pl_u, *pr = pr_u;
        uint8_t *po = po_u;
        // full 16-step groups with the sliding window; the last w % 16 steps (and images narrower than a group) through
        // the reloading loop below, which can start anywhere
        const int groups = t1 / 16;
        if (groups > 0) {
            if (dx > 0) hscan_sliding<LPP, 1>(st, mm, pl, lo_l, pr, lo_r, po, lo_o, ostride, groups, sel_lo, sel_hi, p1p1, p2p2);
            else hscan_sliding<LPP, -1>(st, mm, pl, lo_l, pr, lo_r, po, lo_o, ostride, groups, sel_lo, sel_hi, p1p1, p2p2);
        }
        uint32_t xr[16];
        load_census(pl, lo_l, pr, lo_r, ca);
        load_census(pl + cstride, lo_l, pr + cstride, lo_r, cb);
        int t = groups * 16;
        for (; t + 1 < t1; t += 2) {
            agg_xor(ca, xr);
            __builtin_amdgcn_sched_barrier(0);
            load_census(pl + 2 * cstride, lo_l, pr + 2 * cstride, lo_r, ca);  // step t+2 (reads row padding past the end)
            __builtin_amdgcn_sched_barrier(0);
            agg_step<LPP>(st, mm, xr, sel_lo, sel_hi, p1p1, p2p2, sgpr(po) + pin_v(lo_o));
            __builtin_amdgcn_sched_barrier(0);
            agg_xor(cb, xr);
            __builtin_amdgcn_sched_barrier(0);
            load_census(pl + 3 * cstride, lo_l, pr + 3 * cstride, lo_r, cb);  // step t+3
            __builtin_amdgcn_sched_barrier(0);
            agg_step<LPP>(st, mm, xr, sel_lo, sel_hi, p1p1, p2p2, sgpr(po + ostride) + pin_v(lo_o));
            __builtin_amdgcn_sched_barrier(0);
            pl += 2 * cstride; pr += 2 * cstride; po += 2 * ostride;
        }
        if (t < t1) {
            agg_xor(ca, xr);
            agg_step<LPP>(st, mm, xr, sel_lo, sel_hi, p1p1, p2p2, sgpr(po) + pin_v(lo_o));
        }
        return;
    }

    // ---- vertical / diagonal scans: lines are indexed by their (skewed) entry column j
    // A wave that holds fewer than P lines (the last one of a direction: 1242 columns = 77 x 16 + 10 at D = 64,
    // 155 x 8 + 2 at D = 128) lets its surplus lane groups CLONE its last valid line: same reads, same arithmetic, the
    // same bytes stored to the same cells.  (Sending such waves through the synchronous ragged path instead made one wave
    // per frame and direction walk all its steps at memory latency -- 0.26 ms on an idle GPU, ~0.6 ms under load, which
    // was the whole launch time at D = 64 / 4 paths.)
    const int nv = min(P, nlines - line0);  // valid lines in this wave (wave-uniform)
    const int pgv = min(pg, nv - 1);        // the line of the wave this lane group works on
    const int j = a.dirs[di].jmin + line0 + pgv;
    const int ys = dy > 0 ? 0 : g.h - 1;
    int t0, t1;  // this lane group's active steps
    if (dx > 0) { t0 = max(0, -j); t1 = min(g.h, g.w - j); }
    else if (dx < 0) { t0 = max(0, j - g.w + 1); t1 = min(g.h, j + 1); }
    else { t0 = 0; t1 = g.h; }
    // wave-uniform ranges: [tb, te) = union of the wave's (adjacent) lines, [tm0, tm1) = steps on which
    // every lane group of the wave is active
    const int jf = a.dirs[di].jmin + line0, jl = jf + nv - 1;
    int tb, te, tm0, tm1;
    if (dx > 0) { tb = max(0, -jl); te = min(g.h, g.w - jf); tm0 = max(0, -jf); tm1 = min(g.h, g.w - jl); }
    else if (dx < 0) { tb = max(0, jf - g.w + 1); te = min(g.h, jl + 1); tm0 = max(0, jl - g.w + 1); tm1 = min(g.h, jf + 1); }
    else { tb = 0; te = g.h; tm0 = 0; tm1 = g.h; }
    if (tb >= te) return;
    if (tm0 >= tm1) { tm0 = te; tm1 = te; }  // no step with every line active: everything through the ragged path

    // cooperative window load + this lane's 16 features (window dwords pg + D-16 - 16*gl + k), see Win / WinLane
    WinLane<LPP> wlane;
    wlane.init(lane);
    unsigned (&goff)[WN::NLD] = wlane.goff;
    const int (&lslot)[WN::NLD] = wlane.lslot;
    const int rbase = gl * WN::RS + pgv;
    uint32_t *buf0 = &s_win[wid][0][0], *buf1 = &s_win[wid][1][0];

    // pointers as a function of the step t
    const ptrdiff_t cstride = (ptrdiff_t)dy * g.cpitch + dx;
    const ptrdiff_t ostride = ((ptrdiff_t)dy * g.w + dx) * g.D;
    // uniform bases (line jf = first line of the wave) + per-lane offsets (pg = this lane's line inside the wave)
    const ptrdiff_t cen_off = (ptrdiff_t)frame * (ptrdiff_t)g.census_elems + (ptrdiff_t)ys * g.cpitch + g.cpadl + jf;
    const uint32_t *pw_base = a.cen_r + uniform(cen_off - g.min_disp - (WN::D - 1));  // window start at t = 0
    const uint32_t *pl_u = a.cen_l + uniform(cen_off);
    uint8_t *po_u = a.slabs.frame[frame] + uniform((ptrdiff_t)a.dirs[di].path * (ptrdiff_t)g.slab_bytes + ((ptrdiff_t)ys * g.w + jf) * g.D);
    unsigned lo_l = (unsigned)pgv * 4u, lo_o = (unsigned)(pgv * WN::D + d0);  // bytes

    // ragged start / end of diagonal lines (and waves with invalid lines): simple, fully synchronous steps
    auto ragged = [&](int ta, int tz) {
        for (int t = ta; t < tz; ++t) {
            const uint32_t *pw = pw_base + t * cstride;
#pragma unroll
            for (int i = 0; i < WN::NLD; ++i) buf0[lslot[i]] = ld_u32(pw, goff[i]);
            if (t >= t0 && t < t1) {
                uint32_t xr[16];
                ca.fl = ld_u32(pl_u + t * cstride, lo_l);
                win_read<LPP>(buf0, rbase, ca.r);
                agg_xor(ca, xr);
                agg_step<LPP>(st, mm, xr, sel_lo, sel_hi, p1p1, p2p2, sgpr(po_u + t * ostride) + pin_v(lo_o));
            }
        }
    };
    ragged(tb, tm0);
    if (tm0 < tm1) {
        const uint32_t *pw = pw_base + tm0 * cstride, *pl = pl_u + tm0 * cstride;
        uint8_t *po = po_u + tm0 * ostride;
        // K register sets of prefetched windows: the loads of step t+K are issued at the start of step t.  vmcnt retires in
        // issue order and counts stores, so the loads consumed at step t wait for the slab stores issued up to step t-K:
        // with K = 2 every step of a wave sat out the acknowledgement of a two-step-old store (~2 us under write pressure,
        // i.e. ~1 us per step however few waves shared the SIMD) -- invisible at D = 128 / 8 paths, where enough waves per
        // SIMD cover it, but 0.4 ms per launch at D = 64 / 4 paths, whose vertical waves finish last on their own.
        constexpr int K = v_depth<LPP>();
        static_assert(K % 2 == 0 && K >= 2, "the LDS window buffers alternate");
        uint32_t gs[K][WN::NLD], fs[K];
#pragma unroll
        for (int i = 0; i < WN::NLD; ++i) buf0[lslot[i]] = ld_u32(pw, goff[i]);   // step tm0 straight into its LDS buffer
        fs[0] = ld_u32(pl, lo_l);
#pragma unroll
        for (int j = 1; j < K; ++j) {
#pragma unroll
            for (int i = 0; i < WN::NLD; ++i) gs[j][i] = ld_u32(pw + j * cstride, goff[i]);
            fs[j] = ld_u32(pl + j * cstride, lo_l);
        }
        uint32_t xr[16];
        int t = tm0;
        // sub-step J of a K-step trip: buffer J&1 holds the window of step t+J, fs[J] its left feature, set J is free
        auto sub = [&](auto jc, auto reload) {
            constexpr int J = decltype(jc)::value;
            uint32_t *cur = (J & 1) ? buf1 : buf0, *nxt = (J & 1) ? buf0 : buf1;
            ca.fl = fs[J];
            if constexpr (decltype(reload)::value) {
#pragma unroll
                for (int i = 0; i < WN::NLD; ++i) gs[J][i] = ld_u32(pw + (J + K) * cstride, goff[i]);  // step t+J+K, issued before this step's store
                fs[J] = ld_u32(pl + (J + K) * cstride, lo_l);
            }
            __builtin_amdgcn_sched_barrier(0);
            win_read<LPP>(cur, rbase, ca.r);
            agg_xor(ca, xr);
            agg_step<LPP>(st, mm, xr, sel_lo, sel_hi, p1p1, p2p2, sgpr(po + J * ostride) + pin_v(lo_o));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < WN::NLD; ++i) nxt[lslot[i]] = gs[(J + 1) % K][i];   // window of step t+J+1
        };
        for (; t + K <= tm1; t += K) {
            sub(std::integral_constant<int, 0>{}, std::true_type{});
            sub(std::integral_constant<int, 1>{}, std::true_type{});
            if constexpr (K > 2) {
                sub(std::integral_constant<int, 2>{}, std::true_type{});
                sub(std::integral_constant<int, 3>{}, std::true_type{});
            }
            if constexpr (K > 4) {
                sub(std::integral_constant<int, 4>{}, std::true_type{});
                sub(std::integral_constant<int, 5>{}, std::true_type{});
            }
            pw += K * cstride; pl += K * cstride; po += K * ostride;
        }
        // the last tm1 - t < K steps: their windows are already in flight
        if (t < tm1) sub(std::integral_constant<int, 0>{}, std::false_type{});
        if (t + 1 < tm1) sub(std::integral_constant<int, 1>{}, std::false_type{});
        if constexpr (K > 2) {
            if (t + 2 < tm1) sub(std::integral_constant<int, 2>{}, std::false_type{});
        }
        if constexpr (K > 4) {
            if (t + 3 < tm1) sub(std::integral_constant<int, 3>{}, std::false_type{});
            if (t + 4 < tm1) sub(std::integral_constant<int, 4>{}, std::false_type{});
        }
    }
    ragged(tm1, te);
}

int agg_lines_per_block(int D) { return 64 * kAggWaves / (D / 16); }

// XCD-aware grid decode of the aggregation launch and the fused sweep: on when the launch's frames divide over the 8 XCDs and
// one frame's two census planes are of the order of an XCD's 4 MB L2.  Measured A/B on one box (profiles/r03_xcd.txt):
// 1242x375 D=128 P=8 aggregate 1.607-1.613 vs 1.619-1.631 ms (+0.7 % pairs/s), D=256 P=4 aggregate 1.21-1.23 vs 1.25-1.26,
// D=64 P=4 level; at 1920x1080 (17 MB of census per frame) it costs the aggregate 7 % (7.83 vs 7.32 ms per 8 frames): off there.
bool xcd_placement(const Geometry &g, int n_frames) {
    return n_frames > 0 && (n_frames & 7) == 0 && g.census_elems * 8 <= (size_t)(8u << 20);
}

// 4-wave workgroups of the aggregation launch allowed per CU at a time (0 = no cap: 7 fit).  One row per measured case
// (ms per launch, residency 7 / 5 / 4 / 3 / 2; round 2, one box per row -- DESIGN.md 4.1):
//   ndirs <= 4 (half the work is W-step horizontal scans)
//     D = 64   0.52-0.63 / 0.52-0.57 / - / 0.49-0.51 / 0.52; with the second stream on: 2 per CU 1.04-1.10 ms per step, 3 per CU 1.13-1.18  -> 2
//     D = 128  1.00-1.07 / - / 0.86 / 0.82 / -; with the second stream: 3 per CU 1.68-1.75, 2 per CU 1.72-1.77                                  -> 3
//     D = 256  1.215 / - / 1.19 / - / 1.23 (3 directions + fused sweep)                                                                          -> 4
//   ndirs 7-8
//     D = 256  3.51 / - / 3.29 / - / 3.56 (1080p, 4 frames)                                                                                      -> 4
//     D <= 128, fewer than 16 frames (the frame loop's coalesced groups): 4.73-4.88 k pairs/s against 4.54-4.61 k                              -> 4
//     D <= 128, 16 frames (the headline): 1.574 / 1.547 / 1.540 / 1.60 / 1.69 alone, but beside the second stream's plane kernels the cap
//       costs 1 % (1.61-1.68 against 1.59-1.63)                                                                                                  -> none
//   split horizontal scans (round 4, pairs/s at 2 / 3 / 4 per CU): D = 256 P = 4, 6 frames 5 128 / 5 382 / 5 072, 8 frames 5 134 / 5 360 / 5 332, 12 frames
//     5 428 / 5 542 / 5 565 -> 3;  D = 64 P = 4, 16 frames 15 407 / 14 984 / 14 227, 8 frames 12 853 / 12 337 / 11 789 -> 2 (as without the split)
int agg_residency_cap(int ndirs, int D, int n_frames, bool hsplit) {
    if (ndirs <= 4) return D <= 64 ? 2 : D <= 128 ? 3 : hsplit ? 3 : 4;
    return (D >= 256 || n_frames < 16) ? 4 : 0;
}

// When the horizontal scans run as producer / consumer wave pairs (hsplit_*).  They pay where the launch waits for its W-step chains --
// few directions beside them, or few frames -- and cost 1-2 % where the launch has enough other work (profiles/r04_hsplit.txt; aggregate ms per
// launch, plain / split, means of three alternating runs, 1242x375 unless noted):
//   D = 64  P = 4:  16 frames 0.598 / 0.548   8 frames 0.482 / 0.339   4 frames 0.292 / 0.248   (with the consumer's whole-line stores: 0.503 / 0.305 / 0.244)
//   D = 256 P = 4:  16 frames 1.210 / 1.230   12 frames 0.984 / 0.962   8 frames 0.796 / 0.651   6 frames 0.701 / 0.553   1920x1080, 4 frames 1.493 / 1.448
//   D = 64  P = 8:  16 frames 0.957 / 0.861   8 frames 0.686 / 0.504   (pairs/s 9 253 -> 9 870, 7 353 -> 8 856)
//   D = 128 P = 8:  16 frames 1.453 / 1.456   12 frames 1.147 / 1.166   8 frames 0.782 / 0.785   4 frames 0.473 / 0.451      D = 128 P = 4, 16 frames 0.844 / 0.862
bool agg_hsplit(const Geometry &g, int ndirs, int n_frames) {
    if (g.D == 64) return true;   // whatever else the launch holds: a D = 64 pixel is half a line, and only the split consumer stores whole lines
    return ndirs <= 4 ? n_frames <= 12 : n_frames < 8;
}

void launch_aggregate(const AggArgs &a_in, int n_frames, hipStream_t s) {
    AggArgs a = a_in;
    a.n_frames = n_frames;
    a.xcd_frames = xcd_placement(a.g, n_frames) ? 1 : 0;
    a.hsplit = agg_hsplit(a.g, a.ndirs, n_frames) ? 1 : 0;
    {   // blocks per direction: 4 P lines each, 2 P for the horizontal directions in split mode
        const int lpb_full = agg_lines_per_block(a.g.D);
        int blk = 0;
        for (int i = 0; i < a.ndirs; ++i) {
            const int l = a.hsplit && a.dirs[i].dy == 0 ? lpb_full / 2 : lpb_full;
            a.dirs[i].blk0 = blk;
            blk += (a.dirs[i].nlines + l - 1) / l;
        }
        a.blocks_per_frame = blk;
    }
    dim3 grid(a.blocks_per_frame * n_frames), block(64 * kAggWaves);
    // A cap on the workgroups resident per CU (agg_residency_cap above), enforced with unused dynamic LDS: the others are
    // dispatched as slots free up.  With everything resident at once (7 waves per SIMD fit) the CUs that hold the W-step
    // horizontal scans end up with as many of the short vertical / diagonal scans as the others and finish last; with 2-4
    // workgroups per CU the dispatcher hands the short scans to whichever CU is free, the long scans keep most of their SIMD,
    // and the census planes the directions re-read stay in L2.  Smaller workgroups are slower (two waves or one: the headline's
    // launch 1.85 instead of 1.58 ms), eight-wave ones too except at D=64.
    constexpr int kLdsPerCu = 160 * 1024, kLdsGranule = 1280;
    const int resident = agg_residency_cap(a.ndirs, a.g.D, n_frames, a.hsplit != 0) * 4 / kAggWaves;   // the rule counts 4-wave workgroups
    const int lpp = a.g.D / 16;
    const size_t static_lds = sizeof(uint32_t) * (kAggWaves * 2 * (lpp == 4 ? Win<4>::BUF : lpp == 8 ? Win<8>::BUF : Win<16>::BUF) + (a.hsplit ? (kAggWaves / 2) * kHsCostDwords : 4));
    // (never more than 64 KB per workgroup in all, the limit that needs no opt-in: two of those per CU are still two)
    const size_t pad = resident ? std::min<size_t>(kLdsPerCu / resident - kLdsGranule, 64 * 1024) - static_lds : 0;
    if (a.hsplit) {
        switch (a.g.D) {
            case 64: hipLaunchKernelGGL((aggregate_kernel<4, true>), grid, block, pad, s, a); break;
            case 128: hipLaunchKernelGGL((aggregate_kernel<8, true>), grid, block, pad, s, a); break;
            default: hipLaunchKernelGGL((aggregate_kernel<16, true>), grid, block, pad, s, a); break;
        }
        return;
    }
    switch (a.g.D) {
        case 64: hipLaunchKernelGGL(aggregate_kernel<4>, grid, block, pad, s, a); break;
        case 128: hipLaunchKernelGGL(aggregate_kernel<8>, grid, block, pad, s, a); break;
        default: hipLaunchKernelGGL(aggregate_kernel<16>, grid, block, pad, s, a); break;
    }
}


// ------------------------------------------------------------------ winner takes all
// Block = 64 pixels of one row; a pixel is owned by LPP = D/16 lanes, 16 disparities per lane as 8
// packed u16 pairs.  Per path one 16-byte non-temporal load per lane, the
// bytes are widened by v_perm_b32 and summed with v_pk_add_u16 (1 VALU op per cell and path).
//   * the slab bytes of a lane's 16 disparities arrive in the aggregation kernel's split-halves order, whose even /
//     odd bytes are natural adjacent disparity pairs: one v_and or v_perm plus a plain add per two cells;
//   * argmin (ties -> lowest d, oracle S5): packed keys S*16 + local index, packed min tree, then one
//     32-bit key (S<<16 | d) per lane reduced over the pixel's lanes by DPP;
//   * uniqueness: (float)S*u >= (float)best is monotone in S, so it equals S >= T for the integer
//     threshold T = min{s : (float)s*u >= (float)best}; the pixel is unique iff every S[d] < T lies within
//     |d - best| <= 1, i.e. iff sum_d max(T-S[d],0) equals the same sum over the three neighbours;
//   * the summed costs of the tile live in LDS as u16 [64][D]: sub-pixel neighbours and the right-view
//     diagonal minima S(p+d, d) (oracle S6) come from there; per-tile right minima are merged across tiles
//     with one packed atomicMin per right pixel and tile.
// The fused sweep's right-view rows (one u32 key per right pixel of a block row) can be indexed through rv_slot: one pad per
// 16 entries.  The lanes that own one pixel hold disparity chunks 16 apart, so their candidates for one `da` land 16 entries
// apart -- on TWO of the 32 LDS banks without the pad (8-way conflicts on every ds_min_u32 at D = 256), on 16 different
// banks with it.  Used where it measured faster: D = 256 / 4 paths (sweep 1.58 -> 1.43 ms per 16 pairs).  The 8-path sweeps
// sit at the 168-VGPR limit of three waves per SIMD and the 16 slot addresses spill (D = 256 / 8 paths: 2.5 -> 4.0 ms), and
// the two-kernel WTA did not move (2.03 ms at D = 256 with or without).
__host__ __device__ constexpr int rv_slot(int i) { return i + (i >> 4); }
__host__ __device__ constexpr int rv_size(int n) { return ((n + (n >> 4) + 1) + 3) & ~3; }   // slots for n entries (+ a spare), multiple of 4
// Slots of one right-view row of a fused-sweep block (cols + D - 1 entries + a spare, padded or not), a multiple of 8: in
// the partial buffer a slot is ONE u16 -- (S << log2(cols)) | (d mod cols), 0xffff = empty -- and a burst packs 8 of them
// per lane.  For entry e of a block the candidates are the block's columns xl = 0..cols-1 with d = D-1-e + xl, so d mod cols
// identifies the column and rv_key32 gives the (S << 16 | d) key back; S <= 8 * 255 leaves 5 bits for cols = 32.
__host__ __device__ constexpr int rv_row_slots(int cols, int D, bool padded) { return ((padded ? rv_size(cols + D - 1) : cols + D) + 7) & ~7; }
__host__ __device__ constexpr uint32_t rv_key16(uint32_t key32, int cols) {   // cols = 16 or 32; 0xffffffff -> 0xffff
    return ((((key32 >> 16) << (cols == 32 ? 5 : 4)) | (key32 & (uint32_t)(cols - 1))) & 0xffffu);
}
__host__ __device__ constexpr uint32_t rv_key32(uint32_t key16, int e, int cols, int D) {   // key16 != 0xffff
    const int sh = cols == 32 ? 5 : 4, base = D - 1 - e;
    return ((key16 >> sh) << 16) | (uint32_t)(base + (((int)(key16 & (uint32_t)(cols - 1)) - base) & (cols - 1)));
}

__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}

// smallest s with (float)s*u >= (float)bc, clamped to 4095 (> any reachable cost sum, <= 8*255)
__host__ __device__ __forceinline__ uint32_t uniq_threshold(uint32_t bc, float u) {
    if (bc == 0) return 0;
    if (!(u > 0.f)) return 4095u;
    const float bcf = (float)bc;
    const float q = bcf / u;
    if (q > 4000.f) return 4095u;
    const int g = (int)q;
    int T = g + 3;
#pragma unroll
    for (int c = 2; c >= -2; --c) {
        const int v = g + c;
        if (v >= 0 && (float)v * u >= bcf) T = v;
    }
    return (uint32_t)(T < 4095 ? T : 4095);
}

// test access (cart_debug_uniq_table): the threshold of every best cost 0..2047 for one uniqueness ratio, computed by
// the device code the WTA kernels use, or by the same function compiled for the host
__global__ void uniq_table_kernel(float u, uint16_t *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2048) out[i] = (uint16_t)uniq_threshold((uint32_t)i, u);
}
void launch_uniq_table(float u, uint16_t *out_dev, hipStream_t s) { hipLaunchKernelGGL(uniq_table_kernel, dim3(8), dim3(256), 0, s, u, out_dev); }
void uniq_table_host(float u, uint16_t *out) {
    for (int i = 0; i < 2048; ++i) out[i] = (uint16_t)uniq_threshold((uint32_t)i, u);
}

struct WtaArgs {
    SlabTable slabs;
    uint16_t *wta_l;
    uint32_t *right_pk;
    Geometry g;
    const uint16_t *thr;             // integer uniqueness threshold by best cost, 2048 entries (uniq_threshold of every cost, built at engine create)
    int nslabs;
    int slab_idx[kMaxPaths];
};

__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}

// TOP2 = the S5 variant (CART_OPT_SPEC_S5_TOP2): uniqueness looks at the SECOND-best (cost, d) only -- the second-smallest
// (cost << 16 | d) key of the pixel -- instead of at every disparity.
template <int LPP, bool TOP2 = false>
__global__ __launch_bounds__(256) void wta_kernel(WtaArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t s_lds[];  // [kWtaTileX][DP]
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    constexpr int D = LPP * 16;
    constexpr int DP = D + 8;                 // LDS row pitch in u16: 16 B of padding spread the pixels' rows over the banks
    constexpr int PPP = 256 / LPP;            // pixels per pass
    constexpr int NPASS = kWtaTileX / PPP;    // 1 (D=64), 2 (D=128), 4 (D=256)
    uint16_t *wta_l = a.wta_l;
    uint32_t *right_pk = a.right_pk;
    const Geometry &g = a.g;
    const int x0 = blockIdx.x * kWtaTileX, y = blockIdx.y, frame = blockIdx.z;
    const uint8_t *slabs = a.slabs.frame[frame];   // this frame's path slabs
    const int grp = threadIdx.x / LPP, gl = threadIdx.x % LPP, d0 = gl * 16;
    // Right view: every lane min-reduces its 16 (S << 16 | d) keys into the tile's array indexed by p = x - d (ds_min_u32),
    // slot p - (x0 - (D-1)).  (Walking the tile's diagonals per right pixel instead -- 64 dependent LDS reads on 127 of
    // the 256 threads at D = 64 -- was over half of this kernel's VALU instructions and made the D = 64 variant VALU-bound.)
    __shared__ uint32_t s_rv[kWtaTileX + D];
    for (int i = threadIdx.x; i < kWtaTileX + D; i += 256) s_rv[i] = 0xffffffffu;
    __syncthreads();

    uint32_t pk_res[NPASS], tot_res[NPASS], thr_res[NPASS];   // TOP2: tot_res holds the pixel's second-smallest key
    constexpr bool PREFETCH = NPASS >= 4;
    v4u pf[PREFETCH ? 2 : 1][PREFETCH ? kMaxPaths : 1];
    auto issue_pass = [&](int pass, v4u (&dst)[PREFETCH ? kMaxPaths : 1]) {
        if constexpr (PREFETCH) {
            const int xcp = min(x0 + pass * PPP + grp, g.w - 1);
            const uint8_t *p = slabs + ((size_t)y * g.w + xcp) * D + d0;
#pragma unroll
            for (int r = 0; r < kMaxPaths; ++r)
                if (r < a.nslabs) dst[r] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p + (size_t)a.slab_idx[r] * g.slab_bytes));
        }
    };
    if constexpr (PREFETCH) issue_pass(0, pf[0]);
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int xl = pass * PPP + grp;
        const int xc = min(x0 + xl, g.w - 1);
        // Slab chunk order {0,8,1,9,...}: the even bytes of dword q are disparities (2q, 2q+1), the odd bytes (2q+8, 2q+9).
        // sm[q] = (S[d0+2q], S[d0+2q+1]), sm[4+q] = (S[d0+8+2q], S[d0+9+2q]): natural adjacent pairs.  Even bytes need one
        // v_and, odd bytes one v_perm; the accumulation is a plain 32-bit add (sums stay < 2^16 per half).
        uint32_t sm[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) sm[k] = 0;
        if constexpr (PREFETCH) {
            // D = 256: four passes per block and four blocks per CU (LDS) -- with every pass waiting for its own loads the
            // launch ran at memory latency (7.6 GB in 2.0 ms); the next pass's slab bytes are requested before this pass computes
            if (pass + 1 < NPASS) issue_pass(pass + 1, pf[(pass + 1) & 1]);
#pragma unroll
            for (int r = 0; r < kMaxPaths; ++r) {
                if (r < a.nslabs) {
                    const v4u v = pf[pass & 1][r];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        sm[q] += v[q] & 0x00ff00ffu;
                        sm[4 + q] += perm(0u, v[q], 0x0c030c01u);
                    }
                }
            }
        } else {
            const uint8_t *p = slabs + ((size_t)y * g.w + xc) * D + d0;
            for (int r = 0; r < a.nslabs; ++r) {
                const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p + (size_t)a.slab_idx[r] * g.slab_bytes));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    sm[q] += v[q] & 0x00ff00ffu;
                    sm[4 + q] += perm(0u, v[q], 0x0c030c01u);
                }
            }
        }
        v4u *dst = reinterpret_cast<v4u *>(s_lds + xl * DP + d0);  // LDS tile in natural disparity order
        dst[0] = v4u{sm[0], sm[1], sm[2], sm[3]};
        dst[1] = v4u{sm[4], sm[5], sm[6], sm[7]};
        if (x0 + xl < g.w) {   // columns past the image (clamped duplicates of the last one) have no right view
            uint32_t *rm = &s_rv[xl + D - 1 - d0];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int da = q < 4 ? 2 * q : 8 + 2 * (q - 4);   // local disparity of the low half of sm[q]
                atomicMin(rm - da, (sm[q] << 16) | (uint32_t)(d0 + da));
                atomicMin(rm - da - 1, (sm[q] & 0xffff0000u) | (uint32_t)(d0 + da + 1));
            }
        }
        // packed argmin keys: S*16 + local disparity index
        uint32_t key[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u16x2 kk = __builtin_bit_cast(u16x2, sm[k]) * (u16x2){16, 16} + (u16x2){(uint16_t)(2 * k), (uint16_t)(2 * k + 1)};
            key[k] = __builtin_bit_cast(uint32_t, kk);
        }
        uint32_t m = pk_min(pk_min(pk_min(key[0], key[1]), pk_min(key[2], key[3])), pk_min(pk_min(key[4], key[5]), pk_min(key[6], key[7])));
        m = pk_min(m, __builtin_amdgcn_alignbit(m, m, 16)) & 0xffffu;
        uint32_t pk = ((m >> 4) << 16) | (uint32_t)(d0 + (int)(m & 15u));
        uint32_t cand = 0;
        if constexpr (TOP2) {
            // the lane's two smallest 16-bit keys: a tournament on (min, second) pairs, both halves of the packed registers at once
            uint32_t lo[4], hi[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { lo[k] = pk_min(key[2 * k], key[2 * k + 1]); hi[k] = pk_max(key[2 * k], key[2 * k + 1]); }
            const uint32_t m01 = pk_min(lo[0], lo[1]), s01 = pk_min(pk_max(lo[0], lo[1]), pk_min(hi[0], hi[1]));
            const uint32_t m23 = pk_min(lo[2], lo[3]), s23 = pk_min(pk_max(lo[2], lo[3]), pk_min(hi[2], hi[3]));
            const uint32_t mm4 = pk_min(m01, m23), ss4 = pk_min(pk_max(m01, m23), pk_min(s01, s23));
            const uint32_t mL = mm4 & 0xffffu, mH = mm4 >> 16, sL = ss4 & 0xffffu, sH = ss4 >> 16;
            const uint32_t second = min(max(mL, mH), min(sL, sH));   // (the smaller of mL, mH is `m` above)
            const uint32_t second_full = ((second >> 4) << 16) | (uint32_t)(d0 + (int)(second & 15u));
            const uint32_t best_all = group_allmin<LPP>(pk);
            cand = pk == best_all ? second_full : pk;   // the lane that holds the pixel's best key offers its runner-up
            pk = best_all;
            cand = group_allmin<LPP>(cand);
        } else {
            pk = group_allmin<LPP>(pk);
        }
        // = uniq_threshold(best cost): one load from the engine's 4 KB table (L1-resident) instead of the float search -- a
        // division and five multiply-compares, ~45 VALU instructions per lane and pass, twice (here and where the pixel's
        // first lane decides), about a quarter of this kernel's instructions; at D = 64 the kernel is as much VALU- as HBM-bound
        const uint32_t T = a.thr[pk >> 16];
        const uint32_t tt = T * 0x10001u;
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = pk_add(acc, pk_sub_sat(tt, sm[k]));
        tot_res[pass] = TOP2 ? cand : group_allsum<LPP>((acc & 0xffffu) + (acc >> 16));
        pk_res[pass] = pk;
        thr_res[pass] = T;
    }
    __syncthreads();

    if (gl == 0) {
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int xl = pass * PPP + grp, x = x0 + xl;
            if (x >= g.w) continue;
            const int bd = (int)(pk_res[pass] & 0xffffu), bc = (int)(pk_res[pass] >> 16);
            const int T = (int)thr_res[pass];
            const uint16_t *srow = s_lds + xl * DP;
            const int l = bd > 0 ? srow[bd - 1] : 0x7fff, r = bd < D - 1 ? srow[bd + 1] : 0x7fff;
            const int tot_nbr = max(T - bc, 0) + max(T - l, 0) + max(T - r, 0);
            uint32_t out = kWtaInvalid;
            // TOP2: unique iff the runner-up's cost reaches the threshold or it sits next to the winner
            const bool unique = TOP2 ? ((int)(tot_res[pass] >> 16) >= T || abs((int)(tot_res[pass] & 0xffffu) - bd) <= 1) : (int)tot_res[pass] == tot_nbr;
            if (unique) {
                int subp = bd * 16;
                if (bd > 0 && bd < D - 1) {
                    const int num = l - r, den = l - 2 * bc + r;
                    if (den != 0) subp += (num * 16 + den) / (2 * den);
                }
                out = (uint32_t)subp & 0xffffu;
            }
            wta_l[(size_t)frame * g.npx + (size_t)y * g.w + x] = (uint16_t)out;
        }
    }

    // right view (oracle S6): the tile's minima per right pixel are complete in s_rv, merged across tiles by atomicMin
    for (int pi = threadIdx.x; pi < kWtaTileX + D - 1; pi += 256) {
        const int p = x0 - (D - 1) + pi;
        const uint32_t best = s_rv[pi];
        if (p >= 0 && p < g.w && best != 0xffffffffu) atomicMin(&right_pk[(size_t)frame * g.npx + (size_t)y * g.w + p], best);
    }
}

void launch_wta(const SlabTable &slabs, uint16_t *wta_l, uint32_t *right_pk, const Geometry &g, const uint16_t *thr,
                int n_frames, hipStream_t s, bool top2) {
    dim3 grid((g.w + kWtaTileX - 1) / kWtaTileX, g.h, n_frames), block(256);
    const size_t lds = (size_t)kWtaTileX * (g.D + 8) * sizeof(uint16_t);
    WtaArgs a{slabs, wta_l, right_pk, g, thr, g.P, {0, 1, 2, 3, 4, 5, 6, 7}};
    if (top2) {   // S5 variant (CART_OPT_SPEC_S5_TOP2)
        switch (g.D) {
            case 64: hipLaunchKernelGGL((wta_kernel<4, true>), grid, block, lds, s, a); break;
            case 128: hipLaunchKernelGGL((wta_kernel<8, true>), grid, block, lds, s, a); break;
            default: hipLaunchKernelGGL((wta_kernel<16, true>), grid, block, lds, s, a); break;
        }
        return;
    }
    switch (g.D) {
        case 64: hipLaunchKernelGGL(wta_kernel<4>, grid, block, lds, s, a); break;
        case 128: hipLaunchKernelGGL(wta_kernel<8>, grid, block, lds, s, a); break;
        default: hipLaunchKernelGGL(wta_kernel<16>, grid, block, lds, s, a); break;
    }
}

// ------------------------------------------------------------------ winner takes all, fused with the "up" direction
// For batches the slab of ONE direction never has to exist: this kernel sweeps every column bottom-up, computes the
// "up" path costs on the fly (the same agg_step, registers only), adds the other P-1 slabs and runs the WTA of the row
// it is on.  That removes 1/P of the slab writes and reads (the launch sequence is HBM bound: aggregate writes at
// ~4.6 TB/s, WTA reads at ~6 TB/s).  Block = 4 waves = 4*P adjacent columns, one image row per step; per step
//   * cost recurrence of the block's columns (wave-private right-census window through LDS like aggregate_kernel),
//   * S = L_up + sum of the stored slabs (16-byte non-temporal loads, prefetched one step ahead),
//   * left disparity exactly as wta_kernel (packed keys, integer uniqueness threshold, sub-pixel from the LDS tile),
//   * right view: every lane min-reduces its 16 (S<<16|d) keys into a block-local LDS array indexed by p = x - d
//     (ds_min_u32); the block's NR = 4P + D - 1 minima of the row go to a per-block partial buffer with plain stores
//     and rv_merge_kernel takes the min over the <= ceil((D-1)/4P)+1 blocks that cover a right pixel (global atomics
//     straight from this kernel cost 0.16 ms per 16-frame launch, the partial buffer is 2 % of the slab traffic).
// The row loop has no block barrier (the LDS sum tile is only read by the wave that wrote it); left disparities and
// right-view minima are buffered in LDS for 16 rows and written out in one burst between two barriers, so the row
// loop itself holds loads only and the prefetches stay in flight while a row is processed.
// Waves per block of the fused sweep.  The sweep has frames*W/(64/LPP) waves in total (2484 at 16 x 1242, D=128: 2.4 per
// SIMD), so small blocks spread them evenly over the CUs: with 4-wave blocks a quarter of the CUs carried 3 blocks, the
// rest 2, and the launch took the time of 3.  D=256 keeps 4 waves: its blocks would otherwise be 8 columns wide and the
// right-view partial rows (columns + D - 1 entries per block and row) would grow to 17 % of the slab traffic.
constexpr int fused_waves(int lpp) { return lpp >= 16 ? 4 : 2; }
constexpr int kFusedRB = 16;   // rows buffered in LDS between two bursts of the fused sweep (8 and 32 measured level / slower)
// ... except on wide images at D = 256, where blocks of 8 waves (32 columns) give ~one block per CU and halve the partial
// right-view rows again: 1920x1080, 4 frames: 2.65 instead of 3.08 ms per launch (at 1242 wide 8 waves lose 10 %)
inline int fused_waves_for(const Geometry &g) { return g.D >= 256 && g.w >= 1600 ? 2 * fused_waves(16) : fused_waves(g.D / 16); }


struct FusedArgs {
    const uint32_t *cen_l, *cen_r;
    SlabTable slabs;
    uint16_t *wta_l;
    uint32_t *partial;   // [frame][block][sweep step][rv_row_slots] u16 right-view minima of every block (rv_key16; last slot of a row: unused sink)
    Geometry g;
    const uint16_t *thr; // as WtaArgs::thr
    int xcd_frames;      // xcd_placement(): blocks are decoded per XCD
};
constexpr int kUpPath = 1;   // slab index of the direction computed here and never stored (oracle order: down, up, ...)

template <int LPP, int NP>
struct FusedRegs {
    uint32_t win[Win<LPP>::NLD];
    uint32_t fl;
    uint32_t sv[2][NP - 1][4];   // two rows of slab bytes in flight (see the Little's-law note at the kernel; one row at <= 128 VGPRs measured slower, profiles/r04_fused.txt)
};

// NP = number of paths (compile time: every VMEM instruction of the row loop is unconditional, so that the compiler
// can use exact counted vmcnt waits and the loads of row y-1 stay in flight while row y is processed)
template <int LPP, int NP, int WPB_ = fused_waves(LPP)>
__global__ __launch_bounds__(64 * WPB_, 3) void wta_fused_kernel(FusedArgs a) {  // >= 3 waves per SIMD (HIP: min waves per EU); 4 for the 4-path variants (120 VGPRs, 39 KB of LDS) measured the same
    using WN = Win<LPP>;
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    constexpr int WPB = WPB_, NT = 64 * WPB;
    constexpr bool RVPAD = LPP == 16 && NP == 4;   // see rv_slot
    constexpr int P = WN::P, D = WN::D, COLS = WPB * P, NR = COLS + D - 1, NRP = rv_row_slots(COLS, D, RVPAD);   // slots per row, the last one a spare
    static_assert(COLS == 16 || COLS == 32, "rv_key16 packs the column into 4 or 5 bits");
    static_assert((RVPAD ? rv_slot(NR - 1) : NR - 1) < NRP - 1, "the row's last slot is a spare");
    __shared__ uint32_t s_win[WPB][WN::BUF];
    constexpr int DP = D + 8;                    // LDS pitch of a pixel's sum row (16 B of padding against bank conflicts)
    __shared__ __attribute__((aligned(16))) uint16_t s_tile[1][COLS * DP];
    constexpr int RB = kFusedRB;                 // rows buffered in LDS between two bursts (32 rows cost an LDS-limited block per CU)
    __shared__ __attribute__((aligned(16))) uint32_t s_rmin[RB][NRP];
    __shared__ uint2 s_rec[RB][COLS];            // per pixel: best disparity, unique flag, best cost | its two neighbour costs
    constexpr int NTHR = NP <= 4 ? 1024 : 2048;  // sums are <= NP * 255
    __shared__ uint16_t s_thr[NTHR];             // uniqueness threshold by best cost
    const Geometry &g = a.g;
    keep_f16_denormals();
    const int nblk = (g.w + COLS - 1) / COLS;
    // same XCD placement as aggregate_kernel (frames x, x + 8, ... on XCD x): the sweep re-reads the census planes the
    // aggregation launch has just pulled into that XCD's L2
    int bid = (int)blockIdx.x, frame0 = 0, fstep = 1;
    if (a.xcd_frames) { frame0 = bid & 7; bid >>= 3; fstep = 8; }
    const int frame = frame0 + fstep * (bid / nblk), blk = bid - (bid / nblk) * nblk, x0 = blk * COLS;
    const int hpad = (g.h + RB - 1) / RB * RB + RB;   // rows of one block in the partial buffer (see flush)
    const int lane = threadIdx.x & 63, wid = uniform((int)(threadIdx.x >> 6));
    const int gl = lane % LPP, pg = lane / LPP, d0 = gl * 16;
    const int xl = wid * P + pg, x = x0 + xl;       // column inside the block / the image
    const int xw0 = x0 + wid * P;                   // wave's first column (uniform)
    const bool valid = x < g.w;
    const uint32_t p1p1 = (uint32_t)g.p1 * 0x10001u, p2p2 = (uint32_t)g.p2 * 0x10001u;
    const uint32_t sel_lo = gl == 0 ? 0x05040d0du : 0x05040302u;
    const uint32_t sel_hi = gl == LPP - 1 ? 0x0d0d0302u : 0x05040302u;

    for (int i = threadIdx.x; i < RB * NRP; i += NT) (&s_rmin[0][0])[i] = 0xffffffffu;
    for (int i = threadIdx.x; i < NTHR; i += NT) s_thr[i] = a.thr[i];

    // right-census window of the wave (see aggregate_kernel): cooperative load offsets + this lane's read slots
    WinLane<LPP> wlane;
    wlane.init(lane);
    unsigned (&goff)[WN::NLD] = wlane.goff;
    const int (&lslot)[WN::NLD] = wlane.lslot;
    const int rbase = wlane.rbase;
    uint32_t *wbuf = &s_win[wid][0];

    // row-0 bases (wave-uniform, kept in SGPRs) + constant per-lane byte offsets: every load is "scalar base + VGPR
    // offset".  Columns past the image compute on padding / the clamped last slab column and never write.
    const ptrdiff_t cen0 = (ptrdiff_t)frame * (ptrdiff_t)g.census_elems + g.cpadl + xw0;
    const uint32_t *pw0 = a.cen_r + uniform(cen0 - g.min_disp - (D - 1));
    const uint32_t *pl0 = a.cen_l + uniform(cen0);
    unsigned lo_l = (unsigned)pg * 4u;
    const int xbase = min(xw0, g.w - 1), xc = min(x, g.w - 1);   // xbase <= xc
    const uint8_t *ps0 = a.slabs.frame[frame] + uniform((ptrdiff_t)xbase * D);
    unsigned lo_s = (unsigned)((xc - xbase) * D + d0);
    const ptrdiff_t row_bytes = (ptrdiff_t)g.w * D;

    // Census registers: one set, re-loaded for row y-1 as soon as row y has consumed it.  Slab registers: two sets, each
    // re-loaded for row y-2 when row y has consumed it -- the sweep has only frames*W/P waves (2484 at 16 x 1242, D=128)
    // with 7 KB of slab bytes per wave and row, and one row in flight (17 MB) capped the reads at 4.7 TB/s.
    FusedRegs<LPP, NP> r;
    auto load_census_row = [&](int y) {
        const uint32_t *pw = pw0 + (ptrdiff_t)y * g.cpitch;
#pragma unroll
        for (int i = 0; i < WN::NLD; ++i) r.win[i] = ld_u32(pw, goff[i]);
        r.fl = ld_u32(pl0 + (ptrdiff_t)y * g.cpitch, lo_l);
    };
    auto load_slab_row = [&](int y, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        const uint8_t *ps = ps0 + (ptrdiff_t)y * row_bytes;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (p == kUpPath) continue;
            const int k = p < kUpPath ? p : p - 1;  // compile-time after unrolling
            const v4u v = __builtin_nontemporal_load((const CART_GLOBAL v4u *)((const CART_GLOBAL char *)sgpr(ps + (ptrdiff_t)p * (ptrdiff_t)g.slab_bytes) + pin_v(lo_s)));
            r.sv[SET][k][0] = v.x; r.sv[SET][k][1] = v.y; r.sv[SET][k][2] = v.z; r.sv[SET][k][3] = v.w;
        }
    };

    uint32_t st[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = 0;
    uint32_t mm = 0;

    // The two halves of a row are split so that they can be software-pipelined: agg(y) advances the "up" path state to
    // row y (the only cross-row dependency), wta(...) runs the WTA of the PREVIOUS row on a copy of its state.  The two
    // instruction streams are independent, so the scheduler interleaves them and the long latency chains of one (LDS
    // round trips, DPP reductions) are filled with the other's work.
    auto agg = [&](int y) {   // census registers hold row y; they are re-loaded for row y-1 once consumed
#pragma unroll
        for (int i = 0; i < WN::NLD; ++i) wbuf[lslot[i]] = r.win[i];
        CensusRegs c;
        c.fl = r.fl;
        win_read<LPP>(wbuf, rbase, c.r);
        uint32_t xr[16];
        agg_xor(c, xr);
        load_census_row(max(y - 1, 0));
        agg_step<LPP, false>(st, mm, xr, sel_lo, sel_hi, p1p1, p2p2, nullptr);
    };

    auto wta = [&](const uint32_t (&sp)[8], int lr, int y, auto set_c) {   // sp: path costs of row y; lr: LDS output row
        constexpr int SET = decltype(set_c)::value;
        // ---- S in natural adjacent pairs: sm[q] = (S[d0+2q], S[d0+2q+1]), sm[4+q] = (S[d0+8+2q], S[d0+9+2q])
        uint32_t sm[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sm[q] = perm(sp[2 * q + 1], sp[2 * q], 0x05040100u);
            sm[4 + q] = perm(sp[2 * q + 1], sp[2 * q], 0x07060302u);
        }
#pragma unroll
        for (int k = 0; k < NP - 1; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sm[q] += r.sv[SET][k][q] & 0x00ff00ffu;
                sm[4 + q] += perm(0u, r.sv[SET][k][q], 0x0c030c01u);
            }
        }
        load_slab_row(max(y - 2, 0), set_c);   // this slab set is free again: prefetch row y-2 into it
        uint16_t *tile = &s_tile[0][0];   // single buffer: every wave only touches the rows of its own pixels
        v4u *dst = reinterpret_cast<v4u *>(tile + xl * DP + d0);
        dst[0] = v4u{sm[0], sm[1], sm[2], sm[3]};
        dst[1] = v4u{sm[4], sm[5], sm[6], sm[7]};
        // ---- left view: argmin + uniqueness (same arithmetic as wta_kernel)
        uint32_t key[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // (a shift and an or per register; one v_pk_mad_u16 with the index pair in an SGPR measured level to slower, profiles/r04_fused.txt)
            const u16x2 kk = __builtin_bit_cast(u16x2, sm[k]) * (u16x2){16, 16} + (u16x2){(uint16_t)(2 * k), (uint16_t)(2 * k + 1)};
            key[k] = __builtin_bit_cast(uint32_t, kk);
        }
        uint32_t m = pk_min(pk_min(pk_min(key[0], key[1]), pk_min(key[2], key[3])), pk_min(pk_min(key[4], key[5]), pk_min(key[6], key[7])));
        m = pk_min(m, __builtin_amdgcn_alignbit(m, m, 16)) & 0xffffu;
        uint32_t pk = ((m >> 4) << 16) | (uint32_t)(d0 + (int)(m & 15u));
        pk = group_allmin<LPP>(pk);
        const uint32_t T = s_thr[pk >> 16];  // = uniq_threshold(best cost): the float search runs once per block, not per row
        const uint32_t tt = T * 0x10001u;
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = pk_add(acc, pk_sub_sat(tt, sm[k]));
        const uint32_t tot = group_allsum<LPP>((acc & 0xffffu) + (acc >> 16));
        // ---- right view (oracle S6): key (S<<16 | d) into slot p - (x0 - (D-1)) = xl + D-1 - d.  (Issued before the pixel record below:
        // after it, the record's LDS reads no longer queue behind the atomics, and the sweep was 4 % SLOWER, profiles/r04_fused.txt.)
        if (valid) {
            uint32_t *rrow = &s_rmin[lr][0];
            if constexpr (RVPAD) {
                // Padded rows (rv_slot): slot of entry base - j = s0 - j - [j > b4] with b4 = (xl - 1) & 15 and xl = 4 wid + pg (P = 4 pixels
                // per wave, 16 lanes each): the pad's carry depends on the lane only through pg = lane >> 4, and on the wave only through
                // wid & 3.  One code version per wave class, in which every carry is a COMPILE-TIME lane mask: no lane (slot s0 - j), every
                // lane (s0 - 1 - j), or one of six partial masks (one v_cndmask, shared by the j's with the same mask).  <= 3 selects per
                // row instead of 45 compare / select / shift instructions (246 instead of 288 VALU per wave-step): the D=256 / 4-path
                // sweep 1.36 -> 1.29 ms per 16 frames, means of five alternating runs (profiles/r04_fused.txt).
                static_assert(P == 4 && LPP == 16, "lane masks below assume four 16-lane pixels per wave");
                const int s0 = rv_slot(xl + D - 1 - d0);
                auto emit = [&](auto wc) {
                    constexpr int W4 = decltype(wc)::value;
                    const int ia = s0, ib = s0 - 1;
                    int base = ia;                     // s0 or s0 - 1 per lane: an INDEX into the row (the select works on 32-bit values, never on pointers)
                    unsigned long long prev = 0;       // the carry mask `base` was made for: masks only grow with j, so equal masks are adjacent
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        unsigned long long mask = 0;   // lanes whose slot carries the pad: j > b4(pg)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (j > ((4 * W4 + q - 1) & 15)) mask |= 0xffffull << (16 * q);
                        if (mask != prev) {
                            if (mask == ~0ull) base = ib;
                            // The lane mask goes into VCC as two 32-bit halves.  Handed over as ONE 64-bit "s" operand, the compiler
                            // rematerialises it with s_mov_b64 and a 32-bit literal, and 0x00000000ffff0000 and 0xffffffffffff0000 then
                            // share one encoding (literal 0xffff0000): the last two rows of every sweep came out wrong that way.
                            else asm volatile("s_mov_b32 vcc_lo, %3\n\ts_mov_b32 vcc_hi, %4\n\tv_cndmask_b32 %0, %1, %2, vcc"
                                              : "=v"(base) : "v"(ia), "v"(ib), "i"((int)(uint32_t)mask), "i"((int)(uint32_t)(mask >> 32)) : "vcc");
                            prev = mask;
                        }
                        const int q = j < 8 ? j / 2 : 4 + (j - 8) / 2;   // sm[q] holds local disparities (2q', 2q'+1), low / high half
                        const uint32_t keyv = (j & 1) ? ((sm[q] & 0xffff0000u) | (uint32_t)(d0 + j)) : ((sm[q] << 16) | (uint32_t)(d0 + j));
                        atomicMin(rrow + (base - j), keyv);   // - j rides on the instruction's immediate offset
                    }
                };
                switch (wid & 3) {   // wave-uniform
                    case 0: emit(std::integral_constant<int, 0>{}); break;
                    case 1: emit(std::integral_constant<int, 1>{}); break;
                    case 2: emit(std::integral_constant<int, 2>{}); break;
                    default: emit(std::integral_constant<int, 3>{}); break;
                }
            } else {
                const int s0 = xl + D - 1 - d0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int da = q < 4 ? 2 * q : 8 + 2 * (q - 4);  // local disparity of the low half of sm[q]
                    atomicMin(rrow + s0 - da, (sm[q] << 16) | (uint32_t)(d0 + da));
                    atomicMin(rrow + s0 - da - 1, (sm[q] & 0xffff0000u) | (uint32_t)(d0 + da + 1));
                }
            }
        }
        // No block barrier: the tile rows a lane reads below are its own pixel's, written by lanes of the same wave (LDS
        // operations of one wave execute in order); the block-wide arrays (s_rmin, s_rec) are only read in the burst.
        // The pixel's first lane records (best d, unique?, best cost | neighbour costs); the sub-pixel division is
        // deferred to the burst, where all lanes work on it.
        if (gl == 0) {
            const int bd = (int)(pk & 0xffffu), bc = (int)(pk >> 16);
            const uint16_t *srow = tile + xl * DP;
            const int l = bd > 0 ? srow[bd - 1] : 0x7fff, rr = bd < D - 1 ? srow[bd + 1] : 0x7fff;
            const int Ti = (int)T;
            const int tot_nbr = max(Ti - bc, 0) + max(Ti - l, 0) + max(Ti - rr, 0);
            const uint32_t unique = (int)tot == tot_nbr ? 1u : 0u;
            s_rec[lr][xl] = make_uint2((uint32_t)bd | (unique << 8) | ((uint32_t)bc << 9), (uint32_t)l | ((uint32_t)rr << 16));
        }
    };

    // one pipelined iteration: WTA of sweep step r (image row h-1-r) + path costs of step r+1
    auto iter = [&](int r_, int lr, auto set_c) {
        uint32_t sp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) sp[i] = st[i];
        agg(g.h - 2 - r_);
        wta(sp, lr, g.h - 1 - r_, set_c);
    };

    // Burst of the buffered rows (LDS row r holds image row ytop + nrows-1-r).  Stores inside the row loop would sit
    // between the prefetch loads in vmcnt's in-order retirement; the row loop itself is branch-free and holds loads only.
    // Burst of the buffered rows.  It is branch-free with a fixed number of stores per lane: the partial buffer is laid out
    // [frame][block][sweep step][NRP u16 keys] (rows padded to a multiple of RB), so a burst is one linear copy of RB*NRP slots
    // (rows past the last one land in the padding), and the few left-disparity stores of dead lanes go to a sink entry.
    // With a data-dependent store count (or addresses that spill) the compiler cannot count the VMEM operations between
    // the prefetches issued before the burst and their use after it and waits for everything, including the
    // acknowledgement of the burst's own stores: ~13 us per burst, 0.3 ms per 16-frame launch.
    auto flush = [&](int t0, int nrows) {   // LDS row r = sweep step t0 + r = image row h-1-t0-r
        lds_barrier();
        // partial rows hold u16 keys (rv_key16): NRP / 2 dwords per row
        uint32_t *pbase = a.partial + (((size_t)frame * nblk + blk) * (size_t)hpad + t0) * (NRP / 2);
#pragma unroll
        for (int i0 = 0; i0 < RB * COLS; i0 += NT) {
            const int i = min(i0 + (int)threadIdx.x, RB * COLS - 1);
            const int r = i / COLS, c = i - r * COLS;
            const uint2 rec = s_rec[r][c];
            const int bd = (int)(rec.x & 0xffu), bc = (int)(rec.x >> 9), l = (int)(rec.y & 0xffffu), rr = (int)(rec.y >> 16);
            uint32_t out = kWtaInvalid;
            if (rec.x & 0x100u) {  // oracle S5 sub-pixel
                int subp = bd * 16;
                if (bd > 0 && bd < D - 1) {
                    const int num = l - rr, den = l - 2 * bc + rr;
                    if (den != 0) subp += (num * 16 + den) / (2 * den);
                }
                out = (uint32_t)subp & 0xffffu;
            }
            const bool live = r < nrows && x0 + c < g.w && i0 + (int)threadIdx.x < RB * COLS;
            uint16_t *dst = live ? a.wta_l + (size_t)frame * g.npx + (size_t)(g.h - 1 - t0 - r) * g.w + x0 + c
                                 : reinterpret_cast<uint16_t *>(pbase) + NRP - 1;   // the spare slot of the chunk's first row
            *dst = (uint16_t)out;
        }
        static_assert(RB * NRP % 8 == 0, "a lane packs eight right-view slots into one 16-byte store");
#pragma unroll
        for (int i0 = 0; i0 < RB * NRP / 8; i0 += NT) {
            const int i = i0 + (int)threadIdx.x;
            const bool live = i < RB * NRP / 8;   // excess lanes store ones into the last (never used) row of the block's area
            const v4u ones = v4u{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            v4u v = ones;
            if (live) {
                v4u *src = reinterpret_cast<v4u *>(&s_rmin[0][0]) + 2 * i;
                const v4u t0v = src[0], t1v = src[1];
                src[0] = ones; src[1] = ones;
                v = v4u{rv_key16(t0v.x, COLS) | (rv_key16(t0v.y, COLS) << 16), rv_key16(t0v.z, COLS) | (rv_key16(t0v.w, COLS) << 16),
                        rv_key16(t1v.x, COLS) | (rv_key16(t1v.y, COLS) << 16), rv_key16(t1v.z, COLS) | (rv_key16(t1v.w, COLS) << 16)};
            }
            v4u *dst = live ? reinterpret_cast<v4u *>(pbase) + i
                            : reinterpret_cast<v4u *>(a.partial + (((size_t)frame * nblk + blk + 1) * (size_t)hpad) * (NRP / 2)) - 1;
            *dst = v;
        }
        lds_barrier();
    };

    __syncthreads();
    load_census_row(g.h - 1);
    load_slab_row(g.h - 1, std::integral_constant<int, 0>{});
    load_slab_row(max(g.h - 2, 0), std::integral_constant<int, 1>{});
    agg(g.h - 1);
    flush(0, 0);   // writes nothing that survives (chunk 0 is rewritten by its own burst); it only gives the first entry into
                   // the chunk loop the same VMEM history as every later one, so that the counted waits after a burst stand
    // Sweep steps r = 0..h-1 (image row h-1-r, slab set r & 1).  The pipelined iterations cover r = 0..h-2 in chunks of RB,
    // two per loop trip, straight-line; what is left (one pipelined iteration if h-1 is odd, then the WTA of the last row)
    // runs after the loop: inside it the compiler would have to assume "odd tail, then another chunk" and would shrink the
    // counted waits of slab set 0 to one row in flight.
    const int r_even = (g.h - 1) & ~1;
    for (int r0 = 0; r0 < r_even; r0 += RB) {
        const int nrows = min(RB, r_even - r0);
        // the first pair is peeled so that the waits right after a burst are computed for that history alone (20 stores
        // behind the prefetches) instead of being merged with the loop's back edge
        iter(r0, 0, std::integral_constant<int, 0>{});
        iter(r0 + 1, 1, std::integral_constant<int, 1>{});
        for (int k = 2; k < nrows; k += 2) {
            iter(r0 + k, k, std::integral_constant<int, 0>{});
            iter(r0 + k + 1, k + 1, std::integral_constant<int, 1>{});
        }
        flush(r0, nrows);
    }
    if ((g.h - 1) & 1) {
        iter(r_even, 0, std::integral_constant<int, 0>{});
        wta(st, 1, 0, std::integral_constant<int, 1>{});
        flush(r_even, 2);
    } else {
        wta(st, 0, 0, std::integral_constant<int, 0>{});
        flush(r_even, 1);
    }
}

// right_pk[p] = min over the blocks whose p-range [blk*COLS - (D-1), blk*COLS + COLS - 1] holds p.
// partial = [frame][block][sweep step t = h-1-y][rv_row_slots u16 keys] (rows padded, see wta_fused_kernel's flush)
// At most NB = ceil((D-1)/COLS) + 1 blocks cover a right pixel (17 at D = 256 with 16-column blocks): all NB keys are requested at once,
// out-of-range blocks clamped onto the first one and masked, so that the loads do not wait for each other (as a loop over b0..b1 the
// merge took 0.14 ms per 16-frame launch at D = 256 / 4 paths: a tenth of the sweep it follows).
template <int D, int COLS, bool PADDED>
__global__ __launch_bounds__(256) void rv_merge_kernel(const uint32_t *partial, uint32_t *right_pk, int w, int h, int nblk) {
    constexpr int NB = (D - 1 + COLS - 1) / COLS + 1, NRP = rv_row_slots(COLS, D, PADDED);
    const int p = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), frame = blockIdx.z;
    if (p >= w || y >= h) return;
    const int hpad = (h + kFusedRB - 1) / kFusedRB * kFusedRB + kFusedRB;
    const uint16_t *keys = reinterpret_cast<const uint16_t *>(partial) + ((size_t)frame * nblk * hpad + (size_t)(h - 1 - y)) * NRP;
    const int b0 = p / COLS, b1 = min((p + D - 1) / COLS, nblk - 1);
    uint32_t k[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int b = min(b0 + i, b1);
        const int e = p - (b * COLS - (D - 1));
        k[i] = keys[(size_t)b * hpad * NRP + (PADDED ? rv_slot(e) : e)];
    }
    uint32_t best = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int b = min(b0 + i, b1);   // (a clamped duplicate of the last block changes nothing)
        if (k[i] != 0xffffu) best = min(best, rv_key32(k[i], p - (b * COLS - (D - 1)), COLS, D));
    }
    right_pk[((size_t)frame * h + y) * w + p] = best;
}

inline bool fused_rv_padded(const Geometry &g) { return g.D >= 256 && g.P == 4; }   // = RVPAD of the kernel that will run

size_t wta_fused_partial_elems(const Geometry &g) {
    const int cols = fused_waves_for(g) * (64 / (g.D / 16));
    const int hpad = (g.h + kFusedRB - 1) / kFusedRB * kFusedRB + kFusedRB;
    return (size_t)hpad * ((g.w + cols - 1) / cols) * (rv_row_slots(cols, g.D, fused_rv_padded(g)) / 2);   // u32 elements of u16 keys
}

void launch_wta_fused(const uint32_t *cen_l, const uint32_t *cen_r, const SlabTable &slabs, uint16_t *wta_l, uint32_t *right_pk,
                      uint32_t *partial, const Geometry &g, const uint16_t *thr, int n_frames, hipStream_t s) {
    FusedArgs a{cen_l, cen_r, slabs, wta_l, partial, g, thr, xcd_placement(g, n_frames) ? 1 : 0};
    const int wpb = fused_waves_for(g), cols = wpb * (64 / (g.D / 16));
    const int nblk = (g.w + cols - 1) / cols;
    dim3 grid(nblk * n_frames), block(64 * wpb);
    const bool wide = wpb != fused_waves(g.D / 16);   // D = 256 on wide images: twice the waves per block
    if (g.P == 4) {
        switch (g.D) {
            case 64: hipLaunchKernelGGL((wta_fused_kernel<4, 4>), grid, block, 0, s, a); break;
            case 128: hipLaunchKernelGGL((wta_fused_kernel<8, 4>), grid, block, 0, s, a); break;
            default:
                if (wide) hipLaunchKernelGGL((wta_fused_kernel<16, 4, 2 * fused_waves(16)>), grid, block, 0, s, a);
                else hipLaunchKernelGGL((wta_fused_kernel<16, 4>), grid, block, 0, s, a);
                break;
        }
    } else {
        switch (g.D) {
            case 64: hipLaunchKernelGGL((wta_fused_kernel<4, 8>), grid, block, 0, s, a); break;
            case 128: hipLaunchKernelGGL((wta_fused_kernel<8, 8>), grid, block, 0, s, a); break;
            default:
                if (wide) hipLaunchKernelGGL((wta_fused_kernel<16, 8, 2 * fused_waves(16)>), grid, block, 0, s, a);
                else hipLaunchKernelGGL((wta_fused_kernel<16, 8>), grid, block, 0, s, a);
                break;
        }
    }
    const dim3 mgrid((g.w + 63) / 64, (g.h + 3) / 4, n_frames), mblock(256);
    const bool padded = fused_rv_padded(g);
#define CART_MERGE(DD, CC, PP) hipLaunchKernelGGL((rv_merge_kernel<DD, CC, PP>), mgrid, mblock, 0, s, (const uint32_t *)partial, right_pk, g.w, g.h, nblk)
    if (g.D == 64) CART_MERGE(64, 32, false);
    else if (g.D == 128) CART_MERGE(128, 16, false);
    else if (cols == 32) { if (padded) CART_MERGE(256, 32, true); else CART_MERGE(256, 32, false); }
    else { if (padded) CART_MERGE(256, 16, true); else CART_MERGE(256, 16, false); }
#undef CART_MERGE
}

// ------------------------------------------------------------------ median x2 + LR check + range fix
// Median of 9 from sorted columns: with every 3-element column sorted into (lo, mid, hi),
//   median9 = med3( max3(lo0, lo1, lo2), med3(mid0, mid1, mid2), min3(hi0, hi1, hi2) ).
// A column costs three instructions (v_min3 / v_med3 / v_max3), a median four more, and adjacent pixels share columns.
__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
struct SortedCol { uint32_t lo, mid, hi; };
__device__ __forceinline__ SortedCol sort_col(uint32_t a, uint32_t b, uint32_t c) {
    return SortedCol{min(min(a, b), c), med3u(a, b, c), max(max(a, b), c)};
}
__device__ __forceinline__ uint32_t median_cols(const SortedCol &c0, const SortedCol &c1, const SortedCol &c2) {
    return med3u(max(max(c0.lo, c1.lo), c2.lo), med3u(c0.mid, c1.mid, c2.mid), min(min(c0.hi, c1.hi), c2.hi));
}

// S7 median of the packed right view (low 16 bits) at (x, y); the image border keeps its own value -- or, with the S7 variant
// (CART_OPT_SPEC_S7_REPLICATE_BORDER), is filtered over the replicated border like every other pixel
__device__ __forceinline__ uint32_t right_median_at(const uint32_t *img, int x, int y, int w, int h, bool replicate) {
    const uint32_t *p = img + (size_t)y * w + x;
    if (x < 1 || x >= w - 1 || y < 1 || y >= h - 1) {
        if (!replicate) return p[0] & 0xffffu;
        SortedCol c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t *q = img + min(max(x + k - 1, 0), w - 1);
            c[k] = sort_col(q[(size_t)max(y - 1, 0) * w] & 0xffffu, q[(size_t)y * w] & 0xffffu, q[(size_t)min(y + 1, h - 1) * w] & 0xffffu);
        }
        return median_cols(c[0], c[1], c[2]);
    }
    SortedCol c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = sort_col(p[k - 1 - w] & 0xffffu, p[k - 1] & 0xffffu, p[k - 1 + w] & 0xffffu);
    return median_cols(c[0], c[1], c[2]);
}

// S7 median of the left WTA map at (x, y); border as above
__device__ __forceinline__ uint32_t left_median_at(const uint16_t *img, int x, int y, int w, int h, bool replicate) {
    const uint16_t *p = img + (size_t)y * w + x;
    if (x < 1 || x >= w - 1 || y < 1 || y >= h - 1) {
        if (!replicate) return p[0];
        SortedCol c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint16_t *q = img + min(max(x + k - 1, 0), w - 1);
            c[k] = sort_col(q[(size_t)max(y - 1, 0) * w], q[(size_t)y * w], q[(size_t)min(y + 1, h - 1) * w]);
        }
        return median_cols(c[0], c[1], c[2]);
    }
    SortedCol c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = sort_col(p[k - 1 - w], p[k - 1], p[k - 1 + w]);
    return median_cols(c[0], c[1], c[2]);
}

// One pixel per thread: the right-view median is a gather at x - d, so the launch wants as many independent threads as
// it can get (four pixels per thread with shared left columns measured 50 % slower).
// spec: bit 0 = S8 variant (integer disparity 0 is invalid too), bit 1 = S7 variant (replicated-border medians); 0 = oracle S7 / S8
__device__ __forceinline__ int post_value(const uint16_t *wl, const uint32_t *rp, const uint8_t *gray, int x, int y, const Geometry &g, int spec) {
    const bool replicate = (spec & 2) != 0;
    const uint32_t ml = left_median_at(wl, x, y, g.w, g.h, replicate);
    bool invalid = gray[(size_t)y * g.w + x] == 0 || ml == kWtaInvalid || ((spec & 1) && (ml >> 4) == 0);
    if (!invalid) {
        const int d = (int)(ml >> 4);
        const int k = x - d;
        if (k >= 0 && k < g.w) {
            const int mr = (int)right_median_at(rp, k, y, g.w, g.h, replicate);
            if (abs(mr - d) > 1) invalid = true;
        }
    }
    return invalid ? (g.min_disp - 1) * 16 : (int)ml + g.min_disp * 16;
}

__global__ __launch_bounds__(256) void post_kernel(const uint16_t *wta_l, const uint32_t *right_pk,
                                                   const uint8_t *gray_l, OutBatch out, Geometry g, int spec) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, frame = blockIdx.z;
    if (x >= g.w || y >= g.h) return;
    const int v = post_value(wta_l + (size_t)frame * g.npx, right_pk + (size_t)frame * g.npx, gray_l + (size_t)frame * g.npx, x, y, g, spec);
    uint8_t *obase = out.scattered ? reinterpret_cast<uint8_t *>(out.frames[frame]) : reinterpret_cast<uint8_t *>(out.ptr) + (size_t)frame * out.frame_stride;
    reinterpret_cast<int16_t *>(obase + (size_t)y * out.step)[x] = (int16_t)v;
}

// post_kernel + the first Jacobi pass of disparity::interpolate at radius 2 (interpolateKernel, interpolation.cu:17-82: the 3 x 3 window mean of
// the values inside (min_disp16, max_disp), count > r*r + 1 -- post_kernels.hip, interpolate_r2_kernel) in one launch: a workgroup computes the post
// values of its 64 x 16 tile and a one-pixel halo into LDS (66 x 18: 16 % more post work) and smooths from there, so the intermediate image is never
// written and the step has one launch less.  Out-of-image halo cells hold a value below every valid range (skipped like the reference's
// out-of-image taps).  Same bits as the two launches (tests: every disparity comparison with smoothing_radius = 2).
constexpr int PI_W = 64, PI_H = 16, PI_LW = PI_W + 2, PI_LH = PI_H + 2, PI_PITCH = 68;
__global__ __launch_bounds__(256) void post_interp_kernel(const uint16_t *wta_l, const uint32_t *right_pk, const uint8_t *gray_l, OutBatch out, Geometry g,
                                                          int spec, int min_disp16, int max_disp) {
    __shared__ int16_t tile[PI_LH][PI_PITCH];
    const int x0 = blockIdx.x * PI_W, y0 = blockIdx.y * PI_H, frame = blockIdx.z, tid = threadIdx.x;
    const uint16_t *wl = wta_l + (size_t)frame * g.npx;
    const uint32_t *rp = right_pk + (size_t)frame * g.npx;
    const uint8_t *gray = gray_l + (size_t)frame * g.npx;
    for (int i = tid; i < PI_LH * PI_LW; i += 256) {
        const int ty = i / PI_LW, tx = i - ty * PI_LW;
        const int x = x0 - 1 + tx, y = y0 - 1 + ty;
        int v = -32768;   // outside the image: never inside (min_disp16, max_disp) -- min_disp16 >= 0
        if (x >= 0 && x < g.w && y >= 0 && y < g.h) v = post_value(wl, rp, gray, x, y, g, spec);
        tile[ty][tx] = (int16_t)v;
    }
    __syncthreads();
    // thread -> 4 adjacent pixels of one row: tile columns 4 q + 1 .. 4 q + 4 of tile row r + 1
    const int q = tid & 15, r = tid >> 4;
    const int xb = x0 + 4 * q, y = y0 + r;
    if (xb >= g.w || y >= g.h) return;
    int csum[6], ccnt[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { csum[c] = 0; ccnt[c] = 0; }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int v = tile[r + l][4 * q + c];
            if (v > min_disp16 && v < max_disp) { csum[c] += v; ++ccnt[c]; }
        }
    }
    uint8_t *obase = out.scattered ? reinterpret_cast<uint8_t *>(out.frames[frame]) : reinterpret_cast<uint8_t *>(out.ptr) + (size_t)frame * out.frame_stride;
    int16_t *orow = reinterpret_cast<int16_t *>(obase + (size_t)y * out.step);
    int16_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int sum = csum[i] + csum[i + 1] + csum[i + 2], count = ccnt[i] + ccnt[i + 1] + ccnt[i + 2];
        o[i] = count > 5 ? (int16_t)(int)((float)sum / (float)count) : (int16_t)CART_DISPARITY_INVALID;   // interpolation.cu:33: count > r*r + 1 = 5
    }
    if (((reinterpret_cast<uintptr_t>(orow) | out.step) & 7) == 0 && xb + 4 <= g.w) {
        *reinterpret_cast<uint2 *>(orow + xb) = make_uint2((uint16_t)o[0] | ((uint32_t)(uint16_t)o[1] << 16), (uint16_t)o[2] | ((uint32_t)(uint16_t)o[3] << 16));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (xb + i < g.w) orow[xb + i] = o[i];
    }
}

// can the post stage take the first interpolation pass with it?  (radius 2, the range test representable in the tile's s16 sentinel scheme)
bool post_interp_fusable(int radius, int min_disp16, int max_disp) { return radius == 2 && min_disp16 >= 0 && min_disp16 < (1 << 15) && max_disp > min_disp16; }
void launch_post_interp(const uint16_t *wta_l, const uint32_t *right_pk, const uint8_t *gray_l, const OutBatch &out, const Geometry &g, int n_frames,
                        hipStream_t s, int spec, int min_disp16, int max_disp) {
    dim3 grid((g.w + PI_W - 1) / PI_W, (g.h + PI_H - 1) / PI_H, n_frames), block(256);
    hipLaunchKernelGGL(post_interp_kernel, grid, block, 0, s, wta_l, right_pk, gray_l, out, g, spec, min_disp16, max_disp);
}

void launch_post(const uint16_t *wta_l, const uint32_t *right_pk, const uint8_t *gray_l, const OutBatch &out, const Geometry &g,
                 int n_frames, hipStream_t s, int spec) {
    dim3 grid((g.w + 63) / 64, (g.h + 3) / 4, n_frames), block(64, 4);
    hipLaunchKernelGGL(post_kernel, grid, block, 0, s, wta_l, right_pk, gray_l, out, g, spec);
}

}  // namespace cart_amd
