// gj_device.h -- device-side helpers shared by the encoder and decoder kernels (gfx950, wave64).
#pragma once
// Runtime calls inside the launchers: checked where they are made (gj_runtime.hip keeps the first failure of the thread for gj_hip_last_error;
// gj_hip_encode / gj_hip_decode return -1 when gj_hip_noted() says there was one). Kernel launches report through hipGetLastError() at the end.
extern "C" int gj_hip_note(int hip_error);
extern "C" void gj_hip_note_reset(void);
extern "C" int gj_hip_noted(void);
#define GJ_HIP_CHECK(call) ((void)gj_hip_note((int)(call)))
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gj_hip.h"

#define GJ_WAVE 64

// the handful of helpers that ARE single gfx950 instructions (inline assembly): GJ_KEEP, gj_bfe_u32, gj_pk_min_u16, gj_ubyte_f_opaque,
// gj_scale256_f, and the gj_f2 / gj_ubyte_f they build on
#include <gj_device_asm.h>

// ------------------------------------------------------------------------------------------------
// zig-zag order (ITU T.81 figure A.6): position in scan -> natural (row-major) index.
// Same contents as the reference's gpujpeg_order_natural (src/gpujpeg_table.h:73-84).
// ------------------------------------------------------------------------------------------------
__device__ static constexpr uint8_t GJ_ZZ[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ------------------------------------------------------------------------------------------------
// Segment geometry. A segment is the run of MCUs between two restart markers
// (src/gpujpeg_common.c:813-870); block addressing follows the reference's block list
// (src/gpujpeg_common.c:1040-1085) but is computed arithmetically instead of being stored.
// ------------------------------------------------------------------------------------------------
struct GjSeg {
    int comp;          // component of the scan (non-interleaved), 0 otherwise
    int index_in_scan; // restart marker index = index_in_scan % 8
    int mcu_first;     // first MCU (non-interleaved: first block of the component)
    int nblocks;       // 8x8 blocks coded in this segment
    int last_in_scan;
    int first_in_scan;
    uint64_t first_block; // running block index in coding order; addresses the temp buffer
};

__device__ __forceinline__ GjSeg gj_segment(const gj_geom& g, int s)
{
    GjSeg r;
    if (g.interleaved) {
        const int ri = g.restart_interval ? g.restart_interval : g.mcu_count;
        r.comp = 0;
        r.index_in_scan = s;
        r.mcu_first = s * ri;
        int n = g.mcu_count - r.mcu_first;
        if (n > ri) n = ri;
        r.nblocks = n * g.blocks_per_mcu;
        r.last_in_scan = s == g.segment_count - 1;
        r.first_in_scan = s == 0;
        r.first_block = (uint64_t)r.mcu_first * g.blocks_per_mcu;
        return r;
    }
    int c = 0;
#pragma unroll
    for (int i = 1; i < GJ_MAX_COMP; i++)
        if (i < g.comp_count && s >= g.comp[i].first_segment) c = i;
    const gj_comp_geom& k = g.comp[c];
    const int ri = g.restart_interval ? g.restart_interval : k.mcu_count;
    r.comp = c;
    r.index_in_scan = s - k.first_segment;
    r.mcu_first = r.index_in_scan * ri;
    int n = k.mcu_count - r.mcu_first;
    if (n > ri) n = ri;
    r.nblocks = n;
    r.last_in_scan = r.index_in_scan == k.segment_count - 1;
    r.first_in_scan = r.index_in_scan == 0;
    r.first_block = k.data_offset / 64 + (uint64_t)r.mcu_first;
    return r;
}

// coefficient offset (in int16 units) and component of block k of a segment
__device__ __forceinline__ uint64_t gj_segment_block(const gj_geom& g, const GjSeg& sg, int k, int* comp, int* mcu_pos)
{
    if (!g.interleaved) {
        *comp = sg.comp;
        *mcu_pos = 0;
        return g.comp[sg.comp].data_offset + (uint64_t)(sg.mcu_first + k) * 64;
    }
    const unsigned P = (unsigned)g.blocks_per_mcu;
    const unsigned mi = (unsigned)k / P;
    const unsigned p = (unsigned)k - mi * P;
    const unsigned m = (unsigned)sg.mcu_first + mi;
    const unsigned my = m / (unsigned)g.mcu_count_x;
    const unsigned mx = m - my * (unsigned)g.mcu_count_x;
    const int c = g.mcu_comp[p];
    const gj_comp_geom& kc = g.comp[c];
    const unsigned bx = mx * kc.samp_h + g.mcu_bx[p];
    const unsigned by = my * kc.samp_v + g.mcu_by[p];
    *comp = c;
    *mcu_pos = (int)p;
    return kc.data_offset + ((uint64_t)by * kc.blocks_x + bx) * 64;
}

// LDS written by some lanes of a wave is read by other lanes of the SAME wave: the hardware keeps a wave's LDS operations in
// order, the compiler only has to be told not to move them across this point (no instruction is emitted for the barrier).
__device__ __forceinline__ void gj_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------
// Wave / workgroup prefix sums
// ------------------------------------------------------------------------------------------------
// Inclusive prefix sum over the 64 lanes of a wave with DPP adds (six VALU operations; the shuffle version goes through
// ds_bpermute and waits for LDS six times): shifts inside the rows of 16 lanes, then the last lane of row 0/2 is broadcast
// into row 1/3 and the last lane of the lower half into the upper half (row_bcast:15 / row_bcast:31, gfx9 wave64).
__device__ __forceinline__ uint32_t gj_wave_incl_scan(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false); // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false); // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false); // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
    return (uint32_t)x;
}


// inclusive scan over a 256-thread workgroup; s_tmp needs 4 words; all threads must call
__device__ __forceinline__ uint32_t gj_wg256_incl_scan(uint32_t v, uint32_t* s_tmp, uint32_t* total)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t inc = gj_wave_incl_scan(v);
    __syncthreads(); // protect s_tmp reuse
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    uint32_t a = s_tmp[0], b = s_tmp[1], c = s_tmp[2], d = s_tmp[3];
    uint32_t off = wave == 0 ? 0 : (wave == 1 ? a : (wave == 2 ? a + b : a + b + c));
    if (total) *total = a + b + c + d;
    return inc + off;
}

// ------------------------------------------------------------------------------------------------
// Integer colour transforms. Restates src/gpujpeg_colorspace.h:64-102 (core) and :216-430.
// c*256/255 for c in [-254,255] equals c + (c == 255): integer division truncates toward zero.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int gj_scale256(int v) { return v + (v == 255 ? 1 : 0); }
__device__ __forceinline__ int gj_clamp8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

struct GjMat { int m[9]; int b0, b1, b2; };

__device__ __forceinline__ void gj_matrix_to(int& c0, int& c1, int& c2, const int m0, const int m1, const int m2, const int m3,
                                             const int m4, const int m5, const int m6, const int m7, const int m8, const int b0,
                                             const int b1, const int b2)
{
    const int r0 = gj_scale256(c0), r1 = gj_scale256(c1), r2 = gj_scale256(c2);
    c0 = gj_clamp8(((m0 * r0 + m1 * r1 + m2 * r2 + 128) >> 8) + b0);
    c1 = gj_clamp8(((m3 * r0 + m4 * r1 + m5 * r2 + 128) >> 8) + b1);
    c2 = gj_clamp8(((m6 * r0 + m7 * r1 + m8 * r2 + 128) >> 8) + b2);
}

__device__ __forceinline__ void gj_matrix_from(int& c0, int& c1, int& c2, const int m0, const int m1, const int m2, const int m3,
                                               const int m4, const int m5, const int m6, const int m7, const int m8,
                                               const int b0, const int b1, const int b2)
{
    const int r0 = gj_scale256(c0 - b0), r1 = gj_scale256(c1 - b1), r2 = gj_scale256(c2 - b2);
    c0 = gj_clamp8((m0 * r0 + m1 * r1 + m2 * r2 + 128) >> 8);
    c1 = gj_clamp8((m3 * r0 + m4 * r1 + m5 * r2 + 128) >> 8);
    c2 = gj_clamp8((m6 * r0 + m7 * r1 + m8 * r2 + 128) >> 8);
}

enum { GJ_CS_NONE = 0, GJ_CS_RGB = 1, GJ_CS_BT601 = 2, GJ_CS_BT601_256 = 3, GJ_CS_BT709 = 4, GJ_CS_YUV = 5 };

__device__ __forceinline__ void gj_rgb_to(int cs, int& a, int& b, int& c)
{
    switch (cs) {
    case GJ_CS_BT601: gj_matrix_to(a, b, c, 66, 129, 25, -38, -74, 112, 112, -94, -18, 16, 128, 128); break;
    case GJ_CS_BT601_256: gj_matrix_to(a, b, c, 77, 150, 29, -43, -85, 128, 128, -107, -21, 0, 128, 128); break;
    case GJ_CS_BT709: gj_matrix_to(a, b, c, 47, 157, 16, -26, -87, 112, 112, -102, -10, 16, 128, 128); break;
    case GJ_CS_YUV: gj_matrix_to(a, b, c, 77, 150, 29, -38, -74, 112, 157, -132, -26, 0, 128, 128); break;
    default: break;
    }
}

__device__ __forceinline__ void gj_to_rgb(int cs, int& a, int& b, int& c)
{
    switch (cs) {
    case GJ_CS_BT601: gj_matrix_from(a, b, c, 298, 0, 409, 298, -100, -208, 298, 516, 0, 16, 128, 128); break;
    case GJ_CS_BT601_256: gj_matrix_from(a, b, c, 256, 0, 359, 256, -88, -183, 256, 454, 0, 0, 128, 128); break;
    case GJ_CS_BT709: gj_matrix_from(a, b, c, 298, 0, 459, 298, -55, -136, 298, 541, 0, 16, 128, 128); break;
    case GJ_CS_YUV: gj_matrix_from(a, b, c, 256, 0, 292, 256, -101, -149, 256, 520, 0, 0, 128, 128); break;
    default: break;
    }
}

// generic (runtime) transform; `from`/`to` are wave-uniform so the switches are scalar branches
__device__ __forceinline__ void gj_color_transform(int from, int to, int& a, int& b, int& c)
{
    if (from == to || from == GJ_CS_NONE || to == GJ_CS_NONE) return;
    if (from == GJ_CS_RGB) { gj_rgb_to(to, a, b, c); return; }
    if (to == GJ_CS_RGB) { gj_to_rgb(from, a, b, c); return; }
    // YCbCr -> YCbCr goes through RGB (src/gpujpeg_colorspace.h:354-427); the reference's
    // BT.601-limited -> BT.709 specialisation uses the full-range inverse first (:387-394): kept.
    if (from == GJ_CS_BT601 && to == GJ_CS_BT709) { gj_to_rgb(GJ_CS_BT601_256, a, b, c); gj_rgb_to(GJ_CS_BT709, a, b, c); return; }
    gj_to_rgb(from, a, b, c);
    gj_rgb_to(to, a, b, c);
}

// ------------------------------------------------------------------------------------------------
// The same integer colour transforms evaluated in fp32, two pixels at a time (v_pk_fma_f32), for the fused kernels.
//
// Reference arithmetic (src/gpujpeg_colorspace.h:64-102): out_k = clamp(((sum_j m_kj * c'_j + 128) >> 8) + b_k), c' = c*256/255
// (= c + (c == 255)). Every quantity is an integer below 2^18, so with the matrix pre-divided by 256 all products and
// partial sums are multiples of 1/512 below 2^11: exact in fp32, in any order. floor(x / 256) equals
// rint((x - 127.5) / 256) for integer x (the fraction is never a tie), and v_cvt_pk_u8_f32 rounds to nearest even
// and saturates to 0..255 -- so "shift, add offset, clamp" is the conversion instruction applied to
//     sum_j (m_kj / 256) * c'_j + b_k + 0.5 / 256 .
// The functions return that sum; the caller converts. Checked against the integer path for all 2^24 inputs per matrix
// on the device (tests/test_gpu_parity.py::test_exhaustive_colour_transform_fused).
// ------------------------------------------------------------------------------------------------
// (gj_f2, gj_ubyte_f<K>, gj_ubyte_f_opaque<K> and gj_scale256_f -- c * 256 / 255 for an integer c in [0, 255] -- live in gj_device_asm.h)

__device__ __forceinline__ void gj_matrix_to_f(gj_f2& c0, gj_f2& c1, gj_f2& c2, const int m0, const int m1, const int m2, const int m3, const int m4,
                                               const int m5, const int m6, const int m7, const int m8, const int b0, const int b1, const int b2)
{
    const float s = 1.0f / 256.0f, h = 0.5f / 256.0f;
    const gj_f2 r0 = gj_scale256_f(c0), r1 = gj_scale256_f(c1), r2 = gj_scale256_f(c2);
    c0 = __builtin_elementwise_fma((gj_f2)(m0 * s), r0, __builtin_elementwise_fma((gj_f2)(m1 * s), r1, __builtin_elementwise_fma((gj_f2)(m2 * s), r2, (gj_f2)(b0 + h))));
    c1 = __builtin_elementwise_fma((gj_f2)(m3 * s), r0, __builtin_elementwise_fma((gj_f2)(m4 * s), r1, __builtin_elementwise_fma((gj_f2)(m5 * s), r2, (gj_f2)(b1 + h))));
    c2 = __builtin_elementwise_fma((gj_f2)(m6 * s), r0, __builtin_elementwise_fma((gj_f2)(m7 * s), r1, __builtin_elementwise_fma((gj_f2)(m8 * s), r2, (gj_f2)(b2 + h))));
}

__device__ __forceinline__ void gj_matrix_from_f(gj_f2& c0, gj_f2& c1, gj_f2& c2, const int m0, const int m1, const int m2, const int m3,
                                                 const int m4, const int m5, const int m6, const int m7, const int m8, const int b0, const int b1,
                                                 const int b2)
{
    const float s = 1.0f / 256.0f, h = 0.5f / 256.0f;
    // c - b reaches 255, the one value c * 256 / 255 changes, only when b is 0 (the luminance of the full-range matrices); and the
    // offsets fold into the constant term: sum_j (m_kj / 256) (c_j - b_j) + h = sum_j (m_kj / 256) c_j + (h - sum_j m_kj b_j / 256), every
    // term still a multiple of 1/512 below 2^11. (Written as c - b, the compiler subtracts in integers and converts afterwards:
    // two more instructions per chrominance sample.)
    const gj_f2 r0 = b0 == 0 ? gj_scale256_f(c0) : c0, r1 = b1 == 0 ? gj_scale256_f(c1) : c1, r2 = b2 == 0 ? gj_scale256_f(c2) : c2;
    const float k0 = h - (float)(m0 * b0 + m1 * b1 + m2 * b2) * s, k1 = h - (float)(m3 * b0 + m4 * b1 + m5 * b2) * s, k2 = h - (float)(m6 * b0 + m7 * b1 + m8 * b2) * s;
    c0 = __builtin_elementwise_fma((gj_f2)(m0 * s), r0, __builtin_elementwise_fma((gj_f2)(m1 * s), r1, __builtin_elementwise_fma((gj_f2)(m2 * s), r2, (gj_f2)k0)));
    c1 = __builtin_elementwise_fma((gj_f2)(m3 * s), r0, __builtin_elementwise_fma((gj_f2)(m4 * s), r1, __builtin_elementwise_fma((gj_f2)(m5 * s), r2, (gj_f2)k1)));
    c2 = __builtin_elementwise_fma((gj_f2)(m6 * s), r0, __builtin_elementwise_fma((gj_f2)(m7 * s), r1, __builtin_elementwise_fma((gj_f2)(m8 * s), r2, (gj_f2)k2)));
}

// compile-time colour transform of the fused kernels (the same matrices as gj_rgb_to / gj_to_rgb)
template <int CS_FROM, int CS_TO>
__device__ __forceinline__ void gj_color_f(gj_f2& a, gj_f2& b, gj_f2& c)
{
    if (CS_FROM == CS_TO || CS_FROM == GJ_CS_NONE || CS_TO == GJ_CS_NONE) return;
    if (CS_FROM == GJ_CS_RGB) {
        if (CS_TO == GJ_CS_BT601) gj_matrix_to_f(a, b, c, 66, 129, 25, -38, -74, 112, 112, -94, -18, 16, 128, 128);
        if (CS_TO == GJ_CS_BT601_256) gj_matrix_to_f(a, b, c, 77, 150, 29, -43, -85, 128, 128, -107, -21, 0, 128, 128);
        if (CS_TO == GJ_CS_BT709) gj_matrix_to_f(a, b, c, 47, 157, 16, -26, -87, 112, 112, -102, -10, 16, 128, 128);
        if (CS_TO == GJ_CS_YUV) gj_matrix_to_f(a, b, c, 77, 150, 29, -38, -74, 112, 157, -132, -26, 0, 128, 128);
    } else if (CS_TO == GJ_CS_RGB) {
        if (CS_FROM == GJ_CS_BT601) gj_matrix_from_f(a, b, c, 298, 0, 409, 298, -100, -208, 298, 516, 0, 16, 128, 128);
        if (CS_FROM == GJ_CS_BT601_256) gj_matrix_from_f(a, b, c, 256, 0, 359, 256, -88, -183, 256, 454, 0, 0, 128, 128);
        if (CS_FROM == GJ_CS_BT709) gj_matrix_from_f(a, b, c, 298, 0, 459, 298, -55, -136, 298, 541, 0, 16, 128, 128);
        if (CS_FROM == GJ_CS_YUV) gj_matrix_from_f(a, b, c, 256, 0, 292, 256, -101, -149, 256, 520, 0, 0, 128, 128);
    }
}

// byte `i` (compile-time) of a 24-byte row held in six dwords, as float
template <int I>
__device__ __forceinline__ float gj_row_byte_f(const uint32_t (&px)[6])
{
    const uint32_t w = px[I >> 2];
    return (I & 3) == 0 ? gj_ubyte_f<0>(w) : (I & 3) == 1 ? gj_ubyte_f<1>(w)
         : (I & 3) == 2 ? gj_ubyte_f<2>(w) : gj_ubyte_f<3>(w);
}

// One row of a packed 4:4:4 block: 8 pixels x 3 bytes -> the row of each of the three component blocks (2 dwords each)
template <int CS_FROM, int CS_TO, int X>
__device__ __forceinline__ void gj_color_row_pair(const uint32_t (&px)[6], uint32_t (&o0)[2], uint32_t (&o1)[2], uint32_t (&o2)[2])
{
    gj_f2 a = gj_f2{gj_row_byte_f<3 * X>(px), gj_row_byte_f<3 * X + 3>(px)};
    gj_f2 b = gj_f2{gj_row_byte_f<3 * X + 1>(px), gj_row_byte_f<3 * X + 4>(px)};
    gj_f2 c = gj_f2{gj_row_byte_f<3 * X + 2>(px), gj_row_byte_f<3 * X + 5>(px)};
    gj_color_f<CS_FROM, CS_TO>(a, b, c);
    o0[X >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(a.x, X & 3, o0[X >> 2]);
    o0[X >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(a.y, (X + 1) & 3, o0[X >> 2]);
    o1[X >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(b.x, X & 3, o1[X >> 2]);
    o1[X >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(b.y, (X + 1) & 3, o1[X >> 2]);
    o2[X >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(c.x, X & 3, o2[X >> 2]);
    o2[X >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(c.y, (X + 1) & 3, o2[X >> 2]);
}

template <int CS_FROM, int CS_TO>
__device__ __forceinline__ void gj_color_row(const uint32_t (&px)[6], uint32_t (&o0)[2], uint32_t (&o1)[2], uint32_t (&o2)[2])
{
    o0[0] = o0[1] = o1[0] = o1[1] = o2[0] = o2[1] = 0;
    gj_color_row_pair<CS_FROM, CS_TO, 0>(px, o0, o1, o2);
    gj_color_row_pair<CS_FROM, CS_TO, 2>(px, o0, o1, o2);
    gj_color_row_pair<CS_FROM, CS_TO, 4>(px, o0, o1, o2);
    gj_color_row_pair<CS_FROM, CS_TO, 6>(px, o0, o1, o2);
}

// ------------------------------------------------------------------------------------------------
// 8-point forward DCT (AAN) -- src/gpujpeg_dct_gpu.cu:121-163. Built with -ffp-contract=off: the
// fused operations are exactly the explicit fma calls (fusion map: DESIGN.md section 3).
//
// The transforms are written once for T = float and T = gj_f2 (two floats in a VGPR pair): with gj_f2 every add, mul
// and fma becomes one v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32, i.e. two IEEE-exact fp32 results per instruction
// (CDNA3/4 packed fp32). A block is held as pairs of horizontally adjacent samples for the column pass and
// re-paired (v_pk_mov_b32) into vertically adjacent ones for the row pass; results are bit-identical to the
// scalar sequence.
// ------------------------------------------------------------------------------------------------

template <typename T> __device__ __forceinline__ T gj_fma(T a, T b, T c);
template <> __device__ __forceinline__ float gj_fma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> __device__ __forceinline__ gj_f2 gj_fma<gj_f2>(gj_f2 a, gj_f2 b, gj_f2 c) { return __builtin_elementwise_fma(a, b, c); }

template <typename T>
__device__ __forceinline__ void gj_fdct8(T& x0, T& x1, T& x2, T& x3, T& x4, T& x5, T& x6, T& x7, const float level_shift)
{
    const T diff0 = x0 + x7, diff1 = x1 + x6, diff2 = x2 + x5, diff3 = x3 + x4;
    const T diff4 = x3 - x4, diff5 = x2 - x5, diff6 = x1 - x6, diff7 = x0 - x7;
    const T even0 = diff0 + diff3, even1 = diff1 + diff2, even2 = diff1 - diff2, even3 = diff0 - diff3;
    const T even_diff = even2 + even3;
    const T odd0 = diff4 + diff5, odd1 = diff5 + diff6, odd2 = diff6 + diff7;
    const T odd_diff5 = (odd0 - odd2) * (T)0.382683433f;
    const T odd_diff4 = gj_fma<T>((T)1.306562965f, odd2, odd_diff5);
    const T odd_diff3 = gj_fma<T>(-odd1, (T)0.707106781f, diff7);
    const T odd_diff2 = gj_fma<T>((T)0.541196100f, odd0, odd_diff5);
    const T odd_diff1 = gj_fma<T>(odd1, (T)0.707106781f, diff7);
    // (the row passes add 0.0f in the reference: that can only turn -0 into +0, which the quantiser's rintf erases again)
    x0 = level_shift != 0.0f ? (even0 + even1) + (T)level_shift : even0 + even1;
    x1 = odd_diff1 + odd_diff4;
    x2 = gj_fma<T>(even_diff, (T)0.707106781f, even3);
    x3 = odd_diff3 - odd_diff2;
    x4 = even0 - even1;
    x5 = odd_diff3 + odd_diff2;
    x6 = gj_fma<T>(-even_diff, (T)0.707106781f, even3);
    x7 = odd_diff1 - odd_diff4;
}

// 2-D forward DCT + quantisation of one block.
// px: the 64 unsigned samples, one byte each, row r in px[2r] (columns 0..3) and px[2r + 1] (columns 4..7).
// q = transposed forward table (src/gpujpeg_table.c:112-120): entry [col*8+row].
// out: the quantised coefficients in natural order, two int16 per dword = the layout of the coefficient planes
// (src/gpujpeg_dct_gpu.cu:246-294). rintf(coef * q) is taken with the 1.5 * 2^23 trick: adding it rounds to nearest even
// exactly like v_rndne_f32 and leaves the integer in the low mantissa bits.
// TO_LDS: the rows go to LDS as they are finished (lds_col[dword * 256], the [dword][lane] layout of the coding kernels) instead
// of accumulating in 32 more VGPRs while the 64 of the block are still alive.
template <bool TO_LDS = false>
__device__ __forceinline__ void gj_fdct_quant_pk(const uint32_t (&px)[16], const float* __restrict__ q, uint32_t (&out)[32], uint32_t* lds_col = nullptr)
{
    gj_f2 D[8][4];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t a = px[2 * r], b = px[2 * r + 1];
        D[r][0] = gj_f2{gj_ubyte_f_opaque<0>(a), gj_ubyte_f_opaque<1>(a)};
        D[r][1] = gj_f2{gj_ubyte_f_opaque<2>(a), gj_ubyte_f_opaque<3>(a)};
        D[r][2] = gj_f2{gj_ubyte_f_opaque<0>(b), gj_ubyte_f_opaque<1>(b)};
        D[r][3] = gj_f2{gj_ubyte_f_opaque<2>(b), gj_ubyte_f_opaque<3>(b)};
    }
#pragma unroll
    for (int c = 0; c < 4; c++) // columns first, level shift folded into the DC term
        gj_fdct8<gj_f2>(D[0][c], D[1][c], D[2][c], D[3][c], D[4][c], D[5][c], D[6][c], D[7][c], -1024.0f);
    __builtin_amdgcn_sched_barrier(0); // keep the scheduler from interleaving the passes (register pressure)
    const gj_f2* q2 = reinterpret_cast<const gj_f2*>(q);
#pragma unroll
    for (int rp = 0; rp < 4; rp++) {
        gj_f2 E[8]; // rows 2rp (x) and 2rp + 1 (y)
#pragma unroll
        for (int cp = 0; cp < 4; cp++) {
            E[2 * cp] = gj_f2{D[2 * rp][cp].x, D[2 * rp + 1][cp].x};
            E[2 * cp + 1] = gj_f2{D[2 * rp][cp].y, D[2 * rp + 1][cp].y};
        }
        gj_fdct8<gj_f2>(E[0], E[1], E[2], E[3], E[4], E[5], E[6], E[7], 0.0f);
        uint32_t ux[8], uy[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const gj_f2 u = E[j] * q2[j * 4 + rp] + (gj_f2)12582912.0f;
            const float fx = u.x, fy = u.y; // (bit_cast straight from a vector element reads element 0 for both)
            ux[j] = __builtin_bit_cast(uint32_t, fx);
            uy[j] = __builtin_bit_cast(uint32_t, fy);
        }
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const uint32_t e0 = __builtin_amdgcn_perm(ux[2 * m + 1], ux[2 * m], 0x05040100u);
            const uint32_t e1 = __builtin_amdgcn_perm(uy[2 * m + 1], uy[2 * m], 0x05040100u);
            if (TO_LDS) {
                lds_col[((2 * rp) * 4 + m) * 256] = e0;
                lds_col[((2 * rp + 1) * 4 + m) * 256] = e1;
            } else {
                out[(2 * rp) * 4 + m] = e0;
                out[(2 * rp + 1) * 4 + m] = e1;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------------
// 8-point inverse DCT (lifting scheme) -- src/gpujpeg_dct_gpu.cu:312-366, fusion map DESIGN.md 3.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void gj_idct8(T& v0, T& v1, T& v2, T& v3, T& v4, T& v5, T& v6, T& v7)
{
    const float k0 = 0.4142135623f, k1 = 0.3535533905f, k2 = 0.4619397662f, k3 = 0.1989123673f, k4 = 0.7071067811f;
    const T a2 = v2 * (T)0.5411961f, a4 = v4 * (T)0.509795579f, a5 = v5 * (T)0.601344887f;
    const T t1 = v0 - v1;
    const T b1 = t1 * (T)k1;
    const T b0 = gj_fma<T>(v0, (T)k4, -b1);
    const T b3 = gj_fma<T>(a2, (T)k1, v3 * (T)k2);
    const T b2 = gj_fma<T>(b3, (T)k0, -a2);
    const T b6 = gj_fma<T>(a5, (T)k2, v6 * (T)k0);
    const T b5 = gj_fma<T>(b6, (T)-0.6681786379f, a5);
    const T b7 = gj_fma<T>(a4, (T)k3, v7 * (T)0.49039264f);
    const T b4 = gj_fma<T>(b7, (T)k3, -a4);
    const T c1 = gj_fma<T>(t1, (T)k1, b2);
    const T c2 = gj_fma<T>((T)-2.0f, b2, c1);
    const T c4 = b5 + b4;
    const T c5 = gj_fma<T>((T)2.0f, b5, -c4);
    const T c7 = b6 + b7;
    const T c6 = gj_fma<T>((T)-2.0f, b6, c7);
    const T c0 = b3 + b0;
    const T c3 = gj_fma<T>((T)-2.0f, b3, c0);
    const T d5 = gj_fma<T>(c6, (T)k0, c5);
    const T d6 = gj_fma<T>(d5, (T)-k4, c6);
    const T e5 = gj_fma<T>(d6, (T)k0, d5);
    const T d3 = c3 + c4;
    const T e4 = gj_fma<T>((T)-2.0f, c4, d3);
    const T d2 = c2 + e5;
    const T f5 = gj_fma<T>((T)-2.0f, e5, d2);
    const T e1 = d6 + c1;
    const T e6 = gj_fma<T>((T)-2.0f, d6, e1);
    const T e0 = c0 + c7;
    const T e7 = gj_fma<T>((T)-2.0f, c7, e0);
    v0 = e0; v1 = e1; v2 = d2; v3 = d3; v4 = e4; v5 = f5; v6 = e6; v7 = e7;
}

// Dequantisation + 2-D inverse DCT of one block.
// w: the 64 quantised coefficients in natural order, two int16 per dword (row r in w[4r .. 4r + 3]).
// qf: dequantisation table in natural order as float. coefficient * q is exact in fp32 (|coef| <= 2^15, q <= 255: DQT
// precision is 8 bit, src/gpujpeg_reader.c:682-727), so float(coef) * float(q) equals the reference's
// float(int(coef) * int(q)) (src/gpujpeg_dct_gpu.cu:497-500).
// px: clamped samples, one byte each, laid out like gj_fdct_quant_pk's input. clamp(rintf(x + 128)) is one
// v_cvt_pk_u8_f32 (round to nearest even, saturating; checked against the formula on the device, tests/hooks/cvt_u8_check.hip).
// The permuted operand order {0,4,6,2,7,5,3,1} is src/gpujpeg_dct_gpu.cu:532-539,:583-590.
__device__ __forceinline__ void gj_idct_pk(const uint32_t (&w)[32], const float* __restrict__ qf, uint32_t (&px)[16])
{
    const gj_f2* q2 = reinterpret_cast<const gj_f2*>(qf);
    gj_f2 D[8][4];
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int cp = 0; cp < 4; cp++) {
            const uint32_t v = w[r * 4 + cp];
            D[r][cp] = gj_f2{(float)(int)(int16_t)(v & 0xFFFF), (float)((int)v >> 16)} * q2[r * 4 + cp];
        }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        gj_f2 x0 = D[0][c], x1 = D[4][c], x2 = D[6][c], x3 = D[2][c], x4 = D[7][c], x5 = D[5][c], x6 = D[3][c], x7 = D[1][c];
        gj_idct8<gj_f2>(x0, x1, x2, x3, x4, x5, x6, x7);
        D[0][c] = x0; D[1][c] = x1; D[2][c] = x2; D[3][c] = x3; D[4][c] = x4; D[5][c] = x5; D[6][c] = x6; D[7][c] = x7;
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int rp = 0; rp < 4; rp++) {
        gj_f2 E[8];
#pragma unroll
        for (int cp = 0; cp < 4; cp++) {
            E[2 * cp] = gj_f2{D[2 * rp][cp].x, D[2 * rp + 1][cp].x};
            E[2 * cp + 1] = gj_f2{D[2 * rp][cp].y, D[2 * rp + 1][cp].y};
        }
        gj_f2 X[8] = {E[0], E[4], E[6], E[2], E[7], E[5], E[3], E[1]};
        gj_idct8<gj_f2>(X[0], X[1], X[2], X[3], X[4], X[5], X[6], X[7]);
        uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const gj_f2 lo = X[k] + (gj_f2)128.0f, hi = X[k + 4] + (gj_f2)128.0f;
            a0 = __builtin_amdgcn_cvt_pk_u8_f32(lo.x, k, a0);
            a1 = __builtin_amdgcn_cvt_pk_u8_f32(hi.x, k, a1);
            b0 = __builtin_amdgcn_cvt_pk_u8_f32(lo.y, k, b0);
            b1 = __builtin_amdgcn_cvt_pk_u8_f32(hi.y, k, b1);
        }
        px[4 * rp] = a0; px[4 * rp + 1] = a1; px[4 * rp + 2] = b0; px[4 * rp + 3] = b1;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------------
// Options that act on the planes / the raw image
// ------------------------------------------------------------------------------------------------
// vertical flip of every padded component plane: row y <-> data_height - 1 - y, 4 bytes per thread
// (src/gpujpeg_preprocessor.cu:455-486; the padding rows take part, as in the reference)
static __global__ __launch_bounds__(256) void k_flip_planes(const gj_geom g, uint8_t* __restrict__ planes)
{
    for (int c = 0; c < g.comp_count; c++) {
        const gj_comp_geom& k = g.comp[c];
        const unsigned wd = (unsigned)k.data_width / 4, half = (unsigned)k.data_height / 2;
        uint32_t* p = reinterpret_cast<uint32_t*>(planes + k.data_offset);
        for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < wd * half; i += gridDim.x * 256u) {
            const unsigned y = i / wd, x = i - y * wd;
            uint32_t* a = p + (size_t)y * wd + x;
            uint32_t* b = p + (size_t)((unsigned)k.data_height - 1 - y) * wd + x;
            const uint32_t t = *a;
            *a = *b;
            *b = t;
        }
    }
}

// in-place channel permutation of the raw image (src/gpujpeg_preprocessor.cu:488-559): output channel i takes the source channel in
// nibble i of `map`; 4 = 0xFF, 5 = 0. Formats whose pixels do not share samples: grey, packed 4:4:4 / 4:4:4:4, planar 4:4:4.
static __global__ __launch_bounds__(256) void k_channel_remap(const gj_geom g, uint8_t* __restrict__ raw, const uint32_t map)
{
    const unsigned W = (unsigned)g.width, H = (unsigned)g.height;
    const unsigned pos = blockIdx.x * 256u + threadIdx.x;
    if (pos >= W * H) return;
    const unsigned y = pos / W;
    uint8_t* ch[4] = {nullptr, nullptr, nullptr, nullptr};
    switch (g.pixel_format) {
    case GJ_PF_U8: ch[0] = raw + (size_t)pos + (size_t)g.width_padding * y; break;
    case GJ_PF_444_P012:
        for (int c = 0; c < 3; c++) ch[c] = raw + (size_t)pos * 3 + (size_t)g.width_padding * y + c;
        break;
    case GJ_PF_4444_P0123:
        for (int c = 0; c < 4; c++) ch[c] = raw + (size_t)pos * 4 + (size_t)g.width_padding * y + c;
        break;
    default: // GJ_PF_444_P0P1P2
        for (int c = 0; c < 3; c++) ch[c] = raw + (size_t)c * W * H + pos;
        break;
    }
    const uint8_t fill = g.pixel_format == GJ_PF_U8 ? 128 : 0; // what the loaders give a channel the format does not have
    uint8_t in[8] = {ch[0] ? *ch[0] : (uint8_t)0, ch[1] ? *ch[1] : fill, ch[2] ? *ch[2] : fill, ch[3] ? *ch[3] : (uint8_t)0, 0xFF, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (ch[c]) *ch[c] = in[(map >> (4 * c)) & 7];
}
