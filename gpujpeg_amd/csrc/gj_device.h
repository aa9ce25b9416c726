// gj_device.h -- device-side helpers shared by the encoder and decoder kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gj_hip.h"

#define GJ_WAVE 64

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

// ------------------------------------------------------------------------------------------------
// Wave / workgroup prefix sums
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t gj_wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
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
// 8-point forward DCT (AAN) -- src/gpujpeg_dct_gpu.cu:121-163. Built with -ffp-contract=off: the
// fused operations are exactly the explicit __builtin_fmaf calls (fusion map: DESIGN.md section 3).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gj_fdct8(float& x0, float& x1, float& x2, float& x3, float& x4, float& x5, float& x6, float& x7,
                                         const float level_shift)
{
    const float diff0 = x0 + x7, diff1 = x1 + x6, diff2 = x2 + x5, diff3 = x3 + x4;
    const float diff4 = x3 - x4, diff5 = x2 - x5, diff6 = x1 - x6, diff7 = x0 - x7;
    const float even0 = diff0 + diff3, even1 = diff1 + diff2, even2 = diff1 - diff2, even3 = diff0 - diff3;
    const float even_diff = even2 + even3;
    const float odd0 = diff4 + diff5, odd1 = diff5 + diff6, odd2 = diff6 + diff7;
    const float odd_diff5 = (odd0 - odd2) * 0.382683433f;
    const float odd_diff4 = __builtin_fmaf(1.306562965f, odd2, odd_diff5);
    const float odd_diff3 = __builtin_fmaf(-odd1, 0.707106781f, diff7);
    const float odd_diff2 = __builtin_fmaf(0.541196100f, odd0, odd_diff5);
    const float odd_diff1 = __builtin_fmaf(odd1, 0.707106781f, diff7);
    x0 = (even0 + even1) + level_shift;
    x1 = odd_diff1 + odd_diff4;
    x2 = __builtin_fmaf(even_diff, 0.707106781f, even3);
    x3 = odd_diff3 - odd_diff2;
    x4 = even0 - even1;
    x5 = odd_diff3 + odd_diff2;
    x6 = __builtin_fmaf(-even_diff, 0.707106781f, even3);
    x7 = odd_diff1 - odd_diff4;
}

// 2-D forward DCT + quantisation of one block held in registers (v[row*8+col], unsigned samples).
// q = transposed forward table (src/gpujpeg_table.c:112-120): entry [col*8+row].
// out[i] = natural-order quantised coefficient (src/gpujpeg_dct_gpu.cu:246-294).
__device__ __forceinline__ void gj_fdct_quant(float (&v)[64], const float* __restrict__ q, int (&out)[64])
{
#pragma unroll
    for (int c = 0; c < 8; c++) // columns first, level shift folded into the DC term
        gj_fdct8(v[c], v[8 + c], v[16 + c], v[24 + c], v[32 + c], v[40 + c], v[48 + c], v[56 + c], -1024.0f);
#pragma unroll
    for (int r = 0; r < 8; r++)
        gj_fdct8(v[r * 8], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3], v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7], 0.0f);
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int j = 0; j < 8; j++) out[r * 8 + j] = (int)__builtin_rintf(v[r * 8 + j] * q[j * 8 + r]);
}

// ------------------------------------------------------------------------------------------------
// 8-point inverse DCT (lifting scheme) -- src/gpujpeg_dct_gpu.cu:312-366, fusion map DESIGN.md 3.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gj_idct8(float& v0, float& v1, float& v2, float& v3, float& v4, float& v5, float& v6, float& v7)
{
    const float k0 = 0.4142135623f, k1 = 0.3535533905f, k2 = 0.4619397662f, k3 = 0.1989123673f, k4 = 0.7071067811f;
    const float a2 = v2 * 0.5411961f, a4 = v4 * 0.509795579f, a5 = v5 * 0.601344887f;
    const float t1 = v0 - v1;
    const float b1 = t1 * k1;
    const float b0 = __builtin_fmaf(v0, k4, -b1);
    const float b3 = __builtin_fmaf(v3, k2, a2 * k1);
    const float b2 = __builtin_fmaf(b3, k0, -a2);
    const float b6 = __builtin_fmaf(v6, k0, a5 * k2);
    const float b5 = __builtin_fmaf(b6, -0.6681786379f, a5);
    const float b7 = __builtin_fmaf(v7, 0.49039264f, a4 * k3);
    const float b4 = __builtin_fmaf(b7, k3, -a4);
    const float c1 = __builtin_fmaf(t1, k1, b2);
    const float c2 = __builtin_fmaf(-2.0f, b2, c1);
    const float c4 = b5 + b4;
    const float c5 = __builtin_fmaf(2.0f, b5, -c4);
    const float c7 = b6 + b7;
    const float c6 = __builtin_fmaf(-2.0f, b6, c7);
    const float c0 = b3 + b0;
    const float c3 = __builtin_fmaf(-2.0f, b3, c0);
    const float d5 = __builtin_fmaf(c6, k0, c5);
    const float d6 = __builtin_fmaf(d5, -k4, c6);
    const float e5 = __builtin_fmaf(d6, k0, d5);
    const float d3 = c3 + c4;
    const float e4 = __builtin_fmaf(-2.0f, c4, d3);
    const float d2 = c2 + e5;
    const float f5 = __builtin_fmaf(-2.0f, e5, d2);
    const float e1 = d6 + c1;
    const float e6 = __builtin_fmaf(-2.0f, d6, e1);
    const float e0 = c0 + c7;
    const float e7 = __builtin_fmaf(-2.0f, c7, e0);
    v0 = e0; v1 = e1; v2 = d2; v3 = d3; v4 = e4; v5 = f5; v6 = e6; v7 = e7;
}

// 2-D inverse DCT of one dequantised block in registers (d[row*8+col]); result = clamped samples.
// The permuted operand order {0,4,6,2,7,5,3,1} is src/gpujpeg_dct_gpu.cu:532-539,:583-590.
__device__ __forceinline__ void gj_idct_block(float (&d)[64], int (&out)[64])
{
#pragma unroll
    for (int c = 0; c < 8; c++) {
        float x0 = d[0 * 8 + c], x1 = d[4 * 8 + c], x2 = d[6 * 8 + c], x3 = d[2 * 8 + c];
        float x4 = d[7 * 8 + c], x5 = d[5 * 8 + c], x6 = d[3 * 8 + c], x7 = d[1 * 8 + c];
        gj_idct8(x0, x1, x2, x3, x4, x5, x6, x7);
        d[0 * 8 + c] = x0; d[1 * 8 + c] = x1; d[2 * 8 + c] = x2; d[3 * 8 + c] = x3;
        d[4 * 8 + c] = x4; d[5 * 8 + c] = x5; d[6 * 8 + c] = x6; d[7 * 8 + c] = x7;
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
        float x0 = d[r * 8 + 0], x1 = d[r * 8 + 4], x2 = d[r * 8 + 6], x3 = d[r * 8 + 2];
        float x4 = d[r * 8 + 7], x5 = d[r * 8 + 5], x6 = d[r * 8 + 3], x7 = d[r * 8 + 1];
        gj_idct8(x0, x1, x2, x3, x4, x5, x6, x7);
        out[r * 8 + 0] = gj_clamp8((int)__builtin_rintf(x0 + 128.0f));
        out[r * 8 + 1] = gj_clamp8((int)__builtin_rintf(x1 + 128.0f));
        out[r * 8 + 2] = gj_clamp8((int)__builtin_rintf(x2 + 128.0f));
        out[r * 8 + 3] = gj_clamp8((int)__builtin_rintf(x3 + 128.0f));
        out[r * 8 + 4] = gj_clamp8((int)__builtin_rintf(x4 + 128.0f));
        out[r * 8 + 5] = gj_clamp8((int)__builtin_rintf(x5 + 128.0f));
        out[r * 8 + 6] = gj_clamp8((int)__builtin_rintf(x6 + 128.0f));
        out[r * 8 + 7] = gj_clamp8((int)__builtin_rintf(x7 + 128.0f));
    }
}
