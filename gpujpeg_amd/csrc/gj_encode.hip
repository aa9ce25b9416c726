// gj_encode.hip -- MI355X (gfx950, wave64) JPEG encoder kernels.
//
// Pipeline (all in one stream, the bitstream is assembled on the device):
//   k_encode_rgb444 / k_encode_uyvy422   raw packed pixels -> entropy-coded segments in one kernel (the default for the BASELINE
//                         configurations; no coefficient planes)
//   k_encode_blocks       the same for planar input and for RGB with any chroma sampling: one lane per block in coding order
//   k_fused_rgb444 / k_fused_uyvy422     raw packed pixels -> quantised coefficients (preprocess + DCT + quant fused)
//   k_preprocess/k_copy_planes + k_dct   generic path through padded planes (every pixel format / subsampling)
//   k_huffman             one LANE per 8x8 block: sparse run-length + Huffman coding, bits OR-ed into an LDS stream
//   k_scan_segments       prefix sum of the stuffed segment sizes -> final byte offsets
//   k_assemble            byte stuffing + RSTn + scan headers + EOI, written in stream order
//
// The reference runs preprocess -> (DCT per component) -> codeword kernel -> serialisation kernel -> compaction
// kernel and stitches segments on the host (src/gpujpeg_encoder.c:485-629); the arithmetic below restates
// src/gpujpeg_preprocessor.cu, src/gpujpeg_colorspace.h, src/gpujpeg_dct_gpu.cu and src/gpujpeg_huffman_gpu_encoder.cu.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "gj_device.h"
#include "gj_hip.h"

// -DGJ_TRACE_PHASES (the `trace` target of the Makefile, tools/encoder_phases.py): the first work-item of every workgroup of the fused encoders
// notes the wall clock (100 MHz) at the phase boundaries in a buffer the tool hands over (16 slots per workgroup); the release build has none of it
#ifdef GJ_TRACE_PHASES
static __device__ unsigned long long* gj_trace_buf_e;
extern "C" GJ_HIP_API int gj_hip_trace_set_encoder(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(gj_trace_buf_e), &p, sizeof p) == hipSuccess ? 0 : -1; }
// (gj_hip_trace_stop_encoder(n): every wave ends at stamp n -- the vector instructions of the phases in front of it are what SQ_INSTS_VALU then
// counts, tools/encoder_valu_budget.py takes the differences; the streams of such a launch are garbage)
static __device__ int gj_trace_stop_e = 1 << 30;
extern "C" GJ_HIP_API int gj_hip_trace_stop_encoder(int n) { return hipMemcpyToSymbol(HIP_SYMBOL(gj_trace_stop_e), &n, sizeof n) == hipSuccess ? 0 : -1; }
#define GJ_TRACE_E(slot) do { if (threadIdx.x == 0 && gj_trace_buf_e) gj_trace_buf_e[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
                              if ((slot) >= gj_trace_stop_e) __builtin_amdgcn_endpgm(); } while (0)
#else
#define GJ_TRACE_E(slot) ((void)0)
#endif

// ================================================================================================
// Generic preprocessor: one thread per pixel of the full-resolution grid.
// Restates src/gpujpeg_preprocessor.cu:88-202 (loads, colour transform, point-sampled store).
// ================================================================================================
__global__ __launch_bounds__(256) void k_preprocess(const gj_geom g, const uint8_t* __restrict__ raw, uint8_t* __restrict__ planes)
{
    const unsigned W = (unsigned)g.raw_width, H = (unsigned)g.height;
    const unsigned pos = blockIdx.x * 256u + threadIdx.x;
    if (pos >= W * H) return;
    const unsigned y = pos / W, x = pos - y * W;
    int c0, c1 = 128, c2 = 128, c3 = 0;
    switch (g.pixel_format) {
    case GJ_PF_U8: c0 = raw[(size_t)pos + (size_t)g.width_padding * y]; break;
    case GJ_PF_444_P0P1P2: c0 = raw[pos]; c1 = raw[(size_t)W * H + pos]; c2 = raw[(size_t)2 * W * H + pos]; break;
    case GJ_PF_422_P0P1P2:
        c0 = raw[pos];
        c1 = raw[(size_t)W * H + pos / 2];
        c2 = raw[(size_t)W * H + (size_t)H * ((W + 1) / 2) + pos / 2];
        break;
    case GJ_PF_420_P0P1P2:
        c0 = raw[pos];
        c1 = raw[(size_t)W * H + (size_t)(y / 2) * ((W + 1) / 2) + x / 2];
        c2 = raw[(size_t)W * H + (size_t)((H + 1) / 2 + y / 2) * ((W + 1) / 2) + x / 2];
        break;
    case GJ_PF_444_P012: {
        const uint8_t* p = raw + (size_t)pos * 3 + (size_t)g.width_padding * y;
        c0 = p[0]; c1 = p[1]; c2 = p[2];
        break; }
    case GJ_PF_4444_P0123: {
        const uint8_t* p = raw + (size_t)pos * 4 + (size_t)g.width_padding * y;
        c0 = p[0]; c1 = p[1]; c2 = p[2]; c3 = p[3];
        break; }
    case GJ_PF_422_P1020: {
        const size_t off = (size_t)pos * 2 + (size_t)g.width_padding * y;
        c0 = raw[off + 1];
        if ((off & 3) == 0) { c1 = raw[off]; c2 = raw[off + 2]; }
        else { c1 = raw[off - 2]; c2 = raw[off]; }
        break; }
    default: c0 = 0; break;
    }
    gj_color_transform(g.color_space, g.color_space_internal, c0, c1, c2);
    const int v[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int c = 0; c < GJ_MAX_COMP; c++) {
        if (c >= g.comp_count) break;
        const gj_comp_geom& k = g.comp[c];
        const unsigned sh = (unsigned)k.sub_h, sv = (unsigned)k.sub_v;
        if ((x % sh) || (y % sv)) continue;
        planes[k.data_offset + (size_t)(y / sv) * k.data_width + x / sh] = (uint8_t)v[c];
    }
}

// planar input whose layout already equals the component layout: pitched copies
// (src/gpujpeg_preprocessor.cu:423-453)
__global__ __launch_bounds__(256) void k_copy_planes_in(const gj_geom g, const uint8_t* __restrict__ raw, uint8_t* __restrict__ planes)
{
    size_t src_off = 0;
    for (int c = 0; c < g.comp_count; c++) {
        const gj_comp_geom& k = g.comp[c];
        const size_t spitch = (size_t)k.width + g.width_padding;
        const size_t n = (size_t)k.width * k.height;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            const size_t y = i / k.width, x = i - y * k.width;
            planes[k.data_offset + y * k.data_width + x] = raw[src_off + y * spitch + x];
        }
        src_off += spitch * k.height;
    }
}

// ================================================================================================
// Forward DCT + quantisation, one THREAD per 8x8 block, all components in one launch.
// No LDS, no cross-lane traffic: the whole block lives in 64 VGPRs. A wave reads 64 neighbouring
// blocks, i.e. 512 contiguous bytes per image row.
// ================================================================================================
__device__ __forceinline__ void gj_store_block(int16_t* __restrict__ dst, const uint32_t (&q)[32])
{
    uint4* o = reinterpret_cast<uint4*>(dst);
#pragma unroll
    for (int r = 0; r < 8; r++) o[r] = make_uint4(q[r * 4], q[r * 4 + 1], q[r * 4 + 2], q[r * 4 + 3]);
}

__global__ __launch_bounds__(256) void k_dct(const gj_geom g, const uint8_t* __restrict__ planes, int16_t* __restrict__ coefs,
                                             const float* __restrict__ q_luma, const float* __restrict__ q_chroma)
{
    const unsigned gb = blockIdx.x * 256u + threadIdx.x;
    if (gb >= (unsigned)g.block_count) return;
    int c = 0;
#pragma unroll
    for (int i = 1; i < GJ_MAX_COMP; i++)
        if (i < g.comp_count && (uint64_t)gb * 64 >= g.comp[i].data_offset) c = i;
    const gj_comp_geom& k = g.comp[c];
    const unsigned lb = gb - (unsigned)(k.data_offset / 64);
    const unsigned by = lb / (unsigned)k.blocks_x, bx = lb - by * (unsigned)k.blocks_x;
    const uint8_t* src = planes + k.data_offset + (size_t)by * 8 * k.data_width + bx * 8;
    uint32_t px[16];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint2 w = *reinterpret_cast<const uint2*>(src + (size_t)r * k.data_width);
        px[2 * r] = w.x;
        px[2 * r + 1] = w.y;
    }
    uint32_t q[32];
    gj_fdct_quant_pk(px, k.type ? q_chroma : q_luma, q);
    gj_store_block(coefs + (size_t)gb * 64, q);
}

// ================================================================================================
// Fused fast path: packed 4:4:4 pixels (3 B/pixel) -> coefficients of all three components.
// One thread per block POSITION: it loads its 8 rows x 24 B once (coalesced: a wave covers 1536 contiguous
// bytes of every row), colour-converts the 64 pixels once, then transforms the three component blocks one
// after the other out of byte-packed registers. Removes the planar round trip of the reference
// (1 B/sample written + read again) and its three per-component DCT launches.
// The colour transform is a compile-time choice so that the 64-pixel unrolled body stays branch-free.
// ================================================================================================
template <int CS_FROM, int CS_TO>
__device__ __forceinline__ void gj_color_static(int& a, int& b, int& c)
{
    if (CS_FROM == CS_TO || CS_FROM == GJ_CS_NONE || CS_TO == GJ_CS_NONE) return;
    if (CS_FROM == GJ_CS_RGB) gj_rgb_to(CS_TO, a, b, c);
    else if (CS_TO == GJ_CS_RGB) gj_to_rgb(CS_FROM, a, b, c);
}

// Pixels of one 8x8 block position (packed 4:4:4, 3 B/pixel) -> the three component blocks, one byte per sample.
// gj_load_444 issues all 24 loads (the wave waits for HBM once; the persistent encoder issues them for its NEXT tile while it codes the
// last component of this one), gj_color_444 is the colour transform in fp32 on pixel pairs (gj_color_row). Samples outside the image
// are zero *component* values (src/gpujpeg_common.c:941-944).
template <int R0 = 0, int R1 = 8> // rows [R0, R1) of the block position
__device__ __forceinline__ void gj_load_444(const gj_geom& g, const uint8_t* __restrict__ raw, const unsigned bx, const unsigned by, uint32_t (&px)[8][6])
{
    const size_t pitch = (size_t)g.width * 3 + g.width_padding;
    const bool interior = (bx * 8 + 8 <= (unsigned)g.width) && (by * 8 + 8 <= (unsigned)g.height);
    const bool aligned = ((pitch | (size_t)raw) & 3) == 0;
    if (interior && aligned) {
        // (the row pointers by addition: written as (by * 8 + r) * pitch the compiler multiplies 64-bit numbers for every row)
        const uint8_t* row = raw + (size_t)(by * 8 + R0) * pitch + (size_t)bx * 24;
#pragma unroll
        for (int r = R0; r < R1; r++) {
            const uint2* p = reinterpret_cast<const uint2*>(row);
            const uint2 a = p[0], b = p[1], c = p[2];
            px[r][0] = a.x; px[r][1] = a.y; px[r][2] = b.x; px[r][3] = b.y; px[r][4] = c.x; px[r][5] = c.y;
            row += pitch;
        }
    } else {
#pragma unroll
        for (int r = R0; r < R1; r++) {
            const unsigned y = by * 8 + r;
#pragma unroll
            for (int w = 0; w < 6; w++) {
                uint32_t d = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const unsigned byte = w * 4 + b, x = bx * 8 + byte / 3;
                    if (x < (unsigned)g.width && y < (unsigned)g.height) d |= (uint32_t)raw[(size_t)y * pitch + (size_t)x * 3 + byte % 3] << (8 * b);
                }
                px[r][w] = d;
            }
        }
    }
}

template <int CS_FROM, int CS_TO>
__device__ __forceinline__ void gj_color_444(const gj_geom& g, const unsigned bx, const unsigned by, const uint32_t (&px)[8][6], uint32_t (&pk)[3][16])
{
    const bool interior = (bx * 8 + 8 <= (unsigned)g.width) && (by * 8 + 8 <= (unsigned)g.height);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint32_t o0[2], o1[2], o2[2];
        gj_color_row<CS_FROM, CS_TO>(px[r], o0, o1, o2);
        if (!interior) {
            // byte masks of the samples that lie inside the image: samples outside are zero COMPONENT values (src/gpujpeg_common.c:941-944).
            // (worked out here, inside the branch only the waves at the image's edges take)
            const int cols = min(8, max(0, g.width - (int)(bx * 8))), rows = min(8, max(0, g.height - (int)(by * 8)));
            const uint32_t m_lo = cols >= 4 ? 0xFFFFFFFFu : (1u << (8 * cols)) - 1u;
            const uint32_t m_hi = cols >= 8 ? 0xFFFFFFFFu : cols > 4 ? (1u << (8 * (cols - 4))) - 1u : 0u;
            const uint32_t lo = r < rows ? m_lo : 0u, hi = r < rows ? m_hi : 0u;
            o0[0] &= lo; o0[1] &= hi; o1[0] &= lo; o1[1] &= hi; o2[0] &= lo; o2[1] &= hi;
        }
        pk[0][r * 2] = o0[0]; pk[0][r * 2 + 1] = o0[1];
        pk[1][r * 2] = o1[0]; pk[1][r * 2 + 1] = o1[1];
        pk[2][r * 2] = o2[0]; pk[2][r * 2 + 1] = o2[1];
        // pin the colour transform of this row here (keeps the raw pixels from staying alive into the transforms)
        GJ_KEEP6(pk[0][r * 2], pk[0][r * 2 + 1], pk[1][r * 2], pk[1][r * 2 + 1], pk[2][r * 2], pk[2][r * 2 + 1]);
    }
}

// bx, by: a block position INSIDE the block grid (callers clamp the positions of lanes that have no block of their own to one that exists: what
// such a lane computes is never looked at, and a special case for it -- 48 registers of zeros -- is paid by every wave, round 5)
template <int CS_FROM, int CS_TO>
__device__ __forceinline__ void gj_load_color_444(const gj_geom& g, const uint8_t* __restrict__ raw, const unsigned bx, const unsigned by, uint32_t (&pk)[3][16])
{
    uint32_t px[8][6]; // 8 rows x 24 bytes
    gj_load_444(g, raw, bx, by, px);
    gj_color_444<CS_FROM, CS_TO>(g, bx, by, px, pk);
}

template <int CS_FROM, int CS_TO>
__global__ __launch_bounds__(256, 2) void k_fused_rgb444(const gj_geom g, const uint8_t* __restrict__ raw, int16_t* __restrict__ coefs,
                                                         const float* __restrict__ q_luma, const float* __restrict__ q_chroma)
{
    __shared__ __attribute__((aligned(8))) float s_q[3][64]; // forward tables: read as VGPR pairs for v_pk_mul_f32
    if (threadIdx.x < 192) s_q[threadIdx.x >> 6][threadIdx.x & 63] = (g.comp[threadIdx.x >> 6].type ? q_chroma : q_luma)[threadIdx.x & 63];
    __syncthreads();
    const gj_comp_geom& k0 = g.comp[0];
    const unsigned nb = (unsigned)(k0.blocks_x * k0.blocks_y);
    const unsigned lb = blockIdx.x * 256u + threadIdx.x;
    if (lb >= nb) return;
    const unsigned by = lb / (unsigned)k0.blocks_x, bx = lb - by * (unsigned)k0.blocks_x;
    uint32_t pk[3][16]; // the three component blocks, one byte per sample
    gj_load_color_444<CS_FROM, CS_TO>(g, raw, bx, by, pk);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        uint32_t q[32];
        gj_fdct_quant_pk(pk[c], s_q[c], q);
        gj_store_block(coefs + g.comp[c].data_offset + (size_t)lb * 64, q);
    }
}

// ================================================================================================
// Fused fast path for packed 4:2:2 (UYVY, 2 B/pixel) without colour transform (BASELINE config 4): one thread per MCU
// (16 x 8 pixels = two luminance blocks + Cb + Cr). A wave reads 2 KiB of contiguous bytes per pixel row (2 x 16 B per
// lane), the de-interleave is byte permutes (12 v_perm_b32 per row), then four packed-fp32 transforms in registers.
// Replaces k_preprocess (one thread per pixel, byte loads and stores) + k_dct and their planar round trip.
// ================================================================================================
__global__ __launch_bounds__(256, 2) void k_fused_uyvy422(const gj_geom g, const uint8_t* __restrict__ raw, int16_t* __restrict__ coefs,
                                                          const float* __restrict__ q_luma, const float* __restrict__ q_chroma)
{
    __shared__ __attribute__((aligned(8))) float s_q[2][64];
    if (threadIdx.x < 128) s_q[threadIdx.x >> 6][threadIdx.x & 63] = (threadIdx.x < 64 ? q_luma : q_chroma)[threadIdx.x & 63];
    __syncthreads();
    const gj_comp_geom& kc = g.comp[1];
    const unsigned nm = (unsigned)(kc.blocks_x * kc.blocks_y); // MCUs = chroma blocks
    const unsigned m = blockIdx.x * 256u + threadIdx.x;
    if (m >= nm) return;
    const unsigned my = m / (unsigned)kc.blocks_x, mx = m - my * (unsigned)kc.blocks_x;
    const size_t pitch = (size_t)g.width * 2 + g.width_padding;
    const bool interior = (mx * 16 + 16 <= (unsigned)g.width) && (my * 8 + 8 <= (unsigned)g.height);
    const bool aligned = ((pitch | (size_t)raw) & 15) == 0;
    uint32_t pk[4][16]; // Y0, Y1, Cb, Cr: one byte per sample, row r in [2r], [2r + 1]
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint32_t d[8];
        if (interior && aligned) {
            const uint4* p = reinterpret_cast<const uint4*>(raw + (size_t)(my * 8 + r) * pitch + (size_t)mx * 32);
            const uint4 a = p[0], b = p[1];
            d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
        } else {
            // samples outside the image are zero component values (src/gpujpeg_common.c:941-944); the odd last pixel of an
            // odd-width row shares the chroma of its pair like the generic loader does
            const unsigned y = my * 8 + r;
#pragma unroll
            for (int w = 0; w < 8; w++) {
                uint32_t v = 0;
                const unsigned x0 = mx * 16 + w * 2; // pixels x0, x0 + 1
                if (y < (unsigned)g.height) {
                    const uint8_t* q = raw + (size_t)y * pitch + (size_t)x0 * 2;
                    if (x0 < (unsigned)g.width) v |= (uint32_t)q[1] << 8;
                    if (x0 + 1 < (unsigned)g.width) v |= (uint32_t)q[3] << 24;
                    if (x0 / 2 < (unsigned)kc.width) v |= (uint32_t)q[0] | ((uint32_t)q[2] << 16);
                }
                d[w] = v;
            }
        }
        pk[0][2 * r] = __builtin_amdgcn_perm(d[1], d[0], 0x07050301u);
        pk[0][2 * r + 1] = __builtin_amdgcn_perm(d[3], d[2], 0x07050301u);
        pk[1][2 * r] = __builtin_amdgcn_perm(d[5], d[4], 0x07050301u);
        pk[1][2 * r + 1] = __builtin_amdgcn_perm(d[7], d[6], 0x07050301u);
        const uint32_t uv01 = __builtin_amdgcn_perm(d[1], d[0], 0x06020400u), uv23 = __builtin_amdgcn_perm(d[3], d[2], 0x06020400u);
        const uint32_t uv45 = __builtin_amdgcn_perm(d[5], d[4], 0x06020400u), uv67 = __builtin_amdgcn_perm(d[7], d[6], 0x06020400u);
        pk[2][2 * r] = __builtin_amdgcn_perm(uv23, uv01, 0x05040100u);
        pk[2][2 * r + 1] = __builtin_amdgcn_perm(uv67, uv45, 0x05040100u);
        pk[3][2 * r] = __builtin_amdgcn_perm(uv23, uv01, 0x07060302u);
        pk[3][2 * r + 1] = __builtin_amdgcn_perm(uv67, uv45, 0x07060302u);
    }
#pragma unroll
    for (int b = 0; b < 4; b++) {
#pragma unroll
        for (int t = 0; t < 16; t++) GJ_KEEP(pk[b][t]); // one transform at a time
        const int c = b < 2 ? 0 : b - 1;
        uint32_t q[32];
        gj_fdct_quant_pk(pk[b], s_q[g.comp[c].type ? 1 : 0], q);
        const size_t blk = b < 2 ? (size_t)my * g.comp[0].blocks_x + 2 * mx + b : (size_t)m;
        gj_store_block(coefs + g.comp[c].data_offset + blk * 64, q);
    }
}

// ================================================================================================
// Huffman coder: one LANE per 8x8 block, 256 blocks per workgroup tile.
//
//  1. each lane loads its block (8 x 16 B), reorders it to zig-zag order with byte permutes and parks it in LDS
//     in a [dword][lane] layout (every lane stays in its own bank whatever position it reads later);
//  2. pass A walks only the NON-ZERO coefficients (bit mask + ctz) and sums the code lengths;
//  3. workgroup prefix sums turn the lengths into exact bit positions inside per-segment streams;
//  4. pass B repeats the walk and ORs the codewords into the LDS bit buffer (ds_or_b32), no serial
//     dependency between blocks;
//  5. the finished (still unstuffed) streams are copied to HBM with coalesced dword stores, together with their
//     byte counts and 0xFF counts; byte stuffing happens in k_assemble where the final position is known.
//
// Symbol semantics restate src/gpujpeg_huffman_gpu_encoder.cu:139-294 / src/gpujpeg_huffman_cpu_encoder.c:136-246
// (DC difference per segment, ZRL for runs >= 16, EOB when the block ends with zeros, ones-padding :489).
// ================================================================================================
#define GJ_HUFF_CAP_DW 2048 // LDS bit buffer: 8 KiB (256 bits per block on average); larger tiles take several windows

struct GjEmit {
    uint64_t acc;
    int accbits;
    uint32_t dw;
};

__device__ __forceinline__ void gj_flush32(GjEmit& e, uint32_t* s_bits, uint32_t wbase, uint32_t wend)
{
    const uint32_t v = (uint32_t)(e.acc >> 32);
    if (e.dw >= wbase && e.dw < wend && v) atomicOr(&s_bits[e.dw - wbase], v);
    e.acc <<= 32;
    e.accbits -= 32;
    e.dw++;
}

__device__ __forceinline__ void gj_emit(GjEmit& e, uint32_t code, int n, uint32_t* s_bits, uint32_t wbase, uint32_t wend)
{
    e.acc |= (uint64_t)code << (64 - e.accbits - n);
    e.accbits += n;
    if (e.accbits >= 32) gj_flush32(e, s_bits, wbase, wend);
}

// category (bit length) and magnitude bits of a coefficient (ITU T.81 F.1.2.1.1)
__device__ __forceinline__ void gj_value_bits(int v, int& nbits, uint32_t& bits)
{
    const int a = v < 0 ? -v : v;
    nbits = a ? 32 - __builtin_clz((unsigned)a) : 0;
    const int t = v < 0 ? v - 1 : v;
    bits = (uint32_t)t & ((1u << nbits) - 1u);
}

template <bool EMIT>
__device__ __forceinline__ uint32_t gj_code_block(const uint32_t* s_coef, const uint32_t* s_lut, int lane, int dc_diff, uint64_t mask,
                                                  int table, int pad_bits, GjEmit& e, uint32_t* s_bits, uint32_t wbase, uint32_t wend)
{
    uint32_t len = 0;
    const uint32_t* lut_dc = s_lut + table * 512;
    const uint32_t* lut_ac = lut_dc + 256;
    {
        int nbits;
        uint32_t bits;
        gj_value_bits(dc_diff, nbits, bits);
        const uint32_t ent = lut_dc[nbits];
        const int sz = ent & 0xFF;
        len += sz + nbits;
        if (EMIT) gj_emit(e, ((ent >> 8) << nbits) | bits, sz + nbits, s_bits, wbase, wend);
    }
    int prev = 0;
    uint64_t m = mask & ~1ull;
    const uint32_t zrl = lut_ac[0xF0];
    while (m) {
        const int p = __builtin_ctzll(m);
        m &= m - 1;
        int run = p - prev - 1;
        prev = p;
        const uint32_t d = s_coef[(p >> 1) * 256 + lane];
        const int v = (p & 1) ? ((int)d >> 16) : (int)(int16_t)(d & 0xFFFF);
        while (run >= 16) {
            len += zrl & 0xFF;
            if (EMIT) gj_emit(e, zrl >> 8, zrl & 0xFF, s_bits, wbase, wend);
            run -= 16;
        }
        int nbits;
        uint32_t bits;
        gj_value_bits(v, nbits, bits);
        const uint32_t ent = lut_ac[(run << 4) | nbits];
        const int sz = ent & 0xFF;
        len += sz + nbits;
        if (EMIT) gj_emit(e, ((ent >> 8) << nbits) | bits, sz + nbits, s_bits, wbase, wend);
    }
    if (prev != 63) {
        const uint32_t eob = lut_ac[0];
        len += eob & 0xFF;
        if (EMIT) gj_emit(e, eob >> 8, eob & 0xFF, s_bits, wbase, wend);
    }
    if (EMIT && pad_bits) gj_emit(e, (1u << pad_bits) - 1u, pad_bits, s_bits, wbase, wend);
    return len;
}

__global__ __launch_bounds__(256) void k_huffman(const gj_geom g, const int16_t* __restrict__ coefs, const uint32_t* __restrict__ lut,
                                                 uint8_t* __restrict__ temp, uint32_t* __restrict__ seg_bytes,
                                                 uint32_t* __restrict__ seg_ff)
{
    __shared__ uint32_t s_coef[32 * 256];
    __shared__ uint32_t s_bits[GJ_HUFF_CAP_DW];
    __shared__ uint32_t s_lut[1024];
    __shared__ int s_dc[256];
    __shared__ uint8_t s_comp[256];
    __shared__ uint32_t s_segx[256], s_segend[256], s_segbase[257], s_segbits[256], s_segff[256], s_segblk[256];
    __shared__ uint32_t s_tmp[4];
    __shared__ uint32_t s_carry_val;
    __shared__ int s_carry_dc[GJ_MAX_COMP];

    const int i = threadIdx.x;
    for (int t = i; t < 1024; t += 256) s_lut[t] = lut[t];

    const int B = g.seg_blocks;          // blocks of a full segment
    const int P = g.blocks_per_mcu;
    const bool small = B <= 256;
    const int spt = small ? 256 / B : 1; // segments per tile
    const int tile_blocks = small ? spt * B : (256 / P) * P;
    // j = i / B through a 16.16 reciprocal (exact for i < 256, B <= 256)
    const uint32_t recip = small ? (65536u + (uint32_t)B - 1u) / (uint32_t)B : 0u;
    const int j = small ? (int)(((uint32_t)i * recip) >> 16) : 0;
    const int seg0 = blockIdx.x * spt;
    const int s = seg0 + j;
    const bool seg_valid = (j < spt) && (s < g.segment_count);
    GjSeg sg;
    sg.nblocks = 0;
    if (seg_valid) sg = gj_segment(g, s);
    const int ntiles = small ? 1 : (seg_valid ? (sg.nblocks + tile_blocks - 1) / tile_blocks : 0);

    // carried state of a segment that spans several tiles (only when B > 256; j == 0 then)
    uint32_t carry_bits = 0, bytes_done = 0, ff_done = 0;
    if (i < GJ_MAX_COMP) s_carry_dc[i] = 0;
    if (i == 0) s_carry_val = 0;

    for (int tile = 0; tile < ntiles; tile++) {
        const int k = small ? i - j * B : tile * tile_blocks + i; // block index inside the segment
        const bool active = seg_valid && (small ? true : i < tile_blocks) && k < sg.nblocks;
        const int k_tile_first = small ? 0 : tile * tile_blocks;
        const int k_tile_end = small ? sg.nblocks : min(sg.nblocks, (tile + 1) * tile_blocks);
        const bool finished = seg_valid && k_tile_end == sg.nblocks;

        __syncthreads(); // previous tile fully consumed
        // ---- 1. load, zig-zag, park in LDS
        int comp = 0, mcu_pos = 0, dc = 0;
        uint64_t mask = 0;
        if (active) {
            const uint64_t off = gj_segment_block(g, sg, k, &comp, &mcu_pos);
            const uint4* src = reinterpret_cast<const uint4*>(coefs + off);
            uint32_t n[32];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint4 w = src[r];
                n[r * 4 + 0] = w.x; n[r * 4 + 1] = w.y; n[r * 4 + 2] = w.z; n[r * 4 + 3] = w.w;
            }
            dc = (int)(int16_t)(n[0] & 0xFFFF);
            uint32_t mlo = 0, mhi = 0;
#pragma unroll
            for (int q = 0; q < 32; q++) {
                const int na = GJ_ZZ[2 * q], nb = GJ_ZZ[2 * q + 1];
                // low half <- natural coefficient na, high half <- nb (v_perm_b32)
                const uint32_t sel = (uint32_t)((na & 1) * 2) | ((uint32_t)((na & 1) * 2 + 1) << 8) | ((uint32_t)(4 + (nb & 1) * 2) << 16) |
                                     ((uint32_t)(5 + (nb & 1) * 2) << 24);
                const uint32_t d = __builtin_amdgcn_perm(n[nb >> 1], n[na >> 1], sel);
                s_coef[q * 256 + i] = d;
                const uint32_t m2 = gj_pk_min_u16(d, 0x00010001u); // non-zero flags of the two halves
                const uint32_t f = (m2 | (m2 >> 15)) & 3u;
                if (q < 16) mlo |= f << (2 * q);
                else mhi |= f << (2 * (q - 16));
            }
            mask = ((uint64_t)mhi << 32) | mlo;
        }
        s_dc[i] = dc;
        s_comp[i] = (uint8_t)comp;
        s_segff[i] = 0;
        __syncthreads();

        // ---- 2. DC prediction + pass A (lengths)
        int dc_diff = 0;
        const int table = active ? g.comp[comp].type : 0;
        uint32_t len = 0;
        GjEmit e = {0, 0, 0};
        if (active) {
            const int dist = g.interleaved ? g.mcu_prev[mcu_pos] : 1;
            int pred;
            if (k - dist < 0) pred = 0;                               // first block of this component in the segment
            else if (k - dist < k_tile_first) pred = s_carry_dc[comp]; // predecessor was coded in the previous tile
            else pred = s_dc[i - dist];
            dc_diff = dc - pred;
            len = gj_code_block<false>(s_coef, s_lut, i, dc_diff, mask, table, 0, e, nullptr, 0, 0);
        }

        // ---- 3. bit positions
        uint32_t total_bits;
        const uint32_t incl = gj_wg256_incl_scan(len, s_tmp, &total_bits);
        const uint32_t excl = incl - len;
        if (active && k == k_tile_first) s_segx[j] = excl;
        if (active && k == k_tile_end - 1) s_segend[j] = incl;
        __syncthreads();
        uint32_t my_dw = 0;
        int my_pad = 0;
        const bool seg_in_tile = (i < spt) && (seg0 + i < g.segment_count);
        if (seg_in_tile) {
            // lane i owns the bookkeeping of local segment i
            uint32_t bits = s_segend[i] - s_segx[i] + (small ? 0u : carry_bits);
            // `finished` is uniform when !small; when small every segment finishes in its tile
            const bool fin = small ? true : finished;
            if (fin) {
                my_pad = (int)((8u - (bits & 7u)) & 7u);
                bits += (uint32_t)my_pad;
            }
            s_segbits[i] = bits | (fin ? 0x80000000u : 0u);
            s_segblk[i] = (uint32_t)gj_segment(g, seg0 + i).first_block;
            my_dw = (bits + 31u) >> 5;
        }
        uint32_t total_dw;
        const uint32_t base_incl = gj_wg256_incl_scan(my_dw, s_tmp, &total_dw);
        if (i < spt) s_segbase[i] = base_incl - my_dw;
        if (i == 0) s_segbase[spt] = total_dw;
        // padding is emitted by the lane that codes the last block of a finished segment
        __syncthreads();
        int pad_bits = 0;
        uint32_t start_bit = 0, end_bit = 0;
        if (active) {
            const uint32_t sb = s_segbits[j];
            start_bit = s_segbase[j] * 32u + (small ? 0u : carry_bits) + (excl - s_segx[j]);
            if ((sb & 0x80000000u) && k == sg.nblocks - 1) pad_bits = (int)((8u - ((start_bit + len) & 7u)) & 7u);
            end_bit = start_bit + len + (uint32_t)pad_bits;
        }

        // ---- 4/5. emit window by window, then drain each window to HBM
        for (uint32_t wbase = 0; wbase < total_dw; wbase += GJ_HUFF_CAP_DW) {
            const uint32_t wend = min(total_dw, wbase + (uint32_t)GJ_HUFF_CAP_DW);
            for (uint32_t d = i; d < wend - wbase; d += 256) s_bits[d] = 0;
            __syncthreads();
            if (!small && wbase == 0 && i == 0 && carry_bits) s_bits[0] = s_carry_val; // bits carried from the previous tile
            __syncthreads();
            if (active && end_bit > wbase * 32u && start_bit < wend * 32u && end_bit > start_bit) {
                e.acc = 0;
                e.accbits = (int)(start_bit & 31u);
                e.dw = start_bit >> 5;
                gj_code_block<true>(s_coef, s_lut, i, dc_diff, mask, table, pad_bits, e, s_bits, wbase, wend);
                if (e.accbits > 0) gj_flush32(e, s_bits, wbase, wend);
            }
            __syncthreads();
            for (uint32_t d = wbase + i; d < wend; d += 256) {
                // which local segment owns dword d: binary search in s_segbase[0..spt]
                int lo = 0, hi = spt;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_segbase[mid] <= d) lo = mid; else hi = mid;
                }
                const uint32_t sb = s_segbits[lo];
                const bool fin = (sb & 0x80000000u) != 0;
                const uint32_t bits = sb & 0x7FFFFFFFu;
                const uint32_t el = d - s_segbase[lo];
                const uint32_t nflush = fin ? (bits + 31u) >> 5 : bits >> 5;
                const uint32_t v = s_bits[d - wbase];
                if (el < nflush) {
                    int vb = 4;
                    if (fin && el == nflush - 1) vb = (int)((bits - el * 32u + 7u) >> 3);
                    uint32_t ff = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        if (b < vb && ((v >> (24 - 8 * b)) & 0xFFu) == 0xFFu) ff++;
                    if (ff) atomicAdd(&s_segff[lo], ff);
                    uint32_t* dst = reinterpret_cast<uint32_t*>(temp + (uint64_t)s_segblk[lo] * GJ_TEMP_BYTES_PER_BLOCK + (small ? 0u : bytes_done)) + el;
                    *dst = __builtin_bswap32(v);
                } else if (!fin && el == nflush) {
                    s_carry_val = v; // partial dword travels to the next tile
                }
            }
            __syncthreads();
        }

        // ---- per-segment results / carry
        if (seg_in_tile) {
            const uint32_t sb = s_segbits[i];
            const uint32_t bits = sb & 0x7FFFFFFFu;
            if (sb & 0x80000000u) {
                seg_bytes[seg0 + i] = (small ? 0u : bytes_done) + ((bits + 7u) >> 3);
                seg_ff[seg0 + i] = (small ? 0u : ff_done) + s_segff[i];
            }
        }
        if (!small) {
            // uniform bookkeeping for the single segment of this workgroup
            const uint32_t sb = s_segbits[0];
            const uint32_t bits = sb & 0x7FFFFFFFu;
            bytes_done += (bits >> 5) * 4u;
            ff_done += s_segff[0];
            carry_bits = bits & 31u;
            __syncthreads();
            if (i < g.comp_count) {
                // last DC of each component inside this tile
                for (int l = min(tile_blocks, k_tile_end - k_tile_first) - 1; l >= 0; l--)
                    if (s_comp[l] == i) { s_carry_dc[i] = s_dc[l]; break; }
            }
        }
    }
}

// ================================================================================================
// The coder of the fully fused encoder kernels (k_encode_rgb444, k_encode_uyvy422): one LANE per 8x8 block, 256 block slots per
// workgroup tile, whole restart segments per tile.
//
//   1. the transform stores every quantised coefficient as 16 bits straight to its ZIG-ZAG position in the lane's own LDS column
//      (ds_write_b16 with immediate offsets, layout [z & 31][lane] dwords, half z >> 5): no packing, no reordering pass;
//   2. the lane reads its column back as 32 dwords and forms the 64-bit non-zero mask (v_pk_min_u16 + v_lshl_or_b32 per dword, two
//      v_perm_b32 at the end);
//   3. ONE walk over the non-zero coefficients (mask + ctz) produces the block's bit stream privately: symbols go into a 64-bit
//      register accumulator, every completed dword is stored IN PLACE over the part of the lane's column the walk has already
//      consumed (the halves of dword f go to positions 2f and 2f + 1 once both are behind the walk; true for anything but blocks
//      that average more than 16 bits per coefficient position), the last partial dword stays in a register;
//   4. prefix sums over the block lengths give exact bit positions inside per-segment streams; the rows of the coefficient area
//      above GJ_ENC_PRIV_ROWS become the shared bit window (nothing else lives in LDS: 36 KB per workgroup, four per CU);
//   5. every lane shift-merges its private dwords into the window (ds_or_b32), coalesced copy of the unstuffed segment streams to
//      d_temp with byte and 0xFF counts per segment (k_scan_segments / k_assemble finish the stream).
//
// A block whose stream does not fit in place (noise at q100) continues it in its own slot of d_temp and reads it back for the
// merge; a tile whose streams exceed the window takes several windows.
// Symbol semantics restate src/gpujpeg_huffman_gpu_encoder.cu:139-294 / src/gpujpeg_huffman_cpu_encoder.c:136-246.
// ================================================================================================
#define GJ_ENC_PRIV_ROWS 24                          // rows of the coefficient area whose lower halves may hold private streams (12 dwords per block)
#define GJ_ENC_WIN_DW ((32 - GJ_ENC_PRIV_ROWS) * 256) // shared bit window: the remaining rows, 2048 dwords
#define GJ_ENC_MAX_SPT 64                            // segments per tile the bookkeeping holds (restart intervals of >= 4 blocks)

// natural (row-major) index -> position in the zig-zag scan (inverse of GJ_ZZ)
__device__ static constexpr uint8_t GJ_IZZ[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
                                                  41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
                                                  46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
// byte offset of zig-zag position z inside a lane's column (column base = lane * 4): dword row z & 31, half z >> 5 -- a walk over
// the lower (upper) 32 positions addresses row * 1024 (+ 2) with one shift-add
#define GJ_COL_OFF(z) (((z) & 31) * 1024 + ((z) >> 5) * 2)

// gj_fdct_quant_pk with the stores of step 1: `col` = this lane's column base in LDS (bytes)
__device__ __forceinline__ void gj_fdct_quant_zz(const uint32_t (&px)[16], const float* __restrict__ q, uint8_t* col)
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
    for (int c = 0; c < 4; c++) gj_fdct8<gj_f2>(D[0][c], D[1][c], D[2][c], D[3][c], D[4][c], D[5][c], D[6][c], D[7][c], -1024.0f);
    __builtin_amdgcn_sched_barrier(0);
    const gj_f2* q2 = reinterpret_cast<const gj_f2*>(q);
#pragma unroll
    for (int rp = 0; rp < 4; rp++) {
        gj_f2 E[8];
#pragma unroll
        for (int cp = 0; cp < 4; cp++) {
            E[2 * cp] = gj_f2{D[2 * rp][cp].x, D[2 * rp + 1][cp].x};
            E[2 * cp + 1] = gj_f2{D[2 * rp][cp].y, D[2 * rp + 1][cp].y};
        }
        gj_fdct8<gj_f2>(E[0], E[1], E[2], E[3], E[4], E[5], E[6], E[7], 0.0f);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            // rintf(coef * q) by adding 1.5 * 2^23: the integer sits in the low mantissa bits, its low 16 bits are the int16
            const gj_f2 u = E[j] * q2[j * 4 + rp] + (gj_f2)12582912.0f;
            const float fx = u.x, fy = u.y;
            *reinterpret_cast<uint16_t*>(col + GJ_COL_OFF(GJ_IZZ[(2 * rp) * 8 + j])) = (uint16_t)__builtin_bit_cast(uint32_t, fx);
            *reinterpret_cast<uint16_t*>(col + GJ_COL_OFF(GJ_IZZ[(2 * rp + 1) * 8 + j])) = (uint16_t)__builtin_bit_cast(uint32_t, fy);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

struct GjCoderLds {
    uint32_t* coef;      // [32][256]; rows GJ_ENC_PRIV_ROWS.. double as the shared bit window once the walks are done
    const uint32_t* lut; // [2][272]: per table type AC[(run << 4) | ((16 - nbits) & 15)] then DC[nbits], entry = (code bits + nbits) << 26 | code << nbits
    uint32_t* wsum;      // [4] block-length totals of the waves
    int* edge;           // [4][16] the last sixteen DC terms of each wave (predecessors of the next wave's first lanes)
    uint32_t *segx, *segend, *segbase, *segbits, *segff; // [64] ([65] segbase)
};

// the private stream of a lane while it walks its block
struct GjWalk {
    uint32_t hi;     // accumulator: 64 - room < 32 bits, left-aligned
    int room;        // 64 - the bits in the accumulator: what a code word is shifted left by, less its own length (kept in this form: one subtraction
                     // per symbol where "fill += n; shift = 64 - fill" takes two)
    int produced;    // completed dwords so far
    int stored;      // once a dword has gone to the block's d_temp slot (lim == GJ_ENC_NO_STORE): the first `stored` dwords sit in the lane's column,
                     // the others in the slot; before that every completed dword is in the column (gj_walk_stored)
    int lim;         // 2 * produced + 1 while every completed dword could be stored in place; GJ_ENC_NO_STORE once one could not
};
#define GJ_ENC_NO_STORE 4096

// append the n <= 26 bits `cw` to a lane's private stream; p = zig-zag position of the coefficient being coded (everything up to
// it has been read). The halves of dword f may be written over positions 2f, 2f + 1 (rows 2f, 2f + 1, lower halves) once both are
// behind the walk and the rows are private ones: lim = 2f + 1 <= min(p, GJ_ENC_PRIV_ROWS - 1). From the first dword that cannot,
// the stream continues in the block's own slot of d_temp (`spill`, GJ_TEMP_BYTES_PER_BLOCK bytes = the largest possible block):
// the segment's final stream, written there by the drain, never reaches a slot whose block it has not passed yet.
__device__ __forceinline__ void gj_put(GjWalk& w, const uint32_t cw, const int n, uint8_t* col, uint32_t* __restrict__ spill, const int p)
{
    w.room -= n;
    const uint64_t t = (uint64_t)cw << w.room; // room was > 32, n <= 26: the shift is >= 6
    w.hi |= (uint32_t)(t >> 32);
    if (w.room <= 32) {
        if (w.lim <= min(p, GJ_ENC_PRIV_ROWS - 1)) {
            *reinterpret_cast<uint16_t*>(col + w.produced * 2048) = (uint16_t)(w.hi >> 16);
            *reinterpret_cast<uint16_t*>(col + w.produced * 2048 + 1024) = (uint16_t)w.hi;
            w.lim += 2;
        } else {
            if (w.lim != GJ_ENC_NO_STORE) w.stored = w.produced; // (the first dword that goes to the slot: the ones in front are in the column)
            spill[w.produced] = w.hi;
            w.lim = GJ_ENC_NO_STORE;
        }
        w.produced++;
        w.hi = (uint32_t)t;
        w.room += 32;
    }
}

// category (bit length) and magnitude bits of a coefficient (ITU T.81 F.1.2.1.1). NONZERO: v != 0 is known (AC walk).
template <bool NONZERO>
__device__ __forceinline__ void gj_value_bits2(const int v, int& nbits, uint32_t& bits)
{
    const int s = v >> 31, t = v + s; // t = v - 1 for negative v
    if (NONZERO) {
        nbits = 32 - gj_ffbh_i32(t); // (t is neither 0 nor -1 for a non-zero v: the first bit that differs from the sign is the top bit of |v|)
    } else {
        const uint32_t a = (uint32_t)(t ^ s) | 1u; // |v| = t ^ s (the 1 keeps clz defined for v == 0)
        nbits = v ? 32 - __builtin_clz(a) : 0;
    }
    bits = __builtin_amdgcn_ubfe((uint32_t)t, 0, (uint32_t)nbits);
}

// the AC part of a walk: non-zero coefficients in zig-zag order, ZRL for runs of 16 zeros, EOB unless the block ends non-zero
__device__ __forceinline__ void gj_walk_ac(uint8_t* col, const uint32_t mlo, const uint32_t mhi, const uint32_t* lut_ac, GjWalk& w,
                                           uint32_t* __restrict__ spill)
{
    int prev = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        uint32_t m = half ? mhi : mlo;
        while (m) {
            const int b = __builtin_ctz(m), p = b + 32 * half;
            m &= m - 1;
            int run = p - prev - 1;
            prev = p;
            const int v = *reinterpret_cast<const int16_t*>(col + b * 1024 + half * 2);
            if (run >= 16) {
                const uint32_t zrl = lut_ac[0xF0];
                do {
                    gj_put(w, zrl & 0x03FFFFFFu, (int)(zrl >> 26), col, spill, p);
                    run -= 16;
                } while (run >= 16);
            }
            // category and magnitude bits of the (non-zero) coefficient (ITU T.81 F.1.2.1.1): t = v - 1 for a negative v; the first bit of t that
            // differs from its sign is the top bit of |v|, so k = v_ffbh_i32(t) = 32 - category. The AC table is indexed by (run << 4) | (k & 15)
            // (gj_huffman_coder_lut): with the table's base moved down by 16 entries that is base[(run << 4) + k], two shift-adds
            const int sg = v >> 31, t = v + sg, k = gj_ffbh_i32(t);
            const uint32_t bits = __builtin_amdgcn_ubfe((uint32_t)t, 0, (uint32_t)(32 - k));
            const uint32_t ent = (lut_ac - 16)[gj_lshl_add_u32<4>((uint32_t)run, (uint32_t)k)];
            gj_put(w, (ent & 0x03FFFFFFu) | bits, (int)(ent >> 26), col, spill, p);
        }
    }
    if (prev != 63) {
        const uint32_t eob = lut_ac[0];
        gj_put(w, eob & 0x03FFFFFFu, (int)(eob >> 26), col, spill, 63);
    }
}

// step 5 of gj_code_tile for one lane and one window [wbase, wend) of the tile stream: the lane's stream -- its completed dwords (column, then, for the
// rare block that outgrew it, the block's d_temp slot), the accumulator, the padding -- lands `sh` bits into dword d0 of the tile stream, every output
// dword is a funnel shift of two neighbours ORed into the window. One loop per KIND of source (round 5: a single loop that picked the source of every
// dword behind four lane-dependent conditions cost 28 vector instructions per dword; these take 8), the window's bounds only where a tile's stream
// needs more than one window (WHOLE = false: noise at high qualities).
template <bool WHOLE>
__device__ __forceinline__ void gj_merge_stream(const GjWalk& w, const uint8_t* col, const uint32_t* __restrict__ spill, const uint64_t tail,
                                                const int ndw, const uint32_t sh, const uint32_t d0, uint32_t* s_bits, const uint32_t wbase,
                                                const uint32_t wend)
{
    uint32_t prevv = 0, d = d0;
    auto emit = [&](const uint32_t cur) {
        const uint32_t out = __builtin_amdgcn_alignbit(prevv, cur, sh);
        if (out && (WHOLE || (d >= wbase && d < wend))) atomicOr(&s_bits[d - wbase], out); // (a dword that is not zero lies inside the lane's segment)
        prevv = cur;
        d++;
    };
    const int stored = w.lim == GJ_ENC_NO_STORE ? w.stored : w.produced; // completed dwords that sit in the column
    for (int f = 0; f < stored; f++)
        emit(((uint32_t)*reinterpret_cast<const uint16_t*>(col + f * 2048) << 16) | *reinterpret_cast<const uint16_t*>(col + f * 2048 + 1024));
    for (int f = stored; f < w.produced; f++) emit(spill[f]);
    // the tail: ndw - produced = 0, 1 or 2 dwords of it carry bits; one more step flushes the last carry
    emit((uint32_t)(tail >> 32));
    if (ndw > w.produced) emit((uint32_t)tail);
    if (ndw > w.produced + 1) emit(0u);
}

// Steps 2-5 for one component of a tile. i = thread, j = local segment of the lane's block, k = block inside its segment,
// nblocks = blocks of that segment, table = 0 luminance / 1 chrominance tables, dc_dist = lanes back to the previous block of the
// same component; region = the tile's area of d_temp (GJ_TEMP_BYTES_PER_BLOCK per block: the tile's UNSTUFFED stream from its start --
// every segment on a dword boundary, in the order of the scan --, block i's spill slot at i * GJ_TEMP_BYTES_PER_BLOCK, which the stream
// reaches only after block i has been merged into it), seg_count_left = segments of the scan from the tile's first one on (the last
// one of a scan gets no restart marker); seg_bytes / seg_ff = unstuffed size and 0xFF count per segment (k_gather stuffs).
// Returns the size of the tile's FINISHED stream (stuffed, restart markers included; the same in every thread).
__device__ __forceinline__ uint32_t gj_code_tile(const GjCoderLds& L, const int i, const int j, const int k, const bool active, const int spt,
                                                 const int nblocks, const int table, const int dc_dist, const int seg_count_left,
                                                 uint8_t* __restrict__ region,
                                                 uint32_t* __restrict__ seg_bytes, uint32_t* __restrict__ seg_ff, const uint32_t first_segment,
                                                 const int trace0 = -1)
{
    (void)trace0;
    const int lane = i & 63, wave = i >> 6;
    uint8_t* const col = reinterpret_cast<uint8_t*>(L.coef) + i * 4;
    uint32_t* const s_bits = L.coef + GJ_ENC_PRIV_ROWS * 256;
    const uint32_t* const lut_ac = L.lut + table * 272;
    const uint32_t* const lut_dc = lut_ac + 256;

    // ---- 2. read the column back: non-zero mask, DC term
    uint32_t mlo = 0, mhi = 0;
    int dc = 0;
    {
        uint32_t elo = 0, ehi = 0;
#pragma unroll
        for (int q = 0; q < 32; q++) {
            const uint32_t d = L.coef[q * 256 + i]; // positions q (lower half) and q + 32
            if (q == 0) dc = (int)(int16_t)(d & 0xFFFFu);
            // both halves clamped to 0 / 1 (v_pk_min_u16); elo collects rows 0..15: bit q = position q, bit 16 + q = position q + 32
            const uint32_t m = gj_pk_min_u16(d, 0x00010001u);
            if (q < 16) elo |= m << q;
            else ehi |= m << (q - 16);
        }
        if (active) {
            mlo = __builtin_amdgcn_perm(ehi, elo, 0x05040100u); // lower halves: positions 0..15 | 16..31
            mhi = __builtin_amdgcn_perm(ehi, elo, 0x07060302u); // upper halves: positions 32..47 | 48..63
        }
    }
    if (lane >= 48) L.edge[wave * 16 + (lane - 48)] = dc;
    if (i < GJ_ENC_MAX_SPT) L.segff[i] = 0;
    __syncthreads(); // B1: edges visible (and, for the first component, the tables)

    // ---- 3. the walk
    GjWalk w = {0, 64, 0, 0, 1};
    uint32_t* const spill = reinterpret_cast<uint32_t*>(region + (size_t)i * GJ_TEMP_BYTES_PER_BLOCK); // (lane i = block i of the tile)
    int dc_diff = 0;
    {
        // DC prediction inside the segment (reset at its first block, src/gpujpeg_huffman_gpu_encoder.cu:339-342)
        const int src = lane - dc_dist;
        int pred = __builtin_amdgcn_ds_bpermute((src & 63) << 2, dc);
        if (src < 0 && wave > 0) pred = L.edge[(wave - 1) * 16 + (16 + src)];
        if (k - dc_dist < 0) pred = 0;
        dc_diff = dc - pred;
    }
    if (active) {
        int nbits;
        uint32_t bits;
        gj_value_bits2<false>(dc_diff, nbits, bits);
        const uint32_t ent = lut_dc[nbits];
        gj_put(w, (ent & 0x03FFFFFFu) | bits, (int)(ent >> 26), col, spill, 0);
        gj_walk_ac(col, mlo & ~1u, mhi, lut_ac, w, spill);
    }
    const int fill = 64 - w.room; // bits in the accumulator (< 32)
    const uint32_t len = (uint32_t)w.produced * 32u + (uint32_t)fill;

    if (trace0 >= 0) GJ_TRACE_E(trace0 + 1); // walk done (this wave)
    // ---- 4. bit positions
    const uint32_t winc = gj_wave_incl_scan(len);
    if (lane == 63) L.wsum[wave] = winc;
    __syncthreads(); // B2: wave totals; every walk is finished, so the window rows are free
    uint32_t excl;
    {
        const uint32_t a = L.wsum[0], b = L.wsum[1], c = L.wsum[2];
        const uint32_t incl = winc + (wave == 0 ? 0u : wave == 1 ? a : wave == 2 ? a + b : a + b + c);
        if (active && k == 0) L.segx[j] = incl - len;
        if (active && k == nblocks - 1) L.segend[j] = incl;
        excl = incl - len;
    }
    {   // clear the first window
        uint4* z = reinterpret_cast<uint4*>(s_bits) + i * 2;
        z[0] = make_uint4(0, 0, 0, 0);
        z[1] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads(); // B3: segment ends visible, window cleared
    if (trace0 >= 0) GJ_TRACE_E(trace0 + 2); // positions known
    // segment books, redundantly in every wave (lane l keeps local segment l): bits with ones-padding to a byte, dword base
    uint32_t sbits = 0, sdw = 0;
    if (lane < spt && lane < seg_count_left) {
        sbits = L.segend[lane] - L.segx[lane];
        sbits += (8u - (sbits & 7u)) & 7u;
        sdw = (sbits + 31u) >> 5;
    }
    const uint32_t sbase_incl = gj_wave_incl_scan(sdw);
    const uint32_t sbase = sbase_incl - sdw;
    const uint32_t total_dw = (uint32_t)__builtin_amdgcn_readlane((int)sbase_incl, 63);
    if (wave == 0) {
        if (lane < spt) { L.segbase[lane] = sbase; L.segbits[lane] = sbits; }
        if (lane == 63) L.segbase[spt] = total_dw;
    }
    uint32_t start_bit = 0;
    int pad_bits = 0;
    {
        const uint32_t my_base = (uint32_t)__builtin_amdgcn_ds_bpermute(j << 2, (int)sbase);
        const uint32_t my_x = active ? L.segx[j] : 0u;
        start_bit = my_base * 32u + (excl - my_x);
        if (active && k == nblocks - 1) pad_bits = (int)((8u - ((start_bit + len) & 7u)) & 7u);
    }

    // ---- 5. merge into the window, drain the window to HBM
    // The lane's stream is its `produced` completed dwords, then the accumulator and the ones-padding of a segment's last block as
    // one 64-bit tail; it lands `start_bit & 31` bits into dword `start_bit >> 5` of the tile stream, so every output dword is
    // one funnel shift (v_alignbit_b32) of two neighbouring stream dwords and one ds_or_b32.
    const uint32_t sh = start_bit & 31u, d0 = start_bit >> 5;
    uint64_t tail = (uint64_t)w.hi << 32;
    if (pad_bits) tail |= (uint64_t)((1u << pad_bits) - 1u) << (64 - fill - pad_bits);
    const int ndw = w.produced + (fill + pad_bits > 32 ? 2 : (fill + pad_bits > 0 ? 1 : 0)); // stream dwords incl. the tail
    const int nseg = min(spt, seg_count_left);
    uint32_t* const dst = reinterpret_cast<uint32_t*>(region); // dword d of the tile stream
    for (uint32_t wbase = 0; wbase < total_dw; wbase += GJ_ENC_WIN_DW) {
        const uint32_t wend = min(total_dw, wbase + (uint32_t)GJ_ENC_WIN_DW);
        if (wbase) {
            __syncthreads(); // previous window drained
            for (uint32_t d = i; d < wend - wbase; d += 256) s_bits[d] = 0;
            __syncthreads();
        }
        if (active && d0 + (uint32_t)ndw + 1u > wbase && d0 < wend) {
            if (total_dw <= (uint32_t)GJ_ENC_WIN_DW) gj_merge_stream<true>(w, col, spill, tail, ndw, sh, d0, s_bits, 0u, total_dw);
            else gj_merge_stream<false>(w, col, spill, tail, ndw, sh, d0, s_bits, wbase, wend);
        }
        __syncthreads(); // B4: window complete
        // every wave drains whole segments: no search for the owner of a dword, the 0xFF count of a segment is one wave reduction
        for (int sl = wave; sl < nseg; sl += 4) {
            const uint32_t sb = L.segbase[sl], nfl = (L.segbits[sl] + 31u) >> 5;
            const uint32_t lo = max(sb, wbase), hi = min(sb + nfl, wend);
            uint32_t ffc = 0;
            for (uint32_t d = lo + (uint32_t)lane; d < hi; d += 64) {
                const uint32_t v = s_bits[d - wbase];
                // 0xFF bytes (the unused low bytes of a segment's last dword are zero)
                ffc += (uint32_t)__builtin_popcount(((v & 0x7F7F7F7Fu) + 0x01010101u) & v & 0x80808080u);
                dst[d] = __builtin_bswap32(v);
            }
            ffc = gj_wave_incl_scan(ffc);
            if (lane == 63 && ffc) L.segff[sl] += ffc;
        }
    }
    __syncthreads(); // B5: 0xFF counts complete; the coefficient area may be overwritten by the next component
    if (trace0 >= 0) GJ_TRACE_E(trace0 + 3); // merged and drained
    // the segments' sizes for k_gather, and what the tile's stream will measure once it is stuffed
    uint32_t out = 0;
    if (lane < nseg) {
        const uint32_t nb = (L.segbits[lane] + 7u) >> 3, ff = L.segff[lane];
        out = nb + ff + (lane != seg_count_left - 1 ? 2u : 0u);
        if (wave == 0) {
            seg_bytes[first_segment + lane] = nb;
            seg_ff[first_segment + lane] = ff;
        }
    }
    return (uint32_t)__builtin_amdgcn_readlane((int)gj_wave_incl_scan(out), 63);
}

// ================================================================================================
// k_gather: the tile streams -> the file. Replaces k_scan_segments + k_assemble behind the k_encode_* kernels (the reference:
// serialisation + compaction kernels and the host's stitching, src/gpujpeg_huffman_gpu_encoder.cu:417-613, src/gpujpeg_encoder.c:567-629).
//
// An encoder workgroup leaves the UNSTUFFED stream of its tile in the tile's area of d_temp (segments on dword boundaries), the
// segments' byte and 0xFF counts, the size the tile's stream will have in the file, and adds that size to the total of its group of
// 32 tiles (one atomic, nobody waits for it). Here ONE WAVE takes one tile stream, ~6000 waves at once for an 8K frame, and everything
// it needs is asked for in one trip: the group totals and the sizes of its group's tiles in front of it (its place in the file), its
// segments' counts, and the first 2 KB of its stream (where those are follows from the launch's geometry, not from loaded values).
// Then the stream is stuffed in flight: a lane takes a dword, finds its segment (the segment starts are marked in LDS: a wave prefix
// sum of the marks), counts its 0xFF bytes; a second prefix sum places it; a dword without 0xFF leaves as one unaligned 4-byte store,
// restart markers follow the last dword of a segment. Round 3 needed a launch for the offsets (5 us), a wave per four
// segments with two dependent trips and byte stores (16 us), and 19 MB of traffic for the same.
// ================================================================================================
typedef uint32_t __attribute__((aligned(1))) gj_u32_unaligned;
struct GjTail {
    uint32_t* piece;       // [npieces] size in the file of every tile stream, in FILE order
    uint32_t* group;       // [ngroups] bytes of the tile streams 32 g .. 32 g + 31, added up by the tiles themselves; zero when the encoder kernel starts
    uint32_t* group_other; // the next call's
    uint32_t ngroups, npieces;
    const uint8_t* temp;
    const uint32_t* seg_bytes;
    const uint32_t* seg_ff;
    uint8_t* jpeg;
    uint64_t capacity;
    const uint8_t* scan_hdr;
    uint32_t hdr_end[GJ_MAX_COMP];     // bytes of the scan headers up to and including scan s
    uint32_t scan_first[GJ_MAX_COMP];  // index in the list of the first tile stream of scan s (0xFFFFFFFF behind the last scan)
    uint32_t seg_first[GJ_MAX_COMP];   // global index of the scan's first segment
    uint32_t segs[GJ_MAX_COMP];        // segments of the scan
    uint32_t block_first[GJ_MAX_COMP]; // coding-order index of the scan's first block (addresses d_temp)
    uint32_t spt, seg_blocks;          // segments per tile, blocks per full segment
    uint64_t temp_blocks;              // blocks d_temp has room for (the last segment of a scan may be shorter than seg_blocks: its tile's area ends early)
    uint32_t main_hdr;
    uint32_t* d_result;
    uint32_t* h_result;
    // frame batch (gj_enc_job::batch: blockIdx.z = frame): what lies between the buffers of two frames; all zero for a single frame
    uint64_t f_raw, f_temp, f_jpeg; // bytes
    uint32_t f_seg, f_tail;         // words of seg_bytes / seg_ff, of the tile list and of the group totals
    // k_encode_rgb444: the last tiles of a frame larger than the GPU are coded one component per workgroup (see there)
    uint32_t tail_from, tiles;      // first tile that is split (0xFFFFFFFF: none), tiles of the frame
};

__device__ __forceinline__ uint32_t gj_pick4(const uint32_t (&a)[GJ_MAX_COMP], const uint32_t s)
{
    return s == 0 ? a[0] : s == 1 ? a[1] : s == 2 ? a[2] : a[3];
}
// the scan tile stream p belongs to
__device__ __forceinline__ uint32_t gj_tail_scan_of(const GjTail& T, const uint32_t p)
{
    return (p >= T.scan_first[1] ? 1u : 0u) + (p >= T.scan_first[2] ? 1u : 0u) + (p >= T.scan_first[3] ? 1u : 0u);
}
// an encoder workgroup's entry for tile stream p
__device__ __forceinline__ void gj_piece_put(const GjTail& T, const uint32_t p, const uint32_t size, const size_t frame_words = 0)
{
    T.piece[frame_words + p] = size;
    (void)__hip_atomic_fetch_add(&T.group[frame_words + (p >> 5)], size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define GJ_GATHER_PRELOAD 8 // rounds of 64 dwords whose loads are issued before anything is known about the tile stream
#define GJ_GATHER_MARK_DW 2048 // dwords of a tile stream whose segment starts are marked in LDS at a time
__global__ __launch_bounds__(256) void k_gather(const GjTail T0)
{
    GjTail T = T0;
    { // frame blockIdx.z of a batch: its own tile list, group totals, tile streams, segment counts, stream buffer and result words
        const size_t fz = blockIdx.z;
        T.piece += fz * T0.f_tail; T.group += fz * T0.f_tail; T.group_other += fz * T0.f_tail;
        T.temp += fz * T0.f_temp; T.seg_bytes += fz * T0.f_seg; T.seg_ff += fz * T0.f_seg;
        T.jpeg += fz * T0.f_jpeg; T.d_result += fz * 2;
        if (T.h_result) T.h_result += fz * 2;
    }
    __shared__ uint32_t s_tmp[4];
    __shared__ __attribute__((aligned(16))) uint8_t s_mark[4][GJ_GATHER_MARK_DW];
    const int i = threadIdx.x, lane = i & 63, wave = i >> 6;
    const uint32_t P = T.npieces, NG = T.ngroups;
    const uint32_t p0 = blockIdx.x * 4u, p = p0 + (uint32_t)wave;
    const bool have = p < P;
    if (blockIdx.x == 0) // the next call's group totals
        for (uint32_t g = i; g < NG; g += 256) T.group_other[g] = 0;
    // ---- this wave's tile stream: where its segments and its bytes are (no loaded value needed)
    const uint32_t scan = gj_tail_scan_of(T, have ? p : 0u);
    const uint32_t t = (have ? p : 0u) - gj_pick4(T.scan_first, scan), seg0 = t * T.spt, scan_segs = gj_pick4(T.segs, scan);
    const uint32_t nseg = have ? min(T.spt, scan_segs - seg0) : 0u;
    const uint32_t s0 = gj_pick4(T.seg_first, scan) + seg0;
    const uint64_t src_block = (uint64_t)gj_pick4(T.block_first, scan) + (uint64_t)seg0 * T.seg_blocks;
    const uint32_t* const src = reinterpret_cast<const uint32_t*>(T.temp + src_block * GJ_TEMP_BYTES_PER_BLOCK);
    // dwords of the tile's area: nothing is read behind them, nor behind the buffer (the area of a scan's last tile ends with its last,
    // shorter segment; found by the execution model under AddressSanitizer in round 4: the preload below read up to 35 blocks further)
    const uint64_t room = T.temp_blocks > src_block ? T.temp_blocks - src_block : 0u;
    const uint32_t src_dw = (uint32_t)min((uint64_t)nseg * T.seg_blocks, room) * (GJ_TEMP_BYTES_PER_BLOCK / 4u);
    // ---- one trip: group totals, the sizes of the tiles of the group in front of the workgroup's first one and of the workgroup's
    // own, the segments' counts, the first rounds of the stream
    const uint32_t ga = p0 >> 5;
    uint32_t before = 0, all = 0;
    for (uint32_t g0 = 0; g0 < NG; g0 += 1024) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t g = g0 + (uint32_t)u * 256u + (uint32_t)i;
            v[u] = g < NG ? T.group[g] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t g = g0 + (uint32_t)u * 256u + (uint32_t)i;
            all += v[u];
            before += g < ga ? v[u] : 0u;
        }
    }
    if ((uint32_t)i < (p0 & 31u)) before += T.piece[(ga << 5) + i];
    uint32_t in_front = 0; // of this wave's tile inside the workgroup
    if (wave > 0 && have) in_front += T.piece[p0];
    if (wave > 1 && have) in_front += T.piece[p0 + 1];
    if (wave > 2 && have) in_front += T.piece[p0 + 2];
    uint32_t nb = 0, ff = 0;
    if ((uint32_t)lane < nseg) {
        nb = T.seg_bytes[s0 + lane];
        ff = T.seg_ff[s0 + lane];
    }
    uint32_t pre[GJ_GATHER_PRELOAD];
#pragma unroll
    for (int m = 0; m < GJ_GATHER_PRELOAD; m++) {
        const uint32_t d = (uint32_t)m * 64u + (uint32_t)lane;
        pre[m] = d < src_dw ? src[d] : 0u;
    }
    uint32_t done_bytes, all_bytes;
    gj_wg256_incl_scan(before, s_tmp, &done_bytes);
    gj_wg256_incl_scan(all, s_tmp, &all_bytes);
    const uint64_t total = (uint64_t)T.main_hdr + gj_pick4(T.hdr_end, gj_tail_scan_of(T, P - 1)) + all_bytes + 2u;
    const bool overflow = total > T.capacity;
    if (p0 + 4 >= P && i == 0) { // (the workgroup of the last tile stream)
        T.d_result[0] = (uint32_t)total;
        T.d_result[1] = overflow ? 1u : 0u;
        if (T.h_result) { // the host's (pinned, device-visible) copy: no copy launch behind the kernel
            T.h_result[0] = (uint32_t)total;
            T.h_result[1] = overflow ? 1u : 0u;
        }
    }
    if (!have || overflow) return;
    const uint32_t F = T.main_hdr + gj_pick4(T.hdr_end, scan) + done_bytes + in_front; // the tile stream's first byte in the file
    uint8_t* const out = T.jpeg;
    if (t == 0) { // first tile of a scan: its header (APP13 placeholders + SOS) sits right in front
        const uint32_t h1 = gj_pick4(T.hdr_end, scan), h0 = scan == 0 ? 0u : gj_pick4(T.hdr_end, scan - 1);
        for (uint32_t b = lane; b < h1 - h0; b += 64) out[F - (h1 - h0) + b] = T.scan_hdr[h0 + b];
    }
    // ---- lane sl keeps segment sl: its dwords, its first dword in the tile stream, its first byte in the file, the 0xFF bytes in front
    // of it. Everything a dword needs to know about its segment is two words: where its bytes go if none of the tile's dwords held a
    // 0xFF (minus 4 x its index), and the segment's last dword with the bytes that count in it.
    const uint32_t last = scan_segs - seg0 - 1u; // (local index of the scan's last segment: no restart marker behind it)
    const uint32_t ndw = (nb + 3u) >> 2, olen = nb + ff + ((uint32_t)lane < nseg && (uint32_t)lane != last ? 2u : 0u);
    const uint32_t dwi = gj_wave_incl_scan(ndw), dwb = dwi - ndw;
    const uint32_t oi = gj_wave_incl_scan(olen), ob = F + oi - olen;
    const uint32_t ffi = gj_wave_incl_scan(ff);
    const uint32_t total_dw = (uint32_t)__builtin_amdgcn_readlane((int)dwi, 63);
    const uint32_t end = F + (uint32_t)__builtin_amdgcn_readlane((int)oi, 63);
    if (p == P - 1 && lane == 0) { // EOI
        out[end] = 0xFF;
        out[end + 1] = 0xD9;
    }
    const uint32_t seg_a = ob - 4u * dwb - (ffi - ff);                               // byte q of dword d: seg_a + 4 d + 0xFF bytes in front of d
    const uint32_t seg_e = (dwi - 1u) | ((nb - 4u * (ndw - 1u)) << 28);              // last dword | its bytes (1 .. 4) << 28
    // which dwords begin a segment: a byte per dword in LDS (tile streams of more dwords take several passes)
    uint8_t* const mark = s_mark[wave];
    uint32_t ffrun = 0, segs_before = 0; // 0xFF bytes / segment starts of the dwords of earlier rounds
    for (uint32_t c0 = 0; c0 < total_dw; c0 += GJ_GATHER_MARK_DW) {
        const uint32_t c1 = min(total_dw, c0 + (uint32_t)GJ_GATHER_MARK_DW);
        gj_wave_sync();
#pragma unroll
        for (int z = 0; z < GJ_GATHER_MARK_DW / 256; z++) reinterpret_cast<uint32_t*>(mark)[z * 64 + lane] = 0;
        gj_wave_sync();
        if ((uint32_t)lane < nseg && dwb >= c0 && dwb < c1) mark[dwb - c0] = 1;
        gj_wave_sync();
        for (uint32_t m = c0 >> 6; m * 64u < c1; m++) {
            const uint32_t d = m * 64u + (uint32_t)lane;
            const bool live = d < total_dw;
            uint32_t v = 0;
            if (m < GJ_GATHER_PRELOAD) {
#pragma unroll
                for (int q = 0; q < GJ_GATHER_PRELOAD; q++)
                    if (m == (uint32_t)q) v = pre[q];
            } else if (live) {
                v = src[d];
            }
            const uint32_t seg = gj_wave_incl_scan(live ? mark[d - c0] : 0u) + segs_before - 1u; // the segment of dword d
            const int sidx = (int)((live ? seg : 0u) << 2);
            const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(sidx, (int)seg_a), e = (uint32_t)__builtin_amdgcn_ds_bpermute(sidx, (int)seg_e);
            const bool ends = live && d == (e & 0x0FFFFFFFu);
            const int vb = live ? (ends ? (int)(e >> 28) : 4) : 0;
            const uint32_t ffm = live ? ((v & 0x7F7F7F7Fu) + 0x01010101u) & v & 0x80808080u : 0u; // (the bytes behind a segment's end are zero)
            const uint32_t ffc = (uint32_t)__builtin_popcount(ffm);
            const uint32_t fi = gj_wave_incl_scan(ffc) + ffrun; // 0xFF bytes up to and including this dword
            if (live) {
                uint32_t q = a + 4u * d + (fi - ffc);
                if (ffc == 0 && vb == 4) {
                    *reinterpret_cast<gj_u32_unaligned*>(out + q) = v;
                    q += 4;
                } else {
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        if (b < vb) {
                            const uint32_t byte = (v >> (8 * b)) & 0xFFu;
                            out[q++] = (uint8_t)byte;
                            if (byte == 0xFFu) out[q++] = 0;
                        }
                }
                if (ends && seg != last) { // RSTn (src/gpujpeg_huffman_gpu_encoder.cu:497-502)
                    out[q] = 0xFF;
                    out[q + 1] = (uint8_t)(0xD0 + ((seg0 + seg) & 7u));
                }
            }
            ffrun = (uint32_t)__builtin_amdgcn_readlane((int)fi, 63);
            segs_before = (uint32_t)__builtin_amdgcn_readlane((int)seg, 63) + 1u;
        }
    }
}

// the workgroup's Huffman tables in the layout of GjCoderLds::lut: the host has them ready behind its (code << 8 | size) tables
// (gj_enc_job::d_huff_lut + GJ_CODER_LUT_OFFSET, gj_huffman_coder_lut; worked out in the kernel they cost every wave ~60 vector instructions, round 5)
__device__ __forceinline__ void gj_load_coder_lut(uint32_t* s_lut, const uint32_t* __restrict__ lut, const int i)
{
    static_assert(GJ_CODER_LUT_WORDS == 2 * 272 && GJ_CODER_LUT_WORDS % 4 == 0, "layout of GjCoderLds::lut");
    if (i < GJ_CODER_LUT_WORDS / 4) reinterpret_cast<uint4*>(s_lut)[i] = reinterpret_cast<const uint4*>(lut + GJ_CODER_LUT_OFFSET)[i];
}

// ================================================================================================
// Fully fused fast path: packed 4:4:4 pixels -> per-segment (unstuffed) Huffman streams, no coefficient planes.
//
// k_fused_rgb444 + k_huffman move 2 x 199 MB of int16 coefficients through HBM for an 8K frame; measured, the store half
// alone costs as much as all arithmetic of the kernel. Both kernels already give one thread one 8x8 block, so the
// quantised block can stay with that thread: a workgroup takes spt = 256 / B whole restart segments (B blocks
// each, e.g. 7 x 36 = 252 block positions) of ALL THREE component scans, colour-converts its pixels once, then for one
// component after the other transforms the block into its LDS column and runs the coder above on it. The output (the tile's unstuffed
// stream in d_temp, byte and 0xFF counts per segment, the stream's size in the file) is what k_gather turns into the file.
// Used for non-interleaved 4:4:4 with restart intervals of 4 .. 256 blocks.
// ================================================================================================
template <int CS_FROM, int CS_TO, bool ONE_COMPONENT = false>
__global__ __launch_bounds__(256, 4) void k_encode_rgb444(const gj_geom g, const uint8_t* __restrict__ raw, const float* __restrict__ q_luma,
                                                          const float* __restrict__ q_chroma, const uint32_t* __restrict__ lut,
                                                          uint8_t* __restrict__ temp, uint32_t* __restrict__ seg_bytes,
                                                          uint32_t* __restrict__ seg_ff, const GjTail T)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_coef[32 * 256];
    __shared__ __attribute__((aligned(8))) float s_q[3][64];
    __shared__ __attribute__((aligned(16))) uint32_t s_lut[2 * 272];
    __shared__ uint32_t s_wsum[4];
    __shared__ int s_edge[64];
    __shared__ uint32_t s_segx[GJ_ENC_MAX_SPT], s_segend[GJ_ENC_MAX_SPT], s_segbase[GJ_ENC_MAX_SPT + 1], s_segbits[GJ_ENC_MAX_SPT], s_segff[GJ_ENC_MAX_SPT];
    const GjCoderLds L = {s_coef, s_lut, s_wsum, s_edge, s_segx, s_segend, s_segbase, s_segbits, s_segff};

    const int i = threadIdx.x;
    GJ_TRACE_E(0);
    gj_load_coder_lut(s_lut, lut, i);
    if (i < 192) s_q[i >> 6][i & 63] = (g.comp[i >> 6].type ? q_chroma : q_luma)[i & 63];
    // frame blockIdx.z of a batch (gj_enc_job::batch; strides zero for a single frame)
    const size_t fz = blockIdx.z;
    raw += fz * T.f_raw;
    temp += fz * T.f_temp;
    seg_bytes += fz * T.f_seg;
    seg_ff += fz * T.f_seg;

    const gj_comp_geom& k0 = g.comp[0];
    const int B = g.seg_blocks;
    const int spt = 256 / B;       // segments per workgroup (per component)
    const int tile_blocks = spt * B;
    const uint32_t recip = (65536u + (uint32_t)B - 1u) / (uint32_t)B; // j = i / B through a 16.16 reciprocal (exact for i < 256, B <= 256)
    const int j = min((int)(((uint32_t)i * recip) >> 16), GJ_ENC_MAX_SPT - 1);
    const int k = i - j * B;       // block inside its segment
    // The tile and, when the workgroup codes ONE component of it, which. Three shapes of launch:
    //   all three components per workgroup (frames from ~half a generation of workgroups up),
    //   ONE_COMPONENT with gridDim.y == 3 (small frames: see below),
    //   and a mixture: the LAST tiles of a frame that has more tiles than the GPU has places (8K: 2058 tiles for 1024 places) as three short workgroups
    //   each, behind the whole ones in the grid. A launch ends with the workgroups that started last; the whole tiles of the last, under-filled
    //   generation run on nearly empty CUs at the latency of one workgroup (~36 us), the split ones start as soon as the first places come free
    //   and take ~40 % of that. 8K alone: 80.7 -> 73.9 us with the last 16 ... 96 tiles split, nothing lost with four pipelines up to 32
    //   (profiles/r5_09_encoder_tail_tiles_split.txt); the verdict's "lone-launch tax".
    unsigned tile = blockIdx.x, ntiles = gridDim.x;
    int only = ONE_COMPONENT ? (int)blockIdx.y : -1;
    if (!ONE_COMPONENT && T.tail_from != 0xFFFFFFFFu) {
        ntiles = T.tiles;
        if (blockIdx.x >= T.tail_from) {
            const unsigned r = blockIdx.x - T.tail_from, q = r / 3u;
            tile = T.tail_from + q;
            only = (int)(r - 3u * q);
        }
    }
    const int seg0 = (int)tile * spt; // first segment (inside each component's scan)
    const unsigned nb = (unsigned)(k0.blocks_x * k0.blocks_y);
    const unsigned lb = tile * (unsigned)tile_blocks + (unsigned)i;
    const bool active = i < tile_blocks && lb < nb; // (every component has the same geometry)
    // the block position: the tile's first block by one division of uniform values, the lane's by carrying over the ends of the block rows (a lane
    // without a block of its own -- tile slack, behind the last block -- takes the frame's last one: nobody looks at what it makes of it)
    unsigned bx, by;
    {
        const unsigned bxn = (unsigned)k0.blocks_x, lbc = min(lb, nb - 1u);
        if (bxn >= 256u) {
            const unsigned lb0 = tile * (unsigned)tile_blocks;
            const unsigned by0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(lb0 / bxn));
            bx = lbc - by0 * bxn;
            by = by0;
            if (bx >= bxn) { bx -= bxn; by++; } // (a tile of 256 blocks crosses the end of a block row once at most)
        } else {
            by = lbc / bxn;
            bx = lbc - by * bxn;
        }
    }

    // ---- pixels -> three byte-packed component blocks
    uint32_t pk[3][16];
    gj_load_color_444<CS_FROM, CS_TO>(g, raw, bx, by, pk);
    __syncthreads(); // tables are in LDS
    GJ_TRACE_E(1); // pixels loaded and converted

    // A small frame has fewer tiles than the GPU has places for workgroups (HD: 135 for 1024): ONE_COMPONENT, launched with gridDim.y == 3, codes
    // ONE component of its tile -- the pixels are loaded and converted three times, by CUs that would otherwise idle, and a tile's components
    // run side by side instead of one after the other.
    // (a template parameter: the check costs the three-component instantiation of the 8K frame 0.7 us when it is made at run time)
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (only >= 0 && c != only) continue;
        const gj_comp_geom& kc = g.comp[c];
        // (pinned: the transform of component c + 1 would otherwise be hoisted over the coder of c)
#pragma unroll
        for (int t = 0; t < 16; t++) GJ_KEEP(pk[c][t]);
        gj_fdct_quant_zz(pk[c], s_q[c], reinterpret_cast<uint8_t*>(s_coef) + i * 4);
        GJ_TRACE_E(2 + 4 * c); // transformed (this wave)
        const uint64_t first_block = kc.data_offset / 64 + (uint64_t)seg0 * B; // coding-order index of the tile's first block of this component
        const uint32_t size = gj_code_tile(L, i, j, k, active, spt, active ? min(B, (int)nb - (seg0 + j) * B) : 0, kc.type, 1, k0.segment_count - seg0,
                                           temp + first_block * GJ_TEMP_BYTES_PER_BLOCK, seg_bytes, seg_ff, (uint32_t)(kc.first_segment + seg0), 2 + 4 * c);
        // file order: the luminance scan's tiles, then the two chrominance scans'
        if (i == 0) gj_piece_put(T, (uint32_t)c * ntiles + tile, size, fz * T.f_tail);
    }
}

// ================================================================================================
// k_encode_rgb444's counterpart for interleaved packed 4:2:2 without colour transform (BASELINE config 4): one lane per
// block in CODING order (Y0 Y1 Cb Cr of MCU 0, of MCU 1, ...), a workgroup takes spt = 256 / B whole restart segments
// (B = 4 x restart interval blocks each). All four lanes of an MCU read its 8 x 32 bytes (the same addresses merge in
// the load unit), pick their own samples with byte permutes, transform, and the coder runs once on the whole
// tile -- no coefficient planes, one pass instead of k_encode_rgb444's three.
// (The DC predecessor distance and the table are per lane here.)
// ================================================================================================
__global__ __launch_bounds__(256, 4) void k_encode_uyvy422(const gj_geom g, const uint8_t* __restrict__ raw, const float* __restrict__ q_luma,
                                                           const float* __restrict__ q_chroma, const uint32_t* __restrict__ lut,
                                                           uint8_t* __restrict__ temp, uint32_t* __restrict__ seg_bytes,
                                                           uint32_t* __restrict__ seg_ff, const GjTail T)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_coef[32 * 256];
    __shared__ __attribute__((aligned(8))) float s_q[2][64];
    __shared__ __attribute__((aligned(16))) uint32_t s_lut[2 * 272];
    __shared__ uint32_t s_wsum[4];
    __shared__ int s_edge[64];
    __shared__ uint32_t s_segx[GJ_ENC_MAX_SPT], s_segend[GJ_ENC_MAX_SPT], s_segbase[GJ_ENC_MAX_SPT + 1], s_segbits[GJ_ENC_MAX_SPT], s_segff[GJ_ENC_MAX_SPT];
    const GjCoderLds L = {s_coef, s_lut, s_wsum, s_edge, s_segx, s_segend, s_segbase, s_segbits, s_segff};

    const int i = threadIdx.x;
    gj_load_coder_lut(s_lut, lut, i);
    if (i < 128) s_q[i >> 6][i & 63] = (i < 64 ? q_luma : q_chroma)[i & 63];
    // frame blockIdx.z of a batch (gj_enc_job::batch; strides zero for a single frame)
    raw += (size_t)blockIdx.z * T.f_raw;
    temp += (size_t)blockIdx.z * T.f_temp;
    seg_bytes += (size_t)blockIdx.z * T.f_seg;
    seg_ff += (size_t)blockIdx.z * T.f_seg;

    const gj_comp_geom& kc = g.comp[1];
    const int ri = g.restart_interval;
    const int B = g.seg_blocks;    // 4 x ri
    const int spt = 256 / B;       // segments per workgroup
    const int tile_blocks = spt * B;
    const uint32_t recip = (65536u + (uint32_t)B - 1u) / (uint32_t)B; // j = i / B through a 16.16 reciprocal (exact for i < 256, B <= 256)
    const int j = min((int)(((uint32_t)i * recip) >> 16), GJ_ENC_MAX_SPT - 1);
    const int k = i - j * B;       // block inside its segment
    const int p = k & 3;           // position inside the MCU: Y0 Y1 Cb Cr
    const int seg0 = blockIdx.x * spt;
    const unsigned m = (unsigned)(seg0 + j) * (unsigned)ri + (unsigned)(k >> 2); // MCU
    const unsigned nm = (unsigned)g.mcu_count;
    const bool active = i < tile_blocks && seg0 + j < g.segment_count && m < nm;
    const unsigned my = m / (unsigned)kc.blocks_x, mx = m - my * (unsigned)kc.blocks_x;

    // ---- pixels -> this lane's byte-packed block
    uint32_t px[16];
    {
        const size_t pitch = (size_t)g.width * 2 + g.width_padding;
        const bool interior = (mx * 16 + 16 <= (unsigned)g.width) && (my * 8 + 8 <= (unsigned)g.height);
        const bool aligned = ((pitch | (size_t)raw) & 15) == 0;
        if (!active) {
#pragma unroll
            for (int t = 0; t < 16; t++) px[t] = 0;
        } else if (interior && aligned) {
            const uint4* src = reinterpret_cast<const uint4*>(raw + (size_t)(my * 8) * pitch + (size_t)mx * 32);
            const size_t pitch4 = pitch >> 4;
            const int first = p == 1; // Y1 lives in the second 16 bytes of the row; chroma needs both halves
            const uint32_t selc = p == 2 ? 0x05040100u : 0x07060302u;
            uint4 lo[8], hi[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                lo[r] = src[r * pitch4 + first];
                hi[r] = src[r * pitch4 + 1];
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t y0 = __builtin_amdgcn_perm(lo[r].y, lo[r].x, 0x07050301u), y1 = __builtin_amdgcn_perm(lo[r].w, lo[r].z, 0x07050301u);
                const uint32_t uv01 = __builtin_amdgcn_perm(lo[r].y, lo[r].x, 0x06020400u), uv23 = __builtin_amdgcn_perm(lo[r].w, lo[r].z, 0x06020400u);
                const uint32_t uv45 = __builtin_amdgcn_perm(hi[r].y, hi[r].x, 0x06020400u), uv67 = __builtin_amdgcn_perm(hi[r].w, hi[r].z, 0x06020400u);
                const uint32_t c0 = __builtin_amdgcn_perm(uv23, uv01, selc), c1 = __builtin_amdgcn_perm(uv67, uv45, selc);
                px[2 * r] = p < 2 ? y0 : c0;
                px[2 * r + 1] = p < 2 ? y1 : c1;
            }
        } else {
            // samples outside the image are zero component values (src/gpujpeg_common.c:941-944); the odd last pixel of an
            // odd-width row shares the chroma of its pair like the generic loader does
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const unsigned y = my * 8 + r;
                uint32_t d[2] = {0, 0};
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    uint32_t v = 0;
                    if (y < (unsigned)g.height) {
                        if (p < 2) {
                            const unsigned x = mx * 16 + p * 8 + t;
                            if (x < (unsigned)g.width) v = raw[(size_t)y * pitch + (size_t)x * 2 + 1];
                        } else {
                            const unsigned cx = mx * 8 + t;
                            if (cx < (unsigned)kc.width) v = raw[(size_t)y * pitch + (size_t)cx * 4 + (p == 2 ? 0 : 2)];
                        }
                    }
                    d[t >> 2] |= v << (8 * (t & 3));
                }
                px[2 * r] = d[0];
                px[2 * r + 1] = d[1];
            }
        }
    }
    __syncthreads(); // tables are in LDS
    {
        const int table = p < 2 ? g.comp[0].type : g.comp[1].type;
#pragma unroll
        for (int t = 0; t < 16; t++) GJ_KEEP(px[t]);
        gj_fdct_quant_zz(px, s_q[table ? 1 : 0], reinterpret_cast<uint8_t*>(s_coef) + i * 4);
        const uint64_t first_block = (uint64_t)seg0 * B;
        const uint32_t size = gj_code_tile(L, i, j, k, active, spt, active ? min(B, ((int)nm - (seg0 + j) * ri) * 4) : 0, table,
                                           p == 0 ? 3 : (p == 1 ? 1 : 4) /* Y1 follows the Y0 of its own MCU */, g.segment_count - seg0,
                                           temp + first_block * GJ_TEMP_BYTES_PER_BLOCK, seg_bytes, seg_ff, (uint32_t)seg0);
        if (i == 0) gj_piece_put(T, blockIdx.x, size, (size_t)blockIdx.z * T.f_tail);
    }
}

// ================================================================================================
// The fully fused encoder for every other layout with restart segments of 4 .. 256 blocks: one lane per block in CODING order, whatever
// the scan structure (gj_segment_block gives the lane its component and block position), a workgroup takes spt = 256 / B whole
// restart segments of one scan, every lane fetches the 64 samples of ITS block, transforms them and the coder of k_encode_rgb444
// runs once on the tile. Replaces k_preprocess (one thread per pixel, byte loads and stores) + k_dct + k_huffman and their planes
// (src/gpujpeg_preprocessor.cu:173-292 has one specialised kernel per sampling; here the sampling is the lane's address arithmetic).
//   PLANAR: planar / grey input whose layout equals the component layout (the reference's copy path, :397-453)
//   !PLANAR: packed 4:4:4 pixels with a colour transform from RGB (or none) and point-sampled chroma (:49-63): the lane computes only
//            its own component, out of the pixels (x * sub_h, y * sub_v)
// ================================================================================================
// row of the colour matrix that produces component c (RGB -> CS_TO), pre-divided by 256 with offset + 0.5 / 256 (see gj_matrix_to_f)
__device__ __forceinline__ void gj_matrix_row(const int cs_to, const int c, float& m0, float& m1, float& m2, float& off)
{
    static constexpr int M[3][9] = {{66, 129, 25, -38, -74, 112, 112, -94, -18},    // BT.601 limited
                                    {77, 150, 29, -43, -85, 128, 128, -107, -21},   // BT.601 full range (JPEG)
                                    {47, 157, 16, -26, -87, 112, 112, -102, -10}};  // BT.709
    static constexpr int BASE[3][3] = {{16, 128, 128}, {0, 128, 128}, {16, 128, 128}};
    const int t = cs_to == GJ_CS_BT601 ? 0 : cs_to == GJ_CS_BT601_256 ? 1 : 2;
    const float s = 1.0f / 256.0f;
    m0 = (float)M[t][c * 3] * s;
    m1 = (float)M[t][c * 3 + 1] * s;
    m2 = (float)M[t][c * 3 + 2] * s;
    off = (float)BASE[t][c] + 0.5f / 256.0f;
}

template <bool PLANAR>
__global__ __launch_bounds__(256, 4) void k_encode_blocks(const gj_geom g, const uint8_t* __restrict__ raw, const float* __restrict__ q_luma,
                                                          const float* __restrict__ q_chroma, const uint32_t* __restrict__ lut,
                                                          uint8_t* __restrict__ temp, uint32_t* __restrict__ seg_bytes,
                                                          uint32_t* __restrict__ seg_ff, const GjTail T)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_coef[32 * 256];
    __shared__ __attribute__((aligned(8))) float s_q[2][64];
    __shared__ __attribute__((aligned(16))) uint32_t s_lut[2 * 272];
    __shared__ uint32_t s_wsum[4];
    __shared__ int s_edge[64];
    __shared__ uint32_t s_segx[GJ_ENC_MAX_SPT], s_segend[GJ_ENC_MAX_SPT], s_segbase[GJ_ENC_MAX_SPT + 1], s_segbits[GJ_ENC_MAX_SPT], s_segff[GJ_ENC_MAX_SPT];
    const GjCoderLds L = {s_coef, s_lut, s_wsum, s_edge, s_segx, s_segend, s_segbase, s_segbits, s_segff};

    const int i = threadIdx.x;
    gj_load_coder_lut(s_lut, lut, i);
    if (i < 128) s_q[i >> 6][i & 63] = (i < 64 ? q_luma : q_chroma)[i & 63];
    // frame blockIdx.z of a batch (gj_enc_job::batch; strides zero for a single frame)
    raw += (size_t)blockIdx.z * T.f_raw;
    temp += (size_t)blockIdx.z * T.f_temp;
    seg_bytes += (size_t)blockIdx.z * T.f_seg;
    seg_ff += (size_t)blockIdx.z * T.f_seg;

    const int B = g.seg_blocks;
    const int spt = 256 / B;       // segments per workgroup
    // tiles never cross a scan: only the last segment of a scan may be short, and it has to be the last one of its tile
    int scan = 0, tile = (int)blockIdx.x, scan_first = 0, scan_segs = g.segment_count;
    if (!g.interleaved) {
        for (int c = 0; c < g.comp_count; c++) {
            const int tiles_c = (g.comp[c].segment_count + spt - 1) / spt;
            if (tile < tiles_c || c == g.comp_count - 1) { scan = c; break; }
            tile -= tiles_c;
        }
        scan_first = g.comp[scan].first_segment;
        scan_segs = g.comp[scan].segment_count;
    }
    const int seg0 = tile * spt;   // first segment of the tile inside its scan
    const uint32_t recip = (65536u + (uint32_t)B - 1u) / (uint32_t)B; // j = i / B through a 16.16 reciprocal (exact for i < 256, B <= 256)
    const int j = min((int)(((uint32_t)i * recip) >> 16), GJ_ENC_MAX_SPT - 1);
    const int k = i - j * B;       // block inside its segment
    GjSeg sg;
    sg.nblocks = 0;
    sg.first_block = 0;
    const bool seg_valid = i < spt * B && seg0 + j < scan_segs;
    if (seg_valid) sg = gj_segment(g, scan_first + seg0 + j);
    const bool active = seg_valid && k < sg.nblocks;
    __shared__ uint64_t s_first_block; // coding-order index of the tile's first block (thread 0: j = k = 0)
    if (i == 0) s_first_block = sg.first_block;

    // ---- the lane's block: component, position, samples
    int comp = 0, mcu_pos = 0;
    unsigned bx = 0, by = 0;
    if (active) {
        const uint64_t off = gj_segment_block(g, sg, k, &comp, &mcu_pos);
        const unsigned blk = (unsigned)((off - g.comp[comp].data_offset) >> 6);
        by = blk / (unsigned)g.comp[comp].blocks_x;
        bx = blk - by * (unsigned)g.comp[comp].blocks_x;
    }
    const gj_comp_geom& kc = g.comp[comp];
    uint32_t px[16];
#pragma unroll
    for (int t = 0; t < 16; t++) px[t] = 0;
    if (active && PLANAR) {
        // raw planes back to back, pitch = component width + padding (src/gpujpeg_preprocessor.cu:414-448); outside: zeros
        size_t src_off = 0;
        for (int c = 0; c < comp; c++) src_off += ((size_t)g.comp[c].width + g.width_padding) * g.comp[c].height;
        const size_t pitch = (size_t)kc.width + g.width_padding;
        const uint8_t* p0 = raw + src_off + (size_t)(by * 8) * pitch + bx * 8;
        const bool interior = bx * 8 + 8 <= (unsigned)kc.width && by * 8 + 8 <= (unsigned)kc.height;
        if (interior && ((pitch | (size_t)p0) & 3) == 0) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t* p = reinterpret_cast<const uint32_t*>(p0 + (size_t)r * pitch);
                px[2 * r] = p[0];
                px[2 * r + 1] = p[1];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                uint32_t d[2] = {0, 0};
                if (by * 8 + r < (unsigned)kc.height) {
#pragma unroll
                    for (int t = 0; t < 8; t++)
                        if (bx * 8 + t < (unsigned)kc.width) d[t >> 2] |= (uint32_t)p0[(size_t)r * pitch + t] << (8 * (t & 3));
                }
                px[2 * r] = d[0];
                px[2 * r + 1] = d[1];
            }
        }
    }
    if (active && !PLANAR) {
        const unsigned sh = (unsigned)kc.sub_h, sv = (unsigned)kc.sub_v;
        const size_t pitch = (size_t)g.width * 3 + g.width_padding;
        const bool transform = g.color_space != g.color_space_internal && g.color_space != GJ_CS_NONE && g.color_space_internal != GJ_CS_NONE;
        float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f, off = 0.5f / 256.0f;
        if (transform) gj_matrix_row(g.color_space_internal, comp, m0, m1, m2, off);
        else { m0 = comp == 0 ? 1.0f : 0.0f; m1 = comp == 1 ? 1.0f : 0.0f; m2 = comp == 2 ? 1.0f : 0.0f; off = 0.25f; } // (identity: the chosen channel + 0.25 rounds to itself)
        const unsigned x0 = bx * 8 * sh; // first pixel of the row
        // rows whose 8 * sub_h pixels all exist are fetched as 6 (sub_h = 1) or 12 (sub_h = 2) aligned dwords; measured against one unaligned
        // dword load per sampled pixel (no divergence between the luminance and chrominance lanes of a wave): twice as fast
        const bool fast = sh <= 2 && ((pitch | (size_t)raw) & 3) == 0 && x0 + 8 * sh <= (unsigned)g.width && (by * 8 + 7) * sv < (unsigned)g.height;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const unsigned y = (by * 8 + r) * sv;
            float v[8];
            if (fast) {
                const uint32_t* p = reinterpret_cast<const uint32_t*>(raw + (size_t)y * pitch + (size_t)x0 * 3);
                float rr[8], gg[8], bb[8];
                if (sh == 1) {
                    uint32_t w[6];
#pragma unroll
                    for (int t = 0; t < 6; t++) w[t] = p[t];
                    rr[0] = gj_row_byte_f<0>(w); rr[1] = gj_row_byte_f<3>(w); rr[2] = gj_row_byte_f<6>(w); rr[3] = gj_row_byte_f<9>(w);
                    rr[4] = gj_row_byte_f<12>(w); rr[5] = gj_row_byte_f<15>(w); rr[6] = gj_row_byte_f<18>(w); rr[7] = gj_row_byte_f<21>(w);
                    gg[0] = gj_row_byte_f<1>(w); gg[1] = gj_row_byte_f<4>(w); gg[2] = gj_row_byte_f<7>(w); gg[3] = gj_row_byte_f<10>(w);
                    gg[4] = gj_row_byte_f<13>(w); gg[5] = gj_row_byte_f<16>(w); gg[6] = gj_row_byte_f<19>(w); gg[7] = gj_row_byte_f<22>(w);
                    bb[0] = gj_row_byte_f<2>(w); bb[1] = gj_row_byte_f<5>(w); bb[2] = gj_row_byte_f<8>(w); bb[3] = gj_row_byte_f<11>(w);
                    bb[4] = gj_row_byte_f<14>(w); bb[5] = gj_row_byte_f<17>(w); bb[6] = gj_row_byte_f<20>(w); bb[7] = gj_row_byte_f<23>(w);
                } else { // every other pixel of 16
                    uint32_t lo[6], hi[6];
#pragma unroll
                    for (int t = 0; t < 6; t++) { lo[t] = p[t]; hi[t] = p[6 + t]; }
                    rr[0] = gj_row_byte_f<0>(lo); rr[1] = gj_row_byte_f<6>(lo); rr[2] = gj_row_byte_f<12>(lo); rr[3] = gj_row_byte_f<18>(lo);
                    rr[4] = gj_row_byte_f<0>(hi); rr[5] = gj_row_byte_f<6>(hi); rr[6] = gj_row_byte_f<12>(hi); rr[7] = gj_row_byte_f<18>(hi);
                    gg[0] = gj_row_byte_f<1>(lo); gg[1] = gj_row_byte_f<7>(lo); gg[2] = gj_row_byte_f<13>(lo); gg[3] = gj_row_byte_f<19>(lo);
                    gg[4] = gj_row_byte_f<1>(hi); gg[5] = gj_row_byte_f<7>(hi); gg[6] = gj_row_byte_f<13>(hi); gg[7] = gj_row_byte_f<19>(hi);
                    bb[0] = gj_row_byte_f<2>(lo); bb[1] = gj_row_byte_f<8>(lo); bb[2] = gj_row_byte_f<14>(lo); bb[3] = gj_row_byte_f<20>(lo);
                    bb[4] = gj_row_byte_f<2>(hi); bb[5] = gj_row_byte_f<8>(hi); bb[6] = gj_row_byte_f<14>(hi); bb[7] = gj_row_byte_f<20>(hi);
                }
#pragma unroll
                for (int t = 0; t < 8; t += 2) {
                    gj_f2 a = gj_f2{rr[t], rr[t + 1]}, b = gj_f2{gg[t], gg[t + 1]}, c = gj_f2{bb[t], bb[t + 1]};
                    if (transform) { a = gj_scale256_f(a); b = gj_scale256_f(b); c = gj_scale256_f(c); }
                    const gj_f2 o = __builtin_elementwise_fma((gj_f2)m0, a, __builtin_elementwise_fma((gj_f2)m1, b, __builtin_elementwise_fma((gj_f2)m2, c, (gj_f2)off)));
                    v[t] = o.x;
                    v[t + 1] = o.y;
                }
            } else {
                // edges, other sampling factors, unaligned rows: pixel by pixel; a sample whose pixel lies outside the image is a zero
                // COMPONENT value (src/gpujpeg_common.c:941-944)
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const unsigned x = x0 + t * sh;
                    v[t] = -1.0f; // (converts to 0)
                    if (x < (unsigned)g.width && y < (unsigned)g.height) {
                        const uint8_t* q = raw + (size_t)y * pitch + (size_t)x * 3;
                        float a = (float)q[0], b = (float)q[1], c = (float)q[2];
                        if (transform) { a = fmaxf(a, __builtin_fmaf(a, 256.0f, -65024.0f)); b = fmaxf(b, __builtin_fmaf(b, 256.0f, -65024.0f)); c = fmaxf(c, __builtin_fmaf(c, 256.0f, -65024.0f)); }
                        v[t] = __builtin_fmaf(m0, a, __builtin_fmaf(m1, b, __builtin_fmaf(m2, c, off)));
                    }
                }
            }
            uint32_t d0 = 0, d1 = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                d0 = __builtin_amdgcn_cvt_pk_u8_f32(v[t], t, d0);
                d1 = __builtin_amdgcn_cvt_pk_u8_f32(v[t + 4], t, d1);
            }
            px[2 * r] = d0;
            px[2 * r + 1] = d1;
        }
    }
    __syncthreads(); // tables are in LDS
    {
        const int table = kc.type;
#pragma unroll
        for (int t = 0; t < 16; t++) GJ_KEEP(px[t]);
        gj_fdct_quant_zz(px, s_q[table ? 1 : 0], reinterpret_cast<uint8_t*>(s_coef) + i * 4);
        const uint64_t first_block = s_first_block;
        const uint32_t size = gj_code_tile(L, i, j, k, active, spt, sg.nblocks, table, g.interleaved ? (int)g.mcu_prev[mcu_pos] : 1, scan_segs - seg0,
                                           temp + first_block * GJ_TEMP_BYTES_PER_BLOCK, seg_bytes, seg_ff, (uint32_t)(scan_first + seg0));
        // (workgroups are numbered in file order: the tiles of scan 0, of scan 1, ...)
        if (i == 0) gj_piece_put(T, blockIdx.x, size, (size_t)blockIdx.z * T.f_tail);
    }
}

// ================================================================================================
// Final offsets: exclusive prefix sum over stuffed segment sizes (+2 for RSTn except at the end of a scan) and over
// the scan headers that precede each scan. One launch of ceil(S/1024) workgroups (k_scan_segments below): per-workgroup totals
// published with an epoch tag, every workgroup adds the totals of its predecessors (at most a few hundred values) to its local scan.
// ================================================================================================
__device__ __forceinline__ uint32_t gj_segment_out_size(const gj_enc_job& J, int s, uint32_t* hdr)
{
    const GjSeg sg = gj_segment(J.g, s);
    *hdr = 0;
    if (sg.first_in_scan) {
        const int scan = J.g.interleaved ? 0 : sg.comp;
        *hdr = J.scan_hdr_offset[scan + 1] - J.scan_hdr_offset[scan];
    }
    return J.d_seg_bytes[s] + J.d_seg_ff[s] + (sg.last_in_scan ? 0u : 2u);
}

// inclusive scan over a 1024-thread workgroup; s_w needs 16 words
__device__ __forceinline__ uint32_t gj_wg1024_incl_scan(uint32_t v, uint32_t* s_w, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = gj_wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t off = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const uint32_t x = s_w[w];
        if (w < wave) off += x;
        all += x;
    }
    if (total) *total = all;
    return inc + off;
}

// One launch: every workgroup scans its 1024 segments, publishes its total tagged with the call's epoch, then adds up the
// totals of its predecessors as soon as they appear (all workgroups of a frame are resident at once and are dispatched
// in index order, so a predecessor never waits for a successor). The epoch tag makes clearing the slots unnecessary.
__global__ __launch_bounds__(1024) void k_scan_segments(const gj_enc_job J, unsigned long long* __restrict__ partial, const uint32_t epoch)
{
    __shared__ uint32_t s_w[16];
    const int S = J.g.segment_count;
    const int s = blockIdx.x * 1024 + threadIdx.x;
    uint32_t hdr = 0, v = 0;
    if (s < S) v = gj_segment_out_size(J, s, &hdr);
    uint32_t total;
    const uint32_t inc = gj_wg1024_incl_scan(v + hdr, s_w, &total);
    if (threadIdx.x == 0)
        __hip_atomic_store(&partial[blockIdx.x], ((unsigned long long)epoch << 32) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t pre = 0;
    for (unsigned t = threadIdx.x; t < blockIdx.x; t += 1024) {
        unsigned long long p;
        do {
            p = __hip_atomic_load(&partial[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        } while ((uint32_t)(p >> 32) != epoch);
        pre += (uint32_t)p;
    }
    uint32_t base;
    gj_wg1024_incl_scan(pre, s_w, &base);
    base += J.main_hdr_size;
    if (s < S) J.d_seg_out[s] = base + inc - v; // segment data start (its scan header sits right before)
    if (s == S - 1) {
        const uint32_t end = base + inc;
        const uint32_t size = end + 2; // EOI
        J.d_seg_out[S] = end;
        J.d_result[0] = size;
        J.d_result[1] = (uint64_t)size > J.jpeg_capacity ? 1u : 0u;
        if (J.h_result) { // the host's (pinned, device-visible) copy: no copy launch behind the kernels
            J.h_result[0] = size;
            J.h_result[1] = (uint64_t)size > J.jpeg_capacity ? 1u : 0u;
        }
    }
}

// ================================================================================================
// Stream assembly: one WAVE per segment. Reads the unstuffed bytes, inserts 0x00 after every 0xFF
// (ballot-free: per-lane counts + wave prefix sum), appends RSTn, and the first / last segment of a scan
// also writes the scan header / EOI. Replaces the reference's serialisation + compaction kernels and the
// host-side stitching loop (src/gpujpeg_huffman_gpu_encoder.cu:417-613, src/gpujpeg_encoder.c:567-629).
// ================================================================================================
#define GJ_ASM_SEGS 4 // segments per wave: their sizes, offsets and first 256 bytes are requested together (one memory round trip
                      // instead of four; 43 200 waves of one short segment each spent their time waiting)
__global__ __launch_bounds__(256) void k_assemble(const gj_enc_job J)
{
    const gj_geom& g = J.g;
    const int s0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * GJ_ASM_SEGS;
    const int lane = threadIdx.x & 63;
    if (s0 >= g.segment_count || J.d_result[1]) return;
    uint8_t* out = J.d_jpeg;
    uint32_t raw4[GJ_ASM_SEGS], o4[GJ_ASM_SEGS], w4[GJ_ASM_SEGS];
    const uint8_t* src4[GJ_ASM_SEGS];
#pragma unroll
    for (int q = 0; q < GJ_ASM_SEGS; q++) {
        const int s = min(s0 + q, g.segment_count - 1);
        raw4[q] = s0 + q < g.segment_count ? J.d_seg_bytes[s] : 0u;
        o4[q] = J.d_seg_out[s];
        src4[q] = J.d_temp + gj_segment(g, s).first_block * GJ_TEMP_BYTES_PER_BLOCK;
    }
#pragma unroll
    for (int q = 0; q < GJ_ASM_SEGS; q++) w4[q] = (uint32_t)lane * 4u < raw4[q] ? *reinterpret_cast<const uint32_t*>(src4[q] + lane * 4) : 0u;
#pragma unroll
    for (int q = 0; q < GJ_ASM_SEGS; q++) {
        const int s = s0 + q;
        if (s >= g.segment_count) break;
        const GjSeg sg = gj_segment(g, s);
        const uint32_t raw = raw4[q];
        const uint8_t* src = src4[q];
        uint32_t o = o4[q];
        if (sg.first_in_scan) { // scan header (APP13 placeholders + SOS) right before the first segment
            const int scan = g.interleaved ? 0 : sg.comp;
            const uint32_t h0 = J.scan_hdr_offset[scan], hn = J.scan_hdr_offset[scan + 1] - h0;
            for (uint32_t b = lane; b < hn; b += 64) out[o - hn + b] = J.d_scan_hdr[h0 + b];
        }
        for (uint32_t c0 = 0; c0 < raw; c0 += 256) {
            const uint32_t idx = c0 + lane * 4;
            uint32_t w = w4[q];
            int vb = 0;
            if (idx < raw) {
                if (c0) w = *reinterpret_cast<const uint32_t*>(src + idx);
                vb = (int)min(4u, raw - idx);
            }
            int cnt = vb;
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (b < vb && ((w >> (8 * b)) & 0xFFu) == 0xFFu) cnt++;
            const uint32_t inc = gj_wave_incl_scan((uint32_t)cnt);
            uint32_t p = o + inc - (uint32_t)cnt;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if (b < vb) {
                    const uint32_t byte = (w >> (8 * b)) & 0xFFu;
                    out[p++] = (uint8_t)byte;
                    if (byte == 0xFFu) out[p++] = 0;
                }
            }
            o += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        if (lane == 0) {
            if (!sg.last_in_scan) {
                out[o] = 0xFF;
                out[o + 1] = (uint8_t)(0xD0 + (sg.index_in_scan & 7));
            }
            if (s == g.segment_count - 1) {
                out[o] = 0xFF;
                out[o + 1] = 0xD9;
            }
        }
    }
}

// APP13 segment index (src/gpujpeg_writer.c:522-547): big-endian u32 start of every segment relative to
// the first one of its scan, plus the end position (after the dropped final RSTn).
#define GJ_MAX_HEADER_SIZE (65536 - 100)
__global__ __launch_bounds__(256) void k_segment_info(const gj_enc_job J)
{
    const gj_geom& g = J.g;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= g.segment_count || J.d_result[1]) return;
    const GjSeg sg = gj_segment(g, s);
    const int scan = g.interleaved ? 0 : sg.comp;
    const int first = g.interleaved ? 0 : g.comp[scan].first_segment;
    const int segs = g.interleaved ? g.segment_count : g.comp[scan].segment_count;
    const uint32_t data0 = J.d_seg_out[first];
    const uint32_t hdr_begin = data0 - (J.scan_hdr_offset[scan + 1] - J.scan_hdr_offset[scan]);
    for (int e = sg.index_in_scan; e <= (sg.last_in_scan ? segs : sg.index_in_scan); e++) {
        uint32_t pos;
        if (e < segs) pos = J.d_seg_out[first + e] - data0;
        else pos = J.d_seg_out[first + segs - 1] + J.d_seg_bytes[first + segs - 1] + J.d_seg_ff[first + segs - 1] - data0;
        // payload chunks of GJ_MAX_HEADER_SIZE bytes, each preceded by marker(2)+length(2)+scan(1)
        const uint32_t byte = (uint32_t)e * 4u;
        const uint32_t chunk = byte / GJ_MAX_HEADER_SIZE, within = byte % GJ_MAX_HEADER_SIZE;
        uint8_t* p = J.d_jpeg + hdr_begin + J.scan_info_payload[scan] + chunk * (GJ_MAX_HEADER_SIZE + 5u) + within;
        p[0] = (uint8_t)(pos >> 24); p[1] = (uint8_t)(pos >> 16); p[2] = (uint8_t)(pos >> 8); p[3] = (uint8_t)pos;
    }
}

// ================================================================================================
// Launcher
// ================================================================================================
typedef void (*gj_fused_kernel_t)(const gj_geom, const uint8_t*, int16_t*, const float*, const float*);
typedef void (*gj_encode_kernel_t)(const gj_geom, const uint8_t*, const float*, const float*, const uint32_t*, uint8_t*, uint32_t*, uint32_t*, const GjTail);

// k_gather's arguments (and the encoder kernels': they use the tile list and the group totals) for a launch that leaves
// `pieces` tile streams of `spt` segments; scan s begins with stream scan_first[s]
static GjTail gj_make_tail(const gj_enc_job* job, const unsigned pieces, const unsigned (&scan_first)[GJ_MAX_COMP], const unsigned spt)
{
    const gj_geom& g = job->g;
    GjTail T;
    for (int s = 0; s < GJ_MAX_COMP; s++) {
        T.scan_first[s] = scan_first[s];
        T.hdr_end[s] = job->scan_hdr_offset[s + 1];
        const bool own_scan = !g.interleaved && s < g.comp_count; // (one scan per component, or one for all)
        T.seg_first[s] = own_scan ? (uint32_t)g.comp[s].first_segment : 0u;
        T.segs[s] = own_scan ? (uint32_t)g.comp[s].segment_count : (uint32_t)g.segment_count;
        T.block_first[s] = own_scan ? (uint32_t)(g.comp[s].data_offset / 64) : 0u;
    }
    T.spt = spt;
    T.seg_blocks = (uint32_t)g.seg_blocks;
    const unsigned ngcap = GJ_TAIL_GROUPS_CAP(g.segment_count); // (one tile stream per segment at most)
    T.group = job->d_tail + (job->tail_set & 1) * ngcap;
    T.group_other = job->d_tail + ((job->tail_set + 1) & 1) * ngcap;
    T.ngroups = (pieces + 31) / 32;
    T.piece = job->d_tail + 2 * ngcap;
    T.npieces = pieces;
    T.temp = job->d_temp;
    T.temp_blocks = (uint64_t)g.block_count;
    T.seg_bytes = job->d_seg_bytes;
    T.seg_ff = job->d_seg_ff;
    T.jpeg = job->d_jpeg;
    T.capacity = job->jpeg_capacity;
    T.scan_hdr = job->d_scan_hdr;
    T.main_hdr = job->main_hdr_size;
    T.d_result = job->d_result;
    T.h_result = job->h_result;
    const bool batch = job->batch.count > 1;
    T.f_raw = batch ? job->batch.raw : 0;
    T.f_temp = batch ? job->batch.temp : 0;
    T.f_jpeg = batch ? job->batch.jpeg : 0;
    T.f_seg = batch ? job->batch.seg : 0;
    T.f_tail = batch ? job->batch.tail : 0;
    return T;
}

// fused kernel for this configuration, or nullptr when the generic path has to be used
static gj_fused_kernel_t gj_fused_kernel(const gj_geom& g)
{
    if (g.pixel_format != GJ_PF_444_P012 || g.comp_count != 3) return nullptr;
    for (int c = 0; c < 3; c++)
        if (g.comp[c].samp_h != 1 || g.comp[c].samp_v != 1) return nullptr;
    const int from = g.color_space, to = g.color_space_internal;
    if (from == to || from == GJ_CS_NONE || to == GJ_CS_NONE) return k_fused_rgb444<GJ_CS_NONE, GJ_CS_NONE>;
    if (from == GJ_CS_RGB && to == GJ_CS_BT601_256) return k_fused_rgb444<GJ_CS_RGB, GJ_CS_BT601_256>;
    if (from == GJ_CS_RGB && to == GJ_CS_BT601) return k_fused_rgb444<GJ_CS_RGB, GJ_CS_BT601>;
    if (from == GJ_CS_RGB && to == GJ_CS_BT709) return k_fused_rgb444<GJ_CS_RGB, GJ_CS_BT709>;
    if (from == GJ_CS_BT601_256 && to == GJ_CS_RGB) return k_fused_rgb444<GJ_CS_BT601_256, GJ_CS_RGB>;
    return nullptr;
}

// fully fused kernel for this configuration, or nullptr
static gj_encode_kernel_t gj_encode_kernel(const gj_geom& g, const bool one_component = false)
{
    if (g.pixel_format != GJ_PF_444_P012 || g.comp_count != 3 || g.interleaved || g.restart_interval <= 0 || g.seg_blocks > 256 || g.seg_blocks < 256 / GJ_ENC_MAX_SPT) return nullptr;
    for (int c = 0; c < 3; c++)
        if (g.comp[c].samp_h != 1 || g.comp[c].samp_v != 1) return nullptr;
    const int from = g.color_space, to = g.color_space_internal;
    if (from == to || from == GJ_CS_NONE || to == GJ_CS_NONE) return one_component ? k_encode_rgb444<GJ_CS_NONE, GJ_CS_NONE, true> : k_encode_rgb444<GJ_CS_NONE, GJ_CS_NONE>;
    if (from == GJ_CS_RGB && to == GJ_CS_BT601_256) return one_component ? k_encode_rgb444<GJ_CS_RGB, GJ_CS_BT601_256, true> : k_encode_rgb444<GJ_CS_RGB, GJ_CS_BT601_256>;
    if (from == GJ_CS_RGB && to == GJ_CS_BT601) return one_component ? k_encode_rgb444<GJ_CS_RGB, GJ_CS_BT601, true> : k_encode_rgb444<GJ_CS_RGB, GJ_CS_BT601>;
    if (from == GJ_CS_RGB && to == GJ_CS_BT709) return one_component ? k_encode_rgb444<GJ_CS_RGB, GJ_CS_BT709, true> : k_encode_rgb444<GJ_CS_RGB, GJ_CS_BT709>;
    if (from == GJ_CS_BT601_256 && to == GJ_CS_RGB) return one_component ? k_encode_rgb444<GJ_CS_BT601_256, GJ_CS_RGB, true> : k_encode_rgb444<GJ_CS_BT601_256, GJ_CS_RGB>;
    return nullptr;
}

// k_encode_blocks for this configuration: 1 = planar input in component layout, 0 = packed 4:4:4 with a transform from RGB (or none) and
// any chroma sampling, -1 = neither (generic kernels)
static int gj_blocks_kernel_mode(const gj_geom& g)
{
    if (g.no_transform) return 1;
    if (g.pixel_format != GJ_PF_444_P012 || g.comp_count != 3) return -1;
    const int from = g.color_space, to = g.color_space_internal;
    const bool none = from == to || from == GJ_CS_NONE || to == GJ_CS_NONE;
    if (!none && !(from == GJ_CS_RGB && (to == GJ_CS_BT601 || to == GJ_CS_BT601_256 || to == GJ_CS_BT709))) return -1;
    return 0;
}

// packed 4:2:2 without colour transform in the layout k_fused_uyvy422 / k_encode_uyvy422 take
static bool gj_is_uyvy_layout(const gj_enc_job* job)
{
    const gj_geom& g = job->g;
    return job->use_fused && g.pixel_format == GJ_PF_422_P1020 && g.comp_count == 3 && g.no_transform == 0 &&
           (g.color_space == g.color_space_internal || g.color_space == GJ_CS_NONE || g.color_space_internal == GJ_CS_NONE) &&
           g.comp[0].samp_h == 2 && g.comp[0].samp_v == 1 && g.comp[1].samp_h == 1 && g.comp[1].samp_v == 1 && g.comp[2].samp_h == 1 &&
           g.comp[2].samp_v == 1 && g.comp[0].blocks_x == 2 * g.comp[1].blocks_x && g.comp[0].blocks_y == g.comp[1].blocks_y &&
           g.comp[2].blocks_x == g.comp[1].blocks_x && g.comp[2].blocks_y == g.comp[1].blocks_y;
}
// which kernel leaves the tile streams k_gather takes: 1 = k_encode_uyvy422, 2 = k_encode_blocks, 3 = k_encode_rgb444, 0 = none (coefficient planes + k_huffman)
static int gj_tile_kernel(const gj_enc_job* job)
{
    const gj_geom& g = job->g;
    const bool segs_ok = g.restart_interval > 0 && g.seg_blocks <= 256 && g.seg_blocks >= 256 / GJ_ENC_MAX_SPT;
    gj_encode_kernel_t whole = (job->use_fused && !job->keep_coefs) ? gj_encode_kernel(g) : nullptr;
    if (whole && job->tune.enc_by_blocks > 0 && gj_blocks_kernel_mode(g) == 0) whole = nullptr; // (k_encode_blocks instead)
    if (gj_is_uyvy_layout(job) && !job->keep_coefs && g.interleaved && segs_ok && g.blocks_per_mcu == 4 && g.mcu_count == g.comp[1].blocks_x * g.comp[1].blocks_y &&
        g.comp[1].type == g.comp[2].type && !job->tune.enc_no_whole422)
        return 1;
    if (!whole && job->use_fused && !job->keep_coefs && segs_ok && gj_blocks_kernel_mode(g) >= 0) return 2;
    return whole ? 3 : 0;
}

// a batch of frames takes the kernels that go from pixels to tile streams (k_encode_rgb444, k_encode_uyvy422, k_encode_blocks: every layout with restart
// segments of 4 .. 256 blocks) and k_gather, and no option that touches other buffers
extern "C" int gj_hip_encode_batchable(const gj_enc_job* job)
{
    return job->use_fused && !job->keep_coefs && !job->channel_remap && !job->flipped && !job->segment_info && gj_tile_kernel(job) != 0;
}

extern "C" int gj_hip_encode_tiles(const gj_enc_job* job) { return gj_tile_kernel(job) != 0; }

extern "C" int gj_hip_encode(const gj_enc_job* job, gj_stream_t stream, gj_event_t ev[GJ_ENC_EVENTS])
{
    hipStream_t st = (hipStream_t)stream;
    const gj_geom& g = job->g;
    if (g.blocks_per_mcu > GJ_MAX_MCU_BLOCKS) return -1;
    const unsigned frames = job->batch.count > 1 ? job->batch.count : 1u;
    if (frames > 1 && (!gj_hip_encode_batchable(job) || frames > 65535u)) return -1;
    if (ev) (void)hipEventRecord((hipEvent_t)ev[0], st);
    if (job->channel_remap) { // the reference permutes the channels of the raw image in place first (src/gpujpeg_preprocessor.cu:570-575)
        const unsigned n = (unsigned)g.width * (unsigned)g.height;
        hipLaunchKernelGGL(k_channel_remap, dim3((n + 255) / 256), dim3(256), 0, st, g, const_cast<uint8_t*>(job->d_raw), job->channel_remap & 0xFFFFu);
    }
    bool tiles = true; // k_encode_*: tile streams for k_gather
    GjTail T;
    T.npieces = 0;
    const int tile_kernel = gj_tile_kernel(job);
    gj_encode_kernel_t whole = tile_kernel == 3 ? gj_encode_kernel(g) : nullptr;
    gj_fused_kernel_t fused = job->use_fused ? gj_fused_kernel(g) : nullptr;
    const bool uyvy = gj_is_uyvy_layout(job);
    if (tile_kernel == 1) {
        if (ev) (void)hipEventRecord((hipEvent_t)ev[1], st);
        if (ev) (void)hipEventRecord((hipEvent_t)ev[2], st);
        const int spt = 256 / g.seg_blocks;
        const unsigned wgs = ((unsigned)g.segment_count + spt - 1) / spt;
        T = gj_make_tail(job, wgs, {0u, ~0u, ~0u, ~0u}, (unsigned)spt);
        hipLaunchKernelGGL(k_encode_uyvy422, dim3(wgs, 1, frames), dim3(256), 0, st, g, job->d_raw, job->d_fwd_q[0], job->d_fwd_q[1], job->d_huff_lut,
                           job->d_temp, job->d_seg_bytes, job->d_seg_ff, T);
    } else if (tile_kernel == 2) {
        // every other layout with short restart segments: one lane per block in coding order (k_encode_blocks)
        if (ev) (void)hipEventRecord((hipEvent_t)ev[1], st);
        if (ev) (void)hipEventRecord((hipEvent_t)ev[2], st);
        const int spt = 256 / g.seg_blocks;
        unsigned wgs = 0, scan_first[GJ_MAX_COMP] = {0u, ~0u, ~0u, ~0u};
        if (g.interleaved) wgs = ((unsigned)g.segment_count + spt - 1) / spt;
        else
            for (int c = 0; c < g.comp_count; c++) {
                scan_first[c] = wgs;
                wgs += ((unsigned)g.comp[c].segment_count + spt - 1) / spt;
            }
        T = gj_make_tail(job, wgs, scan_first, (unsigned)spt);
        hipLaunchKernelGGL(gj_blocks_kernel_mode(g) ? k_encode_blocks<true> : k_encode_blocks<false>, dim3(wgs, 1, frames), dim3(256), 0, st, g, job->d_raw, job->d_fwd_q[0],
                           job->d_fwd_q[1], job->d_huff_lut, job->d_temp, job->d_seg_bytes, job->d_seg_ff, T);
    } else if (whole) { // pixels -> segment streams in one kernel, no coefficient planes
        if (ev) (void)hipEventRecord((hipEvent_t)ev[1], st);
        if (ev) (void)hipEventRecord((hipEvent_t)ev[2], st);
        const int spt = 256 / g.seg_blocks;
        const unsigned wgs = ((unsigned)g.comp[0].segment_count + spt - 1) / spt;
        T = gj_make_tail(job, 3 * wgs, {0u, wgs, 2 * wgs, ~0u}, (unsigned)spt);
        // (a component per workgroup while three times the tiles still fit the places the GPU has: GJ_ENC_SPLIT=<tiles> moves the limit, 0 = never)
        const unsigned split_up_to = job->tune.enc_split >= 0 ? (unsigned)job->tune.enc_split : (unsigned)gj_hip_cu_count() * 4u / 3u; // (MI355X: 341)
        const bool split = wgs * frames <= split_up_to;
        if (split) whole = gj_encode_kernel(g, true);
        // (a frame with more tiles than the GPU has places: its last tiles as three workgroups each, see the kernel; GJ_ENC_TAIL=<tiles>, 0 = none)
        unsigned grid_x = wgs;
        const unsigned tail = job->tune.enc_tail >= 0 ? (unsigned)job->tune.enc_tail : 32u;
        T.tail_from = 0xFFFFFFFFu;
        T.tiles = wgs;
        // (an explicit GJ_ENC_TAIL applies to frames of any size: the tests reach the mixture with small frames that way)
        if (!split && frames == 1 && tail > 0 && wgs > tail && (job->tune.enc_tail > 0 || wgs > (unsigned)gj_hip_cu_count() * 4u)) {
            T.tail_from = wgs - tail;
            grid_x = wgs + 2u * tail;
        }
        hipLaunchKernelGGL(whole, dim3(grid_x, split ? 3 : 1, frames), dim3(256), 0, st, g, job->d_raw, job->d_fwd_q[0], job->d_fwd_q[1], job->d_huff_lut, job->d_temp,
                           job->d_seg_bytes, job->d_seg_ff, T);
    } else {
    tiles = false;
    if (uyvy) { // packed 4:2:2 without colour transform: pixels -> coefficients, one thread per MCU
        if (ev) (void)hipEventRecord((hipEvent_t)ev[1], st);
        const unsigned nm = (unsigned)(g.comp[1].blocks_x * g.comp[1].blocks_y);
        hipLaunchKernelGGL(k_fused_uyvy422, dim3((nm + 255) / 256), dim3(256), 0, st, g, job->d_raw, job->d_coefs, job->d_fwd_q[0], job->d_fwd_q[1]);
    } else if (fused) {
        if (ev) (void)hipEventRecord((hipEvent_t)ev[1], st);
        const unsigned nb = (unsigned)(g.comp[0].blocks_x * g.comp[0].blocks_y);
        hipLaunchKernelGGL(fused, dim3((nb + 255) / 256), dim3(256), 0, st, g, job->d_raw, job->d_coefs, job->d_fwd_q[0],
                           job->d_fwd_q[1]);
    } else {
        if (g.no_transform) {
            hipLaunchKernelGGL(k_copy_planes_in, dim3(2048), dim3(256), 0, st, g, job->d_raw, job->d_planes);
        } else {
            const unsigned n = (unsigned)g.raw_width * (unsigned)g.height;
            hipLaunchKernelGGL(k_preprocess, dim3((n + 255) / 256), dim3(256), 0, st, g, job->d_raw, job->d_planes);
        }
        if (job->flipped) hipLaunchKernelGGL(k_flip_planes, dim3(1024), dim3(256), 0, st, g, job->d_planes);
        if (ev) (void)hipEventRecord((hipEvent_t)ev[1], st);
        hipLaunchKernelGGL(k_dct, dim3(((unsigned)g.block_count + 255) / 256), dim3(256), 0, st, g, job->d_planes, job->d_coefs,
                           job->d_fwd_q[0], job->d_fwd_q[1]);
    }
    if (ev) (void)hipEventRecord((hipEvent_t)ev[2], st);
    const int B = g.seg_blocks;
    const int spt = B <= 256 ? 256 / B : 1;
    const unsigned tiles = ((unsigned)g.segment_count + spt - 1) / spt;
    hipLaunchKernelGGL(k_huffman, dim3(tiles), dim3(256), 0, st, g, job->d_coefs, job->d_huff_lut, job->d_temp, job->d_seg_bytes,
                       job->d_seg_ff);
    }
    if (ev) (void)hipEventRecord((hipEvent_t)ev[3], st);
    if (tiles) // the tile streams -> the file: one wave each
        hipLaunchKernelGGL(k_gather, dim3((T.npieces + 3) / 4, 1, frames), dim3(256), 0, st, T);
    const bool seg_info = job->segment_info && g.restart_interval > 0;
    // (behind k_gather the segment offsets are needed for the APP13 index only)
    const unsigned scan_wgs = ((unsigned)g.segment_count + 1023) / 1024;
    if (!tiles || seg_info)
        hipLaunchKernelGGL(k_scan_segments, dim3(scan_wgs), dim3(1024), 0, st, *job, (unsigned long long*)job->d_scan_partial, job->epoch);
    if (ev) (void)hipEventRecord((hipEvent_t)ev[4], st);
    if (!tiles)
        hipLaunchKernelGGL(k_assemble, dim3(((unsigned)g.segment_count + 4 * GJ_ASM_SEGS - 1) / (4 * GJ_ASM_SEGS)), dim3(256), 0, st, *job);
    if (seg_info)
        hipLaunchKernelGGL(k_segment_info, dim3(((unsigned)g.segment_count + 255) / 256), dim3(256), 0, st, *job);
    if (ev) (void)hipEventRecord((hipEvent_t)ev[5], st);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
