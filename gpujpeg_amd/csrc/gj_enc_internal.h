// gj_enc_internal.h -- MI355X (gfx950, wave64) JPEG encoder: what the encoder's device translation units share.
//
// Map of the encoder's device code (round 6: gj_encode.hip was one 2 000-line file; split by kernel family like the decoder's gj_dec_*.hip):
//   gj_encode.hip        gj_hip_encode: picks the kernels of a frame and launches them (the only file the host's C-ABI reaches)
//   gj_enc_tiles.hip     the fully fused encoders -- pixels -> entropy-coded tile streams, no coefficient planes: k_encode_rgb444 (BASELINE
//                        configs 1-3, 5), k_encode_uyvy422 (config 4), k_encode_blocks (every other layout with short restart segments) -- and the
//                        lane-per-block coder they share (gj_code_tile)
//   gj_enc_assemble.hip  tile streams / segment streams -> the file: k_gather (behind the fused encoders); k_scan_segments + k_assemble (behind
//                        k_huffman), k_segment_info (APP13 index)
//   gj_enc_planes.hip    the paths through coefficient planes: k_preprocess / k_copy_planes_in, k_dct, k_fused_rgb444, k_fused_uyvy422, k_huffman
// The reference runs preprocess -> (DCT per component) -> codeword kernel -> serialisation kernel -> compaction
// kernel and stitches segments on the host (src/gpujpeg_encoder.c:485-629); the arithmetic restates
// src/gpujpeg_preprocessor.cu, src/gpujpeg_colorspace.h, src/gpujpeg_dct_gpu.cu and src/gpujpeg_huffman_gpu_encoder.cu.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "gj_device.h"
#include "gj_hip.h"

#define GJ_ENC_MAX_SPT 64 // segments per tile the bookkeeping holds (restart intervals of >= 4 blocks)
#define GJ_ASM_SEGS 4 // segments per wave: their sizes, offsets and first 256 bytes are requested together (one memory round trip

// ---- pixels of a packed 4:4:4 block position (k_fused_rgb444, k_encode_rgb444)
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


// ---- what the fused encoders leave for k_gather (gj_enc_assemble.hip has the description)
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


// ---- kernels and their choice by configuration (defined next to the kernels; nullptr = this configuration does not take that kernel)
typedef void (*gj_fused_kernel_t)(const gj_geom, const uint8_t*, int16_t*, const float*, const float*);
typedef void (*gj_encode_kernel_t)(const gj_geom, const uint8_t*, const float*, const float*, const uint32_t*, uint8_t*, uint32_t*, uint32_t*, const GjTail);
gj_fused_kernel_t gj_fused_kernel(const gj_geom& g);                                   // k_fused_rgb444<from, to>
gj_fused_kernel_t gj_fused_uyvy422_kernel();                                           // k_fused_uyvy422
gj_encode_kernel_t gj_encode_kernel(const gj_geom& g, const bool one_component = false); // k_encode_rgb444<from, to, one component per workgroup>
gj_encode_kernel_t gj_encode_uyvy422_kernel();                                         // k_encode_uyvy422
gj_encode_kernel_t gj_encode_blocks_kernel(const bool planar);                         // k_encode_blocks<planar>
__global__ void k_preprocess(const gj_geom g, const uint8_t* __restrict__ raw, uint8_t* __restrict__ planes);
__global__ void k_copy_planes_in(const gj_geom g, const uint8_t* __restrict__ raw, uint8_t* __restrict__ planes);
__global__ void k_dct(const gj_geom g, const uint8_t* __restrict__ planes, int16_t* __restrict__ coefs, const float* __restrict__ q_luma, const float* __restrict__ q_chroma);
__global__ void k_huffman(const gj_geom g, const int16_t* __restrict__ coefs, const uint32_t* __restrict__ lut, uint8_t* __restrict__ temp, uint32_t* __restrict__ seg_bytes,
                          uint32_t* __restrict__ seg_ff);
__global__ void k_gather(const GjTail T0);
__global__ void k_scan_segments(const gj_enc_job J, unsigned long long* __restrict__ partial, const uint32_t epoch);
__global__ void k_assemble(const gj_enc_job J);
__global__ void k_segment_info(const gj_enc_job J);
