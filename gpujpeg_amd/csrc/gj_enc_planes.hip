// gj_enc_planes.hip -- MI355X (gfx950, wave64) JPEG encoder: the paths through padded component planes and coefficient planes
// (part of the encoder's device code, see gj_enc_internal.h for the map of the files)
//   k_preprocess / k_copy_planes_in + k_dct   generic path (every pixel format / subsampling)
//   k_fused_rgb444 / k_fused_uyvy422          raw packed pixels -> quantised coefficients (preprocess + DCT + quant fused)
//   k_huffman                                 one LANE per 8x8 block: sparse run-length + Huffman coding, bits OR-ed into an LDS stream
// Taken when the caller wants the coefficients (gpujpeg_amd_encoder_keep_coefficients), for restart intervals 0 or > 256 blocks, flipped images and the
// pixel formats the fused encoders of gj_enc_tiles.hip do not cover.
#include "gj_enc_internal.h"

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


// fused kernel for this configuration, or nullptr when the generic path has to be used
gj_fused_kernel_t gj_fused_kernel(const gj_geom& g)
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

gj_fused_kernel_t gj_fused_uyvy422_kernel() { return k_fused_uyvy422; }
