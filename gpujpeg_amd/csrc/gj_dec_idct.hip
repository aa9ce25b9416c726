// gj_dec_idct.hip -- MI355X (gfx950, wave64) JPEG decoder: dequantisation + IDCT (+ colour transform + packed store), from the coefficient planes or from tokens; postprocessor
// Restates src/gpujpeg_dct_gpu.cu:312-366,472-618 and src/gpujpeg_postprocessor.cu:49-217.
// (part of the decoder's device code, see gj_dec_internal.h for the map of the files)
#include "gj_dec_internal.h"

// ================================================================================================
// Dequantisation + IDCT, one thread per block
// ================================================================================================
// `zero`: every block is overwritten with zeros once it has been read, which leaves the coefficient planes ready for the
// entropy decoder of the next frame (it stores non-zero coefficients only) without a separate 2 B/sample memset.
__global__ __launch_bounds__(256) void k_idct(const gj_geom g, int16_t* __restrict__ coefs, const float* __restrict__ qtab,
                                              uint8_t* __restrict__ planes, const int zero)
{
    if (g.fb.sizes != nullptr) { // frame blockIdx.z of a batch (the planes of a frame take as many bytes as its coefficients take elements)
        coefs += (size_t)blockIdx.z * g.fb.coefs;
        planes += (size_t)blockIdx.z * g.fb.coefs;
    }
    const unsigned gb = blockIdx.x * 256u + threadIdx.x;
    if (gb >= (unsigned)g.block_count) return;
    int c = 0;
#pragma unroll
    for (int i = 1; i < GJ_MAX_COMP; i++)
        if (i < g.comp_count && (uint64_t)gb * 64 >= g.comp[i].data_offset) c = i;
    const gj_comp_geom& k = g.comp[c];
    const unsigned lb = gb - (unsigned)(k.data_offset / 64);
    const unsigned by = lb / (unsigned)k.blocks_x, bx = lb - by * (unsigned)k.blocks_x;
    uint32_t w[32];
    {
        uint4* p = reinterpret_cast<uint4*>(coefs + (size_t)gb * 64);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint4 v = p[r];
            if (zero) p[r] = make_uint4(0, 0, 0, 0);
            w[r * 4] = v.x; w[r * 4 + 1] = v.y; w[r * 4 + 2] = v.z; w[r * 4 + 3] = v.w;
        }
    }
    uint32_t px[16];
    gj_idct_pk(w, qtab + k.q_table * 64, px);
    uint8_t* dst = planes + k.data_offset + (size_t)by * 8 * k.data_width + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; r++) *reinterpret_cast<uint2*>(dst + (size_t)r * k.data_width) = make_uint2(px[2 * r], px[2 * r + 1]);
}

// ================================================================================================
// Fused IDCT + colour transform + packed 4:4:4 store (3 B/pixel); one thread per block position.
// ================================================================================================
template <int CS_FROM, int CS_TO>
__device__ __forceinline__ void gj_color_static_d(int& a, int& b, int& c)
{
    if (CS_FROM == CS_TO || CS_FROM == GJ_CS_NONE || CS_TO == GJ_CS_NONE) return;
    if (CS_FROM == GJ_CS_RGB) gj_rgb_to(CS_TO, a, b, c);
    else if (CS_TO == GJ_CS_RGB) gj_to_rgb(CS_FROM, a, b, c);
}

// byte X (compile-time) of an 8-sample row held in two dwords, as float
template <int X>
__device__ __forceinline__ float gj_sample_f(const uint32_t (&c)[2])
{
    return gj_ubyte_f<X & 3>(c[X >> 2]);
}

// pixels X and X + 1 of a row: component samples -> colour transform -> bytes 3X .. 3X + 5 of the packed output row
template <int CS_FROM, int CS_TO, int X>
__device__ __forceinline__ void gj_store_pair(const uint32_t (&c0)[2], const uint32_t (&c1)[2], const uint32_t (&c2)[2], uint32_t (&px)[6])
{
    gj_f2 a = gj_f2{gj_sample_f<X>(c0), gj_sample_f<X + 1>(c0)};
    gj_f2 b = gj_f2{gj_sample_f<X>(c1), gj_sample_f<X + 1>(c1)};
    gj_f2 c = gj_f2{gj_sample_f<X>(c2), gj_sample_f<X + 1>(c2)};
    gj_color_f<CS_FROM, CS_TO>(a, b, c);
    constexpr int B = 3 * X;
    px[(B + 0) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(a.x, (B + 0) & 3, px[(B + 0) >> 2]);
    px[(B + 1) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(b.x, (B + 1) & 3, px[(B + 1) >> 2]);
    px[(B + 2) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(c.x, (B + 2) & 3, px[(B + 2) >> 2]);
    px[(B + 3) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(a.y, (B + 3) & 3, px[(B + 3) >> 2]);
    px[(B + 4) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(b.y, (B + 4) & 3, px[(B + 4) >> 2]);
    px[(B + 5) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(c.y, (B + 5) & 3, px[(B + 5) >> 2]);
}

// Coefficients travel HBM -> LDS in fully coalesced 16 B chunks (a thread-per-block read would touch 64 different 128 B
// lines per load instruction); each thread then takes its own block out of LDS. Blocks are padded to 144 B there, which
// makes both the linear writes and the per-block 16 B reads bank-conflict free (36 dwords: 9 x 4, 9 coprime to 16).
#define GJ_TILE_PITCH 144
// colour transform + packed 4:4:4 store of one block position (three byte-packed component blocks, 8 rows of 24 bytes)
template <int CS_FROM, int CS_TO>
__device__ __forceinline__ void gj_store_rgb444(const gj_geom& g, uint8_t* __restrict__ raw, const uint32_t (&pk)[3][16], const unsigned lb,
                                                const unsigned nb, const unsigned bx, const unsigned by)
{
    const size_t pitch = (size_t)g.width * 3 + g.width_padding;
    const bool interior = lb < nb && (bx * 8 + 8 <= (unsigned)g.width) && (by * 8 + 8 <= (unsigned)g.height);
    const bool aligned = ((pitch | (size_t)raw) & 3) == 0;
    // (the rows' addresses by addition: written as (by * 8 + r) * pitch the compiler multiplies 64-bit numbers for every row, round 5)
    uint8_t* row = raw + (size_t)(by * 8) * pitch + (size_t)bx * 24;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        // colour transform in fp32 on pixel pairs (gj_color_f, exact; see gj_device.h), results packed straight into the 24 output bytes
        uint32_t px[6] = {0, 0, 0, 0, 0, 0};
        const uint32_t c0[2] = {pk[0][2 * r], pk[0][2 * r + 1]}, c1[2] = {pk[1][2 * r], pk[1][2 * r + 1]}, c2[2] = {pk[2][2 * r], pk[2][2 * r + 1]};
        gj_store_pair<CS_FROM, CS_TO, 0>(c0, c1, c2, px);
        gj_store_pair<CS_FROM, CS_TO, 2>(c0, c1, c2, px);
        gj_store_pair<CS_FROM, CS_TO, 4>(c0, c1, c2, px);
        gj_store_pair<CS_FROM, CS_TO, 6>(c0, c1, c2, px);
        const unsigned y = by * 8 + r;
        if (interior && aligned) {
            uint2* p = reinterpret_cast<uint2*>(row);
            p[0] = make_uint2(px[0], px[1]);
            p[1] = make_uint2(px[2], px[3]);
            p[2] = make_uint2(px[4], px[5]);
            row += pitch;
        } else if (lb < nb && y < (unsigned)g.height) {
#pragma unroll
            for (int byte = 0; byte < 24; byte++) {
                const unsigned x = bx * 8 + byte / 3;
                if (x < (unsigned)g.width) raw[(size_t)y * pitch + (size_t)x * 3 + byte % 3] = (uint8_t)(px[byte >> 2] >> ((byte & 3) * 8));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int CS_FROM, int CS_TO>
__global__ __launch_bounds__(256, 3) void k_idct_fused_rgb444(const gj_geom g, int16_t* __restrict__ coefs,
                                                              const float* __restrict__ qtab, uint8_t* __restrict__ raw, const int zero)
{
    if (g.fb.sizes != nullptr) { // frame blockIdx.z of a batch
        coefs += (size_t)blockIdx.z * g.fb.coefs;
        raw += (size_t)blockIdx.z * g.fb.raw;
    }
    __shared__ __attribute__((aligned(16))) uint8_t s_blk[256 * GJ_TILE_PITCH];
    __shared__ __attribute__((aligned(8))) float s_q[3][64]; // dequantisation tables: read as VGPR pairs for v_pk_mul_f32
    if (threadIdx.x < 192) s_q[threadIdx.x >> 6][threadIdx.x & 63] = qtab[g.comp[threadIdx.x >> 6].q_table * 64 + (threadIdx.x & 63)];
    const gj_comp_geom& k0 = g.comp[0];
    const unsigned nb = (unsigned)(k0.blocks_x * k0.blocks_y);
    const unsigned lb0 = blockIdx.x * 256u;
    const unsigned lb = lb0 + threadIdx.x;
    const unsigned nchunk = min(256u, nb - lb0) * 8u; // 16 B chunks of this tile
    const unsigned by = lb / (unsigned)k0.blocks_x, bx = lb - by * (unsigned)k0.blocks_x;
    uint32_t pk[3][16];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        uint4* src = reinterpret_cast<uint4*>(coefs + g.comp[c].data_offset + (size_t)lb0 * 64);
        uint4 w[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const unsigned ch = i * 256u + threadIdx.x;
            w[i] = ch < nchunk ? src[ch] : make_uint4(0, 0, 0, 0);
        }
        if (zero) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const unsigned ch = i * 256u + threadIdx.x;
                if (ch < nchunk) src[ch] = make_uint4(0, 0, 0, 0);
            }
        }
        if (c) __syncthreads(); // everybody has taken the previous component's block out of LDS
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const unsigned ch = i * 256u + threadIdx.x;
            *reinterpret_cast<uint4*>(s_blk + (ch >> 3) * GJ_TILE_PITCH + (ch & 7u) * 16u) = w[i];
        }
        __syncthreads();
        uint32_t wb[32];
        {
            const uint4* p = reinterpret_cast<const uint4*>(s_blk + threadIdx.x * GJ_TILE_PITCH);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint4 v = p[r];
                wb[r * 4] = v.x; wb[r * 4 + 1] = v.y; wb[r * 4 + 2] = v.z; wb[r * 4 + 3] = v.w;
            }
        }
        gj_idct_pk(wb, s_q[c], pk[c]);
        // pin the transform here: otherwise LLVM sinks all three below the last barrier and spills the staged coefficients
#pragma unroll
        for (int i = 0; i < 16; i++) GJ_KEEP(pk[c][i]);
    }
    // (no early return for the threads past the last block: the compiler would sink the three transforms below it and keep
    // every staged coefficient alive until then)
    gj_store_rgb444<CS_FROM, CS_TO>(g, raw, pk, lb, nb, bx, by);
}

// ================================================================================================
// The same, fed by the entropy decoder's TOKENS (DESIGN 4.3 "token mode"): per block a record (first token, count, DC
// term) in coding order and, in one dense array, the non-zero AC coefficients as 16-bit tokens value << 6 | natural position.
// A block costs 8 B + 2 B per non-zero coefficient of HBM traffic instead of 128 B written (twice) and read. Every lane
// clears its own 128-byte slot of the LDS tile, the wave copies the token range of its 64 blocks into LDS with 16-byte
// loads (consecutive blocks of a scan have consecutive tokens; a new range starts where a decoder batch ended), every lane
// scatters its own tokens into its slot (2-byte LDS stores) and reads the block back as rows. Nothing crosses waves, so
// there is no workgroup barrier. Blocks of segments too long for the decoder's LDS stage arrive through the coefficient
// planes as before (count 0xFFFF in the record). Non-interleaved scans only (plane order == coding order).
// ================================================================================================
#define GJ_TOK_STAGE 832 // tokens per wave in LDS (with the 32 KiB tile: four workgroups per CU)

// a lane's 128-byte slot of the block tile: row r (16 bytes) sits at (r ^ (lane & 7)) * 16, which spreads the row reads and
// writes of the 64 lanes over all banks without padding the slot. A token carries its natural position = row << 3 | column in its low
// 6 bits and the value above them: its place in the slot is 2 x (position XOR (lane & 7) << 3), and what is stored there is the token
// with the position masked off = 64 x the value -- the dequantisation table of the token-fed kernels holds q / 64 for the AC positions
// (a power of two: the product is the same fp32 number), so no shift is spent on the value.
__device__ __forceinline__ uint4* gj_slot_row(uint8_t* slot, const int lane, const int r)
{
    return reinterpret_cast<uint4*>(slot + ((uint32_t)(r << 4) ^ (((uint32_t)lane & 7u) << 4)));
}

__device__ __forceinline__ void gj_slot_put(uint8_t* slot, const uint32_t swz /* (lane & 7) << 3 */, const uint32_t tok)
{
    reinterpret_cast<uint16_t*>(slot)[(tok & 63u) ^ swz] = (uint16_t)(tok & 0xFFC0u);
}

// the wave's token range of one component: dense and small enough for the stage (the normal case), with the two 16-byte
// loads per lane that fetch it
struct GjTokRange {
    uint32_t S, E;
    bool fast;
    uint4 t0, t1;
};

__device__ __forceinline__ GjTokRange gj_tok_fetch(const uint16_t* __restrict__ d_tok, const uint32_t start, const uint32_t cnt, const int lane)
{
    GjTokRange r;
    const uint32_t end = start + cnt;
    const uint32_t prev_end = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)end, 0x138, 0xF, 0xF, false); // wave_shr:1
    const unsigned long long breaks = __ballot(lane != 0 && start != prev_end);
    r.S = (uint32_t)__builtin_amdgcn_readlane((int)start, 0) & ~7u; // (16-byte pieces of 8 tokens)
    r.E = (uint32_t)__builtin_amdgcn_readlane((int)end, 63);
    r.fast = breaks == 0 && r.E - r.S <= GJ_TOK_STAGE;
    r.t0 = r.t1 = make_uint4(0, 0, 0, 0);
    if (r.fast) {
        const uint32_t i0 = (uint32_t)lane * 8u, i1 = i0 + 512u;
        if (r.S + i0 < r.E) r.t0 = *reinterpret_cast<const uint4*>(d_tok + r.S + i0);
        if (i1 < GJ_TOK_STAGE && r.S + i1 < r.E) r.t1 = *reinterpret_cast<const uint4*>(d_tok + r.S + i1);
    }
    return r;
}

// One block per lane: zeros, the DC term and the lane's tokens go into its tile slot. `fast`: the wave's tokens are in the stage
// already (dense range starting at token S).
// MULTI: several ranges are staged together when they fit (the kernels fed by the lane-per-segment decoders, where that is the normal case)
template <bool MULTI>
__device__ __forceinline__ void gj_tok_to_slot(uint8_t* slot, uint16_t* stage, const int lane, const bool fast, const uint32_t S, const uint32_t start,
                                               const uint32_t cnt, const uint32_t dc, const bool in_plane, const uint4* __restrict__ plane_block,
                                               const uint16_t* __restrict__ d_tok)
{
#pragma unroll
    for (int r = 0; r < 8; r++) *gj_slot_row(slot, lane, r) = make_uint4(0, 0, 0, 0);
    if (in_plane) { // block of a segment that was decoded piece by piece: it is in the coefficient plane
#pragma unroll
        for (int r = 0; r < 8; r++) *gj_slot_row(slot, lane, r) = plane_block[r];
    }
    const uint32_t end = start + cnt;
    const uint32_t swz = ((uint32_t)lane & 7u) << 3;
    if (fast) {
        gj_wave_sync();
        uint32_t a = start - S;
        const uint32_t b = end - S;
        for (; a + 2 <= b; a += 2) {
            const uint32_t ta = stage[a], tb = stage[a + 1];
            gj_slot_put(slot, swz, ta);
            gj_slot_put(slot, swz, tb);
        }
        if (a < b) gj_slot_put(slot, swz, stage[a]);
    } else {
        // several ranges (a decoder batch ended inside the wave's blocks; the lane-per-segment decoders start a range per restart segment:
        // two or three in every wave of config 4) or more tokens than the stage holds
        const uint32_t prev_end = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)end, 0x138, 0xF, 0xF, false);
        const unsigned long long runs0 = __ballot(lane == 0 || start != prev_end);
        // all ranges behind each other in the stage, each from its 16-byte piece on: their loads are in flight together
        uint32_t total = 0, my_at = 0;
        for (unsigned long long runs = MULTI ? runs0 : 0ull; runs;) {
            const int d = __builtin_ctzll(runs);
            runs &= runs - 1;
            const int dn = runs ? __builtin_ctzll(runs) : 64;
            const uint32_t RS = (uint32_t)__builtin_amdgcn_readlane((int)start, d), RE = (uint32_t)__builtin_amdgcn_readlane((int)end, dn - 1);
            if (lane >= d && lane < dn) my_at = total + (start - (RS & ~7u));
            total += RE > (RS & ~7u) ? (RE - (RS & ~7u) + 7u) & ~7u : 0u;
            if (RE < RS) total = GJ_TOK_STAGE + 1u; // (records of a damaged stream: the careful way below)
        }
        if (MULTI && total <= GJ_TOK_STAGE) {
            gj_wave_sync();
            uint32_t at = 0;
            for (unsigned long long runs = runs0; runs;) {
                const int d = __builtin_ctzll(runs);
                runs &= runs - 1;
                const int dn = runs ? __builtin_ctzll(runs) : 64;
                const uint32_t RS = (uint32_t)__builtin_amdgcn_readlane((int)start, d) & ~7u, RE = (uint32_t)__builtin_amdgcn_readlane((int)end, dn - 1);
                for (uint32_t i = (uint32_t)lane * 8u; RS + i < RE; i += 512u)
                    *reinterpret_cast<uint4*>(stage + at + i) = *reinterpret_cast<const uint4*>(d_tok + RS + i);
                at += RE > RS ? (RE - RS + 7u) & ~7u : 0u;
            }
            gj_wave_sync();
            for (uint32_t a = 0; a < cnt; a++) gj_slot_put(slot, swz, stage[my_at + a]);
        } else {
            // range by range, chunk by chunk
            unsigned long long runs = runs0;
            while (runs) {
                const int d = __builtin_ctzll(runs);
                runs &= runs - 1;
                const int dn = runs ? __builtin_ctzll(runs) : 64;
                const uint32_t RS = (uint32_t)__builtin_amdgcn_readlane((int)start, d), RE = (uint32_t)__builtin_amdgcn_readlane((int)end, dn - 1);
                const bool mine = lane >= d && lane < dn;
                for (uint32_t base = RS & ~7u; base < RE; base += GJ_TOK_STAGE) {
                    gj_wave_sync();
                    for (uint32_t i = (uint32_t)lane * 8u; i < GJ_TOK_STAGE && base + i < RE; i += 512u)
                        *reinterpret_cast<uint4*>(stage + i) = *reinterpret_cast<const uint4*>(d_tok + base + i);
                    gj_wave_sync();
                    if (mine) {
                        const uint32_t b = min(end, base + GJ_TOK_STAGE);
                        for (uint32_t a = max(start, base); a < b; a++) gj_slot_put(slot, swz, stage[a - base]);
                    }
                }
            }
        }
    }
    // the DC term last: a token of a damaged stream that ran past its block's end sits on position 0 (the entropy decoders' zig-zag
    // tables say so) and disappears here, like in the plane kernels
    if (!in_plane) *reinterpret_cast<uint16_t*>(slot + ((lane & 7) << 4)) = (uint16_t)dc;
    gj_wave_sync(); // (the stage is rewritten by the next component)
}

template <int CS_FROM, int CS_TO>
__global__ __launch_bounds__(256, 4) void k_idct_tok_rgb444(const gj_geom g, const int16_t* __restrict__ coefs, const uint2* __restrict__ d_rec,
                                                            const uint16_t* __restrict__ d_tok, const uint32_t tok_cap,
                                                            const float* __restrict__ qtab, uint8_t* __restrict__ raw)
{
    if (g.fb.sizes != nullptr) { // frame blockIdx.z of a batch
        const size_t z = blockIdx.z;
        coefs += z * g.fb.coefs; d_rec += z * g.fb.rec; d_tok += z * g.fb.tok;
        raw += z * g.fb.raw;
    }
    __shared__ __attribute__((aligned(16))) uint8_t s_blk[256 * 128];
    __shared__ __attribute__((aligned(16))) uint16_t s_stage[4][GJ_TOK_STAGE];
    // dequantisation tables: [0] for blocks rebuilt from tokens (AC entries / 64: the slot holds 64 x the value, see gj_slot_put; the DC term is
    // stored as it is), [1] for blocks that arrive through the coefficient planes
    __shared__ __attribute__((aligned(8))) float s_q[2][3][64];
    if (threadIdx.x < 192) {
        const float q = qtab[g.comp[threadIdx.x >> 6].q_table * 64 + (threadIdx.x & 63)];
        s_q[0][threadIdx.x >> 6][threadIdx.x & 63] = (threadIdx.x & 63) ? q * 0.015625f : q;
        s_q[1][threadIdx.x >> 6][threadIdx.x & 63] = q;
    }
    const gj_comp_geom& k0 = g.comp[0];
    const unsigned nb = (unsigned)(k0.blocks_x * k0.blocks_y);
    const unsigned lb = blockIdx.x * 256u + threadIdx.x;
    const unsigned by = lb / (unsigned)k0.blocks_x, bx = lb - by * (unsigned)k0.blocks_x;
    const int lane = threadIdx.x & 63;
    uint8_t* slot = s_blk + threadIdx.x * 128;
    uint16_t* stage = s_stage[threadIdx.x >> 6];

    // ---- 1. the three block records (independent loads)
    // (count and DC term stay packed as the record holds them -- count << 16 | DC, bit 31: "the block is in the coefficient planes" -- until their
    // component is worked on: the kernel has no register to spare, profiles/r5_11)
    uint32_t start[3], cd[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        start[c] = cd[c] = 0;
        if (lb < nb) {
            const uint2 r = d_rec[g.comp[c].data_offset / 64 + lb];
            start[c] = r.x;
            uint32_t n = r.y >> 16;
            const bool planes = n == 0xFFFFu;
            if (planes || n > 63u || start[c] > tok_cap || n > tok_cap - start[c]) n = 0; // (the second: a record nobody wrote, damaged stream)
            cd[c] = (r.y & 0xFFFFu) | (n << 16) | (planes ? 0x80000000u : 0u);
        }
    }
    auto cnt_of = [](const uint32_t x) { return (x >> 16) & 0x7FFFu; };
    __syncthreads(); // (s_q; everything below is private to a wave)

    // ---- 2. per component: the tokens of the wave's 64 blocks go through the LDS stage (consecutive blocks of a scan have
    //         consecutive tokens); the loads of the next component are in flight while this one is transformed
    uint32_t pk[3][16];
    GjTokRange cur = gj_tok_fetch(d_tok, start[0], cnt_of(cd[0]), lane);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const bool fast = cur.fast;
        const uint32_t S = cur.S;
        if (fast) {
            *reinterpret_cast<uint4*>(stage + lane * 8) = cur.t0;
            if (lane * 8 + 512 < GJ_TOK_STAGE) *reinterpret_cast<uint4*>(stage + lane * 8 + 512) = cur.t1;
        }
        if (c < 2) cur = gj_tok_fetch(d_tok, start[c + 1], cnt_of(cd[c + 1]), lane);
        const bool in_plane = (int32_t)cd[c] < 0;
        gj_tok_to_slot<false>(slot, stage, lane, fast, S, start[c], cnt_of(cd[c]), cd[c] & 0xFFFFu, in_plane,
                       reinterpret_cast<const uint4*>(coefs + g.comp[c].data_offset + (size_t)lb * 64), d_tok);
        // the block as rows; dequantisation + IDCT
        uint32_t wb[32];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint4 v = *gj_slot_row(slot, lane, r);
            wb[r * 4] = v.x; wb[r * 4 + 1] = v.y; wb[r * 4 + 2] = v.z; wb[r * 4 + 3] = v.w;
        }
        gj_idct_pk(wb, s_q[in_plane ? 1 : 0][c], pk[c]);
#pragma unroll
        for (int i = 0; i < 16; i++) GJ_KEEP(pk[c][i]); // one transform at a time (see k_idct_fused_rgb444)
    }
    gj_store_rgb444<CS_FROM, CS_TO>(g, raw, pk, lb, nb, bx, by);
}

// ================================================================================================
// Token-fed IDCT for interleaved 4:2:2 scans with packed UYVY output and no colour transform (BASELINE config 4): one lane
// per BLOCK in coding order (Y0 Y1 Cb Cr of MCU 0, of MCU 1, ...), so a workgroup's 256 records and its tokens are dense
// ranges. After the transform the four lanes of an MCU exchange their rows with quad-permute DPP moves and every lane
// stores 8 of the MCU's 32 bytes per pixel row (a wave writes 512 contiguous bytes per row).
// ================================================================================================
__global__ __launch_bounds__(256, 4) void k_idct_tok_uyvy422(const gj_geom g, const int16_t* __restrict__ coefs, const uint2* __restrict__ d_rec,
                                                             const uint16_t* __restrict__ d_tok, const uint32_t tok_cap,
                                                             const float* __restrict__ qtab, uint8_t* __restrict__ raw)
{
    if (g.fb.sizes != nullptr) { // frame blockIdx.z of a batch
        const size_t z = blockIdx.z;
        coefs += z * g.fb.coefs; d_rec += z * g.fb.rec; d_tok += z * g.fb.tok;
        raw += z * g.fb.raw;
    }
    __shared__ __attribute__((aligned(16))) uint8_t s_blk[256 * 128];
    __shared__ __attribute__((aligned(16))) uint16_t s_stage[4][GJ_TOK_STAGE];
    // dequantisation tables: [0] for blocks rebuilt from tokens (AC entries / 64, see gj_slot_put), [1] for blocks from the coefficient planes
    __shared__ __attribute__((aligned(8))) float s_q[2][3][64];
    if (threadIdx.x < 192) {
        const float q = qtab[g.comp[threadIdx.x >> 6].q_table * 64 + (threadIdx.x & 63)];
        s_q[0][threadIdx.x >> 6][threadIdx.x & 63] = (threadIdx.x & 63) ? q * 0.015625f : q;
        s_q[1][threadIdx.x >> 6][threadIdx.x & 63] = q;
    }
    const gj_comp_geom& kc = g.comp[1];
    const unsigned nm = (unsigned)(kc.blocks_x * kc.blocks_y);
    const int p = threadIdx.x & 3; // Y0 Y1 Cb Cr
    const unsigned m = blockIdx.x * 64u + (threadIdx.x >> 2);
    const unsigned my = m / (unsigned)kc.blocks_x, mx = m - my * (unsigned)kc.blocks_x;
    const int c = p < 2 ? 0 : p - 1;
    const int lane = threadIdx.x & 63;
    uint8_t* slot = s_blk + threadIdx.x * 128;
    uint16_t* stage = s_stage[threadIdx.x >> 6];
    uint32_t start = 0, cnt = 0, dc = 0;
    bool in_plane = false;
    if (m < nm) {
        const uint2 r = d_rec[(size_t)m * 4 + p];
        start = r.x;
        cnt = r.y >> 16;
        dc = r.y & 0xFFFFu;
        if (cnt == 0xFFFFu) { in_plane = true; cnt = 0; }
        else if (cnt > 63u || start > tok_cap || cnt > tok_cap - start) cnt = 0; // (a record nobody wrote: damaged stream)
    }
    const GjTokRange tr = gj_tok_fetch(d_tok, start, cnt, lane);
    if (tr.fast) {
        *reinterpret_cast<uint4*>(stage + lane * 8) = tr.t0;
        if (lane * 8 + 512 < GJ_TOK_STAGE) *reinterpret_cast<uint4*>(stage + lane * 8 + 512) = tr.t1;
    }
    __syncthreads(); // (s_q)
    const size_t blk = p < 2 ? (size_t)my * g.comp[0].blocks_x + 2 * mx + p : (size_t)m; // (plane address: blocks of long segments only)
    gj_tok_to_slot<true>(slot, stage, lane, tr.fast, tr.S, start, cnt, dc, in_plane,
                   reinterpret_cast<const uint4*>(coefs + g.comp[c].data_offset + (m < nm ? blk : 0) * 64), d_tok);
    uint32_t wb[32];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint4 v = *gj_slot_row(slot, lane, r);
        wb[r * 4] = v.x; wb[r * 4 + 1] = v.y; wb[r * 4 + 2] = v.z; wb[r * 4 + 3] = v.w;
    }
    uint32_t px[16];
    gj_idct_pk(wb, s_q[in_plane ? 1 : 0][c], px);

    // ---- UYVY: dword k of an MCU row = U_k | Y_2k << 8 | V_k << 16 | Y_2k+1 << 24; lane p writes dwords 2p and 2p + 1
    const size_t pitch = (size_t)g.width * 2 + g.width_padding;
    const bool interior = m < nm && (mx * 16 + 16 <= (unsigned)g.width) && (my * 8 + 8 <= (unsigned)g.height);
    const bool aligned = ((pitch | (size_t)raw) & 7) == 0;
    const uint32_t sel_uv = (p & 1) ? 0x07030602u : 0x05010400u; // [U_2p, V_2p, U_2p+1, V_2p+1] out of the chroma lanes' dwords
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int a0 = (int)px[2 * r], a1 = (int)px[2 * r + 1];
        // quad_perm broadcasts: lane 0 = Y0, 1 = Y1, 2 = Cb, 3 = Cr of this MCU
        const uint32_t y00 = (uint32_t)__builtin_amdgcn_update_dpp(0, a0, 0x00, 0xF, 0xF, false), y01 = (uint32_t)__builtin_amdgcn_update_dpp(0, a1, 0x00, 0xF, 0xF, false);
        const uint32_t y10 = (uint32_t)__builtin_amdgcn_update_dpp(0, a0, 0x55, 0xF, 0xF, false), y11 = (uint32_t)__builtin_amdgcn_update_dpp(0, a1, 0x55, 0xF, 0xF, false);
        const uint32_t u0 = (uint32_t)__builtin_amdgcn_update_dpp(0, a0, 0xAA, 0xF, 0xF, false), u1 = (uint32_t)__builtin_amdgcn_update_dpp(0, a1, 0xAA, 0xF, 0xF, false);
        const uint32_t v0 = (uint32_t)__builtin_amdgcn_update_dpp(0, a0, 0xFF, 0xF, 0xF, false), v1 = (uint32_t)__builtin_amdgcn_update_dpp(0, a1, 0xFF, 0xF, 0xF, false);
        const uint32_t ys = p == 0 ? y00 : p == 1 ? y01 : p == 2 ? y10 : y11; // Y_4p .. Y_4p+3
        const uint32_t us = (p >> 1) ? u1 : u0, vs = (p >> 1) ? v1 : v0;
        const uint32_t uv = __builtin_amdgcn_perm(vs, us, sel_uv);
        const uint32_t d0 = __builtin_amdgcn_perm(ys, uv, 0x05010400u), d1 = __builtin_amdgcn_perm(ys, uv, 0x07030602u);
        const unsigned y = my * 8 + r;
        if (interior && aligned) {
            *reinterpret_cast<uint2*>(raw + (size_t)y * pitch + (size_t)mx * 32 + p * 8) = make_uint2(d0, d1);
        } else if (m < nm && y < (unsigned)g.height) {
            // the generic store writes chroma only with the even pixel and whole pixels only (k_postprocess)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint32_t d = k ? d1 : d0;
                const unsigned x0 = mx * 16 + 2 * (2 * p + k);
                uint8_t* q = raw + (size_t)y * pitch + (size_t)x0 * 2;
                if (x0 < (unsigned)g.raw_width) { q[0] = (uint8_t)d; q[1] = (uint8_t)(d >> 8); }
                if (x0 + 1 < (unsigned)g.raw_width) { q[2] = (uint8_t)(d >> 16); q[3] = (uint8_t)(d >> 24); }
            }
        }
    }
}

// ================================================================================================
// Fused fast path for packed 4:2:2 (UYVY) output without colour transform (BASELINE config 4): one thread per MCU takes
// its two luminance blocks (256 contiguous bytes), Cb and Cr, transforms them in registers, interleaves the samples with
// byte permutes and stores 8 rows x 32 B. Replaces k_idct + k_postprocess (one thread per pixel) and the planar round trip.
// ================================================================================================
__global__ __launch_bounds__(256, 2) void k_idct_fused_uyvy422(const gj_geom g, int16_t* __restrict__ coefs, const float* __restrict__ qtab,
                                                               uint8_t* __restrict__ raw, const int zero)
{
    if (g.fb.sizes != nullptr) { // frame blockIdx.z of a batch
        coefs += (size_t)blockIdx.z * g.fb.coefs;
        raw += (size_t)blockIdx.z * g.fb.raw;
    }
    __shared__ __attribute__((aligned(8))) float s_q[3][64];
    if (threadIdx.x < 192) s_q[threadIdx.x >> 6][threadIdx.x & 63] = qtab[g.comp[threadIdx.x >> 6].q_table * 64 + (threadIdx.x & 63)];
    __syncthreads();
    const gj_comp_geom& kc = g.comp[1];
    const unsigned nm = (unsigned)(kc.blocks_x * kc.blocks_y);
    const unsigned m = blockIdx.x * 256u + threadIdx.x;
    if (m >= nm) return;
    const unsigned my = m / (unsigned)kc.blocks_x, mx = m - my * (unsigned)kc.blocks_x;
    uint32_t pk[4][16];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int c = b < 2 ? 0 : b - 1;
        const size_t blk = b < 2 ? (size_t)my * g.comp[0].blocks_x + 2 * mx + b : (size_t)m;
        uint4* p = reinterpret_cast<uint4*>(coefs + g.comp[c].data_offset + blk * 64);
        uint32_t w[32];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint4 v = p[r];
            if (zero) p[r] = make_uint4(0, 0, 0, 0);
            w[r * 4] = v.x; w[r * 4 + 1] = v.y; w[r * 4 + 2] = v.z; w[r * 4 + 3] = v.w;
        }
        gj_idct_pk(w, s_q[c], pk[b]);
#pragma unroll
        for (int t = 0; t < 16; t++) GJ_KEEP(pk[b][t]); // one transform at a time
    }
    const size_t pitch = (size_t)g.width * 2 + g.width_padding;
    const bool interior = (mx * 16 + 16 <= (unsigned)g.width) && (my * 8 + 8 <= (unsigned)g.height);
    const bool aligned = ((pitch | (size_t)raw) & 15) == 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        // UYVY: dword k = U_k | Y_2k << 8 | V_k << 16 | Y_2k+1 << 24
        uint32_t d[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t yy = pk[k >> 2][2 * r + ((k >> 1) & 1)]; // four luminance samples, two of them ours
            const uint32_t uu = pk[2][2 * r + (k >> 2)], vv = pk[3][2 * r + (k >> 2)];
            // bytes: U_k from uu byte (k & 3), Y from yy bytes 2(k&1), 2(k&1)+1, V_k from vv byte (k & 3)
            const uint32_t uv = __builtin_amdgcn_perm(vv, uu, 0x0C040C00u + (uint32_t)(k & 3) * 0x00010001u); // [U_k, 0, V_k, 0]
            const uint32_t ys = __builtin_amdgcn_perm(0u, yy, (k & 1) ? 0x030C020Cu : 0x010C000Cu);       // [0, Y_2k, 0, Y_2k+1]
            d[k] = uv | ys;
        }
        const unsigned y = my * 8 + r;
        if (interior && aligned) {
            uint4* p = reinterpret_cast<uint4*>(raw + (size_t)y * pitch + (size_t)mx * 32);
            p[0] = make_uint4(d[0], d[1], d[2], d[3]);
            p[1] = make_uint4(d[4], d[5], d[6], d[7]);
        } else if (y < (unsigned)g.height) {
            // the generic store writes chroma only with the even pixel and whole pixels only (k_postprocess)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const unsigned x0 = mx * 16 + 2 * k;
                uint8_t* q = raw + (size_t)y * pitch + (size_t)x0 * 2;
                if (x0 < (unsigned)g.raw_width) { q[0] = (uint8_t)d[k]; q[1] = (uint8_t)(d[k] >> 8); }
                if (x0 + 1 < (unsigned)g.raw_width) { q[2] = (uint8_t)(d[k] >> 16); q[3] = (uint8_t)(d[k] >> 24); }
            }
        }
    }
}

// ================================================================================================
// Generic postprocessor: one thread per output pixel (src/gpujpeg_postprocessor.cu:193-217 and the
// stores of src/gpujpeg_preprocessor_common.cuh:118-203).
// ================================================================================================
__global__ __launch_bounds__(256) void k_postprocess(const gj_geom g, const uint8_t* __restrict__ planes, uint8_t* __restrict__ raw)
{
    if (g.fb.sizes != nullptr) { // frame blockIdx.z of a batch
        planes += (size_t)blockIdx.z * g.fb.coefs;
        raw += (size_t)blockIdx.z * g.fb.raw;
    }
    const unsigned W = (unsigned)g.raw_width, H = (unsigned)g.height;
    const unsigned pos = blockIdx.x * 256u + threadIdx.x;
    if (pos >= W * H) return;
    const unsigned y = pos / W, x = pos - y * W;
    int v[4] = {0, 0, 0, g.pixel_format == GJ_PF_4444_P0123 ? 0xFF : 0};
#pragma unroll
    for (int c = 0; c < GJ_MAX_COMP; c++) {
        if (c >= g.comp_count) break;
        const gj_comp_geom& k = g.comp[c];
        v[c] = planes[k.data_offset + (size_t)(y / (unsigned)k.sub_v) * k.data_width + x / (unsigned)k.sub_h];
    }
    if (g.comp_count == 1) { // single channel expanded for the colour transform (:127-170)
        if (g.color_space_internal == GJ_CS_RGB) v[1] = v[2] = v[0];
        else v[1] = v[2] = 128;
    }
    gj_color_transform(g.color_space_internal, g.color_space, v[0], v[1], v[2]);
    switch (g.pixel_format) {
    case GJ_PF_U8: raw[(size_t)pos + (size_t)g.width_padding * y] = (uint8_t)v[0]; break;
    case GJ_PF_444_P012: {
        uint8_t* p = raw + (size_t)pos * 3 + (size_t)g.width_padding * y;
        p[0] = (uint8_t)v[0]; p[1] = (uint8_t)v[1]; p[2] = (uint8_t)v[2];
        break; }
    case GJ_PF_4444_P0123: {
        uint8_t* p = raw + (size_t)pos * 4 + (size_t)g.width_padding * y;
        p[0] = (uint8_t)v[0]; p[1] = (uint8_t)v[1]; p[2] = (uint8_t)v[2]; p[3] = (uint8_t)v[3];
        break; }
    case GJ_PF_444_P0P1P2:
        raw[pos] = (uint8_t)v[0]; raw[(size_t)W * H + pos] = (uint8_t)v[1]; raw[(size_t)2 * W * H + pos] = (uint8_t)v[2];
        break;
    case GJ_PF_422_P0P1P2:
        raw[pos] = (uint8_t)v[0];
        if ((x & 1) == 0) {
            raw[(size_t)W * H + pos / 2] = (uint8_t)v[1];
            raw[(size_t)W * H + (size_t)H * ((W + 1) / 2) + pos / 2] = (uint8_t)v[2];
        }
        break;
    case GJ_PF_422_P1020: {
        const size_t off = (size_t)pos * 2 + (size_t)g.width_padding * y;
        raw[off + 1] = (uint8_t)v[0];
        raw[off] = (uint8_t)((x & 1) == 0 ? v[1] : v[2]);
        break; }
    case GJ_PF_420_P0P1P2:
        raw[pos] = (uint8_t)v[0];
        if ((pos & 1) == 0 && (y & 1) == 0) {
            raw[(size_t)W * H + (size_t)(y / 2) * ((W + 1) / 2) + x / 2] = (uint8_t)v[1];
            raw[(size_t)W * H + (size_t)((H + 1) / 2 + y / 2) * ((W + 1) / 2) + x / 2] = (uint8_t)v[2];
        }
        break;
    default: break;
    }
}

// planar output whose layout equals the component layout (src/gpujpeg_postprocessor.cu:404-434)
__global__ __launch_bounds__(256) void k_copy_planes_out(const gj_geom g, const uint8_t* __restrict__ planes, uint8_t* __restrict__ raw)
{
    if (g.fb.sizes != nullptr) { // frame blockIdx.z of a batch
        planes += (size_t)blockIdx.z * g.fb.coefs;
        raw += (size_t)blockIdx.z * g.fb.raw;
    }
    size_t dst_off = 0;
    for (int c = 0; c < g.comp_count; c++) {
        const gj_comp_geom& k = g.comp[c];
        const size_t dpitch = (size_t)k.width + g.width_padding;
        const size_t n = (size_t)k.width * k.height;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            const size_t y = i / k.width, x = i - y * k.width;
            raw[dst_off + y * dpitch + x] = planes[k.data_offset + y * k.data_width + x];
        }
        dst_off += dpitch * k.height;
    }
}


// ================================================================================================
// Kernel selection and launch
// ================================================================================================
typedef void (*gj_idct_fused_t)(const gj_geom, int16_t*, const float*, uint8_t*, int);

static gj_idct_fused_t gj_idct_fused_kernel(const gj_geom& g)
{
    if (g.pixel_format != GJ_PF_444_P012 || g.comp_count != 3) return nullptr;
    for (int c = 0; c < 3; c++)
        if (g.comp[c].samp_h != 1 || g.comp[c].samp_v != 1) return nullptr;
    const int from = g.color_space_internal, to = g.color_space;
    if (from == to || from == GJ_CS_NONE || to == GJ_CS_NONE) return k_idct_fused_rgb444<GJ_CS_NONE, GJ_CS_NONE>;
    if (from == GJ_CS_BT601_256 && to == GJ_CS_RGB) return k_idct_fused_rgb444<GJ_CS_BT601_256, GJ_CS_RGB>;
    if (from == GJ_CS_BT601 && to == GJ_CS_RGB) return k_idct_fused_rgb444<GJ_CS_BT601, GJ_CS_RGB>;
    if (from == GJ_CS_BT709 && to == GJ_CS_RGB) return k_idct_fused_rgb444<GJ_CS_BT709, GJ_CS_RGB>;
    if (from == GJ_CS_RGB && to == GJ_CS_BT601_256) return k_idct_fused_rgb444<GJ_CS_RGB, GJ_CS_BT601_256>;
    return nullptr;
}

static gj_idct_tok_t gj_idct_tok_kernel(const gj_geom& g)
{
    if (g.pixel_format != GJ_PF_444_P012 || g.comp_count != 3) return nullptr;
    for (int c = 0; c < 3; c++)
        if (g.comp[c].samp_h != 1 || g.comp[c].samp_v != 1) return nullptr;
    const int from = g.color_space_internal, to = g.color_space;
    if (from == to || from == GJ_CS_NONE || to == GJ_CS_NONE) return k_idct_tok_rgb444<GJ_CS_NONE, GJ_CS_NONE>;
    if (from == GJ_CS_BT601_256 && to == GJ_CS_RGB) return k_idct_tok_rgb444<GJ_CS_BT601_256, GJ_CS_RGB>;
    if (from == GJ_CS_BT601 && to == GJ_CS_RGB) return k_idct_tok_rgb444<GJ_CS_BT601, GJ_CS_RGB>;
    if (from == GJ_CS_BT709 && to == GJ_CS_RGB) return k_idct_tok_rgb444<GJ_CS_BT709, GJ_CS_RGB>;
    if (from == GJ_CS_RGB && to == GJ_CS_BT601_256) return k_idct_tok_rgb444<GJ_CS_RGB, GJ_CS_BT601_256>;
    return nullptr;
}

bool gj_is_uyvy422(const gj_geom& g)
{
    return g.pixel_format == GJ_PF_422_P1020 && g.comp_count == 3 &&
           (g.color_space == g.color_space_internal || g.color_space == GJ_CS_NONE || g.color_space_internal == GJ_CS_NONE) &&
           g.comp[0].samp_h == 2 && g.comp[0].samp_v == 1 && g.comp[1].samp_h == 1 && g.comp[1].samp_v == 1 && g.comp[2].samp_h == 1 &&
           g.comp[2].samp_v == 1 && g.comp[0].blocks_x == 2 * g.comp[1].blocks_x && g.comp[0].blocks_y == g.comp[1].blocks_y &&
           g.comp[2].blocks_x == g.comp[1].blocks_x && g.comp[2].blocks_y == g.comp[1].blocks_y;
}

// the token-fed IDCT kernel for this configuration, or nullptr: non-interleaved 4:4:4 scans (plane order == coding order), and the
// interleaved scan of packed 4:2:2 (one lane per block in coding order)
gj_idct_tok_t gj_idct_tok_for(const gj_geom& g)
{
    if (!g.interleaved) return gj_idct_tok_kernel(g);
    if (gj_is_uyvy422(g) && g.blocks_per_mcu == 4 && g.mcu_count == g.comp[1].blocks_x * g.comp[1].blocks_y && g.mcu_comp[0] == 0 && g.mcu_comp[1] == 0 &&
        g.mcu_comp[2] == 1 && g.mcu_comp[3] == 2 && g.mcu_bx[0] == 0 && g.mcu_bx[1] == 1)
        return k_idct_tok_uyvy422;
    return nullptr;
}

// does the IDCT side of this configuration go through the component planes (generic kernels)? A batch needs a set per frame then.
extern "C" int gj_hip_decode_uses_planes(const gj_geom* g, int use_fused)
{
    return !(use_fused && (gj_is_uyvy422(*g) || gj_idct_fused_kernel(*g) != nullptr));
}

void gj_launch_idct(const gj_dec_job* job, hipStream_t st, gj_idct_tok_t idct_tok, gj_event_t* ev)
{
    const gj_geom& g = job->g;
    const bool uyvy = job->use_fused && gj_is_uyvy422(g);
    gj_idct_fused_t fused = job->use_fused ? gj_idct_fused_kernel(g) : nullptr;
    const unsigned frames = job->batch.count > 1 ? job->batch.count : 1u; // (batches: every kernel below but the flip and the remap)
    if (idct_tok) {
        const unsigned nb = g.interleaved ? (unsigned)g.block_count : (unsigned)(g.comp[0].blocks_x * g.comp[0].blocks_y); // one lane per block (position)
        hipLaunchKernelGGL(idct_tok, dim3((nb + 255) / 256, 1, frames), dim3(256), 0, st, g, job->d_coefs, (const uint2*)job->d_blkrec, (const uint16_t*)job->d_tok, job->tok_cap,
                           job->d_qtabf, job->d_raw);
        if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[2], st));
    } else if (uyvy) {
        const unsigned nm = (unsigned)(g.comp[1].blocks_x * g.comp[1].blocks_y);
        hipLaunchKernelGGL(k_idct_fused_uyvy422, dim3((nm + 255) / 256, 1, frames), dim3(256), 0, st, g, job->d_coefs, job->d_qtabf, job->d_raw, job->zero_coefs);
        if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[2], st));
    } else if (fused) {
        const unsigned nb = (unsigned)(g.comp[0].blocks_x * g.comp[0].blocks_y);
        hipLaunchKernelGGL(fused, dim3((nb + 255) / 256, 1, frames), dim3(256), 0, st, g, job->d_coefs, job->d_qtabf, job->d_raw, job->zero_coefs);
        if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[2], st));
    } else {
        hipLaunchKernelGGL(k_idct, dim3(((unsigned)g.block_count + 255) / 256, 1, frames), dim3(256), 0, st, g, job->d_coefs, job->d_qtabf,
                           job->d_planes, job->zero_coefs);
        if (job->flipped) hipLaunchKernelGGL(k_flip_planes, dim3(1024), dim3(256), 0, st, g, job->d_planes); // src/gpujpeg_postprocessor.cu:447
        if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[2], st));
        if (g.no_transform) {
            hipLaunchKernelGGL(k_copy_planes_out, dim3(2048, 1, frames), dim3(256), 0, st, g, job->d_planes, job->d_raw);
        } else {
            const unsigned n = (unsigned)g.raw_width * (unsigned)g.height;
            hipLaunchKernelGGL(k_postprocess, dim3((n + 255) / 256, 1, frames), dim3(256), 0, st, g, job->d_planes, job->d_raw);
        }
    }
    gj_debug_stage(job->tune.debug_sync != 0, st, "idct / postprocess");
    if (job->channel_remap) { // src/gpujpeg_postprocessor.cu:450,493: the finished image is permuted in place
        const unsigned n = (unsigned)g.width * (unsigned)g.height;
        hipLaunchKernelGGL(k_channel_remap, dim3((n + 255) / 256), dim3(256), 0, st, g, job->d_raw, job->channel_remap & 0xFFFFu);
    }
}
