// gj_decode.hip -- MI355X (gfx950, wave64) JPEG decoder kernels.
//
//   k_huffman_decode      one lane per restart segment, decode tables + one private block per lane in LDS
//   k_idct_fused_rgb444   dequant + IDCT of the three component blocks + colour transform + packed store
//   k_idct / k_postprocess / k_copy_planes_out   generic path through padded planes
//   k_find_rst / k_emit_rst   device-side marker scan (segment table without touching the host)
//
// Restates src/gpujpeg_huffman_gpu_decoder.cu:135-495 (entropy decoding semantics; identical results to
// src/gpujpeg_huffman_cpu_decoder.c:245-372), src/gpujpeg_dct_gpu.cu:312-366,472-618 and
// src/gpujpeg_postprocessor.cu:49-217.
#include <hip/hip_runtime.h>

#include "gj_device.h"
#include "gj_hip.h"

// ================================================================================================
// Entropy decoder: one lane per restart segment (the code is serial inside a segment).
//
// The hot loop touches no global memory on its input side: every lane owns a 256-byte window of its segment in LDS
// (rows of 65 dwords, so both the cooperative fill and the per-lane reads are bank-conflict free). The wave fills
// the windows together -- for lane j, all 64 lanes fetch 256 contiguous bytes -- first for everybody, later only for
// the lanes that have used three quarters of their window (rare: an average q75 segment is ~170 bytes). With no loads
// in the loop, the 2-byte coefficient stores are never waited for (on gfx9 loads and stores share vmcnt).
// Byte stuffing is removed on the fly: a dword without 0xFF (98.5 % of them) is appended with one shift.
// Lanes do not wait for each other at block boundaries: one symbol per iteration, every lane moves on to its next
// block on its own, so a wave needs max-over-lanes(symbols of a segment) iterations.
// ================================================================================================
#define GJ_WIN_DW 64
#define GJ_WIN_STRIDE 65

struct GjBits {
    const uint32_t* src; // global address of window dword 0
    int rd;              // next window dword to consume
    int remaining;       // bytes of the segment not yet moved into the accumulator
    int prev_ff;         // last consumed byte was 0xFF (a following 0x00 is stuffing)
    uint64_t acc;        // valid bits are left aligned
    int n;
};

// (re)fill the windows of the lanes in `mask` from their `src`; all 64 lanes must call this
__device__ __forceinline__ void gj_fill_windows(unsigned long long mask, const uint32_t* src, const uint32_t* end, uint32_t* s_win, int lane)
{
    const unsigned lo = (unsigned)(uintptr_t)src, hi = (unsigned)((uintptr_t)src >> 32);
    while (mask) {
        uint32_t v[4];
        int js[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            js[u] = -1;
            v[u] = 0;
            if (mask) {
                const int j = __builtin_ctzll(mask);
                mask &= mask - 1;
                js[u] = j;
                const uint32_t* a = reinterpret_cast<const uint32_t*>(((uintptr_t)(unsigned)__builtin_amdgcn_readlane((int)hi, j) << 32) |
                                                                      (unsigned)__builtin_amdgcn_readlane((int)lo, j)) + lane;
                if (a < end) v[u] = *a;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (js[u] >= 0) s_win[js[u] * GJ_WIN_STRIDE + lane] = v[u];
    }
}

// canonical search for codes longer than the fast table (ITU T.81 F.2.2.3); rare
__device__ __forceinline__ uint32_t gj_decode_slow(uint32_t hi, const uint16_t* t)
{
    const uint16_t* maxcode = t + 1024;           // [18] as (lo, hi)
    const uint16_t* valptr = t + 1024 + 36;       // [17]
    const uint16_t* mincode = t + 1024 + 36 + 17; // [17] as (lo, hi)
    const uint16_t* vals = t + 1024 + 36 + 17 + 34;
    for (int l = GJ_DEC_FAST_BITS + 1; l <= 16; l++) {
        const int code = (int)(hi >> (32 - l));
        const int mx = (int)((uint32_t)maxcode[2 * l] | ((uint32_t)maxcode[2 * l + 1] << 16));
        if (mx >= 0 && code <= mx) {
            const int mn = (int)((uint32_t)mincode[2 * l] | ((uint32_t)mincode[2 * l + 1] << 16));
            return ((uint32_t)l << 8) | vals[(valptr[l] + code - mn) & 0xFF];
        }
    }
    return (16u << 8); // corrupt stream: consume 16 bits, symbol 0 (output is undefined but in bounds)
}

// The loop body is written to compile to (almost) straight-line predicated code: a lone wave per SIMD pays for every
// divergent branch with exec-mask round trips, which dominated the first versions of this kernel.
template <bool INTERLEAVED>
__global__ __launch_bounds__(256) void k_huffman_decode(const gj_geom g, const uint8_t* __restrict__ jpeg, const uint64_t jpeg_size,
                                                        const uint32_t* __restrict__ seg_pos, const uint32_t* __restrict__ seg_len,
                                                        const uint32_t* __restrict__ seg_index, const uint32_t* __restrict__ seg_count_ptr,
                                                        const int seg_count_max, const uint16_t* __restrict__ tabs, int16_t* __restrict__ coefs)
{
    __shared__ uint16_t s_tab[8 * GJ_DEC_TAB_WORDS];
    __shared__ uint32_t s_win_all[4 * 64 * GJ_WIN_STRIDE];
    __shared__ uint8_t s_zz[64 + 32];
    for (int t = threadIdx.x; t < 8 * GJ_DEC_TAB_WORDS / 2; t += 256)
        reinterpret_cast<uint32_t*>(s_tab)[t] = reinterpret_cast<const uint32_t*>(tabs)[t];
    if (threadIdx.x < 96) s_zz[threadIdx.x] = threadIdx.x < 64 ? GJ_ZZ[threadIdx.x] : 63;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t* s_win = s_win_all + (threadIdx.x >> 6) * 64 * GJ_WIN_STRIDE; // this wave's 64 rows
    const uint32_t* s_row = s_win + lane * GJ_WIN_STRIDE;
    const uint32_t* end = reinterpret_cast<const uint32_t*>((reinterpret_cast<uintptr_t>(jpeg) + jpeg_size + 3) & ~(uintptr_t)3);

    const int seg_count = seg_count_ptr ? min((int)*seg_count_ptr, seg_count_max) : seg_count_max;
    const int si = blockIdx.x * 256 + threadIdx.x;
    uint32_t s = 0xFFFFFFFFu;
    if (si < seg_count) s = seg_index[si];
    GjSeg sg;
    sg.nblocks = 0;
    sg.mcu_first = 0;
    sg.comp = 0;
    if (s < (uint32_t)g.segment_count) sg = gj_segment(g, (int)s);
    int left = sg.nblocks > 0 ? sg.nblocks : 0;

    const uint32_t* src = reinterpret_cast<const uint32_t*>(jpeg); // global address of window dword 0
    int rd = 0;          // next window dword
    int remaining = 0;   // bytes of the segment not yet moved into the accumulator
    int prev_ff = 0;     // last byte moved was 0xFF (a following 0x00 is stuffing)
    uint64_t acc = 0;    // valid bits are left aligned
    int n = 0;
    int lead = 0;
    if (left > 0) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(jpeg) + seg_pos[si];
        src = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
        remaining = (int)seg_len[si];
        lead = (int)(a & 3);
    }
    gj_fill_windows(__ballot(left > 0), src, end, s_win, lane);
    if (lead) { // drop the bytes in front of the segment inside its first dword
        const uint32_t w = s_row[0];
        rd = 1;
        for (int i = lead; i < 4 && remaining > 0; i++) {
            const uint32_t byte = (w >> (8 * i)) & 0xFFu;
            remaining--;
            if (prev_ff && byte == 0) { prev_ff = 0; continue; }
            prev_ff = byte == 0xFFu;
            acc |= (uint64_t)byte << (56 - n);
            n += 8;
        }
    }

    // block cursor
    const int P = g.blocks_per_mcu;
    int p = 0;
    unsigned mx = 0, my = 0;
    int comp = sg.comp;
    uint64_t off;
    if (INTERLEAVED) {
        my = (unsigned)sg.mcu_first / (unsigned)g.mcu_count_x;
        mx = (unsigned)sg.mcu_first - my * (unsigned)g.mcu_count_x;
        comp = g.mcu_comp[0];
        const gj_comp_geom& kc = g.comp[comp];
        off = kc.data_offset + ((uint64_t)(my * kc.samp_v + g.mcu_by[0]) * kc.blocks_x + mx * kc.samp_h + g.mcu_bx[0]) * 64;
    } else {
        off = g.comp[comp].data_offset + (uint64_t)sg.mcu_first * 64;
    }
    const uint16_t* tdc = s_tab + (g.comp[comp].dc_table * 2 + 0) * GJ_DEC_TAB_WORDS;
    const uint16_t* tac = s_tab + (g.comp[comp].ac_table * 2 + 1) * GJ_DEC_TAB_WORDS;
    int dc0 = 0, dc1 = 0, dc2 = 0, dc3 = 0;
    int kk = 0; // 0: DC expected, 1..63: next AC position
    while (__any(left > 0)) {
        // lanes that have used 3/4 of their window get a fresh one starting at their current dword (wave-uniform branch)
        const unsigned long long need = __ballot(left > 0 && rd >= GJ_WIN_DW - 16);
        if (need) {
            if ((need >> lane) & 1) { src += rd; rd = 0; }
            gj_fill_windows(need, src, end, s_win, lane);
        }
        // ---- refill: one dword when fewer than 33 bits are left
        const bool want = left > 0 && n <= 32;
        const uint32_t w = s_row[rd];
        const uint32_t inv = ~w;
        const bool plain = !(((inv - 0x01010101u) & ~inv & 0x80808080u) != 0) && !prev_ff && remaining >= 4;
        if (__any(want && !plain)) { // some lane meets 0xFF, a stuffed zero or the tail of its segment: byte-wise for those lanes
            if (want && !plain) {
                if (remaining <= 0) {
                    n = 64; // zero bits past the end (src/gpujpeg_huffman_cpu_decoder.c:80-118)
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (remaining > 0) {
                            const uint32_t byte = (w >> (8 * i)) & 0xFFu;
                            remaining--;
                            if (prev_ff && byte == 0) {
                                prev_ff = 0;
                            } else {
                                prev_ff = byte == 0xFFu;
                                acc |= (uint64_t)byte << (56 - n);
                                n += 8;
                            }
                        }
                    }
                    rd++;
                }
            }
        }
        if (want && plain) {
            acc |= (uint64_t)__builtin_bswap32(w) << (32 - n);
            n += 32;
            remaining -= 4;
            rd++;
        }
        // ---- one symbol (needs up to 16 + 11 bits)
        const bool go = left > 0 && n >= 27;
        const uint32_t hi = (uint32_t)(acc >> 32);
        const uint16_t* t = kk == 0 ? tdc : tac;
        uint32_t ent = t[hi >> (32 - GJ_DEC_FAST_BITS)];
        if (__any(go && ent == 0)) {
            if (go && ent == 0) ent = gj_decode_slow(hi, t);
        }
        if (go) {
            const int used = (int)(ent >> 8);
            const int sym = (int)(ent & 0xFFu);
            const int run = sym >> 4, sz = sym & 15;
            const uint32_t bits = sz ? (hi << used) >> (32 - sz) : 0u;
            int v = (sz && bits < (1u << (sz - 1))) ? (int)bits - (int)((1u << sz) - 1u) : (int)bits;
            acc <<= (used + sz);
            n -= used + sz;
            const bool is_dc = kk == 0;
            if (is_dc) {
                if (INTERLEAVED) {
                    v += (comp == 0 ? dc0 : comp == 1 ? dc1 : comp == 2 ? dc2 : dc3);
                    if (comp == 0) dc0 = v; else if (comp == 1) dc1 = v; else if (comp == 2) dc2 = v; else dc3 = v;
                } else {
                    v += dc0;
                    dc0 = v;
                }
            }
            const int pos = kk + run; // DC symbols have run 0
            const bool store = (is_dc || sz != 0) && pos < 64;
            if (store) coefs[off + s_zz[pos]] = (int16_t)v;
            kk = (!is_dc && sz == 0) ? (run == 15 ? kk + 16 : 64) : pos + 1;
            if (kk >= 64) { // next block of this segment
                kk = 0;
                left--;
                if (!INTERLEAVED) {
                    off += 64;
                } else {
                    if (++p == P) {
                        p = 0;
                        if (++mx == (unsigned)g.mcu_count_x) { mx = 0; my++; }
                    }
                    comp = g.mcu_comp[p];
                    const gj_comp_geom& kc = g.comp[comp];
                    off = kc.data_offset + ((uint64_t)(my * kc.samp_v + g.mcu_by[p]) * kc.blocks_x + mx * kc.samp_h + g.mcu_bx[p]) * 64;
                    tdc = s_tab + (kc.dc_table * 2 + 0) * GJ_DEC_TAB_WORDS;
                    tac = s_tab + (kc.ac_table * 2 + 1) * GJ_DEC_TAB_WORDS;
                }
            }
        }
    }
}

// ================================================================================================
// Dequantisation + IDCT, one thread per block
// ================================================================================================
__device__ __forceinline__ void gj_load_dequant(const int16_t* __restrict__ src, const uint16_t* __restrict__ q, float (&d)[64])
{
    const uint4* p = reinterpret_cast<const uint4*>(src);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint4 w = p[r];
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int lo = (int)(int16_t)(ws[j] & 0xFFFF), hi = (int)ws[j] >> 16;
            // integer product first, then one conversion (src/gpujpeg_dct_gpu.cu:497-500)
            d[r * 8 + 2 * j] = (float)(lo * (int)q[r * 8 + 2 * j]);
            d[r * 8 + 2 * j + 1] = (float)(hi * (int)q[r * 8 + 2 * j + 1]);
        }
    }
}

__global__ __launch_bounds__(256) void k_idct(const gj_geom g, const int16_t* __restrict__ coefs, const uint16_t* __restrict__ qtab,
                                              uint8_t* __restrict__ planes)
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
    float d[64];
    gj_load_dequant(coefs + (size_t)gb * 64, qtab + k.q_table * 64, d);
    int o[64];
    gj_idct_block(d, o);
    uint8_t* dst = planes + k.data_offset + (size_t)by * 8 * k.data_width + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint2 w;
        w.x = (uint32_t)o[r * 8] | ((uint32_t)o[r * 8 + 1] << 8) | ((uint32_t)o[r * 8 + 2] << 16) | ((uint32_t)o[r * 8 + 3] << 24);
        w.y = (uint32_t)o[r * 8 + 4] | ((uint32_t)o[r * 8 + 5] << 8) | ((uint32_t)o[r * 8 + 6] << 16) | ((uint32_t)o[r * 8 + 7] << 24);
        *reinterpret_cast<uint2*>(dst + (size_t)r * k.data_width) = w;
    }
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

template <int CS_FROM, int CS_TO>
__global__ __launch_bounds__(256) void k_idct_fused_rgb444(const gj_geom g, const int16_t* __restrict__ coefs,
                                                           const uint16_t* __restrict__ qtab, uint8_t* __restrict__ raw)
{
    const gj_comp_geom& k0 = g.comp[0];
    const unsigned nb = (unsigned)(k0.blocks_x * k0.blocks_y);
    const unsigned lb = blockIdx.x * 256u + threadIdx.x;
    if (lb >= nb) return;
    const unsigned by = lb / (unsigned)k0.blocks_x, bx = lb - by * (unsigned)k0.blocks_x;
    uint32_t pk[3][16];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float d[64];
        gj_load_dequant(coefs + g.comp[c].data_offset + (size_t)lb * 64, qtab + g.comp[c].q_table * 64, d);
        int o[64];
        gj_idct_block(d, o);
#pragma unroll
        for (int i = 0; i < 16; i++)
            pk[c][i] = (uint32_t)o[i * 4] | ((uint32_t)o[i * 4 + 1] << 8) | ((uint32_t)o[i * 4 + 2] << 16) | ((uint32_t)o[i * 4 + 3] << 24);
    }
    const size_t pitch = (size_t)g.width * 3 + g.width_padding;
    const bool interior = (bx * 8 + 8 <= (unsigned)g.width) && (by * 8 + 8 <= (unsigned)g.height);
    const bool aligned = ((pitch | (size_t)raw) & 3) == 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint32_t px[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int x = 0; x < 8; x++) {
            const int i = r * 8 + x;
            int c0 = (pk[0][i >> 2] >> ((i & 3) * 8)) & 0xFF;
            int c1 = (pk[1][i >> 2] >> ((i & 3) * 8)) & 0xFF;
            int c2 = (pk[2][i >> 2] >> ((i & 3) * 8)) & 0xFF;
            gj_color_static_d<CS_FROM, CS_TO>(c0, c1, c2);
            const int b0 = x * 3, b1 = x * 3 + 1, b2 = x * 3 + 2;
            px[b0 >> 2] |= (uint32_t)c0 << ((b0 & 3) * 8);
            px[b1 >> 2] |= (uint32_t)c1 << ((b1 & 3) * 8);
            px[b2 >> 2] |= (uint32_t)c2 << ((b2 & 3) * 8);
        }
        const unsigned y = by * 8 + r;
        if (interior && aligned) {
            uint2* p = reinterpret_cast<uint2*>(raw + (size_t)y * pitch + (size_t)bx * 24);
            p[0] = make_uint2(px[0], px[1]);
            p[1] = make_uint2(px[2], px[3]);
            p[2] = make_uint2(px[4], px[5]);
        } else if (y < (unsigned)g.height) {
#pragma unroll
            for (int byte = 0; byte < 24; byte++) {
                const unsigned x = bx * 8 + byte / 3;
                if (x < (unsigned)g.width) raw[(size_t)y * pitch + (size_t)x * 3 + byte % 3] = (uint8_t)(px[byte >> 2] >> ((byte & 3) * 8));
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
// Launcher
// ================================================================================================
typedef void (*gj_idct_fused_t)(const gj_geom, const int16_t*, const uint16_t*, uint8_t*);

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

extern "C" int gj_hip_decode(const gj_dec_job* job, gj_stream_t stream, gj_event_t ev[4])
{
    hipStream_t st = (hipStream_t)stream;
    const gj_geom& g = job->g;
    if (g.blocks_per_mcu > GJ_MAX_MCU_BLOCKS) return -1;
    if (ev) (void)hipEventRecord((hipEvent_t)ev[0], st);
    // the entropy decoder stores only non-zero coefficients
    (void)hipMemsetAsync(job->d_coefs, 0, g.data_size * sizeof(int16_t), st);
    if (job->seg_count > 0)
    {
        auto kernel = g.interleaved ? k_huffman_decode<true> : k_huffman_decode<false>;
        hipLaunchKernelGGL(kernel, dim3(((unsigned)job->seg_count + 255) / 256), dim3(256), 0, st, g, job->d_jpeg, job->jpeg_size,
                           job->d_seg_pos, job->d_seg_len, job->d_seg_index, job->d_seg_count, job->seg_count, job->d_huff_tab, job->d_coefs);
    }
    if (ev) (void)hipEventRecord((hipEvent_t)ev[1], st);
    gj_idct_fused_t fused = job->use_fused ? gj_idct_fused_kernel(g) : nullptr;
    if (fused) {
        const unsigned nb = (unsigned)(g.comp[0].blocks_x * g.comp[0].blocks_y);
        hipLaunchKernelGGL(fused, dim3((nb + 255) / 256), dim3(256), 0, st, g, job->d_coefs, job->d_qtab, job->d_raw);
        if (ev) (void)hipEventRecord((hipEvent_t)ev[2], st);
    } else {
        hipLaunchKernelGGL(k_idct, dim3(((unsigned)g.block_count + 255) / 256), dim3(256), 0, st, g, job->d_coefs, job->d_qtab,
                           job->d_planes);
        if (ev) (void)hipEventRecord((hipEvent_t)ev[2], st);
        if (g.no_transform) {
            hipLaunchKernelGGL(k_copy_planes_out, dim3(2048), dim3(256), 0, st, g, job->d_planes, job->d_raw);
        } else {
            const unsigned n = (unsigned)g.raw_width * (unsigned)g.height;
            hipLaunchKernelGGL(k_postprocess, dim3((n + 255) / 256), dim3(256), 0, st, g, job->d_planes, job->d_raw);
        }
    }
    if (ev) (void)hipEventRecord((hipEvent_t)ev[3], st);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ================================================================================================
// Device-side segment discovery. Inside entropy-coded data 0xFF is followed by 0x00 (stuffing),
// by 0xD0..0xD7 (restart marker = segment boundary) or by the marker that ends the scan. Three small launches turn
// the bytes [begin, size) into the (offset, length, geometric index) table k_huffman_decode consumes, without the host
// touching the stream (the reference walks it with memchr and copies every segment, src/gpujpeg_reader.c:1039-1155):
//   k_marker_count   per 2 KiB chunk: number of RSTn; every other marker is appended (rare) to a small list
//   k_marker_rank    exclusive scan of the chunk counts
//   k_marker_emit    ordered list of RSTn positions
//   k_build_segments segment table for every scan + the summary the host validates (gj_scan_summary)
// ================================================================================================
#define GJ_SCAN_CHUNK 2048

__device__ __forceinline__ int gj_marker_at(const uint8_t* __restrict__ jpeg, uint64_t p, uint64_t size)
{
    // 0: none, 1: RSTn, 2: other marker
    if (p + 1 >= size || jpeg[p] != 0xFF) return 0;
    const int m = jpeg[p + 1];
    if (m == 0x00 || m == 0xFF) return 0;
    return (m & 0xF8) == 0xD0 ? 1 : 2;
}

__global__ __launch_bounds__(256) void k_marker_count(const uint8_t* __restrict__ jpeg, uint64_t begin, uint64_t size,
                                                      uint32_t* __restrict__ chunk_count, gj_scan_summary* __restrict__ sum)
{
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint64_t base = begin + (uint64_t)blockIdx.x * GJ_SCAN_CHUNK + threadIdx.x * 8u;
    uint32_t n = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int k = gj_marker_at(jpeg, base + i, size);
        if (k == 1) n++;
        if (k == 2) { // scan boundary material: keep position, code and the 16 bytes that follow
            const uint32_t slot = atomicAdd(&sum->other_count, 1u);
            if (slot < GJ_SCAN_MAX_OTHER) {
                sum->other_pos[slot] = (uint32_t)(base + i);
                sum->other_code[slot] = jpeg[base + i + 1];
                for (int b = 0; b < 16; b++) sum->other_bytes[slot][b] = base + i + 2 + b < size ? jpeg[base + i + 2 + b] : 0;
            }
        }
    }
    if (n) atomicAdd(&s_n, n);
    __syncthreads();
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = s_n;
}

__global__ __launch_bounds__(1024) void k_marker_rank(uint32_t* __restrict__ chunk_count, uint32_t chunks, gj_scan_summary* __restrict__ sum)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < chunks; base += 1024) {
        const uint32_t i = base + t;
        const uint32_t v = i < chunks ? chunk_count[i] : 0;
        const uint32_t inc = gj_wave_incl_scan(v);
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        uint32_t off = s_carry;
        for (int w = 0; w < wave; w++) off += s_w[w];
        if (i < chunks) chunk_count[i] = off + inc - v;
        __syncthreads();
        if (t == 1023) s_carry = off + inc;
        __syncthreads();
    }
    if (t == 0) sum->rst_count = s_carry;
}

__global__ __launch_bounds__(256) void k_marker_emit(const uint8_t* __restrict__ jpeg, uint64_t begin, uint64_t size,
                                                     const uint32_t* __restrict__ chunk_rank, uint32_t* __restrict__ rst_pos, uint32_t max_rst)
{
    __shared__ uint32_t s_tmp[4];
    const uint64_t base = begin + (uint64_t)blockIdx.x * GJ_SCAN_CHUNK + threadIdx.x * 8u;
    uint32_t mask = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (gj_marker_at(jpeg, base + i, size) == 1) mask |= 1u << i;
    const uint32_t n = (uint32_t)__popc(mask);
    uint32_t total;
    uint32_t r = chunk_rank[blockIdx.x] + gj_wg256_incl_scan(n, s_tmp, &total) - n;
    while (mask) {
        const int i = __builtin_ctz(mask);
        mask &= mask - 1;
        if (r < max_rst) rst_pos[r] = (uint32_t)(base + i);
        r++;
    }
}

// One thread per segment of the table. Scan s is bounded by the "other" markers: it starts after an SOS header and
// ends at the next other marker. Scan 0 starts at `begin` (the host parsed its SOS).
__global__ __launch_bounds__(256) void k_build_segments(const gj_geom g, const uint32_t* __restrict__ rst_pos, uint64_t begin, uint64_t size,
                                                        gj_scan_summary* __restrict__ sum, uint32_t* __restrict__ seg_pos,
                                                        uint32_t* __restrict__ seg_len, uint32_t* __restrict__ seg_index, uint32_t max_segments)
{
    __shared__ uint32_t s_start[GJ_MAX_COMP + 1], s_end[GJ_MAX_COMP + 1], s_first[GJ_MAX_COMP + 2];
    __shared__ int s_scans;
    __shared__ uint32_t s_opos[GJ_SCAN_MAX_OTHER];
    __shared__ uint8_t s_order[GJ_SCAN_MAX_OTHER];
    const uint32_t n_rst = min(sum->rst_count, max_segments);
    const uint32_t n_other = min(sum->other_count, (uint32_t)GJ_SCAN_MAX_OTHER);
    if (threadIdx.x == 0) {
        // order the few other markers by position (insertion sort)
        for (uint32_t i = 0; i < n_other; i++) {
            uint32_t j = i;
            const uint32_t p = sum->other_pos[i];
            while (j > 0 && s_opos[j - 1] > p) { s_opos[j] = s_opos[j - 1]; s_order[j] = s_order[j - 1]; j--; }
            s_opos[j] = p;
            s_order[j] = (uint8_t)i;
        }
        int scans = 0;
        uint32_t start = (uint32_t)begin;
        int status = 0;
        for (uint32_t i = 0; i < n_other && scans < GJ_MAX_COMP; i++) {
            const uint32_t p = s_opos[i];
                        if (p < start) continue; // lies inside a header we already skipped
            s_start[scans] = start;
            s_end[scans] = p;
            scans++;
            const uint8_t* hb = sum->other_bytes[s_order[i]];
            const uint32_t mlen = ((uint32_t)hb[0] << 8) | hb[1];
            const int m = sum->other_code[s_order[i]];
            if (m == 0xDA) { start = p + 2 + mlen; continue; } // next scan
            if (m == 0xD9) { status = 1; break; }              // EOI: done
            status = 2;                                          // something else between scans: let the host walk it
            break;
        }
        if (status == 0) status = 3; // no EOI seen
        // rank of the first RSTn of every scan: binary search in the ordered list
        for (int sc = 0; sc <= scans; sc++) {
            const uint32_t key = sc < scans ? s_start[sc] : 0xFFFFFFFFu;
            uint32_t lo = 0, hi = n_rst;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rst_pos[mid] < key) lo = mid + 1; else hi = mid; }
            s_first[sc] = lo;
        }
        s_scans = scans;
        if (blockIdx.x == 0) {
            sum->scan_count = (uint32_t)scans;
            sum->status = (uint32_t)status;
            sum->segment_count = n_rst + (uint32_t)scans;
            for (int sc = 0; sc < scans; sc++) { sum->scan_start[sc] = s_start[sc]; sum->scan_end[sc] = s_end[sc]; }
        }
    }
    __syncthreads();
    const int scans = s_scans;
    const uint32_t gidx = blockIdx.x * 256u + threadIdx.x;
    if (gidx >= n_rst + (uint32_t)scans || gidx >= max_segments) return;
    int sc = 0;
    while (sc + 1 < scans && gidx >= s_first[sc + 1] + (uint32_t)(sc + 1)) sc++;
    const uint32_t k = gidx - s_first[sc] - (uint32_t)sc;       // index of the segment inside its scan
    const uint32_t c_s = s_first[sc + 1] - s_first[sc];         // RSTn inside this scan
    const uint32_t from = k == 0 ? s_start[sc] : rst_pos[s_first[sc] + k - 1] + 2;
    const uint32_t to = k == c_s ? s_end[sc] : rst_pos[s_first[sc] + k];
    seg_pos[gidx] = from;
    seg_len[gidx] = to > from ? to - from : 0;
    // scan i carries component i when the stream is not interleaved (src/gpujpeg_reader.c:1345)
    const uint32_t first = g.interleaved ? 0u : (uint32_t)g.comp[sc < g.comp_count ? sc : 0].first_segment;
    const uint32_t limit = g.interleaved ? (uint32_t)g.segment_count : (uint32_t)g.comp[sc < g.comp_count ? sc : 0].segment_count;
    seg_index[gidx] = k < limit ? first + k : 0xFFFFFFFFu;
}

extern "C" int gj_hip_find_segments(const gj_geom* g, const uint8_t* d_jpeg, uint64_t begin, uint64_t size, uint32_t* d_seg_pos,
                                    uint32_t* d_seg_len, uint32_t* d_seg_index, uint32_t max_segments, uint32_t* d_scratch,
                                    gj_scan_summary* d_summary, gj_stream_t stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (size <= begin) return -1;
    const uint32_t chunks = (uint32_t)((size - begin + GJ_SCAN_CHUNK - 1) / GJ_SCAN_CHUNK);
    uint32_t* d_chunk = d_scratch;          // [chunks]
    uint32_t* d_rst = d_scratch + chunks;   // [max_segments]
    (void)hipMemsetAsync(d_summary, 0, sizeof(gj_scan_summary), st);
    hipLaunchKernelGGL(k_marker_count, dim3(chunks), dim3(256), 0, st, d_jpeg, begin, size, d_chunk, d_summary);
    hipLaunchKernelGGL(k_marker_rank, dim3(1), dim3(1024), 0, st, d_chunk, chunks, d_summary);
    hipLaunchKernelGGL(k_marker_emit, dim3(chunks), dim3(256), 0, st, d_jpeg, begin, size, d_chunk, d_rst, max_segments);
    hipLaunchKernelGGL(k_build_segments, dim3((max_segments + GJ_MAX_COMP + 255) / 256), dim3(256), 0, st, *g, d_rst, begin, size, d_summary,
                       d_seg_pos, d_seg_len, d_seg_index, max_segments + GJ_MAX_COMP);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" size_t gj_hip_find_segments_scratch_words(uint64_t begin, uint64_t size, uint32_t max_segments)
{
    return (size_t)((size - begin + GJ_SCAN_CHUNK - 1) / GJ_SCAN_CHUNK) + max_segments + 16;
}
