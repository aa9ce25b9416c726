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
// Entropy decoder
// ================================================================================================
struct GjBits {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc; // valid bits are left aligned
    int n;
};

__device__ __forceinline__ void gj_refill(GjBits& b)
{
    while (b.n <= 56) {
        uint32_t byte = 0; // past the end of the segment: zero bits (src/gpujpeg_huffman_cpu_decoder.c:80-118)
        if (b.p < b.end) {
            byte = *b.p++;
            if (byte == 0xFFu && b.p < b.end && *b.p == 0) b.p++; // stuffed zero
        }
        b.acc |= (uint64_t)byte << (56 - b.n);
        b.n += 8;
    }
}

__device__ __forceinline__ uint32_t gj_get_bits(GjBits& b, int n)
{
    const uint32_t v = (uint32_t)(b.acc >> (64 - n));
    b.acc <<= n;
    b.n -= n;
    return v;
}

// table words: see GJ_DEC_TAB_WORDS in gj_hip.h
__device__ __forceinline__ int gj_decode_symbol(GjBits& b, const uint16_t* t)
{
    const uint32_t peek = (uint32_t)(b.acc >> 48); // 16 bits
    const uint32_t fast = t[peek >> (16 - GJ_DEC_FAST_BITS)];
    if (fast) {
        const int len = fast >> 8;
        b.acc <<= len;
        b.n -= len;
        return (int)(fast & 0xFFu);
    }
    // codes longer than 10 bits: canonical search (ITU T.81 F.2.2.3)
    const uint16_t* maxcode = t + 1024;       // [18] as (lo, hi)
    const uint16_t* valptr = t + 1024 + 36;   // [17]
    const uint16_t* mincode = t + 1024 + 36 + 17; // [17] as (lo, hi)
    const uint16_t* vals = t + 1024 + 36 + 17 + 34;
    for (int l = GJ_DEC_FAST_BITS + 1; l <= 16; l++) {
        const int code = (int)(peek >> (16 - l));
        const int mx = (int)((uint32_t)maxcode[2 * l] | ((uint32_t)maxcode[2 * l + 1] << 16));
        if (mx >= 0 && code <= mx) {
            const int mn = (int)((uint32_t)mincode[2 * l] | ((uint32_t)mincode[2 * l + 1] << 16));
            b.acc <<= l;
            b.n -= l;
            return vals[(valptr[l] + code - mn) & 0xFF];
        }
    }
    b.acc <<= 16; // corrupt stream: consume and continue (output is undefined but in bounds)
    b.n -= 16;
    return 0;
}

__device__ __forceinline__ int gj_extend(uint32_t v, int n) // ITU T.81 F.2.2.1
{
    return v < (1u << (n - 1)) ? (int)v - (int)((1u << n) - 1u) : (int)v;
}

__global__ __launch_bounds__(64) void k_huffman_decode(const gj_geom g, const uint8_t* __restrict__ jpeg,
                                                       const uint32_t* __restrict__ seg_pos, const uint32_t* __restrict__ seg_len,
                                                       const uint32_t* __restrict__ seg_index, const int seg_count,
                                                       const uint16_t* __restrict__ tabs, int16_t* __restrict__ coefs)
{
    __shared__ uint16_t s_tab[8 * GJ_DEC_TAB_WORDS];
    // [dword][lane]: one 8x8 block per lane, every lane in its own bank. Written as int16, zeroed and read back
    // as dwords: the dword view must be may_alias or the zero stores get forwarded into the final loads.
    typedef uint32_t __attribute__((may_alias)) u32a;
    __shared__ __attribute__((aligned(16))) int16_t s_blk16[64 * 64];
    u32a* s_blk = reinterpret_cast<u32a*>(s_blk16);
    const int lane = threadIdx.x;
    for (int t = lane; t < 8 * GJ_DEC_TAB_WORDS / 2; t += 64)
        reinterpret_cast<uint32_t*>(s_tab)[t] = reinterpret_cast<const uint32_t*>(tabs)[t];
    __syncthreads();
    const int si = blockIdx.x * 64 + lane;
    if (si >= seg_count) return;
    const int s = (int)seg_index[si];
    if (s >= g.segment_count) return;
    const GjSeg sg = gj_segment(g, s);
    GjBits b;
    b.p = jpeg + seg_pos[si];
    b.end = b.p + seg_len[si];
    b.acc = 0;
    b.n = 0;
    int dc[GJ_MAX_COMP] = {0, 0, 0, 0};
    for (int k = 0; k < sg.nblocks; k++) {
        int comp, mcu_pos;
        const uint64_t off = gj_segment_block(g, sg, k, &comp, &mcu_pos);
        const uint16_t* tdc = s_tab + (g.comp[comp].dc_table * 2 + 0) * GJ_DEC_TAB_WORDS;
        const uint16_t* tac = s_tab + (g.comp[comp].ac_table * 2 + 1) * GJ_DEC_TAB_WORDS;
#pragma unroll
        for (int q = 0; q < 32; q++) s_blk[q * 64 + lane] = 0;
        gj_refill(b);
        {
            const int sz = gj_decode_symbol(b, tdc) & 15;
            int diff = 0;
            if (sz) diff = gj_extend(gj_get_bits(b, sz), sz);
            int d = (comp == 0 ? dc[0] : comp == 1 ? dc[1] : comp == 2 ? dc[2] : dc[3]) + diff;
            if (comp == 0) dc[0] = d; else if (comp == 1) dc[1] = d; else if (comp == 2) dc[2] = d; else dc[3] = d;
            s_blk16[(0 * 64 + lane) * 2] = (int16_t)d;
        }
        for (int kk = 1; kk < 64;) {
            gj_refill(b);
            const int rs = gj_decode_symbol(b, tac);
            const int run = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (run == 15) { kk += 16; continue; }
                break; // EOB
            }
            kk += run;
            if (kk > 63) break;
            const int v = gj_extend(gj_get_bits(b, sz), sz);
            const int nat = GJ_ZZ[kk];
            s_blk16[((nat >> 1) * 64 + lane) * 2 + (nat & 1)] = (int16_t)v;
            kk++;
        }
        uint4* dst = reinterpret_cast<uint4*>(coefs + off);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint4 w;
            w.x = s_blk[(r * 4 + 0) * 64 + lane];
            w.y = s_blk[(r * 4 + 1) * 64 + lane];
            w.z = s_blk[(r * 4 + 2) * 64 + lane];
            w.w = s_blk[(r * 4 + 3) * 64 + lane];
            dst[r] = w;
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
    // blocks of segments that are missing from a damaged stream must still be defined
    if (job->seg_count < g.segment_count) (void)hipMemsetAsync(job->d_coefs, 0, g.data_size * sizeof(int16_t), st);
    if (job->seg_count > 0)
        hipLaunchKernelGGL(k_huffman_decode, dim3(((unsigned)job->seg_count + 63) / 64), dim3(64), 0, st, g, job->d_jpeg, job->d_seg_pos,
                           job->d_seg_len, job->d_seg_index, job->seg_count, job->d_huff_tab, job->d_coefs);
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
// Device-side marker scan: the entropy-coded data of a scan contains 0xFF only as
// "FF 00" (stuffing) or "FF Dn" (restart marker), so every FF Dn pair is a segment boundary.
// Pass 1 counts boundaries per 1 KiB chunk, a single-workgroup scan turns the counts into ranks,
// pass 2 writes the segment table in stream order.
// ================================================================================================
#define GJ_SCAN_CHUNK 1024

__global__ __launch_bounds__(256) void k_rst_count(const uint8_t* __restrict__ jpeg, uint64_t begin, uint64_t end, uint32_t* __restrict__ chunk_count)
{
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint64_t base = begin + (uint64_t)blockIdx.x * GJ_SCAN_CHUNK;
    uint32_t n = 0;
    for (int t = threadIdx.x; t < GJ_SCAN_CHUNK; t += 256) {
        const uint64_t p = base + t;
        if (p + 1 < end && jpeg[p] == 0xFF && (jpeg[p + 1] & 0xF8) == 0xD0) n++;
    }
    if (n) atomicAdd(&s_n, n);
    __syncthreads();
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = s_n;
}

__global__ __launch_bounds__(1024) void k_rst_rank(uint32_t* __restrict__ chunk_count, uint32_t chunks, uint32_t* __restrict__ total)
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
    if (t == 0) *total = s_carry;
}

__global__ __launch_bounds__(64) void k_rst_emit(const uint8_t* __restrict__ jpeg, uint64_t begin, uint64_t end,
                                                 const uint32_t* __restrict__ chunk_rank, uint32_t* __restrict__ marker_pos, uint32_t max_markers)
{
    // one wave per chunk keeps the order: ballot + popcount of the lower lanes
    const uint64_t base = begin + (uint64_t)blockIdx.x * GJ_SCAN_CHUNK;
    uint32_t rank = chunk_rank[blockIdx.x];
    const int lane = threadIdx.x;
    for (int t0 = 0; t0 < GJ_SCAN_CHUNK; t0 += 64) {
        const uint64_t p = base + t0 + lane;
        const bool hit = p + 1 < end && jpeg[p] == 0xFF && (jpeg[p + 1] & 0xF8) == 0xD0;
        const unsigned long long m = __ballot(hit);
        if (hit) {
            const uint32_t r = rank + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (r < max_markers) marker_pos[r] = (uint32_t)p;
        }
        rank += (uint32_t)__popcll(m);
    }
}

// marker positions -> (segment start, segment length); segment i lies between marker i-1 and marker i
__global__ __launch_bounds__(256) void k_rst_segments(const uint32_t* __restrict__ marker_pos, const uint32_t* __restrict__ total,
                                                      uint64_t begin, uint64_t end, uint32_t* __restrict__ seg_pos,
                                                      uint32_t* __restrict__ seg_len, uint32_t max_segments, uint32_t* __restrict__ count)
{
    const uint32_t n = *total;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0) *count = n + 1;
    if (i > n || i >= max_segments) return;
    const uint32_t start = i == 0 ? (uint32_t)begin : marker_pos[i - 1] + 2;
    const uint32_t stop = i == n ? (uint32_t)end : marker_pos[i];
    seg_pos[i] = start;
    seg_len[i] = stop - start;
}

extern "C" int gj_hip_scan_markers(const uint8_t* d_jpeg, uint64_t begin, uint64_t end, uint32_t* d_seg_pos, uint32_t* d_seg_len,
                                   uint32_t max_segments, uint32_t* d_count, gj_stream_t stream)
{
    // scratch: chunk counters and marker positions live behind the caller's seg_len array is not possible in general,
    // so a small cached allocation per thread is used
    static thread_local uint32_t* d_scratch = nullptr;
    static thread_local size_t scratch_words = 0;
    hipStream_t st = (hipStream_t)stream;
    if (end <= begin) return -1;
    const uint32_t chunks = (uint32_t)((end - begin + GJ_SCAN_CHUNK - 1) / GJ_SCAN_CHUNK);
    const size_t need = (size_t)chunks + max_segments + 4;
    if (need > scratch_words) {
        if (d_scratch) (void)hipFree(d_scratch);
        if (hipMalloc((void**)&d_scratch, need * sizeof(uint32_t)) != hipSuccess) { d_scratch = nullptr; scratch_words = 0; return -1; }
        scratch_words = need;
    }
    uint32_t* d_chunk = d_scratch;
    uint32_t* d_total = d_scratch + chunks;
    uint32_t* d_marker = d_scratch + chunks + 4;
    hipLaunchKernelGGL(k_rst_count, dim3(chunks), dim3(256), 0, st, d_jpeg, begin, end, d_chunk);
    hipLaunchKernelGGL(k_rst_rank, dim3(1), dim3(1024), 0, st, d_chunk, chunks, d_total);
    hipLaunchKernelGGL(k_rst_emit, dim3(chunks), dim3(64), 0, st, d_jpeg, begin, end, d_chunk, d_marker, max_segments);
    hipLaunchKernelGGL(k_rst_segments, dim3((max_segments + 255) / 256), dim3(256), 0, st, d_marker, d_total, begin, end, d_seg_pos,
                       d_seg_len, max_segments, d_count);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
