// gj_dec_entropy_serial.hip -- MI355X (gfx950, wave64) JPEG decoder: entropy decoding, one lane per restart segment with stream windows in LDS
// Restates src/gpujpeg_huffman_gpu_decoder.cu:135-495 (identical results to src/gpujpeg_huffman_cpu_decoder.c:245-372).
// (part of the decoder's device code, see gj_dec_internal.h for the map of the files)
#include "gj_dec_internal.h"

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
    gj_wave_sync(); // (the rows are written by all lanes and read by their owners: LDS traffic between the lanes of a wave)
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
    gj_wave_sync();
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
                                                        const int seg_count_max, const uint32_t* __restrict__ sel,
                                                        const uint16_t* __restrict__ tabs, int16_t* __restrict__ coefs)
{
    // `sel` (optional) lists the table entries to decode: the segments the sub-sequence kernel passed on
    const int seg_count = seg_count_ptr ? min((int)*seg_count_ptr, seg_count_max) : seg_count_max;
    if ((int)(blockIdx.x * 256u) >= seg_count) return;
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

    const int slot = blockIdx.x * 256 + threadIdx.x;
    const int si = slot < seg_count ? (sel ? (int)sel[slot] : slot) : 0;
    uint32_t s = 0xFFFFFFFFu;
    if (slot < seg_count) s = seg_index[si];
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


void gj_launch_huffman_serial(const gj_dec_job* job, hipStream_t st)
{
    if (job->seg_count <= 0) return;
    const gj_geom& g = job->g;
    auto kernel = g.interleaved ? k_huffman_decode<true> : k_huffman_decode<false>;
    hipLaunchKernelGGL(kernel, dim3(((unsigned)job->seg_count + 255) / 256), dim3(256), 0, st, g, job->d_jpeg, job->jpeg_size, job->d_seg_pos,
                       job->d_seg_len, job->d_seg_index, job->d_seg_count, job->seg_count, (const uint32_t*)nullptr, job->d_huff_tab, job->d_coefs);
}
