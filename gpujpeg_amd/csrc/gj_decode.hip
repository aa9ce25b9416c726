// gj_decode.hip -- MI355X (gfx950, wave64) JPEG decoder kernels.
//
//   k_huffman_decode_par   sub-sequence parallel entropy decoding of batches of restart segments (the default); output either
//                          the coefficient planes or, in token mode, a dense token array + one record per block
//   k_huffman_decode       one lane per restart segment (streams whose Huffman tables do not fit the two-level tables)
//   k_idct_fused_rgb444    dequant + IDCT of the three component blocks + colour transform + packed store, from the planes
//   k_idct_tok_rgb444      the same, fed by tokens and block records
//   k_idct_fused_uyvy422 / k_idct_tok_uyvy422   packed 4:2:2 output without colour transform
//   k_idct / k_postprocess / k_copy_planes_out  generic path through padded planes
//   k_marker_count / rank / emit, k_build_segments, k_compare_header   segment table built on the device
//
// Restates src/gpujpeg_huffman_gpu_decoder.cu:135-495 (entropy decoding semantics; identical results to
// src/gpujpeg_huffman_cpu_decoder.c:245-372), src/gpujpeg_dct_gpu.cu:312-366,472-618 and
// src/gpujpeg_postprocessor.cu:49-217.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

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

// ================================================================================================
// Entropy decoder, second design: SUB-SEQUENCE PARALLEL inside every restart segment.
//
// The lane-per-segment kernel above leaves an 8K frame with 675 waves and a serial chain of several hundred symbols per
// lane. Here a workgroup takes a batch of consecutive segments and
//   1. copies their bytes into LDS with the stuffed zeros removed (one wave per segment, big-endian dwords),
//   2. cuts every segment into sub-sequences of SUB_BYTES and decodes ALL of them at once: a lane starts at the first bit
//      of its sub-sequence in the state "DC of MCU block 0 expected"; Huffman codes self-synchronise, so most lanes leave
//      their sub-sequence in the right state even though they entered it in a wrong one. Rounds: every sub-sequence whose
//      predecessor now leaves in another state than the one it was entered with goes on a work list and is decoded
//      again, densely packed onto the lanes. The first sub-sequence of a segment is always right, so this converges
//      (most sub-sequences after two rounds; at worst after as many rounds as a segment has sub-sequences),
//   3. turns the per-sub-sequence block counts into block positions with a workgroup prefix sum,
//   4. decodes once more, now storing the AC coefficients to the (pre-zeroed) coefficient planes and the DC differences
//      to an LDS array,
//   5. resolves the DC prediction there (one wave per segment, prefix sum per component) and stores the DC terms.
// The counting passes need only code lengths and zig-zag advances: one 16-bit table entry per symbol (two-level lookup,
// 10 + 6 bits) holds both. (Sub-sequence synchronisation: Klein & Wiseman 2003, Weissenberger & Schmidt 2021; the
// arrangement for short restart segments, the LDS staging and the work lists are specific to this implementation.)
// Segments that do not fit the LDS stage (longer than GJ_PAR_CAP_U bytes) are appended to a list for the
// lane-per-segment kernel. Results are identical to src/gpujpeg_huffman_gpu_decoder.cu:287-495 /
// src/gpujpeg_huffman_cpu_decoder.c:245-372.
// ================================================================================================
// LDS stage of a workgroup: bytes of unstuffed stream per group (incl. 8 B of zero padding per segment) and blocks per batch (DC and
// token-start arrays). What is resident is decided by LDS in steps of 1280 B (tools/ubench/lds_occupancy.hip: 4 workgroups per CU up to
// 40960 B, 5 up to 32000 B -- not the 32768 B the runtime's occupancy query reports). Token mode takes 4 per CU with a stage large
// enough that an 8K frame is ONE generation of workgroups (a second, partial generation doubles the kernel's duration); plane mode keeps
// the smaller stage next to its per-block address array.
#define GJ_PAR_CAP_U_FOR(tok) ((tok) ? 10752 : 8192)
#define GJ_PAR_MAX_BLOCKS_FOR(tok) ((tok) ? 2304 : 1280)
#define GJ_PAR_GMAX 64        // segments per batch
#define GJ_PAR_RESIDENT 1024  // workgroups of the token-mode kernel the GPU holds at once (256 CUs x 4)
#ifndef GJ_PAR_SUB
#define GJ_PAR_SUB 16         // bytes per sub-sequence
#endif

// table entry (gj_hip.h, GJ_DEC2_*): bits [0,5) code length + magnitude bits (0 = second level), [5,9) magnitude bits,
// [9,16) zig-zag advance. State between two symbols: bits [0,5) overshoot into the next sub-sequence, [5,11) zig-zag
// index, [11,16) block inside the MCU.
// LONG (pieces of a segment that does not fit the LDS stage): block addresses are computed, DC differences go to the plane.
// TOK (token mode): the counting passes also count the non-zero AC coefficients (upper half of nblk_out); the
// storing pass appends them as tokens (value | 2 x natural position << 16) to `tok_out` instead of scattering them into the
// planes, and notes for every block where its tokens start (s_btok, relative to the group).
// ZZ2: s_zz holds 2 x the natural position (the kernel's token-mode instantiations, also for their piecewise path through the planes)
#define GJ_TABP(tab, byte_off) (reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(tab) + (byte_off)))
template <bool WRITE, bool INTERLEAVED, bool LONG = false, bool TOK = false, bool ZZ2 = TOK>
__device__ __forceinline__ uint32_t gj_decode_sub(const uint32_t* __restrict__ U, const uint32_t start_bit, const uint32_t end_bit,
                                                  const uint32_t entry, const uint16_t* __restrict__ s_tab, const uint32_t* __restrict__ s_ptab,
                                                  const int P, const uint16_t* tdc, const uint16_t* tac, int& nblk_out,
                                                  int16_t* __restrict__ coefs, const uint32_t first, const uint32_t* __restrict__ s_blk,
                                                  int16_t* __restrict__ s_dc, int blk, const int nblocks, const uint8_t* __restrict__ s_zz,
                                                  const gj_geom* lg = nullptr, const GjSeg* lsg = nullptr, uint32_t* __restrict__ tok_out = nullptr,
                                                  uint16_t* __restrict__ s_btok = nullptr, const uint32_t tok_rel = 0, uint16_t* __restrict__ s_tend = nullptr)
{
    uint32_t bitpos = start_bit + (entry & 31u);
    int z = (int)((entry >> 5) & 63u);
    int p = (int)(entry >> 11);
    if (INTERLEAVED) {
        const uint32_t pt = s_ptab[p];
        tdc = GJ_TABP(s_tab, pt & 0xFFFFu);
        tac = GJ_TABP(s_tab, pt >> 16);
    }
    uint32_t rd = bitpos >> 5;
    uint64_t acc = (uint64_t)U[rd] << (32 + (bitpos & 31u));
    int n = 32 - (int)(bitpos & 31u);
    rd++;
    uint32_t nxt = U[rd];
    int nb = 0;
    uint32_t ntok = 0;
    uint4 tb = make_uint4(0, 0, 0, 0);
    while (bitpos < end_bit) {
        if (n <= 32) {
            acc |= (uint64_t)nxt << (32 - n);
            n += 32;
            rd++;
            nxt = U[rd];
        }
        const uint32_t hi = (uint32_t)(acc >> 32);
        const uint16_t* t = z == 0 ? tdc : tac;
        uint32_t fast; // (as the instruction: written as hi >> 22 the index becomes shift + mask + add instead of bit-field extract + shift-add)
        asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(fast) : "v"(hi), "n"(32 - GJ_DEC_FAST_BITS), "n"(GJ_DEC_FAST_BITS));
        uint32_t e = t[fast];
        if ((e & 31u) == 0) e = t[(e >> 5) + ((hi >> 16) & 63u)]; // codes longer than 10 bits
        const int tot = (int)(e & 31u);
        const int adv = (int)(e >> 9);
        if (TOK && !WRITE) ntok += (z != 0 && (e & 0x1E0u) != 0) ? 1u : 0u;
        if (WRITE && TOK) {
            const int sz = (int)((e >> 5) & 15u);
            const int used = tot - sz;
            const uint32_t bits = sz ? (hi << used) >> (32 - sz) : 0u;
            const int v = (sz && bits < (1u << (sz - 1))) ? (int)bits - (int)((1u << sz) - 1u) : (int)bits;
            if (z == 0) {
                if (blk + nb < nblocks) {
                    s_dc[blk + nb] = (int16_t)v;
                    s_btok[blk + nb] = (uint16_t)(tok_rel + ntok);
                }
            } else if (sz != 0) { // exactly the symbols the counting passes counted (s_zz[64..127] = 63: damaged streams only)
                // four tokens per 16-byte store: what a scattered store costs in the address path does not depend on its width
                const uint32_t tk = ((uint32_t)v & 0xFFFFu) | ((uint32_t)s_zz[z + adv - 1] << 16);
                const uint32_t q = ntok & 3u;
                tb.x = q == 0 ? tk : tb.x;
                tb.y = q == 1 ? tk : tb.y;
                tb.z = q == 2 ? tk : tb.z;
                tb.w = tk;
                ntok++;
                if (q == 3) *reinterpret_cast<uint4*>(tok_out + (ntok - 4u)) = tb; // (dword aligned)
            }
        } else if (WRITE) {
            const int sz = (int)((e >> 5) & 15u);
            const int used = tot - sz;
            const uint32_t bits = sz ? (hi << used) >> (32 - sz) : 0u;
            const int v = (sz && bits < (1u << (sz - 1))) ? (int)bits - (int)((1u << sz) - 1u) : (int)bits;
            const int pos = z + adv - 1;
            if (blk + nb < nblocks) {
                if (LONG) {
                    if (z == 0 || (sz != 0 && pos < 64)) {
                        int c_, m_;
                        const uint64_t off = INTERLEAVED ? gj_segment_block(*lg, *lsg, blk + nb, &c_, &m_) : (uint64_t)(first + (uint32_t)(blk + nb)) * 64;
                        coefs[off + (z == 0 ? 0 : s_zz[pos] >> (ZZ2 ? 1 : 0))] = (int16_t)v;
                    }
                } else if (z == 0) {
                    s_dc[blk + nb] = (int16_t)v;
                } else if (sz != 0 && pos < 64) {
                    const uint32_t b = INTERLEAVED ? s_blk[blk + nb] : first + (uint32_t)(blk + nb);
                    coefs[(uint64_t)b * 64 + (s_zz[pos] >> (ZZ2 ? 1 : 0))] = (int16_t)v;
                }
            }
        }
        acc <<= tot;
        n -= tot;
        bitpos += (uint32_t)tot;
        z += adv;
        if (z >= 64) {
            z = 0;
            nb++;
            if (WRITE && TOK && blk + nb == nblocks) *s_tend = (uint16_t)(tok_rel + ntok); // the last block of the segment ends here
            if (INTERLEAVED) {
                p = p + 1 == P ? 0 : p + 1;
                const uint32_t pt = s_ptab[p];
                tdc = GJ_TABP(s_tab, pt & 0xFFFFu);
                tac = GJ_TABP(s_tab, pt >> 16);
            }
        }
    }
    if (WRITE && TOK) { // the last one to three tokens
        uint32_t* o = tok_out + (ntok & ~3u);
        const uint32_t r = ntok & 3u;
        if (r > 0) o[0] = tb.x;
        if (r > 1) o[1] = tb.y;
        if (r > 2) o[2] = tb.z;
    }
    nblk_out = TOK ? (int)((uint32_t)nb | (ntok << 16)) : nb;
    return (bitpos - end_bit) | ((uint32_t)z << 5) | ((uint32_t)p << 11);
}

// Batches of the sub-sequence decoder: consecutive table entries, cut per scan -- the luminance segments of a photograph carry two to
// three times the bytes of the chrominance ones, and a batch is sized to fill the LDS stage (one batch size for the whole stream
// gave luminance batches that had to be decoded as two groups, and chrominance batches that left half of the lanes idle).
struct GjBatchPlan {
    int n;                       // ranges (scans)
    int first[GJ_MAX_COMP];      // first table entry of range c
    int count[GJ_MAX_COMP];      // entries
    int g[GJ_MAX_COMP];          // segments per batch
    int batch0[GJ_MAX_COMP + 1]; // first batch of range c; [n] = number of batches
};

template <bool INTERLEAVED, int SUB_BYTES, bool TOK>
__global__ __launch_bounds__(256, TOK ? 4 : 1) void k_huffman_decode_par(const gj_geom g, const uint8_t* __restrict__ jpeg, const uint64_t jpeg_size,
                                                            const uint32_t* __restrict__ seg_pos, const uint32_t* __restrict__ seg_len,
                                                            const uint32_t* __restrict__ seg_index, const int seg_count_max,
                                                            const uint32_t* __restrict__ seg_count_ptr, const GjBatchPlan plan,
                                                            const uint16_t* __restrict__ tabs, int16_t* __restrict__ coefs,
                                                            const int zero_fill /* 1: the planes are not known to be zero */,
                                                            uint32_t* __restrict__ d_tok /* TOK: token buffer */, const uint32_t tok_cap,
                                                            uint2* __restrict__ d_rec /* TOK: per block (coding order) token start, count << 16 | DC */)
{
    constexpr int GJ_PAR_CAP_U = GJ_PAR_CAP_U_FOR(TOK), GJ_PAR_MAX_BLOCKS = GJ_PAR_MAX_BLOCKS_FOR(TOK);
    // a group's n segments hold at most (CAP_U - 8 n) unstuffed bytes (8 B of padding each), so they are cut into at most
    // (CAP_U - 8 n) / SUB + n (SUB - 1) / SUB < CAP_U / SUB + n / 2 sub-sequences
    constexpr int MAX_SUBS = GJ_PAR_CAP_U / SUB_BYTES + GJ_PAR_GMAX / 2;
    static_assert(MAX_SUBS <= GJ_PAR_MAX_BLOCKS, "the work list lives in the DC array");
    constexpr uint32_t SUB_BITS = SUB_BYTES * 8;
    __shared__ uint32_t s_U[GJ_PAR_CAP_U / 4 + 4];
    __shared__ __attribute__((aligned(16))) uint16_t s_tab[4 * GJ_DEC2_WORDS];
    __shared__ uint8_t s_zz[64 + 64];
    __shared__ uint32_t s_ptab[GJ_MAX_MCU_BLOCKS];      // per MCU block: byte offsets of its DC | AC << 16 tables in s_tab
    __shared__ uint32_t s_pblk[GJ_MAX_MCU_BLOCKS][4];   // per MCU block: data_offset/64, blocks_x, samp_h | samp_v << 8 | bx << 16 | by << 24, comp
    // per segment of the batch
    __shared__ uint32_t s_pos[GJ_PAR_GMAX], s_len[GJ_PAR_GMAX], s_nblk[GJ_PAR_GMAX], s_first[GJ_PAR_GMAX], s_tabs[GJ_PAR_GMAX];
    __shared__ uint32_t s_bb[GJ_PAR_GMAX + 1], s_ub[GJ_PAR_GMAX + 1], s_ulen[GJ_PAR_GMAX], s_sub0[GJ_PAR_GMAX + 1];
    // per block of the batch
    __shared__ int16_t s_dc[GJ_PAR_MAX_BLOCKS];
    __shared__ uint32_t s_blk[INTERLEAVED && !TOK ? GJ_PAR_MAX_BLOCKS : 1]; // interleaved: block index in the coefficient planes
    __shared__ uint16_t s_btok[TOK ? GJ_PAR_MAX_BLOCKS : 1];                // token mode: first token of every block, relative to the group
    __shared__ uint16_t s_tend[TOK ? GJ_PAR_GMAX : 1];                      // token mode: end of the last block's tokens, per segment
    // per sub-sequence of the group
    __shared__ __attribute__((aligned(8))) uint2 s_rec[MAX_SUBS];
    __shared__ uint8_t s_subseg[MAX_SUBS];
    __shared__ uint32_t s_tmp[4];
    __shared__ int s_j1;
    __shared__ uint32_t s_nwork;
    __shared__ uint32_t s_long[GJ_PAR_GMAX]; // segments too long for the LDS stage: decoded piece by piece afterwards
    __shared__ int s_nlong;

    // LDS is what limits the residency of this kernel (measured: 3 instead of 4 workgroups per CU cost 29 %), so arrays whose lifetimes
    // do not overlap share their space: the work list of the rounds lives in the DC array (written by the storing pass), and the
    // prefix sums of the block / token counts replace the counts in the records.
    uint16_t* const s_work = reinterpret_cast<uint16_t*>(s_dc);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_nlong = 0;
    __syncthreads();
    {
        const uint4* src = reinterpret_cast<const uint4*>(tabs);
        uint4* dst = reinterpret_cast<uint4*>(s_tab);
        for (int t = tid; t < 4 * GJ_DEC2_WORDS / 8; t += 256) dst[t] = src[t];
    }
    if (tid < 128) s_zz[tid] = (uint8_t)((tid < 64 ? GJ_ZZ[tid] : 63) << (TOK ? 1 : 0)); // token mode: 2 x natural position (gj_slot_put)
    const int P = g.blocks_per_mcu;
    if (tid < GJ_MAX_MCU_BLOCKS) {
        const int pp = tid < P ? tid : 0;
        const int c = INTERLEAVED ? g.mcu_comp[pp] : 0;
        const gj_comp_geom& kc = g.comp[c];
        s_ptab[tid] = (uint32_t)((kc.dc_table * 2 + 0) * GJ_DEC2_WORDS * 2) | ((uint32_t)((kc.ac_table * 2 + 1) * GJ_DEC2_WORDS * 2) << 16);
        s_pblk[tid][0] = (uint32_t)(kc.data_offset / 64);
        s_pblk[tid][1] = (uint32_t)kc.blocks_x;
        s_pblk[tid][2] = (uint32_t)kc.samp_h | ((uint32_t)kc.samp_v << 8) | ((uint32_t)g.mcu_bx[pp] << 16) | ((uint32_t)g.mcu_by[pp] << 24);
        s_pblk[tid][3] = (uint32_t)c;
    }
    const uint32_t* end = reinterpret_cast<const uint32_t*>((reinterpret_cast<uintptr_t>(jpeg) + jpeg_size + 3) & ~(uintptr_t)3);

    // ---- batch setup: lane j describes segment j of the batch
    const int seg_count = seg_count_ptr ? min((int)*seg_count_ptr, seg_count_max) : seg_count_max;
    int pc = 0;
    while (pc + 1 < plan.n && (int)blockIdx.x >= plan.batch0[pc + 1]) pc++;
    const int G = plan.g[pc];
    const int si0 = plan.first[pc] + ((int)blockIdx.x - plan.batch0[pc]) * G;
    if (si0 >= seg_count) return;
    const int nseg = min(min(G, plan.first[pc] + plan.count[pc] - si0), seg_count - si0);
    uint32_t my_nblk = 0, my_ucap = 0;
    if (tid < GJ_PAR_GMAX) {
        uint32_t pos = 0, len = 0, nblk = 0, first = 0, tb = 0;
        if (tid < nseg) {
            const uint32_t s = seg_index[si0 + tid];
            if (s < (uint32_t)g.segment_count) {
                const GjSeg sg = gj_segment(g, (int)s);
                nblk = (uint32_t)sg.nblocks;
                pos = seg_pos[si0 + tid];
                len = seg_len[si0 + tid];
                if (INTERLEAVED) {
                    first = (uint32_t)sg.mcu_first; // first MCU
                } else {
                    const gj_comp_geom& kc = g.comp[sg.comp];
                    first = (uint32_t)(kc.data_offset / 64) + (uint32_t)sg.mcu_first; // first block in the coefficient plane
                    tb = (uint32_t)((kc.dc_table * 2 + 0) * GJ_DEC2_WORDS * 2) | ((uint32_t)((kc.ac_table * 2 + 1) * GJ_DEC2_WORDS * 2) << 16);
                }
                if (((len + 3u) & ~3u) + 8u > (uint32_t)GJ_PAR_CAP_U || nblk > (uint32_t)GJ_PAR_MAX_BLOCKS) { // too long for the LDS stage / the per-block arrays: in pieces at the end
                    s_long[atomicAdd(&s_nlong, 1)] = (uint32_t)tid;
                    len = 0;
                    nblk = 0;
                }
            }
        }
        s_pos[tid] = pos;
        s_len[tid] = len;
        s_nblk[tid] = nblk;
        s_first[tid] = first;
        s_tabs[tid] = tb;
        my_nblk = nblk;
        my_ucap = len ? ((len + 3u) & ~3u) + 8u : 0u;
    }
    {
        uint32_t tot;
        const uint32_t a = gj_wg256_incl_scan(my_nblk, s_tmp, &tot);
        if (tid < GJ_PAR_GMAX) s_bb[tid + 1] = a;
        const uint32_t b = gj_wg256_incl_scan(my_ucap, s_tmp, &tot);
        if (tid < GJ_PAR_GMAX) s_ub[tid + 1] = b;
        if (tid == 0) { s_bb[0] = 0; s_ub[0] = 0; }
    }
    __syncthreads();
    const int nblocks_batch = (int)s_bb[nseg];
    if (INTERLEAVED && !TOK) { // where every block of the batch lives in the coefficient planes
        for (int t = tid; t < nblocks_batch; t += 256) {
            int lo = 0, hi = nseg;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_bb[mid] <= (uint32_t)t) lo = mid; else hi = mid;
            }
            const uint32_t kb = (uint32_t)t - s_bb[lo];
            const uint32_t mi = kb / (uint32_t)P, p = kb - mi * (uint32_t)P;
            const uint32_t m = s_first[lo] + mi;
            const uint32_t my = m / (uint32_t)g.mcu_count_x, mx = m - my * (uint32_t)g.mcu_count_x;
            const uint32_t q = s_pblk[p][2];
            const uint32_t bx = mx * (q & 0xFFu) + ((q >> 16) & 0xFFu), by = my * ((q >> 8) & 0xFFu) + (q >> 24);
            s_blk[t] = s_pblk[p][0] + by * s_pblk[p][1] + bx;
        }
    }
    // ---- every block of the batch is filled with zeros (fully coalesced 16 B stores, 128 B per block) before its non-zero
    //      coefficients are scattered into it: the planes need no clearing between frames, and the scattered stores land in
    //      lines this workgroup has just put into L2 instead of pulling the whole plane through partial-line write-backs
    if (INTERLEAVED && !TOK) __syncthreads(); // s_blk is complete
    if (zero_fill && !TOK) {
        for (int j = wave; j < nseg; j += 4) {
            const uint32_t chunks = s_nblk[j] * 8u;
            for (uint32_t c = (uint32_t)lane; c < chunks; c += 64) {
                const uint32_t b = INTERLEAVED ? s_blk[s_bb[j] + (c >> 3)] : s_first[j] + (c >> 3);
                reinterpret_cast<uint4*>(coefs + (uint64_t)b * 64)[c & 7u] = make_uint4(0, 0, 0, 0);
            }
        }
        __syncthreads(); // (orders the zeros before the coefficient stores of the other lanes)
    }

    // (rounds and block positions are shared by the groups of whole segments and by the pieces of long segments)
    auto run_rounds = [&](const int nsub, const uint32_t ub0) {
    // rounds. s_rec[k] = (entry state | exit state << 16, blocks completed) is written with one 64-bit LDS store, so a
    //       record always describes one decoding of sub-sequence k, whoever wrote it last.
    int nwork = nsub;
    for (int round = 0; nwork > 0; round++) {
        for (int w = tid; w < nwork; w += 256) {
            const int k = s_work[w];
            const int j = s_subseg[k];
            const uint32_t k_first = s_sub0[j];
            const uint32_t tb = s_tabs[j];
            const uint32_t* U = s_U + ((s_ub[j] - ub0) >> 2);
            const uint32_t seg_bits = s_ulen[j] * 8u;
            // round 0: the assumed entry state; later: what the predecessor leaves now
            const uint32_t e = round == 0 ? (s_rec[k].x & 0xFFFFu) : (s_rec[k - 1].x >> 16);
            const uint32_t i = (uint32_t)k - k_first;
            int nb;
            const uint32_t x = gj_decode_sub<false, INTERLEAVED, false, TOK>(U, i * SUB_BITS, min((i + 1) * SUB_BITS, seg_bits), e, s_tab, s_ptab, P,
                                                                 GJ_TABP(s_tab, tb & 0xFFFFu), GJ_TABP(s_tab, tb >> 16), nb, nullptr, 0, nullptr, nullptr, 0, 0, s_zz);
            s_rec[k] = make_uint2(e | (x << 16), (uint32_t)nb);
        }
        __syncthreads();
        // next work list: sub-sequences whose predecessor leaves in another state than they were entered with (measured: walking
        // down runs of them with one lane, or seeding interleaved scans with one hypothesis per MCU block, costs more than it saves)
        for (int k0 = 0; k0 < nsub; k0 += 256) {
            const int k = k0 + tid;
            bool cand = false;
            if (k < nsub) {
                const uint32_t first = s_sub0[s_subseg[k]];
                cand = (uint32_t)k != first && (s_rec[k - 1].x >> 16) != (s_rec[k].x & 0xFFFFu);
            }
            const unsigned long long m = __ballot(cand);
            uint32_t base = 0;
            if (lane == 0 && m) base = atomicAdd(&s_nwork, (uint32_t)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (cand) s_work[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)k;
        }
        __syncthreads();
        nwork = (int)s_nwork;
        __syncthreads();
        if (tid == 0) s_nwork = 0;
    }

    };
    auto block_positions = [&](const int nsub) {
    //  block position of every sub-sequence inside its segment (inclusive scan, segment start subtracted below)
    {
        uint32_t carry = 0;
        for (int k0 = 0; k0 < nsub; k0 += 256) {
            const int k = k0 + tid;
            const uint32_t v = k < nsub ? s_rec[k].y : 0;
            uint32_t tot;
            const uint32_t inc = gj_wg256_incl_scan(v, s_tmp, &tot);
            __syncthreads();
            if (k < nsub) s_rec[k].y = carry + inc;
            carry += tot;
        }
    }
    __syncthreads();

    };
    // ---- groups of segments whose unstuffed bytes fit the LDS stage (normally one group)
    for (int j0 = 0; j0 < nseg;) {
        if (tid == 0) { s_j1 = j0 + 1; s_nwork = 0; }
        __syncthreads();
        if (tid > j0 && tid <= nseg && s_ub[tid] - s_ub[j0] <= (uint32_t)GJ_PAR_CAP_U) atomicMax(&s_j1, tid);
        __syncthreads();
        const int j1 = s_j1;
        const uint32_t ub0 = s_ub[j0];

        // -- 1. unstuffed copy, one wave per segment. The first 256 B of all segments of this wave are fetched up front, so
        //       that the wave waits for HBM once and not once per segment.
        {
            uint8_t* U8 = reinterpret_cast<uint8_t*>(s_U);
            // (with at most 8 segments per wave -- batches of long segments -- the upper half of the prefetch registers takes the second
            //  256 B of every segment instead of further segments: a dependent load per segment cost the luminance batches of an 8K
            //  frame 50 us)
            const bool two = G <= GJ_PAR_GMAX / 2;
            uint32_t wpre[GJ_PAR_GMAX / 4];
#pragma unroll
            for (int q = 0; q < GJ_PAR_GMAX / 4; q++) {
                const bool second = two && q >= GJ_PAR_GMAX / 8;
                const int j = j0 + wave + 4 * (second ? q - GJ_PAR_GMAX / 8 : q);
                wpre[q] = 0;
                if (j < j1 && s_len[j]) {
                    const uintptr_t a = reinterpret_cast<uintptr_t>(jpeg) + s_pos[j];
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
                    const uint32_t ndw = ((uint32_t)(a & 3) + s_len[j] + 3u) >> 2;
                    const uint32_t idx = (uint32_t)lane + (second ? 64u : 0u);
                    if (idx < ndw && src + idx < end) wpre[q] = src[idx];
                }
            }
#pragma unroll
            for (int q = 0; q < GJ_PAR_GMAX / 4; q++) {
                const int j = j0 + wave + 4 * q;
                if (j >= j1 || (two && q >= GJ_PAR_GMAX / 8)) break;
                const uint32_t len = s_len[j];
                const uint32_t ubase = s_ub[j] - ub0;
                uint32_t out = 0;
                if (len) {
                    const uintptr_t a = reinterpret_cast<uintptr_t>(jpeg) + s_pos[j];
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
                    const int lead = (int)(a & 3);
                    const uint32_t ndw = ((uint32_t)lead + len + 3u) >> 2;
                    uint32_t carry = 0;
                    // A segment that lies in the prefetch registers and has no stuffed byte (two thirds of the chrominance, one third of
                    // the luminance segments of a photograph at q75) is a plain copy: the lanes' dwords shifted by the start's misalignment and byte-swapped into the stage's
                    // big-endian dwords. 0xFF00 is looked for in all four bytes at once (zero bytes of w under 0xFF bytes of the
                    // stream shifted by one; a borrow can only produce a false alarm, which takes the general path below).
                    bool copied = false;
                    if (ndw <= 64u || (two && ndw <= 128u)) { // (everything the prefetch registers hold)
                        const bool far = ndw > 64u;
                        const uint32_t w0 = wpre[q], w1 = far ? wpre[(q + GJ_PAR_GMAX / 8) % (GJ_PAR_GMAX / 4)] : 0u;
                        uint32_t pw0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w0, 0x138, 0xF, 0xF, false); // wave_shr:1 (lane 0: nothing in front)
                        uint32_t pw1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w1, 0x138, 0xF, 0xF, false);
                        if (lane == 0) pw1 = (uint32_t)__builtin_amdgcn_readlane((int)w0, 63);
                        const uint32_t pb0 = __builtin_amdgcn_alignbit(w0, pw0, 24), pb1 = __builtin_amdgcn_alignbit(w1, pw1, 24); // the stream one byte earlier
                        const uint32_t hit0 = (w0 - 0x01010101u) & ~w0 & (~pb0 - 0x01010101u) & pb0 & 0x80808080u;
                        const uint32_t hit1 = (w1 - 0x01010101u) & ~w1 & (~pb1 - 0x01010101u) & pb1 & 0x80808080u;
                        if (__ballot((hit0 != 0u && (uint32_t)lane < ndw) || (hit1 != 0u && (uint32_t)lane + 64u < ndw)) == 0ull) {
                            uint32_t wn0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w0, 0x130, 0xF, 0xF, false); // wave_shl:1
                            const uint32_t wn1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w1, 0x130, 0xF, 0xF, false); // (lane 63: zero, nothing behind)
                            if (lane == 63) wn0 = (uint32_t)__builtin_amdgcn_readlane((int)w1, 0);
                            uint32_t d0 = __builtin_bswap32(__builtin_amdgcn_alignbyte(wn0, w0, (uint32_t)lead));
                            uint32_t d1 = __builtin_bswap32(__builtin_amdgcn_alignbyte(wn1, w1, (uint32_t)lead));
                            const uint32_t full = len >> 2, rest = len & 3u, cut = 0xFFFFFFFFu << (32u - 8u * rest); // (the bytes behind the end are zero padding)
                            const uint32_t nout = full + (rest ? 1u : 0u), m0 = (uint32_t)lane, m1 = m0 + 64u;
                            if (m0 == full && rest) d0 &= cut;
                            if (m1 == full && rest) d1 &= cut;
                            if (m0 < nout) s_U[(ubase >> 2) + m0] = d0;
                            if (far && m1 < nout) s_U[(ubase >> 2) + m1] = d1;
                            out = len;
                            copied = true;
                        }
                    }
                    for (uint32_t c0 = 0; !copied && c0 < ndw; c0 += 64) {
                        const uint32_t idx = c0 + (uint32_t)lane;
                        uint32_t w = wpre[q];
                        if (c0 == 64 && two) {
                            w = wpre[(q + GJ_PAR_GMAX / 8) % (GJ_PAR_GMAX / 4)];
                        } else if (c0) {
                            w = 0;
                            if (idx < ndw && src + idx < end) w = src[idx];
                        }
                        uint32_t pw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x138, 0xF, 0xF, false); // wave_shr:1
                        if (lane == 0) pw = carry;
                        uint32_t prev = pw >> 24;
                        uint32_t keep = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t b = (w >> (8 * k)) & 0xFFu;
                            const int off = (int)(idx * 4u) + k - lead;
                            const bool valid = off >= 0 && off < (int)len;
                            const bool stuffed = b == 0 && prev == 0xFFu && off > 0;
                            if (valid && !stuffed) keep |= 1u << k;
                            prev = b;
                        }
                        const uint32_t cnt = (uint32_t)__popc(keep);
                        const uint32_t inc = gj_wave_incl_scan(cnt);
                        uint32_t o = ubase + out + inc - cnt;
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (keep & (1u << k)) { U8[o ^ 3u] = (uint8_t)(w >> (8 * k)); o++; }
                        out += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                        carry = (uint32_t)__builtin_amdgcn_readlane((int)w, 63);
                    }
                    for (uint32_t b = out + (uint32_t)lane; b < ((out + 3u) & ~3u) + 8u; b += 64) U8[(ubase + b) ^ 3u] = 0;
                }
                if (lane == 0) s_ulen[j] = out;
            }
        }
        __syncthreads();

        // -- 2. sub-sequence table
        uint32_t my_nsub = 0;
        if (tid >= j0 && tid < j1 && s_nblk[tid]) my_nsub = (s_ulen[tid] + SUB_BYTES - 1) / SUB_BYTES;
        {
            uint32_t tot;
            const uint32_t a = gj_wg256_incl_scan(my_nsub, s_tmp, &tot);
            if (tid >= j0 && tid < j1) s_sub0[tid + 1] = a;
            if (tid == 0) s_sub0[j0] = 0;
        }
        __syncthreads();
        const int nsub = (int)s_sub0[j1];
        for (int k = tid; k < nsub; k += 256) {
            int lo = j0, hi = j1; // segment j with s_sub0[j] <= k < s_sub0[j + 1]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_sub0[mid] <= (uint32_t)k) lo = mid; else hi = mid;
            }
            s_subseg[k] = (uint8_t)lo;
            // assumed entry state: the first sub-sequence starts a block; any other one most likely starts in the middle of one (AC table)
            s_rec[k] = make_uint2((uint32_t)k == s_sub0[lo] ? 0u : (1u << 5), 0u);
            s_work[k] = (uint16_t)k; // round 0: everybody
        }
        __syncthreads();

        // -- 3. rounds, 4. block positions
        run_rounds(nsub, ub0);
        block_positions(nsub);

        // -- 5. decode once more, now storing the coefficients
        // token mode: the group's tokens form one dense run (lanes write their sub-sequences' tokens back to back). It starts at
        // 4 x the byte offset of the group's first segment: a non-zero AC coefficient takes at least 2 bits of the stream, so the
        // runs of different groups cannot overlap, and no allocator or reset is needed between frames.
        uint32_t gbase = 0;
        if (TOK) {
            const uint32_t T = nsub > 0 ? s_rec[nsub - 1].y >> 16 : 0u;
            int jb = j0;
            while (jb + 1 < j1 && s_len[jb] == 0) jb++; // (segments without data carry no position)
            gbase = 4u * s_pos[jb];
            if (gbase > tok_cap || T > tok_cap - gbase) gbase = 0xFFFFFFFFu; // (cannot happen with the capacity the host allocates)
            for (uint32_t b = s_bb[j0] + (uint32_t)tid; b < s_bb[j1]; b += 256) { s_btok[b] = 0xFFFFu; s_dc[b] = 0; } // "block not seen"
            if (tid >= j0 && tid < j1) s_tend[tid] = 0xFFFFu;
            __syncthreads();
        } else {
            for (uint32_t b = s_bb[j0] + (uint32_t)tid; b < s_bb[j1]; b += 256) s_dc[b] = 0; // (blocks a damaged segment never reaches)
            __syncthreads();
        }
        for (int k = tid; k < nsub; k += 256) {
            const int j = s_subseg[k];
            const uint32_t k_first = s_sub0[j];
            const uint32_t i = (uint32_t)k - k_first;
            const uint32_t sc_k = k > 0 ? s_rec[k - 1].y : 0u, sc_f = k_first > 0 ? s_rec[k_first - 1].y : 0u;
            const uint32_t before = TOK ? (sc_k & 0xFFFFu) - (sc_f & 0xFFFFu) : sc_k - sc_f;
            const uint32_t endb = min((i + 1) * SUB_BITS, s_ulen[j] * 8u);
            const uint32_t tb = s_tabs[j];
            int nb;
            if (TOK) {
                if (gbase != 0xFFFFFFFFu)
                    gj_decode_sub<true, INTERLEAVED, false, true>(s_U + ((s_ub[j] - ub0) >> 2), i * SUB_BITS, endb, s_rec[k].x & 0xFFFFu, s_tab, s_ptab, P,
                                                                  GJ_TABP(s_tab, tb & 0xFFFFu), GJ_TABP(s_tab, tb >> 16), nb, nullptr, 0, nullptr, s_dc + s_bb[j], (int)before,
                                                                  (int)s_nblk[j], s_zz, nullptr, nullptr, d_tok + gbase + (sc_k >> 16), s_btok + s_bb[j],
                                                                  sc_k >> 16, s_tend + j);
            } else {
                gj_decode_sub<true, INTERLEAVED>(s_U + ((s_ub[j] - ub0) >> 2), i * SUB_BITS, endb, s_rec[k].x & 0xFFFFu, s_tab, s_ptab, P, GJ_TABP(s_tab, tb & 0xFFFFu),
                                                 GJ_TABP(s_tab, tb >> 16), nb, coefs, s_first[j], s_blk + s_bb[j], s_dc + s_bb[j], (int)before, (int)s_nblk[j], s_zz);
            }
        }
        __syncthreads();

        // -- 6. DC prediction: one wave per segment, prefix sum per component, DC terms to HBM
        for (int j = j0 + wave; j < j1; j += 4) {
            const int nblk = (int)s_nblk[j];
            const uint32_t bb = s_bb[j];
            int carry[GJ_MAX_COMP] = {0, 0, 0, 0};
            for (int k0 = 0; k0 < nblk; k0 += 64) {
                const int kb = k0 + lane;
                const int d = kb < nblk ? (int)s_dc[bb + kb] : 0;
                int comp = 0;
                if (INTERLEAVED) comp = (int)s_pblk[(uint32_t)kb % (uint32_t)P][3];
                int dc = 0;
#pragma unroll
                for (int c = 0; c < GJ_MAX_COMP; c++) {
                    if (c >= (INTERLEAVED ? g.comp_count : 1)) break;
                    const uint32_t inc = gj_wave_incl_scan((uint32_t)((!INTERLEAVED || comp == c) ? d : 0));
                    if (!INTERLEAVED || comp == c) dc = carry[c] + (int)inc;
                    carry[c] += (int)(uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                }
                if (kb < nblk) {
                    if (TOK) { // block record in coding order: where the tokens are, how many, the DC term
                        const uint32_t k_end = s_sub0[j + 1];
                        const uint32_t seg_end = k_end > s_sub0[j] ? s_rec[k_end - 1].y >> 16 : 0u; // tokens of the group up to the end of this segment
                        const uint32_t t0 = s_btok[bb + kb];
                        const uint32_t t1 = (kb + 1 < nblk && s_btok[bb + kb + 1] != 0xFFFFu) ? s_btok[bb + kb + 1]
                                            : (kb + 1 == nblk && s_tend[j] != 0xFFFFu)        ? s_tend[j]
                                                                                              : seg_end;
                        const bool seen = t0 != 0xFFFFu && gbase != 0xFFFFFFFFu;
                        const uint32_t cnt = seen && t1 >= t0 ? min(t1 - t0, 63u) : 0u;
                        const uint32_t r = (INTERLEAVED ? s_first[j] * (uint32_t)P : s_first[j]) + (uint32_t)kb;
                        d_rec[r] = make_uint2(seen ? gbase + t0 : 0u, (cnt << 16) | ((uint32_t)dc & 0xFFFFu));
                    } else {
                        const uint32_t b = INTERLEAVED ? s_blk[bb + kb] : s_first[j] + (uint32_t)kb;
                        coefs[(uint64_t)b * 64] = (int16_t)dc;
                    }
                }
            }
        }
        __syncthreads();
        j0 = j1;
    }
    // ---- segments longer than the LDS stage (restart interval 0 or very large, noise at q100): piece after piece. A piece is
    //      GJ_PAR_PIECE stuffed bytes; it is unstuffed by the whole workgroup, cut into sub-sequences and synchronised like a
    //      segment, except that its first sub-sequence is entered in the state the previous piece was left in. The block count and
    //      the DC predictors are carried along; the DC differences go to the plane and are summed up there, 256 blocks at a time.
    constexpr uint32_t GJ_PAR_PIECE = GJ_PAR_CAP_U - 64;
    const int nlong = s_nlong;
    for (int li = 0; li < nlong; li++) {
        const int jl = (int)s_long[li];
        const GjSeg sg = gj_segment(g, (int)seg_index[si0 + jl]);
        const uint8_t* base = jpeg + seg_pos[si0 + jl];
        const uint32_t len = seg_len[si0 + jl];
        const uint32_t first = s_first[jl];
        if (TOK) // token mode: the blocks of a long segment live in the coefficient planes; their records say so (count 0xFFFF)
            for (uint32_t c = (uint32_t)tid; c < (uint32_t)sg.nblocks; c += 256) d_rec[sg.first_block + c] = make_uint2(0u, 0xFFFF0000u);
        if (zero_fill || TOK) {
            for (uint32_t c = (uint32_t)tid; c < (uint32_t)sg.nblocks * 8u; c += 256) {
                int c_, m_;
                const uint64_t off = INTERLEAVED ? gj_segment_block(g, sg, (int)(c >> 3), &c_, &m_) : (uint64_t)(first + (c >> 3)) * 64;
                reinterpret_cast<uint4*>(coefs + off)[c & 7u] = make_uint4(0, 0, 0, 0);
            }
        }
        uint32_t src_off = 0, entry = 0, blocks_done = 0;
        int dc_carry[GJ_MAX_COMP] = {0, 0, 0, 0};
        while (src_off < len) {
            __syncthreads();
            // -- unstuff [src_off, src_off + chunk) plus up to 16 bytes of look-ahead for the symbol that straddles the piece end
            const uint32_t chunk = min(GJ_PAR_PIECE, len - src_off);
            const uint32_t look = min(16u, len - src_off - chunk);
            const uintptr_t a = reinterpret_cast<uintptr_t>(base) + src_off;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
            const int lead = (int)(a & 3);
            const uint32_t ndw = ((uint32_t)lead + chunk + look + 3u) >> 2;
            uint8_t* U8 = reinterpret_cast<uint8_t*>(s_U);
            uint32_t out = 0, ulen = 0; // bytes written so far; those belonging to the piece proper
            for (uint32_t d0 = 0; d0 < ndw; d0 += 256) {
                const uint32_t idx = d0 + (uint32_t)tid;
                uint32_t w = 0;
                if (idx < ndw && src + idx < end) w = src[idx];
                uint32_t keep = 0, keep_piece = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int off = (int)(idx * 4u) + k - lead; // offset inside [src_off, ...)
                    const bool valid = idx < ndw && off >= 0 && off < (int)(chunk + look);
                    const uint32_t b = (w >> (8 * k)) & 0xFFu;
                    // the byte before (in the stuffed stream): a zero after 0xFF is stuffing; the first byte of a segment never is
                    uint32_t prev = 0;
                    if (valid && src_off + (uint32_t)off > 0) prev = k > 0 ? (w >> (8 * k - 8)) & 0xFFu : (src + idx <= end ? base[src_off + (uint32_t)off - 1] : 0u); // never past the buffer
                    if (valid && !(b == 0 && prev == 0xFFu)) {
                        keep |= 1u << k;
                        if (off < (int)chunk) keep_piece |= 1u << k;
                    }
                }
                const uint32_t cnt = (uint32_t)__popc(keep);
                uint32_t tot;
                const uint32_t inc = gj_wg256_incl_scan(cnt | ((uint32_t)__popc(keep_piece) << 16), s_tmp, &tot); // two 16-bit sums in one scan
                uint32_t o = out + (inc & 0xFFFFu) - cnt;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (keep & (1u << k)) { U8[o ^ 3u] = (uint8_t)(w >> (8 * k)); o++; }
                out += tot & 0xFFFFu;
                ulen += tot >> 16;
            }
            for (uint32_t b = out + (uint32_t)tid; b < ((out + 3u) & ~3u) + 8u; b += 256) U8[b ^ 3u] = 0;
            // -- the piece as one pseudo segment in slot jl
            const int nsub = (int)((ulen + SUB_BYTES - 1) / SUB_BYTES);
            if (tid == 0) { s_sub0[jl] = 0; s_sub0[jl + 1] = (uint32_t)nsub; s_ulen[jl] = ulen; s_ub[jl] = 0; s_nwork = 0; }
            for (int k = tid; k < nsub; k += 256) {
                s_subseg[k] = (uint8_t)jl;
                s_rec[k] = make_uint2(k == 0 ? entry : (1u << 5), 0u);
                s_work[k] = (uint16_t)k;
            }
            __syncthreads();
            run_rounds(nsub, 0u);
            block_positions(nsub);
            // -- coefficients of this piece (DC still as differences)
            const uint32_t tb = s_tabs[jl];
            for (int k = tid; k < nsub; k += 256) {
                const uint32_t before = k > 0 ? (TOK ? s_rec[k - 1].y & 0xFFFFu : s_rec[k - 1].y) : 0u;
                const uint32_t endb = min((uint32_t)(k + 1) * SUB_BITS, ulen * 8u);
                int nb;
                gj_decode_sub<true, INTERLEAVED, true, false, TOK>(s_U, (uint32_t)k * SUB_BITS, endb, s_rec[k].x & 0xFFFFu, s_tab, s_ptab, P, GJ_TABP(s_tab, tb & 0xFFFFu),
                                                       GJ_TABP(s_tab, tb >> 16), nb, coefs, first, nullptr, nullptr, (int)(blocks_done + before), sg.nblocks, s_zz,
                                                       &g, &sg);
            }
            __syncthreads(); // (workgroup-scope fence: the differences are visible to the lanes that sum them up)
            const uint32_t piece_blocks = nsub > 0 ? (TOK ? s_rec[nsub - 1].y & 0xFFFFu : s_rec[nsub - 1].y) : 0u;
            const uint32_t b1 = min(blocks_done + piece_blocks, (uint32_t)sg.nblocks);
            for (uint32_t k0 = blocks_done; k0 < b1; k0 += 256) {
                const uint32_t k = k0 + (uint32_t)tid;
                const bool valid = k < b1;
                int comp = 0, m_ = 0;
                uint64_t off = 0;
                if (valid) off = INTERLEAVED ? gj_segment_block(g, sg, (int)k, &comp, &m_) : (uint64_t)(first + k) * 64;
                const int d = valid ? (int)coefs[off] : 0;
#pragma unroll
                for (int c = 0; c < GJ_MAX_COMP; c++) {
                    if (c >= (INTERLEAVED ? g.comp_count : 1)) break;
                    const bool mine = valid && (!INTERLEAVED || comp == c);
                    uint32_t tot;
                    const uint32_t inc = gj_wg256_incl_scan((uint32_t)(mine ? d : 0), s_tmp, &tot);
                    if (mine) coefs[off] = (int16_t)(dc_carry[c] + (int)inc);
                    dc_carry[c] += (int)tot;
                }
            }
            __syncthreads();
            if (nsub > 0) entry = s_rec[nsub - 1].x >> 16;
            blocks_done += piece_blocks;
            src_off += chunk;
        }
    }
}

// ================================================================================================
// Entropy decoder, third design: ONE LANE PER RESTART SEGMENT over an LDS stage, for interleaved scans with many short segments
// (BASELINE config 4: 172 800 segments of 250 B).
//
// The sub-sequence decoder lives on self-synchronisation. In an interleaved scan a lane that enters a sub-sequence in the wrong block of
// the MCU decodes with the wrong tables and falls into step only by accident: measured on config 4, 6.2 rounds per batch, i.e. the
// correct decoding advances by about one sub-sequence per round and segment -- every symbol is decoded seven times, and a workgroup
// spends 61 of its 99 us in rounds. With this many segments there is enough parallelism without cutting them: a workgroup unstuffs 100
// to 128 segments into LDS (one wave per segment, as above) and then every lane decodes its own segment once, from the first bit, in the
// known state: no counting passes, no rounds, DC prediction in registers, coefficients straight to the (zero-filled) planes.
// A segment that does not fit the stage raises `overflow` and is left alone: the host then decodes the frame with the sub-sequence
// kernel (it knows the longest segment of a stream before the launch, except on the speculative path, where it finds the flag afterwards).
// Results are identical to the other two kernels (tests run all three on the same streams).
// ================================================================================================
#define GJ_SEQ_STAGE 26112 // bytes of unstuffed stream per group (incl. 8 B of zero padding per segment)
#define GJ_SEQ_NS 128      // segments per workgroup

// unstuffs one segment into the stage with one wave; w0 = the lane's dword of the segment's first 256 B (zero behind its end)
__device__ __forceinline__ uint32_t gj_unstuff_segment(const uint8_t* __restrict__ jpeg, const uint32_t* __restrict__ end, const uint32_t pos, const uint32_t len,
                                                       uint32_t* __restrict__ stage, const uint32_t ubase, const int lane, const uint32_t w0)
{
    uint8_t* U8 = reinterpret_cast<uint8_t*>(stage);
    const uintptr_t a = reinterpret_cast<uintptr_t>(jpeg) + pos;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
    const int lead = (int)(a & 3);
    const uint32_t ndw = ((uint32_t)lead + len + 3u) >> 2;
    uint32_t out = 0;
    bool copied = false;
    if (ndw <= 64u) { // no stuffed byte: a shifted, byte-swapped copy (see k_huffman_decode_par)
        const uint32_t pw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w0, 0x138, 0xF, 0xF, false);
        const uint32_t pb = __builtin_amdgcn_alignbit(w0, pw, 24);
        const uint32_t hit = (w0 - 0x01010101u) & ~w0 & (~pb - 0x01010101u) & pb & 0x80808080u;
        if (__ballot(hit != 0u && (uint32_t)lane < ndw) == 0ull) {
            const uint32_t wn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w0, 0x130, 0xF, 0xF, false);
            uint32_t d = __builtin_bswap32(__builtin_amdgcn_alignbyte(wn, w0, (uint32_t)lead));
            const uint32_t full = len >> 2, rest = len & 3u;
            if ((uint32_t)lane == full && rest) d &= 0xFFFFFFFFu << (32u - 8u * rest);
            if ((uint32_t)lane < full + (rest ? 1u : 0u)) stage[(ubase >> 2) + (uint32_t)lane] = d;
            out = len;
            copied = true;
        }
    }
    uint32_t carry = 0;
    for (uint32_t c0 = 0; !copied && c0 < ndw; c0 += 64) {
        const uint32_t idx = c0 + (uint32_t)lane;
        uint32_t w = w0;
        if (c0) {
            w = 0;
            if (idx < ndw && src + idx < end) w = src[idx];
        }
        uint32_t pw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x138, 0xF, 0xF, false); // wave_shr:1
        if (lane == 0) pw = carry;
        uint32_t prev = pw >> 24;
        uint32_t keep = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t b = (w >> (8 * k)) & 0xFFu;
            const int off = (int)(idx * 4u) + k - lead;
            const bool valid = off >= 0 && off < (int)len;
            const bool stuffed = b == 0 && prev == 0xFFu && off > 0;
            if (valid && !stuffed) keep |= 1u << k;
            prev = b;
        }
        const uint32_t cnt = (uint32_t)__popc(keep);
        const uint32_t inc = gj_wave_incl_scan(cnt);
        uint32_t o = ubase + out + inc - cnt;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (keep & (1u << k)) { U8[o ^ 3u] = (uint8_t)(w >> (8 * k)); o++; }
        out += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        carry = (uint32_t)__builtin_amdgcn_readlane((int)w, 63);
    }
    for (uint32_t b = out + (uint32_t)lane; b < ((out + 3u) & ~3u) + 8u; b += 64) U8[(ubase + b) ^ 3u] = 0;
    return out;
}

template <bool INTERLEAVED>
__global__ __launch_bounds__(256, 4) void k_huffman_decode_seq(const gj_geom g, const uint8_t* __restrict__ jpeg, const uint64_t jpeg_size,
                                                               const uint32_t* __restrict__ seg_pos, const uint32_t* __restrict__ seg_len,
                                                               const uint32_t* __restrict__ seg_index, const int seg_count_max,
                                                               const uint32_t* __restrict__ seg_count_ptr, const int NS,
                                                               const uint16_t* __restrict__ tabs, int16_t* __restrict__ coefs, const int zero_fill,
                                                               uint32_t* __restrict__ overflow)
{
    __shared__ uint32_t s_U[GJ_SEQ_STAGE / 4 + 4];
    __shared__ __attribute__((aligned(16))) uint16_t s_tab[4 * GJ_DEC2_WORDS];
    __shared__ uint8_t s_zz[64 + 64];
    __shared__ uint32_t s_pos[GJ_SEQ_NS], s_len[GJ_SEQ_NS], s_idx[GJ_SEQ_NS], s_ub[GJ_SEQ_NS + 1], s_ulen[GJ_SEQ_NS];
    __shared__ uint32_t s_tmp[4];
    __shared__ int s_j1;
    __shared__ uint32_t s_ptab[GJ_MAX_MCU_BLOCKS];    // per MCU block: word offsets of its DC | AC << 16 tables in s_tab
    __shared__ uint32_t s_pblk[GJ_MAX_MCU_BLOCKS][4]; // per MCU block: data_offset / 64, blocks_x, samp_h | samp_v << 8 | bx << 16 | by << 24, comp
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < GJ_MAX_MCU_BLOCKS) { // (the geometry is a kernel argument: indexing it by the MCU block in the symbol loop would be loads from memory)
        const int pp = tid < g.blocks_per_mcu ? tid : 0;
        const int c = INTERLEAVED ? g.mcu_comp[pp] : 0;
        const gj_comp_geom& kc = g.comp[c];
        s_ptab[tid] = (uint32_t)((kc.dc_table * 2 + 0) * GJ_DEC2_WORDS) | ((uint32_t)((kc.ac_table * 2 + 1) * GJ_DEC2_WORDS) << 16);
        s_pblk[tid][0] = (uint32_t)(kc.data_offset / 64);
        s_pblk[tid][1] = (uint32_t)kc.blocks_x;
        s_pblk[tid][2] = (uint32_t)kc.samp_h | ((uint32_t)kc.samp_v << 8) | ((uint32_t)g.mcu_bx[pp] << 16) | ((uint32_t)g.mcu_by[pp] << 24);
        s_pblk[tid][3] = (uint32_t)c;
    }
    {
        const uint4* src = reinterpret_cast<const uint4*>(tabs);
        uint4* dst = reinterpret_cast<uint4*>(s_tab);
        for (int t = tid; t < 4 * GJ_DEC2_WORDS / 8; t += 256) dst[t] = src[t];
    }
    if (tid < 128) s_zz[tid] = tid < 64 ? GJ_ZZ[tid] : 63;
    const uint32_t* end = reinterpret_cast<const uint32_t*>((reinterpret_cast<uintptr_t>(jpeg) + jpeg_size + 3) & ~(uintptr_t)3);
    const int seg_count = seg_count_ptr ? min((int)*seg_count_ptr, seg_count_max) : seg_count_max;
    const int si0 = blockIdx.x * NS;
    if (si0 >= seg_count) return;
    const int nseg = min(NS, seg_count - si0);
    uint32_t my_ucap = 0;
    if (tid < GJ_SEQ_NS) {
        uint32_t pos = 0, len = 0, idx = 0xFFFFFFFFu;
        if (tid < nseg) {
            idx = seg_index[si0 + tid];
            if (idx < (uint32_t)g.segment_count) {
                pos = seg_pos[si0 + tid];
                len = seg_len[si0 + tid];
                if (((len + 3u) & ~3u) + 8u > (uint32_t)GJ_SEQ_STAGE) { // (not for this kernel)
                    *overflow = 1u;
                    len = 0;
                    idx = 0xFFFFFFFFu;
                }
            }
        }
        s_pos[tid] = pos;
        s_len[tid] = len;
        s_idx[tid] = idx;
        my_ucap = len ? ((len + 3u) & ~3u) + 8u : 0u;
    }
    {
        uint32_t tot;
        const uint32_t b = gj_wg256_incl_scan(my_ucap, s_tmp, &tot);
        if (tid < GJ_SEQ_NS) s_ub[tid + 1] = b;
        if (tid == 0) s_ub[0] = 0;
    }
    __syncthreads();
    const int P = g.blocks_per_mcu;
    for (int j0 = 0; j0 < nseg;) {
        // ---- the segments whose unstuffed bytes fit the stage together (normally all of them)
        if (tid == 0) s_j1 = j0 + 1;
        __syncthreads();
        if (tid > j0 && tid <= nseg && s_ub[tid] - s_ub[j0] <= (uint32_t)GJ_SEQ_STAGE) atomicMax(&s_j1, tid);
        __syncthreads();
        const int j1 = s_j1;
        const uint32_t ub0 = s_ub[j0];
        // ---- 1. one wave per segment: its blocks are filled with zeros (the planes need no clearing between frames), its bytes go to
        //         the stage without the stuffed zeros; the first 256 B of eight segments are fetched at a time
        for (int jb = j0 + wave; jb < j1; jb += 32) {
            uint32_t wpre[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int j = jb + 4 * q;
                wpre[q] = 0;
                if (j < j1 && s_len[j]) {
                    const uintptr_t a = reinterpret_cast<uintptr_t>(jpeg) + s_pos[j];
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
                    const uint32_t ndw = ((uint32_t)(a & 3) + s_len[j] + 3u) >> 2;
                    if ((uint32_t)lane < ndw && src + lane < end) wpre[q] = src[lane];
                }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int j = jb + 4 * q;
                if (j >= j1) break;
                if (zero_fill && s_idx[j] != 0xFFFFFFFFu) {
                    const GjSeg sg = gj_segment(g, (int)s_idx[j]);
                    for (int c = lane; c < sg.nblocks * 8; c += 64) {
                        uint64_t off;
                        if (INTERLEAVED) { // (gj_segment_block with the per-block constants from LDS)
                            const unsigned kb = (unsigned)c >> 3, mi = kb / (unsigned)P, pp = kb - mi * (unsigned)P, m = (unsigned)sg.mcu_first + mi;
                            const unsigned my = m / (unsigned)g.mcu_count_x, mx = m - my * (unsigned)g.mcu_count_x;
                            const uint32_t q = s_pblk[pp][2];
                            off = (uint64_t)(s_pblk[pp][0] + (my * ((q >> 8) & 0xFFu) + (q >> 24)) * s_pblk[pp][1] + mx * (q & 0xFFu) + ((q >> 16) & 0xFFu)) * 64;
                        } else {
                            off = g.comp[sg.comp].data_offset + (uint64_t)(sg.mcu_first + (c >> 3)) * 64;
                        }
                        reinterpret_cast<uint4*>(coefs + off)[c & 7] = make_uint4(0, 0, 0, 0);
                    }
                }
                const uint32_t out = s_len[j] ? gj_unstuff_segment(jpeg, end, s_pos[j], s_len[j], s_U, s_ub[j] - ub0, lane, wpre[q]) : 0u;
                if (lane == 0) s_ulen[j] = out;
            }
        }
        __syncthreads(); // (also orders the zeros before the coefficient stores)

        // ---- 2. every lane decodes its segment: src/gpujpeg_huffman_gpu_decoder.cu:397-495 / src/gpujpeg_huffman_cpu_decoder.c:245-372
        const int j = j0 + tid;
        if (j < j1 && s_idx[j] != 0xFFFFFFFFu) {
            const GjSeg sg = gj_segment(g, (int)s_idx[j]);
            const uint32_t* U = s_U + ((s_ub[j] - ub0) >> 2);
            const uint32_t end_bit = s_ulen[j] * 8u;
            int left = sg.nblocks;
            // block cursor
            int p = 0, comp = sg.comp;
            unsigned mx = 0, my = 0;
            uint64_t off;
            const uint16_t *tdc, *tac;
            auto place = [&]() { // plane address, component and tables of block p of MCU (mx, my)
                const uint32_t q = s_pblk[p][2];
                const uint32_t bx = mx * (q & 0xFFu) + ((q >> 16) & 0xFFu), by = my * ((q >> 8) & 0xFFu) + (q >> 24);
                off = (uint64_t)(s_pblk[p][0] + by * s_pblk[p][1] + bx) * 64;
                comp = (int)s_pblk[p][3];
                const uint32_t pt = s_ptab[p];
                tdc = s_tab + (pt & 0xFFFFu);
                tac = s_tab + (pt >> 16);
            };
            if (INTERLEAVED) {
                my = (unsigned)sg.mcu_first / (unsigned)g.mcu_count_x;
                mx = (unsigned)sg.mcu_first - my * (unsigned)g.mcu_count_x;
                place();
            } else {
                off = g.comp[comp].data_offset + (uint64_t)sg.mcu_first * 64;
                tdc = s_tab + (g.comp[comp].dc_table * 2 + 0) * GJ_DEC2_WORDS;
                tac = s_tab + (g.comp[comp].ac_table * 2 + 1) * GJ_DEC2_WORDS;
            }
            int dc0 = 0, dc1 = 0, dc2 = 0, dc3 = 0;
            int z = 0;
            uint32_t bitpos = 0, rd = 1, nxt = U[1];
            uint64_t acc = (uint64_t)U[0] << 32;
            int n = 32;
            while (left > 0) {
                int v = 0, adv = 64; // (data exhausted: the block ends here, its remaining coefficients stay zero)
                bool coef = false;
                if (bitpos < end_bit) {
                    if (n <= 32) {
                        acc |= (uint64_t)nxt << (32 - n);
                        n += 32;
                        rd++;
                        nxt = U[rd];
                    }
                    const uint32_t hi = (uint32_t)(acc >> 32);
                    const uint16_t* t = z == 0 ? tdc : tac;
                    uint32_t e = t[hi >> (32 - GJ_DEC_FAST_BITS)];
                    if ((e & 31u) == 0) e = t[(e >> 5) + ((hi >> 16) & 63u)]; // codes longer than 10 bits
                    const int tot = (int)(e & 31u), sz = (int)((e >> 5) & 15u);
                    adv = tot ? (int)(e >> 9) : 64; // (an entry of a table the stream never defined: give up on the block)
                    const int used = tot - sz;
                    const uint32_t bits = sz ? (hi << used) >> (32 - sz) : 0u;
                    v = (sz && bits < (1u << (sz - 1))) ? (int)bits - (int)((1u << sz) - 1u) : (int)bits;
                    coef = sz != 0;
                    acc <<= tot;
                    n -= tot;
                    bitpos = tot ? bitpos + (uint32_t)tot : end_bit;
                }
                if (z == 0) { // DC: predicted from the previous block of the component inside this segment
                    int pred = dc0;
                    if (INTERLEAVED) pred = comp == 0 ? dc0 : comp == 1 ? dc1 : comp == 2 ? dc2 : dc3;
                    v += pred;
                    if (!INTERLEAVED || comp == 0) dc0 = v; else if (comp == 1) dc1 = v; else if (comp == 2) dc2 = v; else dc3 = v;
                    coefs[off] = (int16_t)v;
                } else if (coef) {
                    const int pos = z + adv - 1;
                    if (pos < 64) coefs[off + s_zz[pos]] = (int16_t)v;
                }
                z += adv;
                if (z >= 64) { // next block of this segment
                    z = 0;
                    left--;
                    if (!INTERLEAVED) {
                        off += 64;
                    } else {
                        if (++p == P) {
                            p = 0;
                            if (++mx == (unsigned)g.mcu_count_x) { mx = 0; my++; }
                        }
                        place();
                    }
                }
            }
        }
        __syncthreads();
        j0 = j1;
    }
}

// ================================================================================================
// Dequantisation + IDCT, one thread per block
// ================================================================================================
// `zero`: every block is overwritten with zeros once it has been read, which leaves the coefficient planes ready for the
// entropy decoder of the next frame (it stores non-zero coefficients only) without a separate 2 B/sample memset.
__global__ __launch_bounds__(256) void k_idct(const gj_geom g, int16_t* __restrict__ coefs, const float* __restrict__ qtab,
                                              uint8_t* __restrict__ planes, const int zero)
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
            uint2* p = reinterpret_cast<uint2*>(raw + (size_t)y * pitch + (size_t)bx * 24);
            p[0] = make_uint2(px[0], px[1]);
            p[1] = make_uint2(px[2], px[3]);
            p[2] = make_uint2(px[4], px[5]);
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
        for (int i = 0; i < 16; i++) asm volatile("" : "+v"(pk[c][i]));
    }
    // (no early return for the threads past the last block: the compiler would sink the three transforms below it and keep
    // every staged coefficient alive until then)
    gj_store_rgb444<CS_FROM, CS_TO>(g, raw, pk, lb, nb, bx, by);
}

// ================================================================================================
// The same, fed by the entropy decoder's TOKENS (DESIGN 4.3 "token mode"): per block a record (first token, count, DC
// term) in coding order and, in one dense array, the non-zero AC coefficients as value | 2 x natural position << 16.
// A block costs 8 B + 4 B per non-zero coefficient of HBM traffic instead of 128 B written (twice) and read. Every lane
// clears its own 128-byte slot of the LDS tile, the wave copies the token range of its 64 blocks into LDS with 16-byte
// loads (consecutive blocks of a scan have consecutive tokens; a new range starts where a decoder batch ended), every lane
// scatters its own tokens into its slot (2-byte LDS stores) and reads the block back as rows. Nothing crosses waves, so
// there is no workgroup barrier. Blocks of segments too long for the decoder's LDS stage arrive through the coefficient
// planes as before (count 0xFFFF in the record). Non-interleaved scans only (plane order == coding order).
// ================================================================================================
#define GJ_TOK_STAGE 416 // tokens per wave in LDS (with the 32 KiB tile: four workgroups per CU)

// a lane's 128-byte slot of the block tile: row r (16 bytes) sits at (r ^ (lane & 7)) * 16, which spreads the row reads and
// writes of the 64 lanes over all banks without padding the slot. A token carries 2 x its natural position = row << 4 | column << 1
// in its upper half, so its place in the slot is that field XOR (lane & 7) << 4: one SDWA and + one xor per token.
__device__ __forceinline__ uint4* gj_slot_row(uint8_t* slot, const int lane, const int r)
{
    return reinterpret_cast<uint4*>(slot + ((uint32_t)(r << 4) ^ (((uint32_t)lane & 7u) << 4)));
}

__device__ __forceinline__ void gj_slot_put(uint8_t* slot, const int lane, const uint32_t tok)
{
    *reinterpret_cast<uint16_t*>(slot + (((tok >> 16) & 0x7Eu) ^ (((uint32_t)lane & 7u) << 4))) = (uint16_t)tok;
}

// the wave's token range of one component: dense and small enough for the stage (the normal case), with the two 16-byte
// loads per lane that fetch it
struct GjTokRange {
    uint32_t S, E;
    bool fast;
    uint4 t0, t1;
};

__device__ __forceinline__ GjTokRange gj_tok_fetch(const uint32_t* __restrict__ d_tok, const uint32_t start, const uint32_t cnt, const int lane)
{
    GjTokRange r;
    const uint32_t end = start + cnt;
    const uint32_t prev_end = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)end, 0x138, 0xF, 0xF, false); // wave_shr:1
    const unsigned long long breaks = __ballot(lane != 0 && start != prev_end);
    r.S = (uint32_t)__builtin_amdgcn_readlane((int)start, 0) & ~3u;
    r.E = (uint32_t)__builtin_amdgcn_readlane((int)end, 63);
    r.fast = breaks == 0 && r.E - r.S <= GJ_TOK_STAGE;
    r.t0 = r.t1 = make_uint4(0, 0, 0, 0);
    if (r.fast) {
        const uint32_t i0 = (uint32_t)lane * 4u, i1 = i0 + 256u;
        if (r.S + i0 < r.E) r.t0 = *reinterpret_cast<const uint4*>(d_tok + r.S + i0);
        if (i1 < GJ_TOK_STAGE && r.S + i1 < r.E) r.t1 = *reinterpret_cast<const uint4*>(d_tok + r.S + i1);
    }
    return r;
}

// One block per lane: zeros, the DC term and the lane's tokens go into its tile slot. `fast`: the wave's tokens are in the stage
// already (dense range starting at token S).
__device__ __forceinline__ void gj_tok_to_slot(uint8_t* slot, uint32_t* stage, const int lane, const bool fast, const uint32_t S, const uint32_t start,
                                               const uint32_t cnt, const uint32_t dc, const bool in_plane, const uint4* __restrict__ plane_block,
                                               const uint32_t* __restrict__ d_tok)
{
#pragma unroll
    for (int r = 0; r < 8; r++) *gj_slot_row(slot, lane, r) = make_uint4(0, 0, 0, 0);
    if (in_plane) { // block of a segment that was decoded piece by piece: it is in the coefficient plane
#pragma unroll
        for (int r = 0; r < 8; r++) *gj_slot_row(slot, lane, r) = plane_block[r];
    } else {
        *reinterpret_cast<uint16_t*>(slot + ((lane & 7) << 4)) = (uint16_t)dc;
    }
    const uint32_t end = start + cnt;
    if (fast) {
        gj_wave_sync();
        uint32_t a = start - S;
        const uint32_t b = end - S;
        for (; a + 2 <= b; a += 2) {
            const uint32_t ta = stage[a], tb = stage[a + 1];
            gj_slot_put(slot, lane, ta);
            gj_slot_put(slot, lane, tb);
        }
        if (a < b) gj_slot_put(slot, lane, stage[a]);
    } else {
        // several ranges (a decoder batch ended inside the wave's blocks) or more tokens than the stage holds: range by range,
        // chunk by chunk
        const uint32_t prev_end = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)end, 0x138, 0xF, 0xF, false);
        unsigned long long runs = __ballot(lane == 0 || start != prev_end);
        while (runs) {
            const int d = __builtin_ctzll(runs);
            runs &= runs - 1;
            const int dn = runs ? __builtin_ctzll(runs) : 64;
            const uint32_t RS = (uint32_t)__builtin_amdgcn_readlane((int)start, d), RE = (uint32_t)__builtin_amdgcn_readlane((int)end, dn - 1);
            const bool mine = lane >= d && lane < dn;
            for (uint32_t base = RS & ~3u; base < RE; base += GJ_TOK_STAGE) {
                gj_wave_sync();
                for (uint32_t i = (uint32_t)lane * 4u; i < GJ_TOK_STAGE && base + i < RE; i += 256u)
                    *reinterpret_cast<uint4*>(stage + i) = *reinterpret_cast<const uint4*>(d_tok + base + i);
                gj_wave_sync();
                if (mine) {
                    const uint32_t b = min(end, base + GJ_TOK_STAGE);
                    for (uint32_t a = max(start, base); a < b; a++) gj_slot_put(slot, lane, stage[a - base]);
                }
            }
        }
    }
    gj_wave_sync(); // (the stage is rewritten by the next component)
}

template <int CS_FROM, int CS_TO>
__global__ __launch_bounds__(256, 4) void k_idct_tok_rgb444(const gj_geom g, const int16_t* __restrict__ coefs, const uint2* __restrict__ d_rec,
                                                            const uint32_t* __restrict__ d_tok, const uint32_t tok_cap,
                                                            const float* __restrict__ qtab, uint8_t* __restrict__ raw)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_blk[256 * 128];
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[4][GJ_TOK_STAGE];
    __shared__ __attribute__((aligned(8))) float s_q[3][64];
    if (threadIdx.x < 192) s_q[threadIdx.x >> 6][threadIdx.x & 63] = qtab[g.comp[threadIdx.x >> 6].q_table * 64 + (threadIdx.x & 63)];
    const gj_comp_geom& k0 = g.comp[0];
    const unsigned nb = (unsigned)(k0.blocks_x * k0.blocks_y);
    const unsigned lb = blockIdx.x * 256u + threadIdx.x;
    const unsigned by = lb / (unsigned)k0.blocks_x, bx = lb - by * (unsigned)k0.blocks_x;
    const int lane = threadIdx.x & 63;
    uint8_t* slot = s_blk + threadIdx.x * 128;
    uint32_t* stage = s_stage[threadIdx.x >> 6];

    // ---- 1. the three block records (independent loads)
    uint32_t start[3], cnt[3], dc[3];
    bool in_plane[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        start[c] = cnt[c] = dc[c] = 0;
        in_plane[c] = false;
        if (lb < nb) {
            const uint2 r = d_rec[g.comp[c].data_offset / 64 + lb];
            start[c] = r.x;
            cnt[c] = r.y >> 16;
            dc[c] = r.y & 0xFFFFu;
            if (cnt[c] == 0xFFFFu) { in_plane[c] = true; cnt[c] = 0; }
            else if (cnt[c] > 63u || start[c] > tok_cap || cnt[c] > tok_cap - start[c]) cnt[c] = 0; // (a record nobody wrote: damaged stream)
        }
    }
    __syncthreads(); // (s_q; everything below is private to a wave)

    // ---- 2. per component: the tokens of the wave's 64 blocks go through the LDS stage (consecutive blocks of a scan have
    //         consecutive tokens); the loads of the next component are in flight while this one is transformed
    uint32_t pk[3][16];
    GjTokRange cur = gj_tok_fetch(d_tok, start[0], cnt[0], lane);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const bool fast = cur.fast;
        const uint32_t S = cur.S;
        if (fast) {
            *reinterpret_cast<uint4*>(stage + lane * 4) = cur.t0;
            if (lane * 4 + 256 < GJ_TOK_STAGE) *reinterpret_cast<uint4*>(stage + lane * 4 + 256) = cur.t1;
        }
        if (c < 2) cur = gj_tok_fetch(d_tok, start[c + 1], cnt[c + 1], lane);
        gj_tok_to_slot(slot, stage, lane, fast, S, start[c], cnt[c], dc[c], in_plane[c],
                       reinterpret_cast<const uint4*>(coefs + g.comp[c].data_offset + (size_t)lb * 64), d_tok);
        // the block as rows; dequantisation + IDCT
        uint32_t wb[32];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint4 v = *gj_slot_row(slot, lane, r);
            wb[r * 4] = v.x; wb[r * 4 + 1] = v.y; wb[r * 4 + 2] = v.z; wb[r * 4 + 3] = v.w;
        }
        gj_idct_pk(wb, s_q[c], pk[c]);
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("" : "+v"(pk[c][i])); // one transform at a time (see k_idct_fused_rgb444)
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
                                                             const uint32_t* __restrict__ d_tok, const uint32_t tok_cap,
                                                             const float* __restrict__ qtab, uint8_t* __restrict__ raw)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_blk[256 * 128];
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[4][GJ_TOK_STAGE];
    __shared__ __attribute__((aligned(8))) float s_q[3][64];
    if (threadIdx.x < 192) s_q[threadIdx.x >> 6][threadIdx.x & 63] = qtab[g.comp[threadIdx.x >> 6].q_table * 64 + (threadIdx.x & 63)];
    const gj_comp_geom& kc = g.comp[1];
    const unsigned nm = (unsigned)(kc.blocks_x * kc.blocks_y);
    const int p = threadIdx.x & 3; // Y0 Y1 Cb Cr
    const unsigned m = blockIdx.x * 64u + (threadIdx.x >> 2);
    const unsigned my = m / (unsigned)kc.blocks_x, mx = m - my * (unsigned)kc.blocks_x;
    const int c = p < 2 ? 0 : p - 1;
    const int lane = threadIdx.x & 63;
    uint8_t* slot = s_blk + threadIdx.x * 128;
    uint32_t* stage = s_stage[threadIdx.x >> 6];
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
        *reinterpret_cast<uint4*>(stage + lane * 4) = tr.t0;
        if (lane * 4 + 256 < GJ_TOK_STAGE) *reinterpret_cast<uint4*>(stage + lane * 4 + 256) = tr.t1;
    }
    __syncthreads(); // (s_q)
    const size_t blk = p < 2 ? (size_t)my * g.comp[0].blocks_x + 2 * mx + p : (size_t)m; // (plane address: blocks of long segments only)
    gj_tok_to_slot(slot, stage, lane, tr.fast, tr.S, start, cnt, dc, in_plane,
                   reinterpret_cast<const uint4*>(coefs + g.comp[c].data_offset + (m < nm ? blk : 0) * 64), d_tok);
    uint32_t wb[32];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint4 v = *gj_slot_row(slot, lane, r);
        wb[r * 4] = v.x; wb[r * 4 + 1] = v.y; wb[r * 4 + 2] = v.z; wb[r * 4 + 3] = v.w;
    }
    uint32_t px[16];
    gj_idct_pk(wb, s_q[c], px);

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
        for (int t = 0; t < 16; t++) asm volatile("" : "+v"(pk[b][t])); // one transform at a time
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

// developer aid (gj_tuning::debug_sync): waits after every launch and names the stage on stderr (which kernel faulted?)
static void gj_debug_stage(const bool on, hipStream_t st, const char* what)
{
    if (!on) return;
    const hipError_t e = hipStreamSynchronize(st);
    fprintf(stderr, "[GPUJPEG] [Debug] %s: %s\n", what, hipGetErrorString(e));
}

// ================================================================================================
// Launcher
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

typedef void (*gj_idct_tok_t)(const gj_geom, const int16_t*, const uint2*, const uint32_t*, uint32_t, const float*, uint8_t*);

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

static bool gj_is_uyvy422(const gj_geom& g)
{
    return g.pixel_format == GJ_PF_422_P1020 && g.comp_count == 3 &&
           (g.color_space == g.color_space_internal || g.color_space == GJ_CS_NONE || g.color_space_internal == GJ_CS_NONE) &&
           g.comp[0].samp_h == 2 && g.comp[0].samp_v == 1 && g.comp[1].samp_h == 1 && g.comp[1].samp_v == 1 && g.comp[2].samp_h == 1 &&
           g.comp[2].samp_v == 1 && g.comp[0].blocks_x == 2 * g.comp[1].blocks_x && g.comp[0].blocks_y == g.comp[1].blocks_y &&
           g.comp[2].blocks_x == g.comp[1].blocks_x && g.comp[2].blocks_y == g.comp[1].blocks_y;
}

// the token-fed IDCT kernel for this configuration, or nullptr
static gj_idct_tok_t gj_idct_tok_for(const gj_geom& g)
{
    if (!g.interleaved) return gj_idct_tok_kernel(g);
    if (gj_is_uyvy422(g) && g.blocks_per_mcu == 4 && g.mcu_count == g.comp[1].blocks_x * g.comp[1].blocks_y && g.mcu_comp[0] == 0 && g.mcu_comp[1] == 0 &&
        g.mcu_comp[2] == 1 && g.mcu_comp[3] == 2 && g.mcu_bx[0] == 0 && g.mcu_bx[1] == 1)
        return k_idct_tok_uyvy422;
    return nullptr;
}

// Does a frame of this geometry and stream size go through token mode (given fused kernels and two-level Huffman tables)? The host
// asks before it allocates the token buffers; the launcher asks again.
// Measured (8K / 16K RGB natural frames at q75, 4.75 B of stream per block: +17 % enc+dec; HD and 4K equal or slightly slower;
// 16K 4:2:2 at q90, 10.4 B per block: -6 %; 8K noise -15 %; crossover at 8K RGB near 9 B per block): tokens pay when the frame
// fills the GPU more than once (the token-fed IDCT has the longer dependency chain per workgroup) and blocks carry few coefficients
// (4 B per coefficient against 128 B per block). gj_tuning::dec_tokens forces either mode (tests, A/B runs).
extern "C" int gj_hip_decode_wants_tokens(const gj_geom* g, uint64_t jpeg_size, const gj_tuning* tune)
{
    if (tune->dec_tokens == 0 || gj_idct_tok_for(*g) == nullptr) return 0;
    if (tune->dec_sub && tune->dec_sub != (g->interleaved ? 32 : GJ_PAR_SUB)) return 0; // (the tuning aid sweeps the plane-mode kernels)
    if (tune->dec_tokens == 1) return 1;
    return g->block_count >= 900000 && jpeg_size <= (uint64_t)g->block_count * 8u;
}

extern "C" int gj_hip_decode(const gj_dec_job* job, gj_stream_t stream, gj_event_t ev[4])
{
    hipStream_t st = (hipStream_t)stream;
    const gj_geom& g = job->g;
    if (g.blocks_per_mcu > GJ_MAX_MCU_BLOCKS) return -1;
    if (ev) (void)hipEventRecord((hipEvent_t)ev[0], st);
    bool par = job->d_huff_tab2 != nullptr && job->seg_count > 0;
    if (job->tune.dec_serial) par = false; // the lane-per-segment kernel (A/B measurements, tests)
    const bool uyvy = job->use_fused && gj_is_uyvy422(g);
    // token mode: the entropy decoder hands the non-zero coefficients to the fused IDCT as a dense token array plus one
    // record per block instead of through the coefficient planes
    gj_idct_tok_t idct_tok = (par && job->tokens && job->use_fused && job->d_tok && job->d_blkrec && gj_hip_decode_wants_tokens(&g, job->jpeg_size, &job->tune))
                                 ? gj_idct_tok_for(g) : nullptr;
    const bool tokens = idct_tok != nullptr;
    // Both entropy decoders store only non-zero coefficients. The sub-sequence kernel zero-fills the blocks of the segments it
    // decodes itself; clear_coefs asks for a full clear first (segments missing from the table, lane-per-segment kernel).
    if (tokens) {
        if (job->clear_coefs) (void)hipMemsetAsync(job->d_blkrec, 0, (size_t)g.block_count * sizeof(uint2), st);
    } else if (job->clear_coefs || !par) {
        (void)hipMemsetAsync(job->d_coefs, 0, g.data_size * sizeof(int16_t), st);
    }
    // interleaved scans with many short segments in plane mode: one lane per segment (k_huffman_decode_seq); the host vouches for the
    // longest segment (this stream's, or the previous frame's on the speculative path, where `d_overflow` is checked afterwards)
    const bool seq = par && !tokens && job->d_overflow != nullptr && job->tune.dec_seq != 2 &&
                     (job->tune.dec_seq == 1 || (g.interleaved && job->max_seg_len != 0 && job->max_seg_len <= 1024u && job->seg_count >= 16384));
    if (seq) {
        const unsigned avg = (unsigned)(job->jpeg_size / (uint64_t)job->seg_count) + 12u;
        const int NS = max(1, min(GJ_SEQ_NS, (int)((GJ_SEQ_STAGE * 7u / 8u) / avg)));
        auto kernel = g.interleaved ? k_huffman_decode_seq<true> : k_huffman_decode_seq<false>;
        hipLaunchKernelGGL(kernel, dim3(((unsigned)job->seg_count + NS - 1) / NS), dim3(256), 0, st, g, job->d_jpeg, job->jpeg_size, job->d_seg_pos, job->d_seg_len,
                           job->d_seg_index, job->seg_count, job->d_seg_count, NS, job->d_huff_tab2, job->d_coefs, job->clear_coefs ? 0 : 1, job->d_overflow);
    } else if (par) {
        // batches: as many segments as fill the LDS stage on average, at most GJ_PAR_MAX_BLOCKS blocks, per scan where the bytes per scan
        // are known (GjBatchPlan)
        const int eg = job->tune.dec_batch, es = job->tune.dec_sub; // tuning aids: segments per batch, bytes per sub-sequence
        const unsigned cap_u = GJ_PAR_CAP_U_FOR(tokens), max_blocks = GJ_PAR_MAX_BLOCKS_FOR(tokens);
        auto batch_size = [&](uint64_t bytes, int segs, unsigned fill /* 32nds of the stage */) {
            const unsigned avg = (unsigned)(bytes / (uint64_t)max(1, segs)) + 12u;
            int G = eg ? eg : (int)((cap_u * fill / 32u) / avg); // (a batch that outgrows the stage is decoded in two groups)
            if (!eg) G = min(G, max(1, job->seg_count / 768)); // small frames: rather more, shorter batches than idle CUs (measured: HD, 4K)
            return max(1, min(G, (int)min((unsigned)GJ_PAR_GMAX, max_blocks / (unsigned)max(1, g.seg_blocks))));
        };
        GjBatchPlan plan = {};
        bool per_scan = !g.interleaved && g.comp_count > 1 && job->seg_count == g.segment_count;
        for (int c = 0; per_scan && c < g.comp_count; c++) per_scan = job->scan_bytes[c] != 0 && g.comp[c].segment_count > 0;
        // 23/32 of the stage on average is the measured optimum; when that gives a little more than one generation of resident
        // workgroups, fuller batches (up to 27/32) that fit into one are better than a second generation of a few
        for (unsigned fill = 23; fill <= 27; fill += 2) {
            if (per_scan) {
                plan.n = g.comp_count;
                int first = 0;
                for (int c = 0; c < g.comp_count; c++) {
                    plan.first[c] = first;
                    plan.count[c] = g.comp[c].segment_count;
                    plan.g[c] = batch_size(job->scan_bytes[c], plan.count[c], fill);
                    plan.batch0[c + 1] = plan.batch0[c] + (plan.count[c] + plan.g[c] - 1) / plan.g[c];
                    first += plan.count[c];
                }
            } else {
                plan.n = 1;
                plan.count[0] = job->seg_count;
                plan.g[0] = batch_size(job->jpeg_size, job->seg_count, fill);
                plan.batch0[1] = (job->seg_count + plan.g[0] - 1) / plan.g[0];
            }
            const int nb = plan.batch0[plan.n];
            if (!tokens || eg || nb <= GJ_PAR_RESIDENT || nb > GJ_PAR_RESIDENT * 5 / 4) break;
        }
        const int sub = es ? es : (g.interleaved ? 32 : GJ_PAR_SUB); // interleaved scans synchronise later (the block inside the MCU
                                                                           // has to fall into step too): measured best with 32 B
        const unsigned batches = (unsigned)plan.batch0[plan.n];
        auto kernel = tokens ? (g.interleaved ? k_huffman_decode_par<true, 32, true> : k_huffman_decode_par<false, GJ_PAR_SUB, true>)
                      : g.interleaved ? (sub == 256 ? k_huffman_decode_par<true, 256, false> : sub == 128 ? k_huffman_decode_par<true, 128, false>
                                         : sub == 64 ? k_huffman_decode_par<true, 64, false> : sub == 32 ? k_huffman_decode_par<true, 32, false>
                                         : sub == 8 ? k_huffman_decode_par<true, 8, false> : k_huffman_decode_par<true, 16, false>)
                                      : (sub == 256 ? k_huffman_decode_par<false, 256, false> : sub == 128 ? k_huffman_decode_par<false, 128, false>
                                         : sub == 64 ? k_huffman_decode_par<false, 64, false> : sub == 32 ? k_huffman_decode_par<false, 32, false>
                                         : sub == 8 ? k_huffman_decode_par<false, 8, false> : k_huffman_decode_par<false, 16, false>);
        hipLaunchKernelGGL(kernel, dim3(batches), dim3(256), 0, st, g, job->d_jpeg, job->jpeg_size, job->d_seg_pos, job->d_seg_len,
                           job->d_seg_index, job->seg_count, job->d_seg_count, plan, job->d_huff_tab2, job->d_coefs, job->clear_coefs ? 0 : 1, job->d_tok, job->tok_cap,
                           (uint2*)job->d_blkrec);
    } else {
        if (job->seg_count > 0) {
            auto kernel = g.interleaved ? k_huffman_decode<true> : k_huffman_decode<false>;
            hipLaunchKernelGGL(kernel, dim3(((unsigned)job->seg_count + 255) / 256), dim3(256), 0, st, g, job->d_jpeg, job->jpeg_size, job->d_seg_pos,
                               job->d_seg_len, job->d_seg_index, job->d_seg_count, job->seg_count, (const uint32_t*)nullptr, job->d_huff_tab,
                               job->d_coefs);
        }
    }
    gj_debug_stage(job->tune.debug_sync != 0, st, "entropy decoder");
    if (ev) (void)hipEventRecord((hipEvent_t)ev[1], st);
    gj_idct_fused_t fused = job->use_fused ? gj_idct_fused_kernel(g) : nullptr;
    if (tokens) {
        const unsigned nb = g.interleaved ? (unsigned)g.block_count : (unsigned)(g.comp[0].blocks_x * g.comp[0].blocks_y); // one lane per block (position)
        hipLaunchKernelGGL(idct_tok, dim3((nb + 255) / 256), dim3(256), 0, st, g, job->d_coefs, (const uint2*)job->d_blkrec, job->d_tok, job->tok_cap,
                           job->d_qtabf, job->d_raw);
        if (ev) (void)hipEventRecord((hipEvent_t)ev[2], st);
    } else if (uyvy) {
        const unsigned nm = (unsigned)(g.comp[1].blocks_x * g.comp[1].blocks_y);
        hipLaunchKernelGGL(k_idct_fused_uyvy422, dim3((nm + 255) / 256), dim3(256), 0, st, g, job->d_coefs, job->d_qtabf, job->d_raw, job->zero_coefs);
        if (ev) (void)hipEventRecord((hipEvent_t)ev[2], st);
    } else if (fused) {
        const unsigned nb = (unsigned)(g.comp[0].blocks_x * g.comp[0].blocks_y);
        hipLaunchKernelGGL(fused, dim3((nb + 255) / 256), dim3(256), 0, st, g, job->d_coefs, job->d_qtabf, job->d_raw, job->zero_coefs);
        if (ev) (void)hipEventRecord((hipEvent_t)ev[2], st);
    } else {
        hipLaunchKernelGGL(k_idct, dim3(((unsigned)g.block_count + 255) / 256), dim3(256), 0, st, g, job->d_coefs, job->d_qtabf,
                           job->d_planes, job->zero_coefs);
        if (job->flipped) hipLaunchKernelGGL(k_flip_planes, dim3(1024), dim3(256), 0, st, g, job->d_planes); // src/gpujpeg_postprocessor.cu:447
        if (ev) (void)hipEventRecord((hipEvent_t)ev[2], st);
        if (g.no_transform) {
            hipLaunchKernelGGL(k_copy_planes_out, dim3(2048), dim3(256), 0, st, g, job->d_planes, job->d_raw);
        } else {
            const unsigned n = (unsigned)g.raw_width * (unsigned)g.height;
            hipLaunchKernelGGL(k_postprocess, dim3((n + 255) / 256), dim3(256), 0, st, g, job->d_planes, job->d_raw);
        }
    }
    gj_debug_stage(job->tune.debug_sync != 0, st, "idct / postprocess");
    if (job->channel_remap) { // src/gpujpeg_postprocessor.cu:450,493: the finished image is permuted in place
        const unsigned n = (unsigned)g.width * (unsigned)g.height;
        hipLaunchKernelGGL(k_channel_remap, dim3((n + 255) / 256), dim3(256), 0, st, g, job->d_raw, job->channel_remap & 0xFFFFu);
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
__global__ __launch_bounds__(256) void k_build_segments(const gj_geom g, const uint8_t* __restrict__ jpeg, const uint32_t* __restrict__ rst_pos, uint64_t begin, uint64_t size,
                                                        gj_scan_summary* __restrict__ sum, uint32_t* __restrict__ seg_pos,
                                                        uint32_t* __restrict__ seg_len, uint32_t* __restrict__ seg_index, uint32_t max_segments)
{
    __shared__ uint32_t s_start[GJ_MAX_COMP + 1], s_end[GJ_MAX_COMP + 1], s_first[GJ_MAX_COMP + 2];
    __shared__ int s_scans;
    __shared__ uint32_t s_opos[GJ_SCAN_MAX_OTHER];
    __shared__ uint8_t s_order[GJ_SCAN_MAX_OTHER];
    // rst_pos holds max_segments - GJ_MAX_COMP valid entries at most (k_marker_emit stops there): a stream with more restart markers
    // than the geometry allows is damaged; the table is cut and the host, seeing the count, rejects it
    const uint32_t n_rst = min(sum->rst_count, max_segments - GJ_MAX_COMP);
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
        s_scans = scans;
        if (blockIdx.x == 0) {
            sum->scan_count = (uint32_t)scans;
            sum->status = (uint32_t)status;
            sum->segment_count = scans ? n_rst + (uint32_t)scans : 0u;
            for (int sc = 0; sc < scans; sc++) { sum->scan_start[sc] = s_start[sc]; sum->scan_end[sc] = s_end[sc]; }
        }
    }
    __syncthreads();
    {   // rank of the first RSTn of every scan (lower bound in the ordered list): wave sc searches for scan sc with 64 probes
        // per round, i.e. three dependent loads instead of sixteen
        const int sc = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (threadIdx.x == 0) s_first[s_scans] = n_rst; // sentinel: everything lies below the end
        if (sc < s_scans) {
            const uint32_t key = s_start[sc];
            uint32_t lo = 0, hi = n_rst;
            while (lo < hi) {
                const uint32_t step = (hi - lo + 63u) / 64u;
                const uint32_t idx = lo + (uint32_t)lane * step;
                const bool below = idx < hi && rst_pos[idx] < key;
                const uint32_t cnt = (uint32_t)__popcll(__ballot(below)); // the probes are ordered: the first cnt are below the key
                if (step == 1) { lo += cnt; break; }
                if (cnt < 64u) hi = min(hi, lo + cnt * step);
                if (cnt) lo += (cnt - 1u) * step + 1u;
            }
            if (lane == 0) s_first[sc] = lo;
        }
    }
    __syncthreads();
    const int scans = s_scans;
    const uint32_t gidx = blockIdx.x * 256u + threadIdx.x;
    if (scans == 0) return; // no scan ends inside the data (truncated file, no marker at all): the host decides what to do
    if (gidx >= n_rst + (uint32_t)scans || gidx >= max_segments) return;
    int sc = 0;
    while (sc + 1 < scans && gidx >= s_first[sc + 1] + (uint32_t)(sc + 1)) sc++;
    const uint32_t k = gidx - s_first[sc] - (uint32_t)sc;       // index of the segment inside its scan
    const uint32_t c_s = s_first[sc + 1] - s_first[sc];         // RSTn inside this scan
    if (k > c_s) return;                                        // (inconsistent ranks: damaged stream)
    const uint32_t from = k == 0 ? s_start[sc] : rst_pos[s_first[sc] + k - 1] + 2;
    const uint32_t to = k == c_s ? s_end[sc] : rst_pos[s_first[sc] + k];
    // the marker that ends segment k must be RST(k mod 8), and the last segment of a scan must not be empty: anything else is a
    // stream the reference reader treats specially, which the host walk reproduces
    if ((k < c_s && jpeg[to + 1] != (uint8_t)(0xD0 + (k & 7u))) || (k == c_s && c_s > 0 && to <= from)) sum->rst_irregular = 1u;
    seg_pos[gidx] = from;
    seg_len[gidx] = to > from ? to - from : 0;
    if (to > from) atomicMax(&sum->max_seg_len, to - from);
    // scan i carries component i when the stream is not interleaved (src/gpujpeg_reader.c:1345)
    const uint32_t first = g.interleaved ? 0u : (uint32_t)g.comp[sc < g.comp_count ? sc : 0].first_segment;
    const uint32_t limit = g.interleaved ? (uint32_t)g.segment_count : (uint32_t)g.comp[sc < g.comp_count ? sc : 0].segment_count;
    seg_index[gidx] = k < limit ? first + k : 0xFFFFFFFFu;
}

extern "C" int gj_hip_find_segments(const gj_geom* g, const uint8_t* d_jpeg, uint64_t begin, uint64_t size, uint32_t* d_seg_pos,
                                    uint32_t* d_seg_len, uint32_t* d_seg_index, uint32_t max_segments, uint32_t* d_scratch,
                                    gj_scan_summary* d_summary, gj_stream_t stream, int debug_sync)
{
    hipStream_t st = (hipStream_t)stream;
    if (size <= begin) return -1;
    const uint32_t chunks = (uint32_t)((size - begin + GJ_SCAN_CHUNK - 1) / GJ_SCAN_CHUNK);
    uint32_t* d_chunk = d_scratch;          // [chunks]
    uint32_t* d_rst = d_scratch + chunks;   // [max_segments]
    (void)hipMemsetAsync(d_summary, 0, sizeof(gj_scan_summary), st);
    hipLaunchKernelGGL(k_marker_count, dim3(chunks), dim3(256), 0, st, d_jpeg, begin, size, d_chunk, d_summary);
    gj_debug_stage(debug_sync != 0, st, "k_marker_count");
    hipLaunchKernelGGL(k_marker_rank, dim3(1), dim3(1024), 0, st, d_chunk, chunks, d_summary);
    hipLaunchKernelGGL(k_marker_emit, dim3(chunks), dim3(256), 0, st, d_jpeg, begin, size, d_chunk, d_rst, max_segments);
    gj_debug_stage(debug_sync != 0, st, "k_marker_rank + k_marker_emit");
    hipLaunchKernelGGL(k_build_segments, dim3((max_segments + GJ_MAX_COMP + 255) / 256), dim3(256), 0, st, *g, d_jpeg, d_rst, begin, size, d_summary,
                       d_seg_pos, d_seg_len, d_seg_index, max_segments + GJ_MAX_COMP);
    gj_debug_stage(debug_sync != 0, st, "k_build_segments");
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

__global__ __launch_bounds__(256) void k_compare_header(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint32_t n,
                                                         gj_scan_summary* __restrict__ sum)
{
    int diff = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 256) diff |= a[i] != b[i];
    diff = __syncthreads_or(diff);
    if (threadIdx.x == 0) sum->header_differs = diff ? 1u : 0u;
}

extern "C" int gj_hip_compare_header(const uint8_t* d_jpeg, const uint8_t* d_ref, uint32_t n, gj_scan_summary* d_summary, gj_stream_t stream)
{
    hipLaunchKernelGGL(k_compare_header, dim3(1), dim3(256), 0, (hipStream_t)stream, d_jpeg, d_ref, n, d_summary);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" size_t gj_hip_find_segments_scratch_words(uint64_t begin, uint64_t size, uint32_t max_segments)
{
    return (size_t)((size - begin + GJ_SCAN_CHUNK - 1) / GJ_SCAN_CHUNK) + max_segments + 16;
}
