// gj_dec_entropy_seq.hip -- MI355X (gfx950, wave64) JPEG decoder: entropy decoding, one lane per restart segment over an LDS stage (interleaved scans with many short segments)
// (part of the decoder's device code, see gj_dec_internal.h for the map of the files)
#include "gj_dec_internal.h"
#include "gj_bitreader.h"

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

// TOK (token mode): instead of scattering the coefficients into the (zero-filled) planes, a lane appends the non-zero AC
// coefficients of its segment as 16-bit tokens (value << 6 | natural position) to the segment's own run of the token array -- it starts at
// token 4 x the segment's byte offset: a token takes at least 3 bits of the stream, so runs cannot overlap -- four tokens per 8-byte store,
// and writes one record per block in coding order (first token, count, DC term). A coefficient beyond a token's 10 value bits raises
// `overflow` (the host decodes the frame again through the planes).
template <bool INTERLEAVED, bool TOK>
__global__ __launch_bounds__(256, 4) void k_huffman_decode_seq(const gj_geom g, const uint8_t* __restrict__ jpeg, const uint64_t jpeg_size,
                                                               const uint32_t* __restrict__ seg_pos, const uint32_t* __restrict__ seg_len,
                                                               const uint32_t* __restrict__ seg_index, const int seg_count_max,
                                                               const uint32_t* __restrict__ seg_count_ptr, const int NS,
                                                               const uint16_t* __restrict__ tabs, int16_t* __restrict__ coefs, const int zero_fill,
                                                               uint32_t* __restrict__ overflow, uint16_t* __restrict__ d_tok, const uint32_t tok_cap,
                                                               uint2* __restrict__ d_rec)
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
    if (tid < 128) s_zz[tid] = tid < 64 ? GJ_ZZ[tid] : 0; // (behind a block's end, damaged streams only: a token there lands on position 0, which the IDCT overwrites
                                                          //  with the DC term, i.e. it is dropped like in the plane kernels; the plane path below asks pos < 64)
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
                if (!TOK && zero_fill && s_idx[j] != 0xFFFFFFFFu) {
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
            // token mode: the segment's run of tokens, the 4-token buffer, the block in progress (record index in coding order, its first token, its DC term)
            const uint32_t tbase = 4u * s_pos[j];
            const bool tok_ok = !TOK || (tbase <= tok_cap && 4u * s_len[j] + 8u <= tok_cap - tbase); // (always, with the capacity the host allocates)
            uint32_t ntok = 0, blk_first = 0, blk_dc = 0, big = 0;
            uint64_t tbuf = 0;
            uint32_t rec = INTERLEAVED ? (uint32_t)sg.mcu_first * (uint32_t)P : (uint32_t)sg.first_block;
            uint32_t bitpos = 0, rd = 1, nxt = U[1];
            uint64_t acc = (uint64_t)U[0] << 32;
            int n = 32;
            while (left > 0) {
                int v = 0, adv = 64; // (data exhausted: the block ends here, its remaining coefficients stay zero)
                bool coef = false;
                uint32_t e_sz = 0;
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
                    adv = tot ? (int)((e >> 9) & 63u) : 64; // (an entry of a table the stream never defined: give up on the block)
                    const int used = tot - sz;
                    const uint32_t bits = sz ? (hi << used) >> (32 - sz) : 0u;
                    v = (sz && bits < (1u << (sz - 1))) ? (int)bits - (int)((1u << sz) - 1u) : (int)bits;
                    coef = sz != 0;
                    e_sz = (uint32_t)sz;
                    acc <<= tot;
                    n -= tot;
                    bitpos = tot ? bitpos + (uint32_t)tot : end_bit;
                }
                if (z == 0) { // DC: predicted from the previous block of the component inside this segment
                    int pred = dc0;
                    if (INTERLEAVED) pred = comp == 0 ? dc0 : comp == 1 ? dc1 : comp == 2 ? dc2 : dc3;
                    v += pred;
                    if (!INTERLEAVED || comp == 0) dc0 = v; else if (comp == 1) dc1 = v; else if (comp == 2) dc2 = v; else dc3 = v;
                    if (TOK) blk_dc = (uint32_t)v;
                    else coefs[off] = (int16_t)v;
                } else if (coef) {
                    const int pos = z + adv - 1;
                    if (TOK) {
                        big |= (e_sz >= 10u) ? 1u : 0u;
                        tbuf = (tbuf >> 16) | ((uint64_t)(uint16_t)(((uint32_t)v << 6) | s_zz[pos]) << 48);
                        ntok++;
                        if ((ntok & 3u) == 0 && tok_ok) *reinterpret_cast<uint2*>(d_tok + tbase + ntok - 4u) = make_uint2((uint32_t)tbuf, (uint32_t)(tbuf >> 32));
                    } else if (pos < 64) {
                        coefs[off + s_zz[pos]] = (int16_t)v;
                    }
                }
                z += adv;
                if (z >= 64) { // next block of this segment
                    z = 0;
                    left--;
                    if (TOK) { // the block's record: where its tokens are, how many, the DC term
                        d_rec[rec] = make_uint2(tok_ok ? tbase + blk_first : 0u, ((tok_ok ? min(ntok - blk_first, 63u) : 0u) << 16) | (blk_dc & 0xFFFFu));
                        rec++;
                        blk_first = ntok;
                        blk_dc = 0;
                    }
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
            if (TOK) {
                if (tok_ok) // the last one to three tokens
                    for (uint32_t r = ntok & 3u, i = 0; i < r; i++) d_tok[tbase + (ntok & ~3u) + i] = (uint16_t)(tbuf >> (16u * (4u - r + i)));
                if (big) *overflow = 1u; // a value beyond a token's 10 bits: the host decodes the frame again through the planes
            }
        }
        __syncthreads();
        j0 = j1;
    }
}


// ================================================================================================
// The same idea without the stage (round 3, token mode only): ONE LANE PER RESTART SEGMENT OVER AN 80-BYTE RING.
//
// What bounds the kernel above is not instruction issue but the latency of a wave's dependent chain (~100 vector instructions and three LDS
// round trips per symbol, 1 700 cycles per symbol and wave measured on config 4) at the 1.4 decoding waves per SIMD its stage admits: a
// lane needs its whole segment (~250 B) in LDS, a workgroup's 26 KB hold 87 segments, and the 172 800 segments of config 4 take two
// generations of workgroups. Here a lane owns a ring of 20 dwords and tops it up itself: all 172 800 segments are resident at once
// (675 workgroups of 256 lanes, 2.6 decoding waves per SIMD), there is no stage, no cooperative unstuffing, no barrier after the tables
// have been loaded, and a segment may be of any length.
//   * refill, wave-synchronous: as soon as ANY lane of a wave has fewer than 16 bytes in front of its bit position, EVERY lane of the
//     wave tops its ring up (a refill on demand per lane would run the refill code in almost every iteration for somebody): 16-byte
//     pieces of the lane's own part of the stream (four guarded dword loads, the next piece's in flight), the stuffed zeros removed
//     dword by dword -- byte classes with SWAR arithmetic, the kept bytes compacted by v_perm_b32 with a selector from a 16-entry LDS
//     table -- and appended to a dword that goes to the ring when it is complete (big-endian, as the bit reader wants it). Behind the
//     last byte of the segment go 8 zero bytes (a symbol that straddles the end reads zeros, src/gpujpeg_huffman_cpu_decoder.c:80-118);
//   * the reader is the one of gj_dec_entropy_tok.hip: bit position - 1 inside the ring, the next 32 bits with one v_alignbit_b32 over two
//     neighbouring slots (slot 20 mirrors slot 0); it never reads a dword the writer has not completed (that is what the 16 bytes in
//     front are for);
//   * symbols as in k_huffman_decode_seq<il, true>; sixteen tokens and four block records are collected in LDS and leave as aligned
//     32-byte pieces (see "Output" below).
// ================================================================================================
// a 16-byte store that does not stay in the L2: the ring kernel's tokens and records are written once and read by the NEXT kernel, while the lines of
// the stream every lane comes back to sixteen bytes later should stay (measured: FETCH_SIZE of the kernel at config 4, profiles/r4_20_*)
typedef uint32_t gj_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gj_store16_stream(void* dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
#ifdef GJ_NO_NT_STORES // (A/B build: make variant NAME=nont DEFS=-DGJ_NO_NT_STORES)
    *reinterpret_cast<gj_u32x4*>(dst) = gj_u32x4{a, b, c, d};
#else
    __builtin_nontemporal_store(gj_u32x4{a, b, c, d}, reinterpret_cast<gj_u32x4*>(dst));
#endif
}

#define GJ_WIN_DW 20     // dwords of a lane's ring
#define GJ_WIN_STRIDE 21 // dwords between the rings of neighbouring lanes: the ring + the mirror of its slot 0 (odd: the lanes of a half wave hit different banks)
#define GJ_WIN_OUT 17    // dwords of a lane's output stage: 16 tokens, 4 block records (+ 1: odd again)
#define GJ_WIN_AHEAD 128 // complete-dword bits the ring must hold in front of the bit position before a symbol is decoded

template <bool INTERLEAVED>
__global__ __launch_bounds__(256, 3) void k_huffman_decode_win(const gj_geom g, const uint8_t* __restrict__ jpeg, uint64_t jpeg_size,
                                                               const uint32_t* __restrict__ seg_pos, const uint32_t* __restrict__ seg_len,
                                                               const uint32_t* __restrict__ seg_index, const int seg_count_max,
                                                               const uint32_t* __restrict__ seg_count_ptr, const uint16_t* __restrict__ tabs,
                                                               uint32_t* __restrict__ overflow, uint16_t* __restrict__ d_tok, const uint32_t tok_cap,
                                                               uint2* __restrict__ d_rec)
{
    if (g.fb.sizes != nullptr) { // frame blockIdx.z of a batch: its stream, its table, its summary words, its tokens and records
        const size_t z = blockIdx.z;
        jpeg += z * g.fb.jpeg;
        jpeg_size = g.fb.sizes[z];
        seg_pos += z * g.fb.seg; seg_len += z * g.fb.seg; seg_index += z * g.fb.seg;
        if (seg_count_ptr) seg_count_ptr += z * (sizeof(gj_scan_summary) / 4);
        overflow += z * (sizeof(gj_scan_summary) / 4);
        d_tok += z * g.fb.tok; d_rec += z * g.fb.rec;
    }
    __shared__ uint32_t s_ring[256 * GJ_WIN_STRIDE];
    __shared__ uint32_t s_out[256 * GJ_WIN_OUT];
    __shared__ __attribute__((aligned(16))) uint16_t s_tab[4 * GJ_DEC2_WORDS];
    __shared__ uint8_t s_zz[64 + 64];
    __shared__ uint32_t s_ptab[GJ_MAX_MCU_BLOCKS]; // per MCU block: byte offsets of its DC | AC << 16 tables in s_tab
    __shared__ uint32_t s_pcomp[GJ_MAX_MCU_BLOCKS];
    __shared__ uint32_t s_sel[16];
    const int tid = threadIdx.x;
    static_assert(GJ_MAX_MCU_BLOCKS >= 16, "the sixteen byte selectors below are set up by the lanes that set up the MCU's blocks");
    if (tid < GJ_MAX_MCU_BLOCKS) {
        const int pp = tid < g.blocks_per_mcu ? tid : 0;
        const int c = INTERLEAVED ? g.mcu_comp[pp] : 0;
        s_ptab[tid] = (uint32_t)((g.comp[c].dc_table * 2 + 0) * GJ_DEC2_WORDS) | ((uint32_t)((g.comp[c].ac_table * 2 + 1) * GJ_DEC2_WORDS) << 16);
        s_pcomp[tid] = (uint32_t)c;
    }
    if (tid < 16) {
        uint32_t sel = 0x0C0C0C0Cu; // (0x0C selects a zero byte)
        int at = 3;
        for (int k = 0; k < 4; k++)
            if (tid & (1 << k)) { sel = (sel & ~(0xFFu << (8 * at))) | ((uint32_t)k << (8 * at)); at--; }
        s_sel[tid] = sel;
    }
    {
        const uint4* src = reinterpret_cast<const uint4*>(tabs);
        uint4* dst = reinterpret_cast<uint4*>(s_tab);
        for (int t = tid; t < 4 * GJ_DEC2_WORDS / 8; t += 256) dst[t] = src[t];
    }
    if (tid < 128) s_zz[tid] = tid < 64 ? GJ_ZZ[tid] : 0; // (behind a block's end, damaged streams only: a token there lands on position 0, which the IDCT overwrites
                                                          //  with the DC term, i.e. it is dropped like in the plane kernels; the plane path below asks pos < 64)
    const uint32_t* const end = reinterpret_cast<const uint32_t*>((reinterpret_cast<uintptr_t>(jpeg) + jpeg_size + 3) & ~(uintptr_t)3);
    const int seg_count = seg_count_ptr ? min((int)*seg_count_ptr, seg_count_max) : seg_count_max;
    const int s = (int)blockIdx.x * 256 + tid;
    uint32_t pos = 0, len = 0, idx = 0xFFFFFFFFu;
    if (s < seg_count) {
        idx = seg_index[s];
        pos = seg_pos[s];
        len = seg_len[s];
        if (idx >= (uint32_t)g.segment_count || (uint64_t)pos + len > jpeg_size) { idx = 0xFFFFFFFFu; len = 0; }
    }
    __syncthreads(); // (tables; nothing below crosses waves)
    bool active = idx != 0xFFFFFFFFu;
    GjSeg sg;
    sg.nblocks = 0; sg.comp = 0; sg.mcu_first = 0; sg.first_block = 0;
    if (active) sg = gj_segment(g, (int)idx);
    int left = active ? sg.nblocks : 0;
    active = left > 0;
    if (__ballot(active) == 0ull) return;
    const int P = g.blocks_per_mcu;
    uint32_t* const R = s_ring + tid * GJ_WIN_STRIDE;

    // ---- the writer: stuffed bytes taken, unstuffed bytes written, the dword being filled, its slot in the ring
    uint32_t src = 0, wr = 0, wacc = 0, wslot = 0;
    bool final = !active;
    // ---- the reader (as in gj_dec_entropy_tok.hip): rp = bit position - 1 inside the ring (0 .. 767, wraps), the next 32 bits are
    //      v_alignbit_b32(R[rp >> 5], R[(rp >> 5) + 1], ~rp) -- slot GJ_WIN_DW mirrors slot 0 --; `ahead` = bits between the bit position and the
    //      end of the complete dwords in the ring; a symbol is decoded while ahead > stop (stop = the zero tail once the segment's last
    //      byte has been written: then that is "bit position < end of the segment")
    uint32_t rp = 32u * GJ_WIN_DW - 1u;
    int ahead = 0, stop = 0;
    auto put = [&](const uint32_t v) { // a complete dword into the ring
        R[wslot] = v;
        if (wslot == 0) R[GJ_WIN_DW] = v;
        wslot = wslot + 1u == (uint32_t)GJ_WIN_DW ? 0u : wslot + 1u;
    };

    // s_sel[m]: byte selector (v_perm_b32) that moves the bytes of a dword whose bit k of m is set to the top of the result, first byte
    // of the stream in the most significant position; the others are zero
    auto refill = [&]() {
        // every lane takes as many bytes as its ring has room for (K), the wave walks through max K in 16-byte pieces; the loads of the
        // next piece are in flight while this one is worked on. In use: from the dword of the bit position to the last byte written
        const uint32_t room = 4u * GJ_WIN_DW - (((uint32_t)ahead + ((rp + 1u) & 31u)) >> 3) - (wr & 3u);
        uint32_t K = final ? 0u : min(len - src, room);
        // the segment's last byte among them: a partial dword and 8 zero bytes follow (12 bytes of room are kept for them, else next time)
        if (!final && K == len - src && room < K + 12u) K = K > 12u ? K - 12u : 0u;
        const bool ends_now = !final && K == len - src;
        const uintptr_t a = reinterpret_cast<uintptr_t>(jpeg) + pos + src;
        const uint32_t* base = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)15);
        uint32_t lo = (uint32_t)(a & 15), hi = lo + K; // the bytes [lo, hi) of the pieces from `base` on are this lane's
        src += K;
        const uint32_t wr0 = wr;
        uint32_t w[4], wn[4];
#pragma unroll
        for (int q = 0; q < 4; q++) w[q] = (hi > 0u && base + q < end) ? base[q] : 0u;
        // 0xFF marks (bit 7 per byte) of the dword in front of the first piece: its last byte decides about a zero at the start of the piece
        // (all bytes in front of `lo` are the segment's own, taken earlier, or what precedes the segment: the end of a marker, never 0xFF)
        uint32_t ffcarry = 0;
        if (hi > 0u && reinterpret_cast<uintptr_t>(base) > reinterpret_cast<uintptr_t>(jpeg) + 4u) {
            const uint32_t x = base[-1];
            ffcarry = ((x & 0x7F7F7F7Fu) + 0x01010101u) & x & 0x80808080u;
        }
        while (__ballot(hi > lo)) {
#pragma unroll
            for (int q = 0; q < 4; q++) wn[q] = (hi > 16u && base + 4 + q < end) ? base[4 + q] : 0u;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t x = w[q];
                const uint32_t eq = ((x & 0x7F7F7F7Fu) + 0x01010101u) & x & 0x80808080u;         // bit 7 of the bytes that are 0xFF
                const uint32_t zr = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;      // ... of the bytes that are 0x00
                const uint32_t before = __builtin_amdgcn_alignbit(eq, ffcarry, 24);               // ... of the bytes behind a 0xFF
                ffcarry = eq;
                // this lane's bytes of the dword: [lo, hi) cut to [4q, 4q + 4)
                const uint32_t b0 = min(max(lo, 4u * q), 4u * q + 4u) - 4u * q, b1 = min(max(hi, 4u * q), 4u * q + 4u) - 4u * q;
                const uint32_t below0 = b0 >= 4u ? 0xFFFFFFFFu : (1u << (8u * b0)) - 1u, below1 = b1 >= 4u ? 0xFFFFFFFFu : (1u << (8u * b1)) - 1u;
                const uint32_t mine = below1 & ~below0;
                const uint32_t keep = mine & ~(before & zr) & 0x80808080u; // (a zero behind 0xFF is stuffing)
                const uint32_t m = (((keep >> 7) * 0x01020408u) >> 24) & 15u;
                const uint32_t c = (uint32_t)__popc(keep);
                const uint32_t comp = __builtin_amdgcn_perm(0u, x, s_sel[m]);
                // append the c bytes of `comp` (left aligned) to the dword being filled
                const uint32_t sh = 8u * (wr & 3u);
                const uint32_t top = wacc | (comp >> sh), rest = __builtin_amdgcn_alignbit(comp, 0u, sh); // (sh = 0: rest = 0)
                const bool full = (wr & 3u) + c >= 4u;
                if (full) put(top);
                wacc = full ? rest : top;
                wr += c;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) w[q] = wn[q];
            lo = lo > 16u ? lo - 16u : 0u;
            hi = hi > 16u ? hi - 16u : 0u;
            base += 4;
        }
        if (ends_now) { // the partial dword and 8 zero bytes (a symbol has at most 27 bits)
            const uint32_t real = wr;
            put(wacc);
            put(0u);
            put(0u);
            wr = ((wr + 3u) & ~3u) + 8u;
            stop = (int)((wr - real) * 8u);
            final = true;
        }
        ahead += (int)(((wr & ~3u) - (wr0 & ~3u)) * 8u);
    };

    refill(); // the first fill

    // ---- decode: src/gpujpeg_huffman_gpu_decoder.cu:397-495 / src/gpujpeg_huffman_cpu_decoder.c:245-372
    int p = 0, comp = INTERLEAVED ? (int)s_pcomp[0] : sg.comp;
    uint32_t tdc, tac; // byte offsets of the block's tables in s_tab
    {
        const uint32_t pt = INTERLEAVED ? s_ptab[0] : ((uint32_t)((g.comp[comp].dc_table * 2 + 0) * GJ_DEC2_WORDS) | ((uint32_t)((g.comp[comp].ac_table * 2 + 1) * GJ_DEC2_WORDS) << 16));
        tdc = (pt & 0xFFFFu) * 2u;
        tac = (pt >> 16) * 2u;
    }
    int dc0 = 0, dc1 = 0, dc2 = 0, dc3 = 0;
    uint32_t z = 0, toff = tdc;
    // Output: the segment's tokens form a run inside its own part of the token array (4 tokens per stream byte: a token takes at least
    // 3 bits) that starts on a 32-byte boundary; sixteen tokens, and the records of four blocks, are collected in LDS and leave as whole
    // aligned 32-byte pieces -- 8-byte stores from 172 800 lanes at once kept two partial lines per lane open in the L2 and wrote 670 MB
    // for 86 MB of tokens and records. (A segment too short to leave room for the alignment keeps the 8-byte boundary it starts on.)
    const uint32_t tbase = len >= 6u ? (4u * pos + 15u) & ~15u : 4u * pos;
    bool tok_ok = 4u * pos <= tok_cap && 4u * len + 8u <= tok_cap - 4u * pos; // (always, with the capacity the host allocates)
    // tokens the run may hold: the segment's share is 4 per stream byte (+ the restart marker's), of which the alignment took up to 15. A token
    // costs 3 bits of stream with the standard tables, but a file with optimised tables may spend 2 (a 1-bit AC code + a magnitude bit):
    // a run that would leave its share raises the overflow flag -- the host decodes the frame through the planes -- and stores nothing more
    const uint32_t room = 4u * pos + 4u * len + 8u - tbase;
    uint32_t* const OT = s_out + tid * GJ_WIN_OUT;
    uint16_t* const OT16 = reinterpret_cast<uint16_t*>(OT);
    uint32_t ntok = 0, blk_first = 0, blk_dc = 0, mx = 0;
    uint32_t rec = INTERLEAVED ? (uint32_t)sg.mcu_first * (uint32_t)P : (uint32_t)sg.first_block, nrec = 0;
    const uint8_t* const tab8 = reinterpret_cast<const uint8_t*>(s_tab);
    while (__ballot(active)) {
        if (__ballot(active && !final && ahead < GJ_WIN_AHEAD)) refill();
        if (active) {
            uint32_t adv = 64, e = 0, sz = 0; // (data exhausted: the block ends here, its remaining coefficients stay zero)
            int v = 0;
            if (ahead > stop) {
                const uint32_t wi = rp >> 5;
                const uint32_t win = __builtin_amdgcn_alignbit(R[wi], R[wi + 1], ~rp); // the next 32 bits of the segment
                const uint16_t* t = reinterpret_cast<const uint16_t*>(tab8 + toff);
                e = t[gj_bfe_u32<32 - GJ_DEC_FAST_BITS, GJ_DEC_FAST_BITS>(win)];
                if ((e & 31u) == 0) e = t[(e >> 5) + ((win >> 16) & 63u)]; // codes longer than 10 bits (the host rejects streams whose scans
                                                                            // refer to tables they never defined: no entry is 0 after this)
                const uint32_t tot = e & 31u;
                sz = (e >> 5) & 15u;
                adv = (e >> 9) & 63u;
                // the sz magnitude bits behind the code, extended (ITU T.81 F.2.2.1), as in gj_tok_decode; meaningless without magnitude bits
                const uint32_t x = win << (tot - sz);
                uint32_t neg = (uint32_t)((int32_t)~x >> 31);
                GJ_KEEP(neg);
                const uint32_t mag = (x ^ neg) >> ((32u - sz) & 31u);
                v = (int)((mag ^ neg) - neg);
                rp += tot;
                rp = min(rp, rp - 32u * GJ_WIN_DW); // (wraps: the smaller of the two is the one inside the ring)
                ahead -= (int)tot;
            }
            if ((int16_t)e < 0) { // a non-zero AC coefficient
                mx = max(mx, sz);
                const uint32_t tok = (((uint32_t)v << 6) | s_zz[z + adv - 1u]) & 0xFFFFu;
                OT16[ntok & 15u] = (uint16_t)tok;
                ntok++;
                if ((ntok & 15u) == 0 && tok_ok && ntok > room) {
                    tok_ok = false;
                    *overflow = 1u;
                }
                if ((ntok & 15u) == 0 && tok_ok) { // sixteen tokens: one 32-byte piece (16-byte aligned when the run starts on the segment's own boundary)
                    uint32_t* dst = reinterpret_cast<uint32_t*>(d_tok + tbase + ntok - 16u);
                    if ((tbase & 7u) == 0) {
                        gj_store16_stream(dst, OT[0], OT[1], OT[2], OT[3]);
                        gj_store16_stream(dst + 4, OT[4], OT[5], OT[6], OT[7]);
                    } else {
                        for (int q = 0; q < 4; q++) reinterpret_cast<uint2*>(dst)[q] = make_uint2(OT[2 * q], OT[2 * q + 1]);
                    }
                }
            } else if (z == 0) { // DC: predicted from the previous block of the component inside this segment
                const int d = sz ? v : 0;
                const bool c0 = !INTERLEAVED || comp == 0, c1 = comp == 1, c2 = comp == 2;
                const int pred = c0 ? dc0 : c1 ? dc1 : c2 ? dc2 : dc3;
                const int dc = d + pred;
                dc0 = c0 ? dc : dc0;
                if (INTERLEAVED) {
                    dc1 = c1 ? dc : dc1;
                    dc2 = c2 ? dc : dc2;
                    dc3 = (c0 || c1 || c2) ? dc3 : dc;
                }
                blk_dc = (uint32_t)dc;
            }
            z += adv;
            toff = tac;
            if (z >= 64u) { // next block of this segment: its record (where its tokens are, how many, the DC term)
                z = 0;
                left--;
                {
                    const uint32_t r0 = tok_ok ? tbase + blk_first : 0u, r1 = ((tok_ok ? min(ntok - blk_first, 63u) : 0u) << 16) | (blk_dc & 0xFFFFu);
                    const uint32_t slot = rec & 3u;
                    OT[8u + 2u * slot] = r0;
                    OT[9u + 2u * slot] = r1;
                    nrec++;
                    if (slot == 3u || left == 0) { // the group of four records ends (or the segment does): whole if all four are this lane's
                        if (nrec == 4u) {
                            uint32_t* dst = reinterpret_cast<uint32_t*>(d_rec + (rec & ~3u));
                            gj_store16_stream(dst, OT[8], OT[9], OT[10], OT[11]);
                            gj_store16_stream(dst + 4, OT[12], OT[13], OT[14], OT[15]);
                        } else {
                            for (uint32_t q = slot + 1u - nrec; q <= slot; q++) d_rec[(rec & ~3u) + q] = make_uint2(OT[8u + 2u * q], OT[9u + 2u * q]);
                        }
                        nrec = 0;
                    }
                }
                rec++;
                blk_first = ntok;
                blk_dc = 0;
                if (INTERLEAVED) {
                    if (++p == P) p = 0;
                    comp = (int)s_pcomp[p];
                    const uint32_t pt = s_ptab[p];
                    tdc = (pt & 0xFFFFu) * 2u;
                    tac = (pt >> 16) * 2u;
                }
                toff = tdc;
                if (left == 0) {
                    active = false;
                    if (tok_ok && ntok > room) {
                        tok_ok = false;
                        *overflow = 1u;
                    }
                    if (tok_ok && (ntok & 15u) != 0) { // the last one to fifteen tokens: as a whole piece when the segment's part of the array has room for it
                        if ((tbase & 7u) == 0 && tbase + ((ntok + 15u) & ~15u) <= 4u * pos + 4u * len + 8u) {
                            uint4* dst = reinterpret_cast<uint4*>(d_tok + tbase + (ntok & ~15u));
                            dst[0] = make_uint4(OT[0], OT[1], OT[2], OT[3]);
                            if ((ntok & 15u) > 8u) dst[1] = make_uint4(OT[4], OT[5], OT[6], OT[7]);
                        } else {
                            for (uint32_t r = ntok & 15u, i = 0; i < r; i++) d_tok[tbase + (ntok & ~15u) + i] = OT16[i];
                        }
                    }
                    if (mx >= 10u) *overflow = 1u; // a value beyond a token's 10 bits: the host decodes the frame again through the planes
                }
            }
        }
    }
}

void gj_launch_huffman_seq(const gj_dec_job* job, hipStream_t st, const bool tokens)
{
    const gj_geom& g = job->g;
#ifndef GJ_SEQ_NO_RING
    if (tokens) { // (token mode: the ring kernel; -DGJ_SEQ_NO_RING builds the A/B variant with the staged one)
        auto kernel = g.interleaved ? k_huffman_decode_win<true> : k_huffman_decode_win<false>;
        hipLaunchKernelGGL(kernel, dim3(((unsigned)job->seg_count + 255u) / 256u, 1, job->batch.count > 1 ? job->batch.count : 1u), dim3(256), 0, st, g, job->d_jpeg,
                           job->jpeg_size, job->d_seg_pos, job->d_seg_len, job->d_seg_index, job->seg_count, job->d_seg_count, job->d_huff_tab2, job->d_overflow,
                           (uint16_t*)job->d_tok, job->tok_cap, (uint2*)job->d_blkrec);
        return;
    }
#endif
    const unsigned avg = (unsigned)(job->jpeg_size / (uint64_t)job->seg_count) + 12u;
    const int NS = max(1, min(GJ_SEQ_NS, (int)((GJ_SEQ_STAGE * 7u / 8u) / avg)));
    auto kernel = tokens ? (g.interleaved ? k_huffman_decode_seq<true, true> : k_huffman_decode_seq<false, true>)
                         : (g.interleaved ? k_huffman_decode_seq<true, false> : k_huffman_decode_seq<false, false>);
    hipLaunchKernelGGL(kernel, dim3(((unsigned)job->seg_count + NS - 1) / NS), dim3(256), 0, st, g, job->d_jpeg, job->jpeg_size, job->d_seg_pos, job->d_seg_len,
                       job->d_seg_index, job->seg_count, job->d_seg_count, NS, job->d_huff_tab2, job->d_coefs, job->clear_coefs ? 0 : 1, job->d_overflow,
                       (uint16_t*)job->d_tok, job->tok_cap, (uint2*)job->d_blkrec);
}
