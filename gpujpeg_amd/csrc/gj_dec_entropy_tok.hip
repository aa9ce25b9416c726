// gj_dec_entropy_tok.hip -- MI355X (gfx950, wave64) JPEG decoder: sub-sequence parallel entropy decoding into TOKENS.
// (part of the decoder's device code, see gj_dec_internal.h for the map of the files)
//
// The default entropy decoder of large non-interleaved frames. A workgroup takes a batch of consecutive restart segments of one scan and
//   1. copies their bytes into LDS without the stuffed zeros and the restart markers -- the whole batch is ONE contiguous piece of the
//      stream, so all 256 lanes take an equal share of it (one prefix sum; segment boundaries fall out of the marker count),
//   2. cuts every segment into sub-sequences of 16 bytes; a lane starts 48 bits in front of its sub-sequence in an assumed state -- Huffman
//      codes self-synchronise, so it most likely enters its sub-sequence in the true state -- and decodes it; sub-sequences whose
//      predecessor leaves in another state than they were entered with go through rounds over dense work lists until nothing changes
//      (Klein & Wiseman 2003, Weissenberger & Schmidt 2021). These passes only count: blocks completed and non-zero AC coefficients,
//   3. turns the counts into block and token positions with one prefix sum,
//   4. decodes once more, now producing 16-bit TOKENS (value << 6 | natural position) for the non-zero AC coefficients. A wave stages the
//      tokens of its 64 sub-sequences in LDS and flushes them with whole 16-byte pieces: the token array is written in full lines,
//   5. resolves the DC prediction (wave prefix sum per segment) and writes one record per block: first token, count, DC term.
// The token-fed IDCT (gj_dec_idct.hip) rebuilds the blocks in LDS; the coefficient planes are never touched.
// What does not fit this scheme raises `overflow` and the host decodes the frame with k_huffman_decode_par through the planes: a segment
// longer than the LDS stage. A coefficient that does not fit the token's 10 value bits (|v| >= 512: DCT coefficients of 8-bit samples
// quantised with a step below 3, or a damaged stream) sends its batch through the planes inside this kernel (records say so).
// Results are identical to src/gpujpeg_huffman_gpu_decoder.cu:287-495 / src/gpujpeg_huffman_cpu_decoder.c:245-372.
#include "gj_dec_internal.h"
#include "gj_bitreader.h"

// LDS budget: 4 workgroups per CU need <= 40960 B (granted in steps of 1280 B, tools/ubench/lds_occupancy.hip), and an 8K frame has to be
// ONE generation of workgroups (a second, partial generation doubles the kernel's duration): hence the stage of 10.5 KB.
#ifndef GJ_TOK_SUB
#define GJ_TOK_SUB 16                                           // bytes per sub-sequence ...
#define GJ_TOK_SUB_MAX 20                                       // ... or up to so many when that saves the group a pass
#endif
// n segments of a group hold at most CAP_U - 8 n unstuffed bytes (8 B of padding each) and every one ends with a partial sub-sequence:
// (CAP_U - 8 n) / 16 + 15 n / 16 < CAP_U / 16 + n / 2 sub-sequences (this bound needs sub-sequences of at most 17 bytes; the table is
// clamped to it all the same)
#define GJ_TOK_MAX_SUBS (GJ_TOK_CAP_U / GJ_TOK_SUB + (GJ_TOK_SUB <= 17 ? GJ_TOK_GMAX / 2 : GJ_TOK_GMAX))
#define GJ_TOK_WSTAGE 944                                       // tokens a wave can stage per flush at least (incl. up to 7 of alignment)
// (Five workgroups per CU -- a 32 000-byte layout: stage 7.7 KB, 1728 blocks and 640..688 staged tokens per wave, 84 VGPRs -- was built and
// measured in round 4, profiles/r4_10_tok_five_per_cu.txt: 16K, many generations, 293 against 298 us with 688 tokens and 315 us with 640; 8K 93 us
// against 80 in one generation, 123 us when the batches tip into a second one; the kernel is not waiting for latency that a fifth workgroup hides.)
#define GJ_TOK_WG_PER_CU 4
#ifndef GJ_TOK_SYNC
#define GJ_TOK_SYNC 48                                          // bits in front of a sub-sequence its lane decodes first to fall into step
#endif
#define GJ_TOK_CHUNK_MAX 48                                     // bytes of the batch's stream per lane in the cooperative copy

// -DGJ_TRACE_PHASES (the `trace` target of the Makefile, tools/decoder_phases.py): the first work-item of every workgroup notes the wall
// clock (100 MHz) at the phase boundaries in a buffer the tool hands over
#ifdef GJ_TRACE_PHASES
static __device__ unsigned long long* gj_trace_buf;
extern "C" GJ_HIP_API int gj_hip_trace_set(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(gj_trace_buf), &p, sizeof p) == hipSuccess ? 0 : -1; }
#define GJ_TRACE(slot) do { if (threadIdx.x == 0 && gj_trace_buf) gj_trace_buf[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GJ_TRACE(slot) ((void)0)
#endif

// -DGJ_TOK_STATS (CPU execution model only, tools/tok_sync_stats.py): how many sub-sequences every round has to decode again
#ifdef GJ_TOK_STATS
extern "C" GJ_HIP_API unsigned long long gj_tok_stats[64];
unsigned long long gj_tok_stats[64];
// lane utilisation of a pass: every lane notes its symbols (tokens + DC and end-of-block per block), lane 0 adds the wave's sum and 64 x max
#define GJ_STAT_WAVE(sm, i, lane, sym) do { gj_wave_sync(); (sm).statsym[threadIdx.x] = (sym); gj_wave_sync(); if ((lane) == 0) { \
    unsigned long long s_ = 0, m_ = 0; for (int q_ = 0; q_ < 64; q_++) { const unsigned v_ = (sm).statsym[(threadIdx.x & ~63) + q_]; s_ += v_; if (v_ > m_) m_ = v_; } \
    GJ_STAT(i, s_); GJ_STAT((i) + 1, 64 * m_); GJ_STAT((i) + 2, 1); } } while (0)
#define GJ_SYMS(c) (((c) >> 16) + 2u * ((c) & 0xFFFFu))
#define GJ_STAT(i, n) __atomic_fetch_add(&gj_tok_stats[i], (unsigned long long)(n), __ATOMIC_RELAXED)
#else
#define GJ_STAT(i, n) ((void)0)
#define GJ_STAT_WAVE(sm, i, lane, sym) ((void)0)
#endif

// The reader's window. GJ_TOK_WIN 0: both dwords of the window are read from the stage for every symbol (ds_read2_b32 in front of the look-up, on the
// symbol's dependent chain). GJ_TOK_WIN 1 (round 6 experiment): the window lives in two registers; the dword BEHIND it is asked for at the top of every
// iteration -- its latency hides under the table look-up -- and moves in when the position crosses a dword (a symbol has at most 26 bits: one step).
#ifndef GJ_TOK_WIN
#define GJ_TOK_WIN 1
#endif
#if GJ_TOK_WIN == 1
#define GJ_TOK_WIN_INIT(sm, p1) uint32_t wcur_ = (p1) >> 5, w0_ = (sm).U[wcur_], w1_ = (sm).U[wcur_ + 1], wnx_ = 0
#define GJ_TOK_WIN_BITS(sm, p1) wnx_ = (sm).U[wcur_ + 2]; const uint32_t win = __builtin_amdgcn_alignbit(w0_, w1_, ~(p1))
#define GJ_TOK_WIN_STEP(p1) do { const uint32_t wn_ = (p1) >> 5; const bool cross_ = wn_ != wcur_; w0_ = cross_ ? w1_ : w0_; w1_ = cross_ ? wnx_ : w1_; wcur_ = wn_; } while (0)
#else
#define GJ_TOK_WIN_INIT(sm, p1) ((void)0)
#define GJ_TOK_WIN_BITS(sm, p1) const uint32_t wi = (p1) >> 5; const uint32_t win = __builtin_amdgcn_alignbit((sm).U[wi], (sm).U[wi + 1], ~(p1))
#define GJ_TOK_WIN_STEP(p1) ((void)0)
#endif

// state between two symbols: bits [0,5) overshoot into the next sub-sequence, [5,11) zig-zag index
// counts of a sub-sequence: bits [0,16) blocks completed, [16,32) tokens

// The workgroup's LDS as ONE object with the stage in front: the stage then sits at LDS address 0, and a bit position turns into a
// dword address with a shift and a mask (with separate arrays the compiler adds the array's base in every symbol).
struct GjTokLds {
    // bytes of unstuffed stream per group (big-endian dwords). Dword 0 is not used: the bit reader addresses the dword of "bit position
    // - 1", see gj_tok_decode; the unstuffed bytes start at dword 1 = bit 32
    uint32_t U[GJ_TOK_CAP_U / 4 + 8];
    __attribute__((aligned(16))) uint16_t tab[2 * GJ_DEC2_WORDS]; // DC table, AC table of the group's component
    __attribute__((aligned(8))) uint2 rec[GJ_TOK_MAX_SUBS];       // per sub-sequence: entry | exit << 11 | segment << 22, counts (then their prefix sums)
    // one pool for the block slots of the batch (per block + one per segment: DC difference | first token << 16) and, behind them, the four
    // waves' token stages: a batch of luminance segments has few blocks and many tokens per sub-sequence, a chrominance batch the opposite
    __attribute__((aligned(16))) uint16_t pool[(GJ_TOK_MAX_BLOCKS + GJ_TOK_GMAX) * 2 + 4 * GJ_TOK_WSTAGE];
    // per segment of the batch: first slot in the pool, capacity offsets (which segments fit the stage together), first record, tables
    uint32_t bb[GJ_TOK_GMAX + 1], cap[GJ_TOK_GMAX + 1], first[GJ_TOK_GMAX];
    uint16_t tabsel[GJ_TOK_GMAX];
    // per segment of the group: first and end bit in the stage, first sub-sequence
    uint32_t sbit[GJ_TOK_GMAX], ebit[GJ_TOK_GMAX], sub0[GJ_TOK_GMAX + 1];
    uint32_t tmp[4];
    uint32_t nwork[2], big;
    uint32_t gpos; // stream position of the group's first segment that has blocks: where the group's token run starts (x 4)
    uint8_t zz[64 + 64];
#ifdef GJ_TOK_STATS
    uint32_t statsym[256];
#endif
};

// The run-in of the first pass: decode from bit `from` (most likely inside a block, hence the AC table) up to the sub-sequence that starts
// at `start_bit`; returns the state there. Huffman codes self-synchronise: after a few symbols this decoding has fallen into step
// with the true one, whatever it started with.
__device__ __forceinline__ uint32_t gj_tok_run_in(const GjTokLds& sm, const uint32_t from, const uint32_t start_bit)
{
    uint32_t p1 = from - 1u;
    const uint32_t e1 = start_bit - 1u;
    uint32_t z = 1;
    uint32_t toff = (uint32_t)GJ_DEC2_WORDS * 2u;
    GJ_TOK_WIN_INIT(sm, p1);
    while (p1 < e1) {
        GJ_TOK_WIN_BITS(sm, p1);
        const uint16_t* t = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(sm.tab) + toff);
        uint32_t e = t[gj_bfe_u32<32 - GJ_DEC_FAST_BITS, GJ_DEC_FAST_BITS>(win)];
        if ((e & 31u) == 0) e = t[(e >> 5) + ((win >> 16) & 63u)];
        p1 += e & 31u;
        GJ_TOK_WIN_STEP(p1);
        z += (e >> 9) & 63u;
        const uint32_t inside = (uint32_t)((int32_t)(z - 64u) >> 31); // all ones until the block is complete
        z &= inside;
        toff = ((uint32_t)GJ_DEC2_WORDS * 2u) & inside;
    }
    return (p1 - e1) | (z << 5);
}

// One pass over a sub-sequence. MODE 0: count. MODE 1: tokens into the wave's LDS stage, DC differences + token positions of the blocks
// into the block slots. MODE 2: coefficients into the planes (batches with a coefficient beyond the token range).
// The bit reader keeps p1 = bit position - 1: the 32 bits at position p1 + 1 are the dword pair (U[p1 >> 5], U[(p1 >> 5) + 1]) shifted right
// by 31 - (p1 & 31) = ~p1 & 31, which is one v_alignbit_b32 (the pair of "bit position" would need a shift by 32 when the position is
// dword aligned, which the instruction does not have).
template <int MODE>
__device__ __forceinline__ uint32_t gj_tok_decode(const GjTokLds& sm, const uint32_t start_bit, const uint32_t end_bit, const uint32_t entry, uint32_t& counts,
                                                  uint16_t* __restrict__ tok_out /* MODE 1: where this sub-sequence's tokens go in the stage */,
                                                  uint32_t* __restrict__ s_blkinfo /* MODE 1, 2: slots of the segment's blocks (+ 1) */,
                                                  const uint32_t tok_rel /* tokens of the group in front of this sub-sequence */, uint32_t blk,
                                                  const uint32_t nblocks, int16_t* __restrict__ coefs /* MODE 2: first block of the segment */,
                                                  uint32_t* __restrict__ s_big)
{
    uint32_t p1 = start_bit + (entry & 31u) - 1u;
    const uint32_t e1 = end_bit - 1u;
    uint32_t z = (entry >> 5) & 63u;
    uint32_t toff = z == 0 ? 0u : (uint32_t)GJ_DEC2_WORDS * 2u; // byte offset of the table: DC in front of a block, AC inside
    uint32_t nb = 0, ntok = 0, mx = 0;
    uint16_t* tp = tok_out; // MODE 1: where the next token goes (ntok follows from it at the end)
    GJ_TOK_WIN_INIT(sm, p1);
    while (p1 < e1) {
        GJ_TOK_WIN_BITS(sm, p1); // win: the next 32 bits of the stream
        const uint16_t* t = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(sm.tab) + toff);
        uint32_t e = t[gj_bfe_u32<32 - GJ_DEC_FAST_BITS, GJ_DEC_FAST_BITS>(win)];
        if ((e & 31u) == 0) e = t[(e >> 5) + ((win >> 16) & 63u)]; // codes longer than 10 bits
        const uint32_t tot = e & 31u, adv = (e >> 9) & 63u, sz = (e >> 5) & 15u;
        if (MODE == 0) {
            ntok += e >> 15; // a non-zero AC coefficient
        } else {
            // the sz magnitude bits behind the code, extended (ITU T.81 F.2.2.1); without magnitude bits the shifts wrap and v is
            // meaningless: no coefficient is made of it, and the DC branch asks
            const uint32_t x = win << (tot - sz);                       // magnitude bits, left aligned
            uint32_t neg = (uint32_t)((int32_t)~x >> 31);               // all ones when the first of them is 0: a negative value, whose
            GJ_KEEP(neg);                                               // magnitude is the complement of the bits (kept as arithmetic:
            const uint32_t mag = (x ^ neg) >> ((32u - sz) & 31u);       // the compiler would turn it into compare + 2 selects, which
            const int v = (int)((mag ^ neg) - neg);                     // issue at half the rate of these)
            if ((int16_t)e < 0) { // a non-zero AC coefficient
                const uint32_t pos = z + adv - 1u; // (zz[64..127] = 0: damaged streams only, see there)
                if (MODE == 1) {
                    mx = max(mx, sz);
                    *tp++ = (uint16_t)(((uint32_t)v << 6) | sm.zz[pos]);
                } else {
                    if (blk + nb < nblocks && pos < 64u) coefs[(uint64_t)(blk + nb) * 64 + sm.zz[pos]] = (int16_t)v;
                    ntok++;
                }
            } else if (z == 0) {
                // slot blk + nb of the segment: DC difference and where the block's tokens start; the slot behind the last block takes the
                // start of a block the segment should not have (damaged stream): it ends the last block's tokens
                const uint32_t t = MODE == 1 ? tok_rel + (uint32_t)(tp - tok_out) : tok_rel + ntok;
                if (blk + nb <= nblocks) s_blkinfo[blk + nb] = (sz ? (uint32_t)v & 0xFFFFu : 0u) | (t << 16);
            }
        }
        p1 += tot;
        GJ_TOK_WIN_STEP(p1);
        z += adv;
        const uint32_t inside = (uint32_t)((int32_t)(z - 64u) >> 31); // all ones until the block is complete
        z &= inside;
        toff = ((uint32_t)GJ_DEC2_WORDS * 2u) & inside;
        nb += 1u + inside;
    }
    if (MODE == 1) ntok = (uint32_t)(tp - tok_out);
    if (MODE == 1 && mx >= 10u) *s_big = 1u; // a value that does not fit a token's 10 bits: the batch goes through the planes
    counts = nb | (ntok << 16);
    return (p1 - e1) | (z << 5);
}

template <bool COOP>
__global__ __launch_bounds__(256, GJ_TOK_WG_PER_CU) void k_huffman_decode_tok(const gj_geom g, const uint8_t* __restrict__ jpeg, uint64_t jpeg_size,
                                                               const uint32_t* __restrict__ seg_pos, const uint32_t* __restrict__ seg_len,
                                                               const uint32_t* __restrict__ seg_index, const int seg_count_max,
                                                               const uint32_t* __restrict__ seg_count_ptr, const GjBatchPlan plan,
                                                               const uint16_t* __restrict__ tabs, int16_t* __restrict__ coefs,
                                                               uint16_t* __restrict__ d_tok, const uint32_t tok_cap, uint2* __restrict__ d_rec,
                                                               uint32_t* __restrict__ overflow, const GjFold F)
{
    constexpr int CAP_U = GJ_TOK_CAP_U, MAX_BLOCKS = GJ_TOK_MAX_BLOCKS, GMAX = GJ_TOK_GMAX, MAX_SUBS = GJ_TOK_MAX_SUBS;
    if (g.fb.sizes != nullptr) { // frame blockIdx.z of a batch: its stream, its table, its summary words, its tokens and records
        const size_t z = blockIdx.z;
        jpeg += z * g.fb.jpeg;
        jpeg_size = g.fb.sizes[z];
        seg_pos += z * g.fb.seg; seg_len += z * g.fb.seg; seg_index += z * g.fb.seg;
        if (seg_count_ptr) seg_count_ptr += z * (sizeof(gj_scan_summary) / 4);
        overflow += z * (sizeof(gj_scan_summary) / 4);
        coefs += z * g.fb.coefs; d_tok += z * g.fb.tok; d_rec += z * g.fb.rec;
    }
    __shared__ GjTokLds sm;
    constexpr int POOL = sizeof(sm.pool) / 2; // 16-bit units
    uint32_t* const s_stage = sm.U + 1;      // where the unstuffed bytes go: bit position 32 of the reader
    uint16_t* const s_pool = sm.pool;
    uint32_t* const s_blkinfo = reinterpret_cast<uint32_t*>(sm.pool);
    uint2* const s_rec = sm.rec;
    uint32_t *const s_bb = sm.bb, *const s_cap = sm.cap, *const s_first = sm.first, *const s_sbit = sm.sbit, *const s_ebit = sm.ebit, *const s_sub0 = sm.sub0;
    uint32_t* const s_tmp = sm.tmp;
    // stream positions and lengths are needed until the stage is filled: they borrow the end of the pool (the last wave's token stage)
    constexpr int BORROW = ((4 * GMAX + 1) * 2 + 7) & ~7;
    uint32_t* const s_pos = reinterpret_cast<uint32_t*>(s_pool + POOL - BORROW);
    uint32_t* const s_len = s_pos + GMAX;
    uint32_t* const s_ubyte = s_len + GMAX;     // [GMAX + 1] byte offset of every segment in the stage
    uint32_t* const s_ulen = s_ubyte + GMAX + 1; // [GMAX] its unstuffed length
    static_assert(BORROW <= GJ_TOK_WSTAGE && MAX_SUBS <= POOL - BORROW, "borrowed space");
    uint16_t* const s_work = s_pool; // (the work list of the rounds: before the block slots are used)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    GJ_TRACE(0);
    // v_perm_b32 selector of lane & 15 read as a set of kept bytes: they move to the top of the result, first byte in the most significant
    // position, zeros (0x0C) below (the cooperative copy fetches the one it needs from lanes 0..15 with ds_bpermute_b32)
    uint32_t sel_of_lane = 0x0C0C0C0Cu;
    {
        int at = 3;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (lane & (1 << k)) { sel_of_lane = (sel_of_lane & ~(0xFFu << (8 * at))) | ((uint32_t)k << (8 * at)); at--; }
    }
    if (tid < 128) sm.zz[tid] = tid < 64 ? GJ_ZZ[tid] : 0; // (behind a block's end, damaged streams only: position 0, which the IDCT overwrites with the DC term -- the
                                                            //  coefficient is dropped, as the plane kernels and the reference's GPU decoder do, src/gpujpeg_huffman_gpu_decoder.cu:370)
    if (tid == 0) sm.U[0] = 0;
    const uint32_t* end = reinterpret_cast<const uint32_t*>((reinterpret_cast<uintptr_t>(jpeg) + jpeg_size + 3) & ~(uintptr_t)3);

    // ---- batch setup: lane j describes segment j of the batch
    int pc = 0;
    while (pc + 1 < plan.n && (int)blockIdx.x >= plan.batch0[pc + 1]) pc++;
    const int G = plan.g[pc];
    const int si0 = plan.first[pc] + ((int)blockIdx.x - plan.batch0[pc]) * G;
    // (the segment table is asked for before the segment count is known: one trip to memory for the four loads)
    uint32_t ld_s = 0xFFFFFFFFu, ld_p = 0, ld_l = 0;
    if (F.recs == nullptr && tid < G && si0 + tid < seg_count_max) { ld_s = seg_index[si0 + tid]; ld_p = seg_pos[si0 + tid]; ld_l = seg_len[si0 + tid]; }
    const int seg_count = seg_count_ptr ? min((int)*seg_count_ptr, seg_count_max) : seg_count_max;
    if (si0 >= seg_count) return;
    const int nseg = min(min(G, plan.first[pc] + plan.count[pc] - si0), seg_count - si0);
    if (F.recs != nullptr) { // the batch's table entries from the marker scan's records (gj_fold_batch, gj_dec_internal.h); scratch: the start of the pool, free until the groups are formed
        static_assert(GJ_FOLD_SCRATCH_WORDS(GJ_TOK_GMAX) * 2 <= (GJ_TOK_MAX_BLOCKS + GJ_TOK_GMAX) * 2, "fold scratch inside the block slots");
        if (!gj_fold_batch<GJ_TOK_GMAX>(F, g, jpeg, jpeg_size, plan, pc, si0, nseg, reinterpret_cast<uint32_t*>(sm.pool), s_tmp, ld_s, ld_p, ld_l)) return;
    }
    // the Huffman tables of the batch's scan are on their way while the segment table is read (a batch cut per scan belongs to component
    // `pc`; otherwise, or when a segment says something else, they are loaded where the groups are formed)
    uint32_t loaded_tabs = 0xFFFFFFFFu;
    auto load_tables = [&](const uint32_t tb) {
        const uint4* src0 = reinterpret_cast<const uint4*>(tabs + (tb & 0xFFu) * GJ_DEC2_WORDS);
        const uint4* src1 = reinterpret_cast<const uint4*>(tabs + (tb >> 8) * GJ_DEC2_WORDS);
        uint4* dst = reinterpret_cast<uint4*>(sm.tab);
        for (int t = tid; t < GJ_DEC2_WORDS / 8; t += 256) { dst[t] = src0[t]; dst[GJ_DEC2_WORDS / 8 + t] = src1[t]; }
        loaded_tabs = tb;
    };
    if (plan.n > 1 && plan.n == g.comp_count) load_tables((uint32_t)(g.comp[pc].dc_table * 2 + 0) | ((uint32_t)(g.comp[pc].ac_table * 2 + 1) << 8));
    // (kept in registers of lanes 0 .. nseg - 1 for the groups: position, length, blocks)
    uint32_t my_pos = 0, my_len = 0, my_nblk = 0;
    {
        uint32_t first = 0, tb = 0;
        if (tid < nseg) {
            const uint32_t s = ld_s, p_ = ld_p, l_ = ld_l;
            if (s < (uint32_t)g.segment_count) {
                const GjSeg sg = gj_segment(g, (int)s);
                const gj_comp_geom& kc = g.comp[sg.comp];
                my_nblk = (uint32_t)sg.nblocks;
                my_pos = p_;
                my_len = l_;
                first = (uint32_t)(kc.data_offset / 64) + (uint32_t)sg.mcu_first; // first block: record index = block index of the planes
                tb = (uint32_t)(kc.dc_table * 2 + 0) | ((uint32_t)(kc.ac_table * 2 + 1) << 8);
                if (((my_len + 3u) & ~3u) + 8u > (uint32_t)CAP_U || my_nblk > (uint32_t)MAX_BLOCKS) { // not for this kernel: the host decodes the frame again
                    *overflow = 1u;
                    my_len = 0;
                    my_nblk = 0;
                }
            }
        }
        if (tid < GMAX) {
            s_first[tid] = first;
            sm.tabsel[tid] = (uint16_t)tb;
        }
        if (wave == 0) { // (a batch has at most GMAX = 64 segments: its table is the business of one wave)
            s_bb[tid + 1] = gj_wave_incl_scan(tid < nseg ? my_nblk + 1u : 0u); // (one slot more per segment, see gj_tok_decode)
            s_cap[tid + 1] = gj_wave_incl_scan(my_len ? ((my_len + 3u) & ~3u) + 8u : 0u);
        }
        if (tid == 0) { s_bb[0] = 0; s_cap[0] = 0; sm.nwork[0] = 0; sm.nwork[1] = 0; }
    }
    static_assert(GMAX == 64, "one wave holds the batch's segment table");
    __syncthreads();
    // the waves' token stages: what the batch's block slots leave of the pool, in four equal 16-byte aligned parts
    const uint32_t stage0 = (2u * s_bb[nseg] + 7u) & ~7u, stage_cap = (((uint32_t)POOL - stage0) >> 2) & ~7u;
    uint16_t* const stage = s_pool + stage0 + (uint32_t)wave * stage_cap;

    // ---- groups: consecutive segments with the same Huffman tables whose unstuffed bytes fit the LDS stage (normally one group = the batch)
    for (int j0 = 0; j0 < nseg;) {
        if (tid == 0) sm.big = 0;
        if (tid < nseg) { s_pos[tid] = my_pos; s_len[tid] = my_len; }
        { // where the group's token run starts: the position of its first segment that has blocks (segments without blocks carry no position)
            const unsigned long long mb = __ballot(lane >= j0 && lane < nseg && my_nblk != 0u);
            if (wave == 0 && lane == (mb ? (int)__builtin_ctzll(mb) : j0)) sm.gpos = my_pos;
        }
        // (every wave works the group out for itself from the batch's table: lane l looks at segment l, then at the end l + 1)
        const uint32_t tb = sm.tabsel[j0];
        const bool other = lane > j0 && lane < nseg && s_bb[lane + 1] - s_bb[lane] != 1u && sm.tabsel[lane] != tb; // other tables (the next scan) end the group
        const unsigned long long mo = __ballot(other);
        const int jstop = mo ? (int)__builtin_ctzll(mo) : nseg;
        const unsigned long long mf = __ballot(lane + 1 > j0 && lane + 1 <= jstop && s_cap[lane + 1] - s_cap[j0] <= (uint32_t)CAP_U);
        const int j1 = max(j0 + 1, mf ? 64 - (int)__builtin_clzll(mf) : 0);
        const int ng = j1 - j0;
        if (COOP) { // the cooperative copy ORs its bytes into the stage: zero what the group can use, and the 8 + 8 bytes the reader may look at behind it
            const uint32_t n16 = min((s_cap[j1] - s_cap[j0] + 16u + 4u + 15u) >> 4, (uint32_t)(sizeof(sm.U) / 16));
            uint4* const z = reinterpret_cast<uint4*>(sm.U);
            for (uint32_t t = (uint32_t)tid; t < n16; t += 256u) z[t] = uint4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
        GJ_TRACE(1);

        // -- 0. the group's Huffman tables
        if (tb != loaded_tabs) load_tables(tb);

        // -- 1. the group's bytes without stuffing and markers. Layout of the stage: segment j at byte s_ubyte[j] (big-endian dwords), 8 zero
        //       bytes behind every segment (a symbol that straddles the end reads zeros, src/gpujpeg_huffman_cpu_decoder.c:80-118)
        bool coop = false;
        if (COOP) {
            // regular: every segment has data and is followed, two bytes later, by the next one (what a scan of restart intervals looks like)
            bool irregular = false;
            if (lane < ng) {
                const int j = j0 + lane;
                irregular = s_len[j] == 0 || (lane + 1 < ng && s_pos[j + 1] != s_pos[j] + s_len[j] + 2u);
            }
            const uint32_t A = s_pos[j0], total = s_pos[j1 - 1] + s_len[j1 - 1] - A;
            coop = __ballot(irregular) == 0ull && total <= 256u * GJ_TOK_CHUNK_MAX && A >= 8u && (uint64_t)A + total <= jpeg_size;
        }
        if (coop) {
            const uint32_t A = s_pos[j0], total = s_pos[j1 - 1] + s_len[j1 - 1] - A;
            const uint32_t C = (((total + 255u) >> 8) + 3u) & ~3u; // bytes per lane, a multiple of 4
            const uint32_t off0 = (uint32_t)tid * C;                // this lane's first byte, relative to A
            const uintptr_t a = reinterpret_cast<uintptr_t>(jpeg) + A + off0;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3) - 1;
            const uint32_t lead = (uint32_t)((reinterpret_cast<uintptr_t>(jpeg) + A) & 3); // (C is a multiple of 4: the same for every lane)
            // window: the dword in front of the lane's first byte ... the dword behind its last one
            uint32_t win[GJ_TOK_CHUNK_MAX / 4 + 3];
#pragma unroll
            for (int i = 0; i < GJ_TOK_CHUNK_MAX / 4 + 3; i++) {
                win[i] = 0;
                if ((uint32_t)i < C / 4 + 3u && off0 < total + 4u && src + i < end) win[i] = src[i];
            }
            // Byte classes four at a time (SWAR on bit 7 of every byte): 0xFF, 0x00, second byte of a restart marker ((b & 0xF8) == 0xD0
            // behind 0xFF). Dropped: stuffed zeros and both marker bytes. `keep` / `mark`: bit b = byte b of the share is kept / starts a marker.
            // (Round 3 classified byte by byte -- ~20 instructions per byte -- and stored the kept bytes one ds_write_b8 each: 13 % of the kernel.)
            constexpr int ND = GJ_TOK_CHUNK_MAX / 4;
            uint64_t keep = 0, mark = 0;
            {
                const uint32_t K7 = 0x7F7F7F7Fu, K8 = 0x80808080u;
                uint32_t eq_prev = 0, m2_prev = 0, st_prev = 0;
                uint32_t ffc = 0; // 0xFF marks of the dword in front (only its last byte matters)
                if (off0 != 0) {
                    const uint32_t x = __builtin_amdgcn_alignbyte(win[1], win[0], lead);
                    ffc = ((x & K7) + 0x01010101u) & x & K8;
                }
#pragma unroll
                for (int i = 0; i <= ND; i++) { // (one dword of look-ahead: a marker may start in the last byte of a dword)
                    if ((uint32_t)(4 * i) <= C) {
                        const uint32_t x = __builtin_amdgcn_alignbyte(win[i + 2], win[i + 1], lead); // bytes 4i .. 4i + 3 of the share
                        const uint32_t t = x & K7;
                        const uint32_t eq = (t + 0x01010101u) & x & K8;
                        const uint32_t zr = ~((t + K7) | x) & K8;
                        const uint32_t y = (x ^ 0xD0D0D0D0u) & 0xF8F8F8F8u;
                        const uint32_t rz = ~(((y & K7) + K7) | y) & K8;
                        const uint32_t before = __builtin_amdgcn_alignbit(eq, ffc, 24); // bytes behind a 0xFF
                        ffc = eq;
                        const uint32_t m2 = before & rz, st = before & zr;
                        if (i > 0) { // dword i - 1 is complete now
                            const uint32_t m1 = eq_prev & __builtin_amdgcn_alignbit(m2, m2_prev, 8); // 0xFF whose next byte is the marker's second
                            const uint32_t kp = K8 & ~(st_prev | m2_prev | m1);
                            keep |= (uint64_t)((((kp >> 7) * 0x01020408u) >> 24) & 15u) << (4 * (i - 1));
                            mark |= (uint64_t)((((m1 >> 7) * 0x01020408u) >> 24) & 15u) << (4 * (i - 1));
                        }
                        eq_prev = eq; m2_prev = m2; st_prev = st;
                    }
                }
                const uint32_t nvalid = off0 >= total ? 0u : min(total - off0, C);
                const uint64_t valid = nvalid >= 64u ? ~0ull : (1ull << nvalid) - 1ull;
                keep &= valid;
                mark &= valid;
            }
            const uint32_t nkeep = (uint32_t)__popcll(keep), nmark = (uint32_t)__popcll(mark);
            uint32_t tot;
            const uint32_t inc = gj_wg256_incl_scan(nkeep | (nmark << 16), s_tmp, &tot);
            uint32_t o = (inc & 0xFFFFu) - nkeep, m = (inc >> 16) - nmark; // kept bytes / markers in front of this lane's share
            coop = (tot >> 16) == (uint32_t)(ng - 1); // (as many markers as the table says; the same for every lane)
            if (coop) {
                // The kept bytes of a dword, compacted by v_perm_b32 (first byte on top: the stage holds big-endian dwords), are ORed into the zeroed
                // stage at their byte position: kept bytes in front + 8 per marker in front (the 8 zero bytes behind every segment). A dword with a
                // marker inside has a part in front of it and a part behind it, 8 bytes further on: both parts are placed by every lane, the second
                // one empty as a rule -- no branch, and the selectors come out of lanes 0..15 of the wave (ds_bpermute_b32), which must be active.
#pragma unroll
                for (int i = 0; i < ND; i++) {
                    if ((uint32_t)(4 * i) < C) {
                        const uint32_t x = __builtin_amdgcn_alignbyte(win[i + 2], win[i + 1], lead);
                        const uint32_t nk = (uint32_t)(keep >> (4 * i)) & 15u, nm = (uint32_t)(mark >> (4 * i)) & 15u;
                        const uint32_t below = nm ? ((nm & (0u - nm)) - 1u) : 15u; // the bytes in front of the (first) marker
                        const uint32_t nA = nk & below, nB = nk & ~below;
                        const uint32_t cA = (uint32_t)__popc(nA);
                        const uint32_t compA = __builtin_amdgcn_perm(0u, x, (uint32_t)__builtin_amdgcn_ds_bpermute((int)(nA << 2), (int)sel_of_lane));
                        const uint32_t compB = __builtin_amdgcn_perm(0u, x, (uint32_t)__builtin_amdgcn_ds_bpermute((int)(nB << 2), (int)sel_of_lane));
                        const uint32_t pA = o + 8u * m, pB = pA + cA + 8u; // (a second marker in the same dword has no kept byte behind it in this dword)
                        const uint32_t shA = 8u * (pA & 3u), shB = 8u * (pB & 3u);
                        uint32_t* const dA = s_stage + (pA >> 2);
                        uint32_t* const dB = s_stage + (pB >> 2);
                        atomicOr(dA, compA >> shA);
                        atomicOr(dA + 1, __builtin_amdgcn_alignbit(compA, 0u, shA));
                        atomicOr(dB, compB >> shB);
                        atomicOr(dB + 1, __builtin_amdgcn_alignbit(compB, 0u, shB));
                        uint32_t mm = m;
                        for (uint32_t q = nm; q; q &= q - 1u) { // the segments that start behind this dword's markers: 8 bytes behind the previous one's last byte
                            mm++;
                            s_ubyte[j0 + (int)mm] = o + (uint32_t)__popc(nk & ((q & (0u - q)) - 1u)) + 8u * mm;
                        }
                        m = mm;
                        o += (uint32_t)__popc(nk);
                    }
                }
                if (tid == 255) s_ubyte[j1] = o + 8u * m + 8u; // behind the last segment (the stage is zero from there on)
                if (tid == 0) s_ubyte[j0] = 0;
            }
        }
        __syncthreads();
        if (tid == 0) GJ_STAT(coop ? 30 : 31, 1); // (groups copied cooperatively / segment by segment)
        if (coop) {
            if (tid >= j0 && tid < j1) s_ulen[tid] = s_ubyte[tid + 1] - s_ubyte[tid] - 8u;
        } else {
            // segment by segment, one wave each (irregular tables: host walk of a damaged stream, APP13 index, missing segments)
            for (int j = j0 + wave; j < j1; j += 4) {
                const uint32_t len = s_len[j], base = s_cap[j] - s_cap[j0];
                uint32_t ulen = 0;
                if (len) {
                    const uintptr_t a = reinterpret_cast<uintptr_t>(jpeg) + s_pos[j];
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
                    const uint32_t ndw = ((uint32_t)(a & 3) + len + 3u) >> 2;
                    uint32_t w0 = 0;
                    if ((uint32_t)lane < ndw && src + lane < end) w0 = src[lane];
                    ulen = gj_unstuff_segment(jpeg, end, s_pos[j], len, s_stage, base, lane, w0);
                }
                if (lane == 0) { s_ubyte[j] = base; s_ulen[j] = ulen; }
            }
        }
        __syncthreads();

        GJ_TRACE(2);
        // -- 2. sub-sequence table
        //       Sub-sequences of GJ_TOK_SUB bytes, unless the group then needs a little more than a whole number of passes of the 256 lanes
        //       (an 8K frame's luminance batches: ~530 sub-sequences of 16 bytes = two full passes and a third one of 18 lanes, which takes as
        //       long as a full one): somewhat longer sub-sequences that fit one pass less are cheaper (2 x 17 instead of 3 x 16).
        uint32_t sub_bytes = GJ_TOK_SUB;
        {
            uint32_t ulen = 0;
            if (tid >= j0 && tid < j1) {
                ulen = my_nblk ? s_ulen[tid] : 0u;
                s_sbit[tid] = s_ubyte[tid] * 8u + 32u; // (the stage starts at dword 1 of the reader's address space)
                s_ebit[tid] = (s_ubyte[tid] + ulen) * 8u + 32u;
            }
            uint32_t tot;
            uint32_t a = gj_wg256_incl_scan(((ulen + GJ_TOK_SUB - 1) / GJ_TOK_SUB) | (ulen << 16), s_tmp, &tot) & 0xFFFFu; // (sums < 2^16 both)
            const uint32_t passes = ((tot & 0xFFFFu) + 255u) >> 8, bytes = tot >> 16;
            if (passes >= 2) {
#pragma unroll
                for (uint32_t s = GJ_TOK_SUB + 1; s <= GJ_TOK_SUB_MAX; s++) // (a segment of n bytes has at most n / s + 1 sub-sequences)
                    if (sub_bytes == GJ_TOK_SUB && bytes / s + (uint32_t)ng <= 256u * (passes - 1u)) sub_bytes = s;
            }
            sub_bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)sub_bytes);
            if (sub_bytes != GJ_TOK_SUB) a = gj_wg256_incl_scan((ulen + sub_bytes - 1u) / sub_bytes, s_tmp, &tot);
            // (a segment of n bytes has n / 16 + 1 sub-sequences at most: MAX_SUBS cannot be exceeded; the clamp keeps a fault in the
            // arithmetic above from becoming a write behind s_rec)
            if (tid >= j0 && tid < j1) s_sub0[tid + 1] = min(a, (uint32_t)MAX_SUBS);
            if (tid == 0) s_sub0[j0] = 0;
        }
        __syncthreads();
        const int nsub = (int)s_sub0[j1];
        const uint32_t sub_bits = sub_bytes * 8u;
        if (tid == 0) { GJ_STAT(32 + min(nsub / 32, 31), 1); GJ_STAT(28, sub_bytes != GJ_TOK_SUB ? 1 : 0); }
        for (int k = tid; k < nsub; k += 256) {
            int lo = j0, hi = j1; // segment j with s_sub0[j] <= k < s_sub0[j + 1]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_sub0[mid] <= (uint32_t)k) lo = mid; else hi = mid;
            }
            s_rec[k].x = (uint32_t)lo << 22;
        }
        __syncthreads();

        GJ_TRACE(3);
        // -- 3. first pass: the first sub-sequence of a segment starts in the true state; the lane of any other one runs in over the last
        //       GJ_TOK_SYNC bits in front of it and takes the state it arrives in (right in most cases: the rounds below find the others)
        for (int k0 = 0; k0 < nsub; k0 += 256) {
            const int k = k0 + tid;
            uint32_t c0 = 0;
            if (k < nsub) {
                const int j = (int)(s_rec[k].x >> 22);
                const uint32_t i = (uint32_t)k - s_sub0[j];
                const uint32_t sb = s_sbit[j] + i * sub_bits, eb = min(sb + sub_bits, s_ebit[j]);
                const uint32_t e0 = i == 0 ? 0u : gj_tok_run_in(sm, sb - (uint32_t)GJ_TOK_SYNC, sb);
                const uint32_t x0 = gj_tok_decode<0>(sm, sb, eb, e0, c0, nullptr, nullptr, 0, 0, 0, nullptr, nullptr);
                s_rec[k] = make_uint2(e0 | (x0 << 11) | ((uint32_t)j << 22), c0);
            }
            GJ_STAT_WAVE(sm, 16, lane, GJ_SYMS(c0));
        }
        GJ_TRACE(4);
        // -- rounds: sub-sequences whose predecessor leaves in another state than they were entered with are decoded again, densely packed
        //    onto the lanes, until there is none (s_rec[k] is written with one 64-bit store: a record always describes one decoding)
        for (int round = 0;; round++) {
            __syncthreads();
            uint32_t* const cnt = &sm.nwork[round & 1];
            for (int k0 = 0; k0 < nsub; k0 += 256) {
                const int k = k0 + tid;
                bool cand = false;
                if (k < nsub) {
                    const uint32_t x = s_rec[k].x;
                    cand = (uint32_t)k != s_sub0[x >> 22] && ((s_rec[k - 1].x >> 11) & 0x7FFu) != (x & 0x7FFu);
                }
                const unsigned long long m = __ballot(cand);
                uint32_t base = 0;
                if (lane == 0 && m) base = atomicAdd(cnt, (uint32_t)__popcll(m));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (cand) s_work[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)k;
            }
            __syncthreads();
            const int nwork = (int)*cnt;
            if (tid == 0) { GJ_STAT(0, round == 0 ? nsub : 0); GJ_STAT(1 + min(round, 13), nwork); GJ_STAT(15, nwork == 0 ? round : 0); }
            if (tid == 0) sm.nwork[(round + 1) & 1] = 0;
            if (nwork == 0) break;
            for (int w = tid; w < nwork; w += 256) {
                const int k = s_work[w];
                const int j = (int)(s_rec[k].x >> 22);
                const uint32_t i = (uint32_t)k - s_sub0[j];
                const uint32_t sb = s_sbit[j] + i * sub_bits, eb = min(sb + sub_bits, s_ebit[j]);
                const uint32_t e = (s_rec[k - 1].x >> 11) & 0x7FFu;
                uint32_t c;
                const uint32_t x = gj_tok_decode<0>(sm, sb, eb, e, c, nullptr, nullptr, 0, 0, 0, nullptr, nullptr);
                s_rec[k] = make_uint2(e | (x << 11) | ((uint32_t)j << 22), c);
            }
        }

        GJ_TRACE(5);
        // -- 4. block and token positions: inclusive prefix sums of both counts (two 16-bit sums in one scan)
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
        for (uint32_t b = s_bb[j0] + (uint32_t)tid; b < s_bb[j1]; b += 256) s_blkinfo[b] = 0xFFFF0000u; // "block not seen", DC difference 0
        __syncthreads();

        GJ_TRACE(6);
        // -- 5. decode once more, now with values. The group's tokens form one dense run that starts at 4 x the byte offset of the group's
        //       first segment (a token takes at least 3 bits of the stream, so the runs of different groups cannot overlap, and no
        //       allocator or reset is needed between frames). Wave w takes the w-th quarter of the sub-sequences, 64 (or as many as
        //       fit its stage) at a time, and flushes their tokens -- consecutive in the run -- with 16-byte pieces.
        const uint32_t T = nsub > 0 ? s_rec[nsub - 1].y >> 16 : 0u;
        uint32_t gbase = 4u * sm.gpos; // (of the group's first segment that has blocks, noted when the group was formed)
        if (sm.gpos > tok_cap / 4u || T > tok_cap - gbase) gbase = 0xFFFFFFFFu; // (cannot happen with the capacity the host allocates)
        if (gbase != 0xFFFFFFFFu) {
            const int ka = (int)(((uint32_t)nsub * (uint32_t)wave) >> 2), kb = (int)(((uint32_t)nsub * (uint32_t)(wave + 1)) >> 2);
            for (int k0 = ka; k0 < kb;) {
                const uint32_t P0 = k0 > 0 ? s_rec[k0 - 1].y >> 16 : 0u; // tokens of the group in front of this chunk
                const uint32_t a0 = (gbase + P0) & 7u;                    // the stage mirrors the alignment of the run: piece p <-> tokens 8p .. 8p + 7
                const int kk = k0 + lane;
                const bool fits = kk < kb && a0 + ((s_rec[min(kk, nsub - 1)].y >> 16) - P0) <= stage_cap;
                const unsigned long long fm = __ballot(fits);
                const int n = fm == ~0ull ? 64 : __builtin_ctzll(~fm); // leading lanes whose tokens fit (at least one: a sub-sequence has < 64 tokens)
                gj_wave_sync(); // (the previous flush has read the stage)
                uint32_t c = 0;
                if (lane < n) {
                    const int j = (int)(s_rec[kk].x >> 22);
                    const uint32_t kf = s_sub0[j];
                    const uint32_t i = (uint32_t)kk - kf;
                    const uint32_t before_k = kk > 0 ? s_rec[kk - 1].y : 0u, before_f = kf > 0 ? s_rec[kf - 1].y : 0u;
                    const uint32_t sb = s_sbit[j] + i * sub_bits, eb = min(sb + sub_bits, s_ebit[j]);
                    gj_tok_decode<1>(sm, sb, eb, s_rec[kk].x & 0x7FFu, c, stage + a0 + ((before_k >> 16) - P0), s_blkinfo + s_bb[j], before_k >> 16,
                                     (before_k & 0xFFFFu) - (before_f & 0xFFFFu), s_bb[j + 1] - s_bb[j] - 1u, nullptr, &sm.big);
                }
                gj_wave_sync();
                GJ_STAT_WAVE(sm, 20, lane, GJ_SYMS(c));
                // flush
                const uint32_t cnt = (s_rec[k0 + n - 1].y >> 16) - P0;
                uint16_t* const dst = d_tok + (size_t)(gbase + P0 - a0); // 16-byte aligned; stage[s] <-> dst[s]
                for (uint32_t p = (uint32_t)lane; p * 8u < a0 + cnt; p += 64) {
                    const uint32_t lo = p * 8u, hi = lo + 8u;
                    if (lo >= a0 && hi <= a0 + cnt) {
                        *reinterpret_cast<uint4*>(dst + lo) = *reinterpret_cast<const uint4*>(stage + lo);
                    } else { // the ends of the chunk: token by token (the neighbours belong to another wave)
                        for (uint32_t s = max(lo, a0); s < min(hi, a0 + cnt); s++) dst[s] = stage[s];
                    }
                }
                k0 += n;
                if (lane == 0) { GJ_STAT(24, 1); GJ_STAT(25, n); GJ_STAT(26, k0 >= kb ? (kb - ka + 63) / 64 : 0); GJ_STAT(27, k0 >= kb ? 1 : 0); }
            }
        }
        __syncthreads();

        GJ_TRACE(7);
        // -- 5b. a coefficient did not fit a token: the group's blocks go through the coefficient planes instead (records say so)
        const bool planes = sm.big != 0 || gbase == 0xFFFFFFFFu;
        if (planes) {
            for (int j = j0 + wave; j < j1; j += 4) {
                const uint32_t chunks = (s_bb[j + 1] - s_bb[j] - 1u) * 8u;
                for (uint32_t c = (uint32_t)lane; c < chunks; c += 64)
                    reinterpret_cast<uint4*>(coefs + (uint64_t)(s_first[j] + (c >> 3)) * 64)[c & 7u] = make_uint4(0, 0, 0, 0);
            }
            __syncthreads(); // (orders the zeros before the coefficient stores of the other lanes)
            for (int k = tid; k < nsub; k += 256) {
                const int j = (int)(s_rec[k].x >> 22);
                const uint32_t kf = s_sub0[j];
                const uint32_t i = (uint32_t)k - kf;
                const uint32_t before_k = k > 0 ? s_rec[k - 1].y : 0u, before_f = kf > 0 ? s_rec[kf - 1].y : 0u;
                const uint32_t sb = s_sbit[j] + i * sub_bits, eb = min(sb + sub_bits, s_ebit[j]);
                uint32_t c;
                gj_tok_decode<2>(sm, sb, eb, s_rec[k].x & 0x7FFu, c, nullptr, s_blkinfo + s_bb[j], 0, (before_k & 0xFFFFu) - (before_f & 0xFFFFu),
                                 s_bb[j + 1] - s_bb[j] - 1u, coefs + (uint64_t)s_first[j] * 64, nullptr);
            }
            __syncthreads();
        }

        // -- 6. DC prediction: one wave per segment, prefix sum over its blocks; one record per block in coding order
        for (int j = j0 + wave; j < j1; j += 4) {
            const uint32_t bb = s_bb[j];
            const int nblk = (int)(s_bb[j + 1] - bb) - 1;
            const uint32_t k_end = s_sub0[j + 1];
            const uint32_t seg_end = k_end > s_sub0[j] ? s_rec[k_end - 1].y >> 16 : 0u; // tokens of the group up to the end of this segment
            int carry = 0;
            for (int kb0 = 0; kb0 < nblk; kb0 += 64) {
                const int kb = kb0 + lane;
                const uint32_t info = kb < nblk ? s_blkinfo[bb + kb] : 0xFFFF0000u;
                const uint32_t inc = gj_wave_incl_scan((uint32_t)(int)(int16_t)(info & 0xFFFFu));
                const int dc = carry + (int)inc;
                carry += (int)(uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                if (kb < nblk) {
                    const uint32_t r = s_first[j] + (uint32_t)kb;
                    if (planes) {
                        coefs[(uint64_t)r * 64] = (int16_t)dc;
                        d_rec[r] = make_uint2(0u, 0xFFFF0000u); // "the block is in the coefficient planes"
                    } else {
                        const uint32_t t0 = info >> 16, nx = s_blkinfo[bb + kb + 1] >> 16;
                        const uint32_t t1 = nx != 0xFFFFu ? nx : seg_end;
                        const bool seen = t0 != 0xFFFFu;
                        const uint32_t cnt = seen && t1 >= t0 ? min(t1 - t0, 63u) : 0u;
                        d_rec[r] = make_uint2(seen ? gbase + t0 : 0u, (cnt << 16) | ((uint32_t)dc & 0xFFFFu));
                    }
                }
            }
        }
        __syncthreads();
        GJ_TRACE(8);
        j0 = j1;
    }
}

void gj_launch_huffman_tok(const gj_dec_job* job, hipStream_t st)
{
    const gj_geom& g = job->g;
    const unsigned frames = job->batch.count > 1 ? job->batch.count : 1u;
    // (a batch of frames fills the device with frames: the fullest batches, no preference for one generation of workgroups)
    const GjBatchPlan plan = gj_plan_batches(job, GJ_TOK_CAP_U, GJ_TOK_MAX_BLOCKS, GJ_TOK_GMAX, frames > 1 ? 0u : GJ_TOK_RESIDENT);
    auto kernel = job->tune.dec_tok_nocoop ? k_huffman_decode_tok<false> : k_huffman_decode_tok<true>;
    GjFold F = {nullptr, nullptr, 0, 0, 0, nullptr, nullptr};
    const bool fold = gj_tok_folds_table(job);
    if (fold) {
        const gj_scan_deferred& sc = job->scan;
        F = GjFold{sc.recs, sc.lists, sc.wgs, sc.part_bytes, sc.begin, sc.h_summary, sc.h_maxlen_parts};
        *sc.maxlen_part_count = (uint32_t)plan.batch0[plan.n]; // (the host takes the maximum over the batches' words)
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)plan.batch0[plan.n], 1, frames), dim3(256), 0, st, g, job->d_jpeg, job->jpeg_size, job->d_seg_pos, job->d_seg_len, job->d_seg_index,
                       job->seg_count, fold ? nullptr : job->d_seg_count, plan, job->d_huff_tab2, job->d_coefs, (uint16_t*)job->d_tok, job->tok_cap, (uint2*)job->d_blkrec,
                       job->d_overflow, F);
}

// the table launch can be folded into this kernel for ONE frame of a non-interleaved stream whose batches are cut per scan (the plan the kernel's
// arithmetic assumes), when the host's array holds a word per batch -- and the geometry says what a regular stream looks like (restart intervals)
bool gj_tok_folds_table(const gj_dec_job* job)
{
    const gj_geom& g = job->g;
    if (!job->scan.valid || job->batch.count > 1 || g.fb.sizes != nullptr || g.interleaved || g.restart_interval <= 0 || job->seg_count != g.segment_count) return false;
    if (job->scan.wgs > 256u || job->scan.h_summary == nullptr || job->scan.h_maxlen_parts == nullptr || job->scan.maxlen_part_count == nullptr) return false;
    const GjBatchPlan plan = gj_plan_batches(job, GJ_TOK_CAP_U, GJ_TOK_MAX_BLOCKS, GJ_TOK_GMAX, GJ_TOK_RESIDENT);
    return plan.n == g.comp_count && (uint32_t)plan.batch0[plan.n] <= job->scan.maxlen_capacity;
}
