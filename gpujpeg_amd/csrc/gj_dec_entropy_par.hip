// gj_dec_entropy_par.hip -- MI355X (gfx950, wave64) JPEG decoder: sub-sequence parallel entropy decoding of batches of restart segments
// (part of the decoder's device code, see gj_dec_internal.h for the map of the files)
#include "gj_dec_internal.h"

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
#define GJ_PAR_CAP_U 8192
#define GJ_PAR_MAX_BLOCKS 1280
#define GJ_PAR_GMAX 64        // segments per batch
#ifndef GJ_PAR_SUB
#define GJ_PAR_SUB 16         // bytes per sub-sequence
#endif

// table entry (gj_hip.h, GJ_DEC2_*): bits [0,5) code length + magnitude bits (0 = second level), [5,9) magnitude bits,
// [9,16) zig-zag advance. State between two symbols: bits [0,5) overshoot into the next sub-sequence, [5,11) zig-zag
// index, [11,16) block inside the MCU.
// LONG (pieces of a segment that does not fit the LDS stage): block addresses are computed, DC differences go to the plane.
#define GJ_TABP(tab, byte_off) (reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(tab) + (byte_off)))
template <bool WRITE, bool INTERLEAVED, bool LONG = false>
__device__ __forceinline__ uint32_t gj_decode_sub(const uint32_t* __restrict__ U, const uint32_t start_bit, const uint32_t end_bit,
                                                  const uint32_t entry, const uint16_t* __restrict__ s_tab, const uint32_t* __restrict__ s_ptab,
                                                  const int P, const uint16_t* tdc, const uint16_t* tac, int& nblk_out,
                                                  int16_t* __restrict__ coefs, const uint32_t first, const uint32_t* __restrict__ s_blk,
                                                  int16_t* __restrict__ s_dc, int blk, const int nblocks, const uint8_t* __restrict__ s_zz,
                                                  const gj_geom* lg = nullptr, const GjSeg* lsg = nullptr)
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
    while (bitpos < end_bit) {
        if (n <= 32) {
            acc |= (uint64_t)nxt << (32 - n);
            n += 32;
            rd++;
            nxt = U[rd];
        }
        const uint32_t hi = (uint32_t)(acc >> 32);
        const uint16_t* t = z == 0 ? tdc : tac;
        const uint32_t fast = gj_bfe_u32<32 - GJ_DEC_FAST_BITS, GJ_DEC_FAST_BITS>(hi);
        uint32_t e = t[fast];
        if ((e & 31u) == 0) e = t[(e >> 5) + ((hi >> 16) & 63u)]; // codes longer than 10 bits
        const int tot = (int)(e & 31u);
        const int adv = (int)((e >> 9) & 63u);
        if (WRITE) {
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
                        coefs[off + (z == 0 ? 0 : s_zz[pos])] = (int16_t)v;
                    }
                } else if (z == 0) {
                    s_dc[blk + nb] = (int16_t)v;
                } else if (sz != 0 && pos < 64) {
                    const uint32_t b = INTERLEAVED ? s_blk[blk + nb] : first + (uint32_t)(blk + nb);
                    coefs[(uint64_t)b * 64 + s_zz[pos]] = (int16_t)v;
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
            if (INTERLEAVED) {
                p = p + 1 == P ? 0 : p + 1;
                const uint32_t pt = s_ptab[p];
                tdc = GJ_TABP(s_tab, pt & 0xFFFFu);
                tac = GJ_TABP(s_tab, pt >> 16);
            }
        }
    }
    nblk_out = nb;
    return (bitpos - end_bit) | ((uint32_t)z << 5) | ((uint32_t)p << 11);
}

template <bool INTERLEAVED, int SUB_BYTES>
__global__ __launch_bounds__(256) void k_huffman_decode_par(const gj_geom g, const uint8_t* __restrict__ jpeg, uint64_t jpeg_size,
                                                            const uint32_t* __restrict__ seg_pos, const uint32_t* __restrict__ seg_len,
                                                            const uint32_t* __restrict__ seg_index, const int seg_count_max,
                                                            const uint32_t* __restrict__ seg_count_ptr, const GjBatchPlan plan,
                                                            const uint16_t* __restrict__ tabs, int16_t* __restrict__ coefs,
                                                            const int zero_fill /* 1: the planes are not known to be zero */, const GjFold F)
{
    // every segment of a group ends with a partial sub-sequence: at most CAP_U / SUB + n sub-sequences for n segments (tighter bounds
    // hold only for short sub-sequences; nsub is clamped below all the same, so that a fault in this arithmetic cannot become a write
    // behind the per-sub-sequence arrays)
    constexpr int MAX_SUBS = GJ_PAR_CAP_U / SUB_BYTES + GJ_PAR_GMAX;
    if (g.fb.sizes != nullptr) { // frame blockIdx.z of a batch: its stream, its table, its summary word, its coefficient planes
        const size_t z = blockIdx.z;
        jpeg += z * g.fb.jpeg;
        jpeg_size = g.fb.sizes[z];
        seg_pos += z * g.fb.seg; seg_len += z * g.fb.seg; seg_index += z * g.fb.seg;
        if (seg_count_ptr) seg_count_ptr += z * (sizeof(gj_scan_summary) / 4);
        coefs += z * g.fb.coefs;
    }
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
    __shared__ uint32_t s_blk[INTERLEAVED ? GJ_PAR_MAX_BLOCKS : 1]; // interleaved: block index in the coefficient planes
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
    if (tid < 128) s_zz[tid] = tid < 64 ? GJ_ZZ[tid] : 63;
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
    // the batch's table entries: from the table, or (no table launch: GjFold, gj_dec_internal.h) from the marker scan's records -- scratch: the stage
    uint32_t ld_s = 0xFFFFFFFFu, ld_p = 0, ld_l = 0;
    if (F.recs == nullptr) {
        if (tid < nseg) { ld_s = seg_index[si0 + tid]; ld_p = seg_pos[si0 + tid]; ld_l = seg_len[si0 + tid]; }
    } else {
        static_assert(GJ_FOLD_SCRATCH_WORDS(GJ_PAR_GMAX) <= GJ_PAR_CAP_U / 4, "fold scratch inside the stage");
        if (!gj_fold_batch<GJ_PAR_GMAX, INTERLEAVED>(F, g, jpeg, jpeg_size, plan, pc, si0, nseg, s_U, s_tmp, ld_s, ld_p, ld_l)) return;
    }
    if (tid < GJ_PAR_GMAX) {
        uint32_t pos = 0, len = 0, nblk = 0, first = 0, tb = 0;
        if (tid < nseg) {
            const uint32_t s = ld_s;
            if (s < (uint32_t)g.segment_count) {
                const GjSeg sg = gj_segment(g, (int)s);
                nblk = (uint32_t)sg.nblocks;
                pos = ld_p;
                len = ld_l;
                if (INTERLEAVED) {
                    first = (uint32_t)sg.mcu_first; // first MCU
                } else {
                    const gj_comp_geom& kc = g.comp[sg.comp];
                    first = (uint32_t)(kc.data_offset / 64) + (uint32_t)sg.mcu_first; // first block in the coefficient plane
                    tb = (uint32_t)((kc.dc_table * 2 + 0) * GJ_DEC2_WORDS * 2) | ((uint32_t)((kc.ac_table * 2 + 1) * GJ_DEC2_WORDS * 2) << 16);
                }
                if (((len + 3u) & ~3u) + 8u > (uint32_t)GJ_PAR_CAP_U || nblk > (uint32_t)GJ_PAR_MAX_BLOCKS) { // too long for the LDS stage / the per-block arrays: in pieces at the end
                    // (the pieces are cut from the table: without one -- GjFold -- the host decodes this frame again the careful way, with the table launch)
                    if (F.recs != nullptr) F.hsum->rst_irregular = 1u;
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
    if (INTERLEAVED) { // where every block of the batch lives in the coefficient planes
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
    if (INTERLEAVED) __syncthreads(); // s_blk is complete
    if (zero_fill) {
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
            const uint32_t x = gj_decode_sub<false, INTERLEAVED>(U, i * SUB_BITS, min((i + 1) * SUB_BITS, seg_bits), e, s_tab, s_ptab, P,
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
                        const uint32_t w0_last = (uint32_t)__builtin_amdgcn_readlane((int)w0, 63); // (cross-lane reads stay outside lane-dependent branches)
                        if (lane == 0) pw1 = w0_last;
                        const uint32_t pb0 = __builtin_amdgcn_alignbit(w0, pw0, 24), pb1 = __builtin_amdgcn_alignbit(w1, pw1, 24); // the stream one byte earlier
                        const uint32_t hit0 = (w0 - 0x01010101u) & ~w0 & (~pb0 - 0x01010101u) & pb0 & 0x80808080u;
                        const uint32_t hit1 = (w1 - 0x01010101u) & ~w1 & (~pb1 - 0x01010101u) & pb1 & 0x80808080u;
                        if (__ballot((hit0 != 0u && (uint32_t)lane < ndw) || (hit1 != 0u && (uint32_t)lane + 64u < ndw)) == 0ull) {
                            uint32_t wn0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w0, 0x130, 0xF, 0xF, false); // wave_shl:1
                            const uint32_t wn1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w1, 0x130, 0xF, 0xF, false); // (lane 63: zero, nothing behind)
                            const uint32_t w1_first = (uint32_t)__builtin_amdgcn_readlane((int)w1, 0);
                            if (lane == 63) wn0 = w1_first;
                            uint32_t d0 = __builtin_bswap32(__builtin_amdgcn_alignbyte(wn0, w0, (uint32_t)lead));
                            uint32_t d1 = __builtin_bswap32(__builtin_amdgcn_alignbyte(wn1, w1, (uint32_t)lead));
                            const uint32_t full = len >> 2, rest = len & 3u, cut = 0xFFFFFFFFu << ((32u - 8u * rest) & 31u); // (the bytes behind the end are zero padding; used when rest != 0)
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
            if (tid >= j0 && tid < j1) s_sub0[tid + 1] = min(a, (uint32_t)MAX_SUBS);
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
        for (uint32_t b = s_bb[j0] + (uint32_t)tid; b < s_bb[j1]; b += 256) s_dc[b] = 0; // (blocks a damaged segment never reaches)
        __syncthreads();
        for (int k = tid; k < nsub; k += 256) {
            const int j = s_subseg[k];
            const uint32_t k_first = s_sub0[j];
            const uint32_t i = (uint32_t)k - k_first;
            const uint32_t sc_k = k > 0 ? s_rec[k - 1].y : 0u, sc_f = k_first > 0 ? s_rec[k_first - 1].y : 0u;
            const uint32_t before = sc_k - sc_f;
            const uint32_t endb = min((i + 1) * SUB_BITS, s_ulen[j] * 8u);
            const uint32_t tb = s_tabs[j];
            int nb;
            gj_decode_sub<true, INTERLEAVED>(s_U + ((s_ub[j] - ub0) >> 2), i * SUB_BITS, endb, s_rec[k].x & 0xFFFFu, s_tab, s_ptab, P, GJ_TABP(s_tab, tb & 0xFFFFu),
                                             GJ_TABP(s_tab, tb >> 16), nb, coefs, s_first[j], INTERLEAVED ? s_blk + s_bb[j] : nullptr, s_dc + s_bb[j], (int)before, (int)s_nblk[j], s_zz);
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
                    const uint32_t b = INTERLEAVED ? s_blk[bb + kb] : s_first[j] + (uint32_t)kb;
                    coefs[(uint64_t)b * 64] = (int16_t)dc;
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
    const int nlong = F.recs != nullptr ? 0 : s_nlong; // (no table, no pieces: the frame is decoded again, see above)
    for (int li = 0; li < nlong; li++) {
        const int jl = (int)s_long[li];
        const GjSeg sg = gj_segment(g, (int)seg_index[si0 + jl]);
        const uint8_t* base = jpeg + seg_pos[si0 + jl];
        const uint32_t len = seg_len[si0 + jl];
        const uint32_t first = s_first[jl];
        if (zero_fill) {
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
                const uint32_t before = k > 0 ? s_rec[k - 1].y : 0u;
                const uint32_t endb = min((uint32_t)(k + 1) * SUB_BITS, ulen * 8u);
                int nb;
                gj_decode_sub<true, INTERLEAVED, true>(s_U, (uint32_t)k * SUB_BITS, endb, s_rec[k].x & 0xFFFFu, s_tab, s_ptab, P, GJ_TABP(s_tab, tb & 0xFFFFu),
                                                       GJ_TABP(s_tab, tb >> 16), nb, coefs, first, nullptr, nullptr, (int)(blocks_done + before), sg.nblocks, s_zz,
                                                       &g, &sg);
            }
            __syncthreads(); // (workgroup-scope fence: the differences are visible to the lanes that sum them up)
            const uint32_t piece_blocks = nsub > 0 ? s_rec[nsub - 1].y : 0u;
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


GjBatchPlan gj_plan_batches(const gj_dec_job* job, const unsigned cap_u, const unsigned max_blocks, const unsigned gmax, const unsigned resident)
{
    const gj_geom& g = job->g;
    // batches: as many segments as fill the LDS stage on average, at most max_blocks blocks, per scan where the bytes per scan are known
    const int eg = job->tune.dec_batch; // tuning aid: segments per batch
    auto batch_size = [&](uint64_t bytes, int segs, unsigned fill /* 32nds of the stage */) {
        const unsigned avg = (unsigned)(bytes / (uint64_t)max(1, segs)) + 12u;
        int G = eg ? eg : (int)((cap_u * fill / 32u) / avg); // (a batch that outgrows the stage is decoded in two groups)
        const long all_segs = (long)job->seg_count * (job->batch.count > 1 ? (long)job->batch.count : 1L); // (a batch of frames: the segments of all of them)
        if (!eg) G = min(G, (int)max(1L, all_segs / 768)); // small frames: rather more, shorter batches than idle CUs (measured: HD, 4K)
        return max(1, min(G, (int)min(gmax, max_blocks / (unsigned)max(1, g.seg_blocks))));
    };
    GjBatchPlan plan = {};
    bool per_scan = !g.interleaved && g.comp_count > 1 && job->seg_count == g.segment_count;
    for (int c = 0; per_scan && c < g.comp_count; c++) per_scan = job->scan_bytes[c] != 0 && g.comp[c].segment_count > 0;
    // 23/32 of the stage on average is the measured optimum; when that gives a little more than one generation of resident
    // workgroups, fuller batches (up to 27/32) that fit into one are better than a second generation of a few
    // (a batch of frames is dozens of generations of workgroups: measured best at 26/32 -- 256 x 4K 21 030 frames/s against 20 860 at 23 and 20 700 at
    // 29, 256 x HD 70 020 against 69 230 and 68 100; profiles/r4_15_frame_batches.txt)
    const unsigned fill0 = job->tune.dec_fill >= 8 && job->tune.dec_fill <= 31 ? (unsigned)job->tune.dec_fill : (job->batch.count > 1 ? 26u : 23u);
    for (unsigned fill = fill0; fill <= (job->tune.dec_fill || job->batch.count > 1 ? fill0 : 27u); fill += 2) {
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
        if (!resident || eg || nb <= (int)resident || nb > (int)resident * 5 / 4) break;
    }
    // One generation of workgroups ends with its slowest batch, and batches of equal BYTES are not equally slow: a workgroup's time grows with
    // its bytes and with its blocks (8K, round-4 phase trace: 27 luminance segments = 7.8 KB and 972 blocks take 59 us, 64 chrominance segments =
    // 7.0 KB and 2 304 blocks 71 us; ~5.5 us per KB + ~12 us per 1000 blocks). When the frame fits one generation, cut the scans so that the
    // estimated times are equal: segments per batch ~ 1 / (cost of a segment of that scan), as small as the resident workgroups allow. What it
    // buys is a full generation for sparse content, where batches of equal bytes are few and long (8K camera frame: 695 batches, 68.9 us; cut
    // this way 1014 batches, 57.5 us). A plan that already fills nine tenths of the generation is left alone: for the natural 8K frame the
    // cut by time gave 74.9 against 74.1 us (a CU's four workgroups share its throughput: moving work between them does not change when the
    // last one ends). OFF by default (GJ_DEC_BALANCE=1): it is a latency optimisation for a lone decode -- with four pipelines sharing the GPU the same
    // camera frame runs at 230 instead of 241 Gpix/s, because more, smaller batches are more work in total and nothing is idle to begin with.
    if (resident && per_scan && !eg && job->tune.dec_balance && plan.batch0[plan.n] <= (int)resident * 9 / 10) {
        double w[GJ_MAX_COMP], work = 0.0, worst_now = 0.0;
        for (int c = 0; c < g.comp_count; c++) {
            const double avg = (double)job->scan_bytes[c] / (double)max(1, plan.count[c]) + 12.0;
            w[c] = avg * 5.5e-3 + (double)g.seg_blocks * 12.1e-3; // us per segment
            work += w[c] * plan.count[c];
            worst_now = max(worst_now, w[c] * plan.g[c]);
        }
        // (a batch also costs ~15 us whatever it holds -- table, tables, stage, barriers: below ~25 us per batch more, smaller batches lose, which is
        // what the 768-batch rule above encodes for HD and 4K: 4K natural 42.8 against 41.5 us, 4K camera 48.7 against 46.5 us when cut this way)
        for (int step = 0; step < 40 && work / (double)resident >= 25.0; step++) {
            const double T = work / (double)resident * (1.0 + 0.01 * step); // time of every batch
            GjBatchPlan q = plan;
            bool ok = true;
            double worst = 0.0;
            for (int c = 0; c < g.comp_count && ok; c++) {
                const unsigned avg = (unsigned)(job->scan_bytes[c] / (uint64_t)max(1, plan.count[c])) + 12u;
                const int cap_g = (int)min(min(gmax, max_blocks / (unsigned)max(1, g.seg_blocks)), (cap_u * 27u / 32u) / avg);
                const int G = min((int)(T / w[c]), cap_g);
                ok = G >= 1;
                if (!ok) break;
                q.g[c] = G;
                q.batch0[c + 1] = q.batch0[c] + (q.count[c] + G - 1) / G;
                worst = max(worst, w[c] * G);
            }
            if (ok && q.batch0[q.n] <= (int)resident) {
                if (worst < worst_now * 0.97) plan = q;
                break;
            }
        }
    }
    return plan;
}

void gj_launch_huffman_par(const gj_dec_job* job, hipStream_t st)
{
    const gj_geom& g = job->g;
    const GjBatchPlan plan = gj_plan_batches(job, GJ_PAR_CAP_U, GJ_PAR_MAX_BLOCKS, GJ_PAR_GMAX, 0);
    const int es = job->tune.dec_sub; // tuning aid: bytes per sub-sequence
    const int sub = es ? es : (g.interleaved ? 32 : GJ_PAR_SUB); // interleaved scans synchronise later (the block inside the MCU
                                                                       // has to fall into step too): measured best with 32 B
    const unsigned batches = (unsigned)plan.batch0[plan.n];
    auto kernel = g.interleaved ? (sub == 256 ? k_huffman_decode_par<true, 256> : sub == 128 ? k_huffman_decode_par<true, 128>
                                   : sub == 64 ? k_huffman_decode_par<true, 64> : sub == 32 ? k_huffman_decode_par<true, 32>
                                   : sub == 8 ? k_huffman_decode_par<true, 8> : k_huffman_decode_par<true, 16>)
                                : (sub == 256 ? k_huffman_decode_par<false, 256> : sub == 128 ? k_huffman_decode_par<false, 128>
                                   : sub == 64 ? k_huffman_decode_par<false, 64> : sub == 32 ? k_huffman_decode_par<false, 32>
                                   : sub == 8 ? k_huffman_decode_par<false, 8> : k_huffman_decode_par<false, 16>);
    GjFold F = {nullptr, nullptr, 0, 0, 0, nullptr, nullptr};
    const bool fold = gj_par_folds_table(job);
    if (fold) {
        const gj_scan_deferred& sc = job->scan;
        F = GjFold{sc.recs, sc.lists, sc.wgs, sc.part_bytes, sc.begin, sc.h_summary, sc.h_maxlen_parts};
        *sc.maxlen_part_count = batches; // (the host takes the maximum over the batches' words)
    }
    hipLaunchKernelGGL(kernel, dim3(batches, 1, job->batch.count > 1 ? job->batch.count : 1u), dim3(256), 0, st, g, job->d_jpeg, job->jpeg_size, job->d_seg_pos, job->d_seg_len,
                       job->d_seg_index, job->seg_count, fold ? nullptr : job->d_seg_count, plan, job->d_huff_tab2, job->d_coefs, job->clear_coefs ? 0 : 1, F);
}

// the table launch folded into this kernel (GjFold, gj_dec_internal.h): one frame whose batches are cut per scan (one scan per component, or one interleaved scan)
bool gj_par_folds_table(const gj_dec_job* job)
{
    const gj_geom& g = job->g;
    if (!job->scan.valid || job->batch.count > 1 || g.fb.sizes != nullptr || g.restart_interval <= 0 || job->seg_count != g.segment_count) return false;
    if (job->scan.wgs > 256u || job->scan.h_summary == nullptr || job->scan.h_maxlen_parts == nullptr || job->scan.maxlen_part_count == nullptr) return false;
    if (job->tune.dec_sub) return false; // (the tuning aid's sub-sequence sizes change the stage's use)
    //  a segment beyond the LDS stage / the per-block arrays is decoded in pieces, and the pieces are cut from the TABLE: the folded
    // kernel can only raise rst_irregular for it, which costs a second decode of every frame of such a sequence. The longest segment of the
    // previous frame with this header (or of this one, after a host walk) says whether that is to be expected; 0 = not known
    if (job->max_seg_len == 0 || ((job->max_seg_len + 3u) & ~3u) + 8u > (uint32_t)GJ_PAR_CAP_U || g.seg_blocks > GJ_PAR_MAX_BLOCKS) return false;
    const GjBatchPlan plan = gj_plan_batches(job, GJ_PAR_CAP_U, GJ_PAR_MAX_BLOCKS, GJ_PAR_GMAX, 0);
    // (the plan the prologue's arithmetic assumes: a range per scan -- one per component, or the one interleaved scan)
    return plan.n == (g.interleaved ? 1 : g.comp_count) && (uint32_t)plan.batch0[plan.n] <= job->scan.maxlen_capacity;
}
