// gj_dec_markers.hip -- MI355X (gfx950, wave64) JPEG decoder: device-side segment discovery (marker scan)
// (part of the decoder's device code, see gj_dec_internal.h for the map of the files)
#include "gj_dec_internal.h"
#include <stddef.h>

// ================================================================================================
// Device-side segment discovery. Inside entropy-coded data 0xFF is followed by 0x00 (stuffing),
// by 0xD0..0xD7 (restart marker = segment boundary) or by the marker that ends the scan. Two short launches turn the bytes
// [begin, size) into the (offset, length, geometric index) table the entropy decoders consume, without the host touching the
// stream (the reference walks it with memchr and copies every segment, src/gpujpeg_reader.c:1039-1155), and the stream is read ONCE:
//
//   k_marker_scan   at most 256 workgroups, each with a contiguous part of the stream, read with coalesced 16-byte loads that are all in
//                   flight together (a lane: 16 bytes of every 4 KB piece of the part). The lane's restart markers are bits in registers;
//                   their order in the part is (piece, lane): a wave prefix sum per piece and ONE workgroup prefix sum over the (piece,
//                   wave) totals give every marker its place in the workgroup's list, which goes to memory. The few other markers (SOS
//                   of the later scans, EOI) go into the workgroup's 64-byte record with the number of restart markers behind each,
//                   next to the number of restart markers and the position of the last one.
//   k_marker_table  the same workgroups: each reads the records of the workgroups IN FRONT of it and its own list (one trip, plain loads)
//                   -- that is all a table entry needs: restart marker number k of a scan ends segment k, the entry's index is the
//                   marker's rank among all markers in front of it plus its scan's number, the segment begins behind the marker in front;
//                   the segment at the end of a scan is written by the owner of the marker that ends the scan. The last workgroup has
//                   seen every record: it writes the summary the host validates (gj_scan_summary) straight to pinned host memory.
//
// Round 3: two launches that both read the stream, every workgroup of the second one reading ~1000 chunk counts (10 + 18 us for the
// 7.4 MB of an 8K frame, 32 + 44 us for config 4's 43 MB). One launch with the records passed between running workgroups was measured
// too (profiles/r4_05_*): a device-scope store-to-load hand-over costs ~6 us on this part, more than a launch.
// ================================================================================================
#ifdef GJ_TRACE_PHASES
static __device__ unsigned long long* gj_trace_buf_m;
extern "C" GJ_HIP_API int gj_hip_trace_set_markers(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(gj_trace_buf_m), &p, sizeof p) == hipSuccess ? 0 : -1; }
#define GJ_TRACE_M(slot) do { if (threadIdx.x == 0 && gj_trace_buf_m) gj_trace_buf_m[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GJ_TRACE_M(slot) ((void)0)
#endif
// (GJ_SCAN_LIST, GJ_SCAN_REC_WORDS and the records' other-marker layout: gj_dec_internal.h -- the token decoder reads them too)
#define GJ_SCAN_WGS 256      // workgroups (= records every workgroup reads in one pass) up to 16 MB of stream
#define GJ_SCAN_WGS_MAX 1024 // ... of one round of 16 pieces each up to 64 MB (k_marker_table reads their records in passes of 256); rounds beyond
// (GJ_SCAN_REC_WORDS = 16, gj_dec_internal.h) 32-bit words of a record:
//   [0] restart markers   [1] position of the last one   [2] other markers (0 .. 2)   [3] 1 = more restart markers than the list holds / more
//   than two other markers   [4 + 6 q ..] other marker q: position, code, restart markers of the workgroup behind it, the 8 bytes behind the
//   code (what the host validates of an SOS header: length, component, tables, spectral selection), one spare word

// the frames of a batch (blockIdx.z = frame): sizes == nullptr for a single stream
struct GjScanBatch {
    const uint32_t* sizes;         // [frames] bytes of every stream
    uint64_t jpeg;                 // bytes between two streams
    uint32_t seg, scratch, maxlen; // words between the frames' tables, scratch areas (records, then lists) and longest-segment words
};

// ITERS pieces of 4 KB per round (16 bytes per lane and piece), `rounds` rounds per workgroup
template <int ITERS>
__global__ __launch_bounds__(256) void k_marker_scan(const uint8_t* __restrict__ jpeg, const uint64_t begin, uint64_t size, const uint32_t rounds,
                                                     uint32_t* __restrict__ recs, uint32_t* __restrict__ lists, gj_scan_summary* __restrict__ hsum /* host memory */,
                                                     const uint8_t* __restrict__ hdr_ref, const uint32_t hdr_n, const GjScanBatch B)
{
    if (B.sizes != nullptr) {
        const size_t z = blockIdx.z;
        jpeg += z * B.jpeg;
        size = B.sizes[z];
        recs += z * B.scratch; lists += z * B.scratch;
        hsum += z;
    }
    __shared__ uint32_t s_tmp[4];
    __shared__ uint32_t s_blk[ITERS * 4 < 4 ? 4 : ITERS * 4]; // restart markers of (piece, wave), then their exclusive prefix sums
    __shared__ uint32_t s_own_other, s_own_q[2], s_own_after[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t me = blockIdx.x;
    // the parts are cut on 16-byte addresses: the first lane's begins in front of `begin`
    const uintptr_t a0 = (reinterpret_cast<uintptr_t>(jpeg) + begin) & ~(uintptr_t)15;
    const uint64_t part0 = (uint64_t)(a0 - reinterpret_cast<uintptr_t>(jpeg)) + (uint64_t)me * 4096u * ITERS * rounds;
    uint32_t* const list = lists + (size_t)me * GJ_SCAN_LIST;
    uint32_t* const rec = recs + (size_t)me * GJ_SCAN_REC_WORDS;
    GJ_TRACE_M(0);
    if (tid == 0) s_own_other = 0;
    if (tid < 2) { s_own_after[tid] = 0; s_own_q[tid] = 0xFFFFFFFFu; }
    uint32_t diff = 0;
    if (me == 0 && hdr_ref != nullptr) // does the stream start with the header the host assumed? (speculative launch)
        for (uint32_t i = tid; i < hdr_n; i += 256) diff |= jpeg[i] != hdr_ref[i];
    uint32_t total = 0, last = 0; // restart markers of the rounds so far, position of the last one
    for (uint32_t rd = 0; rd < rounds; rd++) {
        const uint64_t round0 = part0 + (uint64_t)rd * 4096u * ITERS;
        // ---- the round's bytes: a lane's 16 of every piece, and the four bytes behind them (the next lane's first ones)
        uint4 v[ITERS];
        uint32_t nx[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const uint64_t b = round0 + 4096u * it + 16u * tid;
            v[it] = make_uint4(0, 0, 0, 0);
            if (b < size && b + 16u > begin) v[it] = *reinterpret_cast<const uint4*>(jpeg + b); // (whole 16-byte pieces: the buffer is readable to a multiple of 16)
            nx[it] = 0;
            if (lane == 63 && b + 16u < size) nx[it] = *reinterpret_cast<const uint32_t*>(jpeg + b + 16u);
        }
        __syncthreads(); // (s_blk, s_own_q of the round before are no longer read)
        // ---- markers: bit b of rst[it] = byte b of the lane's 16 of piece it starts a restart marker; the restart markers' codes, 3 bits each.
        // Byte classes for four bytes at a time, without a branch (a wave nearly always holds a 0xFF somewhere, so a "rare" path is taken
        // for every dword): bit 7 of a byte of the masks below says "this byte is 0xFF / 0x00 / 0xD0..0xD7"; a marker is a 0xFF whose NEXT
        // byte is neither 0x00 nor 0xFF (the masks of the next bytes: the same masks moved down by a byte, v_alignbit_b32).
        uint32_t rst[ITERS], num[ITERS], winc[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const uint64_t b0 = round0 + 4096u * it + 16u * tid;
            const uint32_t next = (uint32_t)__builtin_amdgcn_update_dpp((int)nx[it], (int)v[it].x, 0x130, 0xF, 0xF, false); // wave_shl:1: the next lane's first dword
            uint32_t w[5] = {v[it].x, v[it].y, v[it].z, v[it].w, lane == 63 ? nx[it] : next};
            if (b0 < begin || b0 + 20u > size) { // the stream's first and last bytes: what lies outside is no 0xFF and follows none
#pragma unroll
                for (int q = 0; q < 5; q++) {
                    uint32_t keep = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint64_t p = b0 + (uint64_t)(4 * q + k);
                        if (p >= begin && p < size) keep |= 0xFFu << (8 * k);
                    }
                    w[q] &= keep;
                }
            }
            uint32_t ff[5], zero[5], rc[5];
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const uint32_t x = w[q], y = (x ^ 0xD0D0D0D0u) & 0xF8F8F8F8u;
                ff[q] = ((x & 0x7F7F7F7Fu) + 0x01010101u) & x & 0x80808080u;
                zero[q] = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
                rc[q] = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;
            }
            uint32_t r = 0, o = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t nplain = __builtin_amdgcn_alignbit(ff[q + 1] | zero[q + 1], ff[q] | zero[q], 8); // the next byte is 0x00 or 0xFF
                const uint32_t nrst = __builtin_amdgcn_alignbit(rc[q + 1], rc[q], 8);                           // the next byte is 0xD0..0xD7
                const uint32_t mk = ff[q] & ~nplain, mr = mk & nrst, mo = mk & ~nrst;
                // bits 7, 15, 23, 31 -> 0 .. 3
                r |= (((mr >> 7) | (mr >> 14) | (mr >> 21) | (mr >> 28)) & 0xFu) << (4 * q);
                o |= (((mo >> 7) | (mo >> 14) | (mo >> 21) | (mo >> 28)) & 0xFu) << (4 * q);
            }
            uint32_t nm = 0, nn = 0;
            for (uint32_t m = r; m; m &= m - 1, nn++) { // (a restart marker per ~170 bytes)
                const uint32_t bit = (uint32_t)__builtin_ctz(m) + 1u, q = bit >> 2;
                const uint32_t x = q == 0 ? w[0] : q == 1 ? w[1] : q == 2 ? w[2] : q == 3 ? w[3] : w[4];
                nm |= ((x >> (8u * (bit & 3u))) & 7u) << (3u * nn);
            }
            rst[it] = r;
            num[it] = nm;
            winc[it] = gj_wave_incl_scan((uint32_t)__builtin_popcount(r));
            if (lane == 63) s_blk[it * 4 + wave] = winc[it];
            while (o) { // scan boundary material (rare)
                const uint32_t bit = (uint32_t)__builtin_ctz(o);
                o &= o - 1;
                const uint32_t slot = atomicAdd(&s_own_other, 1u);
                if (slot < 2) s_own_q[slot] = (uint32_t)(b0 + bit);
            }
        }
        if (rd == 0) GJ_TRACE_M(1);
        // ---- the places of the markers in the workgroup's list: their order is (piece, wave, lane)
        __syncthreads();
        uint32_t tot;
        const uint32_t blk = tid < ITERS * 4 ? s_blk[tid] : 0u;
        const uint32_t bincl = gj_wg256_incl_scan(blk, s_tmp, &tot);
        if (tid < ITERS * 4) s_blk[tid] = bincl - blk;
        __syncthreads();
        const uint32_t oq0 = s_own_q[0], oq1 = s_own_q[1]; // (as far as they are known: a marker of a later round has none of these behind it)
        uint32_t a_0 = 0, a_1 = 0;
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            uint32_t slot = total + s_blk[it * 4 + wave] + winc[it] - (uint32_t)__builtin_popcount(rst[it]);
            uint32_t nm = num[it];
            for (uint32_t m = rst[it]; m; m &= m - 1, nm >>= 3) {
                const uint32_t o = (uint32_t)(round0 - part0) + 4096u * it + 16u * (uint32_t)tid + (uint32_t)__builtin_ctz(m); // offset in the workgroup's part
                if (slot < (uint32_t)GJ_SCAN_LIST) list[slot] = o | ((nm & 7u) << 24);
                if (slot == total + tot - 1) rec[1] = (uint32_t)part0 + o;
                slot++;
                a_0 += (uint32_t)part0 + o > oq0 ? 1u : 0u;
                a_1 += (uint32_t)part0 + o > oq1 ? 1u : 0u;
            }
        }
        if (a_0) atomicAdd(&s_own_after[0], a_0);
        if (a_1) atomicAdd(&s_own_after[1], a_1);
        total += tot;
        last = tot ? 1u : last;
    }
    diff = (uint32_t)__syncthreads_or((int)diff); // (also: s_own_after complete)
    if (me == 0 && hdr_ref != nullptr && tid == 0) hsum->header_differs = diff ? 1u : 0u;
    // ---- the record (the other markers in the order of their positions)
    const bool too_many = total > (uint32_t)GJ_SCAN_LIST;
    const uint32_t n_own = min(s_own_other, 2u);
    const bool swap = n_own > 1 && s_own_q[1] < s_own_q[0];
    if (tid < 2 && (uint32_t)tid < n_own) { // code and the 8 bytes behind it
        const int src = swap ? 1 - tid : tid;
        const uint32_t q = s_own_q[src];
        uint32_t b[2] = {0, 0};
        for (int k = 0; k < 8; k++)
            if ((uint64_t)q + 2 + k < size) b[k >> 2] |= (uint32_t)jpeg[q + 2 + k] << (8 * (k & 3));
        rec[4 + 6 * tid] = q;
        rec[5 + 6 * tid] = jpeg[q + 1];
        rec[6 + 6 * tid] = s_own_after[src];
        rec[7 + 6 * tid] = b[0];
        rec[8 + 6 * tid] = b[1];
    }
    if (tid == 2) {
        rec[0] = total;
        if (!last || too_many) rec[1] = 0;
        rec[2] = n_own;
        rec[3] = (too_many || s_own_other > 2u) ? 1u : 0u;
    }
    GJ_TRACE_M(2);
}

__global__ __launch_bounds__(256) void k_marker_table(const gj_geom g, const uint8_t* __restrict__ jpeg, const uint64_t begin, uint64_t size, const uint32_t part_bytes,
                                                      const uint32_t* __restrict__ recs, const uint32_t* __restrict__ lists, uint32_t* __restrict__ wg_maxlen /* host memory */,
                                                      gj_scan_summary* __restrict__ sum, gj_scan_summary* __restrict__ hsum /* host memory: what the host reads */,
                                                      uint32_t* __restrict__ seg_pos, uint32_t* __restrict__ seg_len, uint32_t* __restrict__ seg_index,
                                                      const uint32_t max_segments, const GjScanBatch B)
{
    if (B.sizes != nullptr) {
        const size_t z = blockIdx.z;
        jpeg += z * B.jpeg;
        size = B.sizes[z];
        recs += z * B.scratch; lists += z * B.scratch;
        wg_maxlen += z * B.maxlen;
        sum += z; hsum += z;
        seg_pos += z * B.seg; seg_len += z * B.seg; seg_index += z * B.seg;
    }
    __shared__ uint32_t s_mpos[GJ_SCAN_LIST];              // the workgroup's restart markers in order: offset in its part | code & 7 << 24
    __shared__ uint32_t s_tmp[4];
    __shared__ uint32_t s_opos[GJ_SCAN_MAX_OTHER], s_ocode[GJ_SCAN_MAX_OTHER], s_olen[GJ_SCAN_MAX_OTHER], s_orank[GJ_SCAN_MAX_OTHER]; // other markers so far
    __shared__ uint32_t s_ob[GJ_SCAN_MAX_OTHER][2];                                                                                  // ... the 8 bytes behind their codes
    __shared__ uint32_t s_nother, s_own_slot[2], s_own_after[2], s_own_n, s_tot;
    __shared__ int s_prev_wg;
    __shared__ uint32_t s_prev_last, s_maxlen, s_before, s_bad;
    __shared__ uint32_t s_start[GJ_MAX_COMP + 1], s_end[GJ_MAX_COMP + 1], s_first[GJ_MAX_COMP + 1];
    __shared__ int s_scans, s_open, s_status;
    __shared__ uint32_t s_geo_first[GJ_MAX_COMP + 1], s_geo_limit[GJ_MAX_COMP + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t me = blockIdx.x, nwg = gridDim.x;
    const uintptr_t a0 = (reinterpret_cast<uintptr_t>(jpeg) + begin) & ~(uintptr_t)15;
    const uint64_t part0 = (uint64_t)(a0 - reinterpret_cast<uintptr_t>(jpeg)) + (uint64_t)me * part_bytes;
    GJ_TRACE_M(3);
    if (tid == 0) { s_nother = 0; s_prev_wg = -1; s_prev_last = 0; s_maxlen = 0; s_bad = 0; s_own_n = 0; s_tot = 0; }
    if (tid <= GJ_MAX_COMP) { // scan i carries component i when the stream is not interleaved (src/gpujpeg_reader.c:1345)
        const int c = tid < g.comp_count ? tid : 0;
        s_geo_first[tid] = g.interleaved ? 0u : (uint32_t)g.comp[c].first_segment;
        s_geo_limit[tid] = g.interleaved ? (uint32_t)g.segment_count : (uint32_t)g.comp[c].segment_count;
    }
    // ---- one trip: this workgroup's list (as much of it as it can hold) and the records of the workgroups in front (lane i: workgroup i;
    // more than 256 of them: several passes) and its own
    {
        const uint4* l4 = reinterpret_cast<const uint4*>(lists + (size_t)me * GJ_SCAN_LIST);
        uint4* m4 = reinterpret_cast<uint4*>(s_mpos);
        m4[tid] = l4[tid];
        m4[tid + 256] = l4[tid + 256];
    }
    __syncthreads();
    uint32_t before = 0, obefore = 0; // restart / other markers of the passes so far
    for (uint32_t base = 0; base <= me; base += 256) {
        const uint32_t i = base + (uint32_t)tid;
        const bool have = i <= me;
        uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
        if (have) {
            const uint4* r4 = reinterpret_cast<const uint4*>(recs + (size_t)i * GJ_SCAN_REC_WORDS);
            r0 = r4[0];
            r1 = r4[1];
            r2 = r4[2];
            r3 = r4[3];
        }
        const uint32_t n = r0.x, no = min(r0.z, 2u);
        if (have && r0.w) s_bad = 1;
        uint32_t btot, otot;
        const uint32_t incl = gj_wg256_incl_scan(n, s_tmp, &btot) + before;    // restart markers up to and including workgroup i
        const uint32_t oincl = gj_wg256_incl_scan(no, s_tmp, &otot) + obefore; // other markers ...: they are in the order of the workgroups = of their positions
        if (have && i < me && n) atomicMax(&s_prev_wg, (int)i);
        if (have)
            for (uint32_t q = 0; q < no; q++) {
                const GjOther o = gj_rec_other(r1, r2, r3, q); // position, code, restart markers behind it, the 8 bytes behind the code
                const uint32_t slot = oincl - no + q;
                if (slot < GJ_SCAN_MAX_OTHER) {
                    s_opos[slot] = o.pos;
                    s_ocode[slot] = o.code & 0xFFu;
                    s_olen[slot] = gj_other_len(o);
                    s_orank[slot] = incl - o.after; // restart markers in front of it
                    s_ob[slot][0] = o.b03;
                    s_ob[slot][1] = o.b47;
                }
                if (i == me) { s_own_slot[q] = slot; s_own_after[q] = o.after; }
            }
        if (have && i == me) { s_before = incl - n; s_nother = oincl; s_own_n = no; s_tot = n; }
        __syncthreads();
        if (have && i < me && (int)i == s_prev_wg) s_prev_last = r0.y; // the last restart marker in front of this workgroup
        before += btot;
        obefore += otot;
        __syncthreads();
    }
    GJ_TRACE_M(4);
    const uint32_t tot = s_tot, n_own = s_own_n;
    const bool too_many = tot > (uint32_t)GJ_SCAN_LIST;
    // ---- the scan structure as far as it is known here: a walk through the other markers in order (wave 0, scalar values)
    if (wave == 0) {
        const uint32_t n_other = min(s_nother, (uint32_t)GJ_SCAN_MAX_OTHER);
        int scans = 0, status = 0;
        uint32_t start = (uint32_t)begin, first = 0;
        for (uint32_t kk = 0; kk < n_other && scans < GJ_MAX_COMP; kk++) {
            const uint32_t p = s_opos[kk], m = s_ocode[kk], mlen = s_olen[kk], rk = s_orank[kk];
            if (p < start) continue; // lies inside a header we already skipped
            if (lane == 0) { s_start[scans] = start; s_end[scans] = p; s_first[scans] = first; }
            scans++;
            if (m == 0xDA) { // next scan
                start = p + 2 + mlen;
                first = rk;
                continue;
            }
            if (m == 0xD9) { status = 1; break; } // EOI: done
            status = 2;                            // something else between scans: let the host walk it
            break;
        }
        int open = 0;
        if (status == 0) {
            if (scans < GJ_MAX_COMP) { // the scan that is still running behind the last SOS
                if (lane == 0) { s_start[scans] = start; s_end[scans] = 0xFFFFFFFFu; s_first[scans] = first; }
                open = 1;
            }
            status = 3; // no EOI seen (yet)
        }
        if (s_nother > (uint32_t)GJ_SCAN_MAX_OTHER || s_bad) status = 2;
        if (lane == 0) { s_scans = scans; s_open = open; s_status = status; }
    }
    __syncthreads();
    GJ_TRACE_M(5);
    const int scans = s_scans, nsc = scans + s_open;
    uint32_t irregular = too_many ? 1u : 0u, maxlen = 0;
    const uint32_t rank0 = s_before, prev_last = s_prev_last;
    const bool has_prev = s_prev_wg >= 0;
    // ---- this workgroup's restart markers: marker number k of scan sc ends segment k
    for (uint32_t i = (uint32_t)tid; i < tot && !too_many; i += 256) {
        const uint32_t p = (uint32_t)part0 + (s_mpos[i] & 0xFFFFFFu), nm = s_mpos[i] >> 24, rk = rank0 + i;
        int sc = -1;
#pragma unroll
        for (int q = 0; q <= GJ_MAX_COMP; q++)
            if (q < nsc && p >= s_start[q] && p < s_end[q]) sc = q;
        if (sc < 0 || sc >= GJ_MAX_COMP) { irregular = 1u; continue; } // (a restart marker outside every scan)
        if (rk < s_first[sc]) { irregular = 1u; continue; }
        const uint32_t k = rk - s_first[sc];                                           // the marker's number inside its scan = the segment it ends
        const uint32_t before_p = i > 0 ? (uint32_t)part0 + (s_mpos[i - 1] & 0xFFFFFFu) : prev_last; // the marker in front of this one
        if (k > 0 && i == 0 && !has_prev) { irregular = 1u; continue; }
        const uint32_t from = (k == 0) ? s_start[sc] : before_p + 2;
        const uint32_t first = s_geo_first[sc], limit = s_geo_limit[sc];
        const uint32_t e = rk + (uint32_t)sc;
        // the marker that ends segment k must be RST(k mod 8): anything else is a stream the reference reader treats specially, which
        // the host walk reproduces
        if (nm != (k & 7u)) irregular = 1u;
        if (e < max_segments) {
            seg_pos[e] = from;
            seg_len[e] = p > from ? p - from : 0;
            seg_index[e] = k < limit ? first + k : 0xFFFFFFFFu;
            if (p > from) maxlen = max(maxlen, p - from);
        }
    }
    // ---- this workgroup's other markers: the one that ends a scan closes the scan's last segment (which must not be empty)
    if ((uint32_t)tid < n_own && s_own_slot[tid] < GJ_SCAN_MAX_OTHER) {
        const uint32_t slot = s_own_slot[tid], p = s_opos[slot], rk = s_orank[slot]; // rk = restart markers in front of it
        for (int sc = 0; sc < scans; sc++)
            if (s_end[sc] == p) {
                const uint32_t c_s = rk - s_first[sc]; // restart markers of the scan
                // the restart marker in front of p: the last one of this workgroup in front of it, or the last one of the workgroups in front
                uint32_t from = s_start[sc];
                if (c_s > 0) {
                    const uint32_t mine_before = rk - rank0; // (of this workgroup's markers)
                    from = (mine_before > 0 ? (uint32_t)part0 + (s_mpos[mine_before - 1] & 0xFFFFFFu) : prev_last) + 2;
                }
                const uint32_t e = rk + (uint32_t)sc;
                const uint32_t first = s_geo_first[sc], limit = s_geo_limit[sc];
                if (p <= from && c_s > 0) irregular = 1u; // (an empty segment in front of the end of a scan)
                if (e < max_segments) {
                    const uint32_t len = p > from ? p - from : 0;
                    seg_pos[e] = from;
                    seg_len[e] = len;
                    seg_index[e] = c_s < limit ? first + c_s : 0xFFFFFFFFu;
                    maxlen = max(maxlen, len);
                }
            }
        // what the host validates about it
        hsum->other_pos[slot] = p;
        hsum->other_code[slot] = (uint8_t)s_ocode[slot];
        const uint32_t ob[4] = {s_ob[slot][0], s_ob[slot][1], 0, 0}; // (whole words to the host's memory; the host looks at the first eight bytes)
        for (int q = 0; q < 4; q++) reinterpret_cast<uint32_t*>(hsum->other_bytes[slot])[q] = ob[q];
        hsum->other_after[slot] = s_own_after[tid];
    }
    if (maxlen) atomicMax(&s_maxlen, maxlen);
    irregular = (uint32_t)__syncthreads_or((int)irregular);
    GJ_TRACE_M(6);
    if (tid == 0) {
        wg_maxlen[me] = s_maxlen; // (the host takes the maximum)
        if (irregular) hsum->rst_irregular = 1u; // (the host cleared it before the launch)
    }
    if (me == nwg - 1 && tid == 0) { // the last workgroup has seen everything
        const uint32_t total = rank0 + tot;
        sum->segment_count = scans ? total + (uint32_t)scans : 0u; // (speculative launches: the entropy decoder reads the count from here)
        hsum->rst_count = total;
        hsum->other_count = s_nother;
        hsum->scan_count = (uint32_t)scans;
        hsum->status = (uint32_t)s_status;
        hsum->segment_count = scans ? total + (uint32_t)scans : 0u;
        for (int sc = 0; sc < scans; sc++) { hsum->scan_start[sc] = s_start[sc]; hsum->scan_end[sc] = s_end[sc]; }
    }
    GJ_TRACE_M(7);
}

// 4 KB pieces per workgroup and round for a stream of this size (1 .. 16), and rounds (1 .. 4): 256 workgroups up to 16 MB; beyond that MORE
// workgroups of one round each (up to 1024 at 64 MB) rather than rounds -- a round is a dependent trip to memory plus a workgroup prefix sum, and 256
// workgroups are one per CU: config 4's 43 MB took 3 rounds on 225 workgroups, 110 us (round 4, first session), against 672 workgroups of one round
static void gj_scan_shape(uint64_t bytes, uint32_t* iters, uint32_t* rounds)
{
    const uint64_t per_wg = (bytes + 16 + GJ_SCAN_WGS - 1) / GJ_SCAN_WGS;
    uint32_t it = 1, rd = 1;
    while (it < 16 && 4096ull * it < per_wg) it *= 2;
    if (4096ull * it < per_wg) { // (more than 16 MB)
        const uint64_t wgs1 = (bytes + 16 + 65535) / 65536;
        rd = (uint32_t)((wgs1 + GJ_SCAN_WGS_MAX - 1) / GJ_SCAN_WGS_MAX);
        if (rd > 4) rd = 4;
    }
    *iters = it;
    *rounds = rd;
}

extern "C" int gj_hip_find_segments(const gj_geom* g, const uint8_t* d_jpeg, uint64_t begin, uint64_t size, uint32_t* d_seg_pos,
                                    uint32_t* d_seg_len, uint32_t* d_seg_index, uint32_t max_segments, uint32_t* d_scratch,
                                    gj_scan_summary* d_summary, const uint8_t* d_hdr_ref, uint32_t hdr_n, gj_scan_summary* h_summary,
                                    uint32_t* h_maxlen_parts, uint32_t maxlen_capacity, uint32_t* maxlen_part_count, gj_stream_t stream,
                                    const gj_tuning* tune)
{
    return gj_hip_find_segments_batch(g, d_jpeg, begin, size, d_seg_pos, d_seg_len, d_seg_index, max_segments, d_scratch, d_summary, d_hdr_ref, hdr_n, h_summary,
                                      h_maxlen_parts, maxlen_capacity, maxlen_part_count, stream, tune, nullptr);
}

static int gj_find_segments_impl(const gj_geom* g, const uint8_t* d_jpeg, uint64_t begin, uint64_t size, uint32_t* d_seg_pos,
                                 uint32_t* d_seg_len, uint32_t* d_seg_index, uint32_t max_segments, uint32_t* d_scratch,
                                 gj_scan_summary* d_summary, const uint8_t* d_hdr_ref, uint32_t hdr_n, gj_scan_summary* h_summary,
                                 uint32_t* h_maxlen_parts, uint32_t maxlen_capacity, uint32_t* maxlen_part_count, gj_stream_t stream,
                                 const gj_tuning* tune, const gj_batch* batch, gj_scan_deferred* defer);

extern "C" int gj_hip_find_segments_batch(const gj_geom* g, const uint8_t* d_jpeg, uint64_t begin, uint64_t size, uint32_t* d_seg_pos,
                                          uint32_t* d_seg_len, uint32_t* d_seg_index, uint32_t max_segments, uint32_t* d_scratch,
                                          gj_scan_summary* d_summary, const uint8_t* d_hdr_ref, uint32_t hdr_n, gj_scan_summary* h_summary,
                                          uint32_t* h_maxlen_parts, uint32_t maxlen_capacity, uint32_t* maxlen_part_count, gj_stream_t stream,
                                          const gj_tuning* tune, const gj_batch* batch)
{
    return gj_find_segments_impl(g, d_jpeg, begin, size, d_seg_pos, d_seg_len, d_seg_index, max_segments, d_scratch, d_summary, d_hdr_ref, hdr_n, h_summary,
                                 h_maxlen_parts, maxlen_capacity, maxlen_part_count, stream, tune, batch, nullptr);
}

extern "C" int gj_hip_find_segments_deferred(const gj_geom* g, const uint8_t* d_jpeg, uint64_t begin, uint64_t size, uint32_t* d_seg_pos,
                                             uint32_t* d_seg_len, uint32_t* d_seg_index, uint32_t max_segments, uint32_t* d_scratch,
                                             gj_scan_summary* d_summary, const uint8_t* d_hdr_ref, uint32_t hdr_n, gj_scan_summary* h_summary,
                                             uint32_t* h_maxlen_parts, uint32_t maxlen_capacity, uint32_t* maxlen_part_count, gj_stream_t stream,
                                             const gj_tuning* tune, gj_scan_deferred* defer)
{
    defer->valid = 0;
    return gj_find_segments_impl(g, d_jpeg, begin, size, d_seg_pos, d_seg_len, d_seg_index, max_segments, d_scratch, d_summary, d_hdr_ref, hdr_n, h_summary,
                                 h_maxlen_parts, maxlen_capacity, maxlen_part_count, stream, tune, nullptr, defer);
}

// the second launch of the scan: from gj_find_segments_impl, or later from gj_hip_decode (gj_scan_deferred)
void gj_launch_marker_table(const gj_geom* g, const uint8_t* d_jpeg, const gj_scan_deferred* sc, const GjScanBatch& B, unsigned frames, hipStream_t st)
{
    hipLaunchKernelGGL(k_marker_table, dim3(sc->wgs, 1, frames), dim3(256), 0, st, *g, d_jpeg, sc->begin, sc->size, sc->part_bytes, sc->recs, sc->lists, sc->h_maxlen_parts,
                       sc->d_summary, sc->h_summary, sc->d_seg_pos, sc->d_seg_len, sc->d_seg_index, sc->max_segments + GJ_MAX_COMP, B);
}
void gj_launch_marker_table_deferred(const gj_dec_job* job, hipStream_t st)
{
    const GjScanBatch B = {nullptr, 0, 0, 0, 0};
    gj_launch_marker_table(&job->g, job->d_jpeg, &job->scan, B, 1, st);
    gj_debug_stage(job->tune.debug_sync != 0, st, "k_marker_table (deferred)");
}

static int gj_find_segments_impl(const gj_geom* g, const uint8_t* d_jpeg, uint64_t begin, uint64_t size, uint32_t* d_seg_pos,
                                 uint32_t* d_seg_len, uint32_t* d_seg_index, uint32_t max_segments, uint32_t* d_scratch,
                                 gj_scan_summary* d_summary, const uint8_t* d_hdr_ref, uint32_t hdr_n, gj_scan_summary* h_summary,
                                 uint32_t* h_maxlen_parts, uint32_t maxlen_capacity, uint32_t* maxlen_part_count, gj_stream_t stream,
                                 const gj_tuning* tune, const gj_batch* batch, gj_scan_deferred* defer)
{
    hipStream_t st = (hipStream_t)stream;
    GjScanBatch B = {nullptr, 0, 0, 0, 0};
    unsigned frames = 1;
    if (batch && batch->d_sizes != nullptr) { // (also a batch of one frame: its size comes from d_sizes like the others')
        if ((batch->jpeg & 15u) != 0 || batch->count > 65535u || batch->count < 1u) return -1;
        B.sizes = batch->d_sizes;
        B.jpeg = batch->jpeg;
        B.seg = batch->seg;
        B.scratch = batch->scratch;
        B.maxlen = batch->maxlen;
        frames = batch->count;
    }
    if (size <= begin || size > 0xFFFFFFF0ull) return -1;
    uint32_t iters, rounds;
    gj_scan_shape(size - begin, &iters, &rounds);
    if (tune->scan_shape > 0) { // (developer switch: a given shape)
        const uint32_t it = (uint32_t)tune->scan_shape / 100u, rd = (uint32_t)tune->scan_shape % 100u;
        // (only while the scratch -- records and lists for gj_hip_find_segments_max_chunks() workgroups -- and the host's array hold that many
        // workgroups: a forced shape of small parts on a long stream used to write behind the scratch, found under AddressSanitizer in round 4)
        const uint64_t forced = (size - begin + 16 + 4096ull * it * rd - 1) / (4096ull * it * rd);
        if ((it == 1 || it == 2 || it == 4 || it == 8 || it == 16) && rd >= 1 && rd <= 4 && forced <= maxlen_capacity &&
            forced < gj_hip_find_segments_max_chunks(begin, size)) {
            iters = it;
            rounds = rd;
        }
    }
    const uint64_t lead = (reinterpret_cast<uintptr_t>(d_jpeg) + begin) & 15u, part = 4096ull * iters * rounds;
    const uint32_t wgs = (uint32_t)((size - begin + lead + part - 1) / part);
    if (wgs > maxlen_capacity || wgs >= gj_hip_find_segments_max_chunks(begin, size)) return -1;
    *maxlen_part_count = wgs;
    uint32_t* recs = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(d_scratch) + 15) & ~(uintptr_t)15); // [wgs] records, then [wgs] lists
    uint32_t* lists = recs + (size_t)gj_hip_find_segments_max_chunks(begin, size) * GJ_SCAN_REC_WORDS;
    auto kern = iters == 1 ? k_marker_scan<1> : iters == 2 ? k_marker_scan<2> : iters == 4 ? k_marker_scan<4> : iters == 8 ? k_marker_scan<8> : k_marker_scan<16>;
    hipLaunchKernelGGL(kern, dim3(wgs, 1, frames), dim3(256), 0, st, d_jpeg, begin, size, rounds, recs, lists, h_summary, d_hdr_ref, hdr_n, B);
    gj_debug_stage(tune->debug_sync != 0, st, "k_marker_scan");
    gj_scan_deferred sc;
    sc.valid = 1;
    sc.wgs = wgs; sc.part_bytes = (uint32_t)part;
    sc.begin = begin; sc.size = size;
    sc.recs = recs; sc.lists = lists;
    sc.d_seg_pos = d_seg_pos; sc.d_seg_len = d_seg_len; sc.d_seg_index = d_seg_index;
    sc.max_segments = max_segments;
    sc.d_summary = d_summary; sc.h_summary = h_summary;
    sc.h_maxlen_parts = h_maxlen_parts; sc.maxlen_capacity = maxlen_capacity; sc.maxlen_part_count = maxlen_part_count;
    sc.folded = nullptr;
    if (defer != nullptr && frames == 1 && wgs <= 256u) { // (the token decoder reads the records of up to 256 scanning workgroups in one pass)
        *defer = sc;
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    gj_launch_marker_table(g, d_jpeg, &sc, B, frames, st);
    gj_debug_stage(tune->debug_sync != 0, st, "k_marker_table");
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// scratch: a record and a list per workgroup
extern "C" size_t gj_hip_find_segments_scratch_words(uint64_t begin, uint64_t size, uint32_t max_segments)
{
    (void)max_segments;
    return (size_t)gj_hip_find_segments_max_chunks(begin, size) * (GJ_SCAN_REC_WORDS + GJ_SCAN_LIST) + 8;
}

// workgroups the marker scan cuts [begin, size) into at most (capacity of h_maxlen_parts)
extern "C" size_t gj_hip_find_segments_max_chunks(uint64_t begin, uint64_t size)
{
    const uint64_t bytes = size - begin + 16;
    uint32_t it, rd;
    gj_scan_shape(size - begin, &it, &rd); // (the shape gj_hip_find_segments will choose)
    const uint64_t wgs = (bytes + 4096ull * it * rd - 1) / (4096ull * it * rd);
    // (at least 512: a forced shape -- GJ_SCAN_SHAPE, tests -- may cut a small stream into more parts than its size would, which is how the
    // passes of k_marker_table over more than 256 records are exercised without a 16 MB stream)
    return (size_t)(wgs > 2 * GJ_SCAN_WGS ? wgs : 2 * GJ_SCAN_WGS) + 1;
}
