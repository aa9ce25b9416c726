// gj_dec_markers.hip -- MI355X (gfx950, wave64) JPEG decoder: device-side segment discovery (marker scan)
// (part of the decoder's device code, see gj_dec_internal.h for the map of the files)
#include "gj_dec_internal.h"
#include <stddef.h>

// ================================================================================================
// Device-side segment discovery. Inside entropy-coded data 0xFF is followed by 0x00 (stuffing),
// by 0xD0..0xD7 (restart marker = segment boundary) or by the marker that ends the scan. ONE launch turns the bytes
// [begin, size) into the (offset, length, geometric index) table the entropy decoders consume, without the host touching the
// stream (the reference walks it with memchr and copies every segment, src/gpujpeg_reader.c:1039-1155), and reads the stream once:
//
//   * at most 256 workgroups, each with a contiguous part of the stream, every lane with a contiguous 16 .. 1024 bytes of it (16-byte
//     loads, all of them in flight together): the lane's restart markers are bits in registers, the few other markers (SOS of the later
//     scans, EOI) go to a two-entry list;
//   * a workgroup publishes ONE 64-byte record -- restart markers it holds, position of its last one, its other markers with the number
//     of restart markers behind each -- and reads the records of all workgroups IN FRONT of it (a lane each; workgroups are dispatched
//     in order, so waiting for a predecessor cannot deadlock). Every word of a record carries the call's number: no flag to wait for,
//     no clearing between calls;
//   * that is all a table entry needs: restart marker number k of a scan ends segment k, the entry's index is the marker's rank among
//     all markers in front of it plus its scan's number, the segment begins behind the marker in front. The segment at the end of a
//     scan is written by the owner of the marker that ends the scan. The last workgroup has seen every record: it writes the summary
//     the host validates (gj_scan_summary) straight to pinned host memory.
//
// Round 3 used two launches that both read the stream (28 us for the 7.4 MB of an 8K frame, 76 us for config 4's 43 MB).
// ================================================================================================
#ifdef GJ_TRACE_PHASES
static __device__ unsigned long long* gj_trace_buf_m;
extern "C" GJ_HIP_API int gj_hip_trace_set_markers(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(gj_trace_buf_m), &p, sizeof p) == hipSuccess ? 0 : -1; }
#define GJ_TRACE_M(slot) do { if (threadIdx.x == 0 && gj_trace_buf_m) gj_trace_buf_m[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GJ_TRACE_M(slot) ((void)0)
#endif
#define GJ_SCAN_LIST 2048    // restart markers a workgroup may hold (beyond that the host walks the stream)
#define GJ_SCAN_WGS 256      // workgroups (= records every workgroup reads) up to 64 MB of stream; larger streams take more of them
#define GJ_SCAN_REC_WORDS 8  // 64-bit words of a record

// A record word: the call's number in the upper 16 bits, 48 bits of payload.
//   [0] flags << 40 | other markers << 32 | restart markers      flags: 1 = more restart markers than the list holds / more than two other markers
//   [1] position of the last restart marker
//   [2] other marker 0: position << 16 | the 16 bits behind its code (segment length)      [3] code << 32 | restart markers of the workgroup behind it
//   [4], [5] other marker 1
__device__ __forceinline__ void gj_rec_put(uint64_t* rec, const int w, const uint32_t epoch, const uint64_t payload)
{
    __hip_atomic_store(rec + w, ((uint64_t)(epoch & 0xFFFFu) << 48) | (payload & 0xFFFFFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t gj_rec_get(const uint64_t* rec, const int w)
{
    return __hip_atomic_load(rec + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ITERS x 16 bytes per lane
template <int ITERS>
__global__ __launch_bounds__(256) void k_markers(const gj_geom g, const uint8_t* __restrict__ jpeg, const uint64_t begin, const uint64_t size, uint64_t* __restrict__ recs,
                                                 const uint32_t epoch, uint32_t* __restrict__ wg_maxlen /* host memory */, gj_scan_summary* __restrict__ sum,
                                                 gj_scan_summary* __restrict__ hsum /* host memory: what the host reads */, const uint8_t* __restrict__ hdr_ref,
                                                 const uint32_t hdr_n, uint32_t* __restrict__ seg_pos, uint32_t* __restrict__ seg_len,
                                                 uint32_t* __restrict__ seg_index, const uint32_t max_segments)
{
    __shared__ uint32_t s_mpos[GJ_SCAN_LIST];              // the workgroup's restart markers in order: offset in its part | code & 7 << 24
    __shared__ uint32_t s_tmp[4];
    __shared__ uint32_t s_opos[GJ_SCAN_MAX_OTHER], s_ocode[GJ_SCAN_MAX_OTHER], s_olen[GJ_SCAN_MAX_OTHER], s_orank[GJ_SCAN_MAX_OTHER]; // other markers so far
    __shared__ uint32_t s_nother, s_own_other, s_own_q[2], s_own_after[2], s_own_slot[2];
    __shared__ int s_prev_wg;
    __shared__ uint32_t s_prev_last, s_maxlen, s_before, s_bad;
    __shared__ uint32_t s_start[GJ_MAX_COMP + 1], s_end[GJ_MAX_COMP + 1], s_first[GJ_MAX_COMP + 1];
    __shared__ int s_scans, s_open, s_status;
    __shared__ uint32_t s_geo_first[GJ_MAX_COMP + 1], s_geo_limit[GJ_MAX_COMP + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t me = blockIdx.x, nwg = gridDim.x;
    constexpr uint32_t LB = 16u * ITERS; // bytes per lane
    // the parts are cut on 16-byte addresses: the first lane's begins in front of `begin`
    const uintptr_t a0 = (reinterpret_cast<uintptr_t>(jpeg) + begin) & ~(uintptr_t)15;
    const uint64_t part0 = (uint64_t)(a0 - reinterpret_cast<uintptr_t>(jpeg)) + (uint64_t)me * 256u * LB; // (may wrap below zero for the very first bytes: never addressed)
    const uint64_t b0 = part0 + (uint64_t)tid * LB;                                                          // this lane's first byte
    GJ_TRACE_M(0);
    if (tid == 0) { s_nother = 0; s_own_other = 0; s_prev_wg = -1; s_prev_last = 0; s_maxlen = 0; s_bad = 0; }
    if (tid < 2) s_own_after[tid] = 0;
    if (tid <= GJ_MAX_COMP) { // scan i carries component i when the stream is not interleaved (src/gpujpeg_reader.c:1345)
        const int c = tid < g.comp_count ? tid : 0;
        s_geo_first[tid] = g.interleaved ? 0u : (uint32_t)g.comp[c].first_segment;
        s_geo_limit[tid] = g.interleaved ? (uint32_t)g.segment_count : (uint32_t)g.comp[c].segment_count;
    }
    // ---- the lane's bytes: ITERS 16-byte loads and the four bytes behind them
    uint32_t w[ITERS * 4 + 1];
    {
        const uint4* src = reinterpret_cast<const uint4*>(jpeg + b0);
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (b0 + 16u * it < size && b0 + 16u * it + 16u > begin) v = src[it]; // (whole 16-byte pieces: the buffer is readable to a multiple of 16)
            w[4 * it] = v.x; w[4 * it + 1] = v.y; w[4 * it + 2] = v.z; w[4 * it + 3] = v.w;
        }
        w[ITERS * 4] = b0 + LB < size ? *reinterpret_cast<const uint32_t*>(jpeg + b0 + LB) : 0u;
    }
    uint32_t diff = 0;
    if (me == 0 && hdr_ref != nullptr) // does the stream start with the header the host assumed? (speculative launch)
        for (uint32_t i = tid; i < hdr_n; i += 256) diff |= jpeg[i] != hdr_ref[i];
    // ---- markers: bit b of rst[it] / oth[it] = byte 16 it + b of the lane starts a restart / another marker; the restart markers' codes
    uint32_t rst[ITERS], num[ITERS], mine = 0, own_other = 0;
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        uint32_t r = 0, o = 0, nm = 0, nn = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t x = w[4 * it + q], xn = w[4 * it + q + 1];
            if ((~x - 0x01010101u) & x & 0x80808080u) { // 0xFF bytes are rare: one test for the four (a zero byte in ~x; a borrow can only add a false alarm)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t b = (x >> (8 * k)) & 0xFFu;
                    const uint32_t nx = k < 3 ? (x >> (8 * k + 8)) & 0xFFu : xn & 0xFFu;
                    const uint64_t p = b0 + (uint64_t)(16 * it + 4 * q + k);
                    if (b == 0xFFu && nx != 0u && nx != 0xFFu && p >= begin && p + 1 < size) {
                        if ((nx & 0xF8u) == 0xD0u) {
                            r |= 1u << (4 * q + k);
                            nm |= (nx & 7u) << (3u * nn);
                            nn++;
                        } else {
                            o |= 1u << (4 * q + k);
                        }
                    }
                }
            }
        }
        rst[it] = r;
        num[it] = nm;
        mine += (uint32_t)__builtin_popcount(r);
        while (o) { // scan boundary material (rare): position, code and the 16 bits behind it
            const uint32_t bit = (uint32_t)__builtin_ctz(o);
            o &= o - 1;
            const uint32_t slot = atomicAdd(&s_own_other, 1u);
            if (slot < 2) s_own_q[slot] = (uint32_t)(b0 + 16u * it + bit);
            own_other++;
        }
    }
    GJ_TRACE_M(1);
    // ---- the workgroup's restart markers in order (a lane's part is contiguous: one prefix sum)
    uint32_t tot;
    uint32_t r0 = gj_wg256_incl_scan(mine, s_tmp, &tot) - mine; // (also: s_own_q complete)
    const bool too_many = tot > (uint32_t)GJ_SCAN_LIST;
    if (!too_many) {
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            uint32_t nm = num[it];
            for (uint32_t m = rst[it]; m; m &= m - 1, nm >>= 3) {
                const uint32_t o = (uint32_t)tid * LB + 16u * it + (uint32_t)__builtin_ctz(m);
                s_mpos[r0++] = o | ((nm & 7u) << 24);
            }
        }
    }
    // the workgroup's other markers: in order, with the restart markers behind each
    const uint32_t n_own = min(s_own_other, 2u);
    uint32_t oq0 = n_own > 0 ? s_own_q[0] : 0xFFFFFFFFu, oq1 = n_own > 1 ? s_own_q[1] : 0xFFFFFFFFu;
    if (oq1 < oq0) { const uint32_t t = oq0; oq0 = oq1; oq1 = t; }
    if (n_own) {
        uint32_t a_0 = 0, a_1 = 0;
#pragma unroll
        for (int it = 0; it < ITERS; it++)
            for (uint32_t m = rst[it]; m; m &= m - 1) {
                const uint32_t p = (uint32_t)(b0 + 16u * it) + (uint32_t)__builtin_ctz(m);
                a_0 += p > oq0 ? 1u : 0u;
                a_1 += p > oq1 ? 1u : 0u;
            }
        if (a_0) atomicAdd(&s_own_after[0], a_0);
        if (a_1 && n_own > 1) atomicAdd(&s_own_after[1], a_1);
    }
    diff = (uint32_t)__syncthreads_or((int)diff); // (also: s_mpos, s_own_after complete)
    if (me == 0 && hdr_ref != nullptr && tid == 0) hsum->header_differs = diff ? 1u : 0u;
    GJ_TRACE_M(2);
    // ---- publish
    uint64_t* const myrec = recs + (size_t)me * GJ_SCAN_REC_WORDS;
    const uint32_t flags = (too_many || s_own_other > 2u) ? 1u : 0u;
    uint32_t oc[2] = {0, 0}, ol[2] = {0, 0};
    if (tid < 2 && (uint32_t)tid < n_own) { // code and the 16 bits behind it
        const uint32_t q = tid == 0 ? oq0 : oq1;
        const uint32_t code = jpeg[q + 1];
        const uint32_t len = ((q + 2 < size ? (uint32_t)jpeg[q + 2] : 0u) << 8) | (q + 3 < size ? (uint32_t)jpeg[q + 3] : 0u);
        gj_rec_put(myrec, 2 + 2 * tid, epoch, ((uint64_t)q << 16) | len);
        gj_rec_put(myrec, 3 + 2 * tid, epoch, ((uint64_t)code << 32) | s_own_after[tid]);
        oc[tid] = code;
        ol[tid] = len;
    }
    if (tid == 2) {
        gj_rec_put(myrec, 0, epoch, ((uint64_t)flags << 40) | ((uint64_t)n_own << 32) | tot);
        gj_rec_put(myrec, 1, epoch, (uint64_t)(tot && !too_many ? (uint32_t)part0 + (s_mpos[tot - 1] & 0xFFFFFFu) : 0u));
    }
    // ---- the records of the workgroups in front (lane i: workgroup i; more than 256 of them: several passes), and this one's own
    uint32_t before = 0, obefore = 0; // restart / other markers of the passes so far
    for (uint32_t base = 0; base <= me; base += 256) {
        const uint32_t i = base + (uint32_t)tid;
        const bool have = i <= me;
        uint64_t r[6] = {0, 0, 0, 0, 0, 0};
        const uint64_t tag = (uint64_t)(epoch & 0xFFFFu) << 48;
        for (;;) { // (a predecessor is running or done: its words arrive)
            int waiting = 0;
            if (have) {
#pragma unroll
                for (int q = 0; q < 6; q++) r[q] = gj_rec_get(recs + (size_t)i * GJ_SCAN_REC_WORDS, q);
                const uint32_t no = (uint32_t)(r[0] >> 32) & 0xFFu;
                waiting = (r[0] >> 48 << 48) != tag || (r[1] >> 48 << 48) != tag;
                if (!waiting && no > 0) waiting |= (r[2] >> 48 << 48) != tag || (r[3] >> 48 << 48) != tag;
                if (!waiting && no > 1) waiting |= (r[4] >> 48 << 48) != tag || (r[5] >> 48 << 48) != tag;
            }
            if (!__syncthreads_or(waiting)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        const uint32_t n = have ? (uint32_t)r[0] : 0u, no = have ? (uint32_t)(r[0] >> 32) & 0xFFu : 0u;
        if (have && ((r[0] >> 40) & 1u)) s_bad = 1;
        uint32_t btot, otot;
        const uint32_t incl = gj_wg256_incl_scan(n, s_tmp, &btot) + before;    // restart markers up to and including workgroup i
        const uint32_t oincl = gj_wg256_incl_scan(no, s_tmp, &otot) + obefore; // other markers ...: they are in the order of the workgroups = of their positions
        if (have && i < me && n) atomicMax(&s_prev_wg, (int)i);
        if (have)
            for (uint32_t q = 0; q < no; q++) {
                const uint32_t slot = oincl - no + q;
                if (slot < GJ_SCAN_MAX_OTHER) {
                    s_opos[slot] = (uint32_t)(r[2 + 2 * q] >> 16);
                    s_olen[slot] = (uint32_t)r[2 + 2 * q] & 0xFFFFu;
                    s_ocode[slot] = (uint32_t)(r[3 + 2 * q] >> 32) & 0xFFu;
                    s_orank[slot] = incl - (uint32_t)r[3 + 2 * q]; // restart markers in front of it
                }
                if (i == me) s_own_slot[q] = slot;
            }
        if (have && i == me) { s_before = incl - n; s_nother = oincl; }
        __syncthreads();
        if (have && i < me && (int)i == s_prev_wg) s_prev_last = (uint32_t)r[1]; // the last restart marker in front of this workgroup
        before += btot;
        obefore += otot;
        __syncthreads();
    }
    GJ_TRACE_M(3);
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
    GJ_TRACE_M(4);
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
        hsum->other_code[slot] = (uint8_t)oc[tid];
        for (int b = 0; b < 16; b++) hsum->other_bytes[slot][b] = (uint64_t)p + 2 + b < size ? jpeg[p + 2 + b] : 0;
        hsum->other_after[slot] = s_own_after[tid];
        (void)ol;
    }
    if (maxlen) atomicMax(&s_maxlen, maxlen);
    irregular = (uint32_t)__syncthreads_or((int)irregular);
    GJ_TRACE_M(5);
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
    GJ_TRACE_M(6);
}

// bytes per lane for a stream of this size: 16 .. 1024 (256 workgroups of 256 lanes up to 64 MB)
static uint32_t gj_scan_iters(uint64_t bytes)
{
    const uint64_t per_lane = (bytes + 16 + (uint64_t)GJ_SCAN_WGS * 256 - 1) / ((uint64_t)GJ_SCAN_WGS * 256);
    uint32_t it = 1;
    while (it < 64 && 16ull * it < per_lane) it *= 2;
    return it;
}

extern "C" int gj_hip_find_segments(const gj_geom* g, const uint8_t* d_jpeg, uint64_t begin, uint64_t size, uint32_t* d_seg_pos,
                                    uint32_t* d_seg_len, uint32_t* d_seg_index, uint32_t max_segments, uint32_t* d_scratch,
                                    gj_scan_summary* d_summary, const uint8_t* d_hdr_ref, uint32_t hdr_n, gj_scan_summary* h_summary,
                                    uint32_t* h_maxlen_parts, uint32_t maxlen_capacity, uint32_t* maxlen_part_count, uint32_t epoch,
                                    gj_stream_t stream, const gj_tuning* tune)
{
    hipStream_t st = (hipStream_t)stream;
    if (size <= begin || size > 0xFFFFFFF0ull) return -1;
    const uint32_t iters = gj_scan_iters(size - begin);
    const uint64_t lead = (reinterpret_cast<uintptr_t>(d_jpeg) + begin) & 15u;
    const uint32_t wgs = (uint32_t)((size - begin + lead + 256ull * 16 * iters - 1) / (256ull * 16 * iters));
    if (wgs > maxlen_capacity) return -1;
    *maxlen_part_count = wgs;
    uint64_t* recs = reinterpret_cast<uint64_t*>((reinterpret_cast<uintptr_t>(d_scratch) + 7) & ~(uintptr_t)7);
    auto kern = iters == 1 ? k_markers<1> : iters == 2 ? k_markers<2> : iters == 4 ? k_markers<4> : iters == 8 ? k_markers<8> : iters == 16 ? k_markers<16>
              : iters == 32 ? k_markers<32> : k_markers<64>;
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, st, *g, d_jpeg, begin, size, recs, epoch, h_maxlen_parts, d_summary, h_summary, d_hdr_ref, hdr_n, d_seg_pos,
                       d_seg_len, d_seg_index, max_segments + GJ_MAX_COMP);
    gj_debug_stage(tune->debug_sync != 0, st, "k_markers");
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// scratch for the records: one per workgroup
extern "C" size_t gj_hip_find_segments_scratch_words(uint64_t begin, uint64_t size, uint32_t max_segments)
{
    (void)max_segments;
    return 2 * GJ_SCAN_REC_WORDS * gj_hip_find_segments_max_chunks(begin, size) + 4;
}

// workgroups the marker scan cuts [begin, size) into at most (capacity of h_maxlen_parts)
extern "C" size_t gj_hip_find_segments_max_chunks(uint64_t begin, uint64_t size)
{
    const uint64_t bytes = size - begin + 16;
    const uint64_t wgs = (bytes + 256ull * 16 * 64 - 1) / (256ull * 16 * 64); // (at 1024 bytes per lane)
    return (size_t)(wgs > GJ_SCAN_WGS ? wgs : GJ_SCAN_WGS) + 1;
}
