// gj_dec_markers.hip -- MI355X (gfx950, wave64) JPEG decoder: device-side segment discovery (marker scan)
// (part of the decoder's device code, see gj_dec_internal.h for the map of the files)
#include "gj_dec_internal.h"
#include <stddef.h>

// ================================================================================================
// Device-side segment discovery. Inside entropy-coded data 0xFF is followed by 0x00 (stuffing),
// by 0xD0..0xD7 (restart marker = segment boundary) or by the marker that ends the scan. Two launches turn the bytes
// [begin, size) into the (offset, length, geometric index) table the entropy decoders consume, without the host touching the
// stream (the reference walks it with memchr and copies every segment, src/gpujpeg_reader.c:1039-1155):
//   k_marker_scan      per chunk of the stream: number of RSTn and the position of the last one; every other marker is appended (rare)
//                      to a small list; compares the stream's header with the cached one (speculative launches)
//   k_marker_segments  per chunk again: ranks its RSTn (sum of the counts in front, read by every workgroup: at most ~1000 chunks),
//                      sorts the other markers into scans, and writes the table entries its markers end; the summary the host
//                      validates (gj_scan_summary); clears the summary of the NEXT call (two summaries alternate, no memset launch)
// A lane owns `tb` consecutive bytes (8 .. 64, chosen by the host from the stream's size), a workgroup 256 x tb.
// ================================================================================================
#ifdef GJ_TRACE_PHASES
static __device__ unsigned long long* gj_trace_buf_m;
extern "C" GJ_HIP_API int gj_hip_trace_set_markers(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(gj_trace_buf_m), &p, sizeof p) == hipSuccess ? 0 : -1; }
#define GJ_TRACE_M(slot) do { if (threadIdx.x == 0 && gj_trace_buf_m) gj_trace_buf_m[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GJ_TRACE_M(slot) ((void)0)
#endif
#define GJ_SCAN_LIST 2048   // restart markers a chunk may hold (a segment of 8 bytes on average at the largest chunk: beyond that the host walks)

// bit i of the results: byte i of the lane's TB bytes at absolute offset b0 starts a restart marker / another marker; `num`: the low three
// bits of the restart markers' codes, 4 bits per marker in the order of their positions (at most 16 of them are recorded)
template <int TB>
__device__ __forceinline__ void gj_scan_bytes(const uint8_t* __restrict__ jpeg, const uint64_t size, const uint64_t b0, uint64_t& rst, uint64_t& other, uint64_t* nums = nullptr)
{
    rst = other = 0;
    uint64_t nm = 0;
    uint32_t nn = 0;
    if (nums) *nums = 0;
    if (b0 + 1 >= size) return;
    const uintptr_t a = reinterpret_cast<uintptr_t>(jpeg) + b0;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
    const uint32_t lead = (uint32_t)(a & 3);
    const uint32_t* end = reinterpret_cast<const uint32_t*>((reinterpret_cast<uintptr_t>(jpeg) + size + 3) & ~(uintptr_t)3);
    uint32_t win[TB / 4 + 2];
#pragma unroll
    for (int i = 0; i < TB / 4 + 2; i++) win[i] = src + i < end ? src[i] : 0u;
#pragma unroll
    for (int i = 0; i < TB / 4; i++) {
        const uint32_t w = __builtin_amdgcn_alignbyte(win[i + 1], win[i], lead);      // bytes 4i .. 4i + 3 of the lane's share
        const uint32_t wn = __builtin_amdgcn_alignbyte(win[i + 2], win[i + 1], lead); // (its first byte follows the last one of w)
        if ((~w - 0x01010101u) & w & 0x80808080u) { // 0xFF bytes are rare: one test for the four (a zero byte in ~w; a borrow can only add a false alarm)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t b = (w >> (8 * k)) & 0xFFu;
                const uint32_t nx = k < 3 ? (w >> (8 * k + 8)) & 0xFFu : wn & 0xFFu;
                if (b == 0xFFu && nx != 0u && nx != 0xFFu && b0 + (uint64_t)(4 * i + k) + 1 < size) {
                    if ((nx & 0xF8u) == 0xD0u) {
                        rst |= 1ull << (4 * i + k);
                        if (nums) { nm |= (uint64_t)(nx & 7u) << ((4u * nn) & 63u); nn++; }
                    } else {
                        other |= 1ull << (4 * i + k);
                    }
                }
            }
        }
    }
    if (nums) *nums = nm;
}

template <int TB>
__global__ __launch_bounds__(256) void k_marker_scan(const uint8_t* __restrict__ jpeg, const uint64_t begin, const uint64_t size,
                                                     uint16_t* __restrict__ chunk_cnt, uint32_t* __restrict__ chunk_last, gj_scan_summary* __restrict__ sum,
                                                     const uint8_t* __restrict__ hdr_ref, const uint32_t hdr_n)
{
    __shared__ uint32_t s_n, s_last, s_nother, s_oq[4], s_oslot[4], s_oafter[4];
    if (threadIdx.x == 0) { s_n = 0; s_last = 0; s_nother = 0; }
    if (threadIdx.x < 4) s_oafter[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t b0 = begin + ((uint64_t)blockIdx.x * 256u + threadIdx.x) * TB;
    uint64_t rst, other;
    gj_scan_bytes<TB>(jpeg, size, b0, rst, other);
    if (rst) {
        atomicAdd(&s_n, (uint32_t)__popcll(rst));
        atomicMax(&s_last, (uint32_t)(b0 + 63u - (uint32_t)__builtin_clzll(rst)));
    }
    while (other) { // scan boundary material: keep position, code and the 16 bytes that follow
        const uint64_t p = b0 + (uint32_t)__builtin_ctzll(other);
        other &= other - 1;
        const uint32_t slot = atomicAdd(&sum->other_count, 1u);
        if (slot < GJ_SCAN_MAX_OTHER) {
            sum->other_pos[slot] = (uint32_t)p;
            sum->other_code[slot] = jpeg[p + 1];
            for (int b = 0; b < 16; b++) sum->other_bytes[slot][b] = p + 2 + b < size ? jpeg[p + 2 + b] : 0;
            const uint32_t j = atomicAdd(&s_nother, 1u);
            if (j < 4) { s_oq[j] = (uint32_t)p; s_oslot[j] = slot; }
            else sum->other_after[slot] = 0xFFFFFFFFu; // (more than four in one chunk: not a stream the device table is used for)
        }
    }
    if (blockIdx.x == 0 && hdr_ref != nullptr) { // does the stream start with the header the host assumed? (speculative launch)
        int diff = 0;
        for (uint32_t i = threadIdx.x; i < hdr_n; i += 256) diff |= jpeg[i] != hdr_ref[i];
        diff = __syncthreads_or(diff);
        if (threadIdx.x == 0) sum->header_differs = diff ? 1u : 0u;
    } else {
        __syncthreads();
    }
    // restart markers of this chunk behind each of its other markers (an SOS: they are the first ones of the scan it starts)
    const uint32_t no = min(s_nother, 4u);
    if (no) { // (the same for the whole workgroup)
        for (uint32_t j = 0; j < no; j++) {
            const uint64_t q = s_oq[j];
            uint64_t m = rst;
            if (q >= b0) m = q - b0 >= 63 ? 0ull : m & ~((2ull << (q - b0)) - 1ull);
            if (m) atomicAdd(&s_oafter[j], (uint32_t)__popcll(m));
        }
        __syncthreads();
        if (threadIdx.x < no) sum->other_after[s_oslot[threadIdx.x]] = s_oafter[threadIdx.x];
    }
    if (threadIdx.x == 0) { // (a chunk is 16 KiB at most: its count fits 16 bits)
        chunk_cnt[blockIdx.x] = (uint16_t)s_n;
        chunk_last[blockIdx.x] = s_last;
    }
}

// Scan s is bounded by the "other" markers: it starts after an SOS header and ends at the next other marker. Scan 0 starts at
// `begin` (the host parsed its SOS). Table order: the segments of scan 0, of scan 1, ...; restart marker k of a scan ends its segment k
// and starts segment k + 1, so the lane that owns the marker writes the entry of segment k (and, for the last marker of a scan, the one
// of the scan's last segment).
template <int TB>
__global__ __launch_bounds__(256) void k_marker_segments(const gj_geom g, const uint8_t* __restrict__ jpeg, const uint64_t begin, const uint64_t size,
                                                         const uint32_t chunks, const uint16_t* __restrict__ chunk_cnt /* 16-byte aligned, readable up to a multiple of 8 */,
                                                         const uint32_t* __restrict__ chunk_last, uint32_t* __restrict__ chunk_maxlen /* host memory */,
                                                         gj_scan_summary* __restrict__ sum, gj_scan_summary* __restrict__ hsum /* host memory: what the host reads */,
                                                         gj_scan_summary* __restrict__ sum_next, uint32_t* __restrict__ seg_pos,
                                                         uint32_t* __restrict__ seg_len, uint32_t* __restrict__ seg_index, const uint32_t max_segments)
{
    __shared__ uint32_t s_start[GJ_MAX_COMP + 1], s_end[GJ_MAX_COMP + 1], s_first[GJ_MAX_COMP + 2];
    __shared__ uint32_t s_sos_chunk[GJ_MAX_COMP], s_sos_after[GJ_MAX_COMP]; // scan sc > 0: chunk of its SOS, restart markers of that chunk behind the SOS
    __shared__ int s_scans, s_status;
    __shared__ uint32_t s_geo_first[GJ_MAX_COMP], s_geo_limit[GJ_MAX_COMP]; // per scan: geometric index of its first segment, segments it may have
    __shared__ uint32_t s_acc[GJ_MAX_COMP + 2]; // restart markers in the chunks up to: this chunk (excl.), the chunk of every later scan's SOS (incl.); all
    __shared__ int s_prev_chunk;
    __shared__ uint32_t s_tmp[4], s_maxlen;
    __shared__ uint32_t s_mpos[GJ_SCAN_LIST]; // position | code & 7 << 29 would not fit 32-bit positions: offset inside the chunk | code & 7 << 16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr uint32_t chunk_bytes = 256u * TB;
    // everything this workgroup reads from memory is asked for at once: the summary, the counts of all chunks (16-bit: a lane takes eight of
    // them with one 16-byte load, the workgroup 2048; two such loads are kept in registers, more chunks are read again below), the last
    // marker of the chunk in front, and the bytes of its own chunk. Every workgroup reads all the counts: as 32-bit halves of 8-byte records
    // that was 29 MB through the L2 for an 8K frame and a third of this kernel's time.
    const uint32_t cvecs = (chunks + 7u) >> 3; // 16-byte pieces of the count array
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    const uint4 cnt_reg0 = (uint32_t)tid < cvecs ? reinterpret_cast<const uint4*>(chunk_cnt)[tid] : zero4;
    const uint4 cnt_reg1 = (uint32_t)tid + 256u < cvecs ? reinterpret_cast<const uint4*>(chunk_cnt)[tid + 256] : zero4;
    const uint32_t prev1_last = blockIdx.x > 0 ? chunk_last[blockIdx.x - 1] : 0u;
    GJ_TRACE_M(0);
    if (tid < GJ_MAX_COMP) { // scan i carries component i when the stream is not interleaved (src/gpujpeg_reader.c:1345)
        const int c = tid < g.comp_count ? tid : 0;
        s_geo_first[tid] = g.interleaved ? 0u : (uint32_t)g.comp[c].first_segment;
        s_geo_limit[tid] = g.interleaved ? (uint32_t)g.segment_count : (uint32_t)g.comp[c].segment_count;
    }
    // the few other markers: lane i of the first wave takes marker i
    // (all 16 slots are read whether they are in use or not: asking for the count first would be a second trip to memory)
    uint32_t o_pos = 0xFFFFFFFFu, o_code = 0, o_len = 0, o_after = 0;
    if (wave == 0 && lane < GJ_SCAN_MAX_OTHER) {
        o_pos = sum->other_pos[lane];
        o_code = sum->other_code[lane];
        o_len = ((uint32_t)sum->other_bytes[lane][0] << 8) | sum->other_bytes[lane][1];
        o_after = sum->other_after[lane];
    }
    const uint32_t n_other = min(sum->other_count, (uint32_t)GJ_SCAN_MAX_OTHER);
    if ((uint32_t)lane >= n_other) o_pos = 0xFFFFFFFFu;
    const uint64_t c0 = begin + (uint64_t)blockIdx.x * chunk_bytes;
    const uint64_t b0 = c0 + (uint64_t)tid * TB;
    uint64_t rst, other, nums;
    gj_scan_bytes<TB>(jpeg, size, b0, rst, other, &nums);
    if (wave == 0) {
        // their order by position (rank = markers in front), then a walk through them in that order with everything in scalar registers
        // (this was one lane going through arrays in memory: 10 us of dependent loads per workgroup)
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n_other; j++) {
            const uint32_t pj = (uint32_t)__builtin_amdgcn_readlane((int)o_pos, (int)j);
            rank += (pj < o_pos || (pj == o_pos && j < (uint32_t)lane)) ? 1u : 0u;
        }
        int scans = 0, status = 0;
        uint32_t start = (uint32_t)begin, sos_chunk = 0, sos_after = 0;
        for (uint32_t k = 0; k < n_other && scans < GJ_MAX_COMP; k++) {
            const unsigned long long who = __ballot((uint32_t)lane < n_other && rank == k);
            const int l = who ? __builtin_ctzll(who) : 0;
            const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)o_pos, l), m = (uint32_t)__builtin_amdgcn_readlane((int)o_code, l);
            const uint32_t mlen = (uint32_t)__builtin_amdgcn_readlane((int)o_len, l), after = (uint32_t)__builtin_amdgcn_readlane((int)o_after, l);
            if (p < start) continue; // lies inside a header we already skipped
            if (lane == 0) { s_start[scans] = start; s_end[scans] = p; s_sos_chunk[scans] = sos_chunk; s_sos_after[scans] = sos_after; }
            scans++;
            if (m == 0xDA) { // next scan
                start = p + 2 + mlen;
                sos_chunk = (uint32_t)((p - begin) / chunk_bytes);
                sos_after = after;
                if (after == 0xFFFFFFFFu) { status = 2; break; }
                continue;
            }
            if (m == 0xD9) { status = 1; break; } // EOI: done
            status = 2;                            // something else between scans: let the host walk it
            break;
        }
        if (status == 0) status = 3; // no EOI seen
        if (lane == 0) {
            s_scans = scans;
            s_status = status;
            s_prev_chunk = -1;
            s_maxlen = 0;
        }
    }
    if (tid < GJ_MAX_COMP + 2) s_acc[tid] = 0;
    GJ_TRACE_M(1);
    __syncthreads();
    GJ_TRACE_M(2);
    const int scans = s_scans;
    // ---- restart markers in front of this chunk, up to the chunk of every later scan's SOS, and all of them; the last chunk in front
    //      of this one that has a marker
    {
        static_assert(GJ_MAX_COMP == 4, "accumulators below");
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, aall = 0; // (scalars, not an array: the loop below is not unrolled)
        int prev = -1;
        const uint32_t t1 = scans > 1 ? s_sos_chunk[1] + 1u : 0u, t2 = scans > 2 ? s_sos_chunk[2] + 1u : 0u, t3 = scans > 3 ? s_sos_chunk[3] + 1u : 0u;
        for (uint32_t v = (uint32_t)tid, q = 0; v < cvecs; v += 256, q++) {
            uint4 w;
            if (q == 0) w = cnt_reg0;
            else if (q == 1) w = cnt_reg1;
            else w = reinterpret_cast<const uint4*>(chunk_cnt)[v];
#pragma unroll 1 // (unrolled, the forty comparisons' masks cost an eighth of the occupancy in scalar registers)
            for (uint32_t c = v * 8u; c < v * 8u + 8u; c++) {
                const uint32_t n = c < chunks ? w.x & 0xFFFFu : 0u; // (the array's last piece ends with stale counts)
                w.x = __builtin_amdgcn_alignbit(w.y, w.x, 16); // the next count moves down
                w.y = __builtin_amdgcn_alignbit(w.z, w.y, 16);
                w.z = __builtin_amdgcn_alignbit(w.w, w.z, 16);
                w.w >>= 16;
                a0 += c < blockIdx.x ? n : 0u;
                prev = (c < blockIdx.x && n) ? (int)c : prev;
                a1 += c < t1 ? n : 0u;
                a2 += c < t2 ? n : 0u;
                a3 += c < t3 ? n : 0u;
                aall += n;
            }
        }
        const uint32_t acc[GJ_MAX_COMP + 2] = {a0, a1, a2, a3, 0u, aall};
#pragma unroll
        for (int i = 0; i < GJ_MAX_COMP + 2; i++) {
            if (i == GJ_MAX_COMP) continue;
            uint32_t v = gj_wave_incl_scan(acc[i]);
            v = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
            if (lane == 0 && v) atomicAdd(&s_acc[i], v);
        }
        if (prev >= 0) atomicMax(&s_prev_chunk, prev);
    }
    __syncthreads();
    GJ_TRACE_M(3);
    // rank of the first restart marker of every scan: the markers up to its SOS's chunk minus those of that chunk behind the SOS (the SOS
    // header itself holds none); everything lies below the sentinel
    if (tid <= scans) s_first[tid] = tid == 0 ? 0u : (tid == scans ? s_acc[GJ_MAX_COMP + 1] : s_acc[tid] - s_sos_after[tid]);
    const uint32_t total = s_acc[GJ_MAX_COMP + 1];
    // (a stream with more restart markers than the geometry allows is damaged: entries beyond the table are not written and the host,
    //  seeing the count, rejects it)
    // ---- the markers of this chunk, in order
    uint32_t tot;
    const uint32_t mine = (uint32_t)__popcll(rst);
    uint32_t r = gj_wg256_incl_scan(mine, s_tmp, &tot) - mine;
    const bool too_many = tot > (uint32_t)GJ_SCAN_LIST || __syncthreads_or(mine > 16u); // (a lane notes the numbers of 16 markers)
    for (uint64_t m = rst; m && !too_many; m &= m - 1, nums >>= 4) {
        const uint32_t o = (uint32_t)tid * TB + (uint32_t)__builtin_ctzll(m);
        s_mpos[r++] = o | ((uint32_t)(nums & 7u) << 16);
    }
    __syncthreads();
    GJ_TRACE_M(4);
    uint32_t irregular = too_many ? 1u : 0u, maxlen = 0;
    const uint32_t rank0 = s_acc[0];
    // the last marker in front of this chunk (if any): normally in the chunk right in front
    const uint32_t prev_last = s_prev_chunk < 0 ? 0u : (s_prev_chunk == (int)blockIdx.x - 1 ? prev1_last : chunk_last[s_prev_chunk]);
    for (uint32_t i = (uint32_t)tid; i < tot && !too_many && scans > 0; i += 256) {
        const uint32_t p = (uint32_t)c0 + (s_mpos[i] & 0xFFFFu), num = s_mpos[i] >> 16, rk = rank0 + i;
        int sc = -1;
#pragma unroll
        for (int q = 0; q < GJ_MAX_COMP; q++)
            if (q < scans && p >= s_start[q] && p < s_end[q]) sc = q;
        if (sc < 0) { irregular = 1u; continue; } // (a restart marker outside every scan)
        const uint32_t k = rk - s_first[sc];                       // the marker's number inside its scan = the segment it ends
        const uint32_t c_s = s_first[sc + 1] - s_first[sc];        // RSTn inside this scan
        if (rk < s_first[sc] || k >= c_s) { irregular = 1u; continue; }
        const uint32_t before = i > 0 ? (uint32_t)c0 + (s_mpos[i - 1] & 0xFFFFu) : prev_last; // the marker in front of this one
        const uint32_t from = (k == 0) ? s_start[sc] : before + 2;
        const uint32_t first = s_geo_first[sc], limit = s_geo_limit[sc];
        const uint32_t e = s_first[sc] + (uint32_t)sc + k;
        // the marker that ends segment k must be RST(k mod 8), and the last segment of a scan must not be empty: anything else is a
        // stream the reference reader treats specially, which the host walk reproduces
        if (num != (k & 7u)) irregular = 1u;
        if (e < max_segments) {
            seg_pos[e] = from;
            seg_len[e] = p > from ? p - from : 0;
            seg_index[e] = k < limit ? first + k : 0xFFFFFFFFu;
            if (p > from) maxlen = max(maxlen, p - from);
        }
        if (k + 1 == c_s) { // the last marker of the scan: the segment behind it ends with the scan
            const uint32_t from2 = p + 2, to2 = s_end[sc];
            if (to2 <= from2) irregular = 1u;
            if (e + 1 < max_segments) {
                seg_pos[e + 1] = from2;
                seg_len[e + 1] = to2 > from2 ? to2 - from2 : 0;
                seg_index[e + 1] = k + 1 < limit ? first + k + 1 : 0xFFFFFFFFu;
                if (to2 > from2) maxlen = max(maxlen, to2 - from2);
            }
        }
    }
    if (blockIdx.x == 0 && tid < scans) { // scans without a restart marker: one segment
        const int sc = tid;
        if (s_first[sc + 1] == s_first[sc]) {
            const uint32_t e = s_first[sc] + (uint32_t)sc;
            const uint32_t first = s_geo_first[sc], limit = s_geo_limit[sc];
            if (e < max_segments) {
                const uint32_t len = s_end[sc] > s_start[sc] ? s_end[sc] - s_start[sc] : 0;
                seg_pos[e] = s_start[sc];
                seg_len[e] = len;
                seg_index[e] = limit > 0 ? first : 0xFFFFFFFFu;
                maxlen = max(maxlen, len);
            }
        }
    }
    if (maxlen) atomicMax(&s_maxlen, maxlen);
    irregular = (uint32_t)__syncthreads_or((int)irregular);
    GJ_TRACE_M(5);
    if (tid == 0) {
        chunk_maxlen[blockIdx.x] = s_maxlen; // (the host takes the maximum: a thousand workgroups raising one word one after the other took 10 us)
        if (irregular) hsum->rst_irregular = 1u; // (the host cleared it before the launch)
    }
    if (blockIdx.x == 0) {
        // what the host validates goes straight to its (pinned, device-visible) copy -- no copy launch behind the kernels: the other markers
        // as the scan kernel left them, then the scan structure. The words other workgroups and later kernels raise (rst_irregular,
        // seq_overflow) and the one the host fills in (max_seg_len) are left alone.
        const uint32_t w_skip0 = (uint32_t)(offsetof(gj_scan_summary, max_seg_len) / 4), w_skip1 = (uint32_t)(offsetof(gj_scan_summary, seq_overflow) / 4),
                       w_skip2 = (uint32_t)(offsetof(gj_scan_summary, rst_irregular) / 4);
        for (uint32_t i = (uint32_t)tid; i < sizeof(gj_scan_summary) / 4; i += 256)
            if (i != w_skip0 && i != w_skip1 && i != w_skip2) reinterpret_cast<uint32_t*>(hsum)[i] = reinterpret_cast<const uint32_t*>(sum)[i];
        __syncthreads();
        if (tid == 0) {
            sum->segment_count = scans ? total + (uint32_t)scans : 0u; // (speculative launches: the entropy decoder reads the count from here)
            hsum->rst_count = total;
            hsum->scan_count = (uint32_t)scans;
            hsum->status = (uint32_t)s_status;
            hsum->segment_count = scans ? total + (uint32_t)scans : 0u;
            for (int sc = 0; sc < scans; sc++) { hsum->scan_start[sc] = s_start[sc]; hsum->scan_end[sc] = s_end[sc]; }
        }
    }
    GJ_TRACE_M(6);
    // the summary of the next call (the two alternate): its counters start at zero
    if (blockIdx.x == 0 && sum_next != nullptr)
        for (uint32_t i = (uint32_t)tid; i < sizeof(gj_scan_summary) / 4; i += 256) reinterpret_cast<uint32_t*>(sum_next)[i] = 0;
}

static uint32_t gj_scan_lane_bytes(uint64_t begin, uint64_t size)
{
    // at most ~2048 chunks (every workgroup of k_marker_segments reads all the chunk counts: 8 registers per lane), 8 .. 64 bytes per lane
    // (measured at 8K, 7.4 MB: 8 B per lane 9.7 + 29.0 us, 16 B 10.0 + 17.5, 32 B 12.5 + 17.3, 64 B 17.1 + 17.9)
    const uint64_t bytes = size - begin;
    const uint64_t want = (bytes + 256ull * 2048 - 1) / (256ull * 2048);
    return want <= 8 ? 8u : want <= 16 ? 16u : want <= 32 ? 32u : 64u;
}

extern "C" int gj_hip_find_segments(const gj_geom* g, const uint8_t* d_jpeg, uint64_t begin, uint64_t size, uint32_t* d_seg_pos,
                                    uint32_t* d_seg_len, uint32_t* d_seg_index, uint32_t max_segments, uint32_t* d_scratch,
                                    gj_scan_summary* d_summary, gj_scan_summary* d_summary_next, const uint8_t* d_hdr_ref, uint32_t hdr_n,
                                    gj_scan_summary* h_summary, uint32_t* h_maxlen_parts, uint32_t maxlen_capacity, uint32_t* maxlen_part_count,
                                    gj_stream_t stream, const gj_tuning* tune)
{
    const int debug_sync = tune->debug_sync;
    hipStream_t st = (hipStream_t)stream;
    if (size <= begin || size > 0xFFFFFFF0ull) return -1;
    const uint32_t tb = (tune->scan_tb == 8 || tune->scan_tb == 16 || tune->scan_tb == 32 || tune->scan_tb == 64) ? (uint32_t)tune->scan_tb : gj_scan_lane_bytes(begin, size);
    const uint32_t chunks = (uint32_t)((size - begin + 256ull * tb - 1) / (256ull * tb));
    uint32_t* d_last = d_scratch;                                                            // [chunks] position of the chunk's last restart marker
    uint16_t* d_cnt = reinterpret_cast<uint16_t*>(d_scratch + (((size_t)chunks + 3) & ~(size_t)3)); // [chunks rounded up to 8] restart markers in the chunk
    if (chunks > maxlen_capacity) return -1;
    *maxlen_part_count = chunks;
    auto scan = tb == 8 ? k_marker_scan<8> : tb == 16 ? k_marker_scan<16> : tb == 32 ? k_marker_scan<32> : k_marker_scan<64>;
    auto segs = tb == 8 ? k_marker_segments<8> : tb == 16 ? k_marker_segments<16> : tb == 32 ? k_marker_segments<32> : k_marker_segments<64>;
    hipLaunchKernelGGL(scan, dim3(chunks), dim3(256), 0, st, d_jpeg, begin, size, d_cnt, d_last, d_summary, d_hdr_ref, hdr_n);
    gj_debug_stage(debug_sync != 0, st, "k_marker_scan");
    hipLaunchKernelGGL(segs, dim3(chunks), dim3(256), 0, st, *g, d_jpeg, begin, size, chunks, d_cnt, d_last, h_maxlen_parts, d_summary, h_summary, d_summary_next,
                       d_seg_pos, d_seg_len, d_seg_index, max_segments + GJ_MAX_COMP);
    gj_debug_stage(debug_sync != 0, st, "k_marker_segments");
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" size_t gj_hip_find_segments_scratch_words(uint64_t begin, uint64_t size, uint32_t max_segments)
{
    (void)max_segments;
    return 2 * (size_t)((size - begin + 2047) / 2048) + 16; /* (lanes take at least 8 bytes) */
}

extern "C" size_t gj_hip_find_segments_max_chunks(uint64_t begin, uint64_t size) { return (size_t)((size - begin + 2047) / 2048) + 1; }
