// gj_dec_markers.hip -- MI355X (gfx950, wave64) JPEG decoder: device-side segment discovery (marker scan)
// (part of the decoder's device code, see gj_dec_internal.h for the map of the files)
#include "gj_dec_internal.h"

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

