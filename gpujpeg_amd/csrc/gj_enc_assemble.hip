// gj_enc_assemble.hip -- MI355X (gfx950, wave64) JPEG encoder: the streams the coders leave -> the finished file, on the device
// (part of the encoder's device code, see gj_enc_internal.h for the map of the files)
//   k_gather              tile streams of the fused encoders -> the file: stuffing in flight, RSTn, scan headers, EOI, result words
//   k_scan_segments       prefix sum of the stuffed segment sizes -> final byte offsets   } behind k_huffman (coefficient planes),
//   k_assemble            byte stuffing + RSTn + scan headers + EOI, in stream order      } k_scan_segments also for the APP13 index
//   k_segment_info        APP13 segment index (src/gpujpeg_writer.c:522-623)
#include "gj_enc_internal.h"

// ================================================================================================
// k_gather: the tile streams -> the file. Replaces k_scan_segments + k_assemble behind the k_encode_* kernels (the reference:
// serialisation + compaction kernels and the host's stitching, src/gpujpeg_huffman_gpu_encoder.cu:417-613, src/gpujpeg_encoder.c:567-629).
//
// An encoder workgroup leaves the UNSTUFFED stream of its tile in the tile's area of d_temp (segments on dword boundaries), the
// segments' byte and 0xFF counts, the size the tile's stream will have in the file, and adds that size to the total of its group of
// 32 tiles (one atomic, nobody waits for it). Here ONE WAVE takes one tile stream, ~6000 waves at once for an 8K frame, and everything
// it needs is asked for in one trip: the group totals and the sizes of its group's tiles in front of it (its place in the file), its
// segments' counts, and the first 2 KB of its stream (where those are follows from the launch's geometry, not from loaded values).
// Then the stream is stuffed in flight: a lane takes a dword, finds its segment (the segment starts are marked in LDS: a wave prefix
// sum of the marks), counts its 0xFF bytes; a second prefix sum places it; a dword without 0xFF leaves as one unaligned 4-byte store,
// restart markers follow the last dword of a segment. Round 3 needed a launch for the offsets (5 us), a wave per four
// segments with two dependent trips and byte stores (16 us), and 19 MB of traffic for the same.
// ================================================================================================
#define GJ_GATHER_PRELOAD 8 // rounds of 64 dwords whose loads are issued before anything is known about the tile stream
#define GJ_GATHER_MARK_DW 2048 // dwords of a tile stream whose segment starts are marked in LDS at a time
__global__ __launch_bounds__(256) void k_gather(const GjTail T0)
{
    GjTail T = T0;
    { // frame blockIdx.z of a batch: its own tile list, group totals, tile streams, segment counts, stream buffer and result words
        const size_t fz = blockIdx.z;
        T.piece += fz * T0.f_tail; T.group += fz * T0.f_tail; T.group_other += fz * T0.f_tail;
        T.temp += fz * T0.f_temp; T.seg_bytes += fz * T0.f_seg; T.seg_ff += fz * T0.f_seg;
        T.jpeg += fz * T0.f_jpeg; T.d_result += fz * 2;
        if (T.h_result) T.h_result += fz * 2;
    }
    __shared__ uint32_t s_tmp[4];
    __shared__ __attribute__((aligned(16))) uint8_t s_mark[4][GJ_GATHER_MARK_DW];
    const int i = threadIdx.x, lane = i & 63, wave = i >> 6;
    const uint32_t P = T.npieces, NG = T.ngroups;
    const uint32_t p0 = blockIdx.x * 4u, p = p0 + (uint32_t)wave;
    const bool have = p < P;
    if (blockIdx.x == 0) // the next call's group totals
        for (uint32_t g = i; g < NG; g += 256) T.group_other[g] = 0;
    // ---- this wave's tile stream: where its segments and its bytes are (no loaded value needed)
    const uint32_t scan = gj_tail_scan_of(T, have ? p : 0u);
    const uint32_t t = (have ? p : 0u) - gj_pick4(T.scan_first, scan), seg0 = t * T.spt, scan_segs = gj_pick4(T.segs, scan);
    const uint32_t nseg = have ? min(T.spt, scan_segs - seg0) : 0u;
    const uint32_t s0 = gj_pick4(T.seg_first, scan) + seg0;
    const uint64_t src_block = (uint64_t)gj_pick4(T.block_first, scan) + (uint64_t)seg0 * T.seg_blocks;
    const uint32_t* const src = reinterpret_cast<const uint32_t*>(T.temp + src_block * GJ_TEMP_BYTES_PER_BLOCK);
    // dwords of the tile's area: nothing is read behind them, nor behind the buffer (the area of a scan's last tile ends with its last,
    // shorter segment; found by the execution model under AddressSanitizer in round 4: the preload below read up to 35 blocks further)
    const uint64_t room = T.temp_blocks > src_block ? T.temp_blocks - src_block : 0u;
    const uint32_t src_dw = (uint32_t)min((uint64_t)nseg * T.seg_blocks, room) * (GJ_TEMP_BYTES_PER_BLOCK / 4u);
    // ---- one trip: group totals, the sizes of the tiles of the group in front of the workgroup's first one and of the workgroup's
    // own, the segments' counts, the first rounds of the stream
    const uint32_t ga = p0 >> 5;
    uint32_t before = 0, all = 0;
    for (uint32_t g0 = 0; g0 < NG; g0 += 1024) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t g = g0 + (uint32_t)u * 256u + (uint32_t)i;
            v[u] = g < NG ? T.group[g] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t g = g0 + (uint32_t)u * 256u + (uint32_t)i;
            all += v[u];
            before += g < ga ? v[u] : 0u;
        }
    }
    if ((uint32_t)i < (p0 & 31u)) before += T.piece[(ga << 5) + i];
    uint32_t in_front = 0; // of this wave's tile inside the workgroup
    if (wave > 0 && have) in_front += T.piece[p0];
    if (wave > 1 && have) in_front += T.piece[p0 + 1];
    if (wave > 2 && have) in_front += T.piece[p0 + 2];
    uint32_t nb = 0, ff = 0;
    if ((uint32_t)lane < nseg) {
        nb = T.seg_bytes[s0 + lane];
        ff = T.seg_ff[s0 + lane];
    }
    uint32_t pre[GJ_GATHER_PRELOAD];
#pragma unroll
    for (int m = 0; m < GJ_GATHER_PRELOAD; m++) {
        const uint32_t d = (uint32_t)m * 64u + (uint32_t)lane;
        pre[m] = d < src_dw ? src[d] : 0u;
    }
    uint32_t done_bytes, all_bytes;
    gj_wg256_incl_scan(before, s_tmp, &done_bytes);
    gj_wg256_incl_scan(all, s_tmp, &all_bytes);
    const uint64_t total = (uint64_t)T.main_hdr + gj_pick4(T.hdr_end, gj_tail_scan_of(T, P - 1)) + all_bytes + 2u;
    const bool overflow = total > T.capacity;
    if (p0 + 4 >= P && i == 0) { // (the workgroup of the last tile stream)
        T.d_result[0] = (uint32_t)total;
        T.d_result[1] = overflow ? 1u : 0u;
        if (T.h_result) { // the host's (pinned, device-visible) copy: no copy launch behind the kernel
            T.h_result[0] = (uint32_t)total;
            T.h_result[1] = overflow ? 1u : 0u;
        }
    }
    if (!have || overflow) return;
    const uint32_t F = T.main_hdr + gj_pick4(T.hdr_end, scan) + done_bytes + in_front; // the tile stream's first byte in the file
    uint8_t* const out = T.jpeg;
    if (t == 0) { // first tile of a scan: its header (APP13 placeholders + SOS) sits right in front
        const uint32_t h1 = gj_pick4(T.hdr_end, scan), h0 = scan == 0 ? 0u : gj_pick4(T.hdr_end, scan - 1);
        for (uint32_t b = lane; b < h1 - h0; b += 64) out[F - (h1 - h0) + b] = T.scan_hdr[h0 + b];
    }
    // ---- lane sl keeps segment sl: its dwords, its first dword in the tile stream, its first byte in the file, the 0xFF bytes in front
    // of it. Everything a dword needs to know about its segment is two words: where its bytes go if none of the tile's dwords held a
    // 0xFF (minus 4 x its index), and the segment's last dword with the bytes that count in it.
    const uint32_t last = scan_segs - seg0 - 1u; // (local index of the scan's last segment: no restart marker behind it)
    const uint32_t ndw = (nb + 3u) >> 2, olen = nb + ff + ((uint32_t)lane < nseg && (uint32_t)lane != last ? 2u : 0u);
    const uint32_t dwi = gj_wave_incl_scan(ndw), dwb = dwi - ndw;
    const uint32_t oi = gj_wave_incl_scan(olen), ob = F + oi - olen;
    const uint32_t ffi = gj_wave_incl_scan(ff);
    const uint32_t total_dw = (uint32_t)__builtin_amdgcn_readlane((int)dwi, 63);
    const uint32_t end = F + (uint32_t)__builtin_amdgcn_readlane((int)oi, 63);
    if (p == P - 1 && lane == 0) { // EOI
        out[end] = 0xFF;
        out[end + 1] = 0xD9;
    }
    const uint32_t seg_a = ob - 4u * dwb - (ffi - ff);                               // byte q of dword d: seg_a + 4 d + 0xFF bytes in front of d
    const uint32_t seg_e = (dwi - 1u) | ((nb - 4u * (ndw - 1u)) << 28);              // last dword | its bytes (1 .. 4) << 28
    // which dwords begin a segment: a byte per dword in LDS (tile streams of more dwords take several passes)
    uint8_t* const mark = s_mark[wave];
    uint32_t ffrun = 0, segs_before = 0; // 0xFF bytes / segment starts of the dwords of earlier rounds
    for (uint32_t c0 = 0; c0 < total_dw; c0 += GJ_GATHER_MARK_DW) {
        const uint32_t c1 = min(total_dw, c0 + (uint32_t)GJ_GATHER_MARK_DW);
        gj_wave_sync();
#pragma unroll
        for (int z = 0; z < GJ_GATHER_MARK_DW / 256; z++) reinterpret_cast<uint32_t*>(mark)[z * 64 + lane] = 0;
        gj_wave_sync();
        if ((uint32_t)lane < nseg && dwb >= c0 && dwb < c1) mark[dwb - c0] = 1;
        gj_wave_sync();
        for (uint32_t m = c0 >> 6; m * 64u < c1; m++) {
            const uint32_t d = m * 64u + (uint32_t)lane;
            const bool live = d < total_dw;
            uint32_t v = 0;
            if (m < GJ_GATHER_PRELOAD) {
#pragma unroll
                for (int q = 0; q < GJ_GATHER_PRELOAD; q++)
                    if (m == (uint32_t)q) v = pre[q];
            } else if (live) {
                v = src[d];
            }
            const uint32_t seg = gj_wave_incl_scan(live ? mark[d - c0] : 0u) + segs_before - 1u; // the segment of dword d
            const int sidx = (int)((live ? seg : 0u) << 2);
            const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(sidx, (int)seg_a), e = (uint32_t)__builtin_amdgcn_ds_bpermute(sidx, (int)seg_e);
            const bool ends = live && d == (e & 0x0FFFFFFFu);
            const int vb = live ? (ends ? (int)(e >> 28) : 4) : 0;
            const uint32_t ffm = live ? ((v & 0x7F7F7F7Fu) + 0x01010101u) & v & 0x80808080u : 0u; // (the bytes behind a segment's end are zero)
            const uint32_t ffc = (uint32_t)__builtin_popcount(ffm);
            const uint32_t fi = gj_wave_incl_scan(ffc) + ffrun; // 0xFF bytes up to and including this dword
            if (live) {
                uint32_t q = a + 4u * d + (fi - ffc);
                if (ffc == 0 && vb == 4) {
                    *reinterpret_cast<gj_u32_unaligned*>(out + q) = v;
                    q += 4;
                } else {
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        if (b < vb) {
                            const uint32_t byte = (v >> (8 * b)) & 0xFFu;
                            out[q++] = (uint8_t)byte;
                            if (byte == 0xFFu) out[q++] = 0;
                        }
                }
                if (ends && seg != last) { // RSTn (src/gpujpeg_huffman_gpu_encoder.cu:497-502)
                    out[q] = 0xFF;
                    out[q + 1] = (uint8_t)(0xD0 + ((seg0 + seg) & 7u));
                }
            }
            ffrun = (uint32_t)__builtin_amdgcn_readlane((int)fi, 63);
            segs_before = (uint32_t)__builtin_amdgcn_readlane((int)seg, 63) + 1u;
        }
    }
}

// the workgroup's Huffman tables in the layout of GjCoderLds::lut: the host has them ready behind its (code << 8 | size) tables
// (gj_enc_job::d_huff_lut + GJ_CODER_LUT_OFFSET, gj_huffman_coder_lut; worked out in the kernel they cost every wave ~60 vector instructions, round 5)
// ================================================================================================
// Final offsets: exclusive prefix sum over stuffed segment sizes (+2 for RSTn except at the end of a scan) and over
// the scan headers that precede each scan. One launch of ceil(S/1024) workgroups (k_scan_segments below): per-workgroup totals
// published with an epoch tag, every workgroup adds the totals of its predecessors (at most a few hundred values) to its local scan.
// ================================================================================================
__device__ __forceinline__ uint32_t gj_segment_out_size(const gj_enc_job& J, int s, uint32_t* hdr)
{
    const GjSeg sg = gj_segment(J.g, s);
    *hdr = 0;
    if (sg.first_in_scan) {
        const int scan = J.g.interleaved ? 0 : sg.comp;
        *hdr = J.scan_hdr_offset[scan + 1] - J.scan_hdr_offset[scan];
    }
    return J.d_seg_bytes[s] + J.d_seg_ff[s] + (sg.last_in_scan ? 0u : 2u);
}

// inclusive scan over a 1024-thread workgroup; s_w needs 16 words
__device__ __forceinline__ uint32_t gj_wg1024_incl_scan(uint32_t v, uint32_t* s_w, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = gj_wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t off = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const uint32_t x = s_w[w];
        if (w < wave) off += x;
        all += x;
    }
    if (total) *total = all;
    return inc + off;
}

// One launch: every workgroup scans its 1024 segments, publishes its total tagged with the call's epoch, then adds up the
// totals of its predecessors as soon as they appear (all workgroups of a frame are resident at once and are dispatched
// in index order, so a predecessor never waits for a successor). The epoch tag makes clearing the slots unnecessary.
__global__ __launch_bounds__(1024) void k_scan_segments(const gj_enc_job J, unsigned long long* __restrict__ partial, const uint32_t epoch)
{
    __shared__ uint32_t s_w[16];
    const int S = J.g.segment_count;
    const int s = blockIdx.x * 1024 + threadIdx.x;
    uint32_t hdr = 0, v = 0;
    if (s < S) v = gj_segment_out_size(J, s, &hdr);
    uint32_t total;
    const uint32_t inc = gj_wg1024_incl_scan(v + hdr, s_w, &total);
    if (threadIdx.x == 0)
        __hip_atomic_store(&partial[blockIdx.x], ((unsigned long long)epoch << 32) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t pre = 0;
    for (unsigned t = threadIdx.x; t < blockIdx.x; t += 1024) {
        unsigned long long p;
        do {
            p = __hip_atomic_load(&partial[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        } while ((uint32_t)(p >> 32) != epoch);
        pre += (uint32_t)p;
    }
    uint32_t base;
    gj_wg1024_incl_scan(pre, s_w, &base);
    base += J.main_hdr_size;
    if (s < S) J.d_seg_out[s] = base + inc - v; // segment data start (its scan header sits right before)
    if (s == S - 1) {
        const uint32_t end = base + inc;
        const uint32_t size = end + 2; // EOI
        J.d_seg_out[S] = end;
        J.d_result[0] = size;
        J.d_result[1] = (uint64_t)size > J.jpeg_capacity ? 1u : 0u;
        if (J.h_result) { // the host's (pinned, device-visible) copy: no copy launch behind the kernels
            J.h_result[0] = size;
            J.h_result[1] = (uint64_t)size > J.jpeg_capacity ? 1u : 0u;
        }
    }
}

// ================================================================================================
// Stream assembly: one WAVE per segment. Reads the unstuffed bytes, inserts 0x00 after every 0xFF
// (ballot-free: per-lane counts + wave prefix sum), appends RSTn, and the first / last segment of a scan
// also writes the scan header / EOI. Replaces the reference's serialisation + compaction kernels and the
// host-side stitching loop (src/gpujpeg_huffman_gpu_encoder.cu:417-613, src/gpujpeg_encoder.c:567-629).
// ================================================================================================
                      // instead of four; 43 200 waves of one short segment each spent their time waiting)
__global__ __launch_bounds__(256) void k_assemble(const gj_enc_job J)
{
    const gj_geom& g = J.g;
    const int s0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * GJ_ASM_SEGS;
    const int lane = threadIdx.x & 63;
    if (s0 >= g.segment_count || J.d_result[1]) return;
    uint8_t* out = J.d_jpeg;
    uint32_t raw4[GJ_ASM_SEGS], o4[GJ_ASM_SEGS], w4[GJ_ASM_SEGS];
    const uint8_t* src4[GJ_ASM_SEGS];
#pragma unroll
    for (int q = 0; q < GJ_ASM_SEGS; q++) {
        const int s = min(s0 + q, g.segment_count - 1);
        raw4[q] = s0 + q < g.segment_count ? J.d_seg_bytes[s] : 0u;
        o4[q] = J.d_seg_out[s];
        src4[q] = J.d_temp + gj_segment(g, s).first_block * GJ_TEMP_BYTES_PER_BLOCK;
    }
#pragma unroll
    for (int q = 0; q < GJ_ASM_SEGS; q++) w4[q] = (uint32_t)lane * 4u < raw4[q] ? *reinterpret_cast<const uint32_t*>(src4[q] + lane * 4) : 0u;
#pragma unroll
    for (int q = 0; q < GJ_ASM_SEGS; q++) {
        const int s = s0 + q;
        if (s >= g.segment_count) break;
        const GjSeg sg = gj_segment(g, s);
        const uint32_t raw = raw4[q];
        const uint8_t* src = src4[q];
        uint32_t o = o4[q];
        if (sg.first_in_scan) { // scan header (APP13 placeholders + SOS) right before the first segment
            const int scan = g.interleaved ? 0 : sg.comp;
            const uint32_t h0 = J.scan_hdr_offset[scan], hn = J.scan_hdr_offset[scan + 1] - h0;
            for (uint32_t b = lane; b < hn; b += 64) out[o - hn + b] = J.d_scan_hdr[h0 + b];
        }
        for (uint32_t c0 = 0; c0 < raw; c0 += 256) {
            const uint32_t idx = c0 + lane * 4;
            uint32_t w = w4[q];
            int vb = 0;
            if (idx < raw) {
                if (c0) w = *reinterpret_cast<const uint32_t*>(src + idx);
                vb = (int)min(4u, raw - idx);
            }
            int cnt = vb;
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (b < vb && ((w >> (8 * b)) & 0xFFu) == 0xFFu) cnt++;
            const uint32_t inc = gj_wave_incl_scan((uint32_t)cnt);
            uint32_t p = o + inc - (uint32_t)cnt;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if (b < vb) {
                    const uint32_t byte = (w >> (8 * b)) & 0xFFu;
                    out[p++] = (uint8_t)byte;
                    if (byte == 0xFFu) out[p++] = 0;
                }
            }
            o += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        if (lane == 0) {
            if (!sg.last_in_scan) {
                out[o] = 0xFF;
                out[o + 1] = (uint8_t)(0xD0 + (sg.index_in_scan & 7));
            }
            if (s == g.segment_count - 1) {
                out[o] = 0xFF;
                out[o + 1] = 0xD9;
            }
        }
    }
}

// APP13 segment index (src/gpujpeg_writer.c:522-547): big-endian u32 start of every segment relative to
// the first one of its scan, plus the end position (after the dropped final RSTn).
#define GJ_MAX_HEADER_SIZE (65536 - 100)
__global__ __launch_bounds__(256) void k_segment_info(const gj_enc_job J)
{
    const gj_geom& g = J.g;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= g.segment_count || J.d_result[1]) return;
    const GjSeg sg = gj_segment(g, s);
    const int scan = g.interleaved ? 0 : sg.comp;
    const int first = g.interleaved ? 0 : g.comp[scan].first_segment;
    const int segs = g.interleaved ? g.segment_count : g.comp[scan].segment_count;
    const uint32_t data0 = J.d_seg_out[first];
    const uint32_t hdr_begin = data0 - (J.scan_hdr_offset[scan + 1] - J.scan_hdr_offset[scan]);
    for (int e = sg.index_in_scan; e <= (sg.last_in_scan ? segs : sg.index_in_scan); e++) {
        uint32_t pos;
        if (e < segs) pos = J.d_seg_out[first + e] - data0;
        else pos = J.d_seg_out[first + segs - 1] + J.d_seg_bytes[first + segs - 1] + J.d_seg_ff[first + segs - 1] - data0;
        // payload chunks of GJ_MAX_HEADER_SIZE bytes, each preceded by marker(2)+length(2)+scan(1)
        const uint32_t byte = (uint32_t)e * 4u;
        const uint32_t chunk = byte / GJ_MAX_HEADER_SIZE, within = byte % GJ_MAX_HEADER_SIZE;
        uint8_t* p = J.d_jpeg + hdr_begin + J.scan_info_payload[scan] + chunk * (GJ_MAX_HEADER_SIZE + 5u) + within;
        p[0] = (uint8_t)(pos >> 24); p[1] = (uint8_t)(pos >> 16); p[2] = (uint8_t)(pos >> 8); p[3] = (uint8_t)pos;
    }
}

