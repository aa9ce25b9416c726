// gj_encode.hip -- MI355X (gfx950, wave64) JPEG encoder: gj_hip_encode picks the kernels of a frame and launches them.
// The kernels live in gj_enc_*.hip (map: gj_enc_internal.h).
//
// Pipeline (all in one stream, the bitstream is assembled on the device):
//   k_encode_rgb444 / k_encode_uyvy422 / k_encode_blocks -> k_gather            pixels -> tile streams -> file (the default)
//   k_fused_* or k_preprocess + k_dct -> k_huffman -> k_scan_segments + k_assemble   through coefficient planes
#include "gj_enc_internal.h"

// k_gather's arguments (and the encoder kernels': they use the tile list and the group totals) for a launch that leaves
// `pieces` tile streams of `spt` segments; scan s begins with stream scan_first[s]
static GjTail gj_make_tail(const gj_enc_job* job, const unsigned pieces, const unsigned (&scan_first)[GJ_MAX_COMP], const unsigned spt)
{
    const gj_geom& g = job->g;
    GjTail T;
    for (int s = 0; s < GJ_MAX_COMP; s++) {
        T.scan_first[s] = scan_first[s];
        T.hdr_end[s] = job->scan_hdr_offset[s + 1];
        const bool own_scan = !g.interleaved && s < g.comp_count; // (one scan per component, or one for all)
        T.seg_first[s] = own_scan ? (uint32_t)g.comp[s].first_segment : 0u;
        T.segs[s] = own_scan ? (uint32_t)g.comp[s].segment_count : (uint32_t)g.segment_count;
        T.block_first[s] = own_scan ? (uint32_t)(g.comp[s].data_offset / 64) : 0u;
    }
    T.spt = spt;
    T.seg_blocks = (uint32_t)g.seg_blocks;
    const unsigned ngcap = GJ_TAIL_GROUPS_CAP(g.segment_count); // (one tile stream per segment at most)
    T.group = job->d_tail + (job->tail_set & 1) * ngcap;
    T.group_other = job->d_tail + ((job->tail_set + 1) & 1) * ngcap;
    T.ngroups = (pieces + 31) / 32;
    T.piece = job->d_tail + 2 * ngcap;
    T.npieces = pieces;
    T.temp = job->d_temp;
    T.temp_blocks = (uint64_t)g.block_count;
    T.seg_bytes = job->d_seg_bytes;
    T.seg_ff = job->d_seg_ff;
    T.jpeg = job->d_jpeg;
    T.capacity = job->jpeg_capacity;
    T.scan_hdr = job->d_scan_hdr;
    T.main_hdr = job->main_hdr_size;
    T.d_result = job->d_result;
    T.h_result = job->h_result;
    const bool batch = job->batch.count > 1;
    T.f_raw = batch ? job->batch.raw : 0;
    T.f_temp = batch ? job->batch.temp : 0;
    T.f_jpeg = batch ? job->batch.jpeg : 0;
    T.f_seg = batch ? job->batch.seg : 0;
    T.f_tail = batch ? job->batch.tail : 0;
    return T;
}


// k_encode_blocks for this configuration: 1 = planar input in component layout, 0 = packed 4:4:4 with a transform from RGB (or none) and
// any chroma sampling, -1 = neither (generic kernels)
static int gj_blocks_kernel_mode(const gj_geom& g)
{
    if (g.no_transform) return 1;
    if (g.pixel_format != GJ_PF_444_P012 || g.comp_count != 3) return -1;
    const int from = g.color_space, to = g.color_space_internal;
    const bool none = from == to || from == GJ_CS_NONE || to == GJ_CS_NONE;
    if (!none && !(from == GJ_CS_RGB && (to == GJ_CS_BT601 || to == GJ_CS_BT601_256 || to == GJ_CS_BT709))) return -1;
    return 0;
}

// packed 4:2:2 without colour transform in the layout k_fused_uyvy422 / k_encode_uyvy422 take
static bool gj_is_uyvy_layout(const gj_enc_job* job)
{
    const gj_geom& g = job->g;
    return job->use_fused && g.pixel_format == GJ_PF_422_P1020 && g.comp_count == 3 && g.no_transform == 0 &&
           (g.color_space == g.color_space_internal || g.color_space == GJ_CS_NONE || g.color_space_internal == GJ_CS_NONE) &&
           g.comp[0].samp_h == 2 && g.comp[0].samp_v == 1 && g.comp[1].samp_h == 1 && g.comp[1].samp_v == 1 && g.comp[2].samp_h == 1 &&
           g.comp[2].samp_v == 1 && g.comp[0].blocks_x == 2 * g.comp[1].blocks_x && g.comp[0].blocks_y == g.comp[1].blocks_y &&
           g.comp[2].blocks_x == g.comp[1].blocks_x && g.comp[2].blocks_y == g.comp[1].blocks_y;
}
// which kernel leaves the tile streams k_gather takes: 1 = k_encode_uyvy422, 2 = k_encode_blocks, 3 = k_encode_rgb444, 0 = none (coefficient planes + k_huffman)
static int gj_tile_kernel(const gj_enc_job* job)
{
    const gj_geom& g = job->g;
    const bool segs_ok = g.restart_interval > 0 && g.seg_blocks <= 256 && g.seg_blocks >= 256 / GJ_ENC_MAX_SPT;
    gj_encode_kernel_t whole = (job->use_fused && !job->keep_coefs) ? gj_encode_kernel(g) : nullptr;
    if (whole && job->tune.enc_by_blocks > 0 && gj_blocks_kernel_mode(g) == 0) whole = nullptr; // (k_encode_blocks instead)
    if (gj_is_uyvy_layout(job) && !job->keep_coefs && g.interleaved && segs_ok && g.blocks_per_mcu == 4 && g.mcu_count == g.comp[1].blocks_x * g.comp[1].blocks_y &&
        g.comp[1].type == g.comp[2].type && !job->tune.enc_no_whole422)
        return 1;
    if (!whole && job->use_fused && !job->keep_coefs && segs_ok && gj_blocks_kernel_mode(g) >= 0) return 2;
    return whole ? 3 : 0;
}

// a batch of frames takes the kernels that go from pixels to tile streams (k_encode_rgb444, k_encode_uyvy422, k_encode_blocks: every layout with restart
// segments of 4 .. 256 blocks) and k_gather, and no option that touches other buffers
extern "C" int gj_hip_encode_batchable(const gj_enc_job* job)
{
    return job->use_fused && !job->keep_coefs && !job->channel_remap && !job->flipped && !job->segment_info && gj_tile_kernel(job) != 0;
}

extern "C" int gj_hip_encode_tiles(const gj_enc_job* job) { return gj_tile_kernel(job) != 0; }

extern "C" int gj_hip_encode(const gj_enc_job* job, gj_stream_t stream, gj_event_t ev[GJ_ENC_EVENTS])
{
    hipStream_t st = (hipStream_t)stream;
    const gj_geom& g = job->g;
    if (g.blocks_per_mcu > GJ_MAX_MCU_BLOCKS) return -1;
    const unsigned frames = job->batch.count > 1 ? job->batch.count : 1u;
    if (frames > 1 && (!gj_hip_encode_batchable(job) || frames > 65535u)) return -1;
    gj_hip_note_reset();
    if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[0], st));
    if (job->channel_remap) { // the reference permutes the channels of the raw image in place first (src/gpujpeg_preprocessor.cu:570-575)
        const unsigned n = (unsigned)g.width * (unsigned)g.height;
        hipLaunchKernelGGL(k_channel_remap, dim3((n + 255) / 256), dim3(256), 0, st, g, const_cast<uint8_t*>(job->d_raw), job->channel_remap & 0xFFFFu);
    }
    bool tiles = true; // k_encode_*: tile streams for k_gather
    GjTail T;
    T.npieces = 0;
    const int tile_kernel = gj_tile_kernel(job);
    gj_encode_kernel_t whole = tile_kernel == 3 ? gj_encode_kernel(g) : nullptr;
    gj_fused_kernel_t fused = job->use_fused ? gj_fused_kernel(g) : nullptr;
    const bool uyvy = gj_is_uyvy_layout(job);
    if (tile_kernel == 1) {
        if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[1], st));
        if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[2], st));
        const int spt = 256 / g.seg_blocks;
        const unsigned wgs = ((unsigned)g.segment_count + spt - 1) / spt;
        T = gj_make_tail(job, wgs, {0u, ~0u, ~0u, ~0u}, (unsigned)spt);
        hipLaunchKernelGGL(gj_encode_uyvy422_kernel(), dim3(wgs, 1, frames), dim3(256), 0, st, g, job->d_raw, job->d_fwd_q[0], job->d_fwd_q[1], job->d_huff_lut,
                           job->d_temp, job->d_seg_bytes, job->d_seg_ff, T);
    } else if (tile_kernel == 2) {
        // every other layout with short restart segments: one lane per block in coding order (k_encode_blocks)
        if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[1], st));
        if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[2], st));
        const int spt = 256 / g.seg_blocks;
        unsigned wgs = 0, scan_first[GJ_MAX_COMP] = {0u, ~0u, ~0u, ~0u};
        if (g.interleaved) wgs = ((unsigned)g.segment_count + spt - 1) / spt;
        else
            for (int c = 0; c < g.comp_count; c++) {
                scan_first[c] = wgs;
                wgs += ((unsigned)g.comp[c].segment_count + spt - 1) / spt;
            }
        T = gj_make_tail(job, wgs, scan_first, (unsigned)spt);
        hipLaunchKernelGGL(gj_encode_blocks_kernel(gj_blocks_kernel_mode(g) != 0), dim3(wgs, 1, frames), dim3(256), 0, st, g, job->d_raw, job->d_fwd_q[0],
                           job->d_fwd_q[1], job->d_huff_lut, job->d_temp, job->d_seg_bytes, job->d_seg_ff, T);
    } else if (whole) { // pixels -> segment streams in one kernel, no coefficient planes
        if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[1], st));
        if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[2], st));
        const int spt = 256 / g.seg_blocks;
        const unsigned wgs = ((unsigned)g.comp[0].segment_count + spt - 1) / spt;
        T = gj_make_tail(job, 3 * wgs, {0u, wgs, 2 * wgs, ~0u}, (unsigned)spt);
        // (a component per workgroup while three times the tiles still fit the places the GPU has: GJ_ENC_SPLIT=<tiles> moves the limit, 0 = never)
        const unsigned split_up_to = job->tune.enc_split >= 0 ? (unsigned)job->tune.enc_split : (unsigned)gj_hip_cu_count() * 4u / 3u; // (MI355X: 341)
        const bool split = wgs * frames <= split_up_to;
        if (split) whole = gj_encode_kernel(g, true);
        // (a frame with more tiles than the GPU has places: its last tiles as three workgroups each, see the kernel; GJ_ENC_TAIL=<tiles>, 0 = none)
        unsigned grid_x = wgs;
        const unsigned tail = job->tune.enc_tail >= 0 ? (unsigned)job->tune.enc_tail : 32u;
        T.tail_from = 0xFFFFFFFFu;
        T.tiles = wgs;
        // (an explicit GJ_ENC_TAIL applies to frames of any size: the tests reach the mixture with small frames that way)
        if (!split && frames == 1 && tail > 0 && wgs > tail && (job->tune.enc_tail > 0 || wgs > (unsigned)gj_hip_cu_count() * 4u)) {
            T.tail_from = wgs - tail;
            grid_x = wgs + 2u * tail;
        }
        hipLaunchKernelGGL(whole, dim3(grid_x, split ? 3 : 1, frames), dim3(256), 0, st, g, job->d_raw, job->d_fwd_q[0], job->d_fwd_q[1], job->d_huff_lut, job->d_temp,
                           job->d_seg_bytes, job->d_seg_ff, T);
    } else {
    tiles = false;
    if (uyvy) { // packed 4:2:2 without colour transform: pixels -> coefficients, one thread per MCU
        if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[1], st));
        const unsigned nm = (unsigned)(g.comp[1].blocks_x * g.comp[1].blocks_y);
        hipLaunchKernelGGL(gj_fused_uyvy422_kernel(), dim3((nm + 255) / 256), dim3(256), 0, st, g, job->d_raw, job->d_coefs, job->d_fwd_q[0], job->d_fwd_q[1]);
    } else if (fused) {
        if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[1], st));
        const unsigned nb = (unsigned)(g.comp[0].blocks_x * g.comp[0].blocks_y);
        hipLaunchKernelGGL(fused, dim3((nb + 255) / 256), dim3(256), 0, st, g, job->d_raw, job->d_coefs, job->d_fwd_q[0],
                           job->d_fwd_q[1]);
    } else {
        if (g.no_transform) {
            hipLaunchKernelGGL(k_copy_planes_in, dim3(2048), dim3(256), 0, st, g, job->d_raw, job->d_planes);
        } else {
            const unsigned n = (unsigned)g.raw_width * (unsigned)g.height;
            hipLaunchKernelGGL(k_preprocess, dim3((n + 255) / 256), dim3(256), 0, st, g, job->d_raw, job->d_planes);
        }
        if (job->flipped) hipLaunchKernelGGL(k_flip_planes, dim3(1024), dim3(256), 0, st, g, job->d_planes);
        if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[1], st));
        hipLaunchKernelGGL(k_dct, dim3(((unsigned)g.block_count + 255) / 256), dim3(256), 0, st, g, job->d_planes, job->d_coefs,
                           job->d_fwd_q[0], job->d_fwd_q[1]);
    }
    if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[2], st));
    const int B = g.seg_blocks;
    const int spt = B <= 256 ? 256 / B : 1;
    const unsigned tiles = ((unsigned)g.segment_count + spt - 1) / spt;
    hipLaunchKernelGGL(k_huffman, dim3(tiles), dim3(256), 0, st, g, job->d_coefs, job->d_huff_lut, job->d_temp, job->d_seg_bytes,
                       job->d_seg_ff);
    }
    if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[3], st));
    if (tiles) // the tile streams -> the file: one wave each
        hipLaunchKernelGGL(k_gather, dim3((T.npieces + 3) / 4, 1, frames), dim3(256), 0, st, T);
    const bool seg_info = job->segment_info && g.restart_interval > 0;
    // (behind k_gather the segment offsets are needed for the APP13 index only)
    const unsigned scan_wgs = ((unsigned)g.segment_count + 1023) / 1024;
    if (!tiles || seg_info)
        hipLaunchKernelGGL(k_scan_segments, dim3(scan_wgs), dim3(1024), 0, st, *job, (unsigned long long*)job->d_scan_partial, job->epoch);
    if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[4], st));
    if (!tiles)
        hipLaunchKernelGGL(k_assemble, dim3(((unsigned)g.segment_count + 4 * GJ_ASM_SEGS - 1) / (4 * GJ_ASM_SEGS)), dim3(256), 0, st, *job);
    if (seg_info)
        hipLaunchKernelGGL(k_segment_info, dim3(((unsigned)g.segment_count + 255) / 256), dim3(256), 0, st, *job);
    if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[5], st));
    return (hipGetLastError() == hipSuccess && !gj_hip_noted()) ? 0 : -1;
}

