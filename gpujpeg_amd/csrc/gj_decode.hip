// gj_decode.hip -- MI355X (gfx950, wave64) JPEG decoder: gj_hip_decode picks the kernels of a frame and launches them.
// The kernels live in gj_dec_*.hip (map: gj_dec_internal.h).
#include "gj_dec_internal.h"

// Does a frame of this geometry and stream size go through token mode (given fused kernels and two-level Huffman tables)? The host
// asks before it allocates the token buffers; the launcher asks again.
// Measured (8K / 16K RGB natural frames at q75, 4.75 B of stream per block: +17 % enc+dec; HD and 4K equal or slightly slower;
// 8K noise -15 %; crossover at 8K RGB near 9 B per block): tokens pay when the frame fills the GPU more than once (the token-fed IDCT has
// the longer dependency chain per workgroup) and blocks carry few coefficients (2 B per coefficient against 128 B per block).
// gj_tuning::dec_tokens forces either mode (tests, A/B runs).
extern "C" int gj_hip_decode_wants_tokens(const gj_geom* g, uint64_t jpeg_size, const gj_tuning* tune)
{
    if (tune->dec_tokens == 0 || gj_idct_tok_for(*g) == nullptr) return 0;
    if (g->seg_blocks > GJ_TOK_MAX_BLOCKS || g->restart_interval == 0) return 0; // (k_huffman_decode_tok takes whole segments into its LDS stage)
    if (tune->dec_sub) return 0; // (the tuning aid sweeps the plane-mode kernels)
    if (tune->dec_tokens == 1) return 1;
    // (interleaved scans -- packed 4:2:2, BASELINE config 4 at q90: 10.4 B per block -- go through the lane-per-segment kernel, whose plane mode
    // pays for the zero fill and the scattered stores of 128-byte blocks: tokens win up to denser streams there)
    // (round 3, k_huffman_decode_tok: a 4K RGB frame -- 389 K blocks -- gains 9 % enc+dec and 7 % decode-only, an HD frame loses 4 %)
    // (a batch of frames, gj_frame_strides: the blocks of all its frames fill the GPU, so HD frames take token mode too)
    const uint64_t blocks_in_flight = (uint64_t)g->block_count * (g->fb.frames > 1 ? g->fb.frames : 1u);
    // (round 5, k_huffman_decode_tok on denser streams: the camera frame at q100, 11.8 B per block, decodes 11 % faster through tokens than through the
    // planes -- 133.6 against 120.0 Gpix/s decode-only --; `.tst` noise at q75, 26.8 B per block, 5 % slower -- 34.7 against 36.7: the entropy stage takes
    // 0.908 ms either way, ~2.2 symbols per byte at a sub-sequence hit rate that noise's long codes halve, and the token-fed IDCT then moves as many bytes
    // as the planes would. The gate for non-interleaved scans moves from 8 to 16 B per block.)
    return blocks_in_flight >= (g->interleaved ? 900000u : 300000u) && jpeg_size <= (uint64_t)g->block_count * (g->interleaved ? 12u : 16u);
}

// A batch of frames (gj_dec_job::batch) is a speculative launch on one header through the sub-sequence entropy decoders -- tokens for non-interleaved
// 4:4:4, coefficient planes for everything else (the lane-per-segment kernels and the token-fed 4:2:2 IDCT do not know the frame dimension) -- and any
// of the IDCT-side kernels; no option that touches other buffers.
extern "C" int gj_hip_decode_batchable(const gj_dec_job* job)
{
    const gj_geom& g = job->g;
    return job->d_huff_tab2 != nullptr && job->d_overflow != nullptr && job->d_seg_count != nullptr && job->seg_count > 0 &&
           !job->flipped && !job->channel_remap && !job->clear_coefs && !job->tune.dec_serial && !job->tune.dec_careful && job->tune.dec_seq != 1 &&
           g.restart_interval > 0;
}

extern "C" int gj_hip_decode(const gj_dec_job* job, gj_stream_t stream, gj_event_t ev[4])
{
    hipStream_t st = (hipStream_t)stream;
    const gj_geom& g = job->g;
    if (g.blocks_per_mcu > GJ_MAX_MCU_BLOCKS) return -1;
    if ((job->batch.count > 1 || g.fb.sizes != nullptr) && (!gj_hip_decode_batchable(job) || g.fb.sizes == nullptr || job->batch.count > 65535u)) return -1;
    gj_hip_note_reset();
    if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[0], st));
    bool par = job->d_huff_tab2 != nullptr && job->seg_count > 0;
    if (job->tune.dec_serial) par = false; // the lane-per-segment kernel (A/B measurements, tests)
    const bool fast_ok = par && !job->tune.dec_careful && job->d_overflow != nullptr; // kernels that take whole segments into LDS are allowed
    // interleaved scans with many short segments: one lane per segment (k_huffman_decode_seq); the host vouches for the longest segment
    // (this stream's, or the previous frame's on the speculative path, where `d_overflow` is checked afterwards)
    // (a batch of frames: its segments are in flight together; only the token mode of the lane-per-segment decoder -- the ring kernel -- knows the frame dimension)
    const bool batch = g.fb.sizes != nullptr;
    const uint64_t segs_in_flight = (uint64_t)job->seg_count * (batch && g.fb.frames > 1 ? g.fb.frames : 1u);
    bool seq = fast_ok && job->tune.dec_seq != 2 &&
               (job->tune.dec_seq == 1 || (g.interleaved && job->max_seg_len != 0 && job->max_seg_len <= 1024u && segs_in_flight >= 16384u));
    // token mode: the entropy decoder hands the non-zero coefficients to the fused IDCT as a dense token array plus one
    // record per block instead of through the coefficient planes: k_huffman_decode_tok for non-interleaved scans (every segment has to
    // fit its LDS stage), the lane-per-segment kernel for interleaved ones
    const bool tok_wanted = fast_ok && job->tokens && job->use_fused && job->d_tok && job->d_blkrec && gj_hip_decode_wants_tokens(&g, job->jpeg_size, &job->tune);
    const bool tok_sub = tok_wanted && !g.interleaved && !seq && job->max_seg_len != 0 && job->max_seg_len + 12u <= (uint32_t)GJ_TOK_CAP_U;
    const bool tok_seq = tok_wanted && seq;
    gj_idct_tok_t idct_tok = (tok_sub || tok_seq) ? gj_idct_tok_for(g) : nullptr;
    const bool tokens = idct_tok != nullptr;
    if (batch && !tokens) seq = false; // (planes: the sub-sequence decoder)
    // Both entropy decoders store only non-zero coefficients. The sub-sequence kernel zero-fills the blocks of the segments it
    // decodes itself; clear_coefs asks for a full clear first (segments missing from the table, lane-per-segment kernel).
    if (tokens) {
        if (job->clear_coefs) GJ_HIP_CHECK(hipMemsetAsync(job->d_blkrec, 0, (size_t)g.block_count * sizeof(uint2), st));
    } else if (job->clear_coefs || !par) {
        GJ_HIP_CHECK(hipMemsetAsync(job->d_coefs, 0, g.data_size * sizeof(int16_t), st));
    }
    // a marker scan whose table launch was left to us (gj_scan_deferred): the token decoder reads the scan's records itself, everything else needs the table
    const bool takes_par = !(tokens && tok_sub) && !seq && par;
    const bool folded = job->scan.valid && ((tokens && tok_sub && gj_tok_folds_table(job)) || (takes_par && gj_par_folds_table(job)));
    if (job->scan.valid && !folded) gj_launch_marker_table_deferred(job, st);
    if (job->scan.valid && job->scan.folded) *job->scan.folded = folded ? 1 : 0;
    if (tokens && tok_sub) gj_launch_huffman_tok(job, st);
    else if (seq) gj_launch_huffman_seq(job, st, tokens);
    else if (par) gj_launch_huffman_par(job, st);
    else gj_launch_huffman_serial(job, st);
    gj_debug_stage(job->tune.debug_sync != 0, st, "entropy decoder");
    if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[1], st));
    gj_launch_idct(job, st, idct_tok, ev);
    if (ev) GJ_HIP_CHECK(hipEventRecord((hipEvent_t)ev[3], st));
    return (hipGetLastError() == hipSuccess && !gj_hip_noted()) ? 0 : -1;
}
