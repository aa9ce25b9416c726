// gj_dec_internal.h -- what the files of the decoder's device code share.
//
//   gj_decode.hip               gj_hip_decode: picks the kernels of a frame and launches them
//   gj_dec_markers.hip          k_marker_count / rank / emit, k_build_segments, k_compare_header: segment table built on the device
//   gj_dec_entropy_par.hip      k_huffman_decode_par: sub-sequence parallel entropy decoding of batches of restart segments (the default); output
//                               either the coefficient planes or, in token mode, a dense token array + one record per block
//   gj_dec_entropy_seq.hip      k_huffman_decode_seq: one lane per restart segment over an LDS stage (interleaved scans with many short segments)
//   gj_dec_entropy_serial.hip   k_huffman_decode: one lane per restart segment, stream windows (Huffman tables that do not fit the two-level layout)
//   gj_dec_idct.hip             k_idct_fused_* (from the planes), k_idct_tok_* (from tokens), k_idct / k_postprocess / k_copy_planes_out (generic)
//   gj_bitreader.h              unstuffing of a restart segment into an LDS stage, two-level table look-up
//
// Restates src/gpujpeg_huffman_gpu_decoder.cu:135-495 (entropy decoding semantics; identical results to
// src/gpujpeg_huffman_cpu_decoder.c:245-372), src/gpujpeg_dct_gpu.cu:312-366,472-618 and src/gpujpeg_postprocessor.cu:49-217.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "gj_device.h"
#include "gj_hip.h"

// developer aid (gj_tuning::debug_sync): waits after every launch and names the stage on stderr (which kernel faulted?)
static inline void gj_debug_stage(const bool on, hipStream_t st, const char* what)
{
    if (!on) return;
    const hipError_t e = hipStreamSynchronize(st);
    fprintf(stderr, "[GPUJPEG] [Debug] %s: %s\n", what, hipGetErrorString(e));
}

// ---- entropy decoders: each launches its kernel for the whole segment table of the job
void gj_launch_huffman_serial(const gj_dec_job* job, hipStream_t st);
void gj_launch_huffman_par(const gj_dec_job* job, hipStream_t st, bool tokens);
void gj_launch_huffman_seq(const gj_dec_job* job, hipStream_t st);

// ---- IDCT side
typedef void (*gj_idct_tok_t)(const gj_geom, const int16_t*, const uint2*, const uint32_t*, uint32_t, const float*, uint8_t*);
gj_idct_tok_t gj_idct_tok_for(const gj_geom& g); // the token-fed IDCT kernel for this configuration, or nullptr
bool gj_is_uyvy422(const gj_geom& g);
// dequantisation + IDCT + postprocessing of the frame; ev (may be null): events 2 and 3 of gj_hip_decode
void gj_launch_idct(const gj_dec_job* job, hipStream_t st, gj_idct_tok_t idct_tok, gj_event_t* ev);
