// gj_dec_internal.h -- what the files of the decoder's device code share.
//
//   gj_decode.hip               gj_hip_decode: picks the kernels of a frame and launches them
//   gj_dec_markers.hip          k_marker_scan, k_marker_segments: segment table built on the device, summary for the host in pinned memory
//   gj_dec_entropy_tok.hip      k_huffman_decode_tok: sub-sequence parallel entropy decoding into 16-bit TOKENS + one record per block
//                               (the default for large non-interleaved frames)
//   gj_dec_entropy_par.hip      k_huffman_decode_par: sub-sequence parallel entropy decoding into the coefficient planes (segments of any length)
//   gj_dec_entropy_seq.hip      one lane per restart segment (interleaved scans with many short segments): k_huffman_decode_win over a ring the
//                               lane refills itself (token mode), k_huffman_decode_seq over an LDS stage (plane mode)
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

// ---- the marker scan's second launch when it was left to gj_hip_decode (gj_dec_job::scan)
void gj_launch_marker_table_deferred(const gj_dec_job* job, hipStream_t st);
// does gj_launch_huffman_tok derive its batches' table entries from the scan's records itself (no k_marker_table launch)?
bool gj_tok_folds_table(const gj_dec_job* job);

// ---- entropy decoders: each launches its kernel for the whole segment table of the job
void gj_launch_huffman_serial(const gj_dec_job* job, hipStream_t st);
void gj_launch_huffman_par(const gj_dec_job* job, hipStream_t st);
void gj_launch_huffman_seq(const gj_dec_job* job, hipStream_t st, bool tokens);
void gj_launch_huffman_tok(const gj_dec_job* job, hipStream_t st);

// Batches of the sub-sequence decoders: consecutive table entries, cut per scan -- the luminance segments of a photograph carry two to
// three times the bytes of the chrominance ones, and a batch is sized to fill the LDS stage (one batch size for the whole stream
// gave luminance batches that had to be decoded as two groups, and chrominance batches that left half of the lanes idle).
struct GjBatchPlan {
    int n;                       // ranges (scans)
    int first[GJ_MAX_COMP];      // first table entry of range c
    int count[GJ_MAX_COMP];      // entries
    int g[GJ_MAX_COMP];          // segments per batch
    int batch0[GJ_MAX_COMP + 1]; // first batch of range c; [n] = number of batches
};
// cap_u: bytes of the kernel's LDS stage, max_blocks / gmax: blocks / segments a batch may have; resident (0: no preference): the workgroups
// of the kernel the GPU holds at once -- fuller batches are preferred when they keep the launch within one such generation
GjBatchPlan gj_plan_batches(const gj_dec_job* job, unsigned cap_u, unsigned max_blocks, unsigned gmax, unsigned resident);

// k_huffman_decode_tok (shared with the launcher's choice of the kernel)
#define GJ_TOK_CAP_U 10752     // bytes of unstuffed stream per group, incl. 8 B of zero padding per segment
#define GJ_TOK_MAX_BLOCKS 2304 // blocks per batch
#define GJ_TOK_RESIDENT ((unsigned)gj_hip_cu_count() * 4u)  // workgroups of the token decoder the GPU holds at once (MI355X: 256 CUs x 4)
#define GJ_TOK_GMAX 64         // segments per batch

// ---- what k_marker_scan leaves per scanning workgroup (gj_dec_markers.hip), read by k_marker_table and -- when the table launch is folded into it --
// by k_huffman_decode_tok
#define GJ_SCAN_LIST 2048    // restart markers a workgroup may hold (beyond that the host walks the stream)
#define GJ_SCAN_REC_WORDS 16 // 32-bit words of a record (layout: gj_dec_markers.hip)
struct GjOther { uint32_t pos, code, after, b03, b47; }; // a marker that is no restart marker: position, code, restart markers of its workgroup behind it, the 8 bytes behind the code
// other marker q (0, 1) of a record whose words 4 .. 15 are r1, r2, r3
__device__ __forceinline__ GjOther gj_rec_other(const uint4& r1, const uint4& r2, const uint4& r3, const uint32_t q)
{
    return q == 0 ? GjOther{r1.x, r1.y, r1.z, r1.w, r2.x} : GjOther{r2.z, r2.w, r3.x, r3.y, r3.z};
}
// the 16 bits behind the code: the length of the marker's segment (big-endian in the stream)
__device__ __forceinline__ uint32_t gj_other_len(const GjOther& o) { return ((o.b03 & 0xFFu) << 8) | ((o.b03 >> 8) & 0xFFu); }

// ---- IDCT side
typedef void (*gj_idct_tok_t)(const gj_geom, const int16_t*, const uint2*, const uint16_t*, uint32_t, const float*, uint8_t*);
gj_idct_tok_t gj_idct_tok_for(const gj_geom& g); // the token-fed IDCT kernel for this configuration, or nullptr
bool gj_is_uyvy422(const gj_geom& g);
// dequantisation + IDCT + postprocessing of the frame; ev (may be null): events 2 and 3 of gj_hip_decode
void gj_launch_idct(const gj_dec_job* job, hipStream_t st, gj_idct_tok_t idct_tok, gj_event_t* ev);
