/*
 * gj_hip.h -- the thin C-ABI seam between the host C library (encoder/decoder drivers, JFIF writer and
 * reader, tables) and the hand-written gfx950 HIP code. Plain pointers and sizes only; the host side
 * never includes a HIP header.
 *
 * It replaces the reference's internal seam: the CUDA runtime subset the host C uses
 * (cudaMalloc/cudaMemcpyAsync/cudaEvent*, enumerated over the files of src/) and the five CUDA module groups
 *   gpujpeg_preprocessor_encode      src/gpujpeg_preprocessor.h:84-98
 *   gpujpeg_dct_gpu / gpujpeg_idct_gpu  src/gpujpeg_dct_gpu.h:46-55
 *   gpujpeg_huffman_gpu_encoder_*    src/gpujpeg_huffman_gpu_encoder.h:48-68
 *   gpujpeg_huffman_gpu_decoder_*    src/gpujpeg_huffman_gpu_decoder.h:47-60
 *   gpujpeg_postprocessor_decode     src/gpujpeg_postprocessor.h:46-59
 * coarsened into one launcher per direction (gj_hip_encode / gj_hip_decode), because on MI355X the
 * stages are fused and the stream is assembled on the device.
 */
#ifndef GJ_HIP_H
#define GJ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GJ_HIP_API __attribute__((visibility("default")))

typedef void* gj_stream_t; /* hipStream_t */
typedef void* gj_event_t;  /* hipEvent_t */

/* ------------------------------------------------------------------ runtime subset */
GJ_HIP_API int gj_hip_device_count(void);
GJ_HIP_API int gj_hip_get_device(int* device);
GJ_HIP_API int gj_hip_set_device(int device);
GJ_HIP_API int gj_hip_device_reset(void);
GJ_HIP_API int gj_hip_device_props(int device, char name[256], int* major, int* minor, size_t* global_mem, size_t* shared_mem,
                        int* regs_per_block, int* cu_count);
/* compute units of the current device (cached): what "the workgroups the device holds at once" is derived from */
GJ_HIP_API int gj_hip_cu_count(void);
GJ_HIP_API int gj_hip_runtime_version(int* driver, int* runtime);
GJ_HIP_API const char* gj_hip_last_error(void);

GJ_HIP_API void* gj_hip_malloc(size_t size);
GJ_HIP_API void gj_hip_free(void* p);
GJ_HIP_API void* gj_hip_host_alloc(size_t size); /* pinned */
GJ_HIP_API void gj_hip_host_free(void* p);
GJ_HIP_API int gj_hip_memcpy_h2d(void* dst, const void* src, size_t n, gj_stream_t s); /* async on s */
GJ_HIP_API int gj_hip_memcpy_d2h(void* dst, const void* src, size_t n, gj_stream_t s);
GJ_HIP_API int gj_hip_memcpy_d2d(void* dst, const void* src, size_t n, gj_stream_t s);
/* Copies of whole images between host and device. Large ones go through the process's one stream per direction and device ("copy lanes",
 * gj_runtime.hip: the host link carries both directions at once only that way) and are COMPLETE when the call returns: the thread waits for the
 * copy on `done`, an event of the caller that no other call in flight uses; gj_hip_download waits for the work on s first. Small copies,
 * done == NULL and the developer setting GJ_COPY_LANES=0: asynchronous on s, exactly gj_hip_memcpy_h2d / _d2h. */
GJ_HIP_API int gj_hip_upload(void* dst, const void* src, size_t n, gj_stream_t s, gj_event_t done);
GJ_HIP_API int gj_hip_download(void* dst, const void* src, size_t n, gj_stream_t s, gj_event_t done);
/* several such copies of one direction: copy with gj_hip_memcpy_h2d / _d2h on the stream gj_hip_lane_begin returns (the lane -- behind the work on s
 * when `download` -- or s itself), then gj_hip_lane_end: complete on return when that was a lane, asynchronous on s otherwise */
GJ_HIP_API gj_stream_t gj_hip_lane_begin(int download, size_t bytes_each, gj_stream_t s);
GJ_HIP_API int gj_hip_lane_end(gj_stream_t lane, gj_stream_t s, gj_event_t done);
GJ_HIP_API int gj_hip_memset(void* dst, int value, size_t n, gj_stream_t s);
GJ_HIP_API int gj_hip_stream_sync(gj_stream_t s);
/* 1 if p points to device (HBM) memory, 0 for host memory (reference: test/unit/run_tests.c:40-78) */
GJ_HIP_API int gj_hip_is_device_ptr(const void* p);

GJ_HIP_API gj_event_t gj_hip_event_create(void);
GJ_HIP_API void gj_hip_event_destroy(gj_event_t e);
GJ_HIP_API int gj_hip_event_record(gj_event_t e, gj_stream_t s);
GJ_HIP_API float gj_hip_event_elapsed_ms(gj_event_t start, gj_event_t stop); /* synchronises on stop */

/* ------------------------------------------------------------------ geometry shared by host and kernels */
#define GJ_MAX_COMP 4
#define GJ_MAX_MCU_BLOCKS 16     /* blocks per interleaved MCU we accept (JPEG itself allows 10) */
#define GJ_TEMP_BYTES_PER_BLOCK 208 /* >= worst case 1658 bits of an 8x8 block before byte stuffing, 16 B multiple */
/* gj_enc_job.d_tail (k_encode_* -> k_gather): two sets of GJ_TAIL_GROUPS_CAP(segments) group totals, used alternately, then the tile list
 * (one word per tile stream, at most one stream per segment) */
#define GJ_TAIL_GROUPS_CAP(segments) (((unsigned)(segments) + 32u) / 32u + 1u)
#define GJ_TAIL_WORDS(segments) (2u * GJ_TAIL_GROUPS_CAP(segments) + ((unsigned)(segments) + 1u))

enum { GJ_PF_U8 = 0, GJ_PF_444_P012 = 1, GJ_PF_444_P0P1P2 = 2, GJ_PF_422_P1020 = 3, GJ_PF_422_P0P1P2 = 4,
       GJ_PF_420_P0P1P2 = 5, GJ_PF_4444_P0123 = 6 };

typedef struct gj_comp_geom {
    int type;                  /* 0 luminance tables, 1 chrominance tables (encoder) */
    int samp_h, samp_v;
    int sub_h, sub_v;          /* max_h / samp_h, max_v / samp_v */
    int width, height;         /* real size */
    int data_width, data_height;
    int blocks_x, blocks_y;
    int mcu_count_x, mcu_count;
    int segment_count;         /* non-interleaved: segments of this component's scan */
    int first_segment;         /* non-interleaved: global index of its first segment */
    int q_table, dc_table, ac_table; /* decoder: table slots */
    int reserved;
    uint64_t data_offset;      /* offset in samples of the plane / coefficient plane */
} gj_comp_geom;

/* Decoder kernels of a frame batch (gj_dec_job::batch, blockIdx.z = frame): what lies between the buffers of two frames. It travels with
 * the geometry because every decoder kernel takes the geometry by value; sizes == NULL (everything zero) for a single frame. */
typedef struct gj_frame_strides {
    const uint32_t* sizes;    /* [frames] bytes of every frame's stream, device memory */
    uint64_t jpeg, raw;       /* bytes */
    uint64_t coefs, tok, rec; /* int16 coefficients, tokens, block records */
    uint32_t seg;             /* words between the frames' seg_pos (and seg_len, seg_index) */
    uint32_t frames;          /* frames of the launch (what fills the device is the batch, not the frame: gj_hip_decode_wants_tokens) */
} gj_frame_strides;

typedef struct gj_geom {
    int width, height, width_padding;
    int pixel_format, color_space, color_space_internal;
    int comp_count, interleaved, restart_interval;
    int max_h, max_v;
    int mcu_count;             /* interleaved: MCUs in the scan */
    int mcu_count_x;           /* interleaved */
    int segment_count;         /* all scans */
    int scan_count;
    int blocks_per_mcu;        /* 1 when not interleaved */
    int seg_blocks;            /* blocks in a full segment (restart_interval * blocks_per_mcu; whole scan if 0) */
    int block_count;
    int raw_width;             /* width used by the pixel kernels (even for 4:2:2 packed) */
    int no_transform;          /* planar copy path: raw layout equals component layout */
    uint64_t data_size;        /* samples in all planes */
    uint64_t raw_size;
    gj_comp_geom comp[GJ_MAX_COMP];
    uint8_t mcu_comp[GJ_MAX_MCU_BLOCKS]; /* interleaved MCU layout: component of block p */
    uint8_t mcu_bx[GJ_MAX_MCU_BLOCKS];   /* block x inside the MCU */
    uint8_t mcu_by[GJ_MAX_MCU_BLOCKS];
    uint8_t mcu_prev[GJ_MAX_MCU_BLOCKS]; /* distance (in scan order) to the previous block of the same component */
    gj_frame_strides fb;       /* decoder kernels of a frame batch; all zero for a single frame */
} gj_geom;

/* ------------------------------------------------------------------ encoder */
#define GJ_CODER_LUT_OFFSET 1024 /* words of gj_enc_job::d_huff_lut in front of the coder's tables */
#define GJ_CODER_LUT_WORDS 544
/* Developer settings (A/B measurements, forced modes of the tests): process-wide values set through gj_hip_tuning_setting("NAME=VALUE") -- the names
 * in the comments below, public as gpujpeg_amd_tuning (gpujpeg_amd_ext.h) --, copied into a coder ONCE, when it is created (gj_hip_tuning_defaults);
 * the launchers see the coder's copy only. The release library does not read the environment; builds with -DGJ_TUNING_ENV (`make trace`, `make variant`,
 * the CPU execution model of tests/hipemu) also take the settings whose names are in the environment at that moment. */
typedef struct gj_tuning {
    int no_fused;        /* GPUJPEG_NO_FUSED: generic kernels only */
    int enc_split;       /* GJ_ENC_SPLIT=<tiles>: frames of up to so many tiles are coded one component per workgroup (-1: the default limit) */
    int enc_tail;        /* GJ_ENC_TAIL=<tiles>: so many of the last tiles of a frame larger than the GPU are coded one component per workgroup (-1: default 32, 0: none) */
    int dec_balance;     /* GJ_DEC_BALANCE=1: sparse frames' batches cut by estimated time so that they fill one generation (latency of a lone decode) */
    int host_timing;     /* GJ_HOST_TIMING=1: the coder prints where the host's time of its calls went when it is destroyed */
    int host_scan;       /* GPUJPEG_HOST_SCAN: the decoder walks the stream on the host like the reference */
    int enc_no_whole422; /* GJ_ENC_NO_WHOLE422: packed 4:2:2 through k_fused_uyvy422 + k_huffman instead of k_encode_uyvy422 */
    int dec_tokens;      /* GJ_DEC_TOKENS=1 -> 1 (token mode wherever a token-fed IDCT exists), GJ_DEC_NO_TOKENS -> 0, else -1 = by frame */
    int dec_serial;      /* GJ_DEC_ENTROPY=serial: lane-per-segment entropy decoder */
    int dec_batch;       /* GJ_DEC_G: segments per batch of the sub-sequence decoder, 0 = automatic */
    int dec_sub;         /* GJ_DEC_SUB: bytes per sub-sequence, 0 = automatic */
    int dec_no_spec;     /* GJ_DEC_NO_SPEC: no speculative launch on a cached header */
    int dec_seq;         /* GJ_DEC_SEQ: 1 = always the lane-per-segment entropy decoder in plane mode, 2 (GJ_DEC_SEQ=0) = never, 0 = by the frame */
    int debug_sync;      /* GJ_DEC_DEBUG_SYNC=1: wait after every decoder stage and name it on stderr */
    int scan_shape;      /* GJ_SCAN_SHAPE=<pieces * 100 + rounds>: 4 KB pieces per round (1, 2, 4, 8, 16) and rounds (1 .. 4) of a workgroup of the marker scan
                            instead of the choice by the stream's size (tests: every shape on small streams) */
    int dec_tok_nocoop;  /* GJ_DEC_TOK_NOCOOP=1: the token-mode entropy decoder copies its batch segment by segment instead of as one piece (A/B) */
    int enc_by_blocks;   /* GJ_ENC_BLOCKS: packed RGB 4:4:4 through k_encode_blocks (a workgroup codes one component of its tile: three times the
                            workgroups, a third of the work each) 1 = always, -1 = never, 0 = small frames only */
    int dec_fill;        /* GJ_DEC_FILL=<32nds>: how full the sub-sequence decoders' LDS stage is planned on average (default 23, up to 27 when that keeps
                            a single frame within one generation of workgroups); A/B runs of the batch plan */
    int dec_careful;     /* set by the host for ONE call, never from the environment: a kernel that takes whole segments into LDS met one that
                            does not fit (overflow flag) -- this call uses the kernels without that limit */
} gj_tuning;
GJ_HIP_API void gj_hip_tuning_defaults(gj_tuning* t);
GJ_HIP_API int gj_hip_tuning_setting(const char* setting);          /* "NAME=VALUE" or "NAME"; NULL: back to the defaults. 0, or -1 for an unknown name */
GJ_HIP_API const char* const* gj_hip_tuning_names(int* count); /* the names gj_hip_tuning_setting knows */

/* Frame batch (MI355X extension, gpujpeg_amd_{encoder_encode,decoder_decode}_batch): `count` frames of ONE geometry, tables and header behind
 * one set of launches (blockIdx.z = frame) -- an HD frame has 135 encoder tiles for the 1024 workgroup places of the device, a batch fills
 * them. Frame f's buffers lie f x stride behind the job's pointers. count <= 1: a single frame, the strides are ignored. */
typedef struct gj_batch {
    uint32_t count;
    uint64_t raw;            /* bytes between the frames' pixels (d_raw) */
    uint64_t jpeg;           /* bytes between the frames' streams (d_jpeg) */
    uint64_t temp;           /* encoder: bytes between the frames' tile areas (d_temp) */
    uint64_t coefs;          /* decoder: int16 elements between the frames' coefficient planes */
    uint64_t tok;            /* decoder: tokens between the frames' token arrays */
    uint64_t rec;            /* decoder: block records between the frames' record arrays */
    uint32_t seg;            /* words between the frames' per-segment arrays (encoder: seg_bytes, seg_ff; decoder: each of seg_pos, seg_len, seg_index) */
    uint32_t tail;           /* encoder: words between the frames' d_tail */
    uint32_t scratch;        /* decoder: words between the frames' marker-scan scratch */
    uint32_t maxlen;         /* decoder: words between the frames' longest-segment words in pinned host memory */
    const uint32_t* d_sizes; /* decoder: [count] bytes of every frame's stream, in device memory */
} gj_batch;

typedef struct gj_enc_job {
    gj_geom g;
    const uint8_t* d_raw;          /* input pixels in HBM */
    uint8_t* d_planes;             /* padded planar components (generic path only) */
    int16_t* d_coefs;              /* quantised coefficients, 64 per block */
    const float* d_fwd_q[2];       /* forward tables, luminance / chrominance (src/gpujpeg_table.c:103-123 layout) */
    const uint32_t* d_huff_lut;    /* [4][256]: (code << 8) | size ; order lumaDC, lumaAC, chromaDC, chromaAC; behind them, at GJ_CODER_LUT_OFFSET, the
                                      GJ_CODER_LUT_WORDS words of the fused encoders' coder: per table type AC[(run << 4) | ((16 - nbits) & 15)] (256) then DC[nbits] (16),
                                      entry = (code bits + nbits) << 26 | code << nbits (gj_huffman_coder_lut) */
    uint8_t* d_temp;               /* per-segment unstuffed bitstreams */
    uint32_t* d_seg_bytes;         /* [segment_count] unstuffed byte count */
    uint32_t* d_seg_ff;            /* [segment_count] number of 0xFF bytes */
    uint32_t* d_seg_out;           /* [segment_count + 1] final byte offset of each segment in the JPEG */
    uint8_t* d_jpeg;               /* finished stream */
    uint64_t jpeg_capacity;
    uint32_t* d_result;            /* [0] total JPEG size, [1] overflow flag */
    uint32_t* h_result;            /* optional: the same two words in pinned host memory, written by the kernel that computes them */
    uint32_t* d_tail;              /* k_encode_* -> k_gather: GJ_TAIL_WORDS(segment_count) words, see there; the group totals are zero at allocation
                                      and after every call (two sets, used alternately) */
    int tail_set;                  /* which of d_tail's two sets of group totals this call uses: alternates from call to call */
    uint64_t* d_scan_partial;      /* [ceil(segment_count / 1024)] epoch-tagged workgroup totals of the offset scan; zero at allocation */
    uint32_t epoch;                /* differs from the previous call's (and is never 0) */
    gj_tuning tune;
    const uint8_t* d_scan_hdr;     /* scan headers back to back (APP13 placeholders zeroed + SOS) */
    uint32_t scan_hdr_offset[GJ_MAX_COMP + 1];
    uint32_t scan_info_payload[GJ_MAX_COMP]; /* offset inside scan header of the first APP13 payload, 0 = none */
    uint32_t main_hdr_size;        /* bytes already placed at d_jpeg[0..) by the host */
    int segment_info;
    int use_fused;                 /* 1: fused kernels when the format allows (raw -> segment streams, or raw -> coefficients) */
    int keep_coefs;                /* 1: the caller wants the coefficient planes in d_coefs (tests): do not use the fully fused kernel */
    int flipped;                   /* enc_opt_flipped: flip the component planes vertically after the colour stage (generic path) */
    uint32_t channel_remap;        /* enc_opt_channel_remap: packed mapping, 0 = none; applied to d_raw IN PLACE before anything else */
    gj_batch batch;                /* count > 1: a batch of frames (fully fused 4:4:4 kernel only; d_result / h_result hold two words per frame) */
} gj_enc_job;
/* 1 when gj_hip_encode takes a batch (gj_enc_job::batch.count > 1) of this job's configuration */
GJ_HIP_API int gj_hip_encode_batchable(const gj_enc_job* job);
/* 1 when gj_hip_encode codes this job through tile streams + k_gather (the launch that uses and clears gj_enc_job::tail_set's group totals) */
GJ_HIP_API int gj_hip_encode_tiles(const gj_enc_job* job);

/* events (may be NULL): 0 start, 1 after preprocess, 2 after DCT/quant, 3 after k_huffman, 4 after k_scan_segments,
 * 5 after k_assemble (+ segment info) */
#define GJ_ENC_EVENTS 6
GJ_HIP_API int gj_hip_encode(const gj_enc_job* job, gj_stream_t stream, gj_event_t ev[GJ_ENC_EVENTS]);


/* ------------------------------------------------------------------ decoder */
/* A marker scan whose SECOND launch (k_marker_table: records -> segment table) has been left to gj_hip_decode (gj_hip_find_segments_deferred:
 * speculative single-frame launches). k_huffman_decode_tok derives the table entries of its batches from the scan's records itself and writes the
 * summary the host validates -- one launch and ~11 us less per decoded frame --; for any other entropy decoder gj_hip_decode launches k_marker_table
 * first, with the arguments kept here. valid = 0: the table has been written (gj_hip_find_segments). */
struct gj_scan_summary;
typedef struct gj_scan_deferred {
    int valid;
    uint32_t wgs, part_bytes;      /* scanning workgroups (<= 256) and the bytes of the stream each of them took */
    uint64_t begin, size;
    const uint32_t* recs;          /* [wgs] records, [wgs] marker lists (d_scratch of the scan) */
    const uint32_t* lists;
    uint32_t* d_seg_pos;           /* where k_marker_table writes, should it be launched after all */
    uint32_t* d_seg_len;
    uint32_t* d_seg_index;
    uint32_t max_segments;
    struct gj_scan_summary* d_summary;
    struct gj_scan_summary* h_summary; /* pinned host memory */
    uint32_t* h_maxlen_parts;      /* pinned host memory: the longest segment per scanning workgroup (table launch) or per entropy-decoder batch (folded) */
    uint32_t maxlen_capacity;
    uint32_t* maxlen_part_count;   /* host: how many of those words the launch that ran has written */
    int* folded;                   /* host, may be NULL: set to 1 by gj_hip_decode when the table launch was folded into the entropy decoder, to 0 otherwise */
} gj_scan_deferred;

typedef struct gj_dec_job {
    gj_geom g;                     /* pixel_format / color_space describe the requested output */
    const uint8_t* d_jpeg;         /* whole file in HBM */
    uint64_t jpeg_size;
    const uint32_t* d_seg_pos;     /* [seg_count] byte offset of each segment's entropy data */
    const uint32_t* d_seg_len;     /* [seg_count] */
    const uint32_t* d_seg_index;   /* [seg_count] geometric segment index */
    int seg_count;                 /* number of table entries (upper bound when d_seg_count is set) */
    const uint32_t* d_seg_count;   /* optional: actual count in device memory (written by gj_hip_find_segments) */
    const uint16_t* d_huff_tab;    /* [4 slots][2 classes][GJ_DEC_TAB_WORDS] decode tables */
    const uint16_t* d_huff_tab2;   /* [2 slots][2 classes][GJ_DEC2_WORDS] two-level tables of the sub-sequence decoder, or NULL when the
                                      stream's tables do not fit them (then only the lane-per-segment kernel runs) */
    const uint16_t* d_qtab;        /* [4][64] natural order */
    const float* d_qtabf;          /* the same as float: what the IDCT kernels multiply with */
    int16_t* d_coefs;
    uint8_t* d_planes;
    uint8_t* d_raw;                /* output pixels */
    int use_fused;
    int flipped;                   /* dec_opt_flipped: flip the component planes before the colour stage (generic path) */
    uint32_t channel_remap;        /* dec_opt_channel_remap: applied to the finished image in d_raw */
    int clear_coefs;               /* 1: d_coefs is not known to be all zero, clear it first */
    int zero_coefs;                /* 1: the IDCT kernels zero every block after reading it (next call may skip clear_coefs) */
    gj_tuning tune;
    uint32_t max_seg_len;          /* longest restart segment of the stream when known (see scan_bytes), 0 = unknown */
    uint32_t* d_overflow;          /* one word the entropy decoders that take whole segments into LDS (k_huffman_decode_tok, _seq) set when they
                                      meet a segment they cannot stage */
    uint32_t scan_bytes[GJ_MAX_COMP]; /* entropy-coded bytes of every scan when known (this stream's, or the previous frame's on the speculative
                                         path; 0 = unknown): the entropy decoder sizes its batches per scan with them */
    /* token mode: entropy decoder -> fused IDCT without the coefficient planes (used when a token-fed IDCT kernel exists for the
     * configuration; 0 / NULL = planes) */
    int tokens;
    uint16_t* d_tok;               /* [tok_cap + 64] 16-bit tokens: value << 6 | natural position; the run of a decoder group starts at token
                                      4 x the byte offset of its first segment */
    uint32_t tok_cap;              /* >= 4 x jpeg_size (tokens) */
    void* d_blkrec;                /* [g.block_count] uint2 per block in coding order: first token, count << 16 | (uint16) DC term;
                                      count 0xFFFF = the block is in d_coefs (segment decoded piece by piece) */
    gj_scan_deferred scan;         /* valid: the segment table has not been written yet (see gj_scan_deferred) */
    gj_batch batch;                /* count > 1: a batch of frames with the same header (speculative launches only: d_seg_count and d_overflow
                                      point to frame 0's words inside arrays of gj_scan_summary) */
} gj_dec_job;
/* 1 when gj_hip_decode takes a batch (gj_dec_job::batch.count > 1) of this job's configuration */
GJ_HIP_API int gj_hip_decode_batchable(const gj_dec_job* job);
/* 1 when the IDCT side of this configuration goes through the component planes (gj_dec_job::d_planes; a batch needs a set per frame,
 * gj_frame_strides::coefs bytes apart) */
GJ_HIP_API int gj_hip_decode_uses_planes(const gj_geom* g, int use_fused);

/* 1 when a frame of this geometry (requested output included) and stream size is decoded in token mode: a token-fed IDCT kernel
 * exists for it and the measured size / density rule (or the GJ_DEC_TOKENS / GJ_DEC_NO_TOKENS override) says so. The host asks
 * before it allocates d_tok / d_blkrec. */
GJ_HIP_API int gj_hip_decode_wants_tokens(const gj_geom* g, uint64_t jpeg_size, const gj_tuning* tune);

/* decode table layout per (slot, class): 1024 fast entries (len << 8 | symbol, 0 = miss) followed by
 * maxcode[18] (as u16 pairs lo/hi), valptr[17], mincode[17] and the 256 symbol values */
#define GJ_DEC_FAST_BITS 10
#define GJ_DEC_TAB_WORDS (1024 + 36 + 17 + 34 + 256 + 1)

/* two-level table of the sub-sequence decoder: 1024 first-level entries indexed by the next 10 bits, then up to
 * GJ_DEC2_SUBTABLES second-level tables of 64 entries indexed by the 6 bits after those. Entry: bits [0,5) code length +
 * magnitude bits (0 in the first level = go to the second level, whose word offset is entry >> 5), [5,9) magnitude bits,
 * [9,15) zig-zag advance (DC: 1 + high nibble; AC: run + 1, 16 for ZRL, 63 for EOB: the position inside a block is >= 1, so
 * the block is complete at >= 64), bit 15 set for a non-zero AC coefficient (what the token decoder counts and stores). */
#define GJ_DEC2_SUBTABLES 6
#define GJ_DEC2_WORDS (1024 + 64 * GJ_DEC2_SUBTABLES)

/* events (may be NULL): 0 start, 1 after Huffman, 2 after IDCT, 3 after postprocess */
#define GJ_DEC_EVENTS 4
GJ_HIP_API int gj_hip_decode(const gj_dec_job* job, gj_stream_t stream, gj_event_t ev[GJ_DEC_EVENTS]);

/* Device-side segment discovery: finds RSTn and scan boundaries in the device-resident stream [begin, size) and
 * writes the segment table (offset, length, geometric index) in stream order. The small summary is what the host
 * reads back to validate the structure (and to learn the segment count). Replaces the host memchr walk of
 * src/gpujpeg_reader.c:1039-1155. */
#define GJ_SCAN_MAX_OTHER 16
typedef struct gj_scan_summary {
    uint32_t rst_count;                       /* RSTn markers found */
    uint32_t other_count;                     /* every other marker (SOS of later scans, EOI, ...) */
    uint32_t other_pos[GJ_SCAN_MAX_OTHER];
    uint8_t other_code[GJ_SCAN_MAX_OTHER];
    uint8_t other_bytes[GJ_SCAN_MAX_OTHER][16]; /* the 16 bytes after each of them (length + start of the payload) */
    uint32_t other_after[GJ_SCAN_MAX_OTHER];  /* RSTn markers behind it inside its workgroup's part of the stream */
    uint32_t scan_count, segment_count;
    uint32_t status;                          /* 1 = clean (scans ... EOI), 2 = unexpected marker between scans, 3 = no EOI */
    uint32_t scan_start[GJ_MAX_COMP], scan_end[GJ_MAX_COMP];
    uint32_t header_differs;                  /* 1 when the stream does not start with the cached header (speculative launches) */
    uint32_t max_seg_len;                     /* longest segment of the table (filled in by the host from the per-chunk maxima) */
    uint32_t seq_overflow;                    /* set by k_huffman_decode_tok / _seq: a segment did not fit the LDS stage, decode the frame with the other kernel */
    uint32_t rst_irregular;                   /* an RSTn out of sequence, or an empty segment in front of the end of a scan: the reference reader
                                                 resynchronises / drops it (src/gpujpeg_reader.c:1074-1135), so the host walks such a stream */
} gj_scan_summary;

GJ_HIP_API size_t gj_hip_find_segments_scratch_words(uint64_t begin, uint64_t size, uint32_t max_segments);
/* d_scratch: gj_hip_find_segments_scratch_words() words (the workgroups' records and marker lists between the two launches);
 * d_summary: the device's copy, of which only segment_count is written (speculative launches read it).
 * d_hdr_ref (may be NULL): the cached header of a speculative launch; sets h_summary->header_differs = (d_jpeg[0..hdr_n) != d_hdr_ref[0..hdr_n))
 * h_summary, h_maxlen_parts: PINNED HOST memory the kernels write directly (no copy launches behind them): the summary the host validates
 * once the stream has been waited for (the host clears rst_irregular, seq_overflow and header_differs in it before the launch; max_seg_len
 * is not written by the device: the longest segment of the table is the maximum over *maxlen_part_count words of h_maxlen_parts, one per
 * workgroup) */
GJ_HIP_API int gj_hip_find_segments(const gj_geom* g, const uint8_t* d_jpeg, uint64_t begin, uint64_t size, uint32_t* d_seg_pos,
                                    uint32_t* d_seg_len, uint32_t* d_seg_index, uint32_t max_segments, uint32_t* d_scratch,
                                    gj_scan_summary* d_summary, const uint8_t* d_hdr_ref, uint32_t hdr_n, gj_scan_summary* h_summary,
                                    uint32_t* h_maxlen_parts, uint32_t maxlen_capacity, uint32_t* maxlen_part_count, gj_stream_t stream,
                                    const gj_tuning* tune);
/* The same for the frames of a batch (batch->count streams with the same header: begin, and d_hdr_ref compared with every one of them): `size` is
 * the LONGEST stream, batch->d_sizes the streams' sizes, batch->jpeg / seg / scratch / maxlen the strides of d_jpeg, the three tables, d_scratch
 * and h_maxlen_parts; d_summary and h_summary are arrays of batch->count summaries. batch->jpeg must be a multiple of 16. */
GJ_HIP_API int gj_hip_find_segments_batch(const gj_geom* g, const uint8_t* d_jpeg, uint64_t begin, uint64_t size, uint32_t* d_seg_pos,
                                          uint32_t* d_seg_len, uint32_t* d_seg_index, uint32_t max_segments, uint32_t* d_scratch,
                                          gj_scan_summary* d_summary, const uint8_t* d_hdr_ref, uint32_t hdr_n, gj_scan_summary* h_summary,
                                          uint32_t* h_maxlen_parts, uint32_t maxlen_capacity, uint32_t* maxlen_part_count, gj_stream_t stream,
                                          const gj_tuning* tune, const gj_batch* batch);
/* gj_hip_find_segments for ONE frame whose table launch may be left to gj_hip_decode: launches k_marker_scan, and k_marker_table only when the scan has
 * more than 256 workgroups (streams beyond 16 MB); otherwise *defer describes the pending launch (defer->valid = 1) for gj_dec_job::scan. The caller must
 * decode with that job on the same stream next -- nothing else writes the table. */
GJ_HIP_API int gj_hip_find_segments_deferred(const gj_geom* g, const uint8_t* d_jpeg, uint64_t begin, uint64_t size, uint32_t* d_seg_pos,
                                             uint32_t* d_seg_len, uint32_t* d_seg_index, uint32_t max_segments, uint32_t* d_scratch,
                                             gj_scan_summary* d_summary, const uint8_t* d_hdr_ref, uint32_t hdr_n, gj_scan_summary* h_summary,
                                             uint32_t* h_maxlen_parts, uint32_t maxlen_capacity, uint32_t* maxlen_part_count, gj_stream_t stream,
                                             const gj_tuning* tune, gj_scan_deferred* defer);
/* workgroups the marker scan cuts [begin, size) into at most (capacity of h_maxlen_parts) */
GJ_HIP_API size_t gj_hip_find_segments_max_chunks(uint64_t begin, uint64_t size);

#ifdef __cplusplus
}
#endif
#endif
