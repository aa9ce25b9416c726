/*
 * gpujpeg_amd_ext.h -- MI355X-specific additions next to the unchanged libgpujpeg API.
 * They expose intermediate device state for stage-level parity tests and benchmarks; production callers
 * never need them.
 */
#ifndef GPUJPEG_AMD_EXT_H
#define GPUJPEG_AMD_EXT_H

#include <stddef.h>
#include <stdint.h>

#include "libgpujpeg/gpujpeg_common.h"

#ifdef __cplusplus
extern "C" {
#endif

struct gpujpeg_encoder;
struct gpujpeg_decoder;

/* copy the quantised coefficients of the last encode/decode call (int16, 64 per 8x8 block, blocks in
 * raster order per component, components back to back = the reference's d_data_quantized layout,
 * src/gpujpeg_dct_gpu.cu:286-294) to host memory; returns the number of coefficients, 0 on error */
GPUJPEG_API size_t gpujpeg_amd_encoder_read_coefficients(struct gpujpeg_encoder* encoder, int16_t* dst, size_t capacity);
GPUJPEG_API size_t gpujpeg_amd_decoder_read_coefficients(struct gpujpeg_decoder* decoder, int16_t* dst, size_t capacity);
/* padded planar component samples (generic path only) */
GPUJPEG_API size_t gpujpeg_amd_encoder_read_planes(struct gpujpeg_encoder* encoder, uint8_t* dst, size_t capacity);
GPUJPEG_API size_t gpujpeg_amd_decoder_read_planes(struct gpujpeg_decoder* decoder, uint8_t* dst, size_t capacity);
/* 1 = use the fused fast-path kernels when the format allows (default), 0 = always take the generic path */
GPUJPEG_API void gpujpeg_amd_encoder_set_fused(struct gpujpeg_encoder* encoder, int enabled);
GPUJPEG_API void gpujpeg_amd_decoder_set_fused(struct gpujpeg_decoder* decoder, int enabled);
/* 1 = the following encode calls leave the quantised coefficients in HBM for gpujpeg_amd_encoder_read_coefficients (tests).
 * Default 0: where the format allows, pixels go to entropy-coded segments in one kernel and no coefficient planes exist. */
GPUJPEG_API void gpujpeg_amd_encoder_keep_coefficients(struct gpujpeg_encoder* encoder, int enabled);
/* 1 = leave the coefficients of the following decode calls in HBM for gpujpeg_amd_decoder_read_coefficients (tests).
 * Default 0: the IDCT clears each block once it has read it, which saves the per-frame clear of the coefficient planes. */
GPUJPEG_API void gpujpeg_amd_decoder_keep_coefficients(struct gpujpeg_decoder* decoder, int enabled);

/* Developer settings: forced kernel paths for tests and A/B measurements ("GJ_DEC_TOKENS=1", "GJ_ENC_TAIL=0", ...; the table in INTEGRATION.md).
 * Process-wide; a coder takes the values that are set when it is CREATED. `setting` is "NAME=VALUE" or "NAME"; NULL forgets every setting.
 * Returns 0, -1 for a name the library does not know. The library never reads the environment (round 6). */
GPUJPEG_API int gpujpeg_amd_tuning(const char* setting);
/* the names gpujpeg_amd_tuning knows, NULL-terminated */
GPUJPEG_API const char* const* gpujpeg_amd_tuning_names(void);

/* Host-only helper (no device access): the marker segments the encoder would emit for these parameters --
 * everything up to the first scan (SOI .. COM), followed by the scan headers back to back. Parameters are
 * adjusted exactly like gpujpeg_encoder_encode() does on a fresh encoder (comp_count 0, RESTART_AUTO).
 * Returns the number of bytes written, 0 on error. Used by the CPU test-suite to check tables, geometry
 * and the writer against the oracle without a GPU. */
GPUJPEG_API size_t gpujpeg_amd_host_headers(const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* param_image,
                                            int header_type, uint8_t* dst, size_t capacity, size_t* main_header_size);
/* the same with orientation metadata (rotation in quarter turns clockwise, -1 = none; flip) as set by enc_metadata */
GPUJPEG_API size_t gpujpeg_amd_host_headers_md(const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* param_image,
                                               int header_type, int rotation, int flip, uint8_t* dst, size_t capacity, size_t* main_header_size);
/* the same with user Exif tags in the syntax of the encoder option enc_exif_tag ("<ID>:<type>=<value>" or "<name>=<value>");
 * any tag selects the Exif header like the option does */
GPUJPEG_API size_t gpujpeg_amd_host_headers_exif(const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* param_image,
                                                 int header_type, int rotation, int flip, const char* const* exif_tags, int exif_tag_count,
                                                 uint8_t* dst, size_t capacity, size_t* main_header_size);
/* Host-only: decode a BMP / TGA / PNG / GIF file (by extension) with the readers behind gpujpeg_image_load_from_file into caller
 * memory: interleaved 8-bit channels, top-down. dst == NULL: header only. Returns 0, -1 on error or when capacity is too small. */
GPUJPEG_API int gpujpeg_amd_read_raster_file(const char* filename, uint8_t* dst, size_t capacity, int* width, int* height, int* channels);
/* Host-only: geometry summary for the adjusted parameters: out[0] segment_count, [1] block_count, [2] restart interval,
 * [3] blocks per MCU, [4 + 4*c ..] per component data_width, data_height, segment_count, type */
GPUJPEG_API int gpujpeg_amd_host_geometry(const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* param_image, int out[20]);

/* Host-only: builds both decode-table layouts for one DHT table (bits[1..16] = codes per length, vals = symbols) exactly as
 * gpujpeg_decoder_decode does and reports 0 = accepted, -1 = rejected (over-subscribed or more than 256 symbols). Lets the CPU
 * test-suite feed hostile tables to the builders without a GPU. */
GPUJPEG_API int gpujpeg_amd_host_huffman_table_check(const uint8_t bits[17], const uint8_t* vals, int is_ac);

/* per-kernel durations (ms, hipEvents on the coder's stream) of the last call made with perf_stats != 0:
 * encoder: [0] preprocess, [1] DCT+quant (fused path: preprocess included), [2] k_huffman or k_encode_*, [3] k_gather (behind k_huffman:
 *          k_scan_segments), [4] k_assemble (behind k_huffman only)
 * decoder: [0] entropy decoder, [1] IDCT (fused path: postprocess included), [2] postprocess, [3] marker scan (k_markers; 0 when the host walked the stream) */
GPUJPEG_API int gpujpeg_amd_encoder_get_kernel_times(struct gpujpeg_encoder* encoder, float ms[8]);
GPUJPEG_API int gpujpeg_amd_decoder_get_kernel_times(struct gpujpeg_decoder* decoder, float ms[8]);

/* ---- frame batches: many frames of ONE geometry behind one set of kernel launches --------------------------------------------------
 * The libgpujpeg API codes a frame per call. An HD frame is 135 workgroups of the encoder kernel on a device with 1024 places for them,
 * and a call is 2 (encode) or 4 (decode) dependent launches: frame-at-a-time calls cannot fill an MI355X with small frames however many
 * coders run side by side. These two calls take `count` frames that share parameters (encoder) or the header (decoder) and launch every
 * kernel once per chunk of frames (the frame is a grid dimension). Results are identical, byte for byte, to the frame-at-a-time calls;
 * configurations (restart interval 0, flip, channel remap, APP13 index) or streams (another header, damaged markers) the batched kernels do
 * not cover are coded frame by frame inside the call.
 *
 * gpujpeg_amd_encoder_encode_batch: frame f lies at frames + f * frame_stride (device memory = GPU_IMAGE semantics, or host memory: copied);
 *   images_compressed[f] / images_compressed_size[f] receive every frame's stream -- in device memory with enc_opt_out=device, else in
 *   pinned / pageable host memory -- owned by the encoder and valid until its next call. The streams lie a constant number of bytes apart
 *   (images_compressed[1] - images_compressed[0], a multiple of 16), so the pointers can go straight into gpujpeg_amd_decoder_decode_batch.
 *   perf_stats / get_stats are not kept for batch calls.
 * gpujpeg_amd_decoder_decode_batch: stream f lies at streams + f * stream_stride and has sizes[f] bytes (device or host memory); frame f's
 *   pixels go to output + f * output_stride (device memory) in the format set with gpujpeg_decoder_set_output_format. All streams must
 *   decode to the same image parameters; returns 0 when every frame was decoded. */
GPUJPEG_API int gpujpeg_amd_encoder_encode_batch(struct gpujpeg_encoder* encoder, const struct gpujpeg_parameters* param,
                                                 const struct gpujpeg_image_parameters* param_image, const uint8_t* frames, size_t frame_stride,
                                                 int count, uint8_t** images_compressed, size_t* images_compressed_size);
/* The same for frames / streams / destinations that are separate buffers (device or host memory, mixed if need be): buffers that lie a constant
 * distance apart are coded where they are, others are gathered into (scattered from) a staging buffer with one copy per frame on the coder's
 * stream. frame_bytes: room in every destination (>= the decoded frame). */
GPUJPEG_API int gpujpeg_amd_encoder_encode_batch_ptrs(struct gpujpeg_encoder* encoder, const struct gpujpeg_parameters* param,
                                                      const struct gpujpeg_image_parameters* param_image, const uint8_t* const* frames, int count,
                                                      uint8_t** images_compressed, size_t* images_compressed_size);
GPUJPEG_API int gpujpeg_amd_decoder_decode_batch_ptrs(struct gpujpeg_decoder* decoder, const uint8_t* const* streams, const size_t* sizes, int count,
                                                      uint8_t* const* outputs, size_t frame_bytes, struct gpujpeg_image_parameters* param_image);
/* frames per set of launches at most (0 = the default: up to 256, fewer for large frames whose work buffers would exceed 6 / 8 GB); tests use it to cut
 * small batches into several chunks */
GPUJPEG_API void gpujpeg_amd_encoder_set_batch_chunk(struct gpujpeg_encoder* encoder, int frames);
GPUJPEG_API void gpujpeg_amd_decoder_set_batch_chunk(struct gpujpeg_decoder* decoder, int frames);
/* how the frames of the last batch call were coded: by the batched launches / frame by frame inside the call (tests, benchmarks) */
GPUJPEG_API int gpujpeg_amd_encoder_last_batch(struct gpujpeg_encoder* encoder, int* batched, int* single);
GPUJPEG_API int gpujpeg_amd_decoder_last_batch(struct gpujpeg_decoder* decoder, int* batched, int* single);
/* how the decode calls of this decoder ran so far (tests, benchmarks): [0] launched speculatively on the cached header of the previous frame,
 * [1] of those, without the marker scan's second launch (the token decoder derived its segment table from the scan's records itself),
 * [2] speculative launches whose stream turned out not to be what was assumed and that were decoded again the careful way */
GPUJPEG_API int gpujpeg_amd_decoder_get_path_counters(struct gpujpeg_decoder* decoder, long counters[3]);
GPUJPEG_API int gpujpeg_amd_decoder_decode_batch(struct gpujpeg_decoder* decoder, const uint8_t* streams, size_t stream_stride, const size_t* sizes,
                                                 int count, uint8_t* output, size_t output_stride, struct gpujpeg_image_parameters* param_image);

#ifdef __cplusplus
}
#endif
#endif
