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

/* per-kernel durations (ms, hipEvents on the coder's stream) of the last call made with perf_stats != 0:
 * encoder: [0] preprocess, [1] DCT+quant (fused path: preprocess included), [2] k_huffman, [3] k_scan_segments, [4] k_assemble
 * decoder: [0] k_huffman_decode, [1] IDCT (fused path: postprocess included), [2] postprocess */
GPUJPEG_API int gpujpeg_amd_encoder_get_kernel_times(struct gpujpeg_encoder* encoder, float ms[8]);
GPUJPEG_API int gpujpeg_amd_decoder_get_kernel_times(struct gpujpeg_decoder* decoder, float ms[8]);

#ifdef __cplusplus
}
#endif
#endif
