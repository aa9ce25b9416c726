/*
 * ref_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * The reference's host C files (driver, geometry, tables, JFIF writer/reader, CPU Huffman coders)
 * are compiled unmodified from /root/reference into oracle/_ref/libgpujpeg_ref*.so against
 * stub/cuda_runtime.h, and so are three of its five CUDA modules -- src/gpujpeg_preprocessor.cu,
 * gpujpeg_dct_gpu.cu and gpujpeg_postprocessor.cu, i.e. colour transforms, sub/upsampling, fDCT+quantisation
 * and dequantisation+IDCT -- which run on the CPU under the cudaemu execution model (oracle/cudaemu) or on
 * the GPU through hipcc (oracle/hipstub). The two Huffman GPU modules
 * (warp-32 intrinsics) run under cudaemu's warp mode in the CPU builds (round 6: GJREF_EMU_HUFFMAN, all five CUDA modules are the reference's);
 * the hipcc build for gfx950 (wave64) and the timing builds still take this file's stand-ins for their entry points: the reference's own CPU
 * Huffman decoder and the restated segment coder (whose bytes gjref_reencode_cpu_huffman checks against the
 * reference's CPU Huffman encoder). With GJREF_RESTATED_STAGES defined the three arithmetic modules are
 * replaced by the restatement of gj_oracle.c as well (the round-1 build, kept for bisecting a mismatch).
 *
 * It also cross-checks our restated geometry against the reference's on every call (abort on mismatch).
 */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gpujpeg_common_internal.h"
#include "gpujpeg_dct_gpu.h"
#include "gpujpeg_decoder_internal.h"
#include "gpujpeg_encoder_internal.h"
#include "gpujpeg_huffman_cpu_decoder.h"
#include "gpujpeg_huffman_cpu_encoder.h"
#include "gpujpeg_huffman_gpu_decoder.h"
#include "gpujpeg_huffman_gpu_encoder.h"
#include "gpujpeg_marker.h"
#include "gpujpeg_postprocessor.h"
#include "gpujpeg_preprocessor.h"

#include "gj_oracle.h"

#define SHIM_CHECK(cond)                                                                               \
    do {                                                                                               \
        if (!(cond)) {                                                                                 \
            fprintf(stderr, "[ref_shim] geometry mismatch vs reference: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
            abort();                                                                                   \
        }                                                                                              \
    } while (0)

/* build the oracle's view of the coder state and verify it equals the reference's (common.c:629-1106) */
static void image_from_coder(const struct gpujpeg_coder* coder, gjo_image* img)
{
    memset(img, 0, sizeof *img);
    img->width = coder->param_image.width;
    img->height = coder->param_image.height;
    img->width_padding = coder->param_image.width_padding;
    img->pixel_format = coder->param_image.pixel_format;
    img->color_space = coder->param_image.color_space;
    img->comp_count = coder->param.comp_count;
    for (int i = 0; i < coder->param.comp_count; i++) {
        img->samp_h[i] = coder->param.sampling_factor[i].horizontal;
        img->samp_v[i] = coder->param.sampling_factor[i].vertical;
    }
    img->interleaved = coder->param.interleaved;
    img->restart_interval = coder->param.restart_interval;
    img->quality = coder->param.quality;
    img->color_space_internal = coder->param.color_space_internal;
    img->segment_info = coder->param.segment_info;
    SHIM_CHECK(gjo_image_init(img) == 0);
    SHIM_CHECK(img->data_size == coder->data_size);
    SHIM_CHECK(img->segment_count == coder->segment_count);
    SHIM_CHECK(img->block_count == coder->block_count);
    SHIM_CHECK(img->raw_size == coder->data_raw_size);
    for (int c = 0; c < img->comp_count; c++) {
        const struct gpujpeg_component* k = &coder->component[c];
        SHIM_CHECK(img->comp[c].type == (int)k->type);
        SHIM_CHECK(img->comp[c].width == k->width && img->comp[c].height == k->height);
        SHIM_CHECK(img->comp[c].data_width == k->data_width && img->comp[c].data_height == k->data_height);
        SHIM_CHECK(img->comp[c].mcu_size_x == k->mcu_size_x && img->comp[c].mcu_size_y == k->mcu_size_y);
        SHIM_CHECK(img->comp[c].mcu_count_x == k->mcu_count_x && img->comp[c].mcu_count == k->mcu_count);
        SHIM_CHECK(img->comp[c].segment_count == k->segment_count && img->comp[c].segment_mcu_count == k->segment_mcu_count);
        SHIM_CHECK(img->comp[c].data_offset == k->data_quantized_index);
    }
    /* block list parity: our arithmetic block addressing vs the reference's explicit list (common.c:1040-1085) */
    for (int s = 0; s < coder->segment_count; s++) {
        gjo_segment seg;
        gjo_segment_get(img, s, &seg);
        const struct gpujpeg_segment* rs = &coder->segment[s];
        SHIM_CHECK(seg.scan_index == rs->scan_index && seg.scan_segment_index == rs->scan_segment_index && seg.mcu_count == rs->mcu_count);
        SHIM_CHECK(gjo_segment_block_count(img, &seg) == rs->block_count);
        for (int k = 0; k < rs->block_count; k++) {
            int comp;
            size_t off = gjo_segment_block(img, &seg, k, &comp);
            uint64_t packed = coder->block_list[rs->block_index_list_begin + k];
            SHIM_CHECK((uint64_t)off == (packed >> 8));
            SHIM_CHECK(comp == (int)(packed & 0x7f));
            SHIM_CHECK(((packed & 0x80) != 0) == (img->comp[comp].type == GJO_CHROMA));
        }
    }
}

#ifdef GJREF_RESTATED_STAGES
/* ---- preprocessor (reference: src/gpujpeg_preprocessor.cu:316,563) ---- */
int gpujpeg_preprocessor_encoder_init(struct gpujpeg_coder* coder)
{
    coder->preprocessor.kernel = NULL;
    return 0;
}

int gpujpeg_preprocessor_encode(struct gpujpeg_encoder* encoder)
{
    struct gpujpeg_coder* coder = &encoder->coder;
    gjo_image img;
    image_from_coder(coder, &img);
    /* order as in src/gpujpeg_preprocessor.cu:563-586: remap the raw image in place, convert, flip the planes */
    if (coder->preprocessor.channel_remap != 0 && gpujpeg_preprocessor_channel_remap(coder) != 0) return -1;
    gjo_preprocess(&img, coder->d_data_raw, coder->d_data);
    if (coder->preprocessor.flipped) return gpujpeg_preprocessor_flip_lines(coder);
    return 0;
}

int gpujpeg_preprocessor_channel_remap(struct gpujpeg_coder* coder)
{
    gjo_image img;
    image_from_coder(coder, &img);
    if (gjo_channel_remap(&img, coder->d_data_raw, coder->preprocessor.channel_remap) != 0) {
        fprintf(stderr, "[ref_shim] channel remap: wrong channel count or a pixel format whose pixels share samples (not restated)\n");
        return -1;
    }
    return 0;
}

int gpujpeg_preprocessor_flip_lines(struct gpujpeg_coder* coder)
{
    gjo_image img;
    image_from_coder(coder, &img);
    gjo_flip_planes(&img, coder->d_data);
    return 0;
}

/* ---- DCT (reference: src/gpujpeg_dct_gpu.cu:622,682) -- uses the REFERENCE's tables ---- */
int gpujpeg_dct_gpu(struct gpujpeg_encoder* encoder)
{
    struct gpujpeg_coder* coder = &encoder->coder;
    for (int c = 0; c < coder->param.comp_count; c++) {
        struct gpujpeg_component* k = &coder->component[c];
        const float* fwd = encoder->table_quantization[k->type].d_table_forward;
        int bw = k->data_width / 8, bh = k->data_height / 8;
        for (int by = 0; by < bh; by++)
            for (int bx = 0; bx < bw; bx++)
                gjo_fdct_quant_block(k->d_data + (size_t)by * 8 * k->data_width + bx * 8, k->data_width, fwd,
                                     k->d_data_quantized + ((size_t)by * bw + bx) * 64);
    }
    return 0;
}

int gpujpeg_idct_gpu(struct gpujpeg_decoder* decoder)
{
    struct gpujpeg_coder* coder = &decoder->coder;
    for (int c = 0; c < coder->param.comp_count; c++) {
        struct gpujpeg_component* k = &coder->component[c];
        const uint16_t* q = decoder->table_quantization[decoder->comp_table_quantization_map[c]].d_table;
        int bw = k->data_width / 8, bh = k->data_height / 8;
        for (int by = 0; by < bh; by++)
            for (int bx = 0; bx < bw; bx++)
                gjo_idct_block(k->d_data_quantized + ((size_t)by * bw + bx) * 64, q,
                               k->d_data + (size_t)by * 8 * k->data_width + bx * 8, k->data_width);
    }
    return 0;
}

#endif /* GJREF_RESTATED_STAGES */

#ifndef GJREF_EMU_HUFFMAN /* (the cudaemu builds link the reference's own two Huffman GPU modules, oracle/Makefile EMU_WARP_CU: no stand-ins there) */
/* ---- Huffman "GPU" encoder (reference: src/gpujpeg_huffman_gpu_encoder.cu:973,1072) ---- */
struct gpujpeg_huffman_gpu_encoder { int unused; };

struct gpujpeg_huffman_gpu_encoder* gpujpeg_huffman_gpu_encoder_create(const struct gpujpeg_encoder* encoder)
{
    (void)encoder;
    return (struct gpujpeg_huffman_gpu_encoder*)calloc(1, sizeof(struct gpujpeg_huffman_gpu_encoder));
}

void gpujpeg_huffman_gpu_encoder_destroy(struct gpujpeg_huffman_gpu_encoder* h) { free(h); }

int gpujpeg_huffman_gpu_encoder_encode(struct gpujpeg_encoder* encoder, struct gpujpeg_huffman_gpu_encoder* h, unsigned int* output_byte_count)
{
    (void)h;
    struct gpujpeg_coder* coder = &encoder->coder;
    gjo_image img;
    cudaStreamSynchronize(coder->stream);   /* hipstub build: the DCT kernels before us run on the GPU */
    image_from_coder(coder, &img);
    size_t out = 0;
    for (int s = 0; s < coder->segment_count; s++) {
        struct gpujpeg_segment* seg = &coder->d_segment[s];
        uint8_t* dst = coder->d_data_compressed + out;
        size_t n = gjo_huffman_encode_segment(&img, coder->d_data_quantized, s, dst);
        dst[n++] = 0xFF;                                              /* :491-493: every segment ends with RSTn */
        dst[n++] = (uint8_t)(GPUJPEG_MARKER_RST0 + (seg->scan_segment_index % 8));
        seg->data_compressed_index = out;
        seg->data_compressed_size = n;
        out += (n + 15) & ~(size_t)15;                                /* :590 16-byte granules */
        assert(out <= coder->data_compressed_allocated_size);
    }
    *output_byte_count = (unsigned)out;
    return 0;
}

/* ---- Huffman "GPU" decoder (reference: src/gpujpeg_huffman_gpu_decoder.cu:614,664): run the reference's CPU decoder ---- */
struct gpujpeg_huffman_gpu_decoder { int unused; };

struct gpujpeg_huffman_gpu_decoder* gpujpeg_huffman_gpu_decoder_init(void)
{
    return (struct gpujpeg_huffman_gpu_decoder*)calloc(1, sizeof(struct gpujpeg_huffman_gpu_decoder));
}

void gpujpeg_huffman_gpu_decoder_destroy(struct gpujpeg_huffman_gpu_decoder* h) { free(h); }

int gpujpeg_huffman_gpu_decoder_decode(struct gpujpeg_decoder* decoder)
{
    struct gpujpeg_coder* coder = &decoder->coder;
    if (coder->data_quantized == NULL && gpujpeg_coder_allocate_cpu_huffman_buf(coder) != 0) return -1;
    if (gpujpeg_huffman_cpu_decoder_decode(decoder) != 0) return -1;
    memcpy(coder->d_data_quantized, coder->data_quantized, coder->data_size * sizeof(int16_t));
    return 0;
}
#endif /* !GJREF_EMU_HUFFMAN */

#ifdef GJREF_RESTATED_STAGES
/* ---- postprocessor (reference: src/gpujpeg_postprocessor.cu:349,445) ---- */
int gpujpeg_postprocessor_decoder_init(struct gpujpeg_coder* coder)
{
    coder->preprocessor.kernel = NULL;
    return 0;
}

int gpujpeg_postprocessor_decode(struct gpujpeg_coder* coder, cudaStream_t stream)
{
    (void)stream;
    gjo_image img;
    image_from_coder(coder, &img);
    /* order as in src/gpujpeg_postprocessor.cu:445-496: flip the planes, convert, remap the raw image */
    if (coder->preprocessor.flipped) gpujpeg_preprocessor_flip_lines(coder);
    gjo_postprocess(&img, coder->d_data, coder->d_data_raw);
    if (coder->preprocessor.channel_remap != 0 && gpujpeg_preprocessor_channel_remap(coder) != 0) return -1;
    return 0;
}

#endif /* GJREF_RESTATED_STAGES */

/* OpenGL interop is not compiled in; one stray runtime symbol is referenced unconditionally */
int cudaGraphicsUnmapResources(int count, void* resources, cudaStream_t stream)
{
    (void)count; (void)resources; (void)stream;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Test hooks (not part of the reference API)
 * ---------------------------------------------------------------------------------------------- */

/* Re-encode the coefficients of the last gpujpeg_encoder_encode() call with the reference's CPU
 * Huffman coder (src/gpujpeg_huffman_cpu_encoder.c:297), which writes scan headers and RSTn itself.
 * The bytes must equal what the GPU-path stitching (src/gpujpeg_encoder.c:567-629) produced. */
int gjref_reencode_cpu_huffman(struct gpujpeg_encoder* encoder, uint8_t** out, size_t* size)
{
    struct gpujpeg_coder* coder = &encoder->coder;
    {   /* (the geometry cross-check the Huffman stand-in makes on every call: here for the builds that run the reference's own kernels) */
        gjo_image img;
        image_from_coder(coder, &img);
    }
    if (coder->data_quantized == NULL && gpujpeg_coder_allocate_cpu_huffman_buf(coder) != 0) return -1;
    memcpy(coder->data_quantized, coder->d_data_quantized, coder->data_size * sizeof(int16_t));
    encoder->writer->buffer_current = encoder->writer->buffer;
    gpujpeg_writer_write_header(encoder);
    if (gpujpeg_huffman_cpu_encoder_encode(encoder) != 0) return -1;
    gpujpeg_writer_emit_marker(encoder->writer, GPUJPEG_MARKER_EOI);
    *out = encoder->writer->buffer;
    *size = (size_t)(encoder->writer->buffer_current - encoder->writer->buffer);
    return 0;
}

/* expose the coefficient plane / component planes of the last call */
const int16_t* gjref_encoder_coefficients(struct gpujpeg_encoder* encoder, size_t* count)
{
    *count = encoder->coder.data_size;
    return encoder->coder.d_data_quantized;
}

const uint8_t* gjref_encoder_planes(struct gpujpeg_encoder* encoder, size_t* count)
{
    *count = encoder->coder.data_size;
    return encoder->coder.d_data;
}

const int16_t* gjref_decoder_coefficients(struct gpujpeg_decoder* decoder, size_t* count)
{
    *count = decoder->coder.data_size;
    return decoder->coder.d_data_quantized;
}

/* the reference's forward quantisation table as it would be uploaded (src/gpujpeg_table.c:103-129) */
void gjref_quant_tables(int type, int quality, uint8_t raw[64], float fwd[64], uint16_t inv[64])
{
    struct gpujpeg_table_quantization t;
    memset(&t, 0, sizeof t);
    t.d_table_forward = fwd;
    t.d_table = inv;
    gpujpeg_table_quantization_encoder_init(&t, (enum gpujpeg_component_type)type, quality);
    gpujpeg_table_quantization_decoder_init(&t, (enum gpujpeg_component_type)type, quality);
    memcpy(raw, t.table_raw, 64);
}

/* the reference's integer CPU IDCT (src/gpujpeg_dct_cpu.c:178-203, not parity-equal with the CUDA kernels) over the
 * coefficients of the last gpujpeg_decoder_decode() call, on a scratch copy; returns the seconds it took (bench.py cpu_baseline) */
#include <time.h>
#include "gpujpeg_dct_cpu.h"
double gjref_time_idct_cpu(struct gpujpeg_decoder* decoder)
{
    struct gpujpeg_coder* coder = &decoder->coder;
    if (coder->data_quantized == NULL && gpujpeg_coder_allocate_cpu_huffman_buf(coder) != 0) return -1.0;
    memcpy(coder->data_quantized, coder->d_data_quantized, coder->data_size * sizeof(int16_t));
    uint8_t* keep = malloc(coder->data_size); /* gpujpeg_idct_cpu writes the planes the caller may still read */
    if (!keep) return -1.0;
    memcpy(keep, coder->d_data, coder->data_size);
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    gpujpeg_idct_cpu(decoder);
    clock_gettime(CLOCK_MONOTONIC, &b);
    memcpy(coder->d_data, keep, coder->data_size);
    free(keep);
    return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}
