/*
 * libgpujpeg decoder API -- MI355X-native implementation.
 * Replaces the declarations of libgpujpeg/gpujpeg_decoder.h (reference line numbers in brackets).
 */
#ifndef GPUJPEG_DECODER_H
#define GPUJPEG_DECODER_H

#include "gpujpeg_common.h"
#include "gpujpeg_type.h"

#ifdef __cplusplus
extern "C" {
#endif

struct gpujpeg_decoder;

enum gpujpeg_decoder_output_type { /* [49-60] */
    GPUJPEG_DECODER_OUTPUT_INTERNAL_BUFFER,    /* pinned host buffer owned by the decoder */
    GPUJPEG_DECODER_OUTPUT_CUSTOM_BUFFER,      /* caller's host buffer */
    GPUJPEG_DECODER_OUTPUT_OPENGL_TEXTURE,     /* unsupported here */
    GPUJPEG_DECODER_OUTPUT_CUDA_BUFFER,        /* device buffer owned by the decoder */
    GPUJPEG_DECODER_OUTPUT_CUSTOM_CUDA_BUFFER, /* caller's device buffer */
};

struct gpujpeg_decoder_output { /* [66-84] */
    enum gpujpeg_decoder_output_type type;
    uint8_t* data;
    size_t data_size;
    struct gpujpeg_image_parameters param_image;
    struct gpujpeg_opengl_texture* texture;
    const struct gpujpeg_image_metadata* metadata;
};

struct gpujpeg_decoder_init_parameters { /* [90-97] */
    cudaStream_t stream;
    int verbose;
    bool perf_stats;
    bool ff_cs_itu601_is_709;
};

GPUJPEG_API void gpujpeg_decoder_output_set_default(struct gpujpeg_decoder_output* output);                          /* [105-106] */
GPUJPEG_API void gpujpeg_decoder_output_set_custom(struct gpujpeg_decoder_output* output, uint8_t* custom_buffer);   /* [115-116] */
GPUJPEG_API void gpujpeg_decoder_output_set_texture(struct gpujpeg_decoder_output* output,
                                                    struct gpujpeg_opengl_texture* texture);                         /* [124-125] */
GPUJPEG_API void gpujpeg_decoder_output_set_cuda_buffer(struct gpujpeg_decoder_output* output);                      /* [132-133] */
GPUJPEG_API void gpujpeg_decoder_output_set_custom_cuda(struct gpujpeg_decoder_output* output,
                                                        uint8_t* d_custom_buffer);                                   /* [142-143] */

GPUJPEG_API struct gpujpeg_decoder* gpujpeg_decoder_create(cudaStream_t stream);                                     /* [152-153] */
GPUJPEG_API struct gpujpeg_decoder_init_parameters gpujpeg_decoder_default_init_parameters(void);                    /* [155-156] */
GPUJPEG_API struct gpujpeg_decoder* gpujpeg_decoder_create_with_params(const struct gpujpeg_decoder_init_parameters* params); /* [167-168] */
GPUJPEG_API int gpujpeg_decoder_init(struct gpujpeg_decoder* decoder, const struct gpujpeg_parameters* param,
                                     const struct gpujpeg_image_parameters* param_image);                            /* [187-188] */

/* [201-202] image may also be a device pointer (MI355X extension mirroring the encoder's pointer detection):
 * the header is then fetched to the host and the entropy-coded data are used in place. */
GPUJPEG_API int gpujpeg_decoder_decode(struct gpujpeg_decoder* decoder, uint8_t* image, size_t image_size,
                                       struct gpujpeg_decoder_output* output);

GPUJPEG_DEPRECATED GPUJPEG_API int gpujpeg_decoder_get_stats(struct gpujpeg_decoder* decoder,
                                                             struct gpujpeg_duration_stats* stats);                  /* [214-215] */
GPUJPEG_API int gpujpeg_decoder_destroy(struct gpujpeg_decoder* decoder);                                            /* [223-224] */

/* pixel-format placeholders accepted by gpujpeg_decoder_set_output_format() [233-245] */
#define GPUJPEG_PIXFMT_AUTODETECT ((enum gpujpeg_pixel_format)(GPUJPEG_PIXFMT_NONE - 1))
#define GPUJPEG_PIXFMT_NO_ALPHA ((enum gpujpeg_pixel_format)(GPUJPEG_PIXFMT_NONE - 2))
#define GPUJPEG_PIXFMT_STD ((enum gpujpeg_pixel_format)(GPUJPEG_PIXFMT_NONE - 3))
#define GPUJPEG_PIXFMT_NATIVE ((enum gpujpeg_pixel_format)(GPUJPEG_PIXFMT_NONE - 4))
#define GPUJPEG_CS_DEFAULT ((enum gpujpeg_color_space)(GPUJPEG_NONE - 1))

GPUJPEG_API void gpujpeg_decoder_set_output_format(struct gpujpeg_decoder* decoder, enum gpujpeg_color_space color_space,
                                                   enum gpujpeg_pixel_format pixel_format);                          /* [262-265] */

enum {
    GPUJPEG_COUNT_SEG_COUNT_REQ = 1 << 0,
};

struct gpujpeg_image_info { /* [270-283] */
    union {
        struct {
            struct gpujpeg_image_parameters param_image;
            struct gpujpeg_parameters param;
            int segment_count;
            enum gpujpeg_header_type header_type;
            const char* comment;
            struct gpujpeg_image_metadata metadata;
        };
        char reserved[512];
    };
};

GPUJPEG_API int gpujpeg_decoder_get_image_info2(uint8_t* image, size_t image_size, struct gpujpeg_image_info* info,
                                                int verbose, unsigned flags);                                        /* [287-288] */
GPUJPEG_API int gpujpeg_decoder_get_image_info(uint8_t* image, size_t image_size, struct gpujpeg_image_parameters* param_image,
                                               struct gpujpeg_parameters* param, int* segment_count);               /* [290-291] */

#define GPUJPEG_DEC_OPT_TGA_RLE_BOOL "dec_opt_tga_rle"
#define GPUJPEG_DEC_OPT_FLIPPED_BOOL "dec_opt_flipped"
#define GPUJPEG_DEC_OPT_CHANNEL_REMAP "dec_opt_channel_remap"
#define GPUJPEG_DEC_OPT_ALIGNMENT_BYTES_INT "dec_opt_alignment_bytes"

GPUJPEG_API int gpujpeg_decoder_set_option(struct gpujpeg_decoder* decoder, const char* opt, const char* val);       /* [310-311] */
GPUJPEG_API void gpujpeg_decoder_print_options();                                                                    /* [312-313] */

#ifdef __cplusplus
}
#endif

#endif /* GPUJPEG_DECODER_H */
