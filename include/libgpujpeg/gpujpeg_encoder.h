/*
 * libgpujpeg encoder API -- MI355X-native implementation.
 * Replaces the declarations of libgpujpeg/gpujpeg_encoder.h (reference line numbers in brackets).
 */
#ifndef GPUJPEG_ENCODER_H
#define GPUJPEG_ENCODER_H

#include "gpujpeg_common.h"
#include "gpujpeg_type.h"

#ifdef __cplusplus
extern "C" {
#endif

struct gpujpeg_encoder;

enum gpujpeg_encoder_input_type { /* [45-52] */
    GPUJPEG_ENCODER_INPUT_IMAGE,          /* host pointer (a device pointer is detected and used in place) */
    GPUJPEG_ENCODER_INPUT_OPENGL_TEXTURE, /* unsupported here: encode returns GPUJPEG_ERROR */
    GPUJPEG_ENCODER_INPUT_GPU_IMAGE,      /* device (HBM) pointer, used in place */
};

struct gpujpeg_encoder_input { /* [57-67] */
    enum gpujpeg_encoder_input_type type;
    uint8_t* image;
    struct gpujpeg_opengl_texture* texture;
};

GPUJPEG_API void gpujpeg_encoder_input_set_image(struct gpujpeg_encoder_input* input, uint8_t* image);       /* [77-78] */
GPUJPEG_API void gpujpeg_encoder_input_set_gpu_image(struct gpujpeg_encoder_input* input, uint8_t* image);   /* [88-89] */
GPUJPEG_API void gpujpeg_encoder_input_set_texture(struct gpujpeg_encoder_input* input,
                                                   struct gpujpeg_opengl_texture* texture);                  /* [99-100] */
GPUJPEG_API struct gpujpeg_encoder_input gpujpeg_encoder_input_image(uint8_t* image);                        /* [103-104] */
GPUJPEG_API struct gpujpeg_encoder_input gpujpeg_encoder_input_gpu_image(uint8_t* image);                    /* [106-107] */
GPUJPEG_API struct gpujpeg_encoder_input gpujpeg_encoder_input_texture(struct gpujpeg_opengl_texture* texture); /* [109-110] */

/* [118-119] stream: a hipStream_t passed through the opaque cudaStream_t typedef (0 = default stream).
 * The encoder is bound to the HIP device current on the calling thread. Returns NULL on failure. */
GPUJPEG_API struct gpujpeg_encoder* gpujpeg_encoder_create(cudaStream_t stream);

GPUJPEG_API size_t gpujpeg_encoder_max_pixels(struct gpujpeg_parameters* param, struct gpujpeg_image_parameters* param_image,
                                              enum gpujpeg_encoder_input_type image_input_type, size_t memory_size,
                                              int* max_pixels);                                               /* [132-133] */
GPUJPEG_API size_t gpujpeg_encoder_max_memory(struct gpujpeg_parameters* param, struct gpujpeg_image_parameters* param_image,
                                              enum gpujpeg_encoder_input_type image_input_type, int max_pixels); /* [145-146] */
GPUJPEG_API int gpujpeg_encoder_allocate(struct gpujpeg_encoder* encoder, const struct gpujpeg_parameters* param,
                                         const struct gpujpeg_image_parameters* param_image,
                                         enum gpujpeg_encoder_input_type image_input_type);                  /* [157-158] */

/* [173-176] *image_compressed is owned by the encoder and stays valid until the next encode call. */
GPUJPEG_API int gpujpeg_encoder_encode(struct gpujpeg_encoder* encoder, const struct gpujpeg_parameters* param,
                                       const struct gpujpeg_image_parameters* param_image,
                                       const struct gpujpeg_encoder_input* input, uint8_t** image_compressed,
                                       size_t* image_compressed_size);

GPUJPEG_DEPRECATED GPUJPEG_API int gpujpeg_encoder_get_stats(struct gpujpeg_encoder* encoder,
                                                             struct gpujpeg_duration_stats* stats);          /* [188-189] */
GPUJPEG_DEPRECATED GPUJPEG_API void gpujpeg_encoder_set_jpeg_header(struct gpujpeg_encoder* encoder,
                                                                    enum gpujpeg_header_type header_type);   /* [199-200] */
GPUJPEG_API int gpujpeg_encoder_suggest_restart_interval(const struct gpujpeg_image_parameters* param_image,
                                                         gpujpeg_sampling_factor_t subsampling, bool interleaved,
                                                         int verbose);                                        /* [207-209] */

#define GPUJPEG_ENCODER_OPT_OUT_PINNED "enc_out_pinned" /* deprecated spelling */
#define GPUJPEG_ENC_OPT_OUT "enc_opt_out"
#define GPUJPEG_ENC_OUT_VAL_PAGEABLE "enc_out_val_pageable"
#define GPUJPEG_ENC_OUT_VAL_PINNED "enc_out_val_pinned"
/* MI355X extension: keep the finished JPEG in HBM; *image_compressed is then a device pointer */
#define GPUJPEG_ENC_OUT_VAL_DEVICE "enc_out_val_device"

#define GPUJPEG_ENC_OPT_HDR "enc_hdr"
#define GPUJPEG_ENC_HDR_VAL_JFIF "JFIF"
#define GPUJPEG_ENC_HDR_VAL_EXIF "Exif"
#define GPUJPEG_ENC_HDR_VAL_ADOBE "Adobe"
#define GPUJPEG_ENC_HDR_VAL_SPIFF "SPIFF"

#define GPUJPEG_ENC_OPT_FLIPPED_BOOL "enc_opt_flipped"
#define GPUJPEG_ENC_OPT_EXIF_TAG "enc_exif_tag"
#define GPUJPEG_ENC_OPT_METADATA "enc_metadata"
#define GPUJPEG_ENC_OPT_CHANNEL_REMAP "enc_opt_channel_remap"

GPUJPEG_API int gpujpeg_encoder_set_option(struct gpujpeg_encoder* encoder, const char* opt, const char* val);   /* [247-248] */
GPUJPEG_API void gpujpeg_encoder_print_options();                                                                /* [249-250] */
GPUJPEG_API int gpujpeg_encoder_destroy(struct gpujpeg_encoder* encoder);                                        /* [258-259] */

#ifdef __cplusplus
}
#endif

#endif /* GPUJPEG_ENCODER_H */
