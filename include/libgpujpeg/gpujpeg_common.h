/*
 * libgpujpeg common API -- MI355X-native implementation (HIP / gfx950 behind the unchanged C API).
 *
 * Every declaration replaces the reference declaration of the same name in
 * libgpujpeg/gpujpeg_common.h (line numbers in brackets). Struct layouts and enumerator values are
 * ABI and therefore identical; behaviour notes state where the MI355X build differs.
 */
#ifndef GPUJPEG_COMMON_H
#define GPUJPEG_COMMON_H

#ifdef __cplusplus
#include <cstddef>
#include <cstdint>
#else
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#endif

#include "gpujpeg_type.h"

/* The reference exposes streams as cudaStream_t [49-52]. The type is an opaque pointer; pass a
 * hipStream_t through it (0 = default stream). */
#ifndef __DRIVER_TYPES_H__
struct CUstream_st;
typedef struct CUstream_st* cudaStream_t;
#endif

#if __cplusplus >= 201402L || __STDC_VERSION__ >= 202311L
#define GPUJPEG_DEPRECATED [[deprecated]]
#else
#define GPUJPEG_DEPRECATED __attribute__((deprecated))
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define GPUJPEG_API __attribute__((visibility("default")))

GPUJPEG_API int gpujpeg_version(void);                              /* [82] */
GPUJPEG_API const char* gpujpeg_version_to_string(int version);     /* [84] */
GPUJPEG_API double gpujpeg_get_time(void);                          /* [87-88] seconds */

#define GPUJPEG_MAX_DEVICE_COUNT 10

struct gpujpeg_device_info { /* [94-114]; cc_* carry the gfx major/minor (9,5 for gfx950) */
    int id;
    char name[256];
    int cc_major;
    int cc_minor;
    size_t global_memory;
    size_t constant_memory;
    size_t shared_memory;
    int register_count;
    int multiprocessor_count;
};

struct gpujpeg_devices_info { /* [117-123] */
    int device_count;
    struct gpujpeg_device_info device[GPUJPEG_MAX_DEVICE_COUNT];
};

GPUJPEG_API struct gpujpeg_devices_info gpujpeg_get_devices_info(void);   /* [130-131] */
GPUJPEG_API int gpujpeg_print_devices_info(void);                        /* [139-140] */
GPUJPEG_API int gpujpeg_init_device(int device_id, int flags);           /* [154-155] */

enum restart_int {
    RESTART_AUTO = -1,
    RESTART_NONE = 0, /* reference: CPU Huffman; here: still on the GPU, one segment per scan */
};

enum verbosity {
    GPUJPEG_LL_QUIET = -1,
    GPUJPEG_LL_INFO = 0,
    GPUJPEG_LL_STATUS = 1,
    GPUJPEG_LL_VERBOSE = 2,
    GPUJPEG_LL_DEBUG = 3,
    GPUJPEG_LL_DEBUG2 = 4,
};

struct gpujpeg_parameters { /* [176-215] */
    int verbose;
    int perf_stats;
    int quality;          /* 0-100 */
    int restart_interval; /* MCUs per segment; see enum restart_int */
    int interleaved;      /* 1 = single scan with all components */
    int segment_info;     /* emit / use the APP13 segment index */
    int comp_count;       /* 0 = derive from the pixel format */
    struct gpujpeg_component_sampling_factor sampling_factor[GPUJPEG_MAX_COMPONENT_COUNT];
    enum gpujpeg_color_space color_space_internal;
};

GPUJPEG_API void gpujpeg_set_default_parameters(struct gpujpeg_parameters* param);   /* [224-225] */
GPUJPEG_API struct gpujpeg_parameters gpujpeg_default_parameters(void);              /* [231-232] */

typedef uint32_t gpujpeg_sampling_factor_t;
#define MK_SUBSAMPLING(comp1_factor_h, comp1_factor_v, comp2_factor_h, comp2_factor_v, comp3_factor_h, comp3_factor_v, \
                       comp4_factor_h, comp4_factor_v)                                                                 \
    ((comp1_factor_h) << 28U | (comp1_factor_v) << 24U | (comp2_factor_h) << 20U | (comp2_factor_v) << 16U |           \
     (comp3_factor_h) << 12U | (comp3_factor_v) << 8U | (comp4_factor_h) << 4U | (comp4_factor_v) << 0U)

#define GPUJPEG_SUBSAMPLING_UNKNOWN 0U
#define GPUJPEG_SUBSAMPLING_4444 MK_SUBSAMPLING(1, 1, 1, 1, 1, 1, 1, 1)
#define GPUJPEG_SUBSAMPLING_444 MK_SUBSAMPLING(1, 1, 1, 1, 1, 1, 0, 0)
#define GPUJPEG_SUBSAMPLING_440 MK_SUBSAMPLING(1, 2, 1, 1, 1, 1, 0, 0)
#define GPUJPEG_SUBSAMPLING_422 MK_SUBSAMPLING(2, 1, 1, 1, 1, 1, 0, 0)
#define GPUJPEG_SUBSAMPLING_420 MK_SUBSAMPLING(2, 2, 1, 1, 1, 1, 0, 0)
#define GPUJPEG_SUBSAMPLING_411 MK_SUBSAMPLING(4, 1, 1, 1, 1, 1, 0, 0)
#define GPUJPEG_SUBSAMPLING_410 MK_SUBSAMPLING(4, 2, 1, 1, 1, 1, 0, 0)
#define GPUJPEG_SUBSAMPLING_400 MK_SUBSAMPLING(1, 1, 0, 0, 0, 0, 0, 0)
#define GPUJPEG_SUBSAMPLING_442 MK_SUBSAMPLING(1, 2, 1, 2, 1, 1, 0, 0)
#define GPUJPEG_SUBSAMPLING_421 MK_SUBSAMPLING(2, 2, 2, 1, 1, 1, 0, 0)

GPUJPEG_API void gpujpeg_parameters_chroma_subsampling(struct gpujpeg_parameters* param,
                                                       gpujpeg_sampling_factor_t subsampling);   /* [260-262] */
GPUJPEG_API const char* gpujpeg_subsampling_get_name(int comp_count,
                                                     const struct gpujpeg_component_sampling_factor* sampling_factor); /* [268-269] */
GPUJPEG_API gpujpeg_sampling_factor_t gpujpeg_subsampling_from_name(const char* subsampling);   /* [276-277] */

struct gpujpeg_image_parameters { /* [283-294] */
    int width;
    int height;
    enum gpujpeg_color_space color_space;
    enum gpujpeg_pixel_format pixel_format;
    int width_padding; /* bytes appended to each row */
};

GPUJPEG_API void gpujpeg_image_set_default_parameters(struct gpujpeg_image_parameters* param);   /* [302-303] */
GPUJPEG_API struct gpujpeg_image_parameters gpujpeg_default_image_parameters(void);              /* [309-310] */

enum gpujpeg_image_file_format { /* [317-357] */
    GPUJPEG_IMAGE_FILE_UNKNOWN = 0,
    GPUJPEG_IMAGE_FILE_JPEG = 1,
    GPUJPEG_IMAGE_FILE_RAW = 2, /* every following format is raw */
    GPUJPEG_IMAGE_FILE_GRAY,
    GPUJPEG_IMAGE_FILE_RGB,
    GPUJPEG_IMAGE_FILE_RGBA,
    GPUJPEG_IMAGE_FILE_BMP,
    GPUJPEG_IMAGE_FILE_GIF,
    GPUJPEG_IMAGE_FILE_PNG,
    GPUJPEG_IMAGE_FILE_TGA,
    GPUJPEG_IMAGE_FILE_PGM,
    GPUJPEG_IMAGE_FILE_PPM,
    GPUJPEG_IMAGE_FILE_PNM,
    GPUJPEG_IMAGE_FILE_PAM,
    GPUJPEG_IMAGE_FILE_Y4M,
    GPUJPEG_IMAGE_FILE_YUV, /* every following format is YUV */
    GPUJPEG_IMAGE_FILE_YUVA,
    GPUJPEG_IMAGE_FILE_UYVY,
    GPUJPEG_IMAGE_FILE_I420,
    GPUJPEG_IMAGE_FILE_TST, /* synthetic test image described by its file name */
};

#define GPUJPEG_IMAGE_FORMAT_IS_RAW(format) ((format) >= GPUJPEG_IMAGE_FILE_RAW)

struct gpujpeg_duration_stats { /* [368-378] milliseconds, measured with hipEvents on the coder's stream */
    double duration_memory_to;
    double duration_memory_from;
    double duration_memory_map;
    double duration_memory_unmap;
    double duration_preprocessor;
    double duration_dct_quantization;
    double duration_huffman_coder;
    double duration_stream;
    double duration_in_gpu;
};

GPUJPEG_API enum gpujpeg_image_file_format gpujpeg_image_get_file_format(const char* filename);   /* [386-387] */
GPUJPEG_API void gpujpeg_set_device(int index);                                                    /* [394] */
GPUJPEG_API size_t gpujpeg_image_calculate_size(struct gpujpeg_image_parameters* param);           /* [402-403] */
/* [417-418] returns pinned host memory; release with gpujpeg_image_destroy() */
GPUJPEG_API int gpujpeg_image_load_from_file(const char* filename, uint8_t** image, size_t* image_size);
GPUJPEG_API int gpujpeg_image_save_to_file(const char* filename, const uint8_t* image, size_t image_size,
                                           const struct gpujpeg_image_parameters* param_image);   /* [440-442] */
GPUJPEG_API int gpujpeg_image_get_properties(const char* filename, struct gpujpeg_image_parameters* param_image,
                                             int file_exists);                                    /* [453-454] */
GPUJPEG_API int gpujpeg_image_destroy(uint8_t* image);                                             /* [462-463] */
GPUJPEG_API void gpujpeg_image_range_info(const char* filename, int width, int height,
                                          enum gpujpeg_pixel_format sampling_factor);             /* [473-474] */
GPUJPEG_API int gpujpeg_image_convert(const char* input, const char* output,
                                      struct gpujpeg_image_parameters param_image_from,
                                      struct gpujpeg_image_parameters param_image_to);            /* [488-490] defunct upstream */

/* ---- OpenGL interop [492-655]: not available on MI355X compute nodes. The symbols exist so that
 * callers link; they report "not compiled in" exactly like a reference build without BUILD_OPENGL. ---- */
struct gpujpeg_opengl_context;
GPUJPEG_API int gpujpeg_opengl_init(struct gpujpeg_opengl_context** ctx); /* returns -2 */
GPUJPEG_API void gpujpeg_opengl_destroy(struct gpujpeg_opengl_context*);
GPUJPEG_API int gpujpeg_opengl_texture_create(int width, int height, uint8_t* data);
GPUJPEG_API int gpujpeg_opengl_texture_set_data(int texture_id, uint8_t* data);
GPUJPEG_API int gpujpeg_opengl_texture_get_data(int texture_id, uint8_t* data, size_t* data_size);
GPUJPEG_API void gpujpeg_opengl_texture_destroy(int texture_id);

enum gpujpeg_opengl_texture_type { GPUJPEG_OPENGL_TEXTURE_READ = 1, GPUJPEG_OPENGL_TEXTURE_WRITE = 2 };

struct cudaGraphicsResource;
struct gpujpeg_opengl_texture { /* [565-599] layout kept for ABI */
    int texture_id;
    enum gpujpeg_opengl_texture_type texture_type;
    int texture_width;
    int texture_height;
    int texture_pbo_type;
    int texture_pbo_id;
    struct cudaGraphicsResource* texture_pbo_resource;
    void* texture_callback_param;
    void (*texture_callback_attach_opengl)(void* param);
    void (*texture_callback_detach_opengl)(void* param);
};

GPUJPEG_API struct gpujpeg_opengl_texture* gpujpeg_opengl_texture_register(int texture_id,
                                                                          enum gpujpeg_opengl_texture_type texture_type);
GPUJPEG_API void gpujpeg_opengl_texture_unregister(struct gpujpeg_opengl_texture* texture);
GPUJPEG_API uint8_t* gpujpeg_opengl_texture_map(struct gpujpeg_opengl_texture* texture, size_t* data_size);
GPUJPEG_API void gpujpeg_opengl_texture_unmap(struct gpujpeg_opengl_texture* texture);

/* ---- names <-> enums [657-691] ---- */
GPUJPEG_API const char* gpujpeg_color_space_get_name(enum gpujpeg_color_space color_space);
GPUJPEG_API enum gpujpeg_pixel_format gpujpeg_pixel_format_by_name(const char* name);
GPUJPEG_API enum gpujpeg_header_type gpujpeg_header_type_by_name(const char* name);
GPUJPEG_API const char* gpujpeg_header_type_get_name(enum gpujpeg_header_type header_type);
GPUJPEG_API void gpujpeg_print_pixel_formats();
GPUJPEG_API enum gpujpeg_color_space gpujpeg_color_space_by_name(const char* name);
GPUJPEG_API int gpujpeg_pixel_format_get_comp_count(enum gpujpeg_pixel_format pixel_format);
GPUJPEG_API const char* gpujpeg_pixel_format_get_name(enum gpujpeg_pixel_format pixel_format);
GPUJPEG_API int gpujpeg_pixel_format_is_planar(enum gpujpeg_pixel_format pixel_format);
GPUJPEG_API void gpujpeg_device_reset(void);
GPUJPEG_API const char* gpujpeg_orientation_get_name(struct gpujpeg_orientation orientation);

#ifdef __cplusplus
}
#endif

#endif /* GPUJPEG_COMMON_H */
