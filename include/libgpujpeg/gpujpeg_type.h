/*
 * libgpujpeg public types -- MI355X-native implementation.
 * ABI contract: enumerator values, macro values and struct layouts equal the reference's
 * libgpujpeg/gpujpeg_type.h:50-160 so that existing callers link unchanged.
 */
#ifndef GPUJPEG_TYPE_H
#define GPUJPEG_TYPE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef _MSC_VER
#define ATTRIBUTE_UNUSED __attribute__((unused))
#else
#define ATTRIBUTE_UNUSED
#endif

#define GPUJPEG_MAX_COMPONENT_COUNT 4

/* gpujpeg_init_device() flags */
#define GPUJPEG_INIT_DEV_VERBOSE 1
#define GPUJPEG_OPENGL_INTEROPERABILITY 2
#define GPUJPEG_VERBOSE GPUJPEG_INIT_DEV_VERBOSE /* deprecated alias */

#define GPUJPEG_MAX_SEGMENT_INFO_HEADER_COUNT 100

/* return codes */
#define GPUJPEG_NOERR 0
#define GPUJPEG_ERROR (-1)
#define GPUJPEG_ERR_RESTART_CHANGE (-2)

#define GPUJPEG_VAL_TRUE "1"
#define GPUJPEG_VAL_FALSE "0"

enum gpujpeg_color_space {
    GPUJPEG_NONE = 0,
    GPUJPEG_RGB = 1,
    GPUJPEG_YCBCR_BT601 = 2,         /* limited range */
    GPUJPEG_YCBCR_BT601_256LVLS = 3, /* full range = what JFIF carries */
    GPUJPEG_YCBCR_JPEG = GPUJPEG_YCBCR_BT601_256LVLS,
    GPUJPEG_YCBCR_BT709 = 4,         /* limited range */
    GPUJPEG_YCBCR = GPUJPEG_YCBCR_BT709,
    GPUJPEG_YUV = 5                  /* deprecated in the reference; accepted, converted like the reference does */
};

enum gpujpeg_header_type {
    GPUJPEG_HEADER_DEFAULT = 0, /* JFIF for YCbCr-JPEG, Adobe for RGB, SPIFF otherwise */
    GPUJPEG_HEADER_JFIF = 1 << 0,
    GPUJPEG_HEADER_SPIFF = 1 << 1,
    GPUJPEG_HEADER_ADOBE = 1 << 2,
    GPUJPEG_HEADER_EXIF = 1 << 3,
};

enum gpujpeg_pixel_format {
    GPUJPEG_PIXFMT_NONE = -1,
    GPUJPEG_U8 = 0,            /* 1 component */
    GPUJPEG_444_U8_P012 = 1,   /* packed c0 c1 c2 */
    GPUJPEG_444_U8_P0P1P2 = 2, /* planar 4:4:4 */
    GPUJPEG_422_U8_P1020 = 3,  /* packed c1 c0 c2 c0 (UYVY) */
    GPUJPEG_422_U8_P0P1P2 = 4, /* planar 4:2:2 */
    GPUJPEG_420_U8_P0P1P2 = 5, /* planar 4:2:0 */
    GPUJPEG_4444_U8_P0123 = 6, /* packed, 4th byte alpha or unused */
};

struct gpujpeg_component_sampling_factor {
    uint8_t horizontal;
    uint8_t vertical;
};

enum {
    GPUJPEG_METADATA_ORIENTATION,
    GPUJPEG_METADATA_COUNT,
};

struct gpujpeg_orientation { /* SPIFF semantics */
    unsigned rotation : 2;   /* multiples of 90 degrees clockwise */
    unsigned flip : 1;       /* mirrored left-right after rotation */
};

struct gpujpeg_image_metadata {
    struct {
        union {
            struct gpujpeg_orientation orient;
        };
        unsigned set : 1;
    } vals[GPUJPEG_METADATA_COUNT];
};

#ifdef __cplusplus
}
#endif

#endif /* GPUJPEG_TYPE_H */
