/*
 * gj_oracle.h -- CPU restatement of the CESNET/GPUJPEG hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under gpujpeg_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Every stage cites the reference file:line it restates (paths relative to /root/reference).
 *
 * PARITY STATUS: the integer stages (colour transform, subsampling, geometry, Huffman, stream
 * format) are pinned against the reference's own host C code compiled into oracle/_ref
 * (see oracle/Makefile and tests/test_oracle_vs_ref.py). The float stages (forward DCT+quant,
 * dequant+IDCT) exist only as CUDA in the reference and the reference ships no golden vectors,
 * so their FMA-contraction pattern is "parity unpinned": we restate the arithmetic with the
 * explicit fusion map documented in DESIGN.md section 3.
 */
#ifndef GJ_ORACLE_H
#define GJ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* values equal the reference enums (libgpujpeg/gpujpeg_type.h:85-134) */
enum { GJO_CS_NONE = 0, GJO_CS_RGB = 1, GJO_CS_BT601 = 2, GJO_CS_BT601_256 = 3, GJO_CS_BT709 = 4, GJO_CS_YUV = 5 };
enum { GJO_PF_U8 = 0, GJO_PF_444_P012 = 1, GJO_PF_444_P0P1P2 = 2, GJO_PF_422_P1020 = 3,
       GJO_PF_422_P0P1P2 = 4, GJO_PF_420_P0P1P2 = 5, GJO_PF_4444_P0123 = 6 };
enum { GJO_LUMA = 0, GJO_CHROMA = 1 };
#define GJO_MAX_COMP 4

typedef struct gjo_comp {
    int type;                       /* GJO_LUMA / GJO_CHROMA */
    int h, v;                       /* sampling factors */
    int width, height;              /* real component size */
    int data_width, data_height;    /* padded to MCU multiples */
    int mcu_size_x, mcu_size_y, mcu_size;
    int mcu_count_x, mcu_count_y, mcu_count;
    int segment_mcu_count, segment_count;
    size_t data_offset;             /* offset (in samples) of this component's plane / coefficient plane */
} gjo_comp;

typedef struct gjo_image {
    /* ---- inputs ---- */
    int width, height, width_padding;
    int pixel_format;               /* raw pixel format */
    int color_space;                /* raw colour space */
    int comp_count;
    int samp_h[GJO_MAX_COMP], samp_v[GJO_MAX_COMP];
    int interleaved;
    int restart_interval;           /* >= 0 (0 = one segment per scan) */
    int quality;
    int color_space_internal;
    int segment_info;               /* emit APP13 segment index */
    int header_type;                /* 0 default, 1 JFIF, 2 SPIFF, 4 Adobe */
    /* ---- derived by gjo_image_init ---- */
    int max_h, max_v;
    gjo_comp comp[GJO_MAX_COMP];
    size_t data_size;               /* total samples in all padded planes */
    size_t raw_size;
    int mcu_count, segment_count, segment_mcu_count, block_count;
    int scan_count;
} gjo_image;

typedef struct gjo_segment {
    int scan_index, scan_segment_index, mcu_count;
} gjo_segment;

/* ---- parameter helpers (src/gpujpeg_encoder.c:291-346, src/gpujpeg_common.c:140-151,1180-1204) ---- */
int    gjo_pixfmt_comp_count(int pixel_format);
int    gjo_pixfmt_unit_size(int pixel_format);
void   gjo_pixfmt_sampling(int pixel_format, int h[GJO_MAX_COMP], int v[GJO_MAX_COMP]);
size_t gjo_raw_size(int width, int height, int width_padding, int pixel_format);
int    gjo_suggest_restart_interval(int width, int height, int pixel_format, int subsampling_is_444, int interleaved);
/* fill comp_count/sampling from the pixel format when comp_count==0 and resolve restart_interval<0 */
void   gjo_adjust_encoder_params(gjo_image* img);

/* ---- geometry (src/gpujpeg_common.c:675-870) ---- */
int  gjo_image_init(gjo_image* img);
void gjo_segment_get(const gjo_image* img, int segment_index, gjo_segment* seg);
/* offset (in coefficients) of the k-th 8x8 block of a segment, plus its component (src/gpujpeg_common.c:1040-1085) */
size_t gjo_segment_block(const gjo_image* img, const gjo_segment* seg, int k, int* comp);
int    gjo_segment_block_count(const gjo_image* img, const gjo_segment* seg);

/* ---- tables (src/gpujpeg_table.c:35-129,190-306) ---- */
void gjo_quant_table(int type, int quality, uint8_t raw_zigzag[64], float fwd_transposed[64], uint16_t inv_natural[64]);
/* index: 0 = luma DC, 1 = luma AC, 2 = chroma DC, 3 = chroma AC */
void gjo_huffman_spec(int index, const uint8_t** bits17, const uint8_t** vals, int* nvals);
extern const int gjo_zigzag[64];   /* zig-zag position -> natural index */

/* ---- encoder stages ---- */
/* raw pixels -> padded planar components (src/gpujpeg_preprocessor.cu:49-202, src/gpujpeg_colorspace.h) */
void gjo_preprocess(const gjo_image* img, const uint8_t* raw, uint8_t* planes);
/* padded planes -> quantised coefficients, 64 per block, blocks raster per component (src/gpujpeg_dct_gpu.cu:121-295) */
void gjo_fdct_quant(const gjo_image* img, const uint8_t* planes, int16_t* coefs);
void gjo_fdct_quant_block(const uint8_t* src, int stride, const float fwd_transposed[64], int16_t out[64]);
/* one segment -> entropy coded bytes incl. padding, WITHOUT trailing RST (src/gpujpeg_huffman_gpu_encoder.cu:139-294,417-503) */
size_t gjo_huffman_encode_segment(const gjo_image* img, const int16_t* coefs, int segment_index, uint8_t* out);
/* header bytes up to (not including) the first SOS (src/gpujpeg_writer.c:452-520) */
size_t gjo_write_header(const gjo_image* img, uint8_t* out);
size_t gjo_write_scan_header(const gjo_image* img, int scan_index, uint8_t* out);
/* complete JPEG from coefficients / from raw pixels; returns size, 0 on overflow */
size_t gjo_encode_from_coefs(const gjo_image* img, const int16_t* coefs, uint8_t* out, size_t cap);
size_t gjo_encode(gjo_image* img, const uint8_t* raw, uint8_t* out, size_t cap);

/* ---- decoder stages ---- */
typedef struct gjo_stream {
    gjo_image img;                          /* geometry of the coded image (pixel_format/color_space = requested output) */
    uint8_t  qraw[4][64];                   /* DQT tables, zig-zag order */
    uint16_t qinv[4][64];                   /* natural order */
    int      qmap[GJO_MAX_COMP];
    uint8_t  hbits[4][2][17];               /* [Th][Tc] */
    uint8_t  hvals[4][2][256];
    int      hmap[GJO_MAX_COMP][2];         /* component -> table id for DC, AC */
    uint8_t  comp_id[GJO_MAX_COMP];
    int      seg_count;
    size_t*  seg_offset;                    /* byte offset of each segment's entropy data inside the JPEG */
    size_t*  seg_size;
    int*     seg_scan;                      /* scan index of each segment */
    int*     seg_index_in_scan;
} gjo_stream;

/* parse markers, split scans at RSTn (src/gpujpeg_reader.c:682-1155,1257-1372,1620-1707). req_* < 0 => defaults */
int  gjo_parse(const uint8_t* jpeg, size_t size, int req_pixel_format, int req_color_space, gjo_stream* s);
void gjo_stream_free(gjo_stream* s);
/* entropy decode all segments into coefficient planes (src/gpujpeg_huffman_cpu_decoder.c:245-372) */
int  gjo_huffman_decode(const gjo_stream* s, const uint8_t* jpeg, int16_t* coefs);
/* dequant + IDCT + level shift + clamp (src/gpujpeg_dct_gpu.cu:312-366,472-618) */
void gjo_idct(const gjo_stream* s, const int16_t* coefs, uint8_t* planes);
/* 1 (default): the fused multiply-adds of the pinned fusion map; 0: every one of them as multiply, then add (contraction off) */
void gjo_set_fma(int on);
void gjo_idct_block(const int16_t in[64], const uint16_t q_natural[64], uint8_t* dst, int stride);
/* padded planes -> raw pixels (src/gpujpeg_postprocessor.cu:49-217, src/gpujpeg_colorspace.h) */
void gjo_postprocess(const gjo_image* img, const uint8_t* planes, uint8_t* raw);
/* complete decode; out must hold gjo_raw_size(...) of the requested output format. Returns 0 on success */
int  gjo_decode(const uint8_t* jpeg, size_t size, int req_pixel_format, int req_color_space,
                uint8_t* out, size_t cap, gjo_image* info);

/* options acting on the planes / the raw image (src/gpujpeg_preprocessor.cu:455-559, src/gpujpeg_encoder.c:661-699) */
void gjo_flip_planes(const gjo_image* img, uint8_t* planes);
unsigned gjo_parse_channel_remap(const char* val);
int  gjo_channel_remap(const gjo_image* img, uint8_t* raw, unsigned channel_remap);

/* colour transform of one pixel, exposed for exhaustive tests (src/gpujpeg_colorspace.h:64-102,216-430) */
void gjo_color_transform(int cs_from, int cs_to, uint8_t c[3]);

/* .tst synthetic image generator semantics (src/utils/image_delegate.c:562-603) */
void gjo_fill_noise(uint8_t* dst, size_t n, unsigned seed);
void gjo_fill_gradient(uint8_t* dst, int width, int height, int bytes_per_pixel);

#ifdef __cplusplus
}
#endif
#endif
