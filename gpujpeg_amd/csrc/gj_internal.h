/*
 * gj_internal.h -- host-side state of the MI355X libgpujpeg implementation (plain C11).
 * The host never sees HIP types: everything device-related goes through include/gj_hip.h.
 */
#ifndef GJ_INTERNAL_H
#define GJ_INTERNAL_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "gj_hip.h"
#include "libgpujpeg/gpujpeg_common.h"
#include "libgpujpeg/gpujpeg_decoder.h"
#include "libgpujpeg/gpujpeg_encoder.h"
#include "libgpujpeg/gpujpeg_type.h"
#include "libgpujpeg/gpujpeg_version.h"

/* ---- logging, same prefixes as the reference (src/gpujpeg_common_internal.h:131-150) ---- */
extern const char* gj_fg_red;
extern const char* gj_fg_yellow;
extern const char* gj_term_reset;
void gj_init_term_colors(void);
#define GJ_ERROR(...) (fprintf(stderr, "%s[GPUJPEG] [Error]%s ", gj_fg_red, gj_term_reset), fprintf(stderr, __VA_ARGS__))
#define GJ_WARN(...) (fprintf(stderr, "%s[GPUJPEG] [Warning]%s ", gj_fg_yellow, gj_term_reset), fprintf(stderr, __VA_ARGS__))
#define GJ_VERBOSE(level, ...) ((level) >= GPUJPEG_LL_VERBOSE ? (void)(fprintf(stderr, "[GPUJPEG] [Verbose] "), fprintf(stderr, __VA_ARGS__)) : (void)0)
#define GJ_DEBUG(level, ...) ((level) >= GPUJPEG_LL_DEBUG ? (void)(fprintf(stderr, "[GPUJPEG] [Debug] "), fprintf(stderr, __VA_ARGS__)) : (void)0)

enum { GJ_LUMA = 0, GJ_CHROMA = 1 };

/* ---- pixel formats (src/gpujpeg_common.c:130-151) ---- */
int gj_pixfmt_unit_size(enum gpujpeg_pixel_format pf);
const struct gpujpeg_component_sampling_factor* gj_pixfmt_sampling(enum gpujpeg_pixel_format pf);
int gj_pixfmt_is_interleaved(enum gpujpeg_pixel_format pf);
gpujpeg_sampling_factor_t gj_make_sampling_factor(int comp_count, const struct gpujpeg_component_sampling_factor* sf);
int gj_parse_channel_remap(unsigned* out, const char* val, const char* optname); /* src/gpujpeg_encoder.c:661-699 */
bool gj_parameters_equal(const struct gpujpeg_parameters* a, const struct gpujpeg_parameters* b);
bool gj_image_parameters_equal(const struct gpujpeg_image_parameters* a, const struct gpujpeg_image_parameters* b);

/* ---- geometry (src/gpujpeg_common.c:675-870, 1040-1085) ---- */
int gj_geom_init(gj_geom* g, const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* param_image, bool encoder);

/* ---- tables ---- */
extern const uint8_t gj_zigzag[64];
void gj_quant_table_raw(int type, int quality, uint8_t raw_zigzag[64]);                       /* src/gpujpeg_table.c:35-100 */
void gj_quant_table_forward(const uint8_t raw_zigzag[64], float fwd[64]);                    /* src/gpujpeg_table.c:103-123 */
void gj_quant_table_inverse(const uint8_t raw_zigzag[64], uint16_t inv[64]);                 /* src/gpujpeg_table.c:154-160 */
void gj_huffman_std_spec(int type, int is_ac, const uint8_t** bits17, const uint8_t** vals, int* count); /* :190-254 */
void gj_huffman_encoder_lut(uint32_t lut[4 * 256]);                                          /* src/gpujpeg_huffman_gpu_encoder.cu:958-969 */
int gj_huffman_decoder_table(const uint8_t bits17[17], const uint8_t* vals, uint16_t out[GJ_DEC_TAB_WORDS]); /* src/gpujpeg_table.c:384-449 */
int gj_huffman_decoder_table2(const uint8_t bits17[17], const uint8_t* vals, int is_ac, uint16_t out[GJ_DEC2_WORDS]);

/* ---- timers: hipEvents around the stages (src/gpujpeg_common_internal.h:156-205) ---- */
struct gj_timers {
    gj_event_t ev[GJ_ENC_EVENTS];
    /* [0] is recorded in front of every host <-> device copy of a call whether or not statistics are wanted: with an event record in front of
     * the upload AND one in front of the download the HIP runtime overlaps the copies of concurrent coders (8K, four pipelines, pinned buffers both
     * ways: 10.6 against 8.3 Gpix/s; one of the two alone changes nothing; measured in round 4, profiles/r4_08_copy_markers.txt). */
    gj_event_t copy_in[2], copy_out[2];
    /* what the calling thread waits for behind a copy through the process's copy lanes (gj_hip_upload / gj_hip_download) */
    gj_event_t lane_in, lane_out;
    bool valid;
};

/* ---- coder state shared by encoder and decoder ---- */
struct gj_coder {
    struct gpujpeg_parameters param;
    struct gpujpeg_image_parameters param_image;
    gj_geom geom;
    bool configured;
    gj_stream_t stream;
    int device;
    /* device buffers */
    uint8_t* d_raw_own; size_t d_raw_cap;
    uint8_t* d_planes; size_t d_planes_cap;
    int16_t* d_coefs; size_t d_coefs_cap;
    /* statistics (src/gpujpeg_common.c:2170-2254) */
    struct gj_timers timers;
    struct gpujpeg_duration_stats stats;
    float kernel_ms[8]; /* per-kernel durations of the last call (include/gpujpeg_amd_ext.h) */
    double start_time, init_end_time, stop_time;
    double first_frame_duration, aggregate_duration;
    long frames;
    bool encoder;
    /* GJ_HOST_TIMING=1 (developer switch): where the host's time of a call goes -- [0] entry -> first launch, [1] the launches, [2] waiting for the
     * stream, [3] behind the wait; printed when the coder is destroyed */
    bool ht_on;
    double ht[4], ht_mark;
    long ht_calls;
};
#define GJ_HT_START(c) do { if ((c)->ht_on) (c)->ht_mark = gj_now_us(); } while (0)
#define GJ_HT(c, i) do { if ((c)->ht_on) { const double t_ = gj_now_us(); (c)->ht[i] += t_ - (c)->ht_mark; (c)->ht_mark = t_; } } while (0)
double gj_now_us(void);

void gj_coder_process_stats(struct gj_coder* c, bool with_stats);
void gj_coder_process_stats_overall(struct gj_coder* c);
int gj_timers_create(struct gj_timers* t);
void gj_timers_destroy(struct gj_timers* t);
int gj_ensure_device_buffer(void** p, size_t* cap, size_t need);

/* ---- writer (src/gpujpeg_writer.c) ---- */
struct gj_scan_headers {
    uint8_t* bytes;             /* all scan headers back to back */
    size_t size;
    uint32_t offset[GJ_MAX_COMP + 1];
    uint32_t info_payload[GJ_MAX_COMP];
};
struct gj_exif_tags; /* user Exif tags (enc_exif_tag), gj_writer.c */
int gj_exif_add_tag(struct gj_exif_tags** tags, const char* cfg); /* 0 = added, -1 = error or "help" */
void gj_exif_tags_destroy(struct gj_exif_tags* tags);
size_t gj_write_main_header(uint8_t* out, size_t out_cap, const gj_geom* g, const struct gpujpeg_parameters* param, enum gpujpeg_header_type header_type,
                            const uint8_t qraw[2][64], const struct gpujpeg_image_metadata* metadata, const struct gj_exif_tags* exif_tags);
int gj_write_scan_headers(struct gj_scan_headers* sh, const gj_geom* g, const struct gpujpeg_parameters* param);

/* ---- raster file formats behind gpujpeg_image_load_from_file / save_to_file (gj_image_io.c, gj_image_png.c) ---- */
struct gj_raster { int w, h, comps; uint8_t* px; }; /* top-down, tightly packed, comps interleaved 8-bit channels; px is malloc'ed */
int gj_png_decode(const uint8_t* d, size_t n, struct gj_raster* out, int want_pixels);
int gj_gif_decode(const uint8_t* d, size_t n, struct gj_raster* out, int want_pixels);
int gj_png_save(const char* filename, const uint8_t* image, int w, int h, int comps, size_t pitch);

/* ---- reader (src/gpujpeg_reader.c) ---- */
struct gj_reader_result {
    struct gpujpeg_parameters param;
    struct gpujpeg_image_parameters param_image;
    enum gpujpeg_header_type header_type;
    enum gpujpeg_color_space header_color_space;
    uint8_t comp_id[GJ_MAX_COMP];
    int quant_map[GJ_MAX_COMP];
    int huff_map[GJ_MAX_COMP][2];
    uint8_t qraw[4][64];
    bool q_present[4];
    uint8_t hbits[4][2][17];
    uint8_t hvals[4][2][256];
    bool h_present[4][2];
    const char* comment;
    struct gpujpeg_image_metadata metadata;
    /* scans: byte range of the entropy-coded data of each scan inside the file */
    int scan_count;
    size_t scan_begin[GJ_MAX_COMP], scan_end[GJ_MAX_COMP];
    /* APP13 segment info, if present: per scan, pointer to big-endian u32 offsets */
    const uint8_t* seg_info[GJ_MAX_COMP][GPUJPEG_MAX_SEGMENT_INFO_HEADER_COUNT];
    int seg_info_size[GJ_MAX_COMP][GPUJPEG_MAX_SEGMENT_INFO_HEADER_COUNT];
    int seg_info_count[GJ_MAX_COMP];
    size_t header_end; /* offset of the first SOS marker */
    bool eoi_seen;
};
/* parse every marker; when segments != NULL also split scans at RSTn on the host (reference behaviour) */
struct gj_host_segments {
    uint32_t* pos; uint32_t* len; uint32_t* index; /* arrays of capacity cap */
    int count, cap;
};
int gj_reader_parse(const uint8_t* image, size_t size, int verbose, bool ff_cs_itu601_is_709,
                    enum gpujpeg_pixel_format req_pixfmt, enum gpujpeg_color_space req_cs, unsigned req_alignment,
                    struct gj_reader_result* r, bool headers_only);
int gj_reader_split_scans(const uint8_t* image, const struct gj_reader_result* r, const gj_geom* g, struct gj_host_segments* segs, int verbose);

/* ---- image file I/O (src/utils/image_delegate.c, pam.c, y4m.c) ---- */
int gj_image_load(const char* filename, enum gpujpeg_image_file_format fmt, uint8_t** image, size_t* size);
int gj_image_save(const char* filename, enum gpujpeg_image_file_format fmt, const uint8_t* image, size_t size,
                  const struct gpujpeg_image_parameters* param_image);
int gj_image_probe(const char* filename, enum gpujpeg_image_file_format fmt, struct gpujpeg_image_parameters* param_image, int file_exists);

#endif
