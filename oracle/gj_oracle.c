/*
 * gj_oracle.c -- CPU restatement of the CESNET/GPUJPEG hot path (scalar C, one sample at a time).
 *
 * TEST INFRASTRUCTURE ONLY -- see gj_oracle.h. This file is the *checker*: the product under
 * gpujpeg_amd/ never links it. It follows the reference's algorithm stage by stage and cites the
 * reference file:line next to each function (paths relative to /root/reference).
 *
 * Compile with -ffp-contract=off: every fused multiply-add below is an explicit fmaf(). The fusion map is pinned in two
 * steps: with gjo_set_fma(0) every fmaf(a, b, c) becomes a * b + c in two roundings and the result must equal
 * the reference's own CUDA translation units compiled with -ffp-contract=off (oracle/_ref/libgpujpeg_ref_nofma.so, CPU); with
 * fusion on it must equal the same translation units compiled by hipcc for gfx950 (libgpujpeg_refhip.so, GPU box).
 */
#include "gj_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int g_fma = 1;
void gjo_set_fma(int on) { g_fma = on; }
static inline float gjo_fmaf(float a, float b, float c) { return g_fma ? fmaf(a, b, c) : a * b + c; }
#define fmaf(a, b, c) gjo_fmaf(a, b, c)

/* ------------------------------------------------------------------------------------------------
 * Pixel formats / parameters
 * ---------------------------------------------------------------------------------------------- */

/* src/gpujpeg_common.c:140-151 (descriptor table) */
static const struct { int comps, bpp, h[4], v[4]; } k_pixfmt[7] = {
    /* U8          */ {1, 1, {1, 0, 0, 0}, {1, 0, 0, 0}},
    /* 444 P012    */ {3, 3, {1, 1, 1, 0}, {1, 1, 1, 0}},
    /* 444 P0P1P2  */ {3, 0, {1, 1, 1, 0}, {1, 1, 1, 0}},
    /* 422 P1020   */ {3, 2, {2, 1, 1, 0}, {1, 1, 1, 0}},
    /* 422 P0P1P2  */ {3, 0, {2, 1, 1, 0}, {1, 1, 1, 0}},
    /* 420 P0P1P2  */ {3, 0, {2, 1, 1, 0}, {2, 1, 1, 0}},
    /* 4444 P0123  */ {4, 4, {1, 1, 1, 1}, {1, 1, 1, 1}},
};

int gjo_pixfmt_comp_count(int pf) { return (pf >= 0 && pf < 7) ? k_pixfmt[pf].comps : 0; }
int gjo_pixfmt_unit_size(int pf) { return (pf >= 0 && pf < 7) ? k_pixfmt[pf].bpp : 0; }
void gjo_pixfmt_sampling(int pf, int h[GJO_MAX_COMP], int v[GJO_MAX_COMP])
{
    for (int i = 0; i < GJO_MAX_COMP; i++) { h[i] = k_pixfmt[pf].h[i]; v[i] = k_pixfmt[pf].v[i]; }
}

/* src/gpujpeg_common.c:1180-1204 */
size_t gjo_raw_size(int width, int height, int width_padding, int pf)
{
    int bpp = gjo_pixfmt_unit_size(pf);
    if (bpp != 0) return ((size_t)width + width_padding) * height * bpp;
    switch (pf) {
    case GJO_PF_444_P0P1P2: return (size_t)width * height * 3;
    case GJO_PF_422_P0P1P2: return (size_t)width * height + (size_t)2 * ((width + 1) / 2) * height;
    case GJO_PF_420_P0P1P2: return (size_t)width * height + (size_t)2 * ((width + 1) / 2) * ((height + 1) / 2);
    default: return 0;
    }
}

/* src/gpujpeg_encoder.c:291-317 */
int gjo_suggest_restart_interval(int width, int height, int pf, int is_444, int interleaved)
{
    int ri;
    const int comp_count = gjo_pixfmt_comp_count(pf);
    double coefficient = ((double)width * height * comp_count) / (1000000.0 * 3.0);
    if (coefficient < 1.0) ri = 4;
    else if (coefficient < 3.0) ri = 8;
    else if (coefficient < 9.0) ri = 10;
    else ri = 12;
    if (!is_444 && interleaved) ri /= 2;
    if (!interleaved) ri *= comp_count;
    return ri;
}

/* src/gpujpeg_encoder.c:320-346 (first-frame behaviour: img_changed == true) */
void gjo_adjust_encoder_params(gjo_image* img)
{
    if (img->comp_count == 0) {
        int c = gjo_pixfmt_comp_count(img->pixel_format);
        img->comp_count = c < 3 ? c : 3;
        gjo_pixfmt_sampling(img->pixel_format, img->samp_h, img->samp_v);
    }
    if (img->restart_interval < 0) {
        int is444 = img->comp_count == 3;
        for (int i = 0; i < img->comp_count; i++)
            if (img->samp_h[i] != 1 || img->samp_v[i] != 1) is444 = 0;
        /* gpujpeg_make_sampling_factor2 yields GPUJPEG_SUBSAMPLING_444 only for 3 comps all 1x1 */
        img->restart_interval = gjo_suggest_restart_interval(img->width, img->height, img->pixel_format, is444, img->interleaved);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Geometry -- src/gpujpeg_common.c:675-870
 * ---------------------------------------------------------------------------------------------- */
static int round_up_div(int a, int b) { return (a + b - 1) / b; }

int gjo_image_init(gjo_image* img)
{
    if (img->comp_count < 1 || img->comp_count > GJO_MAX_COMP || img->width <= 0 || img->height <= 0) return -1;
    img->raw_size = gjo_raw_size(img->width, img->height, img->width_padding, img->pixel_format);
    img->max_h = img->max_v = 0;
    for (int c = 0; c < img->comp_count; c++) {
        if (img->samp_h[c] < 1 || img->samp_h[c] > 15 || img->samp_v[c] < 1 || img->samp_v[c] > 15) return -1;
        if (img->samp_h[c] > img->max_h) img->max_h = img->samp_h[c];
        if (img->samp_v[c] > img->max_v) img->max_v = img->samp_v[c];
    }
    /* NB: the reference computes the running maximum inside the same loop (common.c:680-689), i.e.
     * a component sees only the maxima of components 0..c. For all standard layouts component 0
     * carries the maximum, so both orders agree; we keep the reference's running form. */
    int run_h = 0, run_v = 0;
    img->data_size = 0;
    for (int c = 0; c < img->comp_count; c++) {
        gjo_comp* k = &img->comp[c];
        k->h = img->samp_h[c];
        k->v = img->samp_v[c];
        if (k->h > run_h) run_h = k->h;
        if (k->v > run_v) run_v = k->v;
        k->type = (img->color_space_internal == GJO_CS_RGB || c == 0 || c == 3) ? GJO_LUMA : GJO_CHROMA; /* :691-694 */
        int div_h = run_h / k->h, div_v = run_v / k->v;
        int width = round_up_div(img->width, div_h) * div_h;
        int height = round_up_div(img->height, div_v) * div_v;
        k->width = (width * k->h) / run_h;
        k->height = (height * k->v) / run_v;
        k->mcu_size_x = 8;
        k->mcu_size_y = 8;
        if (img->interleaved) { k->mcu_size_x *= k->h; k->mcu_size_y *= k->v; }
        k->mcu_size = k->mcu_size_x * k->mcu_size_y;
        k->data_width = round_up_div(k->width, k->mcu_size_x) * k->mcu_size_x;
        k->data_height = round_up_div(k->height, k->mcu_size_y) * k->mcu_size_y;
        k->data_offset = img->data_size;
        img->data_size += (size_t)k->data_width * k->data_height;
        k->mcu_count_x = round_up_div(k->data_width, k->mcu_size_x);
        k->mcu_count_y = round_up_div(k->data_height, k->mcu_size_y);
        k->mcu_count = k->mcu_count_x * k->mcu_count_y;
        k->segment_mcu_count = img->restart_interval;
        if (k->segment_mcu_count == 0) k->segment_mcu_count = k->mcu_count;
        k->segment_count = round_up_div(k->mcu_count, k->segment_mcu_count);
    }
    img->block_count = (int)(img->data_size / 64);
    if (img->interleaved) {
        img->mcu_count = img->comp[0].mcu_count;
        img->segment_count = img->comp[0].segment_count;
        img->segment_mcu_count = img->comp[0].segment_mcu_count;
        for (int c = 1; c < img->comp_count; c++)
            if (img->comp[c].mcu_count != img->mcu_count) return -1; /* reference asserts, common.c:761 */
        img->scan_count = 1;
    } else {
        img->mcu_count = 0;
        img->segment_count = 0;
        img->segment_mcu_count = 0;
        for (int c = 0; c < img->comp_count; c++) {
            img->mcu_count += img->comp[c].mcu_count;
            img->segment_count += img->comp[c].segment_count;
        }
        img->scan_count = img->comp_count;
    }
    return 0;
}

/* src/gpujpeg_common.c:813-870 */
void gjo_segment_get(const gjo_image* img, int index, gjo_segment* seg)
{
    if (img->interleaved) {
        int first = index * img->segment_mcu_count;
        int n = img->segment_mcu_count;
        if (first + n >= img->mcu_count) n = img->mcu_count - first;
        seg->scan_index = 0;
        seg->scan_segment_index = index;
        seg->mcu_count = n;
        return;
    }
    int c = 0;
    while (index >= img->comp[c].segment_count) { index -= img->comp[c].segment_count; c++; }
    const gjo_comp* k = &img->comp[c];
    int first = index * k->segment_mcu_count;
    int n = k->segment_mcu_count;
    if (first + n >= k->mcu_count) n = k->mcu_count - first;
    seg->scan_index = c;
    seg->scan_segment_index = index;
    seg->mcu_count = n;
}

static int blocks_per_mcu(const gjo_image* img)
{
    if (!img->interleaved) return 1;
    int n = 0;
    for (int c = 0; c < img->comp_count; c++) n += img->comp[c].h * img->comp[c].v;
    return n;
}

int gjo_segment_block_count(const gjo_image* img, const gjo_segment* seg) { return seg->mcu_count * blocks_per_mcu(img); }

/* src/gpujpeg_common.c:1040-1085 (the block list) */
size_t gjo_segment_block(const gjo_image* img, const gjo_segment* seg, int kblk, int* comp_out)
{
    if (!img->interleaved) {
        const gjo_comp* k = &img->comp[seg->scan_index];
        *comp_out = seg->scan_index;
        return k->data_offset + ((size_t)seg->scan_segment_index * k->segment_mcu_count + kblk) * 64;
    }
    int per = blocks_per_mcu(img);
    int mcu_in_seg = kblk / per, pos = kblk % per;
    for (int c = 0; c < img->comp_count; c++) {
        const gjo_comp* k = &img->comp[c];
        int n = k->h * k->v;
        if (pos < n) {
            int mcu = seg->scan_segment_index * k->segment_mcu_count + mcu_in_seg;
            int mx = mcu % k->mcu_count_x, my = mcu / k->mcu_count_x;
            int by = pos / k->h, bx = pos % k->h;
            size_t base = k->data_offset + (size_t)my * ((size_t)k->mcu_size * k->mcu_count_x) + (size_t)mx * (k->mcu_size_x * 8);
            size_t row = base + (size_t)by * ((size_t)k->mcu_count_x * k->mcu_size_x * 8);
            *comp_out = c;
            return row + (size_t)bx * 64;
        }
        pos -= n;
    }
    *comp_out = 0;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Tables -- src/gpujpeg_table.c:35-129 (quantisation), :190-254 (Huffman specs, ITU T.81 Annex K)
 * ---------------------------------------------------------------------------------------------- */
const int gjo_zigzag[64] = {
     0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
};

/* ITU T.81 Table K.1 / K.2, listed in zig-zag scan order as the reference stores them */
static const uint8_t k_q_luma[64] = {
    16, 11, 12, 14, 12, 10, 16, 14, 13, 14, 18, 17, 16, 19, 24, 40, 26, 24, 22, 22, 24, 49, 35, 37, 29, 40, 58, 51, 61, 60, 57, 51,
    56, 55, 64, 72, 92, 78, 64, 68, 87, 69, 55, 56, 80, 109, 81, 87, 95, 98, 103, 104, 103, 62, 77, 113, 121, 112, 100, 120, 92, 101, 103, 99};
static const uint8_t k_q_chroma[64] = {
    17, 18, 18, 24, 21, 24, 47, 26, 26, 47, 99, 66, 56, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

void gjo_quant_table(int type, int quality, uint8_t raw[64], float fwd[64], uint16_t inv[64])
{
    const uint8_t* base = type == GJO_LUMA ? k_q_luma : k_q_chroma;
    /* :84-100 IJG quality scaling */
    if (quality <= 0) quality = 1;
    if (quality > 100) quality = 100;
    int s = (quality < 50) ? (5000 / quality) : (200 - 2 * quality);
    for (int i = 0; i < 64; i++) {
        int value = (s * (int)base[i] + 50) / 100;
        if (value == 0) value = 1;
        if (value > 255) value = 255;
        raw[i] = (uint8_t)value;
    }
    /* :103-123 forward table: 1/(q * scale_x * scale_y * 8) in double, stored float, transposed */
    static const double sc[8] = {1.0, 1.387039845, 1.306562965, 1.175875602, 1.0, 0.785694958, 0.541196100, 0.275899379};
    for (int i = 0; i < 64; i++) {
        int x = gjo_zigzag[i] % 8, y = gjo_zigzag[i] / 8;
        if (fwd) fwd[x * 8 + y] = (float)(1.0 / (raw[i] * sc[x] * sc[y] * 8));
        if (inv) inv[gjo_zigzag[i]] = raw[i]; /* :154-160 */
    }
}

/* ITU T.81 Annex K.3 typical Huffman tables (BITS + HUFFVAL) */
static const uint8_t k_dc_luma_bits[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t k_dc_chroma_bits[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t k_dc_vals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t k_ac_luma_bits[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t k_ac_luma_vals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91,
    0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a,
    0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53,
    0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79,
    0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9,
    0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t k_ac_chroma_bits[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t k_ac_chroma_vals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14,
    0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17,
    0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a,
    0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78,
    0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7,
    0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

void gjo_huffman_spec(int index, const uint8_t** bits, const uint8_t** vals, int* nvals)
{
    switch (index) {
    case 0: *bits = k_dc_luma_bits; *vals = k_dc_vals; *nvals = 12; break;
    case 1: *bits = k_ac_luma_bits; *vals = k_ac_luma_vals; *nvals = 162; break;
    case 2: *bits = k_dc_chroma_bits; *vals = k_dc_vals; *nvals = 12; break;
    default: *bits = k_ac_chroma_bits; *vals = k_ac_chroma_vals; *nvals = 162; break;
    }
}

/* code/size per symbol: ITU T.81 Annex C figures C.1-C.3 (src/gpujpeg_table.c:265-306) */
typedef struct { uint16_t code[256]; uint8_t size[256]; } enc_table;

static void build_enc_table(const uint8_t bits[17], const uint8_t* vals, enc_table* t)
{
    memset(t, 0, sizeof *t);
    unsigned code = 0;
    int p = 0;
    for (int len = 1; len <= 16; len++) {
        for (int i = 0; i < bits[len]; i++, p++) {
            t->code[vals[p]] = (uint16_t)code;
            t->size[vals[p]] = (uint8_t)len;
            code++;
        }
        code <<= 1;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Colour transforms -- src/gpujpeg_colorspace.h:50-102 (core), :216-430 (matrices and chains)
 * ---------------------------------------------------------------------------------------------- */
static uint8_t clamp_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

static void matrix_to(uint8_t c[3], const int m[9], int b1, int b2, int b3) /* :64-80 */
{
    int r0 = (int)c[0] * 256 / 255, r1 = (int)c[1] * 256 / 255, r2 = (int)c[2] * 256 / 255;
    c[0] = clamp_u8(((m[0] * r0 + m[1] * r1 + m[2] * r2 + 128) >> 8) + b1);
    c[1] = clamp_u8(((m[3] * r0 + m[4] * r1 + m[5] * r2 + 128) >> 8) + b2);
    c[2] = clamp_u8(((m[6] * r0 + m[7] * r1 + m[8] * r2 + 128) >> 8) + b3);
}

static void matrix_from(uint8_t c[3], const int m[9], int b1, int b2, int b3) /* :86-102 */
{
    int r0 = ((int)c[0] - b1) * 256 / 255, r1 = ((int)c[1] - b2) * 256 / 255, r2 = ((int)c[2] - b3) * 256 / 255;
    c[0] = clamp_u8((m[0] * r0 + m[1] * r1 + m[2] * r2 + 128) >> 8);
    c[1] = clamp_u8((m[3] * r0 + m[4] * r1 + m[5] * r2 + 128) >> 8);
    c[2] = clamp_u8((m[6] * r0 + m[7] * r1 + m[8] * r2 + 128) >> 8);
}

static void rgb_to(int cs, uint8_t c[3])
{
    static const int m601[9] = {66, 129, 25, -38, -74, 112, 112, -94, -18};      /* :216-232 */
    static const int m601f[9] = {77, 150, 29, -43, -85, 128, 128, -107, -21};    /* :251-266 */
    static const int m709[9] = {47, 157, 16, -26, -87, 112, 112, -102, -10};     /* :286-301 */
    static const int myuv[9] = {77, 150, 29, -38, -74, 112, 157, -132, -26};     /* :321-335 */
    switch (cs) {
    case GJO_CS_BT601: matrix_to(c, m601, 16, 128, 128); break;
    case GJO_CS_BT601_256: matrix_to(c, m601f, 0, 128, 128); break;
    case GJO_CS_BT709: matrix_to(c, m709, 16, 128, 128); break;
    case GJO_CS_YUV: matrix_to(c, myuv, 0, 128, 128); break;
    default: break;
    }
}

static void to_rgb(int cs, uint8_t c[3])
{
    static const int m601[9] = {298, 0, 409, 298, -100, -208, 298, 516, 0};      /* :233-249 */
    static const int m601f[9] = {256, 0, 359, 256, -88, -183, 256, 454, 0};      /* :268-284 */
    static const int m709[9] = {298, 0, 459, 298, -55, -136, 298, 541, 0};       /* :303-318 */
    static const int myuv[9] = {256, 0, 292, 256, -101, -149, 256, 520, 0};      /* :337-351 */
    switch (cs) {
    case GJO_CS_BT601: matrix_from(c, m601, 16, 128, 128); break;
    case GJO_CS_BT601_256: matrix_from(c, m601f, 0, 128, 128); break;
    case GJO_CS_BT709: matrix_from(c, m709, 16, 128, 128); break;
    case GJO_CS_YUV: matrix_from(c, myuv, 0, 128, 128); break;
    default: break;
    }
}

void gjo_color_transform(int from, int to, uint8_t c[3])
{
    if (from == to || from == GJO_CS_NONE || to == GJO_CS_NONE) return;   /* :160-213 */
    if (from == GJO_CS_RGB) { rgb_to(to, c); return; }
    if (to == GJO_CS_RGB) { to_rgb(from, c); return; }
    /* YCbCr -> YCbCr chains through RGB (:354-427). Observable quirk kept for parity:
     * BT.601-limited -> BT.709 uses the FULL-range inverse first (:387-394). */
    if (from == GJO_CS_BT601 && to == GJO_CS_BT709) { to_rgb(GJO_CS_BT601_256, c); rgb_to(GJO_CS_BT709, c); return; }
    /* pairs the reference does not specialise fall into the asserting primary template; we chain them */
    to_rgb(from, c);
    rgb_to(to, c);
}

/* ------------------------------------------------------------------------------------------------
 * Preprocessor -- src/gpujpeg_preprocessor.cu:49-202, :358-453
 * ---------------------------------------------------------------------------------------------- */
static int pixfmt_is_planar(int pf) { return pf == GJO_PF_444_P0P1P2 || pf == GJO_PF_422_P0P1P2 || pf == GJO_PF_420_P0P1P2; }

/* load one pixel as (c0,c1,c2,c3), src/gpujpeg_preprocessor.cu:88-159 */
static void raw_load(const gjo_image* img, const uint8_t* raw, int W, int H, int x, int y, uint8_t r[4])
{
    const int pos = y * W + x;
    r[3] = 0;
    switch (img->pixel_format) {
    case GJO_PF_U8: r[0] = raw[pos + img->width_padding * y]; r[1] = 128; r[2] = 128; break;
    case GJO_PF_444_P0P1P2: r[0] = raw[pos]; r[1] = raw[W * H + pos]; r[2] = raw[2 * W * H + pos]; break;
    case GJO_PF_422_P0P1P2:
        r[0] = raw[pos]; r[1] = raw[W * H + pos / 2]; r[2] = raw[W * H + H * ((W + 1) / 2) + pos / 2]; break;
    case GJO_PF_420_P0P1P2:
        r[0] = raw[pos];
        r[1] = raw[W * H + y / 2 * ((W + 1) / 2) + x / 2];
        r[2] = raw[W * H + ((H + 1) / 2 + y / 2) * ((W + 1) / 2) + x / 2];
        break;
    case GJO_PF_444_P012: { const uint8_t* p = raw + (size_t)pos * 3 + (size_t)img->width_padding * y; r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; break; }
    case GJO_PF_4444_P0123: { const uint8_t* p = raw + (size_t)pos * 4 + (size_t)img->width_padding * y; r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; r[3] = p[3]; break; }
    case GJO_PF_422_P1020: {
        size_t off = (size_t)pos * 2 + (size_t)img->width_padding * y;
        r[0] = raw[off + 1];
        if (off % 4 == 0) { r[1] = raw[off]; r[2] = raw[off + 2]; }
        else { r[1] = raw[off - 2]; r[2] = raw[off]; }
        break; }
    default: r[0] = r[1] = r[2] = 0; break;
    }
}

/* src/gpujpeg_preprocessor.cu:294-314 */
static int encode_no_transform(const gjo_image* img)
{
    if (!pixfmt_is_planar(img->pixel_format) && img->pixel_format != GJO_PF_U8) return 0;
    /* NB: gpujpeg_pixel_format_is_interleaved() is true for U8 too? -- no: U8 is neither planar-flagged
     * nor multi-component; reference: interleaved := !planar && comp_count > 1 (common.c) */
    if (img->comp_count == 3 && img->color_space != img->color_space_internal) return 0;
    int h[4], v[4];
    gjo_pixfmt_sampling(img->pixel_format, h, v);
    for (int i = 0; i < img->comp_count; i++)
        if (img->comp[i].h != h[i] || img->comp[i].v != v[i]) return 0;
    return 1;
}

void gjo_preprocess(const gjo_image* img, const uint8_t* raw, uint8_t* planes)
{
    memset(planes, 0, img->data_size); /* src/gpujpeg_common.c:941-944: zero padding */
    if (encode_no_transform(img)) {
        /* planar copy path, src/gpujpeg_preprocessor.cu:423-453 */
        size_t off = 0;
        for (int c = 0; c < img->comp_count; c++) {
            const gjo_comp* k = &img->comp[c];
            int spitch = k->width + img->width_padding;
            for (int y = 0; y < k->height; y++)
                memcpy(planes + k->data_offset + (size_t)y * k->data_width, raw + off + (size_t)y * spitch, k->width);
            off += (size_t)spitch * k->height;
        }
        return;
    }
    int W = img->width, H = img->height;
    if (img->pixel_format == GJO_PF_422_P1020) W = (img->width + 1) & ~1; /* :369-373 */
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            uint8_t r[4];
            raw_load(img, raw, W, H, x, y, r);
            gjo_color_transform(img->color_space, img->color_space_internal, r);
            for (int c = 0; c < img->comp_count; c++) {
                const gjo_comp* k = &img->comp[c];
                int sh = img->max_h / k->h, sv = img->max_v / k->v;   /* :325-329 */
                if ((x % sh) || (y % sv)) continue;                    /* :56-63 point sampling */
                planes[k->data_offset + (size_t)(y / sv) * k->data_width + x / sh] = r[c];
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Forward DCT + quantisation -- src/gpujpeg_dct_gpu.cu:121-163 (1-D AAN), :180-295 (kernel)
 * The reference is CUDA compiled with default -fmad=true; the fusion map below is the one LLVM's
 * DAG combiner produces for this expression sequence under aggressive FMA fusion.
 * ---------------------------------------------------------------------------------------------- */
static void fdct8(const float in[8], float out[8], float level_shift)
{
    const float diff0 = in[0] + in[7], diff1 = in[1] + in[6], diff2 = in[2] + in[5], diff3 = in[3] + in[4];
    const float diff4 = in[3] - in[4], diff5 = in[2] - in[5], diff6 = in[1] - in[6], diff7 = in[0] - in[7];
    const float even0 = diff0 + diff3, even1 = diff1 + diff2, even2 = diff1 - diff2, even3 = diff0 - diff3;
    const float even_diff = even2 + even3;
    const float odd0 = diff4 + diff5, odd1 = diff5 + diff6, odd2 = diff6 + diff7;
    const float odd_diff5 = (odd0 - odd2) * 0.382683433f;
    const float odd_diff4 = fmaf(1.306562965f, odd2, odd_diff5);
    const float odd_diff3 = fmaf(-odd1, 0.707106781f, diff7);
    const float odd_diff2 = fmaf(0.541196100f, odd0, odd_diff5);
    const float odd_diff1 = fmaf(odd1, 0.707106781f, diff7);
    out[0] = (even0 + even1) + level_shift;
    out[1] = odd_diff1 + odd_diff4;
    out[2] = fmaf(even_diff, 0.707106781f, even3);
    out[3] = odd_diff3 - odd_diff2;
    out[4] = even0 - even1;
    out[5] = odd_diff3 + odd_diff2;
    out[6] = fmaf(-even_diff, 0.707106781f, even3);
    out[7] = odd_diff1 - odd_diff4;
}

void gjo_fdct_quant_block(const uint8_t* src, int stride, const float fwd[64], int16_t out[64])
{
    float t[8][8]; /* t[u][col]: vertical frequency u of column col (:246-258) */
    for (int col = 0; col < 8; col++) {
        float in[8], o[8];
        for (int r = 0; r < 8; r++) in[r] = (float)src[r * stride + col];
        fdct8(in, o, -1024.0f);
        for (int u = 0; u < 8; u++) t[u][col] = o[u];
    }
    for (int u = 0; u < 8; u++) { /* :262-266 row pass, then :274-281 quantise with rintf (ties to even) */
        float o[8];
        fdct8(t[u], o, 0.0f);
        for (int j = 0; j < 8; j++) out[u * 8 + j] = (int16_t)(int)rintf(o[j] * fwd[j * 8 + u]);
    }
}

void gjo_fdct_quant(const gjo_image* img, const uint8_t* planes, int16_t* coefs)
{
    uint8_t raw[64];
    float fwd[2][64];
    gjo_quant_table(GJO_LUMA, img->quality, raw, fwd[0], NULL);
    gjo_quant_table(GJO_CHROMA, img->quality, raw, fwd[1], NULL);
    for (int c = 0; c < img->comp_count; c++) {
        const gjo_comp* k = &img->comp[c];
        int bw = k->data_width / 8, bh = k->data_height / 8;
        for (int by = 0; by < bh; by++)
            for (int bx = 0; bx < bw; bx++)
                gjo_fdct_quant_block(planes + k->data_offset + (size_t)by * 8 * k->data_width + bx * 8, k->data_width,
                                     fwd[k->type], coefs + k->data_offset + ((size_t)by * bw + bx) * 64);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Huffman encoder -- semantics of src/gpujpeg_huffman_gpu_encoder.cu:139-294 (symbols) and :103-131,
 * :417-503 (bit packing, byte stuffing, padding); identical stream to src/gpujpeg_huffman_cpu_encoder.c
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint8_t* out; size_t n; uint32_t acc; int bits; } bitw;

static void put_bits(bitw* w, unsigned code, int size)
{
    w->acc = (w->acc << size) | (code & ((1u << size) - 1));
    w->bits += size;
    while (w->bits >= 8) {
        uint8_t b = (uint8_t)(w->acc >> (w->bits - 8));
        w->out[w->n++] = b;
        if (b == 0xFF) w->out[w->n++] = 0;
        w->bits -= 8;
    }
}

static void encode_block(bitw* w, const int16_t* blk, int* dc, const enc_table* tdc, const enc_table* tac)
{
    int temp = blk[0] - *dc, temp2;
    *dc = blk[0];
    temp2 = temp;
    if (temp < 0) { temp = -temp; temp2--; }
    int nbits = 0;
    while (temp) { nbits++; temp >>= 1; }
    put_bits(w, tdc->code[nbits], tdc->size[nbits]);
    if (nbits) put_bits(w, (unsigned)temp2, nbits);
    int r = 0;
    for (int k = 1; k < 64; k++) {
        temp = blk[gjo_zigzag[k]];
        if (temp == 0) { r++; continue; }
        while (r > 15) { put_bits(w, tac->code[0xF0], tac->size[0xF0]); r -= 16; }
        temp2 = temp;
        if (temp < 0) { temp = -temp; temp2--; }
        nbits = 1;
        while ((temp >>= 1)) nbits++;
        int sym = (r << 4) + nbits;
        put_bits(w, tac->code[sym], tac->size[sym]);
        put_bits(w, (unsigned)temp2, nbits);
        r = 0;
    }
    if (r > 0) put_bits(w, tac->code[0], tac->size[0]);
}

static void std_enc_tables(enc_table t[4])
{
    for (int i = 0; i < 4; i++) {
        const uint8_t *bits, *vals;
        int n;
        gjo_huffman_spec(i, &bits, &vals, &n);
        build_enc_table(bits, vals, &t[i]);
    }
}

size_t gjo_huffman_encode_segment(const gjo_image* img, const int16_t* coefs, int segment_index, uint8_t* out)
{
    enc_table t[4];
    std_enc_tables(t);
    gjo_segment seg;
    gjo_segment_get(img, segment_index, &seg);
    bitw w = {out, 0, 0, 0};
    int dc[GJO_MAX_COMP] = {0, 0, 0, 0};  /* predictors reset per segment (:339-342) */
    int nblk = gjo_segment_block_count(img, &seg);
    for (int k = 0; k < nblk; k++) {
        int c;
        size_t off = gjo_segment_block(img, &seg, k, &c);
        int type = img->comp[c].type;
        encode_block(&w, coefs + off, &dc[c], &t[type * 2], &t[type * 2 + 1]);
    }
    if (w.bits > 0) put_bits(&w, 0x7F, 8 - w.bits); /* pad with ones (:489) ; a 0xFF pad byte gets its stuffing zero */
    return w.n;
}

/* ------------------------------------------------------------------------------------------------
 * Stream writer -- src/gpujpeg_writer.c:120-160 (APP0), :172-250 (SPIFF), :255-270 (APP14),
 * :283-300 (DQT), :318-352 (SOF0), :363-405 (DHT), :414-449 (DRI, COM), :452-520 (header order),
 * :550-657 (scan header incl. APP13 segment info)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint8_t* p; size_t n; } bytew;
static void wb(bytew* w, int v) { w->p[w->n++] = (uint8_t)v; }
static void w2(bytew* w, int v) { wb(w, (v >> 8) & 0xFF); wb(w, v & 0xFF); }
static void w4(bytew* w, unsigned v) { wb(w, v >> 24); wb(w, (v >> 16) & 0xFF); wb(w, (v >> 8) & 0xFF); wb(w, v & 0xFF); }
static void wm(bytew* w, int marker) { wb(w, 0xFF); wb(w, marker); }

static int comp_id(const gjo_image* img, int i)
{
    static const uint8_t rgb_ids[4] = {'R', 'G', 'B', 'A'};
    return img->color_space_internal == GJO_CS_RGB ? rgb_ids[i] : i + 1;
}

static void write_app0(bytew* w)
{
    wm(w, 0xE0); w2(w, 16);
    wb(w, 'J'); wb(w, 'F'); wb(w, 'I'); wb(w, 'F'); wb(w, 0);
    wb(w, 1); wb(w, 1); wb(w, 1); w2(w, 300); w2(w, 300); wb(w, 0); wb(w, 0);
}

static void write_app14(bytew* w)
{
    wm(w, 0xEE); w2(w, 14);
    wb(w, 'A'); wb(w, 'd'); wb(w, 'o'); wb(w, 'b'); wb(w, 'e');
    w2(w, 100); w2(w, 0); w2(w, 0); wb(w, 0);
}

static void write_spiff(const gjo_image* img, bytew* w)
{
    int cs;
    if (img->comp_count == 1) cs = 8;
    else switch (img->color_space_internal) {
        case GJO_CS_BT709: cs = 1; break;
        case GJO_CS_BT601_256: cs = 3; break;
        case GJO_CS_BT601: cs = 4; break;
        case GJO_CS_RGB: cs = 10; break;
        default: cs = 2; break;
    }
    wm(w, 0xE8); w2(w, 32);
    wb(w, 'S'); wb(w, 'P'); wb(w, 'I'); wb(w, 'F'); wb(w, 'F'); wb(w, 0);
    w2(w, 0x100); wb(w, (cs == 3 || cs == 8) ? 1 : 0); wb(w, img->comp_count);
    w4(w, (unsigned)img->height); w4(w, (unsigned)img->width);
    wb(w, cs); wb(w, 8); wb(w, 5); wb(w, 0); w4(w, 1); w4(w, 1);
    wm(w, 0xE8); w2(w, 8); w4(w, 1);   /* directory: end-of-directory entry */
    wm(w, 0xD8);                       /* second SOI */
}

size_t gjo_write_header(const gjo_image* img, uint8_t* out)
{
    bytew w = {out, 0};
    wm(&w, 0xD8);
    int hdr = img->header_type;
    if (hdr == 0) { /* default selection :456-474 */
        if (img->comp_count == 4) hdr = 2;
        else if (img->color_space_internal == GJO_CS_BT601 || img->color_space_internal == GJO_CS_BT709) hdr = 2;
        else if (img->color_space_internal == GJO_CS_RGB) hdr = 4;
        else hdr = 1;
    }
    if (hdr == 1) write_app0(&w);
    else if (hdr == 2) write_spiff(img, &w);
    else if (hdr == 4) write_app14(&w);

    uint8_t qraw[2][64];
    gjo_quant_table(GJO_LUMA, img->quality, qraw[0], NULL, NULL);
    gjo_quant_table(GJO_CHROMA, img->quality, qraw[1], NULL, NULL);
    unsigned emitted = 0;
    for (int c = 0; c < img->comp_count; c++) {
        int t = img->comp[c].type;
        if (emitted & (1u << t)) continue;
        emitted |= 1u << t;
        wm(&w, 0xDB); w2(&w, 67); wb(&w, t);
        for (int i = 0; i < 64; i++) wb(&w, qraw[t][i]);
    }
    wm(&w, 0xC0); w2(&w, 8 + 3 * img->comp_count); wb(&w, 8); w2(&w, img->height); w2(&w, img->width); wb(&w, img->comp_count);
    for (int c = 0; c < img->comp_count; c++) {
        wb(&w, comp_id(img, c));
        wb(&w, (img->comp[c].h << 4) + img->comp[c].v);
        wb(&w, img->comp[c].type == GJO_LUMA ? 0 : 1);
    }
    emitted = 0;
    for (int c = 0; c < img->comp_count; c++) {
        int t = img->comp[c].type;
        if (emitted & (1u << t)) continue;
        emitted |= 1u << t;
        for (int ac = 0; ac < 2; ac++) {
            const uint8_t *bits, *vals;
            int n;
            gjo_huffman_spec(t * 2 + ac, &bits, &vals, &n);
            wm(&w, 0xC4); w2(&w, n + 2 + 1 + 16); wb(&w, (ac ? 16 : 0) + t);
            for (int i = 1; i <= 16; i++) wb(&w, bits[i]);
            for (int i = 0; i < n; i++) wb(&w, vals[i]);
        }
    }
    wm(&w, 0xDD); w2(&w, 4); w2(&w, img->restart_interval);
    char com[64];
    int q = img->quality < 1 ? 1 : (img->quality > 100 ? 100 : img->quality);
    int len = snprintf(com, sizeof com, "CREATOR: GPUJPEG, quality = %d", q);
    wm(&w, 0xFE); w2(&w, 2 + len + 1);
    for (int i = 0; i <= len; i++) wb(&w, com[i]);
    if (img->color_space_internal == GJO_CS_BT601) {
        static const char cs601[] = "CS=ITU601";
        wm(&w, 0xFE); w2(&w, 2 + (int)sizeof cs601);
        for (size_t i = 0; i < sizeof cs601; i++) wb(&w, cs601[i]);
    }
    return w.n;
}

#define GJO_MAX_HEADER_SIZE (65536 - 100)

/* writes APP13 placeholders (if enabled) + SOS; *info_pos receives offsets of the placeholder payloads */
static size_t write_scan_header(const gjo_image* img, int scan_index, uint8_t* out, size_t* info_payload, int* info_payload_count)
{
    bytew w = {out, 0};
    if (info_payload_count) *info_payload_count = 0;
    if (img->segment_info && img->restart_interval > 0) {
        int segs = img->interleaved ? img->segment_count : img->comp[scan_index].segment_count;
        int data_size = (segs + 1) * 4;
        while (data_size > 0) {
            int hs = data_size > GJO_MAX_HEADER_SIZE ? GJO_MAX_HEADER_SIZE : data_size;
            data_size -= hs;
            wm(&w, 0xED); w2(&w, 3 + hs); wb(&w, scan_index);
            if (info_payload) info_payload[(*info_payload_count)++] = w.n;
            memset(w.p + w.n, 0, hs);
            w.n += hs;
        }
    }
    wm(&w, 0xDA);
    if (img->interleaved) {
        w2(&w, 6 + 2 * img->comp_count); wb(&w, img->comp_count);
        for (int c = 0; c < img->comp_count; c++) { wb(&w, comp_id(img, c)); wb(&w, img->comp[c].type == GJO_LUMA ? 0 : 0x11); }
    } else {
        w2(&w, 8); wb(&w, 1); wb(&w, comp_id(img, scan_index)); wb(&w, img->comp[scan_index].type == GJO_LUMA ? 0 : 0x11);
    }
    wb(&w, 0); wb(&w, 0x3F); wb(&w, 0);
    return w.n;
}

size_t gjo_write_scan_header(const gjo_image* img, int scan_index, uint8_t* out)
{
    return write_scan_header(img, scan_index, out, NULL, NULL);
}

/* src/gpujpeg_encoder.c:567-629 stream stitching; APP13 index per src/gpujpeg_writer.c:522-547 */
static void put_segment_info(uint8_t* out, const size_t* payload, int index, unsigned position)
{
    size_t byte = (size_t)index * 4;
    uint8_t* p = out + payload[byte / GJO_MAX_HEADER_SIZE] + byte % GJO_MAX_HEADER_SIZE;
    p[0] = (uint8_t)(position >> 24); p[1] = (uint8_t)(position >> 16); p[2] = (uint8_t)(position >> 8); p[3] = (uint8_t)position;
}

size_t gjo_encode_from_coefs(const gjo_image* img, const int16_t* coefs, uint8_t* out, size_t cap)
{
    if (cap < 2048) return 0;
    size_t n = gjo_write_header(img, out);
    int seg = 0;
    /* worst case per block: 64 coefficients x (16 + 11) bits, every byte stuffed */
    size_t max_blocks = 0;
    for (int i = 0; i < img->segment_count; i++) {
        gjo_segment sg;
        gjo_segment_get(img, i, &sg);
        size_t b = (size_t)gjo_segment_block_count(img, &sg);
        if (b > max_blocks) max_blocks = b;
    }
    uint8_t* tmp = (uint8_t*)malloc(max_blocks * 512 + 64);
    if (!tmp) return 0;
    const int with_info = img->segment_info && img->restart_interval > 0;
    for (int scan = 0; scan < img->scan_count; scan++) {
        size_t payload[128];
        int npayload = 0;
        int segs = img->interleaved ? img->segment_count : img->comp[scan].segment_count;
        if (n + (size_t)(segs + 1) * 4 + 128 * 8 + 64 > cap) { free(tmp); return 0; }
        size_t hdr = n;
        n += write_scan_header(img, scan, out + n, payload, &npayload);
        for (int i = 0; i < npayload; i++) payload[i] += hdr;
        const size_t data_start = n;
        for (int s = 0; s < segs; s++, seg++) {
            size_t sz = gjo_huffman_encode_segment(img, coefs, seg, tmp);
            if (n + sz + 4 > cap) { free(tmp); return 0; }
            if (with_info) put_segment_info(out, payload, s, (unsigned)(n - data_start));
            memcpy(out + n, tmp, sz);
            n += sz;
            if (s + 1 < segs) { out[n++] = 0xFF; out[n++] = (uint8_t)(0xD0 + (s & 7)); }
        }
        if (with_info) put_segment_info(out, payload, segs, (unsigned)(n - data_start));
    }
    free(tmp);
    if (n + 2 > cap) return 0;
    out[n++] = 0xFF;
    out[n++] = 0xD9;
    return n;
}

size_t gjo_encode(gjo_image* img, const uint8_t* raw, uint8_t* out, size_t cap)
{
    gjo_adjust_encoder_params(img);
    if (gjo_image_init(img) != 0) return 0;
    uint8_t* planes = (uint8_t*)malloc(img->data_size);
    int16_t* coefs = (int16_t*)malloc(img->data_size * sizeof(int16_t));
    size_t n = 0;
    if (planes && coefs) {
        gjo_preprocess(img, raw, planes);
        gjo_fdct_quant(img, planes, coefs);
        n = gjo_encode_from_coefs(img, coefs, out, cap);
    }
    free(planes);
    free(coefs);
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * Stream reader -- src/gpujpeg_reader.c
 * ---------------------------------------------------------------------------------------------- */
static int rd2(const uint8_t* p) { return (p[0] << 8) | p[1]; }

void gjo_stream_free(gjo_stream* s)
{
    free(s->seg_offset); free(s->seg_size); free(s->seg_scan); free(s->seg_index_in_scan);
    s->seg_offset = s->seg_size = NULL;
    s->seg_scan = s->seg_index_in_scan = NULL;
}

static int push_segment(gjo_stream* s, int* cap, size_t off, size_t size, int scan, int idx)
{
    if (s->seg_count == *cap) {
        *cap = *cap ? *cap * 2 : 1024;
        s->seg_offset = (size_t*)realloc(s->seg_offset, *cap * sizeof(size_t));
        s->seg_size = (size_t*)realloc(s->seg_size, *cap * sizeof(size_t));
        s->seg_scan = (int*)realloc(s->seg_scan, *cap * sizeof(int));
        s->seg_index_in_scan = (int*)realloc(s->seg_index_in_scan, *cap * sizeof(int));
        if (!s->seg_offset || !s->seg_size || !s->seg_scan || !s->seg_index_in_scan) return -1;
    }
    s->seg_offset[s->seg_count] = off;
    s->seg_size[s->seg_count] = size;
    s->seg_scan[s->seg_count] = scan;
    s->seg_index_in_scan[s->seg_count] = idx;
    s->seg_count++;
    return 0;
}

int gjo_parse(const uint8_t* jpeg, size_t size, int req_pf, int req_cs, gjo_stream* s)
{
    memset(s, 0, sizeof *s);
    gjo_image* img = &s->img;
    const uint8_t *p = jpeg, *end = jpeg + size;
    if (size < 4 || p[0] != 0xFF || p[1] != 0xD8) return -1;
    p += 2;
    int header_cs = GJO_CS_NONE, cs_internal = GJO_CS_BT601_256 /* decoder default, decoder.c:110 */, header_adobe = 0;
    int have_sof = 0, scans = 0, cap = 0, comps_seen = 0;
    int restart = 0;
    for (;;) {
        if (end - p < 2 || p[0] != 0xFF) return -1;
        while (p < end && *p == 0xFF) p++;          /* tolerate fill bytes */
        if (p >= end) return -1;
        int marker = *p++;
        if (marker == 0xD9) break;
        if (marker == 0xD8) continue;               /* SPIFF's second SOI */
        if (end - p < 2) return -1;
        int len = rd2(p);
        if (len < 2 || p + len > end) return -1;
        const uint8_t* d = p + 2;
        switch (marker) {
        case 0xE0: /* APP0 JFIF => full-range BT.601 (reader.c:1379-1383) */
            if (len >= 7 && memcmp(d, "JFIF", 5) == 0) header_cs = GJO_CS_BT601_256;
            break;
        case 0xEE: /* APP14 Adobe (reader.c:563-640): transform 0 => RGB (3 comps), 1 => YCbCr */
            if (len >= 14 && memcmp(d, "Adobe", 5) == 0) {
                int transform = d[11];
                header_adobe = 1;
                header_cs = transform == 0 ? GJO_CS_RGB : (transform == 1 ? GJO_CS_BT601_256 : GJO_CS_NONE);
            }
            break;
        case 0xE8: /* APP8 SPIFF (reader.c:449-556) */
            if (len >= 32 && memcmp(d, "SPIFF", 6) == 0) {
                switch (d[18]) {
                case 1: header_cs = GJO_CS_BT709; break;
                case 3: case 8: header_cs = GJO_CS_BT601_256; break;
                case 4: header_cs = GJO_CS_BT601; break;
                case 10: header_cs = GJO_CS_RGB; break;
                default: header_cs = GJO_CS_NONE; break;
                }
            }
            break;
        case 0xFE: /* COM: FFmpeg/GPUJPEG "CS=ITU601" => limited-range BT.601 (reader.c:642-680) */
            if (len >= 2 + 9 && memcmp(d, "CS=ITU601", 9) == 0) header_cs = GJO_CS_BT601;
            break;
        case 0xDB: { /* DQT */
            int l = len - 2;
            while (l >= 65) {
                int pq = d[0] >> 4, tq = d[0] & 15;
                if (pq != 0 || tq > 3) return -1;
                memcpy(s->qraw[tq], d + 1, 64);
                for (int i = 0; i < 64; i++) s->qinv[tq][gjo_zigzag[i]] = d[1 + i];
                d += 65; l -= 65;
            }
            break; }
        case 0xC0: case 0xC1: { /* SOF0 / SOF1 */
            if (d[0] != 8) return -1;
            img->height = rd2(d + 1);
            img->width = rd2(d + 3);
            img->comp_count = d[5];
            if (img->comp_count < 1 || img->comp_count > 4) return -1;
            for (int c = 0; c < img->comp_count; c++) {
                s->comp_id[c] = d[6 + 3 * c];
                img->samp_h[c] = d[7 + 3 * c] >> 4;
                img->samp_v[c] = d[7 + 3 * c] & 15;
                s->qmap[c] = d[8 + 3 * c];
            }
            if (header_cs != GJO_CS_NONE) cs_internal = header_cs;
            else if (img->comp_count >= 3) { /* component-id deduction, reader.c:748-785 */
                if (s->comp_id[0] == 1 && s->comp_id[1] == 2 && s->comp_id[2] == 3) cs_internal = GJO_CS_BT601_256;
                else if ((s->comp_id[0] == 'R' && s->comp_id[1] == 'G' && s->comp_id[2] == 'B') ||
                         (s->comp_id[0] == 'r' && s->comp_id[1] == 'g' && s->comp_id[2] == 'b')) cs_internal = GJO_CS_RGB;
            }
            if (header_adobe && cs_internal == GJO_CS_RGB && img->comp_count == 1) cs_internal = GJO_CS_BT601_256;
            have_sof = 1;
            break; }
        case 0xC4: { /* DHT */
            int l = len - 2;
            while (l > 17) {
                int tc = d[0] >> 4, th = d[0] & 15;
                if (tc > 1 || th > 3) return -1;
                int count = 0;
                s->hbits[th][tc][0] = 0;
                for (int i = 1; i <= 16; i++) { s->hbits[th][tc][i] = d[i]; count += d[i]; }
                if (count > 256 || 17 + count > l) return -1;
                memcpy(s->hvals[th][tc], d + 17, count);
                d += 17 + count; l -= 17 + count;
            }
            break; }
        case 0xDD: restart = rd2(d); break;
        case 0xDA: { /* SOS */
            if (!have_sof) return -1;
            int n = d[0];
            if (len != 6 + 2 * n) return -1;
            if (scans == 0) {
                img->interleaved = (n == 1) ? 0 : 1;
                if (n != 1 && n != img->comp_count) return -1;
                img->restart_interval = restart;
                img->color_space_internal = cs_internal;
                /* output format resolution, reader.c:1591-1618 */
                int cs = req_cs, pf = req_pf;
                if (cs == GJO_CS_NONE) cs = cs_internal;
                if (cs < 0) cs = (pf == GJO_PF_U8 || (pf < 0 && img->comp_count == 1)) ? GJO_CS_BT601_256 : GJO_CS_RGB;
                if (pf < 0) pf = img->comp_count == 1 ? GJO_PF_U8 : (img->comp_count == 3 ? GJO_PF_444_P012 : GJO_PF_4444_P0123);
                img->color_space = cs;
                img->pixel_format = pf;
                img->quality = 0;
                if (gjo_image_init(img) != 0) return -1;
            }
            for (int i = 0; i < n; i++) {
                int id = d[1 + 2 * i], tab = d[2 + 2 * i], ci = -1;
                for (int c = 0; c < img->comp_count; c++) if (s->comp_id[c] == id) { ci = c; break; }
                if (ci < 0) return -1;
                s->hmap[ci][0] = tab >> 4;
                s->hmap[ci][1] = tab & 15;
            }
            comps_seen += n;
            p += len;
            /* scan content by parsing, reader.c:1039-1155 */
            const uint8_t* seg_start = p;
            int idx = 0;
            for (;;) {
                const uint8_t* f = (const uint8_t*)memchr(p, 0xFF, (size_t)(end - p));
                if (!f || f + 1 >= end) return -1;
                int m = f[1];
                p = f + 2;
                if (m == 0) continue;
                if (m >= 0xD0 && m <= 0xD7) {
                    if (push_segment(s, &cap, (size_t)(seg_start - jpeg), (size_t)(f - seg_start), scans, idx++)) return -1;
                    seg_start = p;
                    continue;
                }
                if (m == 0xFF) { p = f + 1; continue; }
                /* any other marker ends the scan */
                if (f - seg_start > 0 || idx == 0)
                    if (push_segment(s, &cap, (size_t)(seg_start - jpeg), (size_t)(f - seg_start), scans, idx++)) return -1;
                p = f;
                break;
            }
            scans++;
            continue; /* p already positioned at next marker */
        }
        default: break;
        }
        p += len;
    }
    (void)comps_seen;
    return (have_sof && scans > 0) ? 0 : -1;
}

/* canonical Huffman decode tables, ITU T.81 F.2.2.3 (src/gpujpeg_table.c:384-449) */
typedef struct { int mincode[17], maxcode[18], valptr[17]; const uint8_t* vals; } dec_table;

static void build_dec_table(const uint8_t bits[17], const uint8_t* vals, dec_table* t)
{
    int code = 0, p = 0;
    for (int l = 1; l <= 16; l++) {
        if (bits[l]) {
            t->valptr[l] = p;
            t->mincode[l] = code;
            p += bits[l];
            code += bits[l];
            t->maxcode[l] = code - 1;
        } else {
            t->maxcode[l] = -1;
            t->mincode[l] = 0;
            t->valptr[l] = 0;
        }
        code <<= 1;
    }
    t->maxcode[17] = 0xFFFFF;
    t->vals = vals;
}

typedef struct { const uint8_t* p; const uint8_t* end; uint32_t acc; int bits; } bitr;

static int get_bit(bitr* r)
{
    if (r->bits == 0) {
        int b = 0; /* past the end: zero bits (src/gpujpeg_huffman_cpu_decoder.c:80-118 pads) */
        if (r->p < r->end) {
            b = *r->p++;
            if (b == 0xFF && r->p < r->end && *r->p == 0) r->p++;
        }
        r->acc = (uint32_t)b;
        r->bits = 8;
    }
    r->bits--;
    return (r->acc >> r->bits) & 1;
}

static int get_bits(bitr* r, int n) { int v = 0; while (n--) v = (v << 1) | get_bit(r); return v; }

static int decode_symbol(bitr* r, const dec_table* t)
{
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | get_bit(r);
        if (t->maxcode[l] >= 0 && code <= t->maxcode[l] && code >= t->mincode[l]) return t->vals[t->valptr[l] + code - t->mincode[l]];
    }
    return 0;
}

static int extend(int v, int n) { return v < (1 << (n - 1)) ? v + (int)((~0u) << n) + 1 : v; } /* F.2.2.1 */

int gjo_huffman_decode(const gjo_stream* s, const uint8_t* jpeg, int16_t* coefs)
{
    const gjo_image* img = &s->img;
    memset(coefs, 0, img->data_size * sizeof(int16_t));
    dec_table t[4][2];
    for (int th = 0; th < 4; th++)
        for (int tc = 0; tc < 2; tc++) build_dec_table(s->hbits[th][tc], s->hvals[th][tc], &t[th][tc]);
    /* map parsed segments to geometric segments: segment order = scan order, index in scan */
    int base = 0, prev_scan = -1, scan_base[GJO_MAX_COMP] = {0, 0, 0, 0};
    if (!img->interleaved) for (int c = 1; c < img->comp_count; c++) scan_base[c] = scan_base[c - 1] + img->comp[c - 1].segment_count;
    (void)base; (void)prev_scan;
    for (int i = 0; i < s->seg_count; i++) {
        int gi = (img->interleaved ? 0 : scan_base[s->seg_scan[i]]) + s->seg_index_in_scan[i];
        if (gi >= img->segment_count) continue;
        gjo_segment seg;
        gjo_segment_get(img, gi, &seg);
        bitr r = {jpeg + s->seg_offset[i], jpeg + s->seg_offset[i] + s->seg_size[i], 0, 0};
        int dc[GJO_MAX_COMP] = {0, 0, 0, 0};
        int nblk = gjo_segment_block_count(img, &seg);
        for (int k = 0; k < nblk; k++) {
            int c;
            size_t off = gjo_segment_block(img, &seg, k, &c);
            int16_t* blk = coefs + off;
            const dec_table* tdc = &t[s->hmap[c][0]][0];
            const dec_table* tac = &t[s->hmap[c][1]][1];
            int sz = decode_symbol(&r, tdc);
            int diff = sz ? extend(get_bits(&r, sz), sz) : 0;
            dc[c] += diff;
            blk[0] = (int16_t)dc[c];
            for (int kk = 1; kk < 64;) {
                int rs = decode_symbol(&r, tac);
                int run = rs >> 4, sz2 = rs & 15;
                if (sz2 == 0) {
                    if (run == 15) { kk += 16; continue; }
                    break; /* EOB */
                }
                kk += run;
                if (kk > 63) break;
                blk[gjo_zigzag[kk]] = (int16_t)extend(get_bits(&r, sz2), sz2);
                kk++;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Dequantisation + inverse DCT -- src/gpujpeg_dct_gpu.cu:312-366 (1-D lifting IDCT), :472-618 (kernel)
 * Fusion map: DESIGN.md section 3.
 * ---------------------------------------------------------------------------------------------- */
static void idct8(float v[8])
{
    const float k0 = 0.4142135623f, k1 = 0.3535533905f, k2 = 0.4619397662f, k3 = 0.1989123673f, k4 = 0.7071067811f;
    const float a2 = v[2] * 0.5411961f, a4 = v[4] * 0.509795579f, a5 = v[5] * 0.601344887f;
    const float t1 = v[0] - v[1];
    const float b1 = t1 * k1;
    const float b0 = fmaf(v[0], k4, -b1);
    const float b3 = fmaf(a2, k1, v[3] * k2);
    const float b2 = fmaf(b3, k0, -a2);
    const float b6 = fmaf(a5, k2, v[6] * k0);
    const float b5 = fmaf(b6, -0.6681786379f, a5);
    const float b7 = fmaf(a4, k3, v[7] * 0.49039264f);
    const float b4 = fmaf(b7, k3, -a4);
    const float c1 = fmaf(t1, k1, b2);
    const float c2 = fmaf(-2.0f, b2, c1);
    const float c4 = b5 + b4;
    const float c5 = fmaf(2.0f, b5, -c4);
    const float c7 = b6 + b7;
    const float c6 = fmaf(-2.0f, b6, c7);
    const float c0 = b3 + b0;
    const float c3 = fmaf(-2.0f, b3, c0);
    const float d5 = fmaf(c6, k0, c5);
    const float d6 = fmaf(d5, -k4, c6);
    const float e5 = fmaf(d6, k0, d5);
    const float d3 = c3 + c4;
    const float e4 = fmaf(-2.0f, c4, d3);
    const float d2 = c2 + e5;
    const float f5 = fmaf(-2.0f, e5, d2);
    const float e1 = d6 + c1;
    const float e6 = fmaf(-2.0f, d6, e1);
    const float e0 = c0 + c7;
    const float e7 = fmaf(-2.0f, c7, e0);
    v[0] = e0; v[1] = e1; v[2] = d2; v[3] = d3; v[4] = e4; v[5] = f5; v[6] = e6; v[7] = e7;
}

void gjo_idct_block(const int16_t in[64], const uint16_t q[64], uint8_t* dst, int stride)
{
    static const int perm[8] = {0, 4, 6, 2, 7, 5, 3, 1};   /* :532-539 */
    float d[8][8];
    for (int i = 0; i < 64; i++) d[i / 8][i % 8] = (float)((int)in[i] * (int)q[i]);   /* :497-500 */
    for (int c = 0; c < 8; c++) {
        float x[8];
        for (int k = 0; k < 8; k++) x[k] = d[perm[k]][c];
        idct8(x);
        for (int k = 0; k < 8; k++) d[k][c] = x[k];
    }
    for (int r = 0; r < 8; r++) {
        float x[8];
        for (int k = 0; k < 8; k++) x[k] = d[r][perm[k]];
        idct8(x);
        for (int i = 0; i < 8; i++) {
            int save = (int)rintf(x[i] + 128.0f);   /* :608-611 */
            dst[r * stride + i] = clamp_u8(save);
        }
    }
}

void gjo_idct(const gjo_stream* s, const int16_t* coefs, uint8_t* planes)
{
    const gjo_image* img = &s->img;
    for (int c = 0; c < img->comp_count; c++) {
        const gjo_comp* k = &img->comp[c];
        int bw = k->data_width / 8, bh = k->data_height / 8;
        for (int by = 0; by < bh; by++)
            for (int bx = 0; bx < bw; bx++)
                gjo_idct_block(coefs + k->data_offset + ((size_t)by * bw + bx) * 64, s->qinv[s->qmap[c]],
                               planes + k->data_offset + (size_t)by * 8 * k->data_width + bx * 8, k->data_width);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Postprocessor -- src/gpujpeg_postprocessor.cu:49-217,:318-347,:445-498 ; stores: preprocessor_common.cuh:118-203
 * ---------------------------------------------------------------------------------------------- */
static int decode_no_transform(const gjo_image* img) /* src/gpujpeg_postprocessor.cu:318-347 */
{
    if (img->comp_count >= 3 && img->color_space != img->color_space_internal) return 0;
    int h[4], v[4];
    gjo_pixfmt_sampling(img->pixel_format, h, v);
    if (gjo_pixfmt_comp_count(img->pixel_format) != img->comp_count) return 0;
    for (int i = 0; i < img->comp_count; i++)
        if (img->comp[i].h != h[i] || img->comp[i].v != v[i]) return 0;
    return 1;
}

void gjo_postprocess(const gjo_image* img, const uint8_t* planes, uint8_t* raw)
{
    const int pf = img->pixel_format;
    if ((pixfmt_is_planar(pf) || pf == GJO_PF_U8) && decode_no_transform(img)) {
        size_t off = 0;
        for (int c = 0; c < img->comp_count; c++) {
            const gjo_comp* k = &img->comp[c];
            int dpitch = k->width + img->width_padding;
            for (int y = 0; y < k->height; y++)
                memcpy(raw + off + (size_t)y * dpitch, planes + k->data_offset + (size_t)y * k->data_width, k->width);
            off += (size_t)dpitch * k->height;
        }
        return;
    }
    int W = img->width, H = img->height;
    if (pf == GJO_PF_422_P1020) W = round_up_div(img->width, 2) * 2;
    /* colour-space selection :371-390: identical spaces => no transform */
    int cs_from = img->color_space_internal, cs_to = img->color_space;
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            uint8_t r[4] = {0, 0, 0, 0};
            if (pf == GJO_PF_4444_P0123) r[3] = 0xFF;                      /* pre_load :116-126 */
            for (int c = 0; c < img->comp_count; c++) {
                const gjo_comp* k = &img->comp[c];
                int sh = img->max_h / k->h, sv = img->max_v / k->v;
                r[c] = planes[k->data_offset + (size_t)(y / sv) * k->data_width + x / sh];  /* nearest neighbour :72-76 */
            }
            if (img->comp_count == 1) {                                    /* post_load / fill_ch_2_3 :127-170 */
                if (cs_from == GJO_CS_RGB) r[1] = r[2] = r[0];
                else r[1] = r[2] = 128;
            }
            if (cs_from != cs_to) gjo_color_transform(cs_from, cs_to, r);
            const int pos = y * W + x;
            switch (pf) {
            case GJO_PF_U8: raw[pos + img->width_padding * y] = r[0]; break;
            case GJO_PF_444_P012: { uint8_t* p = raw + (size_t)pos * 3 + (size_t)img->width_padding * y; p[0] = r[0]; p[1] = r[1]; p[2] = r[2]; break; }
            case GJO_PF_4444_P0123: { uint8_t* p = raw + (size_t)pos * 4 + (size_t)img->width_padding * y; p[0] = r[0]; p[1] = r[1]; p[2] = r[2]; p[3] = r[3]; break; }
            case GJO_PF_444_P0P1P2: raw[pos] = r[0]; raw[W * H + pos] = r[1]; raw[2 * W * H + pos] = r[2]; break;
            case GJO_PF_422_P0P1P2:
                raw[pos] = r[0];
                if ((x % 2) == 0) { raw[W * H + pos / 2] = r[1]; raw[W * H + H * ((W + 1) / 2) + pos / 2] = r[2]; }
                break;
            case GJO_PF_422_P1020: {
                size_t off = (size_t)pos * 2 + (size_t)img->width_padding * y;
                raw[off + 1] = r[0];
                raw[off] = (x % 2) == 0 ? r[1] : r[2];
                break; }
            case GJO_PF_420_P0P1P2:
                raw[pos] = r[0];
                if ((pos % 2) == 0 && (y % 2) == 0) {
                    raw[W * H + y / 2 * ((W + 1) / 2) + x / 2] = r[1];
                    raw[W * H + ((H + 1) / 2 + y / 2) * ((W + 1) / 2) + x / 2] = r[2];
                }
                break;
            default: break;
            }
        }
    }
}

int gjo_decode(const uint8_t* jpeg, size_t size, int req_pf, int req_cs, uint8_t* out, size_t cap, gjo_image* info)
{
    gjo_stream s;
    if (gjo_parse(jpeg, size, req_pf, req_cs, &s) != 0) { gjo_stream_free(&s); return -1; }
    int rc = -1;
    int16_t* coefs = (int16_t*)malloc(s.img.data_size * sizeof(int16_t));
    uint8_t* planes = (uint8_t*)malloc(s.img.data_size + 64);
    if (coefs && planes && s.img.raw_size <= cap) {
        gjo_huffman_decode(&s, jpeg, coefs);
        gjo_idct(&s, coefs, planes);
        gjo_postprocess(&s.img, planes, out);
        rc = 0;
    }
    if (info) *info = s.img;
    free(coefs);
    free(planes);
    gjo_stream_free(&s);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Synthetic inputs -- src/utils/image_delegate.c:562-603
 * ---------------------------------------------------------------------------------------------- */
void gjo_fill_noise(uint8_t* dst, size_t n, unsigned seed)
{
    uint32_t state = seed;
    for (size_t i = 0; i < n; i++) {
        state = (1664525u * state + 1013904223u) % 2147483647u;
        dst[i] = (uint8_t)(state % 256);
    }
}

void gjo_fill_gradient(uint8_t* dst, int width, int height, int bpp)
{
    size_t linesize = (size_t)width * bpp;
    for (int i = 0; i < height; i++) memset(dst + (size_t)i * linesize, i * 255 / height, linesize);
}

/* ---- options that act on the raw image / the planes ---- */

/* vertical flip of every padded component plane, rows 0 .. data_height-1 (src/gpujpeg_preprocessor.cu:455-486:
 * the kernel swaps row y with data_height-1-y, i.e. padding rows take part) */
void gjo_flip_planes(const gjo_image* img, uint8_t* planes)
{
    for (int c = 0; c < img->comp_count; c++) {
        const gjo_comp* k = &img->comp[c];
        uint8_t* p = planes + k->data_offset;
        for (int y = 0; y < k->data_height / 2; y++) {
            uint8_t* a = p + (size_t)y * k->data_width;
            uint8_t* b = p + (size_t)(k->data_height - 1 - y) * k->data_width;
            for (int x = 0; x < k->data_width; x++) { const uint8_t t = a[x]; a[x] = b[x]; b[x] = t; }
        }
    }
}

/* "XYZ"/"XYZW" -> packed mapping (src/gpujpeg_encoder.c:661-699): nibble i = source channel of output channel i,
 * 4 = all ones ('F'), 5 = all zeros ('Z'); bits 24.. = number of channels. Returns 0 for an invalid string. */
unsigned gjo_parse_channel_remap(const char* val)
{
    const int n = (int)strlen(val);
    if (n == 0 || n > GJO_MAX_COMP) return 0;
    unsigned map = 0;
    for (int i = n - 1; i >= 0; i--) {
        int src = val[i] - '0';
        if (val[i] == 'F') src = 4;
        else if (val[i] == 'Z') src = 5;
        else if (src < 0 || src >= n) return 0;
        map = (map << 4) | (unsigned)src;
    }
    return map | ((unsigned)n << 24);
}

/* in-place channel permutation of the raw image, pixel by pixel in raster order
 * (src/gpujpeg_preprocessor.cu:488-559; __byte_perm(val, 0xFF, map): selector 0-3 = channel, 4 = 0xFF, 5-7 = 0).
 * Restated for the formats whose pixels do not share samples (packed 4:4:4 / 4:4:4:4, planar 4:4:4, grey). */
int gjo_channel_remap(const gjo_image* img, uint8_t* raw, unsigned channel_remap)
{
    const int pf = img->pixel_format;
    if (pf != GJO_PF_444_P012 && pf != GJO_PF_4444_P0123 && pf != GJO_PF_444_P0P1P2 && pf != GJO_PF_U8) return -1;
    if ((int)(channel_remap >> 24) != gjo_pixfmt_comp_count(pf)) return -1;
    const unsigned map = channel_remap & 0xFFFF;
    const int W = img->width, H = img->height;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t pos = (size_t)y * W + x;
            uint8_t* ch[4] = {0, 0, 0, 0};
            switch (pf) {
            case GJO_PF_U8: ch[0] = raw + pos + (size_t)img->width_padding * y; break;
            case GJO_PF_444_P012: for (int c = 0; c < 3; c++) ch[c] = raw + pos * 3 + (size_t)img->width_padding * y + c; break;
            case GJO_PF_4444_P0123: for (int c = 0; c < 4; c++) ch[c] = raw + pos * 4 + (size_t)img->width_padding * y + c; break;
            default: for (int c = 0; c < 3; c++) ch[c] = raw + (size_t)c * W * H + pos; break;
            }
            /* the loaders fill missing channels like raw_load does: grey -> (v, 128, 128, 0), three channels -> w = 0 */
            uint8_t in[8] = {ch[0] ? *ch[0] : 0, ch[1] ? *ch[1] : (uint8_t)(pf == GJO_PF_U8 ? 128 : 0), ch[2] ? *ch[2] : (uint8_t)(pf == GJO_PF_U8 ? 128 : 0),
                             ch[3] ? *ch[3] : 0, 0xFF, 0, 0, 0};
            for (int c = 0; c < 4; c++)
                if (ch[c]) *ch[c] = in[(map >> (4 * c)) & 7];
        }
    return 0;
}
