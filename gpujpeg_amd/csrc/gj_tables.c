/*
 * gj_tables.c -- quantisation tables, the typical Huffman tables of ITU T.81 Annex K, and the
 * lookup tables the gfx950 kernels consume. Host counterpart of src/gpujpeg_table.c.
 */
#include <string.h>

#include "gj_internal.h"

/* zig-zag scan: position -> natural index (ITU T.81 figure A.6; reference gpujpeg_order_natural, src/gpujpeg_table.h:73-84) */
const uint8_t gj_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                               41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                               30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* ITU T.81 tables K.1 (luminance) and K.2 (chrominance) in natural order */
static const uint8_t k1_luminance[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                         14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                         18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                         49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t k2_chrominance[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                           99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                           99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

void gj_quant_table_raw(int type, int quality, uint8_t raw[64]) /* src/gpujpeg_table.c:35-100 */
{
    const uint8_t* base = type == GJ_LUMA ? k1_luminance : k2_chrominance;
    if (quality <= 0) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality; /* IJG quality curve */
    for (int i = 0; i < 64; i++) {
        int v = (scale * (int)base[gj_zigzag[i]] + 50) / 100;
        raw[i] = (uint8_t)(v < 1 ? 1 : (v > 255 ? 255 : v));
    }
}

void gj_quant_table_forward(const uint8_t raw[64], float fwd[64]) /* src/gpujpeg_table.c:103-123 */
{
    /* output scaling of the 1-D AAN transform, folded with the 2-D gain of 8 into the quantiser */
    static const double aan[8] = {1.0, 1.387039845, 1.306562965, 1.175875602, 1.0, 0.785694958, 0.541196100, 0.275899379};
    for (int i = 0; i < 64; i++) {
        const int x = gj_zigzag[i] % 8, y = gj_zigzag[i] / 8;
        fwd[x * 8 + y] = (float)(1.0 / (raw[i] * aan[x] * aan[y] * 8));
    }
}

void gj_quant_table_inverse(const uint8_t raw[64], uint16_t inv[64]) /* src/gpujpeg_table.c:154-160 */
{
    for (int i = 0; i < 64; i++) inv[gj_zigzag[i]] = raw[i];
}

/* ---- ITU T.81 Annex K.3.3: typical Huffman tables (BITS, HUFFVAL) ---- */
static const uint8_t dc_bits[2][17] = {{0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0}};
static const uint8_t dc_vals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t ac_bits[2][17] = {{0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d}, {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77}};
static const uint8_t ac_vals[2][162] = {
    {0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
     0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
     0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
     0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
     0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
     0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
     0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
     0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa},
    {0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
     0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
     0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
     0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
     0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
     0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
     0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
     0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa}};

void gj_huffman_std_spec(int type, int is_ac, const uint8_t** bits, const uint8_t** vals, int* count)
{
    if (is_ac) {
        *bits = ac_bits[type];
        *vals = ac_vals[type];
        *count = 162;
    } else {
        *bits = dc_bits[type];
        *vals = dc_vals;
        *count = 12;
    }
}

/* symbol -> (code << 8) | size, canonical assignment of ITU T.81 Annex C (figures C.1-C.3).
 * Table order for the kernels: luminance DC, luminance AC, chrominance DC, chrominance AC. */
void gj_huffman_encoder_lut(uint32_t lut[4 * 256])
{
    memset(lut, 0, 4 * 256 * sizeof(uint32_t));
    for (int type = 0; type < 2; type++) {
        for (int is_ac = 0; is_ac < 2; is_ac++) {
            const uint8_t *bits, *vals;
            int count;
            gj_huffman_std_spec(type, is_ac, &bits, &vals, &count);
            uint32_t* t = lut + (type * 2 + is_ac) * 256;
            unsigned code = 0;
            int p = 0;
            for (int len = 1; len <= 16; len++) {
                for (int i = 0; i < bits[len]; i++) t[vals[p++]] = (code++ << 8) | (unsigned)len;
                code <<= 1;
            }
        }
    }
}

/* decode table for one DHT table: see GJ_DEC_TAB_WORDS (gj_hip.h) for the layout */
int gj_huffman_decoder_table(const uint8_t bits[17], const uint8_t* vals, uint16_t out[GJ_DEC_TAB_WORDS])
{
    memset(out, 0, GJ_DEC_TAB_WORDS * sizeof(uint16_t));
    uint16_t* fast = out;
    uint16_t* maxcode = out + 1024;
    uint16_t* valptr = out + 1024 + 36;
    uint16_t* mincode = out + 1024 + 36 + 17;
    uint16_t* symbols = out + 1024 + 36 + 17 + 34;
    int code = 0, p = 0;
    for (int len = 1; len <= 16; len++) {
        int32_t mx = -1, mn = 0;
        if (bits[len]) {
            valptr[len] = (uint16_t)p;
            mn = code;
            for (int i = 0; i < bits[len]; i++, p++, code++) {
                if (p >= 256 || code >= (1 << len)) return -1; /* over-subscribed table: reject before any write */
                if (len <= GJ_DEC_FAST_BITS) { /* every 10-bit prefix starting with this code */
                    const int shift = GJ_DEC_FAST_BITS - len;
                    for (int f = 0; f < (1 << shift); f++) fast[(code << shift) | f] = (uint16_t)((len << 8) | vals[p]);
                }
            }
            mx = code - 1;
        }
        maxcode[2 * len] = (uint16_t)((uint32_t)mx & 0xFFFF);
        maxcode[2 * len + 1] = (uint16_t)((uint32_t)mx >> 16);
        mincode[2 * len] = (uint16_t)((uint32_t)mn & 0xFFFF);
        mincode[2 * len + 1] = (uint16_t)((uint32_t)mn >> 16);
        code <<= 1;
    }
    for (int i = 0; i < p; i++) symbols[i] = vals[i];
    return 0;
}

/* two-level decode table of the sub-sequence decoder: see GJ_DEC2_WORDS (gj_hip.h) for the layout.
 * Returns 0, -1 for an invalid table, 1 when the table needs more second-level tables than the layout holds. */
int gj_huffman_decoder_table2(const uint8_t bits[17], const uint8_t* vals, int is_ac, uint16_t out[GJ_DEC2_WORDS])
{
    /* what a code that is not in the table decodes to: 16 bits consumed, symbol 0 (as the canonical search of the
     * lane-per-segment kernel does) */
    const uint16_t invalid = (uint16_t)(16 | (0 << 5) | ((is_ac ? 63 : 1) << 9));
    for (int i = 0; i < GJ_DEC2_WORDS; i++) out[i] = invalid;
    int subtables = 0;
    int sub_of_prefix[1024];
    for (int i = 0; i < 1024; i++) sub_of_prefix[i] = -1;
    int code = 0, p = 0;
    for (int len = 1; len <= 16; len++) {
        for (int i = 0; i < bits[len]; i++, p++, code++) {
            if (p >= 256 || code >= (1 << len)) return -1;
            const int sym = vals[p];
            const int run = sym >> 4, sz = sym & 15;
            int adv;
            if (!is_ac) adv = run + 1;
            else if (sz != 0) adv = run + 1;
            else adv = run == 15 ? 16 : 63; /* end of block: from any position inside the block (>= 1) to 64 or beyond */
            const int coefficient = is_ac && sz != 0;
            const uint16_t e = (uint16_t)((len + sz) | (sz << 5) | (adv << 9) | (coefficient << 15));
            if (len <= GJ_DEC_FAST_BITS) {
                const int shift = GJ_DEC_FAST_BITS - len;
                for (int f = 0; f < (1 << shift); f++) out[(code << shift) | f] = e;
            } else {
                const int prefix = code >> (len - GJ_DEC_FAST_BITS);
                if (sub_of_prefix[prefix] < 0) {
                    if (subtables == GJ_DEC2_SUBTABLES) return 1;
                    sub_of_prefix[prefix] = subtables++;
                    out[prefix] = (uint16_t)((1024 + 64 * sub_of_prefix[prefix]) << 5); /* low 5 bits 0: second level */
                }
                const int rest = len - GJ_DEC_FAST_BITS; /* 1..6 bits inside the second level */
                const int low = code & ((1 << rest) - 1);
                uint16_t* sub = out + 1024 + 64 * sub_of_prefix[prefix];
                for (int f = 0; f < (1 << (6 - rest)); f++) sub[(low << (6 - rest)) | f] = e;
            }
        }
        code <<= 1;
    }
    return 0;
}

#include <stdlib.h>

#include "gpujpeg_amd_ext.h"

int gpujpeg_amd_host_huffman_table_check(const uint8_t bits[17], const uint8_t* vals, int is_ac)
{
    uint16_t* t1 = malloc(GJ_DEC_TAB_WORDS * sizeof(uint16_t)); /* exact sizes: an out-of-range write is visible to ASan / valgrind */
    uint16_t* t2 = malloc(GJ_DEC2_WORDS * sizeof(uint16_t));
    int rc = -1;
    if (t1 && t2) {
        const int a = gj_huffman_decoder_table(bits, vals, t1);
        const int b = gj_huffman_decoder_table2(bits, vals, is_ac, t2);
        rc = (a != 0 || b < 0) ? -1 : 0;
    }
    free(t1);
    free(t2);
    return rc;
}
