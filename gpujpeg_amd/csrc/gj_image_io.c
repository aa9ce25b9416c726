/*
 * gj_image_io.c -- image file front-end of the API: gpujpeg_image_load_from_file / save_to_file /
 * get_properties / get_file_format (reference: src/gpujpeg_common.c:380-470,1208-1376 and the delegates
 * in src/utils/image_delegate.c, pam.c, y4m.c).
 *
 * Implemented natively: headerless raw files (.rgb .rgba .yuv .yuva .uyvy .i420 .r .raw), PNM (P5/P6),
 * PAM (P7), Y4M (8-bit 4:4:4 / 4:2:2 / 4:2:0 / mono), the synthetic ".tst" images, BMP and TGA (read and written with the bytes the reference's
 * vendored stb writers produce); PNG (read, written) and GIF (read) live in gj_image_png.c. The reference goes through vendored third-party code for
 * the last four (src/utils/image_delegate.c); nothing of it is used here.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include "gj_internal.h"
#include "gpujpeg_amd_ext.h"

enum gpujpeg_image_file_format gpujpeg_image_get_file_format(const char* filename) /* common.c:380-444 */
{
    static const struct { const char* ext; enum gpujpeg_image_file_format f; } exts[] = {
        {"raw", GPUJPEG_IMAGE_FILE_RAW}, {"rgb", GPUJPEG_IMAGE_FILE_RGB}, {"rgba", GPUJPEG_IMAGE_FILE_RGBA}, {"yuv", GPUJPEG_IMAGE_FILE_YUV},
        {"yuva", GPUJPEG_IMAGE_FILE_YUVA}, {"uyvy", GPUJPEG_IMAGE_FILE_UYVY}, {"i420", GPUJPEG_IMAGE_FILE_I420}, {"r", GPUJPEG_IMAGE_FILE_GRAY},
        {"jpg", GPUJPEG_IMAGE_FILE_JPEG}, {"jpeg", GPUJPEG_IMAGE_FILE_JPEG}, {"jfif", GPUJPEG_IMAGE_FILE_JPEG}, {"bmp", GPUJPEG_IMAGE_FILE_BMP},
        {"gif", GPUJPEG_IMAGE_FILE_GIF}, {"png", GPUJPEG_IMAGE_FILE_PNG}, {"tga", GPUJPEG_IMAGE_FILE_TGA}, {"pnm", GPUJPEG_IMAGE_FILE_PNM},
        {"pgm", GPUJPEG_IMAGE_FILE_PGM}, {"ppm", GPUJPEG_IMAGE_FILE_PPM}, {"pam", GPUJPEG_IMAGE_FILE_PAM}, {"y4m", GPUJPEG_IMAGE_FILE_Y4M},
        {"tst", GPUJPEG_IMAGE_FILE_TST}, {"XXX", GPUJPEG_IMAGE_FILE_RAW}};
    const size_t n = sizeof exts / sizeof exts[0];
    if (strcmp(filename, "help") == 0) {
        fprintf(stderr, "Recognized extensions:\n");
        for (size_t i = 0; i < n; i++)
            if (exts[i].f != GPUJPEG_IMAGE_FILE_RAW) fprintf(stderr, "\t- %s\n", exts[i].ext);
        fprintf(stderr, "\nUse \"help.tst\" (eg. `gpujpegtool help.tst null.jpg`) for test image usage).\n");
        return GPUJPEG_IMAGE_FILE_UNKNOWN;
    }
    if (strcmp(filename, "rawhelp") == 0) {
        fprintf(stderr, "Recognized raw extensions:");
        for (size_t i = 0; i < n; i++)
            if (exts[i].f > GPUJPEG_IMAGE_FILE_RAW) fprintf(stderr, " %s", exts[i].ext);
        fprintf(stderr, "\n");
        return GPUJPEG_IMAGE_FILE_UNKNOWN;
    }
    const char* dot = strrchr(filename, '.');
    if (!dot) return GPUJPEG_IMAGE_FILE_UNKNOWN;
    for (size_t i = 0; i < n; i++)
        if (strcasecmp(dot + 1, exts[i].ext) == 0) return exts[i].f;
    return GPUJPEG_IMAGE_FILE_UNKNOWN;
}

/* ------------------------------------------------------------------ .tst synthetic images (image_delegate.c:383-632) */
enum tst_pattern { TST_GRADIENT, TST_NOISE, TST_RANDOM, TST_BLANK };

static void tst_usage(void)
{
    fprintf(stderr, "Test image usage:\n\t<W>x<H>[.c_<cs>][.p_<pixfmt>][.gradient|.noise|.random[_<seed>]|.blank[_<val>]].tst\n");
}

static int tst_parse(const char* filename, struct gpujpeg_image_parameters* pi, enum tst_pattern* pattern, int* seed, long* blank)
{
    char name[4096];
    snprintf(name, sizeof name, "%s", filename);
    const char* base = strrchr(name, '/');
    char* s = base ? (char*)base + 1 : name;
    char* dot = strrchr(s, '.');
    if (!dot) return -1;
    *dot = '\0';
    char* endp = s;
    pi->width = (int)strtoul(s, &endp, 10);
    if (*endp != 'x') { tst_usage(); return -1; }
    pi->height = (int)strtoul(endp + 1, &endp, 10);
    if (pi->height == 0) { tst_usage(); return -1; }
    pi->color_space = GPUJPEG_RGB;
    pi->pixel_format = GPUJPEG_444_U8_P012;
    pi->width_padding = 0;
    *pattern = TST_GRADIENT;
    *seed = 12345;
    *blank = 0;
    char* save = NULL;
    for (char* item = strtok_r(endp, ".", &save); item; item = strtok_r(NULL, ".", &save)) {
        const char* us = strchr(item, '_');
        if (strncmp(item, "c_", 2) == 0) {
            pi->color_space = gpujpeg_color_space_by_name(us + 1);
            if (pi->color_space == GPUJPEG_NONE) { GJ_ERROR("[tst] Unknown color space: %s\n", us + 1); return -1; }
        } else if (strncmp(item, "p_", 2) == 0) {
            pi->pixel_format = gpujpeg_pixel_format_by_name(us + 1);
            if (pi->pixel_format == GPUJPEG_PIXFMT_NONE) { GJ_ERROR("[tst] Unknown pixel format: %s\n", us + 1); return -1; }
        } else if (strcmp(item, "noise") == 0) *pattern = TST_NOISE;
        else if (strncmp(item, "random", 6) == 0) { *pattern = TST_RANDOM; if (us) *seed = atoi(us + 1); }
        else if (strncmp(item, "blank", 5) == 0) { *pattern = TST_BLANK; if (us) *blank = strtol(us + 1, NULL, 0); }
        else if (strcmp(item, "gradient") == 0) *pattern = TST_GRADIENT;
        else { GJ_ERROR("[tst] unknown test image option: %s!\n", item); return -1; }
    }
    return 0;
}

static int tst_load(const char* filename, uint8_t** image, size_t* size)
{
    struct gpujpeg_image_parameters pi;
    enum tst_pattern pattern;
    int seed;
    long blank;
    if (tst_parse(filename, &pi, &pattern, &seed, &blank) != 0) return -1;
    *size = gpujpeg_image_calculate_size(&pi);
    uint8_t* data = gj_hip_host_alloc(*size);
    if (!data) return -1;
    switch (pattern) {
    case TST_GRADIENT: {
        struct gpujpeg_image_parameters line = pi;
        line.height = 1;
        const size_t linesize = gpujpeg_image_calculate_size(&line);
        for (int i = 0; i < pi.height; i++) memset(data + (size_t)i * linesize, i * 255 / pi.height, linesize);
        break; }
    case TST_NOISE:
        for (size_t i = 0; i < *size; i++) data[i] = (uint8_t)(rand() % 256);
        break;
    case TST_RANDOM: { /* LCG of image_delegate.c:562-582 */
        uint32_t state = (uint32_t)seed;
        for (size_t i = 0; i < *size; i++) {
            state = (1664525u * state + 1013904223u) % 2147483647u;
            data[i] = (uint8_t)(state % 256);
        }
        break; }
    case TST_BLANK: memset(data, (int)blank, *size); break;
    }
    *image = data;
    return 0;
}

/* ------------------------------------------------------------------ PNM / PAM
 * What is accepted, and with which result, follows the reference's reader to the letter (src/utils/pam.c:46-78 PAM header, :88-140 PNM
 * header, :168-232 pam_read; tests/test_file_formats_vs_ref.py feeds both libraries the same files):
 *  - the magic is the first three bytes: "P7\n" exactly, or 'P', a type and ONE white-space byte; P1..P3 (plain) are refused, P4 is parsed
 *    (and then refused for its single level);
 *  - PNM: decimal numbers separated by white space, `#` starts a comment wherever a number is expected, and the byte behind the last number
 *    must be a NEW LINE (a file with CR LF line ends is refused);
 *  - PAM: lines of at most 126 bytes; "ENDHDR\n" ends the header and so does any line without a blank (a file without ENDHDR "parses", its
 *    samples then come up short); lines that begin with `#` and unknown keys are skipped;
 *  - width, height, depth > 0, 0 < maxval <= 65535, and this library (like the reference's delegate, image_delegate.c:186-211) takes 255
 *    levels and 1, 3 or 4 channels only. */
struct pnm_header { int width, height, depth, maxval; bool bitmap; };

static int pnm_read_header(FILE* f, struct pnm_header* hd)
{
    memset(hd, 0, sizeof *hd);
    char magic[4] = {0};
    if (!fgets(magic, sizeof magic, f)) magic[0] = '\0';
    if (strcmp(magic, "P7\n") == 0) {
        char line[128];
        while (fgets(line, sizeof line - 1, f)) {
            if (strcmp(line, "ENDHDR\n") == 0) break;
            if (line[0] == '#') continue;
            char* blank = strchr(line, ' ');
            if (!blank) break;
            *blank = '\0';
            const int v = atoi(blank + 1);
            if (strcmp(line, "WIDTH") == 0) hd->width = v;
            else if (strcmp(line, "HEIGHT") == 0) hd->height = v;
            else if (strcmp(line, "DEPTH") == 0) hd->depth = v;
            else if (strcmp(line, "MAXVAL") == 0) hd->maxval = v;
            else if (strcmp(line, "TUPLTYPE") != 0) fprintf(stderr, "unrecognized key %s in PAM header\n", line);
        }
    } else if (strlen(magic) == 3 && magic[0] == 'P' && isspace((unsigned char)magic[2])) {
        switch (magic[1]) {
        case '1': case '2': case '3': fprintf(stderr, "Plain (ASCII) PNM are not supported, input is P%c\n", magic[1]); return -1;
        case '4': hd->depth = 1; hd->maxval = 1; hd->bitmap = true; break;
        case '5': hd->depth = 1; break;
        case '6': hd->depth = 3; break;
        default: fprintf(stderr, "Wrong PNM type P%c\n", magic[1]); return -1;
        }
        int item = 0;
        bool complete = false;
        while (!complete && !feof(f) && !ferror(f)) {
            int v = 0;
            if (fscanf(f, "%d", &v) == 1) {
                if (item == 0) hd->width = v;
                else if (item == 1) hd->height = v;
                else hd->maxval = v;
                item++;
                complete = item == (hd->bitmap ? 2 : 3);
            } else if (getc(f) == '#') {
                int ch;
                while ((ch = getc(f)) != '\n' && ch != EOF) {}
            } else {
                break;
            }
        }
        if (!complete) { fprintf(stderr, "Problem parsing PNM header, number of hdr items successfully read: %d\n", item); return -1; }
        if (getc(f) != '\n') { fprintf(stderr, "PNM maximal value isn't immediately followed by <NL>\n"); return -1; }
    } else {
        return -1;
    }
    if (hd->width <= 0 || hd->height <= 0 || hd->depth <= 0 || hd->maxval <= 0 || hd->maxval > 65535) return -1;
    return 0;
}

static enum gpujpeg_pixel_format depth_pixfmt(int depth)
{
    return depth == 1 ? GPUJPEG_U8 : depth == 3 ? GPUJPEG_444_U8_P012 : depth == 4 ? GPUJPEG_4444_U8_P0123 : GPUJPEG_PIXFMT_NONE;
}

static int pnm_probe(const char* filename, struct gpujpeg_image_parameters* pi, int file_exists)
{
    if (!file_exists) { /* output file: what the extension can hold (src/utils/image_delegate.c:157-182) */
        const enum gpujpeg_image_file_format fmt = gpujpeg_image_get_file_format(filename);
        pi->pixel_format = fmt == GPUJPEG_IMAGE_FILE_PGM ? GPUJPEG_U8 : fmt == GPUJPEG_IMAGE_FILE_PPM ? GPUJPEG_444_U8_P012
                           : fmt == GPUJPEG_IMAGE_FILE_PNM ? GPUJPEG_PIXFMT_NO_ALPHA : GPUJPEG_PIXFMT_AUTODETECT;
        pi->color_space = fmt == GPUJPEG_IMAGE_FILE_PGM ? GPUJPEG_YCBCR_JPEG : GPUJPEG_CS_DEFAULT;
        return 1;
    }
    FILE* f = fopen(filename, "rb");
    if (!f) { GJ_ERROR("Failed open %s for reading: %s\n", filename, strerror(errno)); return -1; }
    struct pnm_header hd;
    const int rc = pnm_read_header(f, &hd);
    fclose(f);
    if (rc != 0) { GJ_ERROR("File '%s' doesn't seem to be valid PAM or PNM.\n", filename); return -1; }
    if (hd.maxval != 255) { GJ_ERROR("PAM/PNM image %s reports %d levels but only 255 are currently supported!\n", filename, hd.maxval); return -1; }
    if (depth_pixfmt(hd.depth) == GPUJPEG_PIXFMT_NONE) { GJ_ERROR("Unsupported PAM/PNM component count %d!\n", hd.depth); return -1; }
    pi->width = hd.width;
    pi->height = hd.height;
    pi->color_space = hd.depth == 1 ? GPUJPEG_YCBCR_JPEG : GPUJPEG_RGB;
    pi->pixel_format = depth_pixfmt(hd.depth);
    return 0;
}

static int pnm_load(const char* filename, uint8_t** image, size_t* size)
{
    FILE* f = fopen(filename, "rb");
    if (!f) { GJ_ERROR("Failed open %s for reading: %s\n", filename, strerror(errno)); return -1; }
    struct pnm_header hd;
    if (pnm_read_header(f, &hd) != 0 || hd.maxval != 255 || depth_pixfmt(hd.depth) == GPUJPEG_PIXFMT_NONE) { /* (the reference aborts, image_delegate.c:151) */
        fclose(f);
        GJ_ERROR("Unsupported PNM/PAM file %s\n", filename);
        return -1;
    }
    const size_t n = (size_t)hd.width * hd.height * hd.depth;
    if (*size != 0 && *size != n) GJ_WARN("Image size mismatch: expected %zu, file has %zu bytes\n", *size, n);
    uint8_t* data = gj_hip_host_alloc(n);
    if (!data || fread(data, 1, n, f) != n) { fclose(f); gj_hip_host_free(data); GJ_ERROR("Failed to load image data [%zu bytes] from file %s!\n", n, filename); return -1; }
    fclose(f);
    *image = data;
    *size = n;
    return 0;
}

/* src/utils/image_delegate.c:214-253 + src/utils/pam.c:234-294: grey, RGB and (PAM only) RGBA of 255 levels; anything but grey must be RGB */
static int pnm_save(const char* filename, enum gpujpeg_image_file_format fmt, const uint8_t* image, const struct gpujpeg_image_parameters* pi)
{
    if (pi->pixel_format != GPUJPEG_U8 && pi->color_space != GPUJPEG_RGB) {
        GJ_ERROR("Wrong color space %s for PAM!\n", gpujpeg_color_space_get_name(pi->color_space));
        return -1;
    }
    const int depth = pi->pixel_format == GPUJPEG_U8 ? 1 : pi->pixel_format == GPUJPEG_444_U8_P012 ? 3 : pi->pixel_format == GPUJPEG_4444_U8_P0123 ? 4 : 0;
    if (depth == 0) {
        GJ_ERROR("Wrong pixel format %s for PAM/PNM! Only packed formats without subsampling are supported.\n", gpujpeg_pixel_format_get_name(pi->pixel_format));
        return -1;
    }
    FILE* f = fopen(filename, "wb");
    if (!f) { GJ_ERROR("Failed open %s for writing: %s\n", filename, strerror(errno)); return -1; }
    if (fmt == GPUJPEG_IMAGE_FILE_PAM) {
        fprintf(f, "P7\nWIDTH %d\nHEIGHT %d\nDEPTH %d\nMAXVAL 255\nTUPLTYPE %s\nENDHDR\n", pi->width, pi->height, depth,
                depth == 1 ? "GRAYSCALE" : depth == 3 ? "RGB" : "RGB_ALPHA");
    } else {
        if (depth == 4) { /* (the reference creates the file first and leaves it empty, pam.c:246-251) */
            GJ_ERROR("Only 1 or 3 channels supported for PNM!\n");
            fclose(f);
            return -1;
        }
        fprintf(f, "P%d\n%d %d\n255\n", depth == 1 ? 5 : 6, pi->width, pi->height);
    }
    const size_t line = (size_t)pi->width * depth;
    for (int y = 0; y < pi->height; y++) fwrite(image + (size_t)y * (line + pi->width_padding), 1, line, f);
    fclose(f);
    return 0;
}

/* ------------------------------------------------------------------ Y4M (8-bit planar)
 * Reader after src/utils/y4m.c:42-75, 99-160: white-space separated tokens -- "YUV4MPEG2", then W<n>, H<n>, C<chroma>, X... until the token
 * FRAME, which must be followed by a NEW LINE at once (frame parameters are refused). The chroma tag is required: "mono", "444alpha" (refused
 * by the delegate, image_delegate.c:289-291) or a number with an optional p<depth> (420, 422, 444; "420jpeg" reads as 420; more than 8 bits are
 * refused). The range is FULL unless XCOLORRANGE=LIMITED says otherwise. */
static int y4m_read_header(FILE* f, int* w, int* h, enum gpujpeg_pixel_format* pf, bool* limited)
{
    char item[129];
    if (fscanf(f, "%128s", item) != 1 || strcmp(item, "YUV4MPEG2") != 0) return -1;
    int subsampling = 0, depth = 0;
    bool alpha = false;
    *w = *h = 0;
    *limited = false;
    while (fscanf(f, " %128s", item) == 1 && strcmp(item, "FRAME") != 0) {
        if (item[0] == 'W') *w = atoi(item + 1);
        else if (item[0] == 'H') *h = atoi(item + 1);
        else if (item[0] == 'C') {
            depth = 8;
            alpha = false;
            if (strcmp(item + 1, "444alpha") == 0) alpha = true;
            else if (strncmp(item + 1, "mono", 4) == 0) { subsampling = 400; sscanf(item + 1, "mono%d", &depth); }
            else if (sscanf(item + 1, "%dp%d", &subsampling, &depth) == 0) return -1;
        } else if (strcmp(item, "XCOLORRANGE=LIMITED") == 0) *limited = true;
    }
    if (getc(f) != '\n') return -1; /* behind FRAME */
    if (alpha) { GJ_ERROR("[y4m] Planar YCbCr with alpha is not currently supported!\n"); return -1; }
    switch (subsampling) {
    case 400: *pf = GPUJPEG_U8; break;
    case 420: *pf = GPUJPEG_420_U8_P0P1P2; break;
    case 422: *pf = GPUJPEG_422_U8_P0P1P2; break;
    case 444: *pf = GPUJPEG_444_U8_P0P1P2; break;
    default: fprintf(stderr, "Unsupported subsampling '%d'\n", subsampling); return -1;
    }
    if (depth != 8) { GJ_ERROR("Currently only 8-bit Y4M pictures are supported but the file has %d bits!\n", depth); return -1; }
    return 0;
}

static int y4m_probe(const char* filename, struct gpujpeg_image_parameters* pi, int file_exists)
{
    if (!file_exists) { /* src/utils/image_delegate.c:259-263 */
        pi->color_space = GPUJPEG_YCBCR_BT601_256LVLS;
        pi->pixel_format = GPUJPEG_PIXFMT_STD;
        return 0;
    }
    FILE* f = fopen(filename, "rb");
    if (!f) { GJ_ERROR("Failed open %s for reading: %s\n", filename, strerror(errno)); return -1; }
    bool limited;
    int w, h;
    enum gpujpeg_pixel_format pf;
    const int rc = y4m_read_header(f, &w, &h, &pf, &limited);
    fclose(f);
    if (rc != 0) { GJ_ERROR("Unsupported Y4M file %s\n", filename); return -1; }
    pi->width = w;
    pi->height = h;
    pi->pixel_format = pf;
    pi->color_space = limited ? GPUJPEG_YCBCR_BT601 : GPUJPEG_YCBCR_BT601_256LVLS; /* src/utils/image_delegate.c:297 */
    return 0;
}

static int y4m_load(const char* filename, uint8_t** image, size_t* size)
{
    FILE* f = fopen(filename, "rb");
    if (!f) { GJ_ERROR("Failed open %s for reading: %s\n", filename, strerror(errno)); return -1; }
    struct gpujpeg_image_parameters pi = gpujpeg_default_image_parameters();
    bool limited;
    if (y4m_read_header(f, &pi.width, &pi.height, &pi.pixel_format, &limited) != 0 || pi.width <= 0 || pi.height <= 0) {
        fclose(f);
        GJ_ERROR("Unsupported Y4M file %s\n", filename);
        return -1;
    }
    const size_t n = gpujpeg_image_calculate_size(&pi);
    uint8_t* data = gj_hip_host_alloc(n);
    if (!data || fread(data, 1, n, f) != n) { fclose(f); gj_hip_host_free(data); GJ_ERROR("Failed to load image data [%zu bytes] from file %s!\n", n, filename); return -1; }
    fclose(f);
    *image = data;
    *size = n;
    return 0;
}

/* src/utils/image_delegate.c:308-338 + src/utils/y4m.c:162-209: planar YCbCr (any colour space but RGB; limited range unless it is the
 * full-range BT.601 one), one frame */
static int y4m_save(const char* filename, const uint8_t* image, size_t size, const struct gpujpeg_image_parameters* pi)
{
    if (pi->color_space == GPUJPEG_RGB) { GJ_ERROR("Y4M cannot use RGB colorspace!\n"); return -1; }
    const char* c;
    switch (pi->pixel_format) {
    case GPUJPEG_444_U8_P0P1P2: c = "444"; break;
    case GPUJPEG_422_U8_P0P1P2: c = "422"; break;
    case GPUJPEG_420_U8_P0P1P2: c = "420"; break;
    case GPUJPEG_U8: c = "mono"; break;
    default: GJ_ERROR("Wrong pixel format %s for Y4M! Only planar formats are supported.\n", gpujpeg_pixel_format_get_name(pi->pixel_format)); return -1;
    }
    FILE* f = fopen(filename, "wb");
    if (!f) { GJ_ERROR("Failed open %s for writing: %s\n", filename, strerror(errno)); return -1; }
    fprintf(f, "YUV4MPEG2 W%d H%d F25:1 Ip A0:0 C%s XCOLORRANGE=%s\nFRAME\n", pi->width, pi->height, c,
            pi->color_space == GPUJPEG_YCBCR_JPEG ? "FULL" : "LIMITED");
    fwrite(image, 1, size, f);
    fclose(f);
    return 0;
}

/* ------------------------------------------------------------------ BMP and TGA
 * The reference hands these to third-party stb_image / stb_image_write (src/utils/image_delegate.c:188-330): 1, 3 or 4 channels
 * of 8 bits <-> u8 / 444-u8-p012 / 4444-u8-p0123. Written here from the format specifications: BMP uncompressed 8 (grey palette),
 * 24 and 32 bit, bottom-up or top-down; TGA types 2/3 and their run-length coded forms 10/11, both origins. PNG (read, write) and
 * GIF (read) live in gj_image_png.c. */
/* struct gj_raster (gj_internal.h): top-down, RGB(A) or grey, tightly packed */

static unsigned rd16(const uint8_t* p) { return (unsigned)p[0] | (unsigned)p[1] << 8; }
static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

static int file_read_all(const char* filename, uint8_t** data, size_t* n)
{
    FILE* f = fopen(filename, "rb");
    if (!f) { GJ_ERROR("Failed open %s for reading: %s\n", filename, strerror(errno)); return -1; }
    fseek(f, 0, SEEK_END);
    *n = (size_t)ftell(f);
    rewind(f);
    *data = malloc(*n ? *n : 1);
    if (!*data || fread(*data, 1, *n, f) != *n) { fclose(f); free(*data); GJ_ERROR("Failed to read %s\n", filename); return -1; }
    fclose(f);
    return 0;
}

/* header only when out->px == NULL on return is acceptable (probe): pass want_pixels = 0 */
static int bmp_decode(const uint8_t* d, size_t n, struct gj_raster* out, int want_pixels)
{
    if (n < 54 || d[0] != 'B' || d[1] != 'M') return -1;
    const uint32_t off = rd32(d + 10), hsize = rd32(d + 14);
    if (hsize < 40) return -1;
    const int w = (int)rd32(d + 18);
    int h = (int)rd32(d + 22);
    const unsigned bpp = rd16(d + 28);
    const uint32_t compression = rd32(d + 30);
    const int top_down = h < 0;
    if (top_down) h = -h;
    if (w <= 0 || h <= 0 || (bpp != 8 && bpp != 24 && bpp != 32) || (compression != 0 && !(compression == 3 && bpp == 32))) return -1;
    out->w = w; out->h = h; out->comps = bpp == 8 ? 1 : bpp == 24 ? 3 : 4;
    if (!want_pixels) return 0;
    const size_t pitch = (((size_t)w * bpp + 31) / 32) * 4;
    if ((size_t)off + pitch * h > n) return -1;
    const uint8_t* pal = d + 14 + hsize; /* 8 bit: BGRA palette; only grey ramps map to one channel, others are converted to their green */
    out->px = malloc((size_t)w * h * out->comps);
    if (!out->px) return -1;
    for (int y = 0; y < h; y++) {
        const uint8_t* row = d + off + pitch * (size_t)(top_down ? y : h - 1 - y);
        uint8_t* o = out->px + (size_t)y * w * out->comps;
        for (int x = 0; x < w; x++) {
            if (bpp == 8) o[x] = (size_t)(pal - d) + 4u * row[x] + 2 < n ? pal[4 * row[x] + 1] : row[x];
            else if (bpp == 24) { o[3 * x] = row[3 * x + 2]; o[3 * x + 1] = row[3 * x + 1]; o[3 * x + 2] = row[3 * x]; }
            else { o[4 * x] = row[4 * x + 2]; o[4 * x + 1] = row[4 * x + 1]; o[4 * x + 2] = row[4 * x]; o[4 * x + 3] = row[4 * x + 3]; }
        }
    }
    return 0;
}

static int tga_decode(const uint8_t* d, size_t n, struct gj_raster* out, int want_pixels)
{
    if (n < 18) return -1;
    const unsigned idlen = d[0], cmap = d[1], type = d[2], bpp = d[16], desc = d[17];
    const int w = (int)rd16(d + 12), h = (int)rd16(d + 14);
    if (cmap != 0 || w <= 0 || h <= 0) return -1;
    const int rle = type == 10 || type == 11;
    if (!((type == 2 || type == 10) && (bpp == 24 || bpp == 32)) && !((type == 3 || type == 11) && bpp == 8)) return -1;
    out->w = w; out->h = h; out->comps = bpp / 8;
    if (!want_pixels) return 0;
    const int c = out->comps;
    out->px = calloc((size_t)w * h, (size_t)c); /* a truncated file leaves zeros, never heap contents */
    if (!out->px) return -1;
    const uint8_t* p = d + 18 + idlen;
    const uint8_t* e = d + n;
    const size_t total = (size_t)w * h;
    size_t i = 0;
    uint8_t* tmp = out->px; /* file order first, flipped afterwards */
    while (i < total) {
        size_t run = 1;
        int repeat = 0;
        if (rle) {
            if (p >= e) break;
            repeat = *p & 0x80;
            run = (size_t)(*p & 0x7F) + 1;
            p++;
        } else run = total;
        if (run > total - i) run = total - i;
        for (size_t k = 0; k < run; k++) {
            if (p + c > e) { i = total; break; }
            uint8_t* o = tmp + (i + k) * c;
            if (c == 1) o[0] = p[0];
            else { o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; if (c == 4) o[3] = p[3]; }
            if (!repeat) p += c;
        }
        if (repeat) p += c;
        i += run;
    }
    if (!(desc & 0x20)) /* bottom-left origin: flip to top-down */
        for (int y = 0; y < h / 2; y++)
            for (size_t x = 0; x < (size_t)w * c; x++) {
                uint8_t* a = out->px + (size_t)y * w * c + x;
                uint8_t* b = out->px + (size_t)(h - 1 - y) * w * c + x;
                const uint8_t t = *a; *a = *b; *b = t;
            }
    return 0;
}

static const char* raster_name(enum gpujpeg_image_file_format fmt)
{
    return fmt == GPUJPEG_IMAGE_FILE_BMP ? "BMP" : fmt == GPUJPEG_IMAGE_FILE_TGA ? "TGA" : fmt == GPUJPEG_IMAGE_FILE_PNG ? "PNG" : "GIF";
}

static int raster_decode(enum gpujpeg_image_file_format fmt, const uint8_t* d, size_t n, struct gj_raster* r, int want_pixels)
{
    switch (fmt) {
    case GPUJPEG_IMAGE_FILE_BMP: return bmp_decode(d, n, r, want_pixels);
    case GPUJPEG_IMAGE_FILE_TGA: return tga_decode(d, n, r, want_pixels);
    case GPUJPEG_IMAGE_FILE_PNG: return gj_png_decode(d, n, r, want_pixels);
    case GPUJPEG_IMAGE_FILE_GIF: return gj_gif_decode(d, n, r, want_pixels);
    default: return -1;
    }
}

static int raster_probe(const char* filename, enum gpujpeg_image_file_format fmt, struct gpujpeg_image_parameters* pi, int file_exists)
{
    if (!file_exists) { /* output (src/utils/image_delegate.c:523-527): whatever the stream holds, in the default colour space */
        pi->pixel_format = GPUJPEG_PIXFMT_AUTODETECT;
        pi->color_space = GPUJPEG_CS_DEFAULT;
        return 1;
    }
    uint8_t* d;
    size_t n;
    if (file_read_all(filename, &d, &n) != 0) return -1;
    struct gj_raster r = {0};
    const int rc = raster_decode(fmt, d, n, &r, 0);
    free(d);
    if (rc != 0) { GJ_ERROR("Unsupported %s file %s\n", raster_name(fmt), filename); return -1; }
    if (r.comps != 1 && r.comps != 3 && r.comps != 4) { /* src/utils/image_delegate.c:548-551 */
        GJ_ERROR("[stbi] Unsupported channel count %d for %s\n", r.comps, filename);
        return -1;
    }
    pi->width = r.w;
    pi->height = r.h;
    pi->color_space = r.comps == 1 ? GPUJPEG_YCBCR_JPEG : GPUJPEG_RGB;
    pi->pixel_format = depth_pixfmt(r.comps);
    return 0;
}

static int raster_load(const char* filename, enum gpujpeg_image_file_format fmt, uint8_t** image, size_t* size)
{
    uint8_t* d;
    size_t n;
    if (file_read_all(filename, &d, &n) != 0) return -1;
    struct gj_raster r = {0};
    const int rc = raster_decode(fmt, d, n, &r, 1);
    free(d);
    if (rc != 0 || (r.comps != 1 && r.comps != 3 && r.comps != 4)) { free(r.px); GJ_ERROR("Unsupported or damaged %s file %s\n", raster_name(fmt), filename); return -1; }
    const size_t bytes = (size_t)r.w * r.h * r.comps;
    uint8_t* data = gj_hip_host_alloc(bytes);
    if (!data) { free(r.px); return -1; }
    memcpy(data, r.px, bytes);
    free(r.px);
    *image = data;
    *size = bytes;
    return 0;
}

/* gpujpeg_amd_ext.h: the decoders above without the pinned allocation of gpujpeg_image_load_from_file (no device needed) */
int gpujpeg_amd_read_raster_file(const char* filename, uint8_t* dst, size_t capacity, int* width, int* height, int* channels)
{
    const enum gpujpeg_image_file_format fmt = gpujpeg_image_get_file_format(filename);
    uint8_t* d;
    size_t n;
    if (file_read_all(filename, &d, &n) != 0) return -1;
    struct gj_raster r = {0};
    const int rc = raster_decode(fmt, d, n, &r, dst != NULL);
    free(d);
    if (rc != 0) { free(r.px); return -1; }
    if (width) *width = r.w;
    if (height) *height = r.h;
    if (channels) *channels = r.comps;
    int ret = 0;
    if (dst) {
        const size_t bytes = (size_t)r.w * r.h * r.comps;
        if (bytes <= capacity) memcpy(dst, r.px, bytes); else ret = -1;
    }
    free(r.px);
    return ret;
}

static void wr16(uint8_t* p, unsigned v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void wr32(uint8_t* p, uint32_t v) { wr16(p, v & 0xFFFF); wr16(p + 2, v >> 16); }

/* BMP and TGA as the reference writes them -- it hands both to stb_image_write (src/utils/image_delegate.c:476-515), and a drop-in CLI
 * should leave the same files behind (tests/test_file_formats_vs_ref.py compares the bytes). Restated from that writer's behaviour
 * (src/utils/stb_image_write.h:492-507, 532-612):
 *   BMP  bottom-up, B G R; one and three channels as 24 bits with rows padded to four bytes behind a 14 + 40 byte header whose size-of-image
 *        and resolution fields are zero; four channels as 32 bits with BI_BITFIELDS masks in a 108-byte V4 header;
 *   TGA  run-length coded (types 10 / 11), bottom-up, origin bottom-left, B G R (A); a row is cut into packets of at most 128 pixels: a
 *        run packet where a pixel repeats, otherwise a raw packet that is extended while pixel k differs from pixel k - 2 (sic) and
 *        gives its last pixel back when it does not.
 * The channel count is the pixel format's component count, as in the reference (a planar 4:4:4 image is written as if it were packed);
 * formats whose buffer is smaller than width x height x components are refused (the reference reads past the buffer). */
static void tga_pixel(FILE* f, const uint8_t* d, int c)
{
    if (c == 1) { fputc(d[0], f); return; }
    fputc(d[2], f); fputc(d[1], f); fputc(d[0], f);
    if (c == 4) fputc(d[3], f);
}

static int raster_save(const char* filename, enum gpujpeg_image_file_format fmt, const uint8_t* image, const struct gpujpeg_image_parameters* pi)
{
    if (fmt == GPUJPEG_IMAGE_FILE_GIF) { /* src/utils/image_delegate.c:492-495 */
        GJ_ERROR("[stbi] Only gif decoder is present, the encoder is not supported!\n");
        return -1;
    }
    const int c = gpujpeg_pixel_format_get_comp_count(pi->pixel_format);
    const int w = pi->width, h = pi->height;
    if (c != 1 && c != 3 && c != 4) { GJ_ERROR("Pixel format %s cannot be stored in this file type\n", gpujpeg_pixel_format_get_name(pi->pixel_format)); return -1; }
    if (gpujpeg_image_calculate_size((struct gpujpeg_image_parameters*)pi) < (size_t)w * h * c) {
        GJ_ERROR("Pixel format %s (subsampled) cannot be stored in this file type\n", gpujpeg_pixel_format_get_name(pi->pixel_format));
        return -1;
    }
    const size_t spitch = (size_t)w * c + pi->width_padding;
    if (fmt == GPUJPEG_IMAGE_FILE_PNG) {
        if (gj_png_save(filename, image, w, h, c, spitch) != 0) {
            GJ_ERROR("[stbi] Cannot write output file %s\n", filename);
            return -1;
        }
        return 0;
    }
    FILE* f = fopen(filename, "wb");
    if (!f) { GJ_ERROR("[stbi] Cannot write output file %s: %s\n", filename, strerror(errno)); return -1; }
    if (fmt == GPUJPEG_IMAGE_FILE_BMP) {
        const int pad = c == 4 ? 0 : (-w * 3) & 3;
        const uint32_t hdr_size = c == 4 ? 108u : 40u, data_size = c == 4 ? (uint32_t)w * h * 4u : (uint32_t)(w * 3 + pad) * (uint32_t)h;
        uint8_t hdr[14 + 108] = {'B', 'M'};
        wr32(hdr + 2, 14u + hdr_size + data_size);
        wr32(hdr + 10, 14u + hdr_size);
        wr32(hdr + 14, hdr_size);
        wr32(hdr + 18, (uint32_t)w);
        wr32(hdr + 22, (uint32_t)h);
        wr16(hdr + 26, 1);
        wr16(hdr + 28, c == 4 ? 32 : 24);
        if (c == 4) { /* BI_BITFIELDS: red, green, blue, alpha masks */
            wr32(hdr + 30, 3);
            wr32(hdr + 54, 0x00FF0000u); wr32(hdr + 58, 0x0000FF00u); wr32(hdr + 62, 0x000000FFu); wr32(hdr + 66, 0xFF000000u);
        }
        fwrite(hdr, 1, 14 + hdr_size, f);
        const uint8_t zero[4] = {0, 0, 0, 0};
        for (int y = h - 1; y >= 0; y--) {
            const uint8_t* s = image + (size_t)y * spitch;
            for (int x = 0; x < w; x++) {
                const uint8_t* d = s + (size_t)x * c;
                if (c == 1) { fputc(d[0], f); fputc(d[0], f); fputc(d[0], f); }
                else { fputc(d[2], f); fputc(d[1], f); fputc(d[0], f); if (c == 4) fputc(d[3], f); }
            }
            fwrite(zero, 1, (size_t)pad, f);
        }
    } else {
        uint8_t hdr[18] = {0};
        hdr[2] = (uint8_t)((c == 1 ? 3 : 2) + 8);
        wr16(hdr + 12, (unsigned)w);
        wr16(hdr + 14, (unsigned)h);
        hdr[16] = (uint8_t)(c * 8);
        hdr[17] = c == 4 ? 8 : 0;
        fwrite(hdr, 1, 18, f);
        for (int y = h - 1; y >= 0; y--) {
            const uint8_t* row = image + (size_t)y * spitch;
            int len;
            for (int i = 0; i < w; i += len) {
                const uint8_t* begin = row + (size_t)i * c;
                bool raw = true;
                len = 1;
                if (i < w - 1) {
                    len = 2;
                    raw = memcmp(begin, begin + c, (size_t)c) != 0;
                    for (int k = i + 2; k < w && len < 128; k++) {
                        const uint8_t* px = row + (size_t)k * c;
                        if (raw) {
                            if (memcmp(px - 2 * (size_t)c, px, (size_t)c) != 0) len++;
                            else { len--; break; }
                        } else {
                            if (memcmp(begin, px, (size_t)c) == 0) len++;
                            else break;
                        }
                    }
                }
                if (raw) {
                    fputc(len - 1, f);
                    for (int k = 0; k < len; k++) tga_pixel(f, begin + (size_t)k * c, c);
                } else {
                    fputc((uint8_t)(len - 129), f);
                    tga_pixel(f, begin, c);
                }
            }
        }
    }
    fclose(f);
    return 0;
}

/* ------------------------------------------------------------------ API front-end */
int gpujpeg_image_load_from_file(const char* filename, uint8_t** image, size_t* image_size) /* common.c:1217-1253 */
{
    const enum gpujpeg_image_file_format fmt = gpujpeg_image_get_file_format(filename);
    switch (fmt) {
    case GPUJPEG_IMAGE_FILE_TST: return tst_load(filename, image, image_size);
    case GPUJPEG_IMAGE_FILE_PGM: case GPUJPEG_IMAGE_FILE_PPM: case GPUJPEG_IMAGE_FILE_PNM: case GPUJPEG_IMAGE_FILE_PAM:
        return pnm_load(filename, image, image_size);
    case GPUJPEG_IMAGE_FILE_Y4M: return y4m_load(filename, image, image_size);
    case GPUJPEG_IMAGE_FILE_BMP: case GPUJPEG_IMAGE_FILE_TGA: case GPUJPEG_IMAGE_FILE_GIF: case GPUJPEG_IMAGE_FILE_PNG:
        return raster_load(filename, fmt, image, image_size);
    default: break;
    }
    FILE* f = fopen(filename, "rb");
    if (!f) { GJ_ERROR("Failed open %s for reading: %s\n", filename, strerror(errno)); return -1; }
    if (*image_size == 0) {
        fseek(f, 0, SEEK_END);
        *image_size = (size_t)ftell(f);
        rewind(f);
    }
    uint8_t* data = gj_hip_host_alloc(*image_size);
    if (!data) { fclose(f); GJ_ERROR("Initialize host buffer failed: %s\n", gj_hip_last_error()); return -1; }
    if (fread(data, 1, *image_size, f) != *image_size) {
        GJ_ERROR("Failed to load image data [%zu bytes] from file %s!\n", *image_size, filename);
        fclose(f);
        gj_hip_host_free(data);
        return -1;
    }
    fclose(f);
    *image = data;
    return 0;
}

int gpujpeg_image_save_to_file(const char* filename, const uint8_t* image, size_t image_size, const struct gpujpeg_image_parameters* pi)
{ /* common.c:1275-1310 */
    char* dot = strrchr(filename, '.');
    if (dot && strcmp(dot, ".XXX") == 0 && pi) { /* the caller guarantees a writable string in this case */
        const char* ext = (pi->pixel_format != GPUJPEG_U8 && pi->color_space != GPUJPEG_RGB) ? "y4m" : pi->pixel_format == GPUJPEG_4444_U8_P0123 ? "pam" : "pnm";
        strcpy(dot + 1, ext);
    }
    const enum gpujpeg_image_file_format fmt = gpujpeg_image_get_file_format(filename);
    if (pi) {
        switch (fmt) {
        case GPUJPEG_IMAGE_FILE_PGM: case GPUJPEG_IMAGE_FILE_PPM: case GPUJPEG_IMAGE_FILE_PNM: case GPUJPEG_IMAGE_FILE_PAM:
            return pnm_save(filename, fmt, image, pi);
        case GPUJPEG_IMAGE_FILE_Y4M: return y4m_save(filename, image, image_size, pi);
        case GPUJPEG_IMAGE_FILE_BMP: case GPUJPEG_IMAGE_FILE_TGA: case GPUJPEG_IMAGE_FILE_GIF: case GPUJPEG_IMAGE_FILE_PNG:
            return raster_save(filename, fmt, image, pi);
        default: break;
        }
    }
    FILE* f = fopen(filename, "wb");
    if (!f) { GJ_ERROR("Failed open %s for writing: %s\n", filename, strerror(errno)); return -1; }
    if (fwrite(image, 1, image_size, f) != image_size) {
        GJ_ERROR("Failed to write image data [%zu bytes] to file %s!\n", image_size, filename);
        fclose(f);
        return -1;
    }
    fclose(f);
    return 0;
}

int gpujpeg_image_get_properties(const char* filename, struct gpujpeg_image_parameters* pi, int file_exists) /* common.c:1312-1370 */
{
    const enum gpujpeg_image_file_format fmt = gpujpeg_image_get_file_format(filename);
    switch (fmt) {
    case GPUJPEG_IMAGE_FILE_UNKNOWN: GJ_ERROR("GPUJPEG_IMAGE_FILE_UNKNOWN should not be passed!\n"); return -1;
    case GPUJPEG_IMAGE_FILE_JPEG: GJ_ERROR("GPUJPEG_IMAGE_FILE_JPEG should not be passed!\n"); return -1;
    case GPUJPEG_IMAGE_FILE_TST: {
        enum tst_pattern p; int seed; long blank;
        return tst_parse(filename, pi, &p, &seed, &blank);
    }
    case GPUJPEG_IMAGE_FILE_PGM: case GPUJPEG_IMAGE_FILE_PPM: case GPUJPEG_IMAGE_FILE_PNM: case GPUJPEG_IMAGE_FILE_PAM:
        return pnm_probe(filename, pi, file_exists);
    case GPUJPEG_IMAGE_FILE_Y4M: return y4m_probe(filename, pi, file_exists);
    case GPUJPEG_IMAGE_FILE_BMP: case GPUJPEG_IMAGE_FILE_TGA: case GPUJPEG_IMAGE_FILE_GIF: case GPUJPEG_IMAGE_FILE_PNG:
        return raster_probe(filename, fmt, pi, file_exists);
    case GPUJPEG_IMAGE_FILE_RAW: pi->pixel_format = GPUJPEG_PIXFMT_STD; break;
    case GPUJPEG_IMAGE_FILE_GRAY: pi->color_space = GPUJPEG_YCBCR_JPEG; pi->pixel_format = GPUJPEG_U8; break;
    case GPUJPEG_IMAGE_FILE_RGBA: pi->color_space = GPUJPEG_RGB; pi->pixel_format = GPUJPEG_4444_U8_P0123; break;
    case GPUJPEG_IMAGE_FILE_YUVA: pi->color_space = GPUJPEG_YCBCR_JPEG; pi->pixel_format = GPUJPEG_4444_U8_P0123; break;
    case GPUJPEG_IMAGE_FILE_UYVY: pi->color_space = GPUJPEG_YCBCR_JPEG; pi->pixel_format = GPUJPEG_422_U8_P1020; break;
    case GPUJPEG_IMAGE_FILE_I420: pi->color_space = GPUJPEG_YCBCR_JPEG; pi->pixel_format = GPUJPEG_420_U8_P0P1P2; break;
    case GPUJPEG_IMAGE_FILE_RGB: pi->color_space = GPUJPEG_RGB; pi->pixel_format = GPUJPEG_444_U8_P012; break;
    case GPUJPEG_IMAGE_FILE_YUV: pi->color_space = GPUJPEG_YCBCR_JPEG; pi->pixel_format = GPUJPEG_444_U8_P012; break;
    }
    return 1;
}

int gpujpeg_image_destroy(uint8_t* image) /* common.c:1372-1378 */
{
    gj_hip_host_free(image);
    return 0;
}

void gpujpeg_image_range_info(const char* filename, int width, int height, enum gpujpeg_pixel_format pf) /* common.c:1380-1441 */
{
    size_t size = 0;
    uint8_t* data = NULL;
    if (gpujpeg_image_load_from_file(filename, &data, &size) != 0) {
        GJ_ERROR("Failed to load image [%s]!\n", filename);
        return;
    }
    int lo[3] = {256, 256, 256}, hi[3] = {0, 0, 0};
    const size_t pixels = (size_t)width * (size_t)height;
    if (pf == GPUJPEG_444_U8_P012 && size >= pixels * 3) {
        for (size_t i = 0; i < pixels; i++)
            for (int c = 0; c < 3; c++) {
                const int v = data[i * 3 + c];
                if (v < lo[c]) lo[c] = v;
                if (v > hi[c]) hi[c] = v;
            }
    } else if (pf == GPUJPEG_422_U8_P1020 && size >= pixels * 2) {
        for (size_t i = 0; i < pixels; i++) {
            const int y = data[i * 2 + 1], ch = data[i * 2];
            const int c = (i % 2 == 1) ? 1 : 2; /* the reference attributes the odd pixels' byte to component 2, the even ones' to 3 */
            if (y < lo[0]) lo[0] = y;
            if (y > hi[0]) hi[0] = y;
            if (ch < lo[c]) lo[c] = ch;
            if (ch > hi[c]) hi[c] = ch;
        }
    } else {
        fprintf(stderr, "TODO: implement gpujpeg_image_range_info for pixel format %d.", (int)pf);
        gpujpeg_image_destroy(data);
        return;
    }
    printf("Image Samples Range:\n");
    for (int c = 0; c < 3; c++) printf("Component %d: %d - %d\n", c + 1, lo[c], hi[c]);
    gpujpeg_image_destroy(data);
}

int gpujpeg_image_convert(const char* input, const char* output, struct gpujpeg_image_parameters from, struct gpujpeg_image_parameters to)
{
    (void)input; (void)output; (void)from; (void)to;
    GJ_ERROR("gpujpeg_image_convert() is defunct (as in the reference).\n");
    return -1;
}
