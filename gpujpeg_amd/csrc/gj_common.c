/*
 * gj_common.c -- parameters, pixel formats, names, geometry, device management, statistics.
 * Host-side counterpart of the reference's src/gpujpeg_common.c; the behaviour of every exported
 * function follows the reference lines cited next to it.
 */
#define _GNU_SOURCE
#include <assert.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <strings.h>
#include <sys/time.h>
#include <unistd.h>

#include "gj_internal.h"
#include "gpujpeg_amd_ext.h"

/* ------------------------------------------------------------------ logging */
const char* gj_fg_red = "";
const char* gj_fg_yellow = "";
const char* gj_term_reset = "";

void gj_init_term_colors(void) /* src/gpujpeg_common.c:2313-2327 */
{
    static bool done;
    if (done) return;
    done = true;
    if (isatty(fileno(stderr))) {
        gj_fg_red = "\033[31m";
        gj_fg_yellow = "\033[33m";
        gj_term_reset = "\033[0m";
    }
}

/* ------------------------------------------------------------------ version / time */
int gpujpeg_version(void) { return GPUJPEG_VERSION_INT; }

const char* gpujpeg_version_to_string(int version)
{
    static _Thread_local char buf[64];
    snprintf(buf, sizeof buf, "%d.%d.%d", version >> 16, (version >> 8) & 0xFF, version & 0xFF);
    return buf;
}

double gpujpeg_get_time(void) /* src/gpujpeg_common.c:98-103 */
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (double)tv.tv_sec + (double)tv.tv_usec * 1e-6;
}

/* ------------------------------------------------------------------ devices */
struct gpujpeg_devices_info gpujpeg_get_devices_info(void) /* src/gpujpeg_common.c:155-190 */
{
    struct gpujpeg_devices_info info;
    memset(&info, 0, sizeof info);
    int n = gj_hip_device_count();
    if (n > GPUJPEG_MAX_DEVICE_COUNT) {
        GJ_WARN("There are available more devices (%d) than maximum count (%d).\n", n, GPUJPEG_MAX_DEVICE_COUNT);
        n = GPUJPEG_MAX_DEVICE_COUNT;
    }
    info.device_count = n;
    for (int i = 0; i < n; i++) {
        struct gpujpeg_device_info* d = &info.device[i];
        d->id = i;
        gj_hip_device_props(i, d->name, &d->cc_major, &d->cc_minor, &d->global_memory, &d->shared_memory, &d->register_count,
                            &d->multiprocessor_count);
        d->constant_memory = 64 * 1024;
    }
    return info;
}

int gpujpeg_print_devices_info(void) /* src/gpujpeg_common.c:192-217 */
{
    struct gpujpeg_devices_info info = gpujpeg_get_devices_info();
    if (info.device_count == 0) {
        printf("There is no device supporting HIP.\n");
        return -1;
    }
    printf("There %s %d device%s supporting HIP.\n", info.device_count == 1 ? "is" : "are", info.device_count,
           info.device_count == 1 ? "" : "s");
    for (int i = 0; i < info.device_count; i++) {
        const struct gpujpeg_device_info* d = &info.device[i];
        printf("\nDevice #%d: \"%s\"\n", d->id, d->name);
        printf("  Compute capability: %d.%d\n", d->cc_major, d->cc_minor);
        printf("  Total amount of global memory: %zu KiB\n", d->global_memory / 1024);
        printf("  Total amount of shared memory per block: %zu KiB\n", d->shared_memory / 1024);
        printf("  Total number of registers available per block: %d\n", d->register_count);
        printf("  Multiprocessors: %d\n", d->multiprocessor_count);
    }
    return 0;
}

int gpujpeg_init_device(int device_id, int flags) /* src/gpujpeg_common.c:219-287 */
{
    gj_init_term_colors();
    const int n = gj_hip_device_count();
    if (n == 0) {
        GJ_ERROR("No HIP enabled device\n");
        return -1;
    }
    if (device_id < 0 || device_id >= n) {
        GJ_ERROR("Selected device %d is out of bound. Devices on your system are in range %d - %d\n", device_id, 0, n - 1);
        return -1;
    }
    if (flags & GPUJPEG_OPENGL_INTEROPERABILITY) {
        GJ_ERROR("OpenGL interoperability is not available in the MI355X build.\n");
        return -1;
    }
    if (flags & GPUJPEG_INIT_DEV_VERBOSE) {
        char name[256];
        int major, minor, regs, cus, drv = 0, rt = 0;
        size_t gm, sm;
        if (gj_hip_device_props(device_id, name, &major, &minor, &gm, &sm, &regs, &cus) != 0) return -1;
        gj_hip_runtime_version(&drv, &rt);
        printf("HIP driver version:   %d\n", drv);
        printf("HIP runtime version:  %d\n", rt);
        printf("Using Device #%d:       %s (gfx %d.%d, %d CUs)\n", device_id, name, major, minor, cus);
    }
    if (gj_hip_set_device(device_id) != 0) {
        GJ_ERROR("Failed to initialize HIP device: %s\n", gj_hip_last_error());
        return -1;
    }
    /* touch the device so that failures surface here, as the reference does (:274-284) */
    void* probe = gj_hip_malloc(1);
    if (probe == NULL) {
        GJ_ERROR("Failed to initialize HIP device: %s\n", gj_hip_last_error());
        return -1;
    }
    gj_hip_free(probe);
    return 0;
}

void gpujpeg_set_device(int index) { gj_hip_set_device(index); }
void gpujpeg_device_reset(void) { gj_hip_device_reset(); }

/* ------------------------------------------------------------------ parameters */
void gpujpeg_set_default_parameters(struct gpujpeg_parameters* p) /* src/gpujpeg_common.c:292-306 */
{
    memset(p, 0, sizeof *p);
    p->verbose = GPUJPEG_LL_INFO;
    p->quality = 75;
    p->restart_interval = 8;
    for (int c = 0; c < GPUJPEG_MAX_COMPONENT_COUNT; c++) p->sampling_factor[c] = (struct gpujpeg_component_sampling_factor){1, 1};
    p->color_space_internal = GPUJPEG_YCBCR_BT601_256LVLS;
}

struct gpujpeg_parameters gpujpeg_default_parameters(void)
{
    struct gpujpeg_parameters p;
    gpujpeg_set_default_parameters(&p);
    return p;
}

void gpujpeg_parameters_chroma_subsampling(struct gpujpeg_parameters* p, gpujpeg_sampling_factor_t s) /* :316-340 */
{
    p->comp_count = 0;
    int c = 0;
    for (; c < GPUJPEG_MAX_COMPONENT_COUNT; c++) {
        const int h = (s >> 28) & 15, v = (s >> 24) & 15;
        s <<= 8;
        p->sampling_factor[c] = (struct gpujpeg_component_sampling_factor){(uint8_t)h, (uint8_t)v};
        if (h * v == 0) break;
        p->comp_count++;
    }
    for (; c < GPUJPEG_MAX_COMPONENT_COUNT; c++) p->sampling_factor[c] = (struct gpujpeg_component_sampling_factor){0, 0};
}

bool gj_parameters_equal(const struct gpujpeg_parameters* a, const struct gpujpeg_parameters* b) /* :348-368, quality ignored */
{
    if (a->comp_count != b->comp_count || a->restart_interval != b->restart_interval || a->interleaved != b->interleaved ||
        a->segment_info != b->segment_info || a->color_space_internal != b->color_space_internal)
        return false;
    for (int c = 0; c < a->comp_count; c++)
        if (a->sampling_factor[c].horizontal != b->sampling_factor[c].horizontal || a->sampling_factor[c].vertical != b->sampling_factor[c].vertical)
            return false;
    return true;
}

void gpujpeg_image_set_default_parameters(struct gpujpeg_image_parameters* p) /* :371-378 */
{
    p->width = 0;
    p->height = 0;
    p->color_space = GPUJPEG_RGB;
    p->pixel_format = GPUJPEG_444_U8_P012;
    p->width_padding = 0;
}

struct gpujpeg_image_parameters gpujpeg_default_image_parameters(void)
{
    struct gpujpeg_image_parameters p;
    gpujpeg_image_set_default_parameters(&p);
    return p;
}

bool gj_image_parameters_equal(const struct gpujpeg_image_parameters* a, const struct gpujpeg_image_parameters* b)
{
    return a->width == b->width && a->height == b->height && a->color_space == b->color_space && a->pixel_format == b->pixel_format &&
           a->width_padding == b->width_padding;
}

/* ------------------------------------------------------------------ pixel formats (descriptor rows: :140-151) */
static const struct gj_pixfmt_desc {
    enum gpujpeg_pixel_format pf;
    bool planar;
    int comps, bpp;
    const char* name;
    struct gpujpeg_component_sampling_factor sf[GPUJPEG_MAX_COMPONENT_COUNT];
} gj_pixfmts[] = {
    {(enum gpujpeg_pixel_format)(-4), false, 0, 0, "(file standard)", {{0, 0}}},
    {(enum gpujpeg_pixel_format)(-3), false, 0, 0, "(without alpha)", {{0, 0}}},
    {(enum gpujpeg_pixel_format)(-2), false, 0, 0, "(autodetect)", {{0, 0}}},
    {GPUJPEG_PIXFMT_NONE, false, 0, 0, "(unknown)", {{0, 0}}},
    {GPUJPEG_U8, false, 1, 1, "u8", {{1, 1}}},
    {GPUJPEG_444_U8_P012, false, 3, 3, "444-u8-p012", {{1, 1}, {1, 1}, {1, 1}}},
    {GPUJPEG_444_U8_P0P1P2, true, 3, 0, "444-u8-p0p1p2", {{1, 1}, {1, 1}, {1, 1}}},
    {GPUJPEG_422_U8_P1020, false, 3, 2, "422-u8-p1020", {{2, 1}, {1, 1}, {1, 1}}},
    {GPUJPEG_422_U8_P0P1P2, true, 3, 0, "422-u8-p0p1p2", {{2, 1}, {1, 1}, {1, 1}}},
    {GPUJPEG_420_U8_P0P1P2, true, 3, 0, "420-u8-p0p1p2", {{2, 2}, {1, 1}, {1, 1}}},
    {GPUJPEG_4444_U8_P0123, false, 4, 4, "4444-u8-p0123", {{1, 1}, {1, 1}, {1, 1}, {1, 1}}},
};

static const struct gj_pixfmt_desc* pixfmt_desc(enum gpujpeg_pixel_format pf)
{
    for (size_t i = 0; i < sizeof gj_pixfmts / sizeof gj_pixfmts[0]; i++)
        if (gj_pixfmts[i].pf == pf) return &gj_pixfmts[i];
    return NULL;
}

int gpujpeg_pixel_format_get_comp_count(enum gpujpeg_pixel_format pf) { const struct gj_pixfmt_desc* d = pixfmt_desc(pf); return d ? d->comps : 0; }
const char* gpujpeg_pixel_format_get_name(enum gpujpeg_pixel_format pf) { const struct gj_pixfmt_desc* d = pixfmt_desc(pf); return d ? d->name : NULL; }
int gpujpeg_pixel_format_is_planar(enum gpujpeg_pixel_format pf) { const struct gj_pixfmt_desc* d = pixfmt_desc(pf); return d ? d->planar : 0; }
int gj_pixfmt_unit_size(enum gpujpeg_pixel_format pf) { const struct gj_pixfmt_desc* d = pixfmt_desc(pf); return d ? d->bpp : 0; }
const struct gpujpeg_component_sampling_factor* gj_pixfmt_sampling(enum gpujpeg_pixel_format pf) { const struct gj_pixfmt_desc* d = pixfmt_desc(pf); return d ? d->sf : NULL; }
int gj_pixfmt_is_interleaved(enum gpujpeg_pixel_format pf) { return gpujpeg_pixel_format_get_comp_count(pf) > 1 && !gpujpeg_pixel_format_is_planar(pf); }

enum gpujpeg_pixel_format gpujpeg_pixel_format_by_name(const char* name) /* :2031-2042 */
{
    for (size_t i = 0; i < sizeof gj_pixfmts / sizeof gj_pixfmts[0]; i++)
        if (strcmp(gj_pixfmts[i].name, name) == 0) return gj_pixfmts[i].pf;
    if (strcmp(name, "help") == 0) gpujpeg_print_pixel_formats();
    return GPUJPEG_PIXFMT_NONE;
}

void gpujpeg_print_pixel_formats(void)
{
    fprintf(stderr, "                          u8 (grayscale)          420-u8-p0p1p2 (planar 4:2:0)\n"
                    "                          422-u8-p1020 (eg. UYVY) 422-u8-p0p1p2 (planar 4:2:2)\n"
                    "                          444-u8-p012 (eg. RGB)   444-u8-p0p1p2 (planar 4:4:4)\n"
                    "                          4444-u8-p0123 (RGBA)\n");
}

gpujpeg_sampling_factor_t gj_make_sampling_factor(int comp_count, const struct gpujpeg_component_sampling_factor* sf)
{ /* src/gpujpeg_common_internal.h:528-541 */
    gpujpeg_sampling_factor_t v = 0;
    for (int c = 0; c < GPUJPEG_MAX_COMPONENT_COUNT; c++) v = (v << 8) | ((uint32_t)sf[c].horizontal << 4) | sf[c].vertical;
    if (comp_count <= 0) return 0;
    return comp_count >= 4 ? v : v & (0xFFFFFFFFu << (32u - (unsigned)comp_count * 8u));
}

const char* gpujpeg_subsampling_get_name(int comp_count, const struct gpujpeg_component_sampling_factor* sf) /* :1905-1950 */
{
    static _Thread_local char buf[128];
    const int J = 4;
    if (comp_count == 1) return strcpy(buf, "4:0:0");
    if (comp_count == 2 && sf[0].vertical == sf[1].vertical) {
        snprintf(buf, sizeof buf, "4:0:0:%d", J / sf[0].horizontal * sf[1].horizontal);
        return buf;
    }
    /* the reference compares sf[1].vertical with sf[2].HORIZONTAL (:1920); kept */
    if (sf[1].horizontal == sf[2].horizontal && sf[1].vertical == sf[2].horizontal &&
        (comp_count == 3 || (comp_count == 4 && sf[0].vertical == sf[3].vertical))) {
        const int a = J / sf[0].horizontal * sf[1].horizontal;
        const int vert_change = 2 / sf[0].vertical * sf[1].vertical == 2;
        int n = snprintf(buf, sizeof buf, "%d:%d:%d", J, a, a * vert_change);
        if (comp_count == 4) snprintf(buf + n, sizeof buf - n, ":%d", J / sf[0].horizontal * sf[3].horizontal);
        return buf;
    }
    const gpujpeg_sampling_factor_t packed = gj_make_sampling_factor(comp_count, sf);
    if (packed == GPUJPEG_SUBSAMPLING_442) return strcpy(buf, "4:4:2");
    if (packed == GPUJPEG_SUBSAMPLING_421) return strcpy(buf, "4:2:1");
    buf[0] = '\0';
    for (int i = 0; i < comp_count; i++)
        snprintf(buf + strlen(buf), sizeof buf - strlen(buf), "%s%d-%d", i ? ":" : "", sf[i].horizontal, sf[i].vertical);
    return buf;
}

gpujpeg_sampling_factor_t gpujpeg_subsampling_from_name(const char* s) /* :1952-2005 */
{
    if (strcmp(s, "help") == 0) {
        fprintf(stderr, "Set subsampling in usual J:a:b[:alpha] format, eg. 4:2:2. Colons are optional.\n");
        fprintf(stderr, "Non-standard subsamplings 4:4:2 and 4:2:1 are allowed.\n");
        return GPUJPEG_SUBSAMPLING_UNKNOWN;
    }
    int J = 0, a = 0, b = 0, alpha = 0;
    int n = sscanf(s, "%d:%d:%d:%d", &J, &a, &b, &alpha);
    if (n == 1 && J / (1000 == 4 || J / 100 == 4)) { /* digits without ':'; expression kept as in the reference (:1966) */
        if (J / 1000 == 4) { n = 4; alpha = J % 10; J /= 10; }
        else n = 3;
        J -= 400;
        a = J / 10;
        b = J % 10;
        J = 4;
    }
    if (n < 3 || J != 4 || (alpha != 4 && alpha != 0)) return GPUJPEG_SUBSAMPLING_UNKNOWN;
    if (a != b && b != 0) {
        if (a == 4 && b == 2 && alpha == 0) return GPUJPEG_SUBSAMPLING_442;
        if (a == 2 && b == 1 && alpha == 0) return GPUJPEG_SUBSAMPLING_421;
        return GPUJPEG_SUBSAMPLING_UNKNOWN;
    }
    if (a == 0 && b == 0 && alpha == 0) return GPUJPEG_SUBSAMPLING_400;
    struct gpujpeg_component_sampling_factor f[GPUJPEG_MAX_COMPONENT_COUNT] = {{0, 0}};
    f[0].horizontal = (uint8_t)(4 / a);
    f[0].vertical = a == b ? 1 : 2;
    f[1] = f[2] = (struct gpujpeg_component_sampling_factor){1, 1};
    if (alpha == 0) return gj_make_sampling_factor(3, f);
    f[3] = f[0];
    return gj_make_sampling_factor(4, f);
}

const char* gpujpeg_color_space_get_name(enum gpujpeg_color_space cs) /* :2007-2028 */
{
    switch ((int)cs) {
    case GPUJPEG_NONE: return "None";
    case GPUJPEG_RGB: return "RGB";
    case GPUJPEG_YUV: return "YUV";
    case GPUJPEG_YCBCR_BT601: return "YCbCr BT.601 (limtted range)";
    case GPUJPEG_YCBCR_BT601_256LVLS: return "YCbCr BT.601 256 Levels (YCbCr JPEG)";
    case GPUJPEG_YCBCR_BT709: return "YCbCr BT.709 (limited range)";
    case -1: return "(default CS)";
    default: return "Unknown";
    }
}

enum gpujpeg_color_space gpujpeg_color_space_by_name(const char* name) /* :2053-2085 */
{
    static const struct { const char* n; enum gpujpeg_color_space cs; } map[] = {
        {"rgb", GPUJPEG_RGB}, {"yuv", GPUJPEG_YUV}, {"ycbcr", GPUJPEG_YCBCR}, {"ycbcr-jpeg", GPUJPEG_YCBCR_BT601_256LVLS},
        {"ycbcr-bt601", GPUJPEG_YCBCR_BT601}, {"ycbcr-bt709", GPUJPEG_YCBCR_BT709}};
    for (size_t i = 0; i < sizeof map / sizeof map[0]; i++)
        if (strcmp(name, map[i].n) == 0) return map[i].cs;
    if (strcmp(name, "help") == 0)
        fprintf(stderr, "Available color spaces:\n- rgb\n- yuv (deprecated)\n- ycbcr       - same as ycbcr-bt709\n"
                        "- ycbcr-jpeg  - BT.601 full range\n- ycbcr-bt601 - limitted range\n- ycbcr-bt709 - limitted range\n");
    return GPUJPEG_NONE;
}

enum gpujpeg_header_type gpujpeg_header_type_by_name(const char* name) /* :2347-2361 */
{
    if (strcasecmp(name, GPUJPEG_ENC_HDR_VAL_JFIF) == 0) return GPUJPEG_HEADER_JFIF;
    if (strcasecmp(name, GPUJPEG_ENC_HDR_VAL_EXIF) == 0) return GPUJPEG_HEADER_EXIF;
    if (strcasecmp(name, GPUJPEG_ENC_HDR_VAL_ADOBE) == 0) return GPUJPEG_HEADER_ADOBE;
    if (strcasecmp(name, GPUJPEG_ENC_HDR_VAL_SPIFF) == 0) return GPUJPEG_HEADER_SPIFF;
    return GPUJPEG_HEADER_DEFAULT;
}

const char* gpujpeg_header_type_get_name(enum gpujpeg_header_type t)
{
    switch (t) {
    case GPUJPEG_HEADER_DEFAULT: return "undefined";
    case GPUJPEG_HEADER_JFIF: return GPUJPEG_ENC_HDR_VAL_JFIF;
    case GPUJPEG_HEADER_SPIFF: return GPUJPEG_ENC_HDR_VAL_SPIFF;
    case GPUJPEG_HEADER_ADOBE: return GPUJPEG_ENC_HDR_VAL_ADOBE;
    case GPUJPEG_HEADER_EXIF: return GPUJPEG_ENC_HDR_VAL_EXIF;
    }
    abort();
}

const char* gpujpeg_orientation_get_name(struct gpujpeg_orientation o) /* :2380-2393 */
{
    static const char* names[8] = {"normal", "mirror horizontal", "rotated CW 90 deg", "rotated CW 90 and mirrored horizontal",
                                   "rotated 180 deg", "flipped vertical", "rotated CW 270 deg", "rotated CW 270 deg and mirrored horizontal"};
    return names[(o.rotation << 1 | o.flip) & 7];
}

size_t gpujpeg_image_calculate_size(struct gpujpeg_image_parameters* p) /* :1179-1204 */
{
    assert(p->width > 0 && p->height > 0);
    assert(p->width <= 65535 && p->height <= 65535);
    assert(p->width_padding >= 0 && p->width_padding < 10 * 1000 * 1000);
    const int bpp = gj_pixfmt_unit_size(p->pixel_format);
    if (bpp != 0) return ((size_t)p->width + p->width_padding) * p->height * bpp;
    switch (p->pixel_format) {
    case GPUJPEG_444_U8_P0P1P2: return (size_t)p->width * p->height * 3;
    case GPUJPEG_422_U8_P0P1P2: return (size_t)p->width * p->height + (size_t)2 * ((p->width + 1) / 2) * p->height;
    case GPUJPEG_420_U8_P0P1P2: return (size_t)p->width * p->height + (size_t)2 * ((p->width + 1) / 2) * ((p->height + 1) / 2);
    default: assert(0); return 0;
    }
}

/* ------------------------------------------------------------------ geometry */
static int div_up(int a, int b) { return (a + b - 1) / b; }

int gj_geom_init(gj_geom* g, const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* pi, bool encoder)
{
    memset(g, 0, sizeof *g);
    if (param->comp_count < 1 || param->comp_count > GJ_MAX_COMP || pi->width <= 0 || pi->height <= 0) return -1;
    g->width = pi->width;
    g->height = pi->height;
    g->width_padding = pi->width_padding;
    g->pixel_format = pi->pixel_format;
    g->color_space = pi->color_space;
    g->color_space_internal = param->color_space_internal;
    g->comp_count = param->comp_count;
    g->interleaved = param->interleaved ? 1 : 0;
    g->restart_interval = param->restart_interval;
    g->raw_width = pi->pixel_format == GPUJPEG_422_U8_P1020 ? (pi->width + 1) & ~1 : pi->width; /* preprocessor.cu:369-373 */
    struct gpujpeg_image_parameters tmp = *pi;
    g->raw_size = gpujpeg_image_calculate_size(&tmp);

    uint64_t offset = 0;
    int first_segment = 0;
    for (int c = 0; c < g->comp_count; c++) { /* common.c:675-742; maxima accumulate in component order like the reference */
        gj_comp_geom* k = &g->comp[c];
        k->samp_h = param->sampling_factor[c].horizontal;
        k->samp_v = param->sampling_factor[c].vertical;
        if (k->samp_h < 1 || k->samp_h > 15 || k->samp_v < 1 || k->samp_v > 15) return -1;
        if (k->samp_h > g->max_h) g->max_h = k->samp_h;
        if (k->samp_v > g->max_v) g->max_v = k->samp_v;
        k->type = (param->color_space_internal == GPUJPEG_RGB || c == 0 || c == 3) ? GJ_LUMA : GJ_CHROMA;
        const int div_h = g->max_h / k->samp_h, div_v = g->max_v / k->samp_v;
        const int width = div_up(pi->width, div_h) * div_h, height = div_up(pi->height, div_v) * div_v;
        k->width = width * k->samp_h / g->max_h;
        k->height = height * k->samp_v / g->max_v;
        const int mcu_x = 8 * (g->interleaved ? k->samp_h : 1), mcu_y = 8 * (g->interleaved ? k->samp_v : 1);
        k->data_width = div_up(k->width, mcu_x) * mcu_x;
        k->data_height = div_up(k->height, mcu_y) * mcu_y;
        k->blocks_x = k->data_width / 8;
        k->blocks_y = k->data_height / 8;
        k->mcu_count_x = k->data_width / mcu_x;
        k->mcu_count = k->mcu_count_x * (k->data_height / mcu_y);
        const int seg_mcu = param->restart_interval > 0 ? param->restart_interval : k->mcu_count;
        k->segment_count = div_up(k->mcu_count, seg_mcu);
        k->first_segment = first_segment;
        if (!g->interleaved) first_segment += k->segment_count;
        k->data_offset = offset;
        offset += (uint64_t)k->data_width * k->data_height;
    }
    for (int c = 0; c < g->comp_count; c++) {
        g->comp[c].sub_h = g->max_h / g->comp[c].samp_h;
        g->comp[c].sub_v = g->max_v / g->comp[c].samp_v;
        if (g->max_h % g->comp[c].samp_h || g->max_v % g->comp[c].samp_v) return -1; /* preprocessor.cu:321-322 asserts */
    }
    g->data_size = offset;
    g->block_count = (int)(offset / 64);
    if (g->interleaved) {
        g->mcu_count = g->comp[0].mcu_count;
        g->mcu_count_x = g->comp[0].mcu_count_x;
        g->segment_count = g->comp[0].segment_count;
        g->scan_count = 1;
        int p = 0;
        for (int c = 0; c < g->comp_count; c++) {
            const gj_comp_geom* k = &g->comp[c];
            if (k->mcu_count != g->mcu_count) return -1; /* common.c:761 assert */
            const int n = k->samp_h * k->samp_v;
            for (int i = 0; i < n; i++, p++) {
                if (p >= GJ_MAX_MCU_BLOCKS) {
                    GJ_ERROR("Sampling factors need more than %d blocks per MCU, which is not supported.\n", GJ_MAX_MCU_BLOCKS);
                    return -1;
                }
                g->mcu_comp[p] = (uint8_t)c;
                g->mcu_bx[p] = (uint8_t)(i % k->samp_h);
                g->mcu_by[p] = (uint8_t)(i / k->samp_h);
            }
        }
        g->blocks_per_mcu = p;
        for (int q = 0; q < p; q++) { /* distance to the previous block of the same component in coding order */
            const int c = g->mcu_comp[q], n = g->comp[c].samp_h * g->comp[c].samp_v;
            const bool first_of_comp = q == 0 || g->mcu_comp[q - 1] != c;
            g->mcu_prev[q] = (uint8_t)(first_of_comp ? p - (n - 1) : 1);
        }
        g->seg_blocks = (param->restart_interval > 0 ? param->restart_interval : g->mcu_count) * g->blocks_per_mcu;
    } else {
        g->segment_count = first_segment;
        g->scan_count = g->comp_count;
        g->blocks_per_mcu = 1;
        g->mcu_prev[0] = 1;
        int longest = 0;
        for (int c = 0; c < g->comp_count; c++) {
            g->mcu_count += g->comp[c].mcu_count;
            if (g->comp[c].mcu_count > longest) longest = g->comp[c].mcu_count;
        }
        g->seg_blocks = param->restart_interval > 0 ? param->restart_interval : longest;
    }
    /* planar copy path (preprocessor.cu:294-314 for the encoder, postprocessor.cu:318-347 for the decoder) */
    g->no_transform = 0;
    if (!gj_pixfmt_is_interleaved(pi->pixel_format)) {
        bool ok = !(param->comp_count >= 3 && pi->color_space != param->color_space_internal);
        if (encoder && param->comp_count != 3) ok = true; /* the encoder checks the colour space only for 3 components */
        const struct gpujpeg_component_sampling_factor* sf = gj_pixfmt_sampling(pi->pixel_format);
        if (gpujpeg_pixel_format_get_comp_count(pi->pixel_format) != param->comp_count && !encoder) ok = false;
        for (int c = 0; c < param->comp_count && ok; c++)
            if (sf[c].horizontal != g->comp[c].samp_h || sf[c].vertical != g->comp[c].samp_v) ok = false;
        g->no_transform = ok ? 1 : 0;
    }
    return 0;
}

/* ------------------------------------------------------------------ buffers / timers / statistics */
int gj_ensure_device_buffer(void** p, size_t* cap, size_t need)
{
    if (need <= *cap) return 0;
    gj_hip_free(*p);
    *cap = 0;
    *p = gj_hip_malloc(need);
    if (*p == NULL) {
        GJ_ERROR("Device allocation of %zu bytes failed: %s\n", need, gj_hip_last_error());
        return -1;
    }
    *cap = need;
    return 0;
}

int gj_timers_create(struct gj_timers* t)
{
    memset(t, 0, sizeof *t);
    for (int i = 0; i < GJ_ENC_EVENTS; i++)
        if ((t->ev[i] = gj_hip_event_create()) == NULL) return -1;
    for (int i = 0; i < 2; i++)
        if ((t->copy_in[i] = gj_hip_event_create()) == NULL || (t->copy_out[i] = gj_hip_event_create()) == NULL) return -1;
    if ((t->lane_in = gj_hip_event_create()) == NULL || (t->lane_out = gj_hip_event_create()) == NULL) return -1;
    return 0;
}

void gj_timers_destroy(struct gj_timers* t)
{
    for (int i = 0; i < GJ_ENC_EVENTS; i++) gj_hip_event_destroy(t->ev[i]);
    for (int i = 0; i < 2; i++) {
        gj_hip_event_destroy(t->copy_in[i]);
        gj_hip_event_destroy(t->copy_out[i]);
    }
    gj_hip_event_destroy(t->lane_in);
    gj_hip_event_destroy(t->lane_out);
    memset(t, 0, sizeof *t);
}

void gj_coder_process_stats(struct gj_coder* c, bool with_stats) /* common.c:2170-2230 */
{
    if (!with_stats) return;
    c->stop_time = gpujpeg_get_time();
    const double ms = (c->stop_time - c->start_time) * 1000.0, init_ms = (c->init_end_time - c->start_time) * 1000.0;
    if (c->frames == 0) c->first_frame_duration = ms;
    c->aggregate_duration += ms;
    c->frames++;
    if (c->param.verbose < GPUJPEG_LL_STATUS) return;
    const struct gpujpeg_duration_stats* s = &c->stats;
    const char* what = c->encoder ? "Encode" : "Decode";
    if (c->param.verbose >= GPUJPEG_LL_VERBOSE) {
        fprintf(stderr, " -(Re)initialization:%10.4f ms\n", init_ms);
        if (!c->encoder) fprintf(stderr, " -Stream Reader:     %10.4f ms\n", s->duration_stream);
        fprintf(stderr, " -Copy To Device:    %10.4f ms\n", s->duration_memory_to);
        if (c->encoder) {
            fprintf(stderr, " -Preprocessing:     %10.4f ms\n", s->duration_preprocessor);
            fprintf(stderr, " -DCT & Quantization:%10.4f ms\n", s->duration_dct_quantization);
            fprintf(stderr, " -Huffman Encoder:   %10.4f ms\n", s->duration_huffman_coder);
            fprintf(stderr, " -Copy From Device:  %10.4f ms\n", s->duration_memory_from);
            fprintf(stderr, " -Stream Formatter:  %10.4f ms\n", s->duration_stream);
        } else {
            fprintf(stderr, " -Huffman Decoder:   %10.4f ms\n", s->duration_huffman_coder);
            fprintf(stderr, " -DCT & Quantization:%10.4f ms\n", s->duration_dct_quantization);
            fprintf(stderr, " -Postprocessing:    %10.4f ms\n", s->duration_preprocessor);
            fprintf(stderr, " -Copy From Device:  %10.4f ms\n", s->duration_memory_from);
        }
    }
    fprintf(stderr, "%s Image GPU:    %10.4f ms (only in-GPU processing)\n", what, s->duration_in_gpu);
    fprintf(stderr, "%s Image Bare:   %10.4f ms (without copy to/from GPU memory)\n", what, ms - s->duration_memory_to - s->duration_memory_from);
    fprintf(stderr, "%s Image:        %10.4f ms\n", what, ms);
}

double gj_now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}

void gj_coder_process_stats_overall(struct gj_coder* c) /* common.c:2238-2254 */
{
    if (c->ht_on && c->ht_calls > 0)
        fprintf(stderr, "[gj host timing] %s, %ld calls: to the first launch %.1f us, launches %.1f us, waiting %.1f us, behind the wait %.1f us\n",
                c->encoder ? "encoder" : "decoder", c->ht_calls, c->ht[0] / c->ht_calls, c->ht[1] / c->ht_calls, c->ht[2] / c->ht_calls, c->ht[3] / c->ht_calls);
    /* (frames = calls that were TIMED, perf_stats on: a coder whose statistics were off for the rest of its life has nothing to average) */
    if (c->frames <= 1 || c->param.verbose <= GPUJPEG_LL_QUIET || !(c->aggregate_duration > 0.0)) return;
    fprintf(stderr, "\nAvg %s Duration: %10.4f ms\n", c->encoder ? "Encode" : "Decode", c->aggregate_duration / (double)c->frames);
    if (c->param.verbose >= GPUJPEG_LL_VERBOSE)
        fprintf(stderr, "Avg w/o 1st Iter:    %10.4f ms\n", (c->aggregate_duration - c->first_frame_duration) / ((double)c->frames - 1));
    fprintf(stderr, "\n");
}

/* ------------------------------------------------------------------ OpenGL interop: not compiled in (gpujpeg_common.h:501-502) */
int gpujpeg_opengl_init(struct gpujpeg_opengl_context** ctx) { (void)ctx; return -2; }
void gpujpeg_opengl_destroy(struct gpujpeg_opengl_context* ctx) { (void)ctx; }
int gpujpeg_opengl_texture_create(int w, int h, uint8_t* d) { (void)w; (void)h; (void)d; GJ_ERROR("OpenGL support was not compiled in.\n"); return 0; }
int gpujpeg_opengl_texture_set_data(int id, uint8_t* d) { (void)id; (void)d; return -2; }
int gpujpeg_opengl_texture_get_data(int id, uint8_t* d, size_t* n) { (void)id; (void)d; (void)n; return -2; }
void gpujpeg_opengl_texture_destroy(int id) { (void)id; }
struct gpujpeg_opengl_texture* gpujpeg_opengl_texture_register(int id, enum gpujpeg_opengl_texture_type t) { (void)id; (void)t; GJ_ERROR("OpenGL support was not compiled in.\n"); return NULL; }
void gpujpeg_opengl_texture_unregister(struct gpujpeg_opengl_texture* t) { (void)t; }
uint8_t* gpujpeg_opengl_texture_map(struct gpujpeg_opengl_texture* t, size_t* n) { (void)t; if (n) *n = 0; return NULL; }
void gpujpeg_opengl_texture_unmap(struct gpujpeg_opengl_texture* t) { (void)t; }

/* "XYZ" / "XYZW" -> packed channel mapping (src/gpujpeg_encoder.c:661-699): nibble i = source channel of output channel i,
 * 4 = all ones ('F'), 5 = all zeros ('Z'); bits 24.. = number of channels */
int gj_parse_channel_remap(unsigned* out, const char* val, const char* optname)
{
    if (strcmp(val, "help") == 0) {
        printf("syntax for %s:\n", optname);
        printf("\t\"XYZ\" or \"XYZW\" where the letters are input channel indices\n");
        printf("\tplaceholder 'Z' or 'F' can be used to set the channel to all-zeros or all-ones\n");
        printf("\n");
        printf("examples:\n");
        printf("\t\"1230\" or \"123F\" to map ARGB to RGBA\n");
        return GPUJPEG_ERROR;
    }
    const int mapped_count = (int)strlen(val);
    if (mapped_count > GPUJPEG_MAX_COMPONENT_COUNT) {
        GJ_ERROR("Mapping for more than %d channels specified!\n", GPUJPEG_MAX_COMPONENT_COUNT);
        return GPUJPEG_ERROR;
    }
    unsigned map = 0;
    for (const char* ptr = val + mapped_count - 1; ptr >= val; ptr--) {
        int src_chan = *ptr - '0';
        if (*ptr == 'F') src_chan = 4;
        else if (*ptr == 'Z') src_chan = 5;
        else if (src_chan < 0 || src_chan >= mapped_count) {
            GJ_ERROR("Invalid channel index %c for %s (mapping %d channels)!\n", *ptr, optname, mapped_count);
            return GPUJPEG_ERROR;
        }
        map = (map << 4) | (unsigned)src_chan;
    }
    *out = map | ((unsigned)mapped_count << 24);
    return GPUJPEG_NOERR;
}

/* ---- developer settings (include/gpujpeg_amd_ext.h): the library itself never reads the environment */
int gpujpeg_amd_tuning(const char* setting) { return gj_hip_tuning_setting(setting); }
const char* const* gpujpeg_amd_tuning_names(void)
{
    int n;
    return gj_hip_tuning_names(&n);
}
