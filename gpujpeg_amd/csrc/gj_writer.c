/*
 * gj_writer.c -- JPEG stream headers. The entropy-coded data are placed by the device (k_assemble);
 * the host only produces the marker segments, byte for byte as the reference writer does
 * (src/gpujpeg_writer.c:120-160 APP0, :172-250 SPIFF, :255-270 APP14, :283-300 DQT, :318-352 SOF0,
 *  :363-405 DHT, :414-449 DRI/COM, :452-520 order of markers, :550-657 scan header + APP13 placeholders).
 */
#define _POSIX_C_SOURCE 200809L /* localtime_r */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h> /* strncasecmp */
#include <time.h>

#include "gj_internal.h"

struct bw {
    uint8_t* p;
    size_t n;   /* bytes the content needs so far */
    size_t cap; /* bytes of p: nothing is ever written behind them, n keeps counting (the caller compares) */
};
static void b1(struct bw* w, unsigned v) { if (w->n < w->cap) w->p[w->n] = (uint8_t)v; w->n++; }
static void b2(struct bw* w, unsigned v) { b1(w, v >> 8); b1(w, v); }
static void b4(struct bw* w, unsigned v) { b2(w, v >> 16); b2(w, v); }
static void marker(struct bw* w, unsigned m) { b1(w, 0xFF); b1(w, m); }
static void text(struct bw* w, const char* s, size_t n) { for (size_t i = 0; i < n; i++) b1(w, (unsigned char)s[i]); }

static unsigned component_id(const gj_geom* g, int i) /* writer.c:303-311 */
{
    return g->color_space_internal == GPUJPEG_RGB ? (unsigned)"RGBA"[i] : (unsigned)i + 1;
}

static void jfif_app0(struct bw* w)
{
    marker(w, 0xE0);
    b2(w, 16);
    text(w, "JFIF", 5);
    b1(w, 1); b1(w, 1); /* version 1.01 */
    b1(w, 1);           /* dots per inch */
    b2(w, 300); b2(w, 300);
    b1(w, 0); b1(w, 0); /* no thumbnail */
}

static void adobe_app14(struct bw* w)
{
    marker(w, 0xEE);
    b2(w, 14);
    text(w, "Adobe", 5);
    b2(w, 100);
    b2(w, 0); b2(w, 0);
    b1(w, 0); /* transform 0: components are stored as they are (RGB) */
}

static void spiff_app8(struct bw* w, const gj_geom* g, const struct gpujpeg_image_metadata* md)
{
    int cs;
    if (g->comp_count == 1) cs = 8;
    else switch (g->color_space_internal) {
        case GPUJPEG_YCBCR_BT709: cs = 1; break;
        case GPUJPEG_YCBCR_BT601_256LVLS: cs = 3; break;
        case GPUJPEG_YCBCR_BT601: cs = 4; break;
        case GPUJPEG_RGB: cs = 10; break;
        default: cs = 2; break;
    }
    marker(w, 0xE8);
    b2(w, 32);
    text(w, "SPIFF", 6);
    b2(w, 0x100);
    b1(w, (cs == 3 || cs == 8) ? 1 : 0); /* profile */
    b1(w, g->comp_count);
    b4(w, (unsigned)g->height);
    b4(w, (unsigned)g->width);
    b1(w, cs);
    b1(w, 8); /* bits per sample */
    b1(w, 5); /* compression: JPEG */
    b1(w, 0); /* resolution units */
    b4(w, 1); b4(w, 1);
    if (md && md->vals[GPUJPEG_METADATA_ORIENTATION].set) { /* directory entry: orientation */
        marker(w, 0xE8);
        b2(w, 10);
        b4(w, 4);
        b1(w, md->vals[GPUJPEG_METADATA_ORIENTATION].orient.rotation);
        b1(w, md->vals[GPUJPEG_METADATA_ORIENTATION].orient.flip);
        b2(w, 0);
    }
    marker(w, 0xE8); /* end of directory; its length covers the SOI that follows */
    b2(w, 8);
    b4(w, 1);
    marker(w, 0xD8);
}

/* ---------------------------------------------------------------------------------------------------------------
 * Exif APP1 (src/gpujpeg_exif.c:172-450), big endian ("MM"): 0th IFD with Orientation, X/YResolution 72/1, ResolutionUnit
 * inches, DateTime (now), YCbCrPositioning centred and the pointer to the Exif IFD with ExifVersion 0230,
 * ComponentsConfiguration YCbCr, FlashpixVersion 0100, ColorSpace sRGB and the pixel dimensions; user tags (enc_exif_tag,
 * src/gpujpeg_exif.c:455-600) are appended to the IFD their id belongs to (ids below 0x827A: 0th IFD), replace the
 * built-in tag of the same id, and the 12-byte records are sorted by id afterwards. Values longer than 4 bytes follow
 * their IFD in the order the records were written.
 * --------------------------------------------------------------------------------------------------------------- */
enum { XT_BYTE = 1, XT_ASCII = 2, XT_SHORT = 3, XT_LONG = 4, XT_RATIONAL = 5, XT_UNDEFINED = 7, XT_SLONG = 9, XT_SRATIONAL = 10, XT_END = 11 };
enum { XF_NUMERIC = 1, XF_BYTES = 4, XF_RATIONAL = 8 };
static const struct { unsigned size; const char* name; unsigned flags; } exif_types[XT_END] = {
    [XT_BYTE] = {1, "BYTE", XF_NUMERIC},      [XT_ASCII] = {1, "ASCII", XF_BYTES},        [XT_SHORT] = {2, "SHORT", XF_NUMERIC},
    [XT_LONG] = {4, "LONG", XF_NUMERIC},      [XT_RATIONAL] = {8, "RATIONAL", XF_RATIONAL}, [XT_UNDEFINED] = {1, "UNDEFINED", XF_BYTES},
    [XT_SLONG] = {4, "SLONG", XF_NUMERIC},    [XT_SRATIONAL] = {8, "SRATIONAL", XF_RATIONAL},
};
/* the names enc_exif_tag=<name>=<value> understands (src/gpujpeg_exif.c:131-152, spelling included) */
static const struct { uint16_t id; uint8_t type; uint8_t count; const char* name; } exif_names[] = {
    {0x112, XT_SHORT, 1, "Orientation"},       {0x11A, XT_RATIONAL, 1, "XResolution"},   {0x11B, XT_RATIONAL, 1, "YResolution"},
    {0x128, XT_SHORT, 1, "ResolutionUnit"},    {0x131, XT_ASCII, 0, "Sofware"},          {0x132, XT_ASCII, 20, "DateTime"},
    {0x13E, XT_RATIONAL, 2, "WhitePoint"},     {0x213, XT_SHORT, 1, "YCbCrPositioning"}, {0x8769, XT_LONG, 1, "Exif IFD Pointer"},
    {0x9000, XT_UNDEFINED, 4, "ExifVersion"},  {0x9101, XT_UNDEFINED, 4, "ComponentConfiguration"},
    {0xA000, XT_UNDEFINED, 4, "FlashPixVersion"}, {0xA001, XT_SHORT, 1, "ColorSpace"}, {0xA002, XT_SHORT, 1, "PixelXDimension"},
    {0xA003, XT_SHORT, 1, "PixelYDimension"},
};
#define GJ_EXIF_FIRST_PRIVATE 0x827A /* first id of the Exif private tags (src/gpujpeg_exif.c:164) */
#define GJ_EXIF_CUSTOM_MAX 2048      /* bytes of user values: the main header is staged in a 4 KiB buffer */

struct gj_exif_tag {
    uint16_t id;
    uint16_t type;
    uint32_t count;   /* as written into the record */
    uint32_t* u;      /* numeric / rational values (2 per rational) */
    char* s;          /* ASCII / UNDEFINED bytes */
};
struct gj_exif_tags {
    struct gj_exif_tag* v[2]; /* [0] 0th IFD (TIFF tags), [1] Exif IFD */
    size_t n[2];
    size_t bytes;
};

static void exif_usage(void)
{
    printf("Exif value syntax:\n"
           "\t" GPUJPEG_ENC_OPT_EXIF_TAG "=<ID>:<type>=<value>\n"
           "\t" GPUJPEG_ENC_OPT_EXIF_TAG "=<name>=<value>\n"
           "\t\tname must be a tag name known to GPUJPEG\n\n"
           "If mulitple numeric values required, separate with a comma; rationals are in format num/den.\n"
           "UNDEFINED and ASCII should be raw strings.\n\n"
           "recognized tag name (type, count):\n");
    for (size_t i = 0; i < sizeof exif_names / sizeof exif_names[0]; i++)
        printf("\t- %s (%s, %u)\n", exif_names[i].name, exif_types[exif_names[i].type].name, (unsigned)exif_names[i].count);
}

int gj_exif_add_tag(struct gj_exif_tags** tags, const char* cfg)
{
    if (strcmp(cfg, "help") == 0) {
        exif_usage();
        return -1;
    }
    char* q = (char*)cfg;
    long id = 0;
    int type = 0;
    if (isdigit((unsigned char)*q)) { /* <ID>:<type>=<value> */
        id = strtol(q, &q, 0);
        if (*q != ':') {
            GJ_ERROR("Error parsing Exif tag ID or missing type!\n");
            return -1;
        }
        q++;
        for (int t = 1; t < XT_END; t++) {
            if (exif_types[t].name == NULL) continue;
            const size_t len = strlen(exif_types[t].name);
            if (strncasecmp(q, exif_types[t].name, len) == 0) {
                type = t;
                q += len;
                break;
            }
        }
        if (type == 0) {
            GJ_ERROR("Error parsing Exif tag type!\n");
            return -1;
        }
        if (*q != '=') {
            GJ_ERROR("Error parsing Exif - missing value!\n");
            return -1;
        }
    } else { /* <name>=<value> */
        for (size_t i = 0; i < sizeof exif_names / sizeof exif_names[0]; i++) {
            const size_t len = strlen(exif_names[i].name);
            if (strncasecmp(q, exif_names[i].name, len) == 0) {
                id = exif_names[i].id;
                type = exif_names[i].type;
                q += len;
            }
        }
        if (*q != '=' || type == 0) {
            GJ_ERROR("[Exif] Wrong tag name or missing value!\n");
            return -1;
        }
    }
    q++;
    struct gj_exif_tag t;
    memset(&t, 0, sizeof t);
    t.id = (uint16_t)id;
    t.type = (uint16_t)type;
    size_t value_bytes = 0;
    if (exif_types[type].flags & XF_BYTES) {
        t.s = strdup(q);
        if (!t.s) return -1;
        t.count = (uint32_t)strlen(t.s) + (type == XT_ASCII ? 1u : 0u); /* ASCII counts its terminator */
        value_bytes = t.count;
        q += strlen(q);
    } else {
        size_t n = 0;
        do {
            if (*q == ',') q++;
            const int per = (exif_types[type].flags & XF_RATIONAL) ? 2 : 1;
            uint32_t* nu = realloc(t.u, (n + 1) * (size_t)per * sizeof *nu);
            if (!nu) { free(t.u); return -1; }
            t.u = nu;
            t.u[n * (size_t)per] = (uint32_t)strtoull(q, &q, 0);
            if (per == 2) {
                if (*q != '/') GJ_ERROR("[Exif] Malformed rational, expected '/', got '%c'!\n", *q);
                if (*q) q++;
                t.u[n * 2 + 1] = (uint32_t)strtoull(q, &q, 0);
            }
            n++;
        } while (*q == ',');
        t.count = (uint32_t)n;
        value_bytes = n * exif_types[type].size;
    }
    if (*q != '\0') {
        GJ_ERROR("Trainling data in Exif value: %s\n", q);
        free(t.u); free(t.s);
        return -1;
    }
    if (*tags == NULL) *tags = calloc(1, sizeof **tags);
    if (*tags == NULL || (*tags)->bytes + value_bytes + 12 > GJ_EXIF_CUSTOM_MAX) {
        GJ_ERROR("[Exif] Custom tags are limited to %d bytes in total in the MI355X build.\n", GJ_EXIF_CUSTOM_MAX);
        free(t.u); free(t.s);
        return -1;
    }
    const int table = id < GJ_EXIF_FIRST_PRIVATE ? 0 : 1;
    struct gj_exif_tag* nv = realloc((*tags)->v[table], ((*tags)->n[table] + 1) * sizeof *nv);
    if (!nv) { free(t.u); free(t.s); return -1; }
    (*tags)->v[table] = nv;
    nv[(*tags)->n[table]++] = t;
    (*tags)->bytes += value_bytes + 12;
    return 0;
}

void gj_exif_tags_destroy(struct gj_exif_tags* tags)
{
    if (!tags) return;
    for (int k = 0; k < 2; k++) {
        for (size_t i = 0; i < tags->n[k]; i++) { free(tags->v[k][i].u); free(tags->v[k][i].s); }
        free(tags->v[k]);
    }
    free(tags);
}

/* one 12-byte IFD record; a value longer than 4 bytes goes to *end (offsets count from `start`) -- src/gpujpeg_exif.c:180-244 */
static void exif_record(struct bw* w, size_t start, size_t* end, unsigned id, unsigned type, uint32_t count, const uint32_t* u, const char* s)
{
    unsigned size = exif_types[type].size;
    b2(w, id);
    b2(w, type);
    b4(w, count);
    if (exif_types[type].flags & XF_RATIONAL) { /* stored as pairs of LONGs */
        count *= 2;
        size /= 2;
    }
    const size_t total = (size_t)size * count;
    struct bw v = {w->p, w->n, w->cap};
    if (total > 4) {
        b4(w, (unsigned)(*end - start));
        v.n = *end;
    }
    if (exif_types[type].flags & XF_BYTES) {
        for (uint32_t i = 0; i < count; i++) b1(&v, (unsigned char)s[i]);
    } else {
        for (uint32_t c = 0; c < count; c++)
            for (unsigned i = 0; i < size; i++) b1(&v, (u[c] >> (8 * (size - i - 1))) & 0xFFu);
    }
    if (total > 4) {
        *end = v.n;
    } else {
        for (size_t i = total; i < 4; i++) b1(&v, 0); /* left aligned, zero padded */
        w->n = v.n;
    }
}

static int exif_record_cmp(const void* a, const void* b)
{
    const uint8_t* x = a;
    const uint8_t* y = b;
    return ((int)x[0] << 8 | x[1]) - ((int)y[0] << 8 | y[1]);
}

struct exif_builtin { unsigned id, type; uint32_t count; const uint32_t* u; const char* s; };

/* `written`: how many of the (possibly shortened) built-in list are emitted -- the reference passes the original array length for
 * the Exif IFD (src/gpujpeg_exif.c:385-387), so a replaced built-in tag leaves a second copy of the list's last entry behind */
static void exif_ifd(struct bw* w, size_t start, struct exif_builtin* builtin, size_t n_builtin, size_t written, const struct gj_exif_tag* custom, size_t n_custom)
{
    for (size_t i = 0; i < n_custom; i++) /* user tags replace built-in ones (remove_overriden, :319-333) */
        for (size_t j = 0; j < n_builtin; j++)
            if (custom[i].id == builtin[j].id) {
                memmove(builtin + j, builtin + j + 1, (n_builtin - j - 1) * sizeof builtin[0]);
                n_builtin--;
                break;
            }
    if (written == 0) written = n_builtin;
    const size_t all = written + n_custom;
    size_t end = w->n + 2 + all * 12 + 4;
    b2(w, (unsigned)all);
    const size_t first = w->n;
    for (size_t i = 0; i < written; i++) exif_record(w, start, &end, builtin[i].id, builtin[i].type, builtin[i].count, builtin[i].u, builtin[i].s);
    for (size_t i = 0; i < n_custom; i++) exif_record(w, start, &end, custom[i].id, custom[i].type, custom[i].count, custom[i].u, custom[i].s);
    if (n_custom && w->n <= w->cap) qsort(w->p + first, all, 12, exif_record_cmp);
    b4(w, 0); /* no next IFD */
    w->n = end;
}

static void exif_app1(struct bw* w, const gj_geom* g, const struct gpujpeg_image_metadata* md, const struct gj_exif_tags* custom)
{
    static const uint8_t orient_map[8][2] = {{0, 0}, {0, 1}, {2, 0}, {2, 1}, {1, 1}, {1, 0}, {3, 1}, {3, 0}}; /* {rotation, flip} of Exif values 1..8 */
    marker(w, 0xE1);
    const size_t len_at = w->n;
    b2(w, 0);
    text(w, "Exif", 5);
    b1(w, 0);
    const size_t start = w->n; /* offsets count from here */
    text(w, "MM", 2);
    b2(w, 0x002A);
    b4(w, 8);
    uint32_t orientation = 1;
    if (md && md->vals[GPUJPEG_METADATA_ORIENTATION].set)
        for (unsigned i = 0; i < 8; i++)
            if (orient_map[i][0] == md->vals[GPUJPEG_METADATA_ORIENTATION].orient.rotation && orient_map[i][1] == md->vals[GPUJPEG_METADATA_ORIENTATION].orient.flip)
                orientation = i + 1;
    char date_time[20] = "    :  :     :  :  ";
    {
        const time_t now = time(NULL);
        struct tm tmv;
        if (localtime_r(&now, &tmv)) (void)strftime(date_time, sizeof date_time, "%Y:%m:%d %H:%M:%S", &tmv);
    }
    static const uint32_t dpi[2] = {72, 1}, inches = 2, centred = 1, srgb = 1, zero = 0;
    struct exif_builtin tiff[] = {
        {0x0112, XT_SHORT, 1, &orientation, NULL}, {0x011A, XT_RATIONAL, 1, dpi, NULL}, {0x011B, XT_RATIONAL, 1, dpi, NULL},
        {0x0128, XT_SHORT, 1, &inches, NULL},      {0x0132, XT_ASCII, 20, NULL, date_time}, {0x0213, XT_SHORT, 1, &centred, NULL},
        {0x8769, XT_LONG, 1, &zero, NULL}, /* patched below */
    };
    const size_t ifd0 = w->n;
    exif_ifd(w, start, tiff, sizeof tiff / sizeof tiff[0], 0, custom ? custom->v[0] : NULL, custom ? custom->n[0] : 0);
    if (w->n <= w->cap) { /* the Exif IFD starts here: store its offset in the pointer record (the reference reads a compound literal that is out of
         * scope at that point, src/gpujpeg_exif.c:297-300, i.e. writes an unspecified value) */
        const unsigned n = ((unsigned)w->p[ifd0] << 8) | w->p[ifd0 + 1];
        for (unsigned i = 0; i < n; i++) {
            uint8_t* r = w->p + ifd0 + 2 + (size_t)i * 12;
            if (r[0] == 0x87 && r[1] == 0x69 && r[2] == 0 && r[3] == XT_LONG) {
                const uint32_t off = (uint32_t)(w->n - start);
                r[8] = (uint8_t)(off >> 24); r[9] = (uint8_t)(off >> 16); r[10] = (uint8_t)(off >> 8); r[11] = (uint8_t)off;
            }
        }
    }
    const uint32_t width = (uint32_t)g->width, height = (uint32_t)g->height;
    struct exif_builtin priv[] = {
        {0x9000, XT_UNDEFINED, 4, NULL, "0230"}, {0x9101, XT_UNDEFINED, 4, NULL, "\1\2\3\0"}, {0xA000, XT_UNDEFINED, 4, NULL, "0100"},
        {0xA001, XT_SHORT, 1, &srgb, NULL},      {0xA002, XT_SHORT, 1, &width, NULL},          {0xA003, XT_SHORT, 1, &height, NULL},
    };
    /* `written` is deliberately the size of priv[] BEFORE user tags removed entries from it: the reference passes ARR_SIZE(tags)
     * instead of the reduced count for this IFD (src/gpujpeg_exif.c:385-387), so a replaced private tag leaves a duplicate of the
     * last record in its files; we reproduce its bytes (tests/test_oracle_vs_ref.py compares the APP1 segments of both libraries) */
    exif_ifd(w, start, priv, sizeof priv / sizeof priv[0], sizeof priv / sizeof priv[0], custom ? custom->v[1] : NULL, custom ? custom->n[1] : 0);
    const size_t length = w->n - len_at;
    if (len_at + 1 < w->cap) {
        w->p[len_at] = (uint8_t)(length >> 8);
        w->p[len_at + 1] = (uint8_t)length;
    }
}

size_t gj_write_main_header(uint8_t* out, size_t out_cap, const gj_geom* g, const struct gpujpeg_parameters* param, enum gpujpeg_header_type header_type,
                            const uint8_t qraw[2][64], const struct gpujpeg_image_metadata* md, const struct gj_exif_tags* exif_tags)
{
    struct bw w = {out, 0, out_cap}; /* (returns the size the header needs: larger than out_cap = it does not fit, and was cut) */
    marker(&w, 0xD8);
    enum gpujpeg_header_type h = header_type;
    if (h == GPUJPEG_HEADER_DEFAULT) { /* writer.c:456-474 */
        if (g->comp_count == 4 || (md && md->vals[GPUJPEG_METADATA_ORIENTATION].set)) h = GPUJPEG_HEADER_SPIFF;
        else if (g->color_space_internal == GPUJPEG_YCBCR_BT601 || g->color_space_internal == GPUJPEG_YCBCR_BT709) h = GPUJPEG_HEADER_SPIFF;
        else if (g->color_space_internal == GPUJPEG_RGB) h = GPUJPEG_HEADER_ADOBE;
        else h = GPUJPEG_HEADER_JFIF;
    }
    switch (h) {
    case GPUJPEG_HEADER_SPIFF: spiff_app8(&w, g, md); break;
    case GPUJPEG_HEADER_ADOBE: adobe_app14(&w); break;
    case GPUJPEG_HEADER_EXIF:
        if (g->color_space_internal != GPUJPEG_YCBCR_BT601_256LVLS)
            GJ_WARN("[Exif] Color space %s currently not recorded, assumed %s (report)\n", gpujpeg_color_space_get_name((enum gpujpeg_color_space)g->color_space_internal),
                    gpujpeg_color_space_get_name(GPUJPEG_YCBCR_BT601_256LVLS));
        exif_app1(&w, g, md, exif_tags);
        break;
    default: jfif_app0(&w); break;
    }
    unsigned seen = 0;
    for (int c = 0; c < g->comp_count; c++) { /* DQT once per table type in component order */
        const int t = g->comp[c].type;
        if (seen & (1u << t)) continue;
        seen |= 1u << t;
        marker(&w, 0xDB);
        b2(&w, 67);
        b1(&w, t);
        for (int i = 0; i < 64; i++) b1(&w, qraw[t][i]);
    }
    marker(&w, 0xC0);
    b2(&w, 8 + 3 * g->comp_count);
    b1(&w, 8);
    b2(&w, g->height);
    b2(&w, g->width);
    b1(&w, g->comp_count);
    for (int c = 0; c < g->comp_count; c++) {
        b1(&w, component_id(g, c));
        b1(&w, (g->comp[c].samp_h << 4) + g->comp[c].samp_v);
        b1(&w, g->comp[c].type == GJ_LUMA ? 0 : 1);
    }
    seen = 0;
    for (int c = 0; c < g->comp_count; c++) { /* DHT: DC then AC table of each type that occurs */
        const int t = g->comp[c].type;
        if (seen & (1u << t)) continue;
        seen |= 1u << t;
        for (int ac = 0; ac < 2; ac++) {
            const uint8_t *bits, *vals;
            int count;
            gj_huffman_std_spec(t, ac, &bits, &vals, &count);
            marker(&w, 0xC4);
            b2(&w, count + 2 + 1 + 16);
            b1(&w, (ac ? 16 : 0) + t);
            for (int i = 1; i <= 16; i++) b1(&w, bits[i]);
            for (int i = 0; i < count; i++) b1(&w, vals[i]);
        }
    }
    marker(&w, 0xDD);
    b2(&w, 4);
    b2(&w, param->restart_interval);
    char com[64];
    const int q = param->quality < 1 ? 1 : (param->quality > 100 ? 100 : param->quality);
    const int len = snprintf(com, sizeof com, "CREATOR: GPUJPEG, quality = %d", q);
    marker(&w, 0xFE);
    b2(&w, 2 + len + 1);
    text(&w, com, (size_t)len + 1);
    if (g->color_space_internal == GPUJPEG_YCBCR_BT601) {
        marker(&w, 0xFE);
        b2(&w, 2 + 10);
        text(&w, "CS=ITU601", 10);
    }
    return w.n;
}

#define GJ_MAX_HEADER_SIZE (65536 - 100) /* src/gpujpeg_common_internal.h:91 */

int gj_write_scan_headers(struct gj_scan_headers* sh, const gj_geom* g, const struct gpujpeg_parameters* param)
{
    size_t need = 0;
    for (int s = 0; s < g->scan_count; s++) {
        const int segs = g->interleaved ? g->segment_count : g->comp[s].segment_count;
        need += 16 + 2 * GJ_MAX_COMP + (size_t)(segs + 1) * 4 + 5 * ((size_t)(segs + 1) * 4 / GJ_MAX_HEADER_SIZE + 1);
    }
    free(sh->bytes);
    sh->bytes = calloc(1, need);
    if (!sh->bytes) return -1;
    struct bw w = {sh->bytes, 0, need};
    for (int s = 0; s < g->scan_count; s++) {
        sh->offset[s] = (uint32_t)w.n;
        sh->info_payload[s] = 0;
        if (param->segment_info && param->restart_interval > 0) { /* APP13 placeholders, filled by k_segment_info */
            const int segs = g->interleaved ? g->segment_count : g->comp[s].segment_count;
            int data = (segs + 1) * 4, headers = 0;
            while (data > 0) {
                const int chunk = data > GJ_MAX_HEADER_SIZE ? GJ_MAX_HEADER_SIZE : data;
                data -= chunk;
                marker(&w, 0xED);
                b2(&w, 3 + chunk);
                b1(&w, s);
                if (headers++ == 0) sh->info_payload[s] = (uint32_t)(w.n - sh->offset[s]);
                w.n += (size_t)chunk; /* zeros */
                if (headers >= GPUJPEG_MAX_SEGMENT_INFO_HEADER_COUNT) return -1;
            }
        }
        marker(&w, 0xDA);
        if (g->interleaved) {
            b2(&w, 6 + 2 * g->comp_count);
            b1(&w, g->comp_count);
            for (int c = 0; c < g->comp_count; c++) {
                b1(&w, component_id(g, c));
                b1(&w, g->comp[c].type == GJ_LUMA ? 0x00 : 0x11);
            }
        } else {
            b2(&w, 8);
            b1(&w, 1);
            b1(&w, component_id(g, s));
            b1(&w, g->comp[s].type == GJ_LUMA ? 0x00 : 0x11);
        }
        b1(&w, 0);    /* Ss */
        b1(&w, 0x3F); /* Se */
        b1(&w, 0);    /* Ah/Al */
    }
    sh->offset[g->scan_count] = (uint32_t)w.n;
    for (int s = g->scan_count + 1; s <= GJ_MAX_COMP; s++) sh->offset[s] = (uint32_t)w.n;
    sh->size = w.n;
    return 0;
}
