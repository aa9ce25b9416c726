/*
 * gj_writer.c -- JPEG stream headers. The entropy-coded data are placed by the device (k_assemble);
 * the host only produces the marker segments, byte for byte as the reference writer does
 * (src/gpujpeg_writer.c:120-160 APP0, :172-250 SPIFF, :255-270 APP14, :283-300 DQT, :318-352 SOF0,
 *  :363-405 DHT, :414-449 DRI/COM, :452-520 order of markers, :550-657 scan header + APP13 placeholders).
 */
#define _POSIX_C_SOURCE 200809L /* localtime_r */
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gj_internal.h"

struct bw {
    uint8_t* p;
    size_t n;
};
static void b1(struct bw* w, unsigned v) { w->p[w->n++] = (uint8_t)v; }
static void b2(struct bw* w, unsigned v) { b1(w, v >> 8); b1(w, v); }
static void b4(struct bw* w, unsigned v) { b2(w, v >> 16); b2(w, v); }
static void marker(struct bw* w, unsigned m) { b1(w, 0xFF); b1(w, m); }
static void text(struct bw* w, const char* s, size_t n) { memcpy(w->p + w->n, s, n); w->n += n; }

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

/* Exif APP1 with the reference's fixed tag set (src/gpujpeg_exif.c:172-300,337-450), big endian ("MM"): 0th IFD with Orientation,
 * X/YResolution 72/1, ResolutionUnit inches, DateTime (now), YCbCrPositioning centred and the pointer to the Exif IFD with
 * ExifVersion 0230, ComponentsConfiguration YCbCr, FlashpixVersion 0100, ColorSpace sRGB and the pixel dimensions. Values longer
 * than 4 bytes follow their IFD. Custom tags (enc_exif_tag) are not implemented. */
static void exif_app1(struct bw* w, const gj_geom* g, const struct gpujpeg_image_metadata* md)
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
    unsigned orientation = 1;
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
    /* ---- 0th IFD: 7 entries, long values (2 rationals, date) behind it ---- */
    {
        size_t end = w->n + 2 + 7 * 12 + 4; /* where the long values go */
        b2(w, 7);
        b2(w, 0x0112); b2(w, 3); b4(w, 1); b2(w, orientation); b2(w, 0);          /* Orientation SHORT */
        b2(w, 0x011A); b2(w, 5); b4(w, 1); b4(w, (uint32_t)(end - start));         /* XResolution RATIONAL -> offset */
        { struct bw v = {w->p, end}; b4(&v, 72); b4(&v, 1); end = v.n; }
        b2(w, 0x011B); b2(w, 5); b4(w, 1); b4(w, (uint32_t)(end - start));         /* YResolution */
        { struct bw v = {w->p, end}; b4(&v, 72); b4(&v, 1); end = v.n; }
        b2(w, 0x0128); b2(w, 3); b4(w, 1); b2(w, 2); b2(w, 0);                     /* ResolutionUnit: inches */
        b2(w, 0x0132); b2(w, 2); b4(w, 20); b4(w, (uint32_t)(end - start));        /* DateTime ASCII[20] */
        { struct bw v = {w->p, end}; text(&v, date_time, 20); end = v.n; }
        b2(w, 0x0213); b2(w, 3); b4(w, 1); b2(w, 1); b2(w, 0);                     /* YCbCrPositioning: centred */
        b2(w, 0x8769); b2(w, 4); b4(w, 1); b4(w, (uint32_t)(end - start));         /* Exif IFD pointer */
        b4(w, 0);                                                                  /* no next IFD */
        w->n = end;
    }
    /* ---- Exif IFD: 6 entries, all values fit the entry ---- */
    b2(w, 6);
    b2(w, 0x9000); b2(w, 7); b4(w, 4); text(w, "0230", 4);
    b2(w, 0x9101); b2(w, 7); b4(w, 4); text(w, "\1\2\3\0", 4);
    b2(w, 0xA000); b2(w, 7); b4(w, 4); text(w, "0100", 4);
    b2(w, 0xA001); b2(w, 3); b4(w, 1); b2(w, 1); b2(w, 0);
    b2(w, 0xA002); b2(w, 3); b4(w, 1); b2(w, (unsigned)g->width & 0xFFFF); b2(w, 0);
    b2(w, 0xA003); b2(w, 3); b4(w, 1); b2(w, (unsigned)g->height & 0xFFFF); b2(w, 0);
    b4(w, 0);
    const size_t length = w->n - len_at;
    w->p[len_at] = (uint8_t)(length >> 8);
    w->p[len_at + 1] = (uint8_t)length;
}

size_t gj_write_main_header(uint8_t* out, const gj_geom* g, const struct gpujpeg_parameters* param, enum gpujpeg_header_type header_type,
                            const uint8_t qraw[2][64], const struct gpujpeg_image_metadata* md)
{
    struct bw w = {out, 0};
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
        exif_app1(&w, g, md);
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
    struct bw w = {sh->bytes, 0};
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
