/*
 * gj_reader.c -- JPEG marker parser of the decoder. Host counterpart of src/gpujpeg_reader.c:
 * same accepted streams, colour-space deduction, output-format resolution and diagnostics, but it
 * never copies entropy-coded bytes: scans are described by (offset, length) pairs into the original
 * buffer, which the kernels read in place.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <stdlib.h>
#include <string.h>

#include "gj_internal.h"

#define RD2(p) (((unsigned)(p)[0] << 8) | (p)[1])
#define RD4(p) (((unsigned)(p)[0] << 24) | ((unsigned)(p)[1] << 16) | ((unsigned)(p)[2] << 8) | (p)[3])
#define GJ_MAX_HEADER_SIZE (65536 - 100)

static const char* marker_name(int m)
{
    static _Thread_local char buf[16];
    switch (m) {
    case 0xC0: return "SOF0"; case 0xC1: return "SOF1"; case 0xC2: return "SOF2"; case 0xC4: return "DHT";
    case 0xD8: return "SOI"; case 0xD9: return "EOI"; case 0xDA: return "SOS"; case 0xDB: return "DQT";
    case 0xDD: return "DRI"; case 0xFE: return "COM";
    default: snprintf(buf, sizeof buf, "0x%02X", m); return buf;
    }
}

/* ---- output format resolution (src/gpujpeg_reader.c:1494-1618) ---- */
static int gcd(int a, int b) { while (b) { int c = a % b; a = b; b = c; } return a; }

static enum gpujpeg_pixel_format native_pixel_format(struct gpujpeg_parameters* p)
{
    if (p->comp_count == 4) return GPUJPEG_4444_U8_P0123;
    const int n = p->comp_count < 3 ? p->comp_count : 3; /* only the components the stream defines (factors 1..4, checked at SOF0) */
    int gh = p->sampling_factor[0].horizontal, gv = p->sampling_factor[0].vertical;
    for (int i = 1; i < n; i++) { gh = gcd(gh, p->sampling_factor[i].horizontal); gv = gcd(gv, p->sampling_factor[i].vertical); }
    if (gh < 1) gh = 1;
    if (gv < 1) gv = 1;
    for (int i = 0; i < n; i++) { p->sampling_factor[i].horizontal /= gh; p->sampling_factor[i].vertical /= gv; }
    if (n == 3 && p->sampling_factor[1].horizontal == 1 && p->sampling_factor[1].vertical == 1 && p->sampling_factor[2].horizontal == 1 &&
        p->sampling_factor[2].vertical == 1) {
        const int h = p->sampling_factor[0].horizontal, v = p->sampling_factor[0].vertical;
        if (h == 1 && v == 1) return p->interleaved ? GPUJPEG_444_U8_P012 : GPUJPEG_444_U8_P0P1P2;
        if (h == 2 && v == 1) return p->interleaved ? GPUJPEG_422_U8_P1020 : GPUJPEG_422_U8_P0P1P2;
        if (h == 2 && v == 2) return GPUJPEG_420_U8_P0P1P2;
    }
    return p->interleaved ? GPUJPEG_444_U8_P012 : GPUJPEG_444_U8_P0P1P2;
}

static bool sampling_is(const struct gpujpeg_parameters* p, gpujpeg_sampling_factor_t s)
{
    return gj_make_sampling_factor(3, p->sampling_factor) == s;
}

static enum gpujpeg_pixel_format resolve_pixel_format(struct gpujpeg_parameters* p, const struct gpujpeg_image_parameters* pi)
{
    if (p->comp_count == 1) return GPUJPEG_U8;
    if (pi->pixel_format == GPUJPEG_PIXFMT_NATIVE) return native_pixel_format(p);
    if (pi->pixel_format == GPUJPEG_PIXFMT_STD && pi->color_space != GPUJPEG_RGB) {
        if (sampling_is(p, GPUJPEG_SUBSAMPLING_420)) return GPUJPEG_420_U8_P0P1P2;
        if (sampling_is(p, GPUJPEG_SUBSAMPLING_422)) return GPUJPEG_422_U8_P0P1P2;
        return GPUJPEG_444_U8_P0P1P2;
    }
    if (p->comp_count == 3) return GPUJPEG_444_U8_P012;
    return pi->pixel_format == GPUJPEG_PIXFMT_NO_ALPHA ? GPUJPEG_444_U8_P012 : GPUJPEG_4444_U8_P0123;
}

static void resolve_output(struct gpujpeg_parameters* p, struct gpujpeg_image_parameters* pi, unsigned req_alignment)
{
    if (pi->color_space == GPUJPEG_NONE) pi->color_space = p->color_space_internal;
    if (pi->color_space == GPUJPEG_CS_DEFAULT) {
        const bool gray = pi->pixel_format == GPUJPEG_U8 || (pi->pixel_format <= GPUJPEG_PIXFMT_AUTODETECT && p->comp_count == 1);
        pi->color_space = gray ? GPUJPEG_YCBCR_JPEG : GPUJPEG_RGB;
    }
    if (pi->pixel_format <= GPUJPEG_PIXFMT_AUTODETECT) pi->pixel_format = resolve_pixel_format(p, pi);
    if (req_alignment != 0) {
        const unsigned linesize = (unsigned)gj_pixfmt_unit_size(pi->pixel_format) * (unsigned)pi->width;
        pi->width_padding = (int)((linesize + req_alignment - 1) / req_alignment * req_alignment - linesize);
    }
}

/* ---- colour space from component ids (src/gpujpeg_reader.c:748-785) ---- */
static enum gpujpeg_color_space color_space_from_ids(int comp_count, const uint8_t* id, enum gpujpeg_color_space header_cs)
{
    if (comp_count < 3 || header_cs != GPUJPEG_NONE) return GPUJPEG_NONE;
    if (id[0] == 1 && id[1] == 2 && id[2] == 3) return GPUJPEG_YCBCR_BT601_256LVLS;
    if ((id[0] == 'R' && id[1] == 'G' && id[2] == 'B') || (id[0] == 'r' && id[1] == 'g' && id[2] == 'b')) return GPUJPEG_RGB;
    GJ_WARN("SOF0 unexpected component id [%d,%d,%d] was presented!\n", id[0], id[1], id[2]);
    return GPUJPEG_NONE;
}

static int push_segment(struct gj_host_segments* s, uint32_t pos, uint32_t len, uint32_t index)
{
    if (s->count == s->cap) {
        const int cap = s->cap ? s->cap * 2 : 4096;
        uint32_t* a = realloc(s->pos, (size_t)cap * sizeof(uint32_t));
        uint32_t* b = realloc(s->len, (size_t)cap * sizeof(uint32_t));
        uint32_t* c = realloc(s->index, (size_t)cap * sizeof(uint32_t));
        if (a) s->pos = a;
        if (b) s->len = b;
        if (c) s->index = c;
        if (!a || !b || !c) return -1;
        s->cap = cap;
    }
    s->pos[s->count] = pos;
    s->len[s->count] = len;
    s->index[s->count] = index;
    s->count++;
    return 0;
}

/* Walk the entropy-coded data of one scan: returns the offset of the marker that ends it; optionally records the segments between
 * restart markers. Mirrors src/gpujpeg_reader.c:1039-1155 decision by decision, because what a damaged stream decodes to is
 * observable behaviour (tests/test_gpu_refhip.py compares with the reference reader):
 *  - an RSTn that is not the expected one ends the segment right there, everything up to the expected marker is skipped (:1074-1100);
 *    if that one never comes before EOI / SOS, the walk goes on and the marker's two bytes stay inside the segment (:1103-1107);
 *  - EOI, SOS and APPn end the scan; an empty last segment is dropped (FFmpeg bug #8412, :1132-1135);
 *  - any other marker is an error (:1145-1148) -- except 0xFF fill bytes, which the reference rejects and we skip (T.81 B.1.1.2). */
static long walk_scan(const uint8_t* image, size_t begin, size_t size, uint32_t first_index, int max_segments,
                      struct gj_host_segments* segs, int* segment_count)
{
    const uint8_t* p = image + begin;
    const uint8_t* end = image + size;
    const uint8_t* seg_start = p;
    size_t dropped = 0; /* bytes the reference's byte counter did not see (skipped while looking for an expected marker that never came): its
                         * copy of the segment is that much shorter than the segment's extent (:1086,:1112) */
    int idx = 0;
    int previous = 0xD0 - 1;
    for (;;) {
        const uint8_t* f = memchr(p, 0xFF, (size_t)(end - p));
        if (f == NULL || f + 1 >= end) {
            GJ_ERROR("JPEG data unexpected ended while reading SOS marker!\n");
            return -1;
        }
        int m = f[1];
        if (m == 0x00) { p = f + 2; continue; }
        if (m == 0xFF) { p = f + 1; continue; } /* fill byte */
        if (m >= 0xD0 && m <= 0xD7) {
            const int expected = previous < 0xD7 ? previous + 1 : 0xD0;
            const uint8_t* seg_end = f; /* the segment ends in front of this marker, expected or not */
            const uint8_t* next = f + 2;
            if (m != expected) {
                GJ_ERROR("Expected marker 0x%X but 0x%X was presented!\n", expected, m);
                bool found = false;
                size_t skipped = 0;
                const uint8_t* q = f + 2;
                while (q < end) {
                    const uint8_t* h = memchr(q, 0xFF, (size_t)(end - q));
                    if (h == NULL || h + 1 >= end) { q = end; break; }
                    skipped += (size_t)(h - q) + 2;
                    q = h + 2;
                    if (h[1] == expected) {
                        fprintf(stderr, "[GPUJPEG] [Recovery] Skipping %zu bytes of data until marker 0x%X was found!\n", skipped, expected);
                        found = true;
                        break;
                    }
                    if (h[1] == 0xD9 || h[1] == 0xDA) { q = h; break; } /* read again by the main loop */
                }
                if (!found) {
                    GJ_ERROR("No marker 0x%X was found until end of current scan!\n", expected);
                    dropped += (size_t)(q - (f + 2));
                    p = q; /* the segment goes on: it keeps everything up to and including the unexpected marker */
                    if (q >= end) {
                        GJ_ERROR("JPEG data unexpected ended while reading SOS marker!\n");
                        return -1;
                    }
                    continue;
                }
                next = q;
                m = expected;
            }
            previous = m;
            if (segs && idx < max_segments && push_segment(segs, (uint32_t)(seg_start - image), (uint32_t)((size_t)(seg_end - seg_start) - dropped), first_index + (uint32_t)idx) != 0) return -1;
            idx++;
            seg_start = p = next;
            dropped = 0;
            continue;
        }
        if (m == 0xD9 || m == 0xDA || (m >= 0xE0 && m <= 0xEF)) {
            /* end of the scan; an empty trailing segment is dropped */
            if (f > seg_start) {
                if (segs && idx < max_segments && push_segment(segs, (uint32_t)(seg_start - image), (uint32_t)((size_t)(f - seg_start) - dropped), first_index + (uint32_t)idx) != 0) return -1;
                idx++;
            } else if (idx == 0) {
                idx = 0; /* a scan without any data: no segment at all, like the reference (segment_count 1 - 1) */
            }
            *segment_count = idx;
            return (long)(f - image);
        }
        GJ_ERROR("JPEG scan contains unexpected marker 0x%X!\n", m);
        return -1;
    }
}

/* Orientation (tag 0x0112) from the 0th IFD of an Exif APP1 segment -> rotation / flip metadata
 * (src/gpujpeg_exif.c:159-168,646-764). d points behind the length field, dl = payload bytes. */
static void exif_orientation(const uint8_t* d, size_t dl, struct gpujpeg_image_metadata* md)
{
    static const uint8_t map[8][2] = {{0, 0}, {0, 1}, {2, 0}, {2, 1}, {1, 1}, {1, 0}, {3, 1}, {3, 0}}; /* {rotation, flip} of values 1..8 */
    if (dl + 2 < 18 || dl < 14) { GJ_WARN("Insufficient Exif header length %zu!\n", dl + 2); return; }
    const uint8_t* base = d + 6; /* "Exif\0" + padding byte */
    const size_t n = dl - 6;
    bool le;
    if (base[0] == 'I' && base[1] == 'I') le = true;
    else if (base[0] == 'M' && base[1] == 'M') le = false;
    else { GJ_WARN("Unexpected endianity!\n"); return; }
#define XRD2(p) (le ? (unsigned)((p)[0] | (p)[1] << 8) : (unsigned)((p)[0] << 8 | (p)[1]))
#define XRD4(p) (le ? ((uint32_t)(p)[0] | (uint32_t)(p)[1] << 8 | (uint32_t)(p)[2] << 16 | (uint32_t)(p)[3] << 24) \
                   : ((uint32_t)(p)[0] << 24 | (uint32_t)(p)[1] << 16 | (uint32_t)(p)[2] << 8 | (uint32_t)(p)[3]))
    if (XRD2(base + 2) != 0x002A) { GJ_WARN("Wrong TIFF tag, expected 0x%04x!\n", 0x002A); return; }
    const uint32_t off = XRD4(base + 4);
    if ((size_t)off + 2 > n) { GJ_WARN("Unexpected end of file!\n"); return; }
    const unsigned items = XRD2(base + off);
    if ((size_t)off + 2 + (size_t)items * 12 > n) { GJ_WARN("Insufficient space to hold %u IFD0 items!\n", items); return; }
    for (unsigned i = 0; i < items; i++) {
        const uint8_t* e = base + off + 2 + (size_t)i * 12;
        const unsigned tag = XRD2(e), type = XRD2(e + 2);
        uint32_t val = XRD4(e + 8);
        if (!le && (type == 1 || type == 3)) val >>= type == 1 ? 8 : 16; /* :682-686 (short values sit in the upper half) */
        if (tag == 0x0112) {
            if (val == 0 || val > 8) { GJ_WARN("Flawed orientation value %u! Should be 1-8...\n", val); continue; }
            md->vals[GPUJPEG_METADATA_ORIENTATION].orient.rotation = map[val - 1][0];
            md->vals[GPUJPEG_METADATA_ORIENTATION].orient.flip = map[val - 1][1];
            md->vals[GPUJPEG_METADATA_ORIENTATION].set = 1;
        }
    }
#undef XRD2
#undef XRD4
}

/* entry i (big-endian u32 offset relative to the scan's first byte) of the APP13 segment index of a scan; the index is the
 * concatenation of the payloads of the scan's APP13 segments, every one but the last GJ_MAX_HEADER_SIZE bytes long */
static uint32_t seg_info_entry(const struct gj_reader_result* r, int scan, int i)
{
    const uint8_t* q = r->seg_info[scan][(i * 4) / GJ_MAX_HEADER_SIZE] + (i * 4) % GJ_MAX_HEADER_SIZE;
    return RD4(q);
}

/* An APP13 index comes from the file and is not trusted: chunk sizes must make the addressing above valid, there must be at
 * least two entries, offsets must not decrease, every segment but the last must have room for its RSTn, and the scan must end
 * inside the buffer. Anything else: the index is ignored and the scan is walked like a stream without one. */
static bool seg_info_valid(const struct gj_reader_result* r, int scan, size_t begin, size_t size)
{
    const int chunks = r->seg_info_count[scan];
    long total = 0;
    for (int k = 0; k < chunks; k++) {
        if (k + 1 < chunks && r->seg_info_size[scan][k] != GJ_MAX_HEADER_SIZE) return false;
        if (r->seg_info_size[scan][k] <= 0 || r->seg_info_size[scan][k] > GJ_MAX_HEADER_SIZE) return false;
        total += r->seg_info_size[scan][k];
    }
    if (total < 8 || total % 4 != 0) return false;
    const int entries = (int)(total / 4);
    uint32_t prev = seg_info_entry(r, scan, 0);
    for (int i = 1; i < entries; i++) {
        const uint32_t pos = seg_info_entry(r, scan, i);
        if (pos < prev || (i + 1 < entries && pos - prev < 2)) return false;
        prev = pos;
    }
    return (size_t)prev <= size - begin;
}

int gj_reader_parse(const uint8_t* image, size_t size, int verbose, bool ff_cs_itu601_is_709, enum gpujpeg_pixel_format req_pixfmt,
                    enum gpujpeg_color_space req_cs, unsigned req_alignment, struct gj_reader_result* r, bool headers_only)
{
    memset(r, 0, sizeof *r);
    gpujpeg_set_default_parameters(&r->param);
    r->param.verbose = verbose;
    r->param.restart_interval = 0;
    r->param.comp_count = 0;
    gpujpeg_image_set_default_parameters(&r->param_image);
    r->param_image.pixel_format = req_pixfmt;
    r->param_image.color_space = req_cs;
    r->header_color_space = GPUJPEG_NONE;
    const uint8_t* end = image + size;
    const uint8_t* p = image;
    if (size > 0xFFFFFF00u) { /* segment offsets and lengths are 32-bit on the device */
        GJ_ERROR("JPEG data of %zu bytes are not supported (4 GiB limit)!\n", size);
        return -1;
    }
    if (size < 4 || p[0] != 0xFF || p[1] != 0xD8) {
        GJ_ERROR("JPEG data should begin with SOI marker, but marker %s was found!\n", size >= 2 && p[0] == 0xFF ? marker_name(p[1]) : "(none)");
        return -1;
    }
    p += 2;
    bool have_sof = false, in_spiff = false;
    for (;;) {
        if (end - p < 2) {
            GJ_ERROR("JPEG data should end with EOI marker!\n");
            return -1;
        }
        if (p[0] != 0xFF) {
            GJ_ERROR("Failed to read marker from JPEG data (0xFF was expected but 0x%X was presented)\n", p[0]);
            return -1;
        }
        while (p + 1 < end && p[1] == 0xFF) p++; /* fill bytes */
        const int m = p[1];
        p += 2;
        if (m == 0xD9) { r->eoi_seen = true; break; }
        if (m == 0xD8) continue; /* SPIFF repeats SOI after its directory */
        if (end - p < 2) { GJ_ERROR("Marker %s goes beyond end of data\n", marker_name(m)); return -1; }
        const int len = (int)RD2(p);
        if (len < 2 || len > end - p) { GJ_ERROR("Marker %s goes beyond end of data\n", marker_name(m)); return -1; }
        const uint8_t* d = p + 2;
        const int dl = len - 2;
        switch (m) {
        case 0xE0: /* APP0 (reader.c:264-310) */
            if (dl >= 5 && memcmp(d, "JFIF", 5) == 0) { r->header_type = GPUJPEG_HEADER_JFIF; r->header_color_space = GPUJPEG_YCBCR_BT601_256LVLS; }
            break;
        case 0xE1: /* APP1 (reader.c:312-335): Exif implies YCbCr-JPEG; tags themselves are metadata (N4) */
            if (dl >= 5 && memcmp(d, "Exif", 5) == 0) {
                r->header_type = GPUJPEG_HEADER_EXIF;
                r->header_color_space = GPUJPEG_YCBCR_BT601_256LVLS;
                exif_orientation(d, dl, &r->metadata);
            }
            else if (verbose >= 0) GJ_WARN("Skipping unsupported APP1 marker!\n");
            break;
        case 0xE8: /* APP8 SPIFF header / directory (reader.c:387-556) */
            if (!in_spiff && dl >= 30 && memcmp(d, "SPIFF", 6) == 0) {
                const int compression = d[20], cs = d[18];
                if (compression != 5) { GJ_ERROR("Unexpected compression index %d, expected %d (JPEG)\n", compression, 5); return -1; }
                switch (cs) {
                case 1: r->header_color_space = GPUJPEG_YCBCR_BT709; break;
                case 2: break;
                case 3: case 8: r->header_color_space = GPUJPEG_YCBCR_BT601_256LVLS; break;
                case 4: r->header_color_space = GPUJPEG_YCBCR_BT601; break;
                case 10: r->header_color_space = GPUJPEG_RGB; break;
                default: GJ_ERROR("Unsupported or unrecongnized SPIFF color space %d!\n", cs); return -1;
                }
                r->header_type = GPUJPEG_HEADER_SPIFF;
                in_spiff = true;
            } else if (in_spiff && dl >= 4) {
                const unsigned tag = RD4(d);
                if (tag == 4 && dl >= 6) { /* orientation */
                    r->metadata.vals[GPUJPEG_METADATA_ORIENTATION].orient.rotation = d[4] & 3;
                    r->metadata.vals[GPUJPEG_METADATA_ORIENTATION].orient.flip = d[5] & 1;
                    r->metadata.vals[GPUJPEG_METADATA_ORIENTATION].set = 1;
                } else if (tag == 1) {
                    in_spiff = false; /* EOD: its length field counts the SOI that follows but not as payload here */
                    p += 2 + 4; /* marker length is 8 = 2 + 4 + 2(SOI); skip payload only, SOI handled by the loop */
                    continue;
                }
            }
            break;
        case 0xED: /* APP13 segment info (reader.c:229-262,344-385) */
            if (dl >= 1 && !(dl >= 13 && (memcmp(d, "Photoshop 3.0", 13) == 0 || memcmp(d, "Adobe_CM", 8) == 0)) && !(dl >= 18 && memcmp(d, "Adobe_Photoshop2.5", 18) == 0)) {
                const int scan = d[0];
                if (scan == r->scan_count && scan < GJ_MAX_COMP && r->seg_info_count[scan] < GPUJPEG_MAX_SEGMENT_INFO_HEADER_COUNT) {
                    const int k = r->seg_info_count[scan]++;
                    r->seg_info[scan][k] = d + 1;
                    r->seg_info_size[scan][k] = dl - 1;
                } else if (verbose >= 0) {
                    GJ_WARN("APP13 marker (segment info) scan index should be %d but %d was presented! (marker not a segment info?)\n", r->scan_count, scan);
                }
            }
            break;
        case 0xEE: /* APP14 Adobe (reader.c:558-640) */
            if (len >= 14 && memcmp(d, "Adobe", 5) == 0) {
                const int transform = d[11];
                r->header_type = GPUJPEG_HEADER_ADOBE;
                if (transform == 0) r->header_color_space = GPUJPEG_RGB;
                else if (transform == 1) r->header_color_space = GPUJPEG_YCBCR_BT601_256LVLS;
                else GJ_ERROR("Unsupported color transformation value '%d' was presented in APP14 marker!\n", transform);
            } else GJ_WARN("Unknown APP14 marker %dB (%dB) long was presented\n", len, dl);
            break;
        case 0xFE: /* COM (reader.c:642-680) */
            if ((dl == 10 || dl == 9) && strncmp((const char*)d, "CS=ITU601", (size_t)dl) == 0)
                r->header_color_space = ff_cs_itu601_is_709 ? GPUJPEG_YCBCR_BT709 : GPUJPEG_YCBCR_BT601;
            if (dl > 0 && d[dl - 1] == '\0') r->comment = (const char*)d;
            break;
        case 0xDB: { /* DQT (reader.c:682-727) */
            if (dl % 65 != 0) { GJ_ERROR("DQT marker length should be 65 but %d was presented!\n", dl); return -1; }
            for (int o = 0; o < dl; o += 65) {
                const int pq = d[o] >> 4, tq = d[o] & 15;
                if (pq != 0) { GJ_ERROR("Unsupported DQT Pq %d (16-bit table) presented!\n", pq); return -1; }
                if (tq > 3) { GJ_ERROR("DQT marker index should be 0-3 but %d was presented!\n", tq); return -1; }
                memcpy(r->qraw[tq], d + o + 1, 64);
                r->q_present[tq] = true;
            }
            break; }
        case 0xC1: GJ_WARN("Reading SOF1 as it was SOF0 marker (should work but verify it)!\n"); /* fall through */
        case 0xC0: { /* SOF0 (reader.c:806-893) */
            if (len < 8) { GJ_ERROR("SOF0 marker length should be at least 8 but %d was presented!\n", len); return -1; }
            if (r->header_color_space != GPUJPEG_NONE) r->param.color_space_internal = r->header_color_space;
            const int precision = d[0];
            r->param_image.height = (int)RD2(d + 1);
            r->param_image.width = (int)RD2(d + 3);
            r->param.comp_count = d[5];
            if (r->param.comp_count == 0) { GJ_ERROR("SOF0 has 0 components!\n"); return -1; }
            if (r->param.comp_count > GJ_MAX_COMP) { GJ_ERROR("SOF0 has %d components but JPEG can contain at most %d components\n", r->param.comp_count, GJ_MAX_COMP); return -1; }
            if (precision != 8) { GJ_ERROR("SOF0 marker precision should be 8 but %d was presented!\n", precision); return -1; }
            if (dl < 6 + 3 * r->param.comp_count) { GJ_ERROR("SOF0 goes beyond end of data\n"); return -1; }
            for (int c = 0; c < r->param.comp_count; c++) {
                r->comp_id[c] = d[6 + 3 * c];
                r->param.sampling_factor[c].horizontal = d[7 + 3 * c] >> 4;
                r->param.sampling_factor[c].vertical = d[7 + 3 * c] & 15;
                if (r->param.sampling_factor[c].horizontal < 1 || r->param.sampling_factor[c].horizontal > 4 ||
                    r->param.sampling_factor[c].vertical < 1 || r->param.sampling_factor[c].vertical > 4) {
                    GJ_ERROR("SOF0 marker contains unsupported sampling factor %dx%d of component %d (1-4 allowed)!\n",
                             d[7 + 3 * c] >> 4, d[7 + 3 * c] & 15, c);
                    return -1;
                }
                r->quant_map[c] = d[8 + 3 * c];
                if (r->quant_map[c] > 3) { GJ_ERROR("SOF0 marker contains unexpected quantization table index %d!\n", r->quant_map[c]); return -1; }
            }
            const enum gpujpeg_color_space det = color_space_from_ids(r->param.comp_count, r->comp_id, r->header_color_space);
            if (r->header_color_space == GPUJPEG_NONE && det != GPUJPEG_NONE) {
                GJ_VERBOSE(verbose, "Deduced color space %s.\n", gpujpeg_color_space_get_name(det));
                r->param.color_space_internal = det;
            }
            if (r->header_type == GPUJPEG_HEADER_ADOBE && r->param.color_space_internal == GPUJPEG_RGB && r->param.comp_count == 1)
                r->param.color_space_internal = GPUJPEG_YCBCR_BT601_256LVLS;
            resolve_output(&r->param, &r->param_image, req_alignment);
            have_sof = true;
            break; }
        case 0xC4: { /* DHT (reader.c:921-988) */
            int o = 0;
            while (o < dl) {
                const int tc = d[o] >> 4, th = d[o] & 15;
                if (tc > 1) { GJ_ERROR("DHT marker Tc should be 0 or 1 but %d was presented!\n", tc); return -1; }
                if (th > 3) { GJ_ERROR("DHT marker Th should be 0-3 but %d was presented!\n", th); return -1; }
                if (o + 17 > dl) { GJ_ERROR("DHT marker unexpected end when reading bit counts!\n"); return -1; }
                int count = 0;
                r->hbits[th][tc][0] = 0;
                for (int i = 1; i <= 16; i++) { r->hbits[th][tc][i] = d[o + i]; count += d[o + i]; }
                if (count > 256 || o + 17 + count > dl) { GJ_ERROR("DHT marker unexpected end when reading huffman values!\n"); return -1; }
                memcpy(r->hvals[th][tc], d + o + 17, (size_t)count);
                r->h_present[th][tc] = true;
                o += 17 + count;
            }
            break; }
        case 0xDD: { /* DRI (reader.c:997-1026) */
            if (len != 4) { GJ_ERROR("DRI marker length should be 4 but %d was presented!\n", len); return -1; }
            const int ri = (int)RD2(d);
            if (r->param.restart_interval != 0 && r->param.restart_interval != ri) {
                GJ_ERROR("DRI marker can't redefine restart interval (%d to %d)!\n", r->param.restart_interval, ri);
                fprintf(stderr, "This may be caused when more DRI markers are presented which is not supported!\n");
                return GPUJPEG_ERR_RESTART_CHANGE;
            }
            r->param.restart_interval = ri;
            break; }
        case 0xDA: { /* SOS (reader.c:1257-1372) */
            if (!have_sof) { GJ_ERROR("SOS marker before SOF0!\n"); return -1; }
            const int n = d[0];
            if (len != n * 2 + 6) { GJ_ERROR("Wrong SOS length (expected %d, got %d)\n", n * 2 + 6, len); return -1; }
            if (r->scan_count >= GJ_MAX_COMP) { GJ_ERROR("SOS marker reached maximum number of scans (%d)!\n", GJ_MAX_COMP); return -1; }
            if (n == 1) {
                if (r->scan_count == 0) r->param.interleaved = 0;
            } else {
                if (n != r->param.comp_count) { GJ_ERROR("SOS marker component count %d is not supported (should be 1 or equals to total component count)!\n", n); return -1; }
                if (r->scan_count != 0) { GJ_ERROR("SOS marker component count %d is not supported for multiple scans!\n", n); return -1; }
                r->param.interleaved = 1;
            }
            for (int i = 0; i < n; i++) {
                const int id = d[1 + 2 * i], tab = d[2 + 2 * i];
                int ci = -1;
                for (int c = 0; c < r->param.comp_count; c++) if (r->comp_id[c] == id) { ci = c; break; }
                if (ci < 0) { GJ_ERROR("Unexpected component ID '%d' present SOS marker (not defined by SOF marker)!\n", id); return -1; }
                r->huff_map[ci][0] = (tab >> 4) & 15;
                r->huff_map[ci][1] = tab & 15;
            }
            if (d[1 + 2 * n] != 0 || d[2 + 2 * n] != 63 || d[3 + 2 * n] != 0) GJ_WARN("Some of SOS parameters not valid for sequential DCT.\n");
            if (r->scan_count == 0) r->header_end = (size_t)(p - 2 - image);
            const size_t begin = (size_t)(p + len - image);
            const int scan = r->scan_count;
            r->scan_begin[scan] = begin;
            if (headers_only) { r->scan_count++; return 0; }
            long stop;
            if (begin > size) { GJ_ERROR("SOS marker goes beyond end of data\n"); return -1; }
            if (r->seg_info_count[scan] > 0 && !seg_info_valid(r, scan, begin, size)) {
                if (verbose >= 0) GJ_WARN("APP13 segment info of scan %d is inconsistent with the data, ignoring it.\n", scan);
                r->seg_info_count[scan] = 0;
            }
            if (r->seg_info_count[scan] > 0) { /* the index gives the length of the scan directly */
                int total = 0;
                for (int k = 0; k < r->seg_info_count[scan]; k++) total += r->seg_info_size[scan][k];
                stop = (long)(begin + seg_info_entry(r, scan, total / 4 - 1));
            } else {
                int count = 0;
                stop = walk_scan(image, begin, size, 0, 0, NULL, &count);
                if (stop < 0) return -1;
            }
            r->scan_end[scan] = (size_t)stop;
            r->scan_count++;
            p = image + stop;
            continue; }
        case 0xC2: GJ_ERROR("Marker SOF2 (Progressive with Huffman coding) is not supported!\n"); return -1;
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC8: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            GJ_ERROR("Marker %s (not baseline sequential Huffman) is not supported!\n", marker_name(m));
            return -1;
        case 0xCC: case 0xDC: GJ_WARN("JPEG data contains not supported %s marker\n", marker_name(m)); break;
        default:
            if (m >= 0xE0 && m <= 0xEF) {
                if (verbose > 0) GJ_WARN("JPEG data contains not supported %s marker\n", marker_name(m));
                break;
            }
            GJ_ERROR("JPEG data contains not supported %s marker!\n", marker_name(m));
            return -1;
        }
        p += len;
    }
    if (!have_sof || r->scan_count == 0) {
        GJ_ERROR("JPEG data contain no image!\n");
        return -1;
    }
    return 0;
}

/* segment table of every scan: by the APP13 index when present, else by walking the bytes */
int gj_reader_split_scans(const uint8_t* image, const struct gj_reader_result* r, const gj_geom* g, struct gj_host_segments* segs, int verbose)
{
    (void)verbose;
    segs->count = 0;
    for (int scan = 0; scan < r->scan_count; scan++) {
        /* scan i carries component i when not interleaved (reader.c:1345, decoder assumption) */
        const uint32_t first = g->interleaved ? 0u : (uint32_t)g->comp[scan < g->comp_count ? scan : 0].first_segment;
        const int max_segs = g->interleaved ? g->segment_count : g->comp[scan < g->comp_count ? scan : 0].segment_count;
        if (r->seg_info_count[scan] > 0) {
            int total = 0;
            for (int k = 0; k < r->seg_info_count[scan]; k++) total += r->seg_info_size[scan][k];
            const int count = total / 4 - 1;
            uint32_t prev = 0;
            for (int i = 0; i <= count; i++) { /* validated by seg_info_valid() when the scan was parsed */
                const uint32_t pos = seg_info_entry(r, scan, i);
                if (i > 0 && i - 1 < max_segs) {
                    uint32_t len = pos - prev;
                    if (i < count) len -= 2; /* all but the last keep their RSTn (reader.c:1204-1207) */
                    if (push_segment(segs, (uint32_t)r->scan_begin[scan] + prev, len, first + (uint32_t)(i - 1)) != 0) return -1;
                }
                prev = pos;
            }
        } else {
            int count = 0;
            if (walk_scan(image, r->scan_begin[scan], r->scan_end[scan] + 2, first, max_segs, segs, &count) < 0) return -1;
            if (count > max_segs && verbose >= 0) GJ_WARN("%d segments read, expected %d. Broken JPEG?\n", count, max_segs);
        }
    }
    return 0;
}
