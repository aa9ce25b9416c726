/* gj_image_png.c -- PNG (read and write) and GIF (read) for gpujpeg_image_load_from_file / save_to_file / get_properties.
 *
 * The reference delegates these formats to stb_image / stb_image_write (src/utils/image_delegate.c:476-553): files are read
 * with as many 8-bit channels as they carry (grey 1, RGB 3, RGBA 4; a palette becomes RGB or RGBA; 16-bit samples keep their
 * high byte; GIF always yields RGBA of the first frame), PNG files are written with 1, 3 or 4 channels of 8 bits. Everything
 * here is written from the format specifications: RFC 1950 / 1951 (zlib, deflate), the PNG specification (chunks, the five
 * scanline filters, Adam7 interlacing) and GIF89a (LZW with variable code width). The deflate writer uses fixed Huffman codes
 * and a single-probe hash match search: files are valid and compact enough, not byte-identical to stb's (neither tool
 * promises that). Host-only code; nothing on the GPU path depends on it. */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gj_internal.h"

/* ================================================================================================ inflate (RFC 1951) */
struct bitsrc {
    const uint8_t* p;
    size_t n, pos;
    uint32_t acc;
    int cnt;
    int bad;
};

static unsigned take(struct bitsrc* s, int need) /* least significant bit first */
{
    while (s->cnt < need) {
        if (s->pos >= s->n) { s->bad = 1; return 0; }
        s->acc |= (uint32_t)s->p[s->pos++] << s->cnt;
        s->cnt += 8;
    }
    const unsigned v = s->acc & ((1u << need) - 1u);
    s->acc >>= need;
    s->cnt -= need;
    return v;
}

/* canonical prefix code: count[l] codes of length l, symbols ordered by (length, value) */
struct prefix { uint16_t count[16]; uint16_t symbol[288]; };

static int prefix_build(struct prefix* h, const uint8_t* lengths, int n)
{
    memset(h->count, 0, sizeof h->count);
    for (int i = 0; i < n; i++) h->count[lengths[i]]++;
    h->count[0] = 0;
    int left = 1; /* over-subscription check */
    for (int l = 1; l < 16; l++) {
        left = left * 2 - h->count[l];
        if (left < 0) return -1;
    }
    uint16_t start[16];
    start[1] = 0;
    for (int l = 1; l < 15; l++) start[l + 1] = (uint16_t)(start[l] + h->count[l]);
    for (int i = 0; i < n; i++)
        if (lengths[i]) h->symbol[start[lengths[i]]++] = (uint16_t)i;
    return 0;
}

static int prefix_decode(struct bitsrc* s, const struct prefix* h)
{
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; l++) {
        code |= (int)take(s, 1);
        const int c = h->count[l];
        if (code - c < first) return h->symbol[index + (code - first)];
        index += c;
        first = (first + c) * 2;
        code *= 2;
        if (s->bad) return -1;
    }
    return -1;
}

/* out must hold `cap` bytes; returns the number of bytes produced or -1 */
static long inflate_raw(const uint8_t* src, size_t n, uint8_t* out, size_t cap)
{
    static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    struct bitsrc s = {src, n, 0, 0, 0, 0};
    size_t o = 0;
    int last;
    struct prefix* lit = malloc(sizeof *lit);
    struct prefix* dst = malloc(sizeof *dst);
    if (!lit || !dst) { free(lit); free(dst); return -1; }
    long result = -1;
    do {
        last = (int)take(&s, 1);
        const unsigned type = take(&s, 2);
        if (s.bad) goto done;
        if (type == 0) { /* stored */
            s.acc = 0;
            s.cnt = 0;
            if (s.pos + 4 > s.n) goto done;
            const unsigned len = (unsigned)s.p[s.pos] | (unsigned)s.p[s.pos + 1] << 8, nlen = (unsigned)s.p[s.pos + 2] | (unsigned)s.p[s.pos + 3] << 8;
            s.pos += 4;
            if ((len ^ nlen) != 0xFFFFu || s.pos + len > s.n || o + len > cap) goto done;
            memcpy(out + o, s.p + s.pos, len);
            s.pos += len;
            o += len;
            continue;
        }
        if (type == 3) goto done;
        uint8_t lengths[320];
        if (type == 1) { /* fixed code */
            for (int i = 0; i < 288; i++) lengths[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            prefix_build(lit, lengths, 288);
            for (int i = 0; i < 30; i++) lengths[i] = 5;
            prefix_build(dst, lengths, 30);
        } else { /* code lengths are themselves prefix coded */
            const int nlen = (int)take(&s, 5) + 257, ndist = (int)take(&s, 5) + 1, ncode = (int)take(&s, 4) + 4;
            if (nlen > 286 || ndist > 30) goto done;
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)take(&s, 3);
            if (prefix_build(lit, cl, 19) != 0) goto done;
            int idx = 0;
            while (idx < nlen + ndist) {
                const int sym = prefix_decode(&s, lit);
                if (sym < 0) goto done;
                if (sym < 16) { lengths[idx++] = (uint8_t)sym; continue; }
                int rep, val = 0;
                if (sym == 16) {
                    if (idx == 0) goto done;
                    val = lengths[idx - 1];
                    rep = 3 + (int)take(&s, 2);
                } else if (sym == 17) rep = 3 + (int)take(&s, 3);
                else rep = 11 + (int)take(&s, 7);
                if (idx + rep > nlen + ndist) goto done;
                while (rep--) lengths[idx++] = (uint8_t)val;
            }
            if (lengths[256] == 0) goto done;
            if (prefix_build(lit, lengths, nlen) != 0) goto done;
            if (prefix_build(dst, lengths + nlen, ndist) != 0) goto done;
        }
        for (;;) {
            const int sym = prefix_decode(&s, lit);
            if (sym < 0 || s.bad) goto done;
            if (sym < 256) {
                if (o >= cap) goto done;
                out[o++] = (uint8_t)sym;
            } else if (sym == 256) {
                break;
            } else {
                if (sym > 285) goto done;
                const unsigned len = len_base[sym - 257] + take(&s, len_extra[sym - 257]);
                const int ds = prefix_decode(&s, dst);
                if (ds < 0 || ds > 29) goto done;
                const unsigned dist = dist_base[ds] + take(&s, dist_extra[ds]);
                if (dist > o || o + len > cap || s.bad) goto done;
                for (unsigned i = 0; i < len; i++, o++) out[o] = out[o - dist];
            }
        }
    } while (!last);
    result = (long)o;
done:
    free(lit);
    free(dst);
    return result;
}

/* ================================================================================================ PNG reader */
static uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

static int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : pb <= pc ? b : c;
}

/* undo the scanline filters of one (sub)image of `rows` lines of `rowbytes` bytes, in place; `data` has a filter byte per line */
static int png_unfilter(uint8_t* data, size_t rowbytes, int rows, int bpp)
{
    for (int y = 0; y < rows; y++) {
        uint8_t* cur = data + (size_t)y * (rowbytes + 1);
        const uint8_t* up = y ? cur - rowbytes : NULL; /* previous line without its filter byte (already reconstructed) */
        const int f = cur[0];
        uint8_t* x = cur + 1;
        for (size_t i = 0; i < rowbytes; i++) {
            const int a = i >= (size_t)bpp ? x[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
            switch (f) {
            case 0: break;
            case 1: x[i] = (uint8_t)(x[i] + a); break;
            case 2: x[i] = (uint8_t)(x[i] + b); break;
            case 3: x[i] = (uint8_t)(x[i] + ((a + b) >> 1)); break;
            case 4: x[i] = (uint8_t)(x[i] + paeth(a, b, c)); break;
            default: return -1;
            }
        }
    }
    return 0;
}

int gj_png_decode(const uint8_t* d, size_t n, struct gj_raster* out, int want_pixels)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 + 25 || memcmp(d, sig, 8) != 0) return -1;
    size_t pos = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0, have_trns = 0, pal_n = 0;
    uint8_t pal[256][4];
    uint16_t trns_key[3] = {0, 0, 0};
    uint8_t* idat = NULL;
    size_t idat_n = 0;
    int seen_ihdr = 0, rc = -1;
    memset(pal, 255, sizeof pal);
    while (pos + 12 <= n) {
        const uint32_t len = be32(d + pos);
        const uint8_t* type = d + pos + 4;
        const uint8_t* body = d + pos + 8;
        if (len > n - pos - 12) goto done;
        if (memcmp(type, "IHDR", 4) == 0 && len >= 13) {
            w = (int)be32(body); h = (int)be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
            if (w <= 0 || h <= 0 || body[10] != 0 || body[11] != 0 || interlace > 1) goto done;
            if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) goto done;
            if (!(ctype == 0 || ctype == 2 || ctype == 3 || ctype == 4 || ctype == 6)) goto done;
            if ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8) goto done;
            if (ctype == 3 && depth == 16) goto done;
            seen_ihdr = 1;
        } else if (memcmp(type, "PLTE", 4) == 0) {
            pal_n = (int)(len / 3);
            if (pal_n > 256) goto done;
            for (int i = 0; i < pal_n; i++) { pal[i][0] = body[3 * i]; pal[i][1] = body[3 * i + 1]; pal[i][2] = body[3 * i + 2]; }
        } else if (memcmp(type, "tRNS", 4) == 0) {
            have_trns = 1;
            if (ctype == 3) { for (uint32_t i = 0; i < len && i < 256; i++) pal[i][3] = body[i]; }
            else if (ctype == 0 && len >= 2) trns_key[0] = (uint16_t)(body[0] << 8 | body[1]);
            else if (ctype == 2 && len >= 6) for (int k = 0; k < 3; k++) trns_key[k] = (uint16_t)(body[2 * k] << 8 | body[2 * k + 1]);
        } else if (memcmp(type, "IDAT", 4) == 0) {
            if (!want_pixels) { /* the header is complete */ }
            else {
                uint8_t* ni = realloc(idat, idat_n + len + 1);
                if (!ni) goto done;
                idat = ni;
                memcpy(idat + idat_n, body, len);
                idat_n += len;
            }
        } else if (memcmp(type, "IEND", 4) == 0) {
            break;
        }
        pos += 12 + (size_t)len;
        if (!want_pixels && seen_ihdr && (memcmp(type, "IDAT", 4) == 0)) break;
    }
    if (!seen_ihdr) goto done;
    const int chan_in = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
    int comps = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 3; /* what stb reports */
    if (have_trns && ctype != 4 && ctype != 6) comps += 1;
    out->w = w; out->h = h; out->comps = comps; out->px = NULL;
    if (!want_pixels) { rc = 0; goto done; }
    if (idat_n < 2) goto done;

    /* sizes of the seven Adam7 passes (or of the one non-interlaced image) */
    static const int px0[7] = {0, 4, 0, 2, 0, 1, 0}, py0[7] = {0, 0, 4, 0, 2, 0, 1}, pdx[7] = {8, 8, 4, 4, 2, 2, 1}, pdy[7] = {8, 8, 8, 4, 4, 2, 2};
    const int passes = interlace ? 7 : 1;
    const int bits_px = chan_in * depth;
    size_t raw_size = 0;
    for (int p = 0; p < passes; p++) {
        const int pw = interlace ? (w - px0[p] + pdx[p] - 1) / pdx[p] : w, ph = interlace ? (h - py0[p] + pdy[p] - 1) / pdy[p] : h;
        if (pw > 0 && ph > 0) raw_size += ((size_t)(((size_t)pw * bits_px + 7) / 8) + 1) * (size_t)ph;
    }
    uint8_t* raw = malloc(raw_size ? raw_size : 1);
    uint8_t* px = malloc((size_t)w * h * comps);
    if (!raw || !px) { free(raw); free(px); goto done; }
    if (inflate_raw(idat + 2, idat_n - 2, raw, raw_size) != (long)raw_size) { free(raw); free(px); goto done; }
    const int bpp = bits_px >= 8 ? bits_px / 8 : 1;
    size_t off = 0;
    for (int p = 0; p < passes; p++) {
        const int pw = interlace ? (w - px0[p] + pdx[p] - 1) / pdx[p] : w, ph = interlace ? (h - py0[p] + pdy[p] - 1) / pdy[p] : h;
        if (pw <= 0 || ph <= 0) continue;
        const size_t rowbytes = ((size_t)pw * bits_px + 7) / 8;
        if (png_unfilter(raw + off, rowbytes, ph, bpp) != 0) { free(raw); free(px); goto done; }
        for (int yy = 0; yy < ph; yy++) {
            const uint8_t* row = raw + off + (size_t)yy * (rowbytes + 1) + 1;
            const int y = interlace ? py0[p] + yy * pdy[p] : yy;
            for (int xx = 0; xx < pw; xx++) {
                const int x = interlace ? px0[p] + xx * pdx[p] : xx;
                uint16_t s16[4] = {0, 0, 0, 0}; /* samples at file precision */
                for (int c = 0; c < chan_in; c++) {
                    const size_t bit = ((size_t)xx * chan_in + c) * depth;
                    if (depth == 16) s16[c] = (uint16_t)(row[bit / 8] << 8 | row[bit / 8 + 1]);
                    else if (depth == 8) s16[c] = row[bit / 8];
                    else s16[c] = (uint16_t)((row[bit / 8] >> (8 - depth - (int)(bit % 8))) & ((1 << depth) - 1));
                }
                uint8_t* o = px + ((size_t)y * w + x) * comps;
                if (ctype == 3) {
                    const uint8_t* e = pal[s16[0] & 255];
                    o[0] = e[0]; o[1] = e[1]; o[2] = e[2];
                    if (comps == 4) o[3] = e[3];
                } else {
                    int transparent = have_trns && (ctype == 0 || ctype == 2);
                    for (int c = 0; c < chan_in; c++) {
                        if (transparent && s16[c] != trns_key[c]) transparent = 0;
                        /* 16 bit: high byte; below 8 bit: spread over 0..255 */
                        o[c] = depth == 16 ? (uint8_t)(s16[c] >> 8) : depth == 8 ? (uint8_t)s16[c] : (uint8_t)(s16[c] * 255 / ((1 << depth) - 1));
                    }
                    if (have_trns && (ctype == 0 || ctype == 2)) o[chan_in] = transparent ? 0 : 255;
                }
            }
        }
        off += (rowbytes + 1) * (size_t)ph;
    }
    free(raw);
    out->px = px;
    rc = 0;
done:
    free(idat);
    return rc;
}

/* ================================================================================================ PNG writer */
static uint32_t crc_table[256];
static void crc_init(void)
{
    if (crc_table[1]) return;
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_table[i] = c;
    }
}
static uint32_t crc_update(uint32_t c, const uint8_t* p, size_t n)
{
    for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 255] ^ (c >> 8);
    return c;
}

struct bitdst { uint8_t* p; size_t n, cap; uint32_t acc; int cnt; };
static int put_bits(struct bitdst* b, uint32_t v, int n) /* least significant bit first */
{
    b->acc |= v << b->cnt;
    b->cnt += n;
    while (b->cnt >= 8) {
        if (b->n == b->cap) {
            const size_t nc = b->cap * 2 + 4096;
            uint8_t* np = realloc(b->p, nc);
            if (!np) return -1;
            b->p = np;
            b->cap = nc;
        }
        b->p[b->n++] = (uint8_t)b->acc;
        b->acc >>= 8;
        b->cnt -= 8;
    }
    return 0;
}
static uint32_t rev(uint32_t v, int n) /* Huffman codes go out most significant bit first */
{
    uint32_t r = 0;
    for (int i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}
static int put_fixed_literal(struct bitdst* b, int sym) /* fixed code of RFC 1951, 3.2.6 */
{
    if (sym < 144) return put_bits(b, rev(0x30u + (uint32_t)sym, 8), 8);
    if (sym < 256) return put_bits(b, rev(0x190u + (uint32_t)(sym - 144), 9), 9);
    if (sym < 280) return put_bits(b, rev((uint32_t)(sym - 256), 7), 7);
    return put_bits(b, rev(0xC0u + (uint32_t)(sym - 280), 8), 8);
}

/* zlib stream of `src` with one fixed-Huffman block; greedy matches found through a hash of three bytes (one candidate each) */
static int deflate_fixed(const uint8_t* src, size_t n, struct bitdst* b)
{
    static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    enum { HASH_BITS = 15 };
    int32_t* head = malloc(sizeof(int32_t) << HASH_BITS);
    if (!head) return -1;
    for (size_t i = 0; i < ((size_t)1 << HASH_BITS); i++) head[i] = -1;
    int rc = put_bits(b, 0x78, 8) | put_bits(b, 0x01, 8); /* zlib header: deflate, 32 KiB window, no preset dictionary */
    rc |= put_bits(b, 1, 1) | put_bits(b, 1, 2);          /* last block, fixed codes */
    uint32_t s1 = 1, s2 = 0;                              /* Adler-32 */
    size_t i = 0;
    while (i < n && rc == 0) {
        size_t best = 0, dist = 0;
        if (i + 3 <= n) {
            const uint32_t hsh = (((uint32_t)src[i] << 16 | (uint32_t)src[i + 1] << 8 | src[i + 2]) * 2654435761u) >> (32 - HASH_BITS);
            const int32_t cand = head[hsh];
            head[hsh] = (int32_t)i;
            if (cand >= 0 && i - (size_t)cand <= 32768) {
                const size_t max = n - i < 258 ? n - i : 258;
                size_t l = 0;
                while (l < max && src[(size_t)cand + l] == src[i + l]) l++;
                if (l >= 3) { best = l; dist = i - (size_t)cand; }
            }
        }
        if (best) {
            int ls = 28;
            while (len_base[ls] > best) ls--;
            rc |= put_fixed_literal(b, 257 + ls) | put_bits(b, (uint32_t)(best - len_base[ls]), len_extra[ls]);
            int ds = 29;
            while (dist_base[ds] > dist) ds--;
            rc |= put_bits(b, rev((uint32_t)ds, 5), 5) | put_bits(b, (uint32_t)(dist - dist_base[ds]), dist_extra[ds]);
        } else {
            rc |= put_fixed_literal(b, src[i]);
            best = 1;
        }
        for (size_t k = 0; k < best; k++) {
            s1 += src[i + k];
            if (s1 >= 65521u) s1 -= 65521u;
            s2 += s1;
            if (s2 >= 65521u) s2 -= 65521u;
        }
        i += best;
    }
    rc |= put_fixed_literal(b, 256);
    if (b->cnt) rc |= put_bits(b, 0, 8 - b->cnt);
    const uint32_t adler = s2 << 16 | s1;
    for (int k = 3; k >= 0; k--) rc |= put_bits(b, (adler >> (8 * k)) & 255u, 8);
    free(head);
    return rc;
}

static int png_chunk(FILE* f, const char* type, const uint8_t* body, size_t n)
{
    uint8_t hdr[8] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n, (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
    uint32_t c = crc_update(0xFFFFFFFFu, hdr + 4, 4);
    c = crc_update(c, body, n) ^ 0xFFFFFFFFu;
    const uint8_t tail[4] = {(uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c};
    return (fwrite(hdr, 1, 8, f) == 8 && fwrite(body, 1, n, f) == n && fwrite(tail, 1, 4, f) == 4) ? 0 : -1;
}

/* 8-bit grey (1), RGB (3) or RGBA (4); every scanline takes the filter (none / sub / up) with the smallest sum of magnitudes */
int gj_png_save(const char* filename, const uint8_t* image, int w, int h, int comps, size_t pitch)
{
    crc_init();
    const size_t rowbytes = (size_t)w * comps;
    uint8_t* filtered = malloc((rowbytes + 1) * (size_t)h);
    uint8_t* trial = malloc(rowbytes);
    if (!filtered || !trial) { free(filtered); free(trial); return -1; }
    for (int y = 0; y < h; y++) {
        const uint8_t* cur = image + (size_t)y * pitch;
        const uint8_t* up = y ? cur - pitch : NULL;
        uint8_t* dst = filtered + (size_t)y * (rowbytes + 1);
        unsigned long best_cost = ~0ul;
        for (int f = 0; f < 3; f++) {
            if (f == 2 && !up) continue;
            unsigned long cost = 0;
            for (size_t i = 0; i < rowbytes; i++) {
                const int pred = f == 0 ? 0 : f == 1 ? (i >= (size_t)comps ? cur[i - comps] : 0) : up[i];
                trial[i] = (uint8_t)(cur[i] - pred);
                cost += (unsigned long)abs((int)(int8_t)trial[i]);
            }
            if (cost < best_cost) {
                best_cost = cost;
                dst[0] = (uint8_t)f;
                memcpy(dst + 1, trial, rowbytes);
            }
        }
    }
    free(trial);
    struct bitdst b = {NULL, 0, 0, 0, 0};
    const int rc = deflate_fixed(filtered, (rowbytes + 1) * (size_t)h, &b);
    free(filtered);
    if (rc != 0) { free(b.p); return -1; }
    FILE* f = fopen(filename, "wb");
    if (!f) { free(b.p); GJ_ERROR("Failed open %s for writing: %s\n", filename, strerror(errno)); return -1; }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h,
                        8, (uint8_t)(comps == 1 ? 0 : comps == 3 ? 2 : 6), 0, 0, 0};
    int wrc = fwrite(sig, 1, 8, f) == 8 ? 0 : -1;
    wrc |= png_chunk(f, "IHDR", ihdr, 13);
    wrc |= png_chunk(f, "IDAT", b.p, b.n);
    wrc |= png_chunk(f, "IEND", NULL, 0);
    free(b.p);
    if (fclose(f) != 0) wrc = -1;
    return wrc;
}

/* ================================================================================================ GIF reader (first frame) */
int gj_gif_decode(const uint8_t* d, size_t n, struct gj_raster* out, int want_pixels)
{
    if (n < 13 || memcmp(d, "GIF8", 4) != 0 || (d[4] != '7' && d[4] != '9') || d[5] != 'a') return -1;
    const int sw = d[6] | d[7] << 8, sh = d[8] | d[9] << 8;
    if (sw <= 0 || sh <= 0) return -1;
    out->w = sw; out->h = sh; out->comps = 4; out->px = NULL; /* stb always reports RGBA for GIF */
    if (!want_pixels) return 0;
    uint8_t gpal[256][3], lpal[256][3];
    memset(gpal, 0, sizeof gpal);
    size_t pos = 13;
    if (d[10] & 0x80) {
        const int cnt = 2 << (d[10] & 7);
        if (pos + (size_t)cnt * 3 > n) return -1;
        memcpy(gpal, d + pos, (size_t)cnt * 3);
        pos += (size_t)cnt * 3;
    }
    int transparent = -1;
    uint8_t* px = calloc((size_t)sw * sh, 4); /* undrawn pixels stay (0, 0, 0, 0) like in stb's first frame */
    if (!px) return -1;
    while (pos < n) {
        const int tag = d[pos++];
        if (tag == 0x21) { /* extension: a graphic control extension may name a transparent index */
            if (pos >= n) break;
            const int label = d[pos++];
            while (pos < n && d[pos] != 0) {
                const int len = d[pos];
                if (label == 0xF9 && len >= 4 && pos + 4 < n) transparent = (d[pos + 1] & 1) ? d[pos + 4] : -1;
                pos += 1 + (size_t)len;
            }
            pos++;
        } else if (tag == 0x2C) {
            if (pos + 9 > n) break;
            const int ix = d[pos] | d[pos + 1] << 8, iy = d[pos + 2] | d[pos + 3] << 8, iw = d[pos + 4] | d[pos + 5] << 8, ih = d[pos + 6] | d[pos + 7] << 8;
            const int flags = d[pos + 8];
            pos += 9;
            const uint8_t(*palette)[3] = gpal;
            if (flags & 0x80) {
                const int cnt = 2 << (flags & 7);
                if (pos + (size_t)cnt * 3 > n) break;
                memset(lpal, 0, sizeof lpal);
                memcpy(lpal, d + pos, (size_t)cnt * 3);
                pos += (size_t)cnt * 3;
                palette = lpal;
            }
            if (pos >= n || ix + iw > sw || iy + ih > sh) break;
            const int min_bits = d[pos++];
            if (min_bits < 2 || min_bits > 8) break;
            /* LZW: codes of growing width, least significant bit first, packed into sub-blocks of up to 255 bytes */
            static const int ilace_start[4] = {0, 4, 2, 1}, ilace_step[4] = {8, 8, 4, 2};
            struct { int16_t prefix; uint8_t first, suffix; } tab[4096];
            uint8_t stack[4096];
            const int clear = 1 << min_bits, eoi = clear + 1;
            int avail = clear + 2, width = min_bits + 1, old = -1;
            for (int i = 0; i < clear; i++) { tab[i].prefix = -1; tab[i].first = tab[i].suffix = (uint8_t)i; }
            uint32_t acc = 0;
            int cnt = 0, block = 0, ended = 0;
            long drawn = 0;
            const long total = (long)iw * ih;
            int pass = 0, line = (flags & 0x40) ? ilace_start[0] : 0, col = 0;
            while (!ended) {
                while (cnt < width) {
                    if (block == 0) {
                        if (pos >= n) { ended = 1; break; }
                        block = d[pos++];
                        if (block == 0) { ended = 1; break; }
                    }
                    if (pos >= n) { ended = 1; break; }
                    acc |= (uint32_t)d[pos++] << cnt;
                    cnt += 8;
                    block--;
                }
                if (ended) break;
                int code = (int)(acc & ((1u << width) - 1u));
                acc >>= width;
                cnt -= width;
                if (code == clear) { avail = clear + 2; width = min_bits + 1; old = -1; continue; }
                if (code == eoi) break;
                if (code > avail || (code == avail && old < 0)) break; /* damaged stream */
                int sp = 0, cur = code;
                if (code == avail) { stack[sp++] = tab[old].first; cur = old; } /* the "KwKwK" case */
                while (cur >= 0 && sp < 4096) { stack[sp++] = tab[cur].suffix; cur = tab[cur].prefix; }
                const uint8_t first = stack[sp - 1];
                if (old >= 0 && avail < 4096) {
                    tab[avail].prefix = (int16_t)old;
                    tab[avail].first = tab[old].first;
                    tab[avail].suffix = first;
                    avail++;
                    if (avail == (1 << width) && width < 12) width++;
                }
                old = code;
                while (sp > 0 && drawn < total) {
                    const int idx = stack[--sp];
                    if (idx != transparent) {
                        uint8_t* o = px + ((size_t)(iy + line) * sw + (size_t)(ix + col)) * 4;
                        o[0] = palette[idx][0]; o[1] = palette[idx][1]; o[2] = palette[idx][2]; o[3] = 255;
                    }
                    drawn++;
                    if (++col == iw) {
                        col = 0;
                        if (flags & 0x40) {
                            line += ilace_step[pass];
                            while (line >= ih && pass < 3) { pass++; line = ilace_start[pass]; }
                        } else {
                            line++;
                        }
                    }
                }
                if (drawn >= total) break;
            }
            out->px = px;
            return 0; /* first frame only */
        } else {
            break; /* trailer or garbage */
        }
    }
    free(px);
    return -1;
}
