/*
 * gj_decoder.c -- the libgpujpeg decoder API on MI355X. Host driver counterpart of
 * src/gpujpeg_decoder.c: same entry points, output types, ownership and error behaviour. The JPEG
 * bytes are uploaded once and decoded in place (no per-segment host memcpy as in
 * src/gpujpeg_reader.c:1115,1128).
 */
#define _GNU_SOURCE
#include <assert.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include "gj_internal.h"
#include "gpujpeg_amd_ext.h"

struct gpujpeg_decoder {
    struct gj_coder coder;
    enum gpujpeg_pixel_format req_pixel_format;
    enum gpujpeg_color_space req_color_space;
    bool ff_cs_itu601_is_709;
    unsigned req_alignment;
    struct gpujpeg_image_metadata metadata;
    /* device */
    uint8_t* d_jpeg; size_t d_jpeg_cap;
    uint32_t* d_seg; size_t d_seg_cap; /* pos | len | index */
    uint16_t* d_huff_tab;
    uint16_t* d_qtab;
    /* host */
    uint8_t* h_raw; size_t h_raw_cap; /* pinned internal output buffer */
    uint32_t* h_seg; size_t h_seg_cap; /* pinned staging of the segment table */
    uint16_t* h_tabs;                  /* pinned staging: 8 decode tables + 4 quant tables */
    struct gj_host_segments segs;
    int use_fused;
    bool flipped;                 /* dec_opt_flipped */
    unsigned channel_remap;       /* dec_opt_channel_remap, packed; 0 = none */
    int keep_coefs;               /* 1: leave the coefficients in HBM after the call (gpujpeg_amd_decoder_keep_coefficients) */
    uint16_t* d_tok; size_t d_tok_cap;     /* token mode (gj_hip.h): non-zero AC coefficients of the frame */
    void* d_blkrec; size_t d_blkrec_cap;   /* token mode: one record per block */
    bool coefs_clean;             /* d_coefs is all zero: the previous call's IDCT cleared what it read */
    /* device-side segment discovery */
    uint8_t* h_hdr;               /* pinned: first bytes of a device-resident stream, for header parsing */
    uint32_t* d_scan_scratch; size_t d_scan_scratch_cap;
    gj_scan_summary* d_summary;   /* the device's copy: only segment_count is used (speculative launches read it) */
    bool scan_timed;              /* this call's marker scan ran on the device between events 4 and 5 */
    uint32_t last_max_seg_len;    /* longest segment of the last frame decoded with this header (speculative path) */
    uint32_t last_scan_bytes[GJ_MAX_COMP]; /* entropy-coded bytes per scan of the last frame decoded with this header (speculative path) */
    gj_scan_summary* h_summary;   /* pinned */
    uint32_t* h_maxlen; size_t h_maxlen_cap; /* pinned: per-chunk longest segments of the marker scan */
    uint32_t maxlen_parts;
    int host_scan;                /* 1: always walk the stream on the host (reference behaviour) */
    gj_tuning tune;               /* developer switches, read from the environment when the decoder is created */
    /* header cache: a stream that starts with the same bytes (SOI .. first SOS header) as the previous one has the same
     * tables and geometry, so the call goes straight to the kernels and is validated after the fact */
    uint8_t* hdr_cache; uint8_t* d_hdr_cache; size_t hdr_cache_len; bool hdr_cache_valid;
    bool need_planes; /* a frame with the cached header raised the entropy decoders' overflow flag (a segment too long for the LDS stage, a coefficient
                         too large for a token): the following frames with that header go through the kernels without those limits at once,
                         instead of being decoded twice each */
    struct gj_reader_result hdr_cache_r;
    bool tab2_ok;
    /* frame batches (gpujpeg_amd_decoder_decode_batch): per-frame words for ALL frames of a call, work buffers for one chunk of frames */
    gj_scan_summary* b_dsum; size_t b_dsum_cap;    /* device: segment counts */
    gj_scan_summary* bh_sum; size_t bh_sum_cap;    /* pinned: what the host validates */
    uint32_t* bh_maxlen; size_t bh_maxlen_cap;     /* pinned: longest segments per scanning workgroup */
    uint32_t* b_sizes; size_t b_sizes_cap;         /* device: stream sizes */
    uint32_t* bh_sizes; size_t bh_sizes_cap;       /* pinned staging of the same */
    uint32_t* b_seg; size_t b_seg_cap;
    uint32_t* b_scratch; size_t b_scratch_cap;
    int16_t* b_coefs; size_t b_coefs_cap;
    uint8_t* b_planes; size_t b_planes_cap;        /* component planes per frame (configurations whose IDCT side is the generic kernels) */
    uint16_t* b_tok; size_t b_tok_cap;
    void* b_rec; size_t b_rec_cap;
    uint8_t* b_jpeg; size_t b_jpeg_cap;            /* streams handed over in host memory */
    uint8_t* b_raw; size_t b_raw_cap;              /* pixels wanted in host memory */
    uint8_t* b_gather; size_t b_gather_cap;        /* streams given as separate buffers (decode_batch_ptrs), gathered 16-byte aligned */
    uint8_t* b_scatter; size_t b_scatter_cap;      /* ... and the frames decoded back to back before they go to separate buffers */
    int b_last_batched, b_last_single;             /* frames of the last batch call that the batched launches decoded / that went the ordinary way */
    int last_folded;                               /* the last decode call did without the k_marker_table launch (gj_scan_deferred) */
    long n_spec, n_folded, n_again;                /* calls launched on the cached header / of those without the table launch / decoded again the careful way */
    int b_chunk;                                   /* gpujpeg_amd_decoder_set_batch_chunk: frames per launch at most, 0 = the default */
};

#define GJ_HDR_WINDOW 65536

/* ------------------------------------------------------------------ output helpers (gpujpeg_decoder.h:105-143) */
void gpujpeg_decoder_output_set_default(struct gpujpeg_decoder_output* o) { o->type = GPUJPEG_DECODER_OUTPUT_INTERNAL_BUFFER; o->data = NULL; o->data_size = 0; o->texture = NULL; }
void gpujpeg_decoder_output_set_custom(struct gpujpeg_decoder_output* o, uint8_t* buf) { o->type = GPUJPEG_DECODER_OUTPUT_CUSTOM_BUFFER; o->data = buf; o->data_size = 0; o->texture = NULL; }
void gpujpeg_decoder_output_set_texture(struct gpujpeg_decoder_output* o, struct gpujpeg_opengl_texture* t) { o->type = GPUJPEG_DECODER_OUTPUT_OPENGL_TEXTURE; o->data = NULL; o->data_size = 0; o->texture = t; }
void gpujpeg_decoder_output_set_cuda_buffer(struct gpujpeg_decoder_output* o) { o->type = GPUJPEG_DECODER_OUTPUT_CUDA_BUFFER; o->data = NULL; o->data_size = 0; o->texture = NULL; }
void gpujpeg_decoder_output_set_custom_cuda(struct gpujpeg_decoder_output* o, uint8_t* d_buf) { o->type = GPUJPEG_DECODER_OUTPUT_CUSTOM_CUDA_BUFFER; o->data = d_buf; o->data_size = 0; o->texture = NULL; }

/* ------------------------------------------------------------------ create / destroy (src/gpujpeg_decoder.c:97-183, 560-584) */
#define GJ_TABS_QF_OFFSET (8 * GJ_DEC_TAB_WORDS + 4 * 64 + 4 * GJ_DEC2_WORDS) /* u16 words; 4-byte aligned */
#define GJ_TABS_WORDS (GJ_TABS_QF_OFFSET + 4 * 64 * 2)

struct gpujpeg_decoder* gpujpeg_decoder_create(cudaStream_t stream)
{
    gj_init_term_colors();
    struct gpujpeg_decoder* d = calloc(1, sizeof *d);
    if (!d) return NULL;
    d->coder.stream = (gj_stream_t)stream;
    d->req_pixel_format = GPUJPEG_PIXFMT_AUTODETECT;
    d->req_color_space = GPUJPEG_CS_DEFAULT;
    gj_hip_tuning_defaults(&d->tune);
    d->coder.ht_on = d->tune.host_timing != 0;
    d->use_fused = !d->tune.no_fused;
    gpujpeg_set_default_parameters(&d->coder.param);
    gpujpeg_image_set_default_parameters(&d->coder.param_image);
    d->coder.param.comp_count = 0;
    d->coder.param.restart_interval = 0;
    if (gj_hip_get_device(&d->coder.device) != 0) goto fail;
    if (gj_timers_create(&d->coder.timers) != 0) goto fail;
    d->d_huff_tab = gj_hip_malloc(GJ_TABS_WORDS * sizeof(uint16_t));
    d->h_tabs = gj_hip_host_alloc(GJ_TABS_WORDS * sizeof(uint16_t));
    if (!d->d_huff_tab || !d->h_tabs) goto fail;
    d->d_qtab = d->d_huff_tab + 8 * GJ_DEC_TAB_WORDS;
    d->h_hdr = gj_hip_host_alloc(GJ_HDR_WINDOW);
    d->d_summary = gj_hip_malloc(sizeof(gj_scan_summary));
    d->h_summary = gj_hip_host_alloc(sizeof(gj_scan_summary));
    if (!d->h_hdr || !d->d_summary || !d->h_summary) goto fail;
    if (gj_hip_memset(d->d_summary, 0, sizeof(gj_scan_summary), d->coder.stream) != 0 || gj_hip_stream_sync(d->coder.stream) != 0) goto fail;
    d->host_scan = d->tune.host_scan;
    return d;
fail:
    GJ_ERROR("Decoder initialisation failed: %s\n", gj_hip_last_error());
    gpujpeg_decoder_destroy(d);
    return NULL;
}

struct gpujpeg_decoder_init_parameters gpujpeg_decoder_default_init_parameters(void)
{
    return (struct gpujpeg_decoder_init_parameters){(cudaStream_t)0, 0, false, false};
}

struct gpujpeg_decoder* gpujpeg_decoder_create_with_params(const struct gpujpeg_decoder_init_parameters* params)
{
    struct gpujpeg_decoder* d = gpujpeg_decoder_create(params->stream);
    if (!d) return NULL;
    d->coder.param.verbose = params->verbose;
    d->coder.param.perf_stats = params->perf_stats;
    d->ff_cs_itu601_is_709 = params->ff_cs_itu601_is_709;
    return d;
}

int gpujpeg_decoder_destroy(struct gpujpeg_decoder* d)
{
    if (!d) return -1;
    gj_coder_process_stats_overall(&d->coder);
    gj_timers_destroy(&d->coder.timers);
    gj_hip_free(d->d_jpeg); gj_hip_free(d->d_seg); gj_hip_free(d->d_huff_tab);
    gj_hip_free(d->coder.d_raw_own); gj_hip_free(d->coder.d_planes); gj_hip_free(d->coder.d_coefs);
    gj_hip_free(d->d_tok); gj_hip_free(d->d_blkrec);
    gj_hip_host_free(d->h_raw); gj_hip_host_free(d->h_seg); gj_hip_host_free(d->h_tabs);
    free(d->hdr_cache); gj_hip_free(d->d_hdr_cache);
    gj_hip_host_free(d->h_hdr); gj_hip_host_free(d->h_summary); gj_hip_host_free(d->h_maxlen); gj_hip_free(d->d_summary); gj_hip_free(d->d_scan_scratch);
    free(d->segs.pos); free(d->segs.len); free(d->segs.index);
    gj_hip_free(d->b_dsum); gj_hip_free(d->b_sizes); gj_hip_free(d->b_seg); gj_hip_free(d->b_scratch); gj_hip_free(d->b_coefs); gj_hip_free(d->b_planes); gj_hip_free(d->b_tok);
    gj_hip_free(d->b_rec); gj_hip_free(d->b_jpeg); gj_hip_free(d->b_raw); gj_hip_free(d->b_gather); gj_hip_free(d->b_scatter);
    gj_hip_host_free(d->bh_sum); gj_hip_host_free(d->bh_maxlen); gj_hip_host_free(d->bh_sizes);
    free(d);
    return 0;
}

void gpujpeg_decoder_set_output_format(struct gpujpeg_decoder* d, enum gpujpeg_color_space cs, enum gpujpeg_pixel_format pf)
{
    d->req_color_space = cs;
    d->req_pixel_format = pf;
    d->hdr_cache_valid = false; /* the cached parse result embeds the requested output format */
}

/* ------------------------------------------------------------------ configuration (src/gpujpeg_decoder.c:185-233) */
static int decoder_configure(struct gpujpeg_decoder* d, const struct gpujpeg_parameters* p, const struct gpujpeg_image_parameters* pi)
{
    struct gj_coder* c = &d->coder;
    const int verbose = c->param.verbose, perf = c->param.perf_stats;
    if (c->configured && gj_parameters_equal(&c->param, p) && gj_image_parameters_equal(&c->param_image, pi)) return 0;
    if (c->configured && verbose >= GPUJPEG_LL_INFO) fprintf(stderr, "[GPUJPEG] [Info] Reinitializing decoder.\n");
    c->configured = false;
    d->coefs_clean = false;
    c->param = *p;
    c->param.verbose = verbose;
    c->param.perf_stats = perf;
    c->param_image = *pi;
    gj_geom* g = &c->geom;
    if (gj_geom_init(g, p, pi, false) != 0) {
        GJ_ERROR("Failed to init coder image!\n");
        return -1;
    }
    if (gj_ensure_device_buffer((void**)&c->d_coefs, &c->d_coefs_cap, g->data_size * sizeof(int16_t)) != 0) return -1;
    if (gj_ensure_device_buffer((void**)&c->d_planes, &c->d_planes_cap, g->data_size) != 0) return -1;
    if (gj_ensure_device_buffer((void**)&d->d_seg, &d->d_seg_cap, ((size_t)g->segment_count * 3 + 4) * sizeof(uint32_t)) != 0) return -1;
    if ((size_t)g->segment_count * 3 * sizeof(uint32_t) > d->h_seg_cap) {
        gj_hip_host_free(d->h_seg);
        d->h_seg_cap = (size_t)g->segment_count * 3 * sizeof(uint32_t);
        d->h_seg = gj_hip_host_alloc(d->h_seg_cap);
        if (!d->h_seg) { d->h_seg_cap = 0; return -1; }
    }
    c->configured = true;
    return 0;
}

int gpujpeg_decoder_init(struct gpujpeg_decoder* d, const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* pi)
{
    d->coder.param.verbose = param->verbose;
    d->coder.param.perf_stats = param->perf_stats || param->verbose >= GPUJPEG_LL_STATUS;
    if ((size_t)pi->width * pi->height * param->comp_count == 0) return 0;
    if (decoder_configure(d, param, pi) != 0) return -1;
    gpujpeg_decoder_set_output_format(d, pi->color_space, pi->pixel_format);
    return 0;
}

/* ------------------------------------------------------------------ decode (src/gpujpeg_decoder.c:235-465) */

/* Validate what the device found: scans in component order, each introduced by a well-formed SOS, EOI at the end.
 * Fills the Huffman table selectors of the later scans. Returns 0 when the device table can be used. */
static int accept_device_scan(const gj_scan_summary* su, struct gj_reader_result* r, const gj_geom* g)
{
    if (su->status != 1 || su->other_count > GJ_SCAN_MAX_OTHER || su->rst_irregular) return -1;
    const int expect_scans = g->interleaved ? 1 : g->comp_count;
    if ((int)su->scan_count != expect_scans) return -1;
    if (su->segment_count > (uint32_t)g->segment_count) return -1;
    /* order the markers by position and check them one by one */
    int order[GJ_SCAN_MAX_OTHER];
    const int n = (int)su->other_count;
    for (int i = 0; i < n; i++) {
        int j = i;
        while (j > 0 && su->other_pos[order[j - 1]] > su->other_pos[i]) { order[j] = order[j - 1]; j--; }
        order[j] = i;
    }
    int scan = 1;
    for (int i = 0; i < n; i++) {
        const int k = order[i];
        const uint8_t* b = su->other_bytes[k];
        if (su->other_code[k] == 0xD9) return (scan == expect_scans) ? 0 : -1;
        if (su->other_code[k] != 0xDA || scan >= expect_scans) return -1;
        /* SOS of a non-interleaved scan: length 8, one component, id of component `scan`, tables, 0, 63, 0 */
        if (b[0] != 0 || b[1] != 8 || b[2] != 1 || b[3] != r->comp_id[scan] || b[5] != 0 || b[6] != 63 || b[7] != 0) return -1;
        r->huff_map[scan][0] = (b[4] >> 4) & 15;
        r->huff_map[scan][1] = b[4] & 15;
        scan++;
    }
    return -1; /* no EOI */
}

/* the longest restart segment of the device's table: the maximum of the per-chunk values the marker scan left (valid once the stream has been
 * waited for) */
static void summary_take_maxlen(struct gpujpeg_decoder* d)
{
    uint32_t m = 0;
    for (uint32_t i = 0; i < d->maxlen_parts; i++)
        if (d->h_maxlen[i] > m) m = d->h_maxlen[i];
    d->h_summary->max_seg_len = m;
}

/* careful: this call must not use the entropy decoders that take whole restart segments into LDS (a previous attempt on this stream met a
 * segment that does not fit), and it does not launch on the header cache */
static int decoder_decode(struct gpujpeg_decoder* d, uint8_t* image, size_t image_size, struct gpujpeg_decoder_output* output, bool careful)
{
    struct gj_coder* c = &d->coder;
    GJ_HT_START(c);
    const bool stats = c->param.perf_stats != 0 || c->param.verbose >= GPUJPEG_LL_STATUS;
    c->start_time = stats ? gpujpeg_get_time() : 0;
    memset(&c->stats, 0, sizeof c->stats);
    const double t_read0 = stats ? gpujpeg_get_time() : 0;
    uint8_t* host_copy = NULL;
    int rc = -1;

    /* ---- 1. headers. A device-resident stream (MI355X extension) is parsed from a 64 KiB window. ---- */
    const bool jpeg_on_device = gj_hip_is_device_ptr(image) != 0;
    const uint8_t* himage = image; /* host view of (at least the headers of) the stream */
    size_t hsize = image_size;
    struct gj_reader_result r;
    bool device_scan = !d->host_scan;
    /* speculative path: same header as last time (compared on the device for a device-resident stream) */
    bool spec = device_scan && d->hdr_cache_valid && image_size > d->hdr_cache_len && !d->tune.dec_no_spec && !careful;
    if (spec && !jpeg_on_device) spec = memcmp(image, d->hdr_cache, d->hdr_cache_len) == 0;
    if (spec) {
        r = d->hdr_cache_r;
        rc = 0;
    } else {
    if (jpeg_on_device) {
        hsize = image_size < GJ_HDR_WINDOW ? image_size : GJ_HDR_WINDOW;
        if (gj_hip_memcpy_d2h(d->h_hdr, image, hsize, c->stream) != 0 || gj_hip_stream_sync(c->stream) != 0) return -1;
        himage = d->h_hdr;
    }
    rc = device_scan ? gj_reader_parse(himage, hsize, hsize < image_size ? GPUJPEG_LL_QUIET - 1 : c->param.verbose, d->ff_cs_itu601_is_709,
                                       d->req_pixel_format, d->req_color_space, d->req_alignment, &r, true)
                     : -1;
    if (rc != 0 || r.seg_info_count[0] > 0) device_scan = false; /* odd header, or an APP13 index that makes scanning unnecessary */
    if (!device_scan) {
        if (jpeg_on_device) { /* the host walk needs the whole stream */
            host_copy = malloc(image_size);
            if (!host_copy || gj_hip_memcpy_d2h(host_copy, image, image_size, c->stream) != 0 || gj_hip_stream_sync(c->stream) != 0) goto out;
            himage = host_copy;
            hsize = image_size;
        }
        rc = gj_reader_parse(himage, image_size, c->param.verbose, d->ff_cs_itu601_is_709, d->req_pixel_format, d->req_color_space,
                             d->req_alignment, &r, false);
        if (rc != 0) {
            GJ_ERROR("Decoder failed when decoding image data!\n");
            goto out;
        }
    }
    }
    rc = -1;
    for (int i = 0; i < GPUJPEG_METADATA_COUNT; i++) d->metadata.vals[i] = r.metadata.vals[i];
    if (decoder_configure(d, &r.param, &r.param_image) != 0) goto out;
    c->init_end_time = stats ? gpujpeg_get_time() : 0;
    gj_geom* g = &c->geom;

    /* ---- 2. stream to HBM ---- */
    if (stats || !jpeg_on_device) gj_hip_event_record(c->timers.copy_in[0], c->stream); /* (without perf_stats too: see gj_internal.h, copy markers) */
    const uint8_t* d_jpeg;
    if (jpeg_on_device) {
        d_jpeg = image;
    } else {
        if (gj_ensure_device_buffer((void**)&d->d_jpeg, &d->d_jpeg_cap, image_size + 64) != 0) goto out;
        if (gj_hip_memcpy_h2d(d->d_jpeg, image, image_size, c->stream) != 0) goto out; /* (on the coder's own stream, not in the upload lane behind other coders' images) */
        d_jpeg = d->d_jpeg;
    }

    /* ---- 3. segment table ---- */
    int seg_count = 0;
    const uint32_t* d_seg_count = NULL;
    gj_scan_summary* sum_cur = d->d_summary;
    gj_scan_deferred scan_deferred;
    memset(&scan_deferred, 0, sizeof scan_deferred);
    const size_t S = (size_t)g->segment_count + GJ_MAX_COMP;
    if (gj_ensure_device_buffer((void**)&d->d_seg, &d->d_seg_cap, (S * 4 + 8) * sizeof(uint32_t)) != 0) goto out;
    if (device_scan) {
        const size_t words = gj_hip_find_segments_scratch_words(r.scan_begin[0], image_size, (uint32_t)g->segment_count);
        if (gj_ensure_device_buffer((void**)&d->d_scan_scratch, &d->d_scan_scratch_cap, words * sizeof(uint32_t)) != 0) goto out;
        /* (a speculative launch on a device-resident stream has its header compared with the cached one by the scan's first kernel) */
        const bool cmp = spec && jpeg_on_device;
        /* the kernels write what the host validates straight into pinned host memory */
        /* (a word per scanning workgroup, or -- when the table launch is folded into the token decoder -- per batch of that kernel: at most a few thousand) */
        size_t max_chunks = gj_hip_find_segments_max_chunks(r.scan_begin[0], image_size);
        if (max_chunks < 4096) max_chunks = 4096;
        int frc = 0;
        if (max_chunks * sizeof(uint32_t) > d->h_maxlen_cap) {
            gj_hip_host_free(d->h_maxlen);
            d->h_maxlen_cap = max_chunks * sizeof(uint32_t) * 2;
            d->h_maxlen = gj_hip_host_alloc(d->h_maxlen_cap);
            if (!d->h_maxlen) { d->h_maxlen_cap = 0; frc = -1; }
        }
        d->h_summary->rst_irregular = 0;
        d->h_summary->seq_overflow = 0;
        d->h_summary->header_differs = 0;
        GJ_HT(c, 0);
        if (stats) gj_hip_event_record(c->timers.ev[4], c->stream); /* (the marker scan is GPU time of this call: events 4 and 5 bracket it) */
        /* speculative launches leave the scan's second launch to gj_hip_decode: the token decoder does without it (gj_scan_deferred) */
        if (frc == 0 && spec)
            frc = gj_hip_find_segments_deferred(g, d_jpeg, r.scan_begin[0], image_size, d->d_seg, d->d_seg + S, d->d_seg + 2 * S, (uint32_t)g->segment_count,
                                                d->d_scan_scratch, sum_cur, cmp ? d->d_hdr_cache : NULL, cmp ? (uint32_t)d->hdr_cache_len : 0u, d->h_summary,
                                                d->h_maxlen, (uint32_t)(d->h_maxlen_cap / sizeof(uint32_t)), &d->maxlen_parts, c->stream, &d->tune, &scan_deferred);
        else if (frc == 0)
            frc = gj_hip_find_segments(g, d_jpeg, r.scan_begin[0], image_size, d->d_seg, d->d_seg + S, d->d_seg + 2 * S, (uint32_t)g->segment_count,
                                       d->d_scan_scratch, sum_cur, cmp ? d->d_hdr_cache : NULL, cmp ? (uint32_t)d->hdr_cache_len : 0u, d->h_summary,
                                       d->h_maxlen, (uint32_t)(d->h_maxlen_cap / sizeof(uint32_t)), &d->maxlen_parts, c->stream, &d->tune);
        if (stats) gj_hip_event_record(c->timers.ev[5], c->stream);
        d->scan_timed = stats;
        if (frc != 0 || (!spec && gj_hip_stream_sync(c->stream) != 0)) {
            GJ_ERROR("Marker scan failed: %s\n", gj_hip_last_error());
            goto out;
        }
        if (!spec) summary_take_maxlen(d);
        if (spec) { /* the kernels take the segment count from the device; the summary is checked once everything has run */
            seg_count = g->segment_count;
            d_seg_count = &sum_cur->segment_count;
        } else if (accept_device_scan(d->h_summary, &r, g) == 0) {
            seg_count = (int)d->h_summary->segment_count;
            d_seg_count = NULL;
        } else {
            /* unusual stream (markers between scans, damaged data...): the host walk decides */
            GJ_DEBUG(c->param.verbose, "device marker scan not applicable (status %u), walking the stream on the host\n", d->h_summary->status);
            device_scan = false;
            if (jpeg_on_device && !host_copy) {
                host_copy = malloc(image_size);
                if (!host_copy || gj_hip_memcpy_d2h(host_copy, image, image_size, c->stream) != 0 || gj_hip_stream_sync(c->stream) != 0) goto out;
                himage = host_copy;
            } else if (!jpeg_on_device) {
                himage = image;
            }
            rc = gj_reader_parse(himage, image_size, c->param.verbose, d->ff_cs_itu601_is_709, d->req_pixel_format, d->req_color_space,
                                 d->req_alignment, &r, false);
            if (rc != 0) {
                GJ_ERROR("Decoder failed when decoding image data!\n");
                goto out;
            }
            rc = -1;
        }
    }
    if (!device_scan) {
        if (gj_reader_split_scans(himage, &r, g, &d->segs, c->param.verbose) != 0) goto out;
        if (d->segs.count > g->segment_count) {
            GJ_ERROR("Decoder can't decode image that has segment count %d (maximum segment count for specified parameters is %d)!\n", d->segs.count, g->segment_count);
            goto out;
        }
        seg_count = d->segs.count;
        const size_t ns = (size_t)seg_count;
        if (ns * 3 * sizeof(uint32_t) > d->h_seg_cap) {
            gj_hip_host_free(d->h_seg);
            d->h_seg_cap = ns * 3 * sizeof(uint32_t);
            d->h_seg = gj_hip_host_alloc(d->h_seg_cap);
            if (!d->h_seg) { d->h_seg_cap = 0; goto out; }
        }
        memcpy(d->h_seg, d->segs.pos, ns * sizeof(uint32_t));
        memcpy(d->h_seg + ns, d->segs.len, ns * sizeof(uint32_t));
        memcpy(d->h_seg + 2 * ns, d->segs.index, ns * sizeof(uint32_t));
        if (gj_hip_memcpy_h2d(d->d_seg, d->h_seg, ns * sizeof(uint32_t), c->stream) != 0 ||
            gj_hip_memcpy_h2d(d->d_seg + S, d->h_seg + ns, ns * sizeof(uint32_t), c->stream) != 0 ||
            gj_hip_memcpy_h2d(d->d_seg + 2 * S, d->h_seg + 2 * ns, ns * sizeof(uint32_t), c->stream) != 0)
            goto out;
    }
    if (seg_count != g->segment_count && c->param.verbose >= 0) GJ_WARN("%d segments read, expected %d. Broken JPEG?\n", seg_count, g->segment_count);
    c->stats.duration_stream = stats ? (gpujpeg_get_time() - t_read0) * 1000.0 : 0;

    /* ---- 4. tables: decode tables for every DHT slot present, natural-order quantisation tables ---- */
    for (int i = 0; i < g->comp_count; i++) {
        g->comp[i].q_table = r.quant_map[i];
        g->comp[i].dc_table = r.huff_map[i][0];
        g->comp[i].ac_table = r.huff_map[i][1];
    }
    bool tab2_ok = d->tab2_ok;
    if (!spec) {
    memset(d->h_tabs, 0, GJ_TABS_WORDS * sizeof(uint16_t));
    for (int th = 0; th < 4; th++)
        for (int tc = 0; tc < 2; tc++)
            if (r.h_present[th][tc] && gj_huffman_decoder_table(r.hbits[th][tc], r.hvals[th][tc], d->h_tabs + (th * 2 + tc) * GJ_DEC_TAB_WORDS) != 0) {
                GJ_ERROR("Invalid Huffman table %d/%d!\n", th, tc);
                goto out;
            }
    for (int t = 0; t < 4; t++)
        if (r.q_present[t]) {
            uint16_t* qi = d->h_tabs + 8 * GJ_DEC_TAB_WORDS + t * 64;
            float* qf = (float*)(void*)(d->h_tabs + GJ_TABS_QF_OFFSET) + t * 64;
            gj_quant_table_inverse(r.qraw[t], qi);
            for (int i = 0; i < 64; i++) qf[i] = (float)qi[i];
        }
    /* two-level tables of the sub-sequence decoder: slots 0 and 1 only, every table has to fit the layout */
    tab2_ok = true;
    for (int i = 0; i < g->comp_count; i++)
        if (g->comp[i].dc_table > 1 || g->comp[i].ac_table > 1) tab2_ok = false;
    for (int th = 0; th < 2 && tab2_ok; th++)
        for (int tc = 0; tc < 2; tc++)
            if (r.h_present[th][tc] && gj_huffman_decoder_table2(r.hbits[th][tc], r.hvals[th][tc], tc,
                                                                  d->h_tabs + 8 * GJ_DEC_TAB_WORDS + 4 * 64 + (th * 2 + tc) * GJ_DEC2_WORDS) != 0)
                tab2_ok = false;
    for (int i = 0; i < g->comp_count; i++) {
        if (g->comp[i].q_table > 3 || g->comp[i].dc_table > 3 || g->comp[i].ac_table > 3 || !r.q_present[g->comp[i].q_table] ||
            !r.h_present[g->comp[i].dc_table][0] || !r.h_present[g->comp[i].ac_table][1]) {
            GJ_ERROR("Component %d refers to a table that was not defined!\n", i);
            goto out;
        }
    }
    if (gj_hip_memcpy_h2d(d->d_huff_tab, d->h_tabs, GJ_TABS_WORDS * sizeof(uint16_t), c->stream) != 0) {
        GJ_ERROR("Decoder copy compressed data failed: %s\n", gj_hip_last_error());
        goto out;
    }
    d->tab2_ok = tab2_ok;
    }
    if (stats) gj_hip_event_record(c->timers.copy_in[1], c->stream);

    /* ---- 5. destination (:336-375) ---- */
    uint8_t* d_raw;
    if (output->type == GPUJPEG_DECODER_OUTPUT_CUSTOM_CUDA_BUFFER) {
        d_raw = output->data;
    } else if (output->type == GPUJPEG_DECODER_OUTPUT_OPENGL_TEXTURE) {
        GJ_ERROR("OpenGL texture output is not supported by the MI355X build.\n");
        goto out;
    } else {
        if (gj_ensure_device_buffer((void**)&c->d_raw_own, &c->d_raw_cap, g->raw_size) != 0) goto out;
        d_raw = c->d_raw_own;
    }

    gj_dec_job job;
    memset(&job, 0, sizeof job);
    job.g = *g;
    job.d_jpeg = d_jpeg;
    job.jpeg_size = image_size;
    job.d_seg_pos = d->d_seg;
    job.d_seg_len = d->d_seg + S;
    job.d_seg_index = d->d_seg + 2 * S;
    job.seg_count = seg_count;
    job.d_seg_count = d_seg_count;
    job.scan = scan_deferred;
    d->last_folded = 0;
    job.scan.folded = &d->last_folded;
    if (spec) d->n_spec++;
    job.d_huff_tab = d->d_huff_tab;
    job.d_qtab = d->d_qtab;
    job.d_qtabf = (const float*)(const void*)(d->d_huff_tab + GJ_TABS_QF_OFFSET);
    job.d_huff_tab2 = tab2_ok ? d->d_qtab + 4 * 64 : NULL;
    job.d_coefs = c->d_coefs;
    job.d_planes = c->d_planes;
    job.d_raw = d_raw;
    job.use_fused = d->use_fused && !d->flipped;
    job.flipped = d->flipped;
    job.channel_remap = d->channel_remap;
    if (d->channel_remap) {
        const enum gpujpeg_pixel_format pf = c->param_image.pixel_format;
        if ((int)(d->channel_remap >> 24) != gpujpeg_pixel_format_get_comp_count(pf)) {
            GJ_ERROR("Wrong channel remapping given, given %u channels but pixel format has %d!\n", d->channel_remap >> 24, gpujpeg_pixel_format_get_comp_count(pf));
            goto out;
        }
        if (pf != GPUJPEG_U8 && pf != GPUJPEG_444_U8_P012 && pf != GPUJPEG_4444_U8_P0123 && pf != GPUJPEG_444_U8_P0P1P2) {
            GJ_ERROR("Channel remapping is implemented for pixel formats whose pixels do not share samples (u8, 444-u8-p012, 4444-u8-p0123, 444-u8-p0p1p2).\n");
            goto out;
        }
    }
    /* the sub-sequence entropy decoder zero-fills every block it decodes; a full clear is needed only when segments are missing
     * from the table (their blocks would keep the previous frame's coefficients) -- unknown before the end in the speculative path,
     * which is validated and repeated then */
    job.clear_coefs = !spec && seg_count != g->segment_count;
    job.zero_coefs = 0;
    /* token mode buffers (gj_hip.h): one record per block, 4 tokens per stream byte at most */
    if (!d->keep_coefs && job.use_fused && tab2_ok && image_size < ((size_t)1 << 29) && gj_hip_decode_wants_tokens(&job.g, image_size, &d->tune)) {
        const size_t tok_need = (image_size * 4 + 64) * sizeof(uint16_t);
        if (tok_need > d->d_tok_cap) { /* (grown with headroom: frames of a sequence vary in size) */
            if (gj_ensure_device_buffer((void**)&d->d_tok, &d->d_tok_cap, tok_need + tok_need / 4) != 0) goto out;
        }
        if ((size_t)g->block_count * 8 + 64 > d->d_blkrec_cap) {
            if (gj_ensure_device_buffer((void**)&d->d_blkrec, &d->d_blkrec_cap, (size_t)g->block_count * 8 + 64) != 0) goto out;
            gj_hip_memset(d->d_blkrec, 0, d->d_blkrec_cap, c->stream);
        }
        job.tokens = 1;
        job.d_tok = d->d_tok;
        job.tok_cap = (uint32_t)(image_size * 4);
        job.d_blkrec = d->d_blkrec;
    }
    job.tune = d->tune;
    job.tune.dec_careful = careful || d->need_planes;
    /* bytes per scan: what the entropy decoder's batch sizes are cut to (luminance segments are 2-3 x the chrominance ones) */
    /* the word the entropy decoders raise when they meet a segment they cannot stage: in the host's (pinned, device-visible) summary */
    d->h_summary->seq_overflow = 0;
    job.d_overflow = &d->h_summary->seq_overflow;
    if (spec) {
        memcpy(job.scan_bytes, d->last_scan_bytes, sizeof job.scan_bytes);
        job.max_seg_len = d->last_max_seg_len;
    } else {
        if (device_scan) {
            job.max_seg_len = d->h_summary->max_seg_len;
        } else {
            job.max_seg_len = 0;
            for (int i = 0; i < d->segs.count; i++)
                if (d->segs.len[i] > job.max_seg_len) job.max_seg_len = d->segs.len[i];
        }
        for (int sc = 0; sc < GJ_MAX_COMP; sc++) {
            if (device_scan) job.scan_bytes[sc] = sc < (int)d->h_summary->scan_count ? d->h_summary->scan_end[sc] - d->h_summary->scan_start[sc] : 0;
            else job.scan_bytes[sc] = sc < r.scan_count && r.scan_end[sc] > r.scan_begin[sc] ? (uint32_t)(r.scan_end[sc] - r.scan_begin[sc]) : 0;
        }
        if (seg_count != g->segment_count) memset(job.scan_bytes, 0, sizeof job.scan_bytes); /* (table and geometry out of step: one batch size) */
    }
    if (gj_hip_decode(&job, c->stream, stats ? c->timers.ev : NULL) != 0) {
        GJ_ERROR("Decoder kernels failed: %s\n", gj_hip_last_error());
        goto out;
    }
    /* (the lane-per-segment entropy decoder may have met a segment it cannot stage: known once everything has run) */

    output->data_size = g->raw_size;
    output->param_image = c->param_image;
    if (output->param_image.color_space == GPUJPEG_NONE) output->param_image.color_space = c->param.color_space_internal;
    if (output->type == GPUJPEG_DECODER_OUTPUT_INTERNAL_BUFFER || output->type == GPUJPEG_DECODER_OUTPUT_CUSTOM_BUFFER) {
        uint8_t* dst;
        if (output->type == GPUJPEG_DECODER_OUTPUT_INTERNAL_BUFFER) {
            if (g->raw_size > d->h_raw_cap) {
                gj_hip_host_free(d->h_raw);
                d->h_raw = gj_hip_host_alloc(g->raw_size);
                d->h_raw_cap = d->h_raw ? g->raw_size : 0;
                if (!d->h_raw) goto out;
            }
            dst = d->h_raw;
            output->data = d->h_raw;
        } else {
            assert(output->data != NULL);
            dst = output->data;
        }
        gj_hip_event_record(c->timers.copy_out[0], c->stream);
        if (gj_hip_download(dst, d_raw, g->raw_size, c->stream, stats ? NULL : c->timers.lane_out) != 0) goto out; /* (the process's download lane) */
        if (stats) gj_hip_event_record(c->timers.copy_out[1], c->stream);
    } else {
        output->data = d_raw;
    }
    if (d->last_folded) d->n_folded++;
    GJ_HT(c, 1);
    if (gj_hip_stream_sync(c->stream) != 0) {
        GJ_ERROR("Decoder failed: %s\n", gj_hip_last_error());
        goto out;
    }
    GJ_HT(c, 2);

    if (!spec && d->h_summary->seq_overflow && !careful) { /* (a kernel forced on a stream with segments it cannot stage; or, after a host
                                                               walk, a table whose longest segment the host did not foresee) */
        GJ_DEBUG(c->param.verbose, "a restart segment did not fit the entropy decoder's LDS stage: decoding again through the coefficient planes\n");
        free(host_copy);
        d->need_planes = true; /* (and so for the frames that follow with this header) */
        return decoder_decode(d, image, image_size, output, true);
    }
    if (spec) { /* now the summary of this stream is on the host: was it what we assumed? */
        summary_take_maxlen(d);
        struct gj_reader_result chk = r;
        bool ok = d->h_summary->header_differs == 0 && d->h_summary->seq_overflow == 0 && accept_device_scan(d->h_summary, &chk, g) == 0 &&
                  (int)d->h_summary->segment_count == g->segment_count; /* (a stream with missing segments needs the planes cleared first) */
        for (int i = 0; ok && i < g->comp_count; i++)
            if (chk.huff_map[i][0] != r.huff_map[i][0] || chk.huff_map[i][1] != r.huff_map[i][1]) ok = false;
        if (!ok) { /* different header or unusual scan structure: decode again the careful way */
            d->n_again++;
            if (d->last_folded) d->n_folded--; /* (counted at the launch: only accepted launches count as folded) */
            const bool overflow_only = d->h_summary->seq_overflow != 0 && d->h_summary->header_differs == 0;
            if (overflow_only) d->need_planes = true; /* (the header was the assumed one: it stays cached, the next frames launch on it with the other kernels) */
            else d->hdr_cache_valid = false;
            free(host_copy);
            return decoder_decode(d, image, image_size, output, overflow_only);
        }
        for (int sc = 0; sc < GJ_MAX_COMP; sc++)
            d->last_scan_bytes[sc] = sc < (int)d->h_summary->scan_count ? d->h_summary->scan_end[sc] - d->h_summary->scan_start[sc] : 0;
        d->last_max_seg_len = d->h_summary->max_seg_len;
        if ((int)d->h_summary->segment_count != g->segment_count && c->param.verbose >= 0)
            GJ_WARN("%d segments read, expected %d. Broken JPEG?\n", (int)d->h_summary->segment_count, g->segment_count);
    } else if (device_scan && r.scan_begin[0] <= GJ_HDR_WINDOW && r.scan_begin[0] < image_size) {
        /* remember this header (SOI .. first SOS header) for the next call */
        const size_t n = r.scan_begin[0];
        if (!d->hdr_cache) d->hdr_cache = malloc(GJ_HDR_WINDOW);
        if (!d->d_hdr_cache) d->d_hdr_cache = gj_hip_malloc(GJ_HDR_WINDOW);
        if (d->hdr_cache && d->d_hdr_cache) {
            /* another header than the cached one: what was learnt about the old one's frames does not apply (unless this very call is the
             * second attempt that learnt it) */
            if (!careful && !(d->hdr_cache_len == n && memcmp(d->hdr_cache, himage, n) == 0)) d->need_planes = false;
            memcpy(d->hdr_cache, himage, n);
            if (gj_hip_memcpy_h2d(d->d_hdr_cache, d->hdr_cache, n, c->stream) == 0 && gj_hip_stream_sync(c->stream) == 0) {
                d->hdr_cache_len = n;
                d->hdr_cache_r = r;
                d->hdr_cache_r.comment = NULL;
                d->hdr_cache_valid = true;
                d->last_max_seg_len = d->h_summary->max_seg_len;
                for (int sc = 0; sc < GJ_MAX_COMP; sc++)
                    d->last_scan_bytes[sc] = sc < (int)d->h_summary->scan_count ? d->h_summary->scan_end[sc] - d->h_summary->scan_start[sc] : 0;
            }
        }
    } else {
        d->hdr_cache_valid = false;
    }
    if (stats) {
        struct gpujpeg_duration_stats* s = &c->stats;
        s->duration_huffman_coder = gj_hip_event_elapsed_ms(c->timers.ev[0], c->timers.ev[1]);
        s->duration_dct_quantization = gj_hip_event_elapsed_ms(c->timers.ev[1], c->timers.ev[2]);
        s->duration_preprocessor = gj_hip_event_elapsed_ms(c->timers.ev[2], c->timers.ev[3]);
        s->duration_in_gpu = gj_hip_event_elapsed_ms(c->timers.ev[0], c->timers.ev[3]);
        for (int k = 0; k < GJ_DEC_EVENTS - 1; k++) c->kernel_ms[k] = gj_hip_event_elapsed_ms(c->timers.ev[k], c->timers.ev[k + 1]);
        c->kernel_ms[3] = d->scan_timed ? gj_hip_event_elapsed_ms(c->timers.ev[4], c->timers.ev[5]) : 0.0f; /* the marker scan, when the device did it */
        s->duration_in_gpu += c->kernel_ms[3];
        c->timers.valid = true;
        s->duration_memory_to = gj_hip_event_elapsed_ms(c->timers.copy_in[0], c->timers.copy_in[1]);
        if (output->type == GPUJPEG_DECODER_OUTPUT_INTERNAL_BUFFER || output->type == GPUJPEG_DECODER_OUTPUT_CUSTOM_BUFFER)
            s->duration_memory_from = gj_hip_event_elapsed_ms(c->timers.copy_out[0], c->timers.copy_out[1]);
    }
    gj_coder_process_stats(c, stats);
    if (c->param.verbose >= GPUJPEG_LL_STATUS)
        fprintf(stderr, "Decompressed Size:%13zu bytes %dx%d %s %s\n", output->data_size, output->param_image.width, output->param_image.height,
                gpujpeg_pixel_format_get_name(output->param_image.pixel_format), gpujpeg_color_space_get_name(output->param_image.color_space));
    output->metadata = &d->metadata;
    rc = 0;
    GJ_HT(c, 3);
    c->ht_calls++;
out:
    free(host_copy);
    return rc;
}

int gpujpeg_decoder_decode(struct gpujpeg_decoder* d, uint8_t* image, size_t image_size, struct gpujpeg_decoder_output* output)
{
    return decoder_decode(d, image, image_size, output, false);
}

int gpujpeg_decoder_get_stats(struct gpujpeg_decoder* d, struct gpujpeg_duration_stats* stats)
{
    if (!d || !stats) return -1;
    *stats = d->coder.stats;
    return 0;
}


/* ------------------------------------------------------------------ frame batches (MI355X extension, include/gpujpeg_amd_ext.h) */
/* Streams with ONE header (a sequence from one encoder) decoded by one set of launches per chunk of frames -- marker scan, marker table, entropy
 * decoder, IDCT, each with blockIdx.z = frame. It is the speculative path of decoder_decode (kernels launched on the cached header, the device
 * compares every stream's header with it, the host validates every frame's summary afterwards) for many frames at once; a frame whose
 * summary does not pass -- another header, an unusual scan structure, a segment the fast kernels cannot stage -- is decoded again by
 * gpujpeg_decoder_decode's own path, and so is everything the batched kernels do not cover. */
#define GJ_DEC_BATCH_CHUNK_MAX 256
#define GJ_DEC_BATCH_BYTES ((size_t)8 << 30) /* work buffers of one chunk */

static int pinned_ensure(void** p, size_t* cap, size_t need)
{
    if (need <= *cap) return 0;
    gj_hip_host_free(*p);
    *p = gj_hip_host_alloc(need + need / 2);
    *cap = *p ? need + need / 2 : 0;
    return *p ? 0 : -1;
}

/* one frame of a batch the ordinary way: into the decoder's own buffer first -- a stream that is not what the caller promised (other dimensions) must
 * not be written over the neighbours' slots --, then into its slot; *frame_raw = the size every frame of the batch decodes to (0: not known yet) */
static int batch_decode_one(struct gpujpeg_decoder* d, const uint8_t* stream, size_t size, uint8_t* d_out, size_t out_stride, size_t* frame_raw)
{
    struct gpujpeg_decoder_output o;
    gpujpeg_decoder_output_set_cuda_buffer(&o);
    if (decoder_decode(d, (uint8_t*)(uintptr_t)stream, size, &o, false) != 0) return -1;
    const size_t raw = d->coder.geom.raw_size;
    if (*frame_raw == 0) *frame_raw = raw;
    if (raw != *frame_raw || raw > out_stride) {
        GJ_ERROR("A frame of the batch decodes to %zu B, the others to %zu B (output stride %zu)!\n", raw, *frame_raw, out_stride);
        return -1;
    }
    if (gj_hip_memcpy_d2d(d_out, o.data, raw, d->coder.stream) != 0 || gj_hip_stream_sync(d->coder.stream) != 0) return -1;
    return 0;
}

int gpujpeg_amd_decoder_decode_batch(struct gpujpeg_decoder* d, const uint8_t* streams, size_t stream_stride, const size_t* sizes, int count,
                                     uint8_t* output, size_t output_stride, struct gpujpeg_image_parameters* param_image)
{
    if (!d || !streams || !sizes || count < 1 || !output) return -1;
    struct gj_coder* c = &d->coder;
    const bool streams_on_device = gj_hip_is_device_ptr(streams) != 0, out_on_device = gj_hip_is_device_ptr(output) != 0;
    int rc = -1;
    uint8_t* frame_done = calloc((size_t)count, 1);
    if (!frame_done) return -1;
    /* where the pixels go on the device: the caller's buffer, or a staging area that is copied out at the end */
    uint8_t* d_out = output;
    size_t d_out_stride = output_stride;
    int first = 0;
    /* the header cache is what a batch launches on: the first frame goes the ordinary way when there is none (or when it has to) */
    const uint8_t* s0 = streams;
    size_t frame_raw = 0; /* what every frame decodes to */
    bool force_first = false; /* the cached header turned out to be another sequence's: frame 0 the ordinary way (which replaces the cache), then the batch */
again:
    if (first == 0 && (!out_on_device || !d->hdr_cache_valid || force_first)) {
        /* (the output size is known once a frame has been parsed: decode frame 0 into the decoder's own buffer first) */
        struct gpujpeg_decoder_output o;
        gpujpeg_decoder_output_set_cuda_buffer(&o);
        if (decoder_decode(d, (uint8_t*)(uintptr_t)s0, sizes[0], &o, false) != 0) goto out;
        const size_t raw = c->geom.raw_size;
        frame_raw = raw;
        if (output_stride < raw) {
            GJ_ERROR("Output stride %zu is smaller than a decoded frame (%zu B)!\n", output_stride, raw);
            goto out;
        }
        if (!out_on_device) {
            if (gj_ensure_device_buffer((void**)&d->b_raw, &d->b_raw_cap, raw * (size_t)count) != 0) goto out;
            d_out = d->b_raw;
            d_out_stride = raw;
        }
        if (gj_hip_memcpy_d2d(d_out, o.data, raw, c->stream) != 0 || gj_hip_stream_sync(c->stream) != 0) goto out;
        frame_done[0] = 1;
        first = 1;
    }
    const gj_geom* g = &c->geom;
    bool batched = d->hdr_cache_valid && !d->need_planes && !d->host_scan && !d->tune.dec_no_spec && !d->keep_coefs && c->configured &&
                   (streams_on_device ? (stream_stride & 15u) == 0 : true) && d->tab2_ok;
    size_t max_size = 0;
    for (int f = first; f < count; f++) {
        if (sizes[f] > max_size) max_size = sizes[f];
        if (sizes[f] <= d->hdr_cache_len + 2 || sizes[f] > 0x1FFFFFF0u) batched = false;
    }
    gj_dec_job job;
    memset(&job, 0, sizeof job);
    if (batched && first < count) {
        struct gj_reader_result r = d->hdr_cache_r;
        if (decoder_configure(d, &r.param, &r.param_image) != 0) goto out; /* (the geometry of the cached header, whatever the coder was last set up for) */
        if (output_stride < g->raw_size || (frame_raw != 0 && frame_raw != g->raw_size)) {
            /* the cached header is an OLDER sequence's (a larger image than the caller's slots, ADVICE r4: this used to fail the call): these
             * streams are strangers to it. Frame 0 goes the ordinary way, which replaces the cache, and the batch is planned again */
            if (first == 0 && !force_first) {
                force_first = true;
                frame_raw = 0;
                goto again;
            }
            batched = false;
        }
    }
    if (batched && first < count) {
        struct gj_reader_result r = d->hdr_cache_r;
        frame_raw = g->raw_size;
        for (int i = 0; i < c->geom.comp_count; i++) {
            c->geom.comp[i].q_table = r.quant_map[i];
            c->geom.comp[i].dc_table = r.huff_map[i][0];
            c->geom.comp[i].ac_table = r.huff_map[i][1];
        }
        job.g = *g;
        job.seg_count = g->segment_count;
        job.d_huff_tab = d->d_huff_tab;
        job.d_qtab = d->d_qtab;
        job.d_qtabf = (const float*)(const void*)(d->d_huff_tab + GJ_TABS_QF_OFFSET);
        job.d_huff_tab2 = d->d_qtab + 4 * 64;
        job.d_planes = c->d_planes;
        job.use_fused = d->use_fused && !d->flipped;
        job.flipped = d->flipped;
        job.channel_remap = d->channel_remap;
        job.tune = d->tune;
        job.tune.dec_careful = 0;
        job.max_seg_len = d->last_max_seg_len;
        memcpy(job.scan_bytes, d->last_scan_bytes, sizeof job.scan_bytes);
        job.jpeg_size = max_size;
        /* (the two pointers only have to be non-null for the question; they are set per chunk below) */
        job.d_seg_count = &d->d_summary->segment_count;
        job.d_overflow = &d->h_summary->seq_overflow;
        if (!gj_hip_decode_batchable(&job)) batched = false;
    }
    if (batched && first < count) {
        const int n_all = count - first;
        const size_t begin = d->hdr_cache_r.scan_begin[0];
        /* streams in host memory: one staging area, slots on 16-byte addresses */
        const uint8_t* d_streams = streams + (size_t)first * stream_stride;
        size_t d_stride = stream_stride;
        if (!streams_on_device) {
            d_stride = (max_size + 64 + 15) & ~(size_t)15;
            if (gj_ensure_device_buffer((void**)&d->b_jpeg, &d->b_jpeg_cap, d_stride * (size_t)n_all) != 0) goto out;
            gj_hip_event_record(c->timers.copy_in[0], c->stream); /* (copy marker, see gj_internal.h) */
            for (int f = first; f < count; f++)
                if (gj_hip_memcpy_h2d(d->b_jpeg + (size_t)(f - first) * d_stride, streams + (size_t)f * stream_stride, sizes[f], c->stream) != 0) goto out;
            d_streams = d->b_jpeg;
        }
        /* per-frame words of all frames */
        const size_t max_chunks = gj_hip_find_segments_max_chunks(begin, max_size);
        if (gj_ensure_device_buffer((void**)&d->b_sizes, &d->b_sizes_cap, (size_t)n_all * sizeof(uint32_t)) != 0) goto out;
        if (pinned_ensure((void**)&d->bh_sizes, &d->bh_sizes_cap, (size_t)n_all * sizeof(uint32_t)) != 0) goto out;
        if (pinned_ensure((void**)&d->bh_sum, &d->bh_sum_cap, (size_t)n_all * sizeof(gj_scan_summary)) != 0) goto out;
        if (pinned_ensure((void**)&d->bh_maxlen, &d->bh_maxlen_cap, (size_t)n_all * max_chunks * sizeof(uint32_t)) != 0) goto out;
        {
            const size_t had = d->b_dsum_cap;
            if (gj_ensure_device_buffer((void**)&d->b_dsum, &d->b_dsum_cap, (size_t)n_all * sizeof(gj_scan_summary)) != 0) goto out;
            if (d->b_dsum_cap != had && gj_hip_memset(d->b_dsum, 0, d->b_dsum_cap, c->stream) != 0) goto out;
        }
        memset(d->bh_sum, 0, (size_t)n_all * sizeof(gj_scan_summary));
        for (int i = 0; i < n_all; i++) d->bh_sizes[i] = (uint32_t)sizes[first + i];
        if (gj_hip_memcpy_h2d(d->b_sizes, d->bh_sizes, (size_t)n_all * sizeof(uint32_t), c->stream) != 0) goto out;
        /* work buffers of a chunk */
        const size_t S = (size_t)g->segment_count + GJ_MAX_COMP;
        const size_t seg_frame = (3 * S + 8 + 3) & ~(size_t)3;                                                                   /* words */
        const size_t scratch_frame = (gj_hip_find_segments_scratch_words(begin, max_size, (uint32_t)g->segment_count) + 3) & ~(size_t)3; /* words */
        const size_t coefs_frame = ((size_t)g->data_size + 63) & ~(size_t)63;                                                   /* int16 */
        const int chunk_cap = d->b_chunk > 0 && d->b_chunk < GJ_DEC_BATCH_CHUNK_MAX ? d->b_chunk : GJ_DEC_BATCH_CHUNK_MAX;
        job.g.fb.frames = (uint32_t)(n_all < chunk_cap ? n_all : chunk_cap); /* (what the token / plane choice looks at) */
        const bool tokens = gj_hip_decode_wants_tokens(&job.g, max_size, &d->tune) != 0;
        const size_t tok_frame = tokens ? (max_size * 4 + 64 + 63) & ~(size_t)63 : 0;                                            /* tokens */
        const size_t rec_frame = tokens ? ((size_t)g->block_count + 8 + 7) & ~(size_t)7 : 0;                                    /* records */
        const bool planes = gj_hip_decode_uses_planes(g, job.use_fused) != 0;
        const size_t frame_bytes = seg_frame * 4 + scratch_frame * 4 + coefs_frame * 2 + tok_frame * 2 + rec_frame * 8 + (planes ? coefs_frame : 0);
        int chunk = (int)(GJ_DEC_BATCH_BYTES / frame_bytes);
        if (chunk > chunk_cap) chunk = chunk_cap;
        if (chunk > n_all) chunk = n_all;
        if (chunk < 1) chunk = 1;
        if (gj_ensure_device_buffer((void**)&d->b_seg, &d->b_seg_cap, seg_frame * 4 * (size_t)chunk) != 0) goto out;
        if (gj_ensure_device_buffer((void**)&d->b_scratch, &d->b_scratch_cap, scratch_frame * 4 * (size_t)chunk + 64) != 0) goto out;
        if (gj_ensure_device_buffer((void**)&d->b_coefs, &d->b_coefs_cap, coefs_frame * 2 * (size_t)chunk) != 0) goto out;
        if (planes && gj_ensure_device_buffer((void**)&d->b_planes, &d->b_planes_cap, coefs_frame * (size_t)chunk) != 0) goto out;
        if (planes) job.d_planes = d->b_planes;
        if (tokens) {
            if (gj_ensure_device_buffer((void**)&d->b_tok, &d->b_tok_cap, tok_frame * 2 * (size_t)chunk) != 0) goto out;
            const size_t had = d->b_rec_cap;
            if (gj_ensure_device_buffer((void**)&d->b_rec, &d->b_rec_cap, rec_frame * 8 * (size_t)chunk) != 0) goto out;
            if (d->b_rec_cap != had && gj_hip_memset(d->b_rec, 0, d->b_rec_cap, c->stream) != 0) goto out;
        }
        job.d_seg_pos = d->b_seg;
        job.d_seg_len = d->b_seg + S;
        job.d_seg_index = d->b_seg + 2 * S;
        job.d_coefs = d->b_coefs;
        job.tokens = tokens ? 1 : 0;
        job.d_tok = tokens ? d->b_tok : NULL;
        job.tok_cap = tokens ? (uint32_t)(max_size * 4) : 0;
        job.d_blkrec = tokens ? d->b_rec : NULL;
        for (int a = 0; a < n_all; a += chunk) {
            const int n = n_all - a < chunk ? n_all - a : chunk;
            gj_batch B;
            memset(&B, 0, sizeof B);
            B.count = (uint32_t)n;
            B.jpeg = d_stride;
            B.raw = d_out_stride;
            B.coefs = coefs_frame;
            B.tok = tok_frame;
            B.rec = rec_frame;
            B.seg = (uint32_t)seg_frame;
            B.scratch = (uint32_t)scratch_frame;
            B.maxlen = (uint32_t)max_chunks;
            B.d_sizes = d->b_sizes + a;
            job.batch = B;
            job.g.fb.sizes = B.d_sizes;
            job.g.fb.jpeg = B.jpeg;
            job.g.fb.raw = B.raw;
            job.g.fb.coefs = B.coefs;
            job.g.fb.tok = B.tok;
            job.g.fb.rec = B.rec;
            job.g.fb.seg = B.seg;
            job.d_jpeg = d_streams + (size_t)a * d_stride;
            job.d_raw = d_out + (size_t)(first + a) * d_out_stride;
            job.d_seg_count = &d->b_dsum[a].segment_count;
            job.d_overflow = &d->bh_sum[a].seq_overflow;
            uint32_t parts = 0;
            if (gj_hip_find_segments_batch(g, job.d_jpeg, begin, max_size, d->b_seg, d->b_seg + S, d->b_seg + 2 * S, (uint32_t)g->segment_count, d->b_scratch,
                                           d->b_dsum + a, d->d_hdr_cache, (uint32_t)d->hdr_cache_len, d->bh_sum + a, d->bh_maxlen + (size_t)a * max_chunks,
                                           (uint32_t)max_chunks, &parts, c->stream, &d->tune, &B) != 0 ||
                gj_hip_decode(&job, c->stream, NULL) != 0) {
                GJ_ERROR("Decoder kernels failed: %s\n", gj_hip_last_error());
                goto out;
            }
            d->maxlen_parts = parts;
        }
        if (gj_hip_stream_sync(c->stream) != 0) {
            GJ_ERROR("Decoder failed: %s\n", gj_hip_last_error());
            goto out;
        }
        /* every frame's summary: was the stream what the launch assumed? (the checks of decoder_decode's speculative path) */
        for (int i = 0; i < n_all; i++) {
            gj_scan_summary* su = d->bh_sum + i;
            uint32_t m = 0;
            for (uint32_t k = 0; k < d->maxlen_parts; k++)
                if (d->bh_maxlen[(size_t)i * max_chunks + k] > m) m = d->bh_maxlen[(size_t)i * max_chunks + k];
            su->max_seg_len = m;
            struct gj_reader_result chk = d->hdr_cache_r;
            bool ok = su->header_differs == 0 && su->seq_overflow == 0 && accept_device_scan(su, &chk, g) == 0 && (int)su->segment_count == g->segment_count;
            for (int k = 0; ok && k < g->comp_count; k++)
                if (chk.huff_map[k][0] != d->hdr_cache_r.huff_map[k][0] || chk.huff_map[k][1] != d->hdr_cache_r.huff_map[k][1]) ok = false;
            if (ok) {
                frame_done[first + i] = 1;
                for (int sc = 0; sc < GJ_MAX_COMP; sc++) d->last_scan_bytes[sc] = sc < (int)su->scan_count ? su->scan_end[sc] - su->scan_start[sc] : 0;
                d->last_max_seg_len = su->max_seg_len;
            }
        }
    }
    if (batched && first == 0 && !force_first) {
        /* not one frame passed: the cached header (an older sequence's, of dimensions the slots happen to hold) is not these streams' */
        int accepted = 0;
        for (int f = 0; f < count; f++) accepted += frame_done[f];
        if (accepted == 0) {
            force_first = true;
            frame_raw = 0;
            goto again;
        }
    }
    /* whatever is left: the ordinary call, frame by frame */
    d->b_last_batched = 0;
    for (int f = first; f < count; f++) d->b_last_batched += frame_done[f] ? 1 : 0;
    d->b_last_single = count - d->b_last_batched;
    for (int f = 0; f < count; f++) {
        if (frame_done[f]) continue;
        if (batch_decode_one(d, streams + (size_t)f * stream_stride, sizes[f], d_out + (size_t)f * d_out_stride, d_out_stride, &frame_raw) != 0) goto out;
    }
    if (!out_on_device) {
        gj_hip_event_record(c->timers.copy_out[0], c->stream); /* (copy marker) */
        const gj_stream_t down = gj_hip_lane_begin(1, frame_raw, c->stream); /* (the process's download lane for frames of 1 MiB and more) */
        int copies = 0;
        for (int f = 0; f < count && copies == 0; f++)
            copies = gj_hip_memcpy_d2h(output + (size_t)f * output_stride, d_out + (size_t)f * d_out_stride, frame_raw, down);
        /* (a copy that could not be queued: the ones before it are still on their way INTO THE CALLER'S BUFFERS -- wait for the lane before the error leaves) */
        if (gj_hip_lane_end(down, c->stream, c->timers.lane_out) != 0 || copies != 0 || gj_hip_stream_sync(c->stream) != 0) goto out;
    }
    if (param_image) {
        *param_image = c->param_image;
        if (param_image->color_space == GPUJPEG_NONE) param_image->color_space = c->param.color_space_internal;
    }
    rc = 0; /* (c->frames counts the frames TIMED with perf_stats, src/gpujpeg_common.c:2238-2254: a batch adds none) */
out:
    free(frame_done);
    return rc;
}

/* The same for streams and destinations that are separate buffers (device or host memory). The streams are gathered 16 bytes aligned in a staging
 * buffer (they are small), the frames are decoded back to back and copied to their destinations; `frame_bytes` = room in every destination. */
int gpujpeg_amd_decoder_decode_batch_ptrs(struct gpujpeg_decoder* d, const uint8_t* const* streams, const size_t* sizes, int count, uint8_t* const* outputs,
                                          size_t frame_bytes, struct gpujpeg_image_parameters* param_image)
{
    if (!d || !streams || !sizes || !outputs || count < 1) return -1;
    size_t longest = 0;
    for (int f = 0; f < count; f++) {
        if (!streams[f] || !outputs[f]) return -1;
        if (sizes[f] > longest) longest = sizes[f];
    }
    struct gj_coder* c = &d->coder;
    if (frame_bytes == 0) return -1;
    /* buffers of one kind a constant distance apart (streams: a multiple of 16 bytes in device memory): used where they are */
    bool in_strided = true, out_strided = true;
    /* (addresses compared as integers: the buffers may be unrelated allocations, whose pointers C does not let us subtract) */
#define GJ_ADDR_STEP(a, b) ((ptrdiff_t)((uintptr_t)(a) - (uintptr_t)(b)))
    const ptrdiff_t in_step = count > 1 ? GJ_ADDR_STEP(streams[1], streams[0]) : (ptrdiff_t)((longest + 64 + 15) & ~(size_t)15);
    const ptrdiff_t out_step = count > 1 ? GJ_ADDR_STEP(outputs[1], outputs[0]) : (ptrdiff_t)frame_bytes;
    const int in_dev = gj_hip_is_device_ptr(streams[0]), out_dev = gj_hip_is_device_ptr(outputs[0]);
    for (int f = 1; f < count; f++) {
        in_strided = in_strided && GJ_ADDR_STEP(streams[f], streams[f - 1]) == in_step && gj_hip_is_device_ptr(streams[f]) == in_dev;
        out_strided = out_strided && GJ_ADDR_STEP(outputs[f], outputs[f - 1]) == out_step && gj_hip_is_device_ptr(outputs[f]) == out_dev;
    }
#undef GJ_ADDR_STEP
    in_strided = in_strided && in_step > 0 && (size_t)in_step >= longest && (!in_dev || (in_step & 15) == 0);
    out_strided = out_strided && out_step >= (ptrdiff_t)frame_bytes;
    const uint8_t* src = streams[0];
    size_t in_stride = (size_t)in_step;
    if (!in_strided) {
        in_stride = (longest + 64 + 15) & ~(size_t)15;
        if (gj_ensure_device_buffer((void**)&d->b_gather, &d->b_gather_cap, in_stride * (size_t)count) != 0) return -1;
        for (int f = 0; f < count; f++) {
            const int rc = gj_hip_is_device_ptr(streams[f]) ? gj_hip_memcpy_d2d(d->b_gather + (size_t)f * in_stride, streams[f], sizes[f], c->stream)
                                                            : gj_hip_memcpy_h2d(d->b_gather + (size_t)f * in_stride, streams[f], sizes[f], c->stream);
            if (rc != 0) return -1;
        }
        if (gj_hip_stream_sync(c->stream) != 0) return -1; /* (the first frame may be parsed on the host from the staged copy) */
        src = d->b_gather;
    }
    struct gpujpeg_image_parameters pi;
    if (out_strided) { /* (device or host memory: decode_batch takes both) */
        if (gpujpeg_amd_decoder_decode_batch(d, src, in_stride, sizes, count, outputs[0], (size_t)out_step, &pi) != 0) return -1;
        if (param_image) *param_image = pi;
        return 0;
    }
    if (gj_ensure_device_buffer((void**)&d->b_scatter, &d->b_scatter_cap, frame_bytes * (size_t)count) != 0) return -1;
    if (gpujpeg_amd_decoder_decode_batch(d, src, in_stride, sizes, count, d->b_scatter, frame_bytes, &pi) != 0) return -1;
    const size_t raw = c->geom.raw_size;
    for (int f = 0; f < count; f++) {
        const int rc = gj_hip_is_device_ptr(outputs[f]) ? gj_hip_memcpy_d2d(outputs[f], d->b_scatter + (size_t)f * frame_bytes, raw, c->stream)
                                                        : gj_hip_memcpy_d2h(outputs[f], d->b_scatter + (size_t)f * frame_bytes, raw, c->stream);
        if (rc != 0) return -1;
    }
    if (gj_hip_stream_sync(c->stream) != 0) return -1;
    if (param_image) *param_image = pi;
    return 0;
}

/* ------------------------------------------------------------------ image info (src/gpujpeg_reader.c:1739-1872) */
int gpujpeg_decoder_get_image_info2(uint8_t* image, size_t image_size, struct gpujpeg_image_info* info, int verbose, unsigned flags)
{
    struct gj_reader_result r;
    const bool count_segments = (flags & GPUJPEG_COUNT_SEG_COUNT_REQ) != 0;
    const int rc = gj_reader_parse(image, image_size, verbose, false, GPUJPEG_PIXFMT_NATIVE, GPUJPEG_NONE, 0, &r, !count_segments);
    if (rc != 0) return rc;
    memset(info, 0, sizeof *info);
    info->param_image = r.param_image;
    info->param = r.param;
    info->header_type = r.header_type;
    info->comment = r.comment;
    info->metadata = r.metadata;
    if (info->param_image.color_space == GPUJPEG_NONE) info->param_image.color_space = r.param.color_space_internal;
    if (count_segments) {
        gj_geom g;
        struct gj_host_segments segs = {0};
        if (gj_geom_init(&g, &r.param, &r.param_image, false) == 0 && gj_reader_split_scans(image, &r, &g, &segs, verbose) == 0)
            info->segment_count = segs.count;
        free(segs.pos); free(segs.len); free(segs.index);
    }
    return 0;
}

int gpujpeg_decoder_get_image_info(uint8_t* image, size_t image_size, struct gpujpeg_image_parameters* pi, struct gpujpeg_parameters* param, int* segment_count)
{
    struct gpujpeg_image_info info;
    const int rc = gpujpeg_decoder_get_image_info2(image, image_size, &info, param ? param->verbose : 0, segment_count ? GPUJPEG_COUNT_SEG_COUNT_REQ : 0);
    if (rc != 0) return rc;
    if (pi) *pi = info.param_image;
    if (param) { const int v = param->verbose, ps = param->perf_stats; *param = info.param; param->verbose = v; param->perf_stats = ps; }
    if (segment_count) *segment_count = info.segment_count;
    return 0;
}

/* ------------------------------------------------------------------ options (src/gpujpeg_decoder.c:486-558) */
int gpujpeg_decoder_set_option(struct gpujpeg_decoder* d, const char* opt, const char* val)
{
    if (d == NULL || opt == NULL || val == NULL) return GPUJPEG_ERROR;
    if (strcmp(opt, GPUJPEG_DEC_OPT_ALIGNMENT_BYTES_INT) == 0) {
        const int a = atoi(val);
        if (a < 0) { GJ_ERROR("Wrong alignment: %s\n", val); return GPUJPEG_ERROR; }
        d->req_alignment = (unsigned)a;
        d->hdr_cache_valid = false;
        return GPUJPEG_NOERR;
    }
    if (strcmp(opt, GPUJPEG_DEC_OPT_TGA_RLE_BOOL) == 0) return GPUJPEG_NOERR; /* only affects file output */
    if (strcmp(opt, GPUJPEG_DEC_OPT_FLIPPED_BOOL) == 0) { /* src/gpujpeg_decoder.c:499-501 */
        if (strcasecmp(val, GPUJPEG_VAL_TRUE) == 0 || strcmp(val, "1") == 0) d->flipped = true;
        else if (strcasecmp(val, GPUJPEG_VAL_FALSE) == 0 || strcmp(val, "0") == 0) d->flipped = false;
        else { GJ_ERROR("Unknown option %s for " GPUJPEG_DEC_OPT_FLIPPED_BOOL "\n", val); return GPUJPEG_ERROR; }
        return GPUJPEG_NOERR;
    }
    if (strcmp(opt, GPUJPEG_DEC_OPT_CHANNEL_REMAP) == 0) return gj_parse_channel_remap(&d->channel_remap, val, opt);
    GJ_ERROR("Invalid decoder option: %s!\n", opt);
    return GPUJPEG_ERROR;
}

void gpujpeg_decoder_print_options(void)
{
    printf("\t" GPUJPEG_DEC_OPT_ALIGNMENT_BYTES_INT "=<n> - required line alignment of the decoded image in bytes\n");
    printf("\t" GPUJPEG_DEC_OPT_FLIPPED_BOOL "=[" GPUJPEG_VAL_TRUE "|" GPUJPEG_VAL_FALSE "] - whether the output image should be vertically flipped\n");
    printf("\t" GPUJPEG_DEC_OPT_CHANNEL_REMAP "=XYZ[W] - output channel remapping, 'help' for details\n");
}

/* ------------------------------------------------------------------ MI355X extensions (include/gpujpeg_amd_ext.h) */

void gpujpeg_amd_decoder_set_batch_chunk(struct gpujpeg_decoder* d, int frames) { if (d) d->b_chunk = frames > 0 ? frames : 0; }

int gpujpeg_amd_decoder_get_path_counters(struct gpujpeg_decoder* d, long counters[3])
{
    if (!d || !counters) return -1;
    counters[0] = d->n_spec;
    counters[1] = d->n_folded;
    counters[2] = d->n_again;
    return 0;
}

int gpujpeg_amd_decoder_last_batch(struct gpujpeg_decoder* d, int* batched, int* single)
{
    if (!d) return -1;
    if (batched) *batched = d->b_last_batched;
    if (single) *single = d->b_last_single;
    return 0;
}

size_t gpujpeg_amd_decoder_read_coefficients(struct gpujpeg_decoder* d, int16_t* dst, size_t capacity)
{
    const size_t n = d->coder.geom.data_size;
    if (!d->coder.configured || capacity < n) return 0;
    if (gj_hip_memcpy_d2h(dst, d->coder.d_coefs, n * sizeof(int16_t), d->coder.stream) != 0 || gj_hip_stream_sync(d->coder.stream) != 0) return 0;
    return n;
}

size_t gpujpeg_amd_decoder_read_planes(struct gpujpeg_decoder* d, uint8_t* dst, size_t capacity)
{
    const size_t n = d->coder.geom.data_size;
    if (!d->coder.configured || capacity < n) return 0;
    if (gj_hip_memcpy_d2h(dst, d->coder.d_planes, n, d->coder.stream) != 0 || gj_hip_stream_sync(d->coder.stream) != 0) return 0;
    return n;
}

void gpujpeg_amd_decoder_set_fused(struct gpujpeg_decoder* d, int enabled) { d->use_fused = enabled != 0; }
void gpujpeg_amd_decoder_keep_coefficients(struct gpujpeg_decoder* d, int enabled) { d->keep_coefs = enabled != 0; }

/* durations of the kernels of the last decode: [0] k_huffman_decode, [1] IDCT (fused: incl. postprocess), [2] postprocess */
int gpujpeg_amd_decoder_get_kernel_times(struct gpujpeg_decoder* d, float ms[8])
{
    if (!d->coder.timers.valid) return -1;
    memcpy(ms, d->coder.kernel_ms, 8 * sizeof(float));
    return 0;
}
