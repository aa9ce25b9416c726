/*
 * gj_encoder.c -- the libgpujpeg encoder API on MI355X. Host driver counterpart of
 * src/gpujpeg_encoder.c: same entry points, argument meaning, ownership and error behaviour; the
 * work itself is one call into the HIP layer (gj_hip_encode) which leaves a finished JPEG in HBM.
 */
#define _GNU_SOURCE
#include <assert.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include "gj_internal.h"
#include "gpujpeg_amd_ext.h"

enum { GJ_OUT_PAGEABLE = 0, GJ_OUT_PINNED = 1, GJ_OUT_DEVICE = 2 };
#define GJ_MAIN_HEADER_CAP 4096 /* staging of the main header (the writer never stores behind it and reports the size it would need) */

struct gpujpeg_encoder {
    struct gj_coder coder;
    enum gpujpeg_header_type header_type;
    struct gpujpeg_image_metadata metadata;
    int out_location;
    bool flipped;
    unsigned channel_remap; /* packed like the reference's preprocessor.channel_remap, 0 = none */
    /* tables */
    int table_quality; /* quality the uploaded tables were computed for, -1 = none */
    uint8_t qraw[2][64];
    float* d_fwd_q[2];
    uint32_t* d_huff_lut;
    /* device work buffers */
    uint8_t* d_temp; size_t d_temp_cap;
    uint32_t* d_seg; size_t d_seg_cap; /* bytes | ff | out(+1) */
    uint8_t* d_jpeg; size_t d_jpeg_cap;
    uint32_t* d_result;
    uint64_t* d_scan_partial; size_t d_scan_partial_cap;
    uint32_t* d_tail; size_t d_tail_cap; /* k_encode_* -> k_gather: two sets of group totals + one entry per tile stream (GJ_TAIL_WORDS) */
    int tail_set;                        /* the set the next call uses */
    uint32_t epoch;
    uint8_t* d_scan_hdr; size_t d_scan_hdr_cap;
    struct gj_scan_headers scan_hdrs;
    /* host side */
    uint32_t* h_result; /* pinned */
    uint8_t* h_header;  /* pinned staging for the main header */
    uint8_t* hdr_sent; size_t hdr_sent_len; const uint8_t* hdr_sent_to; /* the header bytes that are at the start of d_jpeg already */
    struct gj_exif_tags* exif_tags; /* enc_exif_tag */
    uint8_t* out_buf; size_t out_cap; bool out_buf_pinned;
    int use_fused;
    gj_tuning tune; /* developer switches, read from the environment when the encoder is created */
    int keep_coefs; /* gpujpeg_amd_encoder_keep_coefficients */
    /* frame batches (gpujpeg_amd_encoder_encode_batch): work buffers of one chunk of frames, the streams and result words of all frames */
    uint8_t* b_temp; size_t b_temp_cap;
    uint32_t* b_seg; size_t b_seg_cap;
    uint32_t* b_tail; size_t b_tail_cap;
    uint8_t* b_jpeg; size_t b_jpeg_cap;
    uint32_t* b_result; size_t b_result_cap;
    uint32_t* bh_result; size_t bh_result_cap; /* pinned */
    uint8_t* b_raw; size_t b_raw_cap;          /* frames handed over in host memory */
    uint8_t* b_hdr_sent; size_t b_hdr_len; const uint8_t* b_hdr_to; size_t b_hdr_slot; int b_hdr_frames; /* the header bytes at the start of b_jpeg's slots */
    uint8_t* b_out; size_t b_out_cap; bool b_out_pinned; /* streams handed back in host memory */
    uint8_t* b_gather; size_t b_gather_cap;              /* frames given as separate buffers (encode_batch_ptrs), gathered back to back */
    int b_last_batched, b_last_single; /* frames of the last batch call coded by batched launches / frame by frame */
    int b_chunk;                       /* gpujpeg_amd_encoder_set_batch_chunk: frames per launch at most, 0 = the default */
};

/* ------------------------------------------------------------------ input helpers (gpujpeg_encoder.h:77-110) */
void gpujpeg_encoder_input_set_image(struct gpujpeg_encoder_input* in, uint8_t* image) { in->type = GPUJPEG_ENCODER_INPUT_IMAGE; in->image = image; in->texture = NULL; }
void gpujpeg_encoder_input_set_gpu_image(struct gpujpeg_encoder_input* in, uint8_t* image) { in->type = GPUJPEG_ENCODER_INPUT_GPU_IMAGE; in->image = image; in->texture = NULL; }
void gpujpeg_encoder_input_set_texture(struct gpujpeg_encoder_input* in, struct gpujpeg_opengl_texture* t) { in->type = GPUJPEG_ENCODER_INPUT_OPENGL_TEXTURE; in->image = NULL; in->texture = t; }
struct gpujpeg_encoder_input gpujpeg_encoder_input_image(uint8_t* image) { struct gpujpeg_encoder_input i; gpujpeg_encoder_input_set_image(&i, image); return i; }
struct gpujpeg_encoder_input gpujpeg_encoder_input_gpu_image(uint8_t* image) { struct gpujpeg_encoder_input i; gpujpeg_encoder_input_set_gpu_image(&i, image); return i; }
struct gpujpeg_encoder_input gpujpeg_encoder_input_texture(struct gpujpeg_opengl_texture* t) { struct gpujpeg_encoder_input i; gpujpeg_encoder_input_set_texture(&i, t); return i; }

/* ------------------------------------------------------------------ create / destroy (src/gpujpeg_encoder.c:105-163, 797-823) */
struct gpujpeg_encoder* gpujpeg_encoder_create(cudaStream_t stream)
{
    gj_init_term_colors();
    struct gpujpeg_encoder* e = calloc(1, sizeof *e);
    if (!e) return NULL;
    e->coder.encoder = true;
    e->coder.stream = (gj_stream_t)stream;
    e->table_quality = -1;
    gj_hip_tuning_defaults(&e->tune);
    e->coder.ht_on = e->tune.host_timing != 0;
    e->use_fused = !e->tune.no_fused;
    gpujpeg_set_default_parameters(&e->coder.param);
    gpujpeg_image_set_default_parameters(&e->coder.param_image);
    e->coder.param.comp_count = 0;
    if (gj_hip_get_device(&e->coder.device) != 0) goto fail;
    if (gj_timers_create(&e->coder.timers) != 0) goto fail;
    e->d_fwd_q[0] = gj_hip_malloc(64 * sizeof(float));
    e->d_fwd_q[1] = gj_hip_malloc(64 * sizeof(float));
    e->d_huff_lut = gj_hip_malloc((GJ_CODER_LUT_OFFSET + GJ_CODER_LUT_WORDS) * sizeof(uint32_t));
    e->d_result = gj_hip_malloc(4 * sizeof(uint32_t));
    e->h_result = gj_hip_host_alloc(4 * sizeof(uint32_t));
    e->h_header = gj_hip_host_alloc(GJ_MAIN_HEADER_CAP);
    if (!e->d_fwd_q[0] || !e->d_fwd_q[1] || !e->d_huff_lut || !e->d_result || !e->h_result || !e->h_header) goto fail;
    uint32_t lut[GJ_CODER_LUT_OFFSET + GJ_CODER_LUT_WORDS];
    gj_huffman_encoder_lut(lut);
    /* the same tables as the fused encoders' coder reads them (gj_encode.hip: GjCoderLds::lut): per table type 256 AC entries indexed by
     * (run << 4) | ((16 - nbits) & 15), then 16 DC entries indexed by nbits; entry = (code length + nbits) << 26 | code << nbits */
    for (int t = 0; t < GJ_CODER_LUT_WORDS; t++) {
        const int type = t >= 272, idx = t - type * 272, ac = idx < 256;
        const int sym = ac ? idx : idx - 256, nbits = ac ? (sym & 15) : sym;
        const uint32_t old = lut[(type * 2 + ac) * 256 + sym];
        /* (the AC entries sit at (run << 4) | ((16 - nbits) & 15): the walk indexes them with 32 - nbits as v_ffbh_i32 delivers it; EOB and ZRL, nbits = 0, stay where they were) */
        const int at = ac ? type * 272 + ((sym & 0xF0) | ((16 - nbits) & 15)) : t;
        lut[GJ_CODER_LUT_OFFSET + at] = (((old & 0xFFu) + (uint32_t)nbits) << 26) | ((old >> 8) << nbits);
    }
    if (gj_hip_memcpy_h2d(e->d_huff_lut, lut, sizeof lut, e->coder.stream) != 0 || gj_hip_stream_sync(e->coder.stream) != 0) goto fail;
    return e;
fail:
    GJ_ERROR("Encoder initialisation failed: %s\n", gj_hip_last_error());
    gpujpeg_encoder_destroy(e);
    return NULL;
}

int gpujpeg_encoder_destroy(struct gpujpeg_encoder* e)
{
    if (!e) return -1;
    gj_coder_process_stats_overall(&e->coder);
    gj_timers_destroy(&e->coder.timers);
    gj_hip_free(e->d_fwd_q[0]); gj_hip_free(e->d_fwd_q[1]); gj_hip_free(e->d_huff_lut); gj_hip_free(e->d_result);
    gj_hip_free(e->d_temp); gj_hip_free(e->d_tail); gj_hip_free(e->d_scan_partial); gj_hip_free(e->d_seg); gj_hip_free(e->d_jpeg); gj_hip_free(e->d_scan_hdr);
    gj_hip_free(e->coder.d_raw_own); gj_hip_free(e->coder.d_planes); gj_hip_free(e->coder.d_coefs);
    gj_hip_host_free(e->h_result); gj_hip_host_free(e->h_header); free(e->hdr_sent);
    gj_hip_free(e->b_gather);
    gj_hip_free(e->b_temp); gj_hip_free(e->b_seg); gj_hip_free(e->b_tail); gj_hip_free(e->b_jpeg); gj_hip_free(e->b_result); gj_hip_free(e->b_raw);
    gj_hip_host_free(e->bh_result); free(e->b_hdr_sent);
    if (e->b_out_pinned) gj_hip_host_free(e->b_out); else free(e->b_out);
    gj_exif_tags_destroy(e->exif_tags);
    if (e->out_buf_pinned) gj_hip_host_free(e->out_buf); else free(e->out_buf);
    free(e->scan_hdrs.bytes);
    free(e);
    return 0;
}

/* ------------------------------------------------------------------ parameter adjustment (src/gpujpeg_encoder.c:291-346) */
int gpujpeg_encoder_suggest_restart_interval(const struct gpujpeg_image_parameters* pi, gpujpeg_sampling_factor_t subsampling, bool interleaved, int verbose)
{
    const int comps = gpujpeg_pixel_format_get_comp_count(pi->pixel_format);
    const double mpix3 = ((double)pi->width * pi->height * comps) / (1000000.0 * 3.0);
    int ri = mpix3 < 1.0 ? 4 : mpix3 < 3.0 ? 8 : mpix3 < 9.0 ? 10 : 12;
    if (subsampling != GPUJPEG_SUBSAMPLING_444 && interleaved) ri /= 2; /* bigger MCUs */
    if (!interleaved) ri *= comps;                                       /* one scan per component */
    GJ_VERBOSE(verbose, "Auto-adjusting restart interval to %d for better performance.\n", ri);
    return ri;
}

static struct gpujpeg_parameters adjust_params(const struct gj_coder* c, const struct gpujpeg_parameters* param,
                                               const struct gpujpeg_image_parameters* pi, bool img_changed)
{
    struct gpujpeg_parameters p = *param;
    if (param->comp_count == 0) {
        if (img_changed) {
            const int n = gpujpeg_pixel_format_get_comp_count(pi->pixel_format);
            p.comp_count = n < 3 ? n : 3;
            memcpy(p.sampling_factor, gj_pixfmt_sampling(pi->pixel_format), sizeof p.sampling_factor);
        } else {
            p.comp_count = c->param.comp_count;
            memcpy(p.sampling_factor, c->param.sampling_factor, sizeof p.sampling_factor);
        }
    }
    if (param->restart_interval == RESTART_AUTO) {
        if (img_changed || p.interleaved != c->param.interleaved)
            p.restart_interval = gpujpeg_encoder_suggest_restart_interval(pi, gj_make_sampling_factor(p.comp_count, p.sampling_factor), p.interleaved, p.verbose);
        else
            p.restart_interval = c->param.restart_interval;
    }
    return p;
}

/* ------------------------------------------------------------------ (re)configuration */
static int encoder_configure(struct gpujpeg_encoder* e, const struct gpujpeg_parameters* p, const struct gpujpeg_image_parameters* pi)
{
    struct gj_coder* c = &e->coder;
    if (c->configured && gj_parameters_equal(&c->param, p) && gj_image_parameters_equal(&c->param_image, pi)) {
        c->param = *p; /* verbose / perf_stats / quality may change without reconfiguration (common.c:632-637) */
        c->param_image = *pi;
        return 0;
    }
    GJ_DEBUG(p->verbose, "coder image reconfiguration\n");
    c->configured = false;
    c->param = *p;
    c->param_image = *pi;
    gj_geom* g = &c->geom;
    if (gj_geom_init(g, p, pi, true) != 0) {
        GJ_ERROR("Failed to init image encoding!\n");
        return -1;
    }
    if (gj_ensure_device_buffer((void**)&c->d_coefs, &c->d_coefs_cap, g->data_size * sizeof(int16_t)) != 0) return -1;
    /* planes are needed by the generic path only; zero filled once, padding is never written (common.c:941-944) */
    if (gj_ensure_device_buffer((void**)&c->d_planes, &c->d_planes_cap, g->data_size) != 0) return -1;
    if (gj_hip_memset(c->d_planes, 0, g->data_size, c->stream) != 0) return -1;
    if (gj_ensure_device_buffer((void**)&e->d_temp, &e->d_temp_cap, (size_t)g->block_count * GJ_TEMP_BYTES_PER_BLOCK + 256) != 0) return -1;
    /* k_gather's counters and group totals are zero between calls (the kernel leaves them so); cleared here in case a failed launch did not */
    if (gj_ensure_device_buffer((void**)&e->d_tail, &e->d_tail_cap, (size_t)GJ_TAIL_WORDS(g->segment_count) * sizeof(uint32_t)) != 0) return -1;
    if (gj_hip_memset(e->d_tail, 0, 2 * (size_t)GJ_TAIL_GROUPS_CAP(g->segment_count) * sizeof(uint32_t), c->stream) != 0) return -1;
    e->tail_set = 0;
    {
        const size_t need = (((size_t)g->segment_count + 1023) / 1024 + 1) * sizeof(uint64_t);
        const size_t had = e->d_scan_partial_cap;
        if (gj_ensure_device_buffer((void**)&e->d_scan_partial, &e->d_scan_partial_cap, need) != 0) return -1;
        if (e->d_scan_partial_cap != had && gj_hip_memset(e->d_scan_partial, 0, e->d_scan_partial_cap, c->stream) != 0) return -1;
    }
    if (gj_ensure_device_buffer((void**)&e->d_seg, &e->d_seg_cap, ((size_t)g->segment_count * 3 + 16 + ((size_t)g->segment_count + 1023) / 1024) * sizeof(uint32_t)) != 0) return -1;
    if (gj_write_scan_headers(&e->scan_hdrs, g, p) != 0) return -1;
    if (gj_ensure_device_buffer((void**)&e->d_scan_hdr, &e->d_scan_hdr_cap, e->scan_hdrs.size + 16) != 0) return -1;
    if (gj_hip_memcpy_h2d(e->d_scan_hdr, e->scan_hdrs.bytes, e->scan_hdrs.size, c->stream) != 0) return -1;
    if (gj_hip_stream_sync(c->stream) != 0) return -1; /* scan_hdrs.bytes is pageable: finish before it can change */
    /* same sizing rule as the reference writer (writer.c:66-69) plus the scan headers */
    const size_t jpeg_cap = 1000 + e->scan_hdrs.size + (size_t)pi->width * pi->height * p->comp_count * 2 + 4096;
    if (gj_ensure_device_buffer((void**)&e->d_jpeg, &e->d_jpeg_cap, jpeg_cap) != 0) return -1;
    e->hdr_sent_to = NULL; /* (the buffer may be a new one: the main header has to be uploaded again) */
    c->configured = true;
    return 0;
}

static int ensure_out_buffer(struct gpujpeg_encoder* e, size_t need)
{
    const bool want_pinned = e->out_location == GJ_OUT_PINNED;
    if (need <= e->out_cap && want_pinned == e->out_buf_pinned) return 0;
    if (e->out_buf_pinned) gj_hip_host_free(e->out_buf); else free(e->out_buf);
    e->out_buf = want_pinned ? gj_hip_host_alloc(need) : malloc(need);
    e->out_buf_pinned = want_pinned;
    e->out_cap = e->out_buf ? need : 0;
    return e->out_buf ? 0 : -1;
}

/* ------------------------------------------------------------------ encode (src/gpujpeg_encoder.c:352-644) */
int gpujpeg_encoder_encode(struct gpujpeg_encoder* e, const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* pi,
                           const struct gpujpeg_encoder_input* input, uint8_t** image_compressed, size_t* image_compressed_size)
{
    assert(param->comp_count <= GPUJPEG_MAX_COMPONENT_COUNT);
    assert(param->quality >= 0 && param->quality <= 100);
    assert(param->restart_interval >= RESTART_AUTO);
    assert(param->interleaved == 0 || param->interleaved == 1);
    struct gj_coder* c = &e->coder;
    GJ_HT_START(c);
    const bool img_changed = !c->configured || !gj_image_parameters_equal(&c->param_image, pi);
    struct gpujpeg_parameters p = adjust_params(c, param, pi, img_changed);
    p.perf_stats = param->perf_stats || param->verbose >= GPUJPEG_LL_STATUS;
    const bool stats = p.perf_stats != 0;
    c->start_time = stats ? gpujpeg_get_time() : 0;

    if (e->table_quality != param->quality) { /* :369-380 */
        for (int t = 0; t < 2; t++) {
            float fwd[64];
            gj_quant_table_raw(t, param->quality, e->qraw[t]);
            gj_quant_table_forward(e->qraw[t], fwd);
            if (gj_hip_memcpy_h2d(e->d_fwd_q[t], fwd, sizeof fwd, c->stream) != 0) return -1;
        }
        if (gj_hip_stream_sync(c->stream) != 0) return -1; /* stack source */
        e->table_quality = param->quality;
    }
    if (encoder_configure(e, &p, pi) != 0) return -1;
    const gj_geom* g = &c->geom;
    c->init_end_time = stats ? gpujpeg_get_time() : 0;
    memset(&c->stats, 0, sizeof c->stats);

    /* input (:397-475) */
    const uint8_t* d_raw = NULL;
    if (input->type == GPUJPEG_ENCODER_INPUT_GPU_IMAGE || (input->type == GPUJPEG_ENCODER_INPUT_IMAGE && gj_hip_is_device_ptr(input->image))) {
        d_raw = input->image;
    } else if (input->type == GPUJPEG_ENCODER_INPUT_IMAGE) {
        if (gj_ensure_device_buffer((void**)&c->d_raw_own, &c->d_raw_cap, g->raw_size) != 0) return -1;
        gj_hip_event_record(c->timers.copy_in[0], c->stream); /* (also without perf_stats: see gj_internal.h, copy markers) */
        /* (through the process's upload lane unless the copy is timed on the coder's stream) */
        if (gj_hip_upload(c->d_raw_own, input->image, g->raw_size, c->stream, stats ? NULL : c->timers.lane_in) != 0) {
            GJ_ERROR("Encoder raw data copy failed: %s\n", gj_hip_last_error());
            return -1;
        }
        if (stats) gj_hip_event_record(c->timers.copy_in[1], c->stream);
        d_raw = c->d_raw_own;
    } else {
        GJ_ERROR("OpenGL texture input is not supported by the MI355X build.\n");
        return -1;
    }

    /* main header: host bytes, placed at the start of the device stream */
    const size_t hdr = gj_write_main_header(e->h_header, GJ_MAIN_HEADER_CAP, g, &p, e->header_type, (const uint8_t(*)[64])e->qraw, &e->metadata, e->exif_tags);
    if (hdr > GJ_MAIN_HEADER_CAP) {
        GJ_ERROR("The main header (%zu B: Exif tags, metadata) does not fit its %d B staging buffer.\n", hdr, GJ_MAIN_HEADER_CAP);
        return -1;
    }
    /* (the kernels write behind it, so it is uploaded again only when it changes -- parameters, or the wall clock of an Exif header -- or
     * the stream buffer was reallocated) */
    if (!e->hdr_sent) e->hdr_sent = malloc(GJ_MAIN_HEADER_CAP);
    if (!e->hdr_sent) return -1;
    if (e->hdr_sent_to != e->d_jpeg || e->hdr_sent_len != hdr || memcmp(e->hdr_sent, e->h_header, hdr) != 0) {
        if (gj_hip_memcpy_h2d(e->d_jpeg, e->h_header, hdr, c->stream) != 0 || gj_hip_stream_sync(c->stream) != 0) return -1; /* (h_header is rewritten by the next call) */
        memcpy(e->hdr_sent, e->h_header, hdr);
        e->hdr_sent_len = hdr;
        e->hdr_sent_to = e->d_jpeg;
    }

    gj_enc_job job;
    memset(&job, 0, sizeof job);
    job.g = *g;
    job.d_raw = d_raw;
    job.flipped = e->flipped;
    job.channel_remap = e->channel_remap;
    if (e->channel_remap) {
        const enum gpujpeg_pixel_format pf = c->param_image.pixel_format;
        if ((int)(e->channel_remap >> 24) != gpujpeg_pixel_format_get_comp_count(pf)) {
            GJ_ERROR("Wrong channel remapping given, given %u channels but pixel format has %d!\n", e->channel_remap >> 24, gpujpeg_pixel_format_get_comp_count(pf));
            return -1;
        }
        if (pf != GPUJPEG_U8 && pf != GPUJPEG_444_U8_P012 && pf != GPUJPEG_4444_U8_P0123 && pf != GPUJPEG_444_U8_P0P1P2) {
            GJ_ERROR("Channel remapping is implemented for pixel formats whose pixels do not share samples (u8, 444-u8-p012, 4444-u8-p0123, 444-u8-p0p1p2).\n");
            return -1;
        }
    }
    job.d_planes = c->d_planes;
    job.d_coefs = c->d_coefs;
    job.d_fwd_q[0] = e->d_fwd_q[0];
    job.d_fwd_q[1] = e->d_fwd_q[1];
    job.d_huff_lut = e->d_huff_lut;
    job.d_temp = e->d_temp;
    job.d_seg_bytes = e->d_seg;
    job.d_seg_ff = e->d_seg + g->segment_count;
    job.d_seg_out = e->d_seg + 2 * (size_t)g->segment_count;
    job.d_jpeg = e->d_jpeg;
    job.jpeg_capacity = e->d_jpeg_cap;
    job.d_result = e->d_result;
    job.h_result = e->h_result;
    job.d_scan_partial = e->d_scan_partial;
    job.d_tail = e->d_tail;
    job.tail_set = e->tail_set;
    if (++e->epoch == 0) e->epoch = 1;
    job.epoch = e->epoch;
    job.tune = e->tune;
    job.d_scan_hdr = e->d_scan_hdr;
    memcpy(job.scan_hdr_offset, e->scan_hdrs.offset, sizeof job.scan_hdr_offset);
    memcpy(job.scan_info_payload, e->scan_hdrs.info_payload, sizeof job.scan_info_payload);
    job.main_hdr_size = (uint32_t)hdr;
    job.segment_info = p.segment_info;
    job.use_fused = e->use_fused && !e->flipped;
    job.keep_coefs = e->keep_coefs;
    /* the sets of group totals alternate only between calls that end in k_gather, which clears the idle one: a call that takes the
     * coefficient planes (flip, kept coefficients, fused path off) leaves both as they are (it used to flip, and the next
     * tile-path call added its sizes to an older frame's totals) */
    if (gj_hip_encode_tiles(&job)) e->tail_set ^= 1;
    GJ_HT(c, 0);
    if (gj_hip_encode(&job, c->stream, stats ? c->timers.ev : NULL) != 0) {
        GJ_ERROR("Encoder kernels failed: %s\n", gj_hip_last_error());
        c->configured = false; /* (the next call sets the device-side state up again) */
        return -1;
    }
    GJ_HT(c, 1);
    /* size first, then the bytes (:550-563) */
    if (gj_hip_stream_sync(c->stream) != 0) { /* (the two result words are in host memory once the kernels have run) */
        GJ_ERROR("Encoder failed: %s\n", gj_hip_last_error());
        c->configured = false;
        return -1;
    }
    GJ_HT(c, 2);
    const size_t size = e->h_result[0];
    if (e->h_result[1]) {
        GJ_ERROR("Compressed stream (%zu B) does not fit the output buffer (%zu B)!\n", size, e->d_jpeg_cap);
        return -1;
    }
    if (e->out_location == GJ_OUT_DEVICE) {
        *image_compressed = e->d_jpeg;
    } else {
        if (ensure_out_buffer(e, e->d_jpeg_cap) != 0) return -1;
        gj_hip_event_record(c->timers.copy_out[0], c->stream); /* (copy marker) */
        /* (the compressed stream stays on the coder's own stream: queued behind other coders' images in the download lane it came out 10-40 %
         * slower with four coders, profiles/r5_07) */
        if (gj_hip_memcpy_d2h(e->out_buf, e->d_jpeg, size, c->stream) != 0) return -1;
        if (stats) gj_hip_event_record(c->timers.copy_out[1], c->stream);
        if (gj_hip_stream_sync(c->stream) != 0) return -1;
        *image_compressed = e->out_buf;
    }
    *image_compressed_size = size;

    if (stats) {
        struct gpujpeg_duration_stats* s = &c->stats;
        s->duration_preprocessor = gj_hip_event_elapsed_ms(c->timers.ev[0], c->timers.ev[1]);
        s->duration_dct_quantization = gj_hip_event_elapsed_ms(c->timers.ev[1], c->timers.ev[2]);
        s->duration_huffman_coder = gj_hip_event_elapsed_ms(c->timers.ev[2], c->timers.ev[5]);
        s->duration_in_gpu = gj_hip_event_elapsed_ms(c->timers.ev[0], c->timers.ev[5]);
        for (int k = 0; k < GJ_ENC_EVENTS - 1; k++) c->kernel_ms[k] = gj_hip_event_elapsed_ms(c->timers.ev[k], c->timers.ev[k + 1]);
        if (d_raw == c->d_raw_own) s->duration_memory_to = gj_hip_event_elapsed_ms(c->timers.copy_in[0], c->timers.copy_in[1]);
        if (e->out_location != GJ_OUT_DEVICE) s->duration_memory_from = gj_hip_event_elapsed_ms(c->timers.copy_out[0], c->timers.copy_out[1]);
        c->timers.valid = true;
    }
    gj_coder_process_stats(c, stats);
    if (c->param.verbose >= GPUJPEG_LL_STATUS) {
        const char* il = p.comp_count == 1 ? "" : (p.interleaved ? " interleaved" : " non-interleaved");
        fprintf(stderr, "Compressed Size:%15zu bytes %dx%d %s %s%s\n", size, pi->width, pi->height,
                gpujpeg_color_space_get_name(p.color_space_internal), gpujpeg_subsampling_get_name(p.comp_count, p.sampling_factor), il);
    }
    GJ_HT(c, 3);
    c->ht_calls++;
    return 0;
}

int gpujpeg_encoder_get_stats(struct gpujpeg_encoder* e, struct gpujpeg_duration_stats* stats)
{
    if (!e || !stats) return -1;
    *stats = e->coder.stats;
    return 0;
}


/* ------------------------------------------------------------------ frame batches (MI355X extension, include/gpujpeg_amd_ext.h) */
/* Frames of one geometry coded by ONE pair of launches per chunk (k_encode_rgb444 + k_gather with blockIdx.z = frame): an HD frame has 135
 * tiles for the 1024 workgroup places of the device and six dependent launches per encode + decode, so frame-at-a-time calls leave the GPU
 * mostly idle however many coders run side by side; a batch fills it. Same bytes as gpujpeg_encoder_encode frame by frame (which is what
 * happens for configurations outside the fully fused 4:4:4 kernel). */
#define GJ_BATCH_CHUNK_MAX 256 /* frames per launch (measured: 256 x HD through one pipeline 69.3k frames/s at 64, 71.6k at 128, 73.9k at 256; larger
                                   frames are bounded by the bytes below) */
#define GJ_BATCH_TEMP_BYTES ((size_t)6 << 30) /* tile areas of one chunk (address space: only the streams' bytes are touched) */

int gpujpeg_amd_encoder_encode_batch(struct gpujpeg_encoder* e, const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* pi,
                                     const uint8_t* frames, size_t frame_stride, int count, uint8_t** images_compressed, size_t* images_compressed_size)
{
    if (!e || !param || !pi || !frames || count < 1 || !images_compressed || !images_compressed_size) return -1;
    struct gj_coder* c = &e->coder;
    const bool img_changed = !c->configured || !gj_image_parameters_equal(&c->param_image, pi);
    struct gpujpeg_parameters p = adjust_params(c, param, pi, img_changed);
    p.perf_stats = 0;
    if (e->table_quality != param->quality) {
        for (int t = 0; t < 2; t++) {
            float fwd[64];
            gj_quant_table_raw(t, param->quality, e->qraw[t]);
            gj_quant_table_forward(e->qraw[t], fwd);
            if (gj_hip_memcpy_h2d(e->d_fwd_q[t], fwd, sizeof fwd, c->stream) != 0) return -1;
        }
        if (gj_hip_stream_sync(c->stream) != 0) return -1;
        e->table_quality = param->quality;
    }
    if (encoder_configure(e, &p, pi) != 0) return -1;
    const gj_geom* g = &c->geom;
    if (frame_stride < g->raw_size) {
        GJ_ERROR("Frame stride %zu is smaller than a frame (%zu B)!\n", frame_stride, (size_t)g->raw_size);
        return -1;
    }
    const bool frames_on_device = gj_hip_is_device_ptr(frames) != 0;
    /* the streams of ALL frames stay valid until the next call: a slot per frame, sized like the single-frame buffer */
    const size_t slot = (e->d_jpeg_cap + 255) & ~(size_t)255;
    if (gj_ensure_device_buffer((void**)&e->b_jpeg, &e->b_jpeg_cap, slot * (size_t)count) != 0) return -1;
    if (gj_ensure_device_buffer((void**)&e->b_result, &e->b_result_cap, (size_t)count * 2 * sizeof(uint32_t)) != 0) return -1;
    if ((size_t)count * 2 * sizeof(uint32_t) > e->bh_result_cap) {
        gj_hip_host_free(e->bh_result);
        e->bh_result_cap = (size_t)count * 2 * sizeof(uint32_t) * 2;
        e->bh_result = gj_hip_host_alloc(e->bh_result_cap);
        if (!e->bh_result) { e->bh_result_cap = 0; return -1; }
    }

    gj_enc_job job;
    memset(&job, 0, sizeof job);
    job.g = *g;
    job.flipped = e->flipped;
    job.channel_remap = e->channel_remap;
    job.d_planes = c->d_planes;
    job.d_coefs = c->d_coefs;
    job.d_fwd_q[0] = e->d_fwd_q[0];
    job.d_fwd_q[1] = e->d_fwd_q[1];
    job.d_huff_lut = e->d_huff_lut;
    job.tune = e->tune;
    job.d_scan_hdr = e->d_scan_hdr;
    memcpy(job.scan_hdr_offset, e->scan_hdrs.offset, sizeof job.scan_hdr_offset);
    memcpy(job.scan_info_payload, e->scan_hdrs.info_payload, sizeof job.scan_info_payload);
    job.segment_info = p.segment_info;
    job.use_fused = e->use_fused && !e->flipped;
    job.keep_coefs = e->keep_coefs;

    if (!gj_hip_encode_batchable(&job)) {
        /* another configuration than the fully fused kernel's: frame by frame, every stream copied to its slot */
        for (int f = 0; f < count; f++) {
            struct gpujpeg_encoder_input in;
            if (frames_on_device) gpujpeg_encoder_input_set_gpu_image(&in, (uint8_t*)(uintptr_t)(frames + (size_t)f * frame_stride));
            else gpujpeg_encoder_input_set_image(&in, (uint8_t*)(uintptr_t)(frames + (size_t)f * frame_stride));
            const int loc = e->out_location;
            e->out_location = GJ_OUT_DEVICE;
            uint8_t* one = NULL;
            size_t n = 0;
            const int rc = gpujpeg_encoder_encode(e, param, pi, &in, &one, &n);
            e->out_location = loc;
            if (rc != 0 || n > slot) return -1;
            if (gj_hip_memcpy_d2d(e->b_jpeg + (size_t)f * slot, one, n, c->stream) != 0 || gj_hip_stream_sync(c->stream) != 0) return -1;
            e->bh_result[2 * f] = (uint32_t)n;
            e->bh_result[2 * f + 1] = 0;
        }
        e->b_hdr_to = NULL; /* (the slots' headers were overwritten by whole streams) */
        e->b_last_batched = 0;
        e->b_last_single = count;
    } else {
        e->b_last_batched = count;
        e->b_last_single = 0;
        /* main header: the same bytes at the start of every slot, uploaded when they change (as gpujpeg_encoder_encode does for its one buffer) */
        const size_t hdr = gj_write_main_header(e->h_header, GJ_MAIN_HEADER_CAP, g, &p, e->header_type, (const uint8_t(*)[64])e->qraw, &e->metadata, e->exif_tags);
        if (hdr > GJ_MAIN_HEADER_CAP) {
            GJ_ERROR("The main header (%zu B: Exif tags, metadata) does not fit its %d B staging buffer.\n", hdr, GJ_MAIN_HEADER_CAP);
            return -1;
        }
        if (!e->b_hdr_sent) e->b_hdr_sent = malloc(GJ_MAIN_HEADER_CAP);
        if (!e->b_hdr_sent) return -1;
        if (e->b_hdr_to != e->b_jpeg || e->b_hdr_slot != slot || e->b_hdr_frames < count || e->b_hdr_len != hdr || memcmp(e->b_hdr_sent, e->h_header, hdr) != 0) {
            for (int f = 0; f < count; f++)
                if (gj_hip_memcpy_h2d(e->b_jpeg + (size_t)f * slot, e->h_header, hdr, c->stream) != 0) return -1;
            if (gj_hip_stream_sync(c->stream) != 0) return -1;
            memcpy(e->b_hdr_sent, e->h_header, hdr);
            e->b_hdr_len = hdr;
            e->b_hdr_to = e->b_jpeg;
            e->b_hdr_slot = slot;
            e->b_hdr_frames = count;
        }
        /* chunks: as many frames per launch as their tile areas may take */
        const size_t temp_frame = (((size_t)g->block_count * GJ_TEMP_BYTES_PER_BLOCK + 256) + 255) & ~(size_t)255;
        const size_t seg_frame = ((size_t)g->segment_count * 2 + 16 + 3) & ~(size_t)3;          /* words: bytes | ff */
        const size_t tail_frame = ((size_t)GJ_TAIL_WORDS(g->segment_count) + 3) & ~(size_t)3; /* words */
        int chunk = (int)(GJ_BATCH_TEMP_BYTES / temp_frame);
        if (chunk > GJ_BATCH_CHUNK_MAX) chunk = GJ_BATCH_CHUNK_MAX;
        if (e->b_chunk > 0 && chunk > e->b_chunk) chunk = e->b_chunk;
        if (chunk > count) chunk = count;
        if (chunk < 1) chunk = 1;
        if (gj_ensure_device_buffer((void**)&e->b_temp, &e->b_temp_cap, temp_frame * (size_t)chunk) != 0) return -1;
        if (gj_ensure_device_buffer((void**)&e->b_seg, &e->b_seg_cap, seg_frame * (size_t)chunk * sizeof(uint32_t)) != 0) return -1;
        if (gj_ensure_device_buffer((void**)&e->b_tail, &e->b_tail_cap, tail_frame * (size_t)chunk * sizeof(uint32_t)) != 0) return -1;
        const uint8_t* d_frames = frames;
        size_t d_stride = frame_stride;
        if (!frames_on_device) {
            if (gj_ensure_device_buffer((void**)&e->b_raw, &e->b_raw_cap, (size_t)g->raw_size * (size_t)count) != 0) return -1;
            gj_hip_event_record(c->timers.copy_in[0], c->stream); /* (copy marker, see gj_internal.h) */
            const gj_stream_t up = gj_hip_lane_begin(0, g->raw_size, c->stream); /* (the process's upload lane for frames of 1 MiB and more) */
            int copies = 0;
            for (int f = 0; f < count && copies == 0; f++)
                copies = gj_hip_memcpy_h2d(e->b_raw + (size_t)f * g->raw_size, frames + (size_t)f * frame_stride, g->raw_size, up);
            /* (a copy that could not be queued: the ones before it still read THE CALLER'S BUFFERS -- wait for the lane before the error leaves) */
            if (gj_hip_lane_end(up, c->stream, c->timers.lane_in) != 0 || copies != 0) return -1;
            d_frames = e->b_raw;
            d_stride = g->raw_size;
        }
        job.main_hdr_size = (uint32_t)hdr;
        job.jpeg_capacity = e->d_jpeg_cap;
        job.d_temp = e->b_temp;
        job.d_seg_bytes = e->b_seg;
        job.d_seg_ff = e->b_seg + g->segment_count;
        job.d_seg_out = NULL;
        job.d_scan_partial = e->d_scan_partial;
        job.d_tail = e->b_tail;
        job.tail_set = 0;
        job.batch.raw = d_stride;
        job.batch.jpeg = slot;
        job.batch.temp = temp_frame;
        job.batch.seg = (uint32_t)seg_frame;
        job.batch.tail = (uint32_t)tail_frame;
        for (int f0 = 0; f0 < count; f0 += chunk) {
            const int n = count - f0 < chunk ? count - f0 : chunk;
            /* (the group totals a launch adds to: zero for every frame of the chunk, whatever the chunks before it looked like) */
            if (gj_hip_memset(e->b_tail, 0, tail_frame * (size_t)n * sizeof(uint32_t), c->stream) != 0) return -1;
            job.d_raw = d_frames + (size_t)f0 * d_stride;
            job.d_jpeg = e->b_jpeg + (size_t)f0 * slot;
            job.d_result = e->b_result + 2 * (size_t)f0;
            job.h_result = e->bh_result + 2 * (size_t)f0;
            job.batch.count = (uint32_t)n;
            if (++e->epoch == 0) e->epoch = 1;
            job.epoch = e->epoch;
            if (gj_hip_encode(&job, c->stream, NULL) != 0) {
                GJ_ERROR("Encoder kernels failed: %s\n", gj_hip_last_error());
                c->configured = false;
                return -1;
            }
        }
        if (gj_hip_stream_sync(c->stream) != 0) {
            GJ_ERROR("Encoder failed: %s\n", gj_hip_last_error());
            c->configured = false;
            return -1;
        }
    }
    size_t longest = 0;
    for (int f = 0; f < count; f++) {
        if (e->bh_result[2 * f + 1]) {
            GJ_ERROR("Compressed stream (%u B) of frame %d does not fit the output buffer (%zu B)!\n", e->bh_result[2 * f], f, e->d_jpeg_cap);
            return -1;
        }
        images_compressed_size[f] = e->bh_result[2 * f];
        if (e->bh_result[2 * f] > longest) longest = e->bh_result[2 * f];
    }
    /* (host copies too lie a constant number of bytes apart -- images_compressed[1] - images_compressed[0], a multiple of 16 --, so that the
     * pointers can go straight into gpujpeg_amd_decoder_decode_batch) */
    const size_t host_stride = (longest + 64 + 15) & ~(size_t)15, total = host_stride * (size_t)count;
    if (e->out_location == GJ_OUT_DEVICE) {
        for (int f = 0; f < count; f++) images_compressed[f] = e->b_jpeg + (size_t)f * slot;
    } else {
        const bool want_pinned = e->out_location == GJ_OUT_PINNED;
        if (total > e->b_out_cap || want_pinned != e->b_out_pinned) {
            if (e->b_out_pinned) gj_hip_host_free(e->b_out); else free(e->b_out);
            e->b_out = want_pinned ? gj_hip_host_alloc(total + total / 4) : malloc(total + total / 4);
            e->b_out_pinned = want_pinned;
            e->b_out_cap = e->b_out ? total + total / 4 : 0;
            if (!e->b_out) return -1;
        }
        gj_hip_event_record(c->timers.copy_out[0], c->stream); /* (copy marker) */
        for (int f = 0; f < count; f++) {
            if (gj_hip_memcpy_d2h(e->b_out + (size_t)f * host_stride, e->b_jpeg + (size_t)f * slot, images_compressed_size[f], c->stream) != 0) return -1;
            images_compressed[f] = e->b_out + (size_t)f * host_stride;
        }
        if (gj_hip_stream_sync(c->stream) != 0) return -1;
    }
    return 0; /* (c->frames counts the frames timed with perf_stats: a batch adds none, see gj_coder_process_stats_overall) */
}

/* The same for frames that are separate buffers (device or host memory, mixed if need be). Buffers that happen to lie a constant distance apart
 * are coded where they are; otherwise they are gathered back to back in a staging buffer first (one device copy per frame in front of the kernels:
 * ~4 us each, i.e. a third of the time an HD frame takes in a batch and a tenth of a 4K frame's). */
int gpujpeg_amd_encoder_encode_batch_ptrs(struct gpujpeg_encoder* e, const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* pi,
                                          const uint8_t* const* frames, int count, uint8_t** images_compressed, size_t* images_compressed_size)
{
    if (!e || !param || !pi || !frames || count < 1) return -1;
    for (int f = 0; f < count; f++)
        if (!frames[f]) return -1;
    const size_t raw = gpujpeg_image_calculate_size((struct gpujpeg_image_parameters*)(uintptr_t)pi);
    if (raw == 0) return -1;
    /* a constant stride between buffers of one kind: nothing to gather */
    bool strided = true;
    const int dev0 = gj_hip_is_device_ptr(frames[0]);
    /* (addresses compared as integers: the buffers may be unrelated allocations, whose pointers C does not let us subtract) */
    const ptrdiff_t step = count > 1 ? (ptrdiff_t)((uintptr_t)frames[1] - (uintptr_t)frames[0]) : (ptrdiff_t)raw;
    for (int f = 1; f < count && strided; f++)
        strided = (ptrdiff_t)((uintptr_t)frames[f] - (uintptr_t)frames[f - 1]) == step && gj_hip_is_device_ptr(frames[f]) == dev0;
    if (strided && step >= (ptrdiff_t)raw)
        return gpujpeg_amd_encoder_encode_batch(e, param, pi, frames[0], (size_t)step, count, images_compressed, images_compressed_size);
    if (gj_ensure_device_buffer((void**)&e->b_gather, &e->b_gather_cap, raw * (size_t)count) != 0) return -1;
    for (int f = 0; f < count; f++) {
        const int rc = gj_hip_is_device_ptr(frames[f]) ? gj_hip_memcpy_d2d(e->b_gather + (size_t)f * raw, frames[f], raw, e->coder.stream)
                                                       : gj_hip_memcpy_h2d(e->b_gather + (size_t)f * raw, frames[f], raw, e->coder.stream);
        if (rc != 0) return -1;
    }
    /* (the copies and the kernels are on the coder's stream: ordered. Host sources must stay untouched until the call returns, which it does after
     * a synchronisation) */
    return gpujpeg_amd_encoder_encode_batch(e, param, pi, e->b_gather, raw, count, images_compressed, images_compressed_size);
}

/* ------------------------------------------------------------------ memory planning (src/gpujpeg_encoder.c:165-288) */
size_t gpujpeg_encoder_max_memory(struct gpujpeg_parameters* param, struct gpujpeg_image_parameters* pi, enum gpujpeg_encoder_input_type type, int max_pixels)
{
    struct gpujpeg_image_parameters t = *pi;
    t.width = (int)(max_pixels > 0 ? (max_pixels < 65535 ? max_pixels : 65535) : 1);
    t.height = (max_pixels + t.width - 1) / t.width;
    if (t.height < 1) t.height = 1;
    struct gpujpeg_parameters p = *param;
    if (p.comp_count == 0) {
        const int n = gpujpeg_pixel_format_get_comp_count(pi->pixel_format);
        p.comp_count = n < 3 ? n : 3;
        memcpy(p.sampling_factor, gj_pixfmt_sampling(pi->pixel_format), sizeof p.sampling_factor);
    }
    if (p.restart_interval == RESTART_AUTO) p.restart_interval = gpujpeg_encoder_suggest_restart_interval(&t, gj_make_sampling_factor(p.comp_count, p.sampling_factor), p.interleaved, -1);
    gj_geom g;
    if (gj_geom_init(&g, &p, &t, true) != 0) return 0;
    size_t total = g.data_size * 3 + (size_t)g.block_count * GJ_TEMP_BYTES_PER_BLOCK + (size_t)g.segment_count * 20 +
                   1000 + (size_t)t.width * t.height * p.comp_count * 2;
    if (type == GPUJPEG_ENCODER_INPUT_IMAGE) total += g.raw_size;
    return total;
}

size_t gpujpeg_encoder_max_pixels(struct gpujpeg_parameters* param, struct gpujpeg_image_parameters* pi, enum gpujpeg_encoder_input_type type,
                                  size_t memory_size, int* max_pixels)
{
    int lo = 0, hi = 1 << 30;
    while (hi - lo > 1024) { /* bisection over the monotone memory model */
        const int mid = lo + (hi - lo) / 2;
        const size_t need = gpujpeg_encoder_max_memory(param, pi, type, mid);
        if (need != 0 && need <= memory_size) lo = mid; else hi = mid;
    }
    if (max_pixels) *max_pixels = lo;
    return lo ? gpujpeg_encoder_max_memory(param, pi, type, lo) : 0;
}

int gpujpeg_encoder_allocate(struct gpujpeg_encoder* e, const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* pi,
                             enum gpujpeg_encoder_input_type type)
{
    struct gpujpeg_parameters p = adjust_params(&e->coder, param, pi, true);
    if (encoder_configure(e, &p, pi) != 0) return -1;
    if (type == GPUJPEG_ENCODER_INPUT_IMAGE && gj_ensure_device_buffer((void**)&e->coder.d_raw_own, &e->coder.d_raw_cap, e->coder.geom.raw_size) != 0)
        return -1;
    return 0;
}

/* ------------------------------------------------------------------ options (src/gpujpeg_encoder.c:648-795) */
void gpujpeg_encoder_set_jpeg_header(struct gpujpeg_encoder* e, enum gpujpeg_header_type t) { e->header_type = t; }

static int parse_bool(bool* out, const char* val, const char* opt)
{
    if (strcasecmp(val, GPUJPEG_VAL_TRUE) == 0) { *out = true; return GPUJPEG_NOERR; }
    if (strcasecmp(val, GPUJPEG_VAL_FALSE) == 0) { *out = false; return GPUJPEG_NOERR; }
    GJ_ERROR("Unknown option %s for %s\n", val, opt);
    return GPUJPEG_ERROR;
}

int gpujpeg_encoder_set_option(struct gpujpeg_encoder* e, const char* opt, const char* val)
{
    if (e == NULL || opt == NULL || val == NULL) return GPUJPEG_ERROR;
    if (strcmp(opt, GPUJPEG_ENCODER_OPT_OUT_PINNED) == 0) {
        bool b;
        if (parse_bool(&b, val, opt) != GPUJPEG_NOERR) return GPUJPEG_ERROR;
        e->out_location = b ? GJ_OUT_PINNED : GJ_OUT_PAGEABLE;
        return GPUJPEG_NOERR;
    }
    if (strcmp(opt, GPUJPEG_ENC_OPT_OUT) == 0) {
        if (strcmp(val, GPUJPEG_ENC_OUT_VAL_PAGEABLE) == 0) e->out_location = GJ_OUT_PAGEABLE;
        else if (strcmp(val, GPUJPEG_ENC_OUT_VAL_PINNED) == 0) e->out_location = GJ_OUT_PINNED;
        else if (strcmp(val, GPUJPEG_ENC_OUT_VAL_DEVICE) == 0) e->out_location = GJ_OUT_DEVICE;
        else {
            GJ_ERROR("Unknown encoder output type: %s\n", val);
            return GPUJPEG_ERROR;
        }
        return GPUJPEG_NOERR;
    }
    if (strcmp(opt, GPUJPEG_ENC_OPT_HDR) == 0) {
        const enum gpujpeg_header_type t = gpujpeg_header_type_by_name(val);
        if (t == GPUJPEG_HEADER_DEFAULT) {
            GJ_ERROR("Unknown encoder header type: %s\n", val);
            return GPUJPEG_ERROR;
        }
        e->header_type = t;
        return GPUJPEG_NOERR;
    }
    if (strcmp(opt, GPUJPEG_ENC_OPT_FLIPPED_BOOL) == 0) return parse_bool(&e->flipped, val, opt); /* src/gpujpeg_encoder.c:767-769 */
    if (strcmp(opt, GPUJPEG_ENC_OPT_CHANNEL_REMAP) == 0) return gj_parse_channel_remap(&e->channel_remap, val, opt);
    if (strcmp(opt, GPUJPEG_ENC_OPT_METADATA) == 0) { /* src/gpujpeg_encoder.c:700-732: orientation=<deg>[-], carried by the SPIFF directory */
        if (strstr(val, "help") != NULL) {
            printf(GPUJPEG_ENC_OPT_METADATA " usage:\n");
            printf("\t" GPUJPEG_ENC_OPT_METADATA "=orientation=<deg>[-]\n");
            printf("\t\t<deg> - clockwise rotation - 0, 90, 180 or 270 degrees\n");
            printf("\t\t'-'   - mirror the image horizontally after rotation applied\n");
            return GPUJPEG_ERROR;
        }
        if (strstr(val, "orientation=") != val) {
            printf("Wrong metadata item: %s\n", val);
            return GPUJPEG_ERROR;
        }
        char* endptr = NULL;
        const int deg = (int)strtol(strchr(val, '=') + 1, &endptr, 10);
        bool flip = false;
        if (*endptr == '-') { flip = true; endptr++; }
        if (*endptr != '\0' || deg < 0 || deg > 270 || deg % 90 != 0) {
            printf("Wrong orientation value: %s\n", val);
            return GPUJPEG_ERROR;
        }
        e->metadata.vals[GPUJPEG_METADATA_ORIENTATION].set = 1;
        e->metadata.vals[GPUJPEG_METADATA_ORIENTATION].orient.rotation = (unsigned)deg / 90;
        e->metadata.vals[GPUJPEG_METADATA_ORIENTATION].orient.flip = flip;
        return GPUJPEG_NOERR;
    }
    if (strcmp(opt, GPUJPEG_ENC_OPT_EXIF_TAG) == 0) { /* src/gpujpeg_encoder.c:773-776 */
        e->header_type = GPUJPEG_HEADER_EXIF;
        return gj_exif_add_tag(&e->exif_tags, val) == 0 ? GPUJPEG_NOERR : GPUJPEG_ERROR;
    }
    GJ_ERROR("Invalid encoder option: %s!\n", opt);
    return GPUJPEG_ERROR;
}

void gpujpeg_encoder_print_options(void)
{
    printf("\t" GPUJPEG_ENC_OPT_OUT "=[" GPUJPEG_ENC_OUT_VAL_PAGEABLE "|" GPUJPEG_ENC_OUT_VAL_PINNED "|" GPUJPEG_ENC_OUT_VAL_DEVICE
           "] - compressed data buffer allocation property (" GPUJPEG_ENC_OUT_VAL_DEVICE ": the stream stays in device memory)\n");
    printf("\t" GPUJPEG_ENC_OPT_HDR "=[" GPUJPEG_ENC_HDR_VAL_JFIF "|" GPUJPEG_ENC_HDR_VAL_ADOBE "|" GPUJPEG_ENC_HDR_VAL_EXIF "|" GPUJPEG_ENC_HDR_VAL_SPIFF
           "] - output JPEG header\n");
    printf("\t" GPUJPEG_ENC_OPT_FLIPPED_BOOL "=[" GPUJPEG_VAL_FALSE "|" GPUJPEG_VAL_TRUE
           "] - whether is the input image should be vertically flipped (prior encode)\n");
    printf("\t" GPUJPEG_ENC_OPT_CHANNEL_REMAP "=XYZ[W] - input channel mapping, eg. '210F' for GBRX,\n"
           "\t\t'210' for GBR; special placeholders 'F' and 'Z' to set a channel to all-ones or all-zeros\n");
    printf("\t" GPUJPEG_ENC_OPT_EXIF_TAG "=<key>=<value>|help - custom EXIF tag (use help for syntax)\n");
    printf("\t" GPUJPEG_ENC_OPT_METADATA "=<key>=<value>|help - set image metadata\n");
}

/* ------------------------------------------------------------------ MI355X extensions (include/gpujpeg_amd_ext.h) */

void gpujpeg_amd_encoder_set_batch_chunk(struct gpujpeg_encoder* e, int frames) { if (e) e->b_chunk = frames > 0 ? frames : 0; }

int gpujpeg_amd_encoder_last_batch(struct gpujpeg_encoder* e, int* batched, int* single)
{
    if (!e) return -1;
    if (batched) *batched = e->b_last_batched;
    if (single) *single = e->b_last_single;
    return 0;
}

size_t gpujpeg_amd_encoder_read_coefficients(struct gpujpeg_encoder* e, int16_t* dst, size_t capacity)
{
    const size_t n = e->coder.geom.data_size;
    if (!e->coder.configured || capacity < n) return 0;
    if (gj_hip_memcpy_d2h(dst, e->coder.d_coefs, n * sizeof(int16_t), e->coder.stream) != 0 || gj_hip_stream_sync(e->coder.stream) != 0) return 0;
    return n;
}

size_t gpujpeg_amd_encoder_read_planes(struct gpujpeg_encoder* e, uint8_t* dst, size_t capacity)
{
    const size_t n = e->coder.geom.data_size;
    if (!e->coder.configured || capacity < n) return 0;
    if (gj_hip_memcpy_d2h(dst, e->coder.d_planes, n, e->coder.stream) != 0 || gj_hip_stream_sync(e->coder.stream) != 0) return 0;
    return n;
}

void gpujpeg_amd_encoder_set_fused(struct gpujpeg_encoder* e, int enabled) { e->use_fused = enabled != 0; }
void gpujpeg_amd_encoder_keep_coefficients(struct gpujpeg_encoder* e, int enabled) { e->keep_coefs = enabled != 0; }

/* durations of the kernels of the last encode: [0] preprocess (generic path), [1] DCT+quant (fused: incl. preprocess),
 * [2] k_huffman, [3] k_scan_segments, [4] k_assemble; needs perf_stats or verbose >= 1 */
int gpujpeg_amd_encoder_get_kernel_times(struct gpujpeg_encoder* e, float ms[8])
{
    if (!e->coder.timers.valid) return -1;
    memcpy(ms, e->coder.kernel_ms, 8 * sizeof(float));
    return 0;
}

static int host_adjusted(const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* pi, struct gpujpeg_parameters* p, gj_geom* g)
{
    struct gj_coder fresh;
    memset(&fresh, 0, sizeof fresh);
    *p = adjust_params(&fresh, param, pi, true);
    return gj_geom_init(g, p, pi, true);
}

size_t gpujpeg_amd_host_headers(const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* pi, int header_type,
                                uint8_t* dst, size_t capacity, size_t* main_header_size)
{
    return gpujpeg_amd_host_headers_md(param, pi, header_type, -1, 0, dst, capacity, main_header_size);
}

size_t gpujpeg_amd_host_headers_md(const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* pi, int header_type,
                                   int rotation, int flip, uint8_t* dst, size_t capacity, size_t* main_header_size)
{
    return gpujpeg_amd_host_headers_exif(param, pi, header_type, rotation, flip, NULL, 0, dst, capacity, main_header_size);
}

size_t gpujpeg_amd_host_headers_exif(const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* pi, int header_type,
                                     int rotation, int flip, const char* const* exif_tags, int exif_tag_count, uint8_t* dst, size_t capacity,
                                     size_t* main_header_size)
{
    struct gj_exif_tags* tags = NULL;
    for (int i = 0; i < exif_tag_count; i++) {
        if (gj_exif_add_tag(&tags, exif_tags[i]) != 0) {
            gj_exif_tags_destroy(tags);
            return 0;
        }
        header_type = GPUJPEG_HEADER_EXIF;
    }
    struct gpujpeg_image_metadata md;
    memset(&md, 0, sizeof md);
    if (rotation >= 0) {
        md.vals[GPUJPEG_METADATA_ORIENTATION].set = 1;
        md.vals[GPUJPEG_METADATA_ORIENTATION].orient.rotation = (unsigned)rotation & 3u;
        md.vals[GPUJPEG_METADATA_ORIENTATION].orient.flip = flip != 0;
    }
    struct gpujpeg_parameters p;
    gj_geom g;
    if (host_adjusted(param, pi, &p, &g) != 0) { gj_exif_tags_destroy(tags); return 0; }
    uint8_t qraw[2][64];
    gj_quant_table_raw(0, p.quality, qraw[0]);
    gj_quant_table_raw(1, p.quality, qraw[1]);
    uint8_t hdr[GJ_MAIN_HEADER_CAP];
    const size_t n = gj_write_main_header(hdr, sizeof hdr, &g, &p, (enum gpujpeg_header_type)header_type, (const uint8_t(*)[64])qraw, &md, tags);
    gj_exif_tags_destroy(tags);
    struct gj_scan_headers sh;
    memset(&sh, 0, sizeof sh);
    if (gj_write_scan_headers(&sh, &g, &p) != 0) return 0;
    size_t total = 0;
    if (n <= sizeof hdr && n + sh.size <= capacity) {
        memcpy(dst, hdr, n);
        memcpy(dst + n, sh.bytes, sh.size);
        total = n + sh.size;
        if (main_header_size) *main_header_size = n;
    }
    free(sh.bytes);
    return total;
}

int gpujpeg_amd_host_geometry(const struct gpujpeg_parameters* param, const struct gpujpeg_image_parameters* pi, int out[20])
{
    struct gpujpeg_parameters p;
    gj_geom g;
    if (host_adjusted(param, pi, &p, &g) != 0) return -1;
    memset(out, 0, 20 * sizeof(int));
    out[0] = g.segment_count; out[1] = g.block_count; out[2] = p.restart_interval; out[3] = g.blocks_per_mcu;
    for (int c = 0; c < g.comp_count; c++) {
        out[4 + 4 * c] = g.comp[c].data_width; out[5 + 4 * c] = g.comp[c].data_height;
        out[6 + 4 * c] = g.interleaved ? g.segment_count : g.comp[c].segment_count; out[7 + 4 * c] = g.comp[c].type;
    }
    return 0;
}
