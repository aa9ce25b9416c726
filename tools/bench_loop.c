/*
 * bench_loop -- the inner loop of bench.py in C: N frames through gpujpeg_encoder_encode / gpujpeg_decoder_decode of one pipeline,
 * public API only (what a C caller of the reference writes, cf. README.md:178-260 and test/misc/mt_encode.c:12-45). bench.py's
 * launch threads call it once per timed region, so that no interpreter work (and no GIL hand-over between the four threads) sits
 * between two API calls; the per-frame Python loop remains for the short regions that read per-kernel times after every call.
 */
#define _GNU_SOURCE
#include <libgpujpeg/gpujpeg.h>
#include <stdint.h>
#include <stddef.h>
#include <time.h>

struct gj_bench_lane {
    struct gpujpeg_encoder* enc;
    struct gpujpeg_decoder* dec;
    struct gpujpeg_parameters* param;
    struct gpujpeg_image_parameters* param_image;
    uint8_t* const* images; /* image_count frames, walked round-robin */
    int image_count;
    int images_on_device; /* GPUJPEG_ENCODER_INPUT_GPU_IMAGE instead of _IMAGE */
    uint8_t* out;         /* decoder's custom buffer */
    int out_on_device;
    uint8_t* const* outs; /* out_count custom buffers walked round-robin (NULL: `out` for every frame) */
    int out_count;
};

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* mode 0: encode then decode every frame; 1: encode only; 2: decode only (of the stream passed in *jpeg / *size).
 * On return *jpeg / *size are the last stream, seconds[0] / [1] the time spent inside the encoder / decoder calls,
 * *bytes the sum of the stream sizes. Returns 0, or the failing call's code. */
__attribute__((visibility("default"))) int gj_bench_run(const struct gj_bench_lane* ln, int frames, int mode, uint8_t** jpeg, size_t* size,
                                                         double* seconds, size_t* bytes)
{
    struct gpujpeg_encoder_input in;
    struct gpujpeg_decoder_output out;
    double te = 0.0, td = 0.0;
    size_t total = 0;
    for (int f = 0; f < frames; f++) {
        const double a = now();
        if (mode != 2) {
            uint8_t* img = ln->images[f % ln->image_count];
            if (ln->images_on_device) gpujpeg_encoder_input_set_gpu_image(&in, img);
            else gpujpeg_encoder_input_set_image(&in, img);
            const int rc = gpujpeg_encoder_encode(ln->enc, ln->param, ln->param_image, &in, jpeg, size);
            if (rc != 0) return rc;
        }
        const double b = now();
        if (mode != 1) {
            uint8_t* dst = ln->outs != NULL && ln->out_count > 0 ? ln->outs[f % ln->out_count] : ln->out;
            if (ln->out_on_device) gpujpeg_decoder_output_set_custom_cuda(&out, dst);
            else gpujpeg_decoder_output_set_custom(&out, dst);
            const int rc = gpujpeg_decoder_decode(ln->dec, *jpeg, *size, &out);
            if (rc != 0) return rc;
        }
        const double c = now();
        te += b - a;
        td += c - b;
        total += *size;
    }
    seconds[0] = te;
    seconds[1] = td;
    *bytes = total;
    return 0;
}
