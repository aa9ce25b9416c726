/*
 * gpujpegtool -- command line front-end, drop-in for the reference CLI (src/main.c): same options,
 * same encode/decode file loop, same "-n" iteration and timing printouts (those come from the library
 * at verbosity >= 1). Uses only the public libgpujpeg API.
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "libgpujpeg/gpujpeg.h"

static void print_help(bool full)
{
    printf("gpujpegtool [options] input.rgb output.jpg [input2.rgb output2.jpg ...]\n"
           "   -h, --help             print help\n");
    if (!full) printf("   -H, --fullhelp         print all options\n");
    printf("   -v, --verbose          verbose output (multiply to increase verbosity - max 3) \n"
           "   -D, --device           set HIP device id (default 0)\n"
           "   -L, --device-list      list HIP devices\n\n"
           "   -s, --size             set input image size in pixels, e.g. 1920x1080\n"
           "   -f, --pixel-format     set input/output image pixel format, one of the\n"
           "                          following (example in parenthesis):\n");
    gpujpeg_print_pixel_formats();
    printf("\n   -c, --colorspace       set input/output image colorspace, e.g. rgb, ycbcr-jpeg (full\n"
           "                          range BT.601), ycbcr-bt601 (limited 601), ycbcr-bt709 (limited)\n\n"
           "   -q, --quality          set JPEG encoder quality level 0-100 (default 75)\n"
           "   -r, --restart          set JPEG encoder restart interval (default: chosen automatically)\n"
           "   -S, --subsampled[=<s>] set JPEG encoder chroma subsampling in J:a:b[:A] format (default 4:2:0)\n"
           "   -i  --interleaved      set JPEG encoder to use interleaved stream\n"
           "   -g  --segment-info     set JPEG encoder to use segment info in stream\n"
           "                          for fast decoding\n\n"
           "   -e, --encode           perform JPEG encoding\n"
           "   -d, --decode           perform JPEG decoding\n"
           "   -R, --component-range  show samples range for each component in image\n\n"
           "   -n  --iterate          perform encoding/decoding in specified number of\n"
           "                          iterations for each image\n"
           "   -I  --info             print JPEG file info\n"
           "   -a  --alpha            encode/decode alpha channel (otherwise stripped)\n"
           "   -N  --native           create native JPEG (Adobe RGB for RGB, SPIFF for Y709;\n"
           "                                              works also for decoding)\n"
           "   -V  --version          print GPUJPEG version\n");
    if (full)
        printf("   -b, --debug            debug helpers (reset GPU for leakcheck)\n"
               "   -O <key>=<value>|help  set encoder/decoder option, 'help' for list\n");
    printf("recognized raw input/output file extensions: rgb, yuv, pnm... (use`gpujpegtool exts` for the full list)\n");
}

static int print_image_info(const char* filename, int verbose)
{
    uint8_t* data = NULL;
    size_t size = 0;
    if (gpujpeg_image_load_from_file(filename, &data, &size) != 0) return 1;
    struct gpujpeg_image_info info;
    const int rc = gpujpeg_decoder_get_image_info2(data, size, &info, verbose, GPUJPEG_COUNT_SEG_COUNT_REQ);
    if (rc == 0) {
        printf("width: %d\nheight: %d\n", info.param_image.width, info.param_image.height);
        printf("component count: %d\n", info.param.comp_count);
        printf("color space: %s\n", gpujpeg_color_space_get_name(info.param.color_space_internal));
        printf("internal representation: %s (%s)\n", gpujpeg_pixel_format_get_name(info.param_image.pixel_format),
               gpujpeg_subsampling_get_name(info.param.comp_count, info.param.sampling_factor));
        printf("segment count: %d (DRI = %d)\n", info.segment_count, info.param.restart_interval);
        printf("JPEG header type: %s\n", gpujpeg_header_type_get_name(info.header_type));
        if (info.comment) printf("comment: %s\n", info.comment);
    }
    gpujpeg_image_destroy(data);
    return rc == 0 ? 0 : 1;
}

struct options {
    int device, iterate;
    bool keep_alpha, native, debug, encode, decode, range;
    gpujpeg_sampling_factor_t subsampling;
    char* enc_opts[32]; int enc_opt_count;
};

/* What the command line leaves open is taken from the raw file: the input when encoding, the (not yet existing) output when decoding
 * (src/main.c:251-291). The result of the probe is not checked: an unknown extension (/dev/zero, .XXX) simply contributes nothing. */
static bool adjust_params(struct gpujpeg_parameters* param, struct gpujpeg_image_parameters* pi, const char* raw_file, bool encode, const struct options* o)
{
    struct gpujpeg_image_parameters file_pi = gpujpeg_default_image_parameters();
    if (pi->width == 0 || pi->height == 0 || pi->pixel_format == GPUJPEG_PIXFMT_NONE || pi->color_space == GPUJPEG_NONE)
        (void)gpujpeg_image_get_properties(raw_file, &file_pi, encode);
    if (!pi->width) pi->width = file_pi.width;
    if (!pi->height) pi->height = file_pi.height;
    if (pi->color_space == GPUJPEG_NONE) pi->color_space = file_pi.color_space ? file_pi.color_space : GPUJPEG_RGB;
    if (pi->pixel_format == GPUJPEG_PIXFMT_NONE) {
        pi->pixel_format = file_pi.pixel_format;
        if (!encode && !o->keep_alpha && file_pi.pixel_format == GPUJPEG_PIXFMT_AUTODETECT) pi->pixel_format = GPUJPEG_PIXFMT_NO_ALPHA; /* alpha only on request */
    }
    if (o->keep_alpha && encode && pi->pixel_format == GPUJPEG_4444_U8_P0123) {
        gpujpeg_sampling_factor_t subs = GPUJPEG_SUBSAMPLING_4444;
        if (o->subsampling != GPUJPEG_SUBSAMPLING_UNKNOWN && (o->subsampling & 0xFF) == 0) subs = o->subsampling | o->subsampling >> 24; /* alpha sampled like Y */
        gpujpeg_parameters_chroma_subsampling(param, subs);
    }
    if (encode && (pi->width <= 0 || pi->height <= 0)) {
        fprintf(stderr, "Image dimensions must be set to nonzero values!\n");
        return false;
    }
    if (encode && pi->pixel_format == GPUJPEG_PIXFMT_NONE) {
        fprintf(stderr, "Pixel format must be set!\n");
        return false;
    }
    return true;
}

/* -b: an input that is not a regular file with content (test pattern, /dev/zero ...) is written out as input-<name>.XXX, where
 * XXX becomes the extension that fits the pixel format (src/main.c:309-350) */
static void dump_infile(const char* filename, const uint8_t* image, size_t size, const struct gpujpeg_image_parameters* pi)
{
    FILE* f = fopen(filename, "rb");
    long file_size = 0;
    if (f) {
        fseek(f, 0, SEEK_END);
        file_size = ftell(f);
        fclose(f);
    }
    if (file_size > 0) return;
    const char* base = strrchr(filename, '/');
    base = base ? base + 1 : filename;
    char name[256];
    snprintf(name, sizeof name - 4, "input-%s", base);
    char* dot = strrchr(name, '.');
    if (!dot) { dot = name + strlen(name); *dot = '.'; }
    strcpy(dot + 1, "XXX");
    if (gpujpeg_image_save_to_file(name, image, size, pi) == 0) printf("Input data saved to file %s.\n", name);
}

int main(int argc, char* argv[])
{
    struct gpujpeg_parameters param = gpujpeg_default_parameters();
    param.restart_interval = RESTART_AUTO;
    param.verbose = GPUJPEG_LL_STATUS;
    struct gpujpeg_image_parameters pi = gpujpeg_default_image_parameters();
    pi.color_space = GPUJPEG_NONE;
    pi.pixel_format = GPUJPEG_PIXFMT_NONE;
    struct options o = {.iterate = 1};
    bool restart_set = false;
    static const struct option longopts[] = {
        {"alpha", no_argument, 0, 'a'}, {"debug", no_argument, 0, 'b'}, {"help", no_argument, 0, 'h'}, {"fullhelp", no_argument, 0, 'H'},
        {"verbose", optional_argument, 0, 'v'}, {"device", required_argument, 0, 'D'}, {"device-list", no_argument, 0, 'L'},
        {"size", required_argument, 0, 's'}, {"pixel-format", required_argument, 0, 'f'}, {"colorspace", required_argument, 0, 'c'},
        {"quality", required_argument, 0, 'q'}, {"restart", required_argument, 0, 'r'}, {"segment-info", optional_argument, 0, 'g'},
        {"subsampled", optional_argument, 0, 'S'}, {"interleaved", optional_argument, 0, 'i'}, {"encode", no_argument, 0, 'e'},
        {"decode", no_argument, 0, 'd'}, {"component-range", no_argument, 0, 'R'}, {"iterate", required_argument, 0, 'n'},
        {"use-opengl", no_argument, 0, 'o'}, {"info", required_argument, 0, 'I'}, {"native", no_argument, 0, 'N'},
        {"version", no_argument, 0, 'V'}, {0}};
    int ch;
    while ((ch = getopt_long(argc, argv, "CD:HI:LNO:RS::Vabc:edf:ghin:oq:r:s:v", longopts, NULL)) != -1) {
        switch (ch) {
        case 'a': o.keep_alpha = true; break;
        case 'b': o.debug = true; break;
        case 'h': print_help(false); return 0;
        case 'H': print_help(true); return 0;
        case 'v': param.verbose += 1; if (optarg) param.verbose += (int)strlen(optarg); break;
        case 'D': o.device = atoi(optarg); break;
        case 'L': return gpujpeg_print_devices_info() < 0 ? 1 : 0;
        case 's': {
            char* x = strchr(optarg, 'x');
            if (!x) { fprintf(stderr, "Incorrect image size '%s'! Use a format 'WxH'.\n", optarg); return 1; }
            pi.width = atoi(optarg);
            pi.height = atoi(x + 1);
            break; }
        case 'f':
            pi.pixel_format = gpujpeg_pixel_format_by_name(optarg);
            if (pi.pixel_format == GPUJPEG_PIXFMT_NONE) { fprintf(stderr, "Unknown pixel format '%s'!\n", optarg); return 1; }
            break;
        case 'c':
            pi.color_space = gpujpeg_color_space_by_name(optarg);
            if (pi.color_space == GPUJPEG_NONE) { if (strcmp(optarg, "help") != 0) fprintf(stderr, "Unknown color space '%s'!\n", optarg); return 1; }
            break;
        case 'q': param.quality = atoi(optarg); if (param.quality < 0) param.quality = 0; if (param.quality > 100) param.quality = 100; break;
        case 'r': param.restart_interval = atoi(optarg); if (param.restart_interval < 0) param.restart_interval = 0; restart_set = true; break;
        case 'g': param.segment_info = optarg ? atoi(optarg) : 1; break;
        case 'S':
            o.subsampling = optarg ? gpujpeg_subsampling_from_name(optarg) : GPUJPEG_SUBSAMPLING_420;
            if (o.subsampling == GPUJPEG_SUBSAMPLING_UNKNOWN) { fprintf(stderr, "Unknown subsampling '%s'!\n", optarg ? optarg : ""); return 1; }
            break;
        case 'i': param.interleaved = optarg ? atoi(optarg) : 1; break;
        case 'e': o.encode = true; break;
        case 'd': o.decode = true; break;
        case 'R': o.range = true; break;
        case 'n': o.iterate = atoi(optarg); if (o.iterate < 1) o.iterate = 1; break;
        case 'o': fprintf(stderr, "OpenGL is not available in the MI355X build.\n"); return 1;
        case 'I': return print_image_info(optarg, param.verbose);
        case 'N': o.native = true; break;
        case 'V': printf("GPUJPEG version: %s (MI355X / HIP build)\n", gpujpeg_version_to_string(gpujpeg_version())); return 0;
        case 'O':
            if (strcmp(optarg, "help") == 0) { printf("Encoder options:\n"); gpujpeg_encoder_print_options(); printf("Decoder options:\n"); gpujpeg_decoder_print_options(); return 0; }
            if (o.enc_opt_count < 32) o.enc_opts[o.enc_opt_count++] = optarg;
            break;
        case 'C': fprintf(stderr, "Conversion is not supported (defunct in the reference too).\n"); return 1;
        default: print_help(false); return 1;
        }
    }
    (void)restart_set;
    argc -= optind;
    argv += optind;
    if (argc == 1 && strcmp(argv[0], "exts") == 0) { gpujpeg_image_get_file_format("help"); return 0; }
    if (argc == 0 || (argc % 2) != 0) {
        if (argc) fprintf(stderr, "Wrong number of file arguments: pairs of input/output files are expected.\n");
        print_help(false);
        return 1;
    }
    /* direction from the file names when neither -e nor -d is given (main.c:655-684) */
    if (!o.encode && !o.decode) {
        const enum gpujpeg_image_file_format in = gpujpeg_image_get_file_format(argv[0]), out = gpujpeg_image_get_file_format(argv[1]);
        if (out == GPUJPEG_IMAGE_FILE_JPEG && in != GPUJPEG_IMAGE_FILE_JPEG) o.encode = true;
        else if (in == GPUJPEG_IMAGE_FILE_JPEG) o.decode = true;
        else { fprintf(stderr, "Cannot determine the operation from file extensions, use -e or -d.\n"); return 1; }
    }
    if (gpujpeg_init_device(o.device, param.verbose >= GPUJPEG_LL_VERBOSE ? GPUJPEG_INIT_DEV_VERBOSE : 0) != 0) return 1;

    int rc = 0;
    if (o.encode) {
        struct gpujpeg_encoder* enc = gpujpeg_encoder_create(0);
        if (!enc) { fprintf(stderr, "Failed to create encoder!\n"); return 1; }
        for (int k = 0; k < o.enc_opt_count; k++) {
            char* eq = strchr(o.enc_opts[k], '=');
            if (!eq) { fprintf(stderr, "Option must be key=value\n"); return 1; }
            *eq = '\0';
            if (gpujpeg_encoder_set_option(enc, o.enc_opts[k], eq + 1) != 0) return 1;
        }
        for (int i = 0; i < argc; i += 2) {
            const char *in = argv[i], *out = argv[i + 1];
            struct gpujpeg_image_parameters fpi = pi;
            struct gpujpeg_parameters p = param;
            if (o.subsampling) gpujpeg_parameters_chroma_subsampling(&p, o.subsampling);
            if (!adjust_params(&p, &fpi, in, true, &o)) { rc = 1; continue; }
            if (fpi.pixel_format == GPUJPEG_PIXFMT_STD) fpi.pixel_format = GPUJPEG_444_U8_P012;
            if (o.native) p.color_space_internal = fpi.color_space; /* main.c:769-771 */
            uint8_t* image = NULL;
            size_t size = gpujpeg_image_calculate_size(&fpi);
            if (gpujpeg_image_load_from_file(in, &image, &size) != 0) { fprintf(stderr, "Failed to load image [%s]!\n", in); rc = 1; continue; }
            if (o.debug) dump_infile(in, image, size, &fpi);
            struct gpujpeg_encoder_input input = gpujpeg_encoder_input_image(image);
            uint8_t* jpeg = NULL;
            size_t jpeg_size = 0;
            int erc = 0;
            for (int it = 0; it < o.iterate && erc == 0; it++) {
                if (o.iterate > 1 && p.verbose >= GPUJPEG_LL_STATUS) printf("\nIteration #%d:\n", it + 1);
                erc = gpujpeg_encoder_encode(enc, &p, &fpi, &input, &jpeg, &jpeg_size);
            }
            if (erc != 0) { fprintf(stderr, "Failed to encode image [%s]!\n", in); rc = 1; }
            else if (gpujpeg_image_save_to_file(out, jpeg, jpeg_size, NULL) != 0) { fprintf(stderr, "Failed to save image [%s]!\n", out); rc = 1; }
            gpujpeg_image_destroy(image);
        }
        gpujpeg_encoder_destroy(enc);
    } else {
        struct gpujpeg_decoder_init_parameters ip = gpujpeg_decoder_default_init_parameters();
        ip.verbose = param.verbose;
        ip.perf_stats = param.verbose >= GPUJPEG_LL_STATUS;
        struct gpujpeg_decoder* dec = gpujpeg_decoder_create_with_params(&ip);
        if (!dec) { fprintf(stderr, "Failed to create decoder!\n"); return 1; }
        for (int k = 0; k < o.enc_opt_count; k++) {
            char* eq = strchr(o.enc_opts[k], '=');
            if (!eq) { fprintf(stderr, "Option must be key=value\n"); return 1; }
            *eq = '\0';
            if (gpujpeg_decoder_set_option(dec, o.enc_opts[k], eq + 1) != 0) return 1;
        }
        for (int i = 0; i < argc; i += 2) {
            const char* in = argv[i];
            char out[4096];
            snprintf(out, sizeof out, "%s", argv[i + 1]);
            struct gpujpeg_image_parameters fpi = pi;
            struct gpujpeg_parameters p = param;
            adjust_params(&p, &fpi, out, false, &o);
            if (o.native) fpi.color_space = GPUJPEG_NONE; /* main.c:906-908 */
            gpujpeg_decoder_set_output_format(dec, fpi.color_space, fpi.pixel_format);
            uint8_t* jpeg = NULL;
            size_t size = 0;
            if (gpujpeg_image_load_from_file(in, &jpeg, &size) != 0) { fprintf(stderr, "Failed to load image [%s]!\n", in); rc = 1; continue; }
            struct gpujpeg_decoder_output output;
            gpujpeg_decoder_output_set_default(&output);
            int drc = 0;
            for (int it = 0; it < o.iterate && drc == 0; it++) {
                if (o.iterate > 1 && param.verbose >= GPUJPEG_LL_STATUS) printf("\nIteration #%d:\n", it + 1);
                drc = gpujpeg_decoder_decode(dec, jpeg, size, &output);
            }
            if (drc != 0) { fprintf(stderr, "Failed to decode image [%s]!\n", in); rc = 1; }
            else if (gpujpeg_image_save_to_file(out, output.data, output.data_size, &output.param_image) != 0) { fprintf(stderr, "Failed to save image [%s]!\n", out); rc = 1; }
            else if (o.range) gpujpeg_image_range_info(out, output.param_image.width, output.param_image.height, output.param_image.pixel_format);
            gpujpeg_image_destroy(jpeg);
        }
        gpujpeg_decoder_destroy(dec);
    }
    if (o.debug) gpujpeg_device_reset();
    return rc;
}
