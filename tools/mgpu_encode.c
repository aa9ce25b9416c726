/*
 * mgpu_encode -- frame-sharded encoding (and optionally decoding) over the GPUs of one node through the libgpujpeg C API, the way
 * SURVEY.md 8(e) describes the deployment: no collective, one host thread per coder, every coder bound to one device and one
 * stream, frames pre-staged in PINNED host memory, C coders per device so that the upload of one frame overlaps the kernels of
 * another (PCIe, not xGMI, is the shared resource). Mirrors test/misc/mt_encode.c:12-45 of the reference (one encoder + stream
 * per thread) and extends it over devices.
 *
 *   mgpu_encode <frames_total> <width> <height> [devices=all] [coders_per_device=2] [decode=0] [pin=1] [batch=0]
 *
 * batch: B > 0 = every coder hands B frames at a time to gpujpeg_amd_encoder_encode_batch (and the B streams to
 * gpujpeg_amd_decoder_decode_batch): one set of kernel launches per B frames instead of per frame (include/gpujpeg_amd_ext.h; B <= 16
 * here, the distinct frames lie back to back in one pinned allocation). 0 = one libgpujpeg call per frame, the reference's API.
 *
 * pin: every coder thread is bound to one core of the NUMA node its GPU hangs off (/sys/bus/pci/devices/<bdf>/local_cpulist), the
 * devices of one node taking disjoint cores -- launch latency and the pinned staging traffic stay on the near socket. The frames are
 * allocated before the threads exist, portable pinned memory, first touched by the main thread: on a two-socket node pass
 * `numactl --interleave=all` for them.
 *
 * Thread t = (device d, coder c) takes frames t, t + T, t + 2T, ... (static round-robin like gpujpeg_amd/sharding.py). Host buffers
 * on both sides: this is the full-API figure (PCIe included), the one a drop-in caller sees. Prints one JSON line.
 */
#define _GNU_SOURCE
#include <hip/hip_runtime_api.h>
#include <libgpujpeg/gpujpeg.h>
#include <gpujpeg_amd_ext.h>
#include <ctype.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define DISTINCT 16 /* distinct synthetic frames, reused round-robin (one pinned allocation, back to back) */

struct shared {
    int frames_total, width, height, threads, decode, batch;
    uint8_t* frame[DISTINCT]; /* pinned */
    pthread_barrier_t start;
};

struct worker {
    struct shared* sh;
    int index, device;
    int cpu; /* core this thread is bound to, -1 = not bound */
    pthread_t tid;
    int rc;
    long frames;
    size_t jpeg_bytes;
    uint64_t fdig[DISTINCT]; /* FNV-1a (sampled bytes + size) of the stream of distinct frame k as this thread coded it */
    int fseen[DISTINCT];
    int self_mismatch;       /* the same frame gave two different streams on this coder */
    double seconds;
};

/* cores next to a device: the kernel's list format ("0-31,64-95") of /sys/bus/pci/devices/<bdf>/local_cpulist */
static int near_cpus(int device, int* cpus, int cap, char* key, size_t key_cap)
{
    char bdf[32] = {0}, path[128];
    key[0] = '\0';
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) return 0;
    for (char* c = bdf; *c; c++) *c = (char)tolower((unsigned char)*c);
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bdf);
    FILE* f = fopen(path, "r");
    if (!f) return 0;
    char line[1024] = {0};
    if (!fgets(line, sizeof line, f)) line[0] = '\0';
    fclose(f);
    snprintf(key, key_cap, "%s", line);
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return 0;
    int n = 0;
    for (char* q = line; *q && n < cap;) {
        char* end;
        const long lo = strtol(q, &end, 10);
        if (end == q) break;
        long hi = lo;
        if (*end == '-') hi = strtol(end + 1, &end, 10);
        for (long c = lo; c <= hi && n < cap; c++)
            if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET((int)c, &allowed)) cpus[n++] = (int)c;
        q = *end == ',' ? end + 1 : end;
        if (*end != ',') break;
    }
    return n;
}

/* core for coder `slot` of `device`: the near cores are split evenly between the devices that share them */
static int pick_cpu(int device, int ndev, int slot, int per_dev)
{
    enum { CAP = 1024 };
    static int cpus[CAP];
    char key[1024], other[1024];
    const int n = near_cpus(device, cpus, CAP, key, sizeof key);
    if (n == 0) return -1;
    int rank_on_node = 0, on_node = 0, scratch[8];
    for (int d = 0; d < ndev; d++) {
        near_cpus(d, scratch, 8, other, sizeof other);
        if (strcmp(other, key) == 0) {
            if (d < device) rank_on_node++;
            on_node++;
        }
    }
    int per = n / (on_node > 0 ? on_node : 1);
    if (per < 1) per = 1;
    (void)per_dev;
    return cpus[(rank_on_node * per + slot % per) % n];
}

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static void fill_frame(uint8_t* p, int w, int h, unsigned seed)
{
    unsigned s = 12345u + seed; /* smooth structure + texture + a little LCG noise: compresses like a photograph */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            s = s * 1664525u + 1013904223u;
            const int n = (int)((s >> 24) & 7) - 3;
            const int a = (x * 255 / w + (int)seed * 9) & 255, b = y * 255 / h, c = ((x ^ y) >> 2) & 63;
            uint8_t* q = p + ((size_t)y * w + x) * 3;
            q[0] = (uint8_t)((a + c / 4 + n) & 255);
            q[1] = (uint8_t)((b + c / 8 + n < 0 ? 0 : (b + c / 8 + n > 255 ? 255 : b + c / 8 + n)));
            q[2] = (uint8_t)(((a + b) / 2 + n) & 255);
        }
}

static void* run(void* arg)
{
    struct worker* wk = arg;
    struct shared* sh = wk->sh;
    wk->rc = 1;
    if (wk->cpu >= 0) {
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(wk->cpu, &one);
        if (pthread_setaffinity_np(pthread_self(), sizeof one, &one) != 0) wk->cpu = -1;
    }
    if (gpujpeg_init_device(wk->device, 0) != 0) return NULL; /* binds this thread to its device (src/gpujpeg_common.c:215) */
    hipStream_t stream;
    if (hipStreamCreate(&stream) != hipSuccess) return NULL;
    struct gpujpeg_encoder* enc = gpujpeg_encoder_create((cudaStream_t)stream);
    struct gpujpeg_decoder* dec = sh->decode ? gpujpeg_decoder_create((cudaStream_t)stream) : NULL;
    if (!enc || (sh->decode && !dec)) return NULL;
    gpujpeg_encoder_set_option(enc, "enc_opt_out", "enc_out_val_pinned");
    struct gpujpeg_parameters param;
    gpujpeg_set_default_parameters(&param);
    param.restart_interval = RESTART_AUTO;
    param.verbose = GPUJPEG_LL_QUIET;
    struct gpujpeg_image_parameters pi;
    gpujpeg_image_set_default_parameters(&pi);
    pi.width = sh->width;
    pi.height = sh->height;
    uint8_t* out = NULL;
    const size_t raw = (size_t)sh->width * sh->height * 3;
    if (sh->decode && hipHostMalloc((void**)&out, raw, hipHostMallocDefault) != hipSuccess) return NULL;
    /* one untimed call: allocations, tables */
    struct gpujpeg_encoder_input in;
    gpujpeg_encoder_input_set_image(&in, sh->frame[0]);
    uint8_t* jpeg = NULL;
    size_t size = 0;
    if (gpujpeg_encoder_encode(enc, &param, &pi, &in, &jpeg, &size) != 0) return NULL;
    uint8_t* outs = NULL; /* batch mode: room for the decoded frames of one batch */
    if (sh->batch > 0 && sh->decode && hipHostMalloc((void**)&outs, raw * (size_t)sh->batch, hipHostMallocDefault) != hipSuccess) return NULL;
    pthread_barrier_wait(&sh->start);
    const double t0 = now();
    int it = 0;
    if (sh->batch > 0) { /* this thread's share of the frames, B at a time: frames 0 .. B - 1 of the distinct set */
        long mine = 0;
        for (int f = wk->index; f < sh->frames_total; f += sh->threads) mine++;
        while (mine > 0) {
            const int n = mine < sh->batch ? (int)mine : sh->batch;
            uint8_t* js[DISTINCT];
            size_t sz[DISTINCT];
            if (gpujpeg_amd_encoder_encode_batch(enc, &param, &pi, sh->frame[0], raw, n, js, sz) != 0) return NULL;
            for (int k = 0; k < n; k++) {
                uint64_t dg = 1469598103934665603ull;
                for (size_t i = 0; i < sz[k]; i += 97) dg = (dg ^ js[k][i]) * 1099511628211ull;
                dg = (dg ^ (uint64_t)sz[k]) * 1099511628211ull;
                if (wk->fseen[k] && wk->fdig[k] != dg) wk->self_mismatch = 1;
                wk->fdig[k] = dg;
                wk->fseen[k] = 1;
                wk->jpeg_bytes += sz[k];
            }
            if (sh->decode) {
                struct gpujpeg_image_parameters opi;
                if (gpujpeg_amd_decoder_decode_batch(dec, js[0], n > 1 ? (size_t)(js[1] - js[0]) : ((sz[0] + 79) & ~(size_t)15), sz, n, outs, raw, &opi) != 0) return NULL;
            }
            wk->frames += n;
            mine -= n;
        }
    } else
    for (int f = wk->index; f < sh->frames_total; f += sh->threads, it++) {
        const int k = (wk->index + 3 * it) % DISTINCT; /* (3 is coprime to DISTINCT: every thread walks through all the distinct frames) */
        gpujpeg_encoder_input_set_image(&in, sh->frame[k]);
        if (gpujpeg_encoder_encode(enc, &param, &pi, &in, &jpeg, &size) != 0) return NULL;
        uint64_t dg = 1469598103934665603ull;
        for (size_t i = 0; i < size; i += 97) dg = (dg ^ jpeg[i]) * 1099511628211ull; /* sampled: the digest is not the timed work */
        dg = (dg ^ (uint64_t)size) * 1099511628211ull;
        if (wk->fseen[k] && wk->fdig[k] != dg) wk->self_mismatch = 1;
        wk->fdig[k] = dg;
        wk->fseen[k] = 1;
        if (sh->decode) {
            struct gpujpeg_decoder_output o;
            gpujpeg_decoder_output_set_custom(&o, out);
            if (gpujpeg_decoder_decode(dec, jpeg, size, &o) != 0) return NULL;
        }
        wk->frames++;
        wk->jpeg_bytes += size;
    }
    wk->seconds = now() - t0;
    gpujpeg_encoder_destroy(enc);
    if (dec) gpujpeg_decoder_destroy(dec);
    if (out) (void)hipHostFree(out);
    if (outs) (void)hipHostFree(outs);
    (void)hipStreamDestroy(stream);
    wk->rc = 0;
    return NULL;
}

int main(int argc, char** argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: %s <frames_total> <width> <height> [devices=all] [coders_per_device=2] [decode=0] [pin=1] [batch=0]\n", argv[0]);
        return 2;
    }
    struct shared sh;
    memset(&sh, 0, sizeof sh);
    sh.frames_total = atoi(argv[1]);
    sh.width = atoi(argv[2]);
    sh.height = atoi(argv[3]);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { fprintf(stderr, "no device\n"); return 1; }
    const int devices = argc > 4 && atoi(argv[4]) > 0 ? atoi(argv[4]) : ndev;
    const int per_dev = argc > 5 && atoi(argv[5]) > 0 ? atoi(argv[5]) : 2;
    sh.decode = argc > 6 ? atoi(argv[6]) : 0;
    const int pin = argc > 7 ? atoi(argv[7]) : 1;
    sh.batch = argc > 8 ? atoi(argv[8]) : 0;
    if (sh.batch < 0 || sh.batch > DISTINCT) { fprintf(stderr, "batch must be 0 .. %d\n", DISTINCT); return 2; }
    sh.threads = devices * per_dev;
    const size_t raw = (size_t)sh.width * sh.height * 3;
    uint8_t* all_frames = NULL;
    if (hipHostMalloc((void**)&all_frames, raw * DISTINCT, hipHostMallocPortable) != hipSuccess) { fprintf(stderr, "pinned allocation failed\n"); return 1; }
    for (int k = 0; k < DISTINCT; k++) {
        sh.frame[k] = all_frames + (size_t)k * raw;
        fill_frame(sh.frame[k], sh.width, sh.height, (unsigned)k);
    }
    pthread_barrier_init(&sh.start, NULL, (unsigned)sh.threads + 1);
    struct worker* wk = calloc((size_t)sh.threads, sizeof *wk);
    int coders_on[64] = {0};
    for (int t = 0; t < sh.threads; t++) {
        wk[t].sh = &sh;
        wk[t].index = t;
        wk[t].device = (t % devices) % ndev; /* (more requested devices than present: several threads share one, the 1-GPU proof) */
        wk[t].cpu = pin ? pick_cpu(wk[t].device, ndev, coders_on[wk[t].device % 64]++, per_dev) : -1;
        pthread_create(&wk[t].tid, NULL, run, &wk[t]);
    }
    pthread_barrier_wait(&sh.start);
    const double t0 = now();
    long frames = 0;
    size_t bytes = 0;
    int rc = 0;
    for (int t = 0; t < sh.threads; t++) {
        pthread_join(wk[t].tid, NULL);
        rc |= wk[t].rc;
        frames += wk[t].frames;
        bytes += wk[t].jpeg_bytes;
    }
    const double dt = now() - t0;
    /* equal frames must give equal streams on every coder and device: every distinct frame's digest is compared between all the
     * threads that coded it (and inside a thread between its repetitions); `compared` counts the cross-thread comparisons made */
    int consistent = 1, pinned = 0;
    long compared = 0;
    for (int a = 0; a < sh.threads; a++) pinned += wk[a].cpu >= 0;
    for (int a = 0; a < sh.threads; a++) {
        if (wk[a].self_mismatch) consistent = 0;
        for (int b = a + 1; b < sh.threads; b++)
            for (int k = 0; k < DISTINCT; k++)
                if (wk[a].fseen[k] && wk[b].fseen[k]) {
                    compared++;
                    if (wk[a].fdig[k] != wk[b].fdig[k]) consistent = 0;
                }
    }
    printf("{\"tool\": \"mgpu_encode\", \"ok\": %s, \"frames\": %ld, \"width\": %d, \"height\": %d, \"devices\": %d, \"devices_present\": %d, "
           "\"coders_per_device\": %d, \"decode\": %d, \"batch\": %d, \"seconds\": %.4f, \"frames_s\": %.2f, \"mpix_s\": %.1f, \"jpeg_bytes\": %zu, "
           "\"streams_consistent\": %s, \"digest_comparisons\": %ld, \"threads_pinned\": %d, \"first_thread_cpu\": %d, \"io\": \"pinned host buffers in and out (PCIe included)\"}\n",
           rc == 0 && frames == sh.frames_total ? "true" : "false", frames, sh.width, sh.height, devices, ndev, per_dev, sh.decode, sh.batch, dt,
           (double)frames / dt, (double)frames * sh.width * sh.height / dt / 1e6, bytes, consistent ? "true" : "false", compared, pinned, wk[0].cpu);
    (void)hipHostFree(all_frames);
    free(wk);
    return rc == 0 && frames == sh.frames_total && consistent ? 0 : 1;
}
