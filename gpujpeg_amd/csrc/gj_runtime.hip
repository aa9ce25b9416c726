// gj_runtime.hip -- HIP runtime subset behind the C-ABI of include/gj_hip.h.
// Replaces the ~25 CUDA runtime entry points the reference's host C calls directly
// (cudaMalloc, cudaMallocHost, cudaMemcpyAsync, cudaEvent*, ... enumerated over /root/reference/src/*.c).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "gj_hip.h"

static thread_local hipError_t g_last = hipSuccess;

static int chk(hipError_t e)
{
    if (e != hipSuccess) {
        g_last = e;
        return -1;
    }
    return 0;
}

extern "C" {

// the launchers (gj_hip_encode / gj_hip_decode and what they call) check every runtime call where they make it: the first failure of a thread is kept
// for gj_hip_last_error, gj_hip_noted says whether there has been one since gj_hip_note_reset
int gj_hip_note(int hip_error) { return chk((hipError_t)hip_error); }
void gj_hip_note_reset(void) { g_last = hipSuccess; }
int gj_hip_noted(void) { return g_last != hipSuccess; }

int gj_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gj_hip_get_device(int* device) { return chk(hipGetDevice(device)); }
int gj_hip_set_device(int device) { return chk(hipSetDevice(device)); }
static void gj_lanes_forget(int dev);
int gj_hip_device_reset(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) gj_lanes_forget(dev); // (the reset destroys the device's streams: the copy lanes are created again on demand)
    return chk(hipDeviceReset());
}

int gj_hip_device_props(int device, char name[256], int* major, int* minor, size_t* global_mem, size_t* shared_mem,
                        int* regs_per_block, int* cu_count)
{
    hipDeviceProp_t p;
    if (chk(hipGetDeviceProperties(&p, device))) return -1;
    std::snprintf(name, 256, "%s", p.name);
    *major = p.major;
    *minor = p.minor;
    *global_mem = p.totalGlobalMem;
    *shared_mem = p.sharedMemPerBlock;
    *regs_per_block = p.regsPerBlock;
    *cu_count = p.multiProcessorCount;
    return 0;
}

// compute units of the CURRENT device (cached per device): the batch plans and kernel choices that depend on how many workgroups the
// device holds at once ask here instead of assuming the MI355X's 256 (other parts, partitioned modes: CPX / NPS)
int gj_hip_cu_count(void)
{
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        hipDeviceProp_t p;
        const int n = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
        cached[dev] = n;
    }
    return cached[dev];
}

int gj_hip_runtime_version(int* driver, int* runtime)
{
    if (chk(hipDriverGetVersion(driver))) return -1;
    return chk(hipRuntimeGetVersion(runtime));
}

const char* gj_hip_last_error(void)
{
    hipError_t e = g_last != hipSuccess ? g_last : hipGetLastError();
    g_last = hipSuccess;
    return hipGetErrorString(e);
}

void* gj_hip_malloc(size_t size)
{
    void* p = nullptr;
    if (chk(hipMalloc(&p, size ? size : 1))) return nullptr;
    return p;
}
void gj_hip_free(void* p)
{
    if (p) (void)hipFree(p);
}
void* gj_hip_host_alloc(size_t size)
{
    void* p = nullptr;
    if (chk(hipHostMalloc(&p, size ? size : 1, hipHostMallocDefault))) return nullptr;
    return p;
}
void gj_hip_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

int gj_hip_memcpy_h2d(void* d, const void* s, size_t n, gj_stream_t st) { return chk(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, (hipStream_t)st)); }
int gj_hip_memcpy_d2h(void* d, const void* s, size_t n, gj_stream_t st) { return chk(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, (hipStream_t)st)); }
int gj_hip_memcpy_d2d(void* d, const void* s, size_t n, gj_stream_t st) { return chk(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, (hipStream_t)st)); }

// ---- copy lanes: whole images cross the host link on ONE stream per direction and device, whatever coder they belong to, and the calling thread
// waits for them -- no stream waits for another stream's event.
// Measured on the MI355X box (tools/ubench/pcie_duplex.py, tools/ubench/host_link_patterns.hip, profiles/r5_07): a direction alone carries 55-57 GB/s.
// N synchronous coders that upload 99.5 MB, run a kernel and download 99.5 MB per call move
//   28-33 GB/s each way   copying on their own streams (what the library did before: the link behaves as if it were half duplex),
//   20-39 GB/s            through one upload and one download stream tied into the coder's stream with hipStreamWaitEvent both ways,
//   15-28 GB/s            with an upload, a compute and a download stream per coder and events between them,
//   46-47 GB/s            through one upload and one download stream when the THREAD waits for its copy (hipEventSynchronize) -- from two coders up.
// The calls of the API are synchronous anyway, so the last form costs nothing but two host wake-ups per call. Copies below g_lane_min_bytes
// (tables, headers, the streams of small frames), calls without an event and GJ_COPY_LANES=0 stay asynchronous on the caller's stream.
static size_t g_lane_min_bytes = (size_t)1 << 20; // (the developer setting GJ_COPY_LANES=<MiB> moves it, 0 turns the lanes off: gj_hip_tuning_setting)
static std::mutex g_lane_mutex;
static hipStream_t g_lane[64][2];
static int g_lanes_enabled = 1;
static void gj_lanes_forget(int dev)
{
    std::lock_guard<std::mutex> lk(g_lane_mutex);
    if (dev >= 0 && dev < 64) g_lane[dev][0] = g_lane[dev][1] = nullptr;
}
static hipStream_t gj_lane(int dir, size_t n)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_lane_mutex);
    if (!g_lanes_enabled || n < g_lane_min_bytes) return nullptr;
    if (!g_lane[dev][dir] && hipStreamCreateWithFlags(&g_lane[dev][dir], hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        g_lane[dev][dir] = nullptr;
    }
    return g_lane[dev][dir];
}
// upload: what `st` holds does not touch dst (the callers' staging buffers are idle at this point of a call); complete on return when it went
// through the lane, asynchronous on st otherwise -- either way ordered in front of what the caller launches on st next
int gj_hip_upload(void* d, const void* s, size_t n, gj_stream_t st, gj_event_t done)
{
    hipStream_t lane = done ? gj_lane(0, n) : nullptr;
    if (!lane) return chk(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, (hipStream_t)st));
    if (chk(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, lane)) || chk(hipEventRecord((hipEvent_t)done, lane))) return -1;
    return chk(hipEventSynchronize((hipEvent_t)done));
}
// download of what the work on `st` produced: through the lane the thread waits for st first and for the copy afterwards (complete on return);
// otherwise the copy is asynchronous on st and the caller synchronises as before
int gj_hip_download(void* d, const void* s, size_t n, gj_stream_t st, gj_event_t done)
{
    hipStream_t lane = done ? gj_lane(1, n) : nullptr;
    if (!lane) return chk(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, (hipStream_t)st));
    if (chk(hipStreamSynchronize((hipStream_t)st))) return -1;
    if (chk(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, lane)) || chk(hipEventRecord((hipEvent_t)done, lane))) return -1;
    return chk(hipEventSynchronize((hipEvent_t)done));
}
// the same for several copies of one direction behind each other (the frames of a batch call): _begin says which stream they go on -- the lane, after
// the work on st when they are downloads -- and _end makes the thread wait for them when that was a lane
gj_stream_t gj_hip_lane_begin(int download, size_t bytes_each, gj_stream_t st)
{
    hipStream_t lane = gj_lane(download ? 1 : 0, bytes_each);
    if (!lane) return st;
    if (download && chk(hipStreamSynchronize((hipStream_t)st))) return st;
    return (gj_stream_t)lane;
}
int gj_hip_lane_end(gj_stream_t lane, gj_stream_t st, gj_event_t done)
{
    if (lane == st) return 0;
    if (chk(hipEventRecord((hipEvent_t)done, (hipStream_t)lane))) return -1;
    return chk(hipEventSynchronize((hipEvent_t)done));
}
int gj_hip_memset(void* d, int v, size_t n, gj_stream_t st) { return chk(hipMemsetAsync(d, v, n, (hipStream_t)st)); }
int gj_hip_stream_sync(gj_stream_t st) { return chk(hipStreamSynchronize((hipStream_t)st)); }

int gj_hip_is_device_ptr(const void* p)
{
    hipPointerAttribute_t a;
    std::memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError(); // plain malloc'd memory reports an error: that is "host"
        return 0;
    }
    return a.type == hipMemoryTypeDevice ? 1 : 0;
}

gj_event_t gj_hip_event_create(void)
{
    hipEvent_t e = nullptr;
    if (chk(hipEventCreate(&e))) return nullptr;
    return (gj_event_t)e;
}
void gj_hip_event_destroy(gj_event_t e)
{
    if (e) (void)hipEventDestroy((hipEvent_t)e);
}
int gj_hip_event_record(gj_event_t e, gj_stream_t s) { return chk(hipEventRecord((hipEvent_t)e, (hipStream_t)s)); }
float gj_hip_event_elapsed_ms(gj_event_t a, gj_event_t b)
{
    float ms = 0.0f;
    if (hipEventSynchronize((hipEvent_t)b) != hipSuccess) return 0.0f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) return 0.0f;
    return ms;
}

} // extern "C"

// ---- The developer settings (gj_hip.h: gj_tuning). Process-wide values, set through gj_hip_tuning_setting (public: gpujpeg_amd_tuning,
// include/gpujpeg_amd_ext.h) and copied into a coder when it is created; the launchers see only the coder's copy. The RELEASE library never reads
// the environment: builds with -DGJ_TUNING_ENV -- the `trace` / `variant` targets of the Makefile and the CPU execution model of
// tests/hipemu, i.e. what the measurement tools load -- additionally take every setting whose name is in the environment when a coder is created.
static std::mutex g_tune_mutex;
static gj_tuning g_tune = {/*no_fused*/ 0, /*enc_split*/ -1, /*enc_tail*/ -1, 0, 0, 0, 0, /*dec_tokens*/ -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
static const char* const g_tune_names[] = {"GPUJPEG_NO_FUSED", "GPUJPEG_HOST_SCAN", "GJ_ENC_NO_WHOLE422", "GJ_DEC_TOKENS", "GJ_DEC_NO_TOKENS", "GJ_DEC_ENTROPY", "GJ_DEC_G",
                                           "GJ_DEC_SUB", "GJ_DEC_NO_SPEC", "GJ_DEC_SEQ", "GJ_DEC_DEBUG_SYNC", "GJ_DEC_TOK_NOCOOP", "GJ_SCAN_SHAPE", "GJ_ENC_BLOCKS",
                                           "GJ_HOST_TIMING", "GJ_DEC_BALANCE", "GJ_ENC_SPLIT", "GJ_ENC_TAIL", "GJ_DEC_FILL", "GJ_COPY_LANES", nullptr};
// one setting: `name` as in INTEGRATION.md's table, `v` its value (may be empty: "the name is present"); false for an unknown name
static bool gj_tune_apply(gj_tuning* t, const char* name, const char* v)
{
    const auto is = [&](const char* n) { return std::strcmp(name, n) == 0; };
    const int num = std::atoi(v);
    if (is("GPUJPEG_NO_FUSED")) t->no_fused = 1;
    else if (is("GPUJPEG_HOST_SCAN")) t->host_scan = 1;
    else if (is("GJ_ENC_NO_WHOLE422")) t->enc_no_whole422 = 1;
    else if (is("GJ_DEC_TOKENS")) { if (v[0] == '1') t->dec_tokens = 1; }
    else if (is("GJ_DEC_NO_TOKENS")) t->dec_tokens = 0;
    else if (is("GJ_DEC_ENTROPY")) t->dec_serial = v[0] == 's';
    else if (is("GJ_DEC_G")) t->dec_batch = num;
    else if (is("GJ_DEC_SUB")) t->dec_sub = num;
    else if (is("GJ_DEC_NO_SPEC")) t->dec_no_spec = 1;
    else if (is("GJ_DEC_SEQ")) t->dec_seq = v[0] == '1' ? 1 : 2;
    else if (is("GJ_DEC_DEBUG_SYNC")) t->debug_sync = v[0] == '1';
    else if (is("GJ_DEC_TOK_NOCOOP")) t->dec_tok_nocoop = v[0] == '1';
    else if (is("GJ_SCAN_SHAPE")) t->scan_shape = num;
    else if (is("GJ_ENC_BLOCKS")) t->enc_by_blocks = num;
    else if (is("GJ_HOST_TIMING")) t->host_timing = v[0] == '1';
    else if (is("GJ_DEC_BALANCE")) t->dec_balance = v[0] == '1';
    else if (is("GJ_ENC_SPLIT")) t->enc_split = num;
    else if (is("GJ_ENC_TAIL")) t->enc_tail = num;
    else if (is("GJ_DEC_FILL")) t->dec_fill = num;
    else if (is("GJ_COPY_LANES")) { // (process-wide, not per coder: 0 turns the copy lanes off, <MiB> moves the size from which a copy takes them)
        std::lock_guard<std::mutex> lk(g_lane_mutex);
        g_lanes_enabled = !(v[0] == '0' && v[1] == 0);
        g_lane_min_bytes = num > 0 ? (size_t)num << 20 : (size_t)1 << 20;
    } else return false;
    return true;
}
extern "C" int gj_hip_tuning_setting(const char* setting)
{
    std::lock_guard<std::mutex> lk(g_tune_mutex);
    if (setting == nullptr) { // back to the defaults
        g_tune = gj_tuning{0, -1, -1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        std::lock_guard<std::mutex> lk2(g_lane_mutex);
        g_lanes_enabled = 1;
        g_lane_min_bytes = (size_t)1 << 20;
        return 0;
    }
    char name[40];
    const char* eq = std::strchr(setting, '=');
    const size_t n = eq ? (size_t)(eq - setting) : std::strlen(setting);
    if (n == 0 || n >= sizeof name) return -1;
    std::memcpy(name, setting, n);
    name[n] = 0;
    return gj_tune_apply(&g_tune, name, eq ? eq + 1 : "") ? 0 : -1;
}
extern "C" const char* const* gj_hip_tuning_names(int* count)
{
    *count = (int)(sizeof g_tune_names / sizeof g_tune_names[0]) - 1; // (the array ends with a null pointer)
    return g_tune_names;
}
extern "C" void gj_hip_tuning_defaults(gj_tuning* t)
{
    {
        std::lock_guard<std::mutex> lk(g_tune_mutex);
        *t = g_tune;
    }
#ifdef GJ_TUNING_ENV
    for (const char* n : g_tune_names)
        if (n != nullptr)
            if (const char* e = getenv(n)) gj_tune_apply(t, n, e);
#endif
}
