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
static size_t g_lane_min_bytes = (size_t)1 << 20; // (GJ_COPY_LANES=<MiB> moves it, 0 turns the lanes off)
static std::mutex g_lane_mutex;
static hipStream_t g_lane[64][2];
static int g_lanes_enabled = -1;
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
    if (g_lanes_enabled < 0) {
        const char* e = getenv("GJ_COPY_LANES");
        g_lanes_enabled = !(e && e[0] == '0' && e[1] == 0);
        if (e && atoi(e) > 0) g_lane_min_bytes = (size_t)atoi(e) << 20;
    }
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

// the developer switches, read once per coder (gj_hip.h)
extern "C" void gj_hip_tuning_from_env(gj_tuning* t)
{
    memset(t, 0, sizeof *t);
    const char* e;
    t->no_fused = getenv("GPUJPEG_NO_FUSED") != nullptr;
    t->host_scan = getenv("GPUJPEG_HOST_SCAN") != nullptr;
    t->enc_no_whole422 = getenv("GJ_ENC_NO_WHOLE422") != nullptr;
    t->dec_tokens = -1;
    if ((e = getenv("GJ_DEC_TOKENS")) && e[0] == '1') t->dec_tokens = 1;
    if (getenv("GJ_DEC_NO_TOKENS")) t->dec_tokens = 0;
    t->dec_serial = (e = getenv("GJ_DEC_ENTROPY")) && e[0] == 's';
    t->dec_batch = (e = getenv("GJ_DEC_G")) ? atoi(e) : 0;
    t->dec_sub = (e = getenv("GJ_DEC_SUB")) ? atoi(e) : 0;
    t->dec_no_spec = getenv("GJ_DEC_NO_SPEC") != nullptr;
    if ((e = getenv("GJ_DEC_SEQ"))) t->dec_seq = e[0] == '1' ? 1 : 2;
    t->debug_sync = (e = getenv("GJ_DEC_DEBUG_SYNC")) && e[0] == '1';
    t->dec_tok_nocoop = (e = getenv("GJ_DEC_TOK_NOCOOP")) && e[0] == '1';
    t->scan_shape = (e = getenv("GJ_SCAN_SHAPE")) ? atoi(e) : 0;
    t->enc_by_blocks = (e = getenv("GJ_ENC_BLOCKS")) ? atoi(e) : 0;
    t->host_timing = (e = getenv("GJ_HOST_TIMING")) && e[0] == '1';
    t->dec_balance = (e = getenv("GJ_DEC_BALANCE")) && e[0] == '1';
    t->enc_split = (e = getenv("GJ_ENC_SPLIT")) ? atoi(e) : -1;
    t->enc_tail = (e = getenv("GJ_ENC_TAIL")) ? atoi(e) : -1;
    t->dec_fill = (e = getenv("GJ_DEC_FILL")) ? atoi(e) : 0;
}
