/*
 * TEST INFRASTRUCTURE ONLY -- the scheduler and runtime behind tests/hipemu/include/hip/hip_runtime.h (see there).
 * x86-64 System V only.
 */
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

thread_local hipemu_idx threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

extern "C" void hipemu_switch(void** save_sp, void* new_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {
extern const char* g_kernel_name_ptr();
namespace {
enum { STACK_BYTES = 256 * 1024, MAX_THREADS = 1024, MAX_WAVES = MAX_THREADS / 64 };
enum State { RUN = 0, AT_BARRIER = 1, WAIT_WAVE = 2 };

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = true;
    State state = RUN;
    unsigned wait_gen = 0;
};

struct Wave {
    uint64_t vals[64];
    uint64_t res[64];
    uint64_t arrived = 0, res_mask = 0, live = 0;
    unsigned gen = 0;
    hipemu_site site = {0, "", 0};
};

struct Worker {
    std::vector<Fiber> fibers;
    Wave waves[MAX_WAVES];
    void* sched_sp = nullptr;
    int current = -1;
    const std::function<void()>* body = nullptr;
    dim3 block;
    int barrier_or = 0, barrier_result = 0;
};
thread_local Worker tl_worker;

void fiber_main()
{
    Worker& w = tl_worker;
    for (;;) { /* a fiber is re-armed by resetting its stack, so this returns only by switching away */
        (*w.body)();
        Fiber& f = w.fibers[w.current];
        f.done = true;
        hipemu_switch(&f.sp, w.sched_sp);
    }
}

void arm(Fiber& f)
{
    if (!f.stack) {
        f.stack = (char*)mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (f.stack == (char*)MAP_FAILED) abort();
    }
    /* initial frame: six callee-saved registers, then the entry address; after the `ret` the stack pointer is 8 modulo 16, as after a call */
    uintptr_t end = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
    void** p = (void**)(end - 32);
    p[0] = (void*)&fiber_main;
    p[1] = nullptr;
    for (int i = 1; i <= 6; i++) p[-i] = nullptr;
    f.sp = (void*)(p - 6);
    f.done = false;
    f.state = RUN;
}

void set_ids(const Worker& w, unsigned i)
{
    threadIdx.x = i % w.block.x;
    threadIdx.y = (i / w.block.x) % w.block.y;
    threadIdx.z = i / (w.block.x * w.block.y);
}

[[noreturn]] void deadlock(Worker& w, unsigned n, const char* why)
{
    fprintf(stderr, "hipemu: %s in kernel `%s`, block (%u,%u,%u)\n", why, g_kernel_name_ptr(), blockIdx.x, blockIdx.y, blockIdx.z);
    for (unsigned wv = 0; wv < (n + 63) / 64; wv++) {
        const Wave& W = w.waves[wv];
        unsigned long long at_barrier = 0, waiting = 0, done = 0;
        for (unsigned l = 0; l < 64 && wv * 64 + l < n; l++) {
            const Fiber& f = w.fibers[wv * 64 + l];
            if (f.done) done |= 1ull << l;
            else if (f.state == AT_BARRIER) at_barrier |= 1ull << l;
            else if (f.state == WAIT_WAVE) waiting |= 1ull << l;
        }
        fprintf(stderr, "  wave %u: done %016llx, at barrier %016llx, in a cross-lane operation (%s:%d) %016llx\n", wv, done, at_barrier, W.site.file, W.site.line, waiting);
    }
    abort();
}

void run_block(Worker& w, dim3 block)
{
    const unsigned n = block.x * block.y * block.z;
    if (n > MAX_THREADS) abort();
    if (w.fibers.size() < n) w.fibers.resize(n);
    w.block = block;
    for (unsigned i = 0; i < n; i++) arm(w.fibers[i]);
    for (unsigned wv = 0; wv < (n + 63) / 64; wv++) {
        Wave& W = w.waves[wv];
        W.arrived = 0;
        W.live = (n - wv * 64 >= 64) ? ~0ull : ((1ull << (n - wv * 64)) - 1ull);
    }
    unsigned remaining = n;
    while (remaining) {
        bool progress = false;
        unsigned at_barrier = 0;
        for (unsigned i = 0; i < n; i++) {
            Fiber& f = w.fibers[i];
            if (f.done) continue;
            if (f.state == AT_BARRIER) { at_barrier++; continue; }
            if (f.state == WAIT_WAVE) {
                const Wave& W = w.waves[i >> 6];
                if (W.gen == f.wait_gen && (W.arrived & W.live) != W.live) continue; /* still waiting for lanes of its wave */
            }
            f.state = RUN;
            set_ids(w, i);
            w.current = (int)i;
            hipemu_switch(&w.sched_sp, f.sp);
            progress = true;
            if (f.done) {
                remaining--;
                w.waves[i >> 6].live &= ~(1ull << (i & 63));
            } else if (f.state == AT_BARRIER) {
                at_barrier++;
            }
        }
        if (remaining && at_barrier == remaining) { /* everybody alive has arrived: release */
            w.barrier_result = w.barrier_or;
            w.barrier_or = 0;
            for (unsigned i = 0; i < n; i++)
                if (!w.fibers[i].done) w.fibers[i].state = RUN;
            progress = true;
        }
        if (!progress) deadlock(w, n, "deadlock (a barrier or cross-lane operation that not all live work-items reach)");
    }
}

/* ---- worker pool: persistent threads, so that the fibers' stacks and the kernels' thread_local LDS arrays are allocated once */
struct Pool {
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::vector<std::thread> threads;
    unsigned long job = 0;
    unsigned active = 0;
    unsigned use = 0; /* workers that take part in the current job */
    dim3 grid, block;
    const std::function<void()>* body = nullptr;
    std::atomic<unsigned long> next{0};
    unsigned long nblocks = 0;
    bool stop = false;

    void work(unsigned id)
    {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv_job.wait(lk, [&] { return stop || job != seen; });
                if (stop) return;
                seen = job;
            }
            if (id < use) {
                Worker& w = tl_worker;
                w.body = body;
                gridDim = grid;
                blockDim = block;
                for (;;) {
                    const unsigned long b = next.fetch_add(1);
                    if (b >= nblocks) break;
                    blockIdx.x = (unsigned)(b % grid.x);
                    blockIdx.y = (unsigned)((b / grid.x) % grid.y);
                    blockIdx.z = (unsigned)(b / ((unsigned long)grid.x * grid.y));
                    run_block(w, block);
                }
            }
            {
                std::lock_guard<std::mutex> lk(m);
                if (--active == 0) cv_done.notify_all();
            }
        }
    }

    Pool()
    {
        unsigned n = std::thread::hardware_concurrency();
        if (const char* e = getenv("HIPEMU_THREADS")) n = (unsigned)atoi(e);
        if (n < 1) n = 1;
        if (n > 64) n = 64;
        for (unsigned t = 0; t < n; t++) threads.emplace_back([this, t] { work(t); });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_job.notify_all();
        for (auto& t : threads) t.join();
    }
    void run(dim3 g, dim3 b, const std::function<void()>& fn)
    {
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return active == 0; });
        grid = g;
        block = b;
        body = &fn;
        nblocks = (unsigned long)g.x * g.y * g.z;
        next = 0;
        use = nblocks < 4 ? 1u : (unsigned)threads.size();
        active = (unsigned)threads.size();
        job++;
        cv_job.notify_all();
        cv_done.wait(lk, [&] { return active == 0; });
    }
};
std::mutex g_pool_mutex; /* one launch at a time (the product's host threads may launch concurrently) */
Pool& pool()
{
    static Pool* p = new Pool(); /* (never destroyed: the library may be unloaded while threads wait) */
    return *p;
}
} // namespace

void sync_threads()
{
    Worker& w = tl_worker;
    Fiber& f = w.fibers[w.current];
    f.state = AT_BARRIER;
    hipemu_switch(&f.sp, w.sched_sp);
}

int sync_threads_or(int v)
{
    Worker& w = tl_worker;
    if (v) w.barrier_or = 1;
    sync_threads();
    return w.barrier_result;
}

int lane_id() { return tl_worker.current & 63; }

const uint64_t* wave_exchange(hipemu_site site, uint64_t v, uint64_t* mask)
{
    Worker& w = tl_worker;
    Wave& W = w.waves[w.current >> 6];
    const int lane = w.current & 63;
    Fiber& f = w.fibers[w.current];
    if (W.arrived == 0) {
        W.site = site;
    } else if (W.site.id != site.id) {
        fprintf(stderr, "hipemu: lanes of one wave meet in different cross-lane operations (%s:%d and %s:%d): divergent control flow around one\n", W.site.file, W.site.line, site.file, site.line);
        deadlock(w, w.block.x * w.block.y * w.block.z, "divergent cross-lane operation");
    }
    W.vals[lane] = v;
    W.arrived |= 1ull << lane;
    const unsigned my_gen = W.gen;
    for (;;) {
        if (W.gen != my_gen) break; /* completed by the lane that arrived last */
        if ((W.arrived & W.live) == W.live) {
            memcpy(W.res, W.vals, sizeof W.res);
            W.res_mask = W.arrived;
            W.arrived = 0;
            W.gen++;
            break;
        }
        f.state = WAIT_WAVE;
        f.wait_gen = my_gen;
        hipemu_switch(&f.sp, w.sched_sp);
    }
    *mask = W.res_mask;
    return W.res;
}

/* v_mov_b32_dpp semantics (gfx9, wave64): which lane `lane` reads under `ctrl`, or -1 when the source is outside the wave / row */
static int dpp_source(int lane, int ctrl)
{
    const int row = lane & ~15, l = lane & 15;
    if (ctrl >= 0x000 && ctrl <= 0x0FF) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);           /* quad_perm */
    if (ctrl >= 0x101 && ctrl <= 0x10F) { const int s = l + (ctrl & 15); return s < 16 ? row + s : -1; }  /* row_shl */
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const int s = l - (ctrl & 15); return s >= 0 ? row + s : -1; }  /* row_shr */
    if (ctrl >= 0x121 && ctrl <= 0x12F) return row + ((l - (ctrl & 15)) & 15);                            /* row_ror */
    switch (ctrl) {
    case 0x130: return lane + 1 < 64 ? lane + 1 : -1; /* wave_shl:1 */
    case 0x134: return (lane + 1) & 63;               /* wave_rol:1 */
    case 0x138: return lane - 1;                      /* wave_shr:1 (lane 0: -1) */
    case 0x13C: return (lane - 1) & 63;               /* wave_ror:1 */
    case 0x140: return row + 15 - l;                  /* row_mirror */
    case 0x141: return row + (l < 8 ? 7 - l : 23 - l); /* row_half_mirror */
    case 0x142: return row > 0 ? row - 1 : -1;        /* row_bcast:15: lane 15 of the previous row */
    case 0x143: return lane >= 32 ? 31 : -1;          /* row_bcast:31 */
    default: fprintf(stderr, "hipemu: DPP control 0x%x not modelled\n", ctrl); abort();
    }
}

uint32_t dpp(hipemu_site site, uint32_t old, uint32_t src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    uint64_t m;
    const uint64_t* v = wave_exchange(site, src, &m);
    const int lane = lane_id();
    if (!((row_mask >> (lane >> 4)) & 1) || !((bank_mask >> ((lane >> 2) & 3)) & 1)) return old; /* this lane is not written */
    const int s = dpp_source(lane, ctrl);
    if (s < 0 || !((m >> s) & 1)) return bound_ctrl ? 0u : old; /* no source (or an inactive one): the destination keeps its value */
    return (uint32_t)v[s];
}

static const char* g_kernel_name = "";
const char* g_kernel_name_ptr() { return g_kernel_name; }
void run_grid(const char* name, dim3 grid, dim3 block, const std::function<void()>& body)
{
    if ((unsigned long)grid.x * grid.y * grid.z == 0) return;
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    g_kernel_name = name;
    static const bool trace = getenv("HIPEMU_TRACE") != nullptr;
    if (trace) fprintf(stderr, "hipemu: launch %s grid %u x %u block %u\n", name, grid.x, grid.y, block.x);
    pool().run(grid, block, body);
}
} // namespace hipemu

/* ------------------------------------------------------------------ runtime API subset: device memory is host memory */
namespace {
std::mutex g_mem_mutex;
std::set<std::pair<uintptr_t, size_t>> g_device_ranges;
struct Ev { double t; };
} // namespace
struct hipemu_event { double t; };

hipError_t hipMalloc(void** p, size_t n)
{
    *p = aligned_alloc(256, (n + 255) & ~(size_t)255);
    if (!*p) return hipErrorInvalidValue;
    memset(*p, 0xA5, n); /* (fresh device memory is not zero) */
    std::lock_guard<std::mutex> lk(g_mem_mutex);
    g_device_ranges.insert({(uintptr_t)*p, n});
    return hipSuccess;
}
hipError_t hipFree(void* p)
{
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_mem_mutex);
        for (auto it = g_device_ranges.begin(); it != g_device_ranges.end(); ++it)
            if (it->first == (uintptr_t)p) { g_device_ranges.erase(it); break; }
    }
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = aligned_alloc(256, (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
// (everything is synchronous here, every stream is the same; the handle is NOT null, so that code which asks "did I get a stream of my own?" -- the
// library's copy lanes, gj_runtime.hip -- takes the path it takes on the GPU: ADVICE r5)
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned) { static int one_stream; *st = reinterpret_cast<hipStream_t>(&one_stream); return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hipemu error"; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipDeviceReset(void) { return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int)
{
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "hipemu (CPU execution model of gfx950, test infrastructure)");
    p->totalGlobalMem = (size_t)16 << 30;
    p->sharedMemPerBlock = 160 * 1024;
    p->regsPerBlock = 131072;
    p->major = 9;
    p->minor = 5;
    p->multiProcessorCount = 256;
    return hipSuccess;
}
hipError_t hipDriverGetVersion(int* v) { *v = 70200000; return hipSuccess; }
hipError_t hipRuntimeGetVersion(int* v) { *v = 70200000; return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p)
{
    std::lock_guard<std::mutex> lk(g_mem_mutex);
    const uintptr_t x = (uintptr_t)p;
    for (const auto& r : g_device_ranges)
        if (x >= r.first && x < r.first + r.second) {
            a->type = hipMemoryTypeDevice;
            a->device = 0;
            a->devicePointer = (void*)p;
            a->hostPointer = nullptr;
            return hipSuccess;
        }
    return hipErrorInvalidValue;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event{0.0}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = now_ms(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
