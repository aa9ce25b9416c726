/*
 * TEST INFRASTRUCTURE ONLY -- the scheduler behind oracle/cudaemu/cuda_runtime.h (see there).
 * x86-64 System V only (the authoring container and the GPU box's host).
 */
#include <cuda_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <link.h>
#include <thread>
#include <vector>

thread_local uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

extern "C" void cudaemu_switch(void** save_sp, void* new_sp);
asm(R"(
    .text
    .globl cudaemu_switch
    .type cudaemu_switch,@function
cudaemu_switch:
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
    .size cudaemu_switch,.-cudaemu_switch
)");

namespace cudaemu {
namespace {
enum { STACK_BYTES = 64 * 1024 };

/* warp mode: where a lane has stopped */
enum LaneState { LANE_RUN, LANE_AT_READ, LANE_AT_VOTE, LANE_AT_BARRIER, LANE_PAUSED, LANE_DONE };

struct Fiber {
    void* sp;
    char* stack;
    bool done;
    int state;        /* LaneState */
    unsigned vote;    /* the lane's predicate / the warp's result */
};

struct Worker {
    std::vector<Fiber> fibers;
    void* sched_sp = nullptr;
    int current = -1;
    void (*fn)(void*) = nullptr;
    void* ctx = nullptr;
    bool used_shared = false;
    long barriers = 0;
    bool warp_mode = false;   /* this launch runs its warps in lock step (see run_block_warps) */
    bool read_round = false;  /* lanes that stopped at a shared-memory read are being let through, one after the other */
    unsigned long stat_votes = 0, stat_rounds = 0;
    uintptr_t tls_lo = 0, tls_hi = 0; /* this thread's block of the module's thread-local storage: where `__shared__` arrays live */
    ~Worker() { for (auto& f : fibers) free(f.stack); }
};
thread_local Worker tl_worker;

void fiber_main()
{
    Worker& w = tl_worker;
    for (;;) {                       /* a fiber is re-armed by resetting its stack, so this returns only by switching */
        w.fn(w.ctx);
        Fiber& f = w.fibers[w.current];
        f.done = true;
        cudaemu_switch(&f.sp, w.sched_sp);
    }
}

void arm(Fiber& f)
{
    /* initial frame: six callee-saved registers, then the entry address; after the `ret` the stack pointer is
     * 8 modulo 16, as after a call */
    uintptr_t end = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
    void** p = (void**)(end - 32);
    p[0] = (void*)&fiber_main;
    p[1] = nullptr;
    for (int i = 1; i <= 6; i++) p[-i] = nullptr;
    f.sp = (void*)(p - 6);
    f.done = false;
    f.state = LANE_RUN;
    f.vote = 0;
}

/* ------------------------------------------------------------------------------------------------------------------------------------
 * WARP MODE. The reference's Huffman encoder kernel (src/gpujpeg_huffman_gpu_encoder.cu:304-404) is warp-synchronous code of the pre-Volta
 * kind: the 32 lanes of a warp exchange code words through shared memory with no barrier between the stores and the loads, and vote
 * (__ballot_sync) four times per block. What such code relies on is that a lane's shared-memory READ sees every store the other lanes issue
 * in front of it in program order. The emulator gives exactly that, at the granularity the program can observe:
 *   - a lane runs until it is about to READ shared memory, votes, reaches __syncthreads() or ends; shared-memory WRITES do not stop it;
 *   - a read goes ahead only when every other lane of the warp has stopped too (so all stores in front of that point are done), and the lanes
 *     that wait at reads are then let through one after the other, each up to its NEXT access to shared memory of either kind -- no store
 *     of the same round can slip in front of another lane's read;
 *   - a vote completes when every lane that has not ended waits at a vote; __syncthreads() when every live lane of the block waits at it.
 * The loads and stores are seen through the compiler's -fsanitize=thread hooks (__tsan_read* / __tsan_write*, defined at the end of this
 * file; no sanitizer runtime is linked), `__shared__` arrays are recognised as addresses inside this thread's block of thread-local storage.
 * Limits, stated: lanes that diverge around a shared-memory read are ordered by "reads wait for everybody", not by program counter (a store a
 * lane issues AFTER skipping a read another lane still waits at would be seen by that read); and a shared-to-shared struct copy shows its
 * two hooks in front of the copy. Neither pattern occurs in the two Huffman modules; their results are checked byte for byte against the
 * reference's CPU Huffman coders by tests/test_oracle_vs_ref.py.
 * ---------------------------------------------------------------------------------------------------------------------------------- */
void stop_lane(Worker& w, int state)
{
    Fiber& f = w.fibers[w.current];
    f.state = state;
    cudaemu_switch(&f.sp, w.sched_sp);
}

inline void pause_if_round(Worker& w)
{
    if (w.read_round) stop_lane(w, LANE_PAUSED);
}

void resume_lane(Worker& w, dim3 block, unsigned i)
{
    Fiber& f = w.fibers[i];
    threadIdx.x = i % block.x;
    threadIdx.y = (i / block.x) % block.y;
    threadIdx.z = i / (block.x * block.y);
    w.current = (int)i;
    f.state = LANE_RUN;
    cudaemu_switch(&w.sched_sp, f.sp);
    if (f.done) f.state = LANE_DONE;
}

/* runs the lanes [a, b) of one warp until each of them waits at __syncthreads() or has ended */
void run_warp(Worker& w, dim3 block, unsigned a, unsigned b)
{
    for (;;) {
        for (unsigned i = a; i < b; i++)
            if (w.fibers[i].state == LANE_RUN) resume_lane(w, block, i);
        unsigned live = 0, at_read = 0, at_vote = 0, at_barrier = 0;
        for (unsigned i = a; i < b; i++) {
            const int st = w.fibers[i].state;
            live += st != LANE_DONE;
            at_read += st == LANE_AT_READ;
            at_vote += st == LANE_AT_VOTE;
            at_barrier += st == LANE_AT_BARRIER;
        }
        if (live == 0 || at_barrier == live) return;
        if (at_read) { /* everybody has stopped: the reads see all stores in front of them */
            w.stat_rounds++;
            w.read_round = true;
            for (unsigned i = a; i < b; i++)
                if (w.fibers[i].state == LANE_AT_READ) resume_lane(w, block, i); /* ... up to its next shared access (LANE_PAUSED), vote, barrier or end */
            w.read_round = false;
            for (unsigned i = a; i < b; i++)
                if (w.fibers[i].state == LANE_PAUSED) w.fibers[i].state = LANE_RUN;
            continue;
        }
        if (at_vote == live) {
            w.stat_votes++;
            unsigned mask = 0;
            for (unsigned i = a; i < b; i++)
                if (w.fibers[i].state == LANE_AT_VOTE && w.fibers[i].vote) mask |= 1u << (i - a);
            for (unsigned i = a; i < b; i++)
                if (w.fibers[i].state == LANE_AT_VOTE) { w.fibers[i].vote = mask; w.fibers[i].state = LANE_RUN; }
            continue;
        }
        fprintf(stderr, "[cudaemu] warp dead-locked: %u lanes live, %u at a vote, %u at __syncthreads()\n", live, at_vote, at_barrier);
        abort();
    }
}

int tls_phdr(struct dl_phdr_info* info, size_t, void* data)
{
    Worker& w = *static_cast<Worker*>(data);
    const uintptr_t here = (uintptr_t)&tls_phdr;
    bool mine = false;
    size_t tls_size = 0;
    for (int i = 0; i < info->dlpi_phnum; i++) {
        const ElfW(Phdr)& ph = info->dlpi_phdr[i];
        if (ph.p_type == PT_LOAD && here >= info->dlpi_addr + ph.p_vaddr && here < info->dlpi_addr + ph.p_vaddr + ph.p_memsz) mine = true;
        if (ph.p_type == PT_TLS) tls_size = ph.p_memsz;
    }
    if (!mine || !info->dlpi_tls_data) return 0;
    w.tls_lo = (uintptr_t)info->dlpi_tls_data;
    w.tls_hi = w.tls_lo + tls_size;
    return 1;
}

void run_block_warps(Worker& w, dim3 block)
{
    const unsigned n = block.x * block.y * block.z;
    while (w.fibers.size() < n) {
        Fiber f;
        f.stack = (char*)malloc(STACK_BYTES);
        if (!f.stack) abort();
        w.fibers.push_back(f);
    }
    if (!w.tls_hi) {
        dl_iterate_phdr(tls_phdr, &w);
        if (!w.tls_hi) { fprintf(stderr, "[cudaemu] no thread-local storage block found for this module\n"); abort(); }
    }
    for (unsigned i = 0; i < n; i++) arm(w.fibers[i]);
    for (;;) {
        for (unsigned a = 0; a < n; a += 32) run_warp(w, block, a, a + 32 < n ? a + 32 : n);
        unsigned live = 0;
        for (unsigned i = 0; i < n; i++) live += w.fibers[i].state != LANE_DONE;
        if (!live) break;
        for (unsigned i = 0; i < n; i++) /* every live lane of the block waits at the barrier (run_warp returns on nothing else) */
            if (w.fibers[i].state == LANE_AT_BARRIER) w.fibers[i].state = LANE_RUN;
    }
}

void run_block(Worker& w, dim3 block)
{
    const unsigned n = block.x * block.y * block.z;
    while (w.fibers.size() < n) {
        Fiber f;
        f.stack = (char*)malloc(STACK_BYTES);
        if (!f.stack) abort();
        w.fibers.push_back(f);
    }
    for (int pass = 0; pass < 2; pass++) {
        w.used_shared = false;
        w.barriers = 0;
        for (unsigned i = 0; i < n; i++) arm(w.fibers[i]);
        unsigned remaining = n;
        while (remaining) {
            /* one round: every unfinished thread runs up to its next barrier (or to its end) */
            for (unsigned i = 0; i < n; i++) {
                Fiber& f = w.fibers[i];
                if (f.done) continue;
                threadIdx.x = i % block.x;
                threadIdx.y = (i / block.x) % block.y;
                threadIdx.z = i / (block.x * block.y);
                w.current = (int)i;
                cudaemu_switch(&w.sched_sp, f.sp);
                if (f.done) remaining--;
            }
        }
        /* warp-synchronous kernel (shared memory, no barrier): second pass over the block, see cuda_runtime.h */
        if (!(w.used_shared && w.barriers == 0)) break;
    }
}
} // namespace

shared_mark::shared_mark() { tl_worker.used_shared = true; }

void sync_threads()
{
    Worker& w = tl_worker;
    if (w.warp_mode) {
        pause_if_round(w);
        stop_lane(w, LANE_AT_BARRIER);
        return;
    }
    w.barriers++;
    Fiber& f = w.fibers[w.current];
    cudaemu_switch(&f.sp, w.sched_sp);
}

unsigned ballot(int predicate)
{
    Worker& w = tl_worker;
    if (!w.warp_mode) { fprintf(stderr, "[cudaemu] warp vote in a translation unit compiled without -DCUDAEMU_WARP\n"); abort(); }
    pause_if_round(w);
    w.fibers[w.current].vote = predicate != 0;
    stop_lane(w, LANE_AT_VOTE);
    return w.fibers[w.current].vote;
}

unsigned atomic_add(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

static std::atomic<unsigned long> g_warp_launches(0), g_warp_blocks(0), g_warp_votes(0), g_warp_read_rounds(0);

void run_grid(dim3 grid, dim3 block, void (*fn)(void*), void* ctx, bool warp)
{
    const unsigned long nblocks = (unsigned long)grid.x * grid.y * grid.z;
    unsigned nthreads = std::thread::hardware_concurrency();
    if (const char* e = getenv("CUDAEMU_THREADS")) nthreads = (unsigned)atoi(e);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if (nblocks < 64) nthreads = 1;
    std::atomic<unsigned long> next(0);
    auto work = [&]() {
        Worker& w = tl_worker;
        w.fn = fn;
        w.ctx = ctx;
        w.warp_mode = warp;
        gridDim = grid;
        blockDim = block;
        for (;;) {
            unsigned long b0 = next.fetch_add(64);
            if (b0 >= nblocks) break;
            unsigned long b1 = b0 + 64 < nblocks ? b0 + 64 : nblocks;
            for (unsigned long b = b0; b < b1; b++) {
                blockIdx.x = (unsigned)(b % grid.x);
                blockIdx.y = (unsigned)((b / grid.x) % grid.y);
                blockIdx.z = (unsigned)(b / ((unsigned long)grid.x * grid.y));
                if (warp) run_block_warps(w, block);
                else run_block(w, block);
            }
            if (warp) g_warp_blocks += b1 - b0;
        }
        w.warp_mode = false;
        g_warp_votes += w.stat_votes;
        g_warp_read_rounds += w.stat_rounds;
        w.stat_votes = w.stat_rounds = 0;
    };
    if (warp) g_warp_launches++;
    if (nthreads == 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nthreads; t++) pool.emplace_back(work);
        for (auto& t : pool) t.join();
    }
}
} // namespace cudaemu

/* ---- the compiler's -fsanitize=thread hooks (only the two Huffman modules are compiled with them): shared-memory loads and stores of warp mode */
namespace {
inline bool is_shared(const cudaemu::Worker& w, const void* p)
{
    const uintptr_t a = (uintptr_t)p;
    if (a < w.tls_lo || a >= w.tls_hi) return false;
    /* the built-in variables live in the same block */
    if ((a >= (uintptr_t)&threadIdx && a < (uintptr_t)(&threadIdx + 1)) || (a >= (uintptr_t)&blockIdx && a < (uintptr_t)(&blockIdx + 1)) ||
        (a >= (uintptr_t)&blockDim && a < (uintptr_t)(&blockDim + 1)) || (a >= (uintptr_t)&gridDim && a < (uintptr_t)(&gridDim + 1)) ||
        (a >= (uintptr_t)&cudaemu::tl_worker && a < (uintptr_t)(&cudaemu::tl_worker + 1)))
        return false;
    return true;
}
inline void shared_load(const void* p)
{
    cudaemu::Worker& w = cudaemu::tl_worker;
    if (!w.warp_mode || !is_shared(w, p)) return;
    cudaemu::pause_if_round(w);
    cudaemu::stop_lane(w, cudaemu::LANE_AT_READ);
}
inline void shared_store(const void* p)
{
    cudaemu::Worker& w = cudaemu::tl_worker;
    if (!w.warp_mode || !is_shared(w, p)) return;
    cudaemu::pause_if_round(w);
}
} // namespace

extern "C" {
/* test hook: launches / thread blocks / completed votes / read rounds of warp mode so far (tests/test_oracle_vs_ref.py asserts that the
 * reference's Huffman GPU kernels really ran) */
__attribute__((visibility("default"))) void cudaemu_warp_stats(unsigned long out[4])
{
    out[0] = cudaemu::g_warp_launches; out[1] = cudaemu::g_warp_blocks; out[2] = cudaemu::g_warp_votes; out[3] = cudaemu::g_warp_read_rounds;
}
void __tsan_init(void) {}
void __tsan_func_entry(void*) {}
void __tsan_func_exit(void) {}
void __tsan_read1(void* p) { shared_load(p); }
void __tsan_read2(void* p) { shared_load(p); }
void __tsan_read4(void* p) { shared_load(p); }
void __tsan_read8(void* p) { shared_load(p); }
void __tsan_read16(void* p) { shared_load(p); }
void __tsan_unaligned_read2(void* p) { shared_load(p); }
void __tsan_unaligned_read4(void* p) { shared_load(p); }
void __tsan_unaligned_read8(void* p) { shared_load(p); }
void __tsan_unaligned_read16(void* p) { shared_load(p); }
void __tsan_read_range(void* p, unsigned long) { shared_load(p); }
void __tsan_write1(void* p) { shared_store(p); }
void __tsan_write2(void* p) { shared_store(p); }
void __tsan_write4(void* p) { shared_store(p); }
void __tsan_write8(void* p) { shared_store(p); }
void __tsan_write16(void* p) { shared_store(p); }
void __tsan_unaligned_write2(void* p) { shared_store(p); }
void __tsan_unaligned_write4(void* p) { shared_store(p); }
void __tsan_unaligned_write8(void* p) { shared_store(p); }
void __tsan_unaligned_write16(void* p) { shared_store(p); }
void __tsan_write_range(void* p, unsigned long) { shared_store(p); }
void __tsan_vptr_update(void**, void*) {}
void __tsan_vptr_read(void**) {}
}
