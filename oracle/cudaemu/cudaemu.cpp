/*
 * TEST INFRASTRUCTURE ONLY -- the scheduler behind oracle/cudaemu/cuda_runtime.h (see there).
 * x86-64 System V only (the authoring container and the GPU box's host).
 */
#include <cuda_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
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

struct Fiber {
    void* sp;
    char* stack;
    bool done;
};

struct Worker {
    std::vector<Fiber> fibers;
    void* sched_sp = nullptr;
    int current = -1;
    void (*fn)(void*) = nullptr;
    void* ctx = nullptr;
    bool used_shared = false;
    long barriers = 0;
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
    w.barriers++;
    Fiber& f = w.fibers[w.current];
    cudaemu_switch(&f.sp, w.sched_sp);
}

void run_grid(dim3 grid, dim3 block, void (*fn)(void*), void* ctx)
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
                run_block(w, block);
            }
        }
    };
    if (nthreads == 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nthreads; t++) pool.emplace_back(work);
        for (auto& t : pool) t.join();
    }
}
} // namespace cudaemu
