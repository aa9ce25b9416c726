/*
 * TEST INFRASTRUCTURE ONLY -- not part of the product.
 *
 * cudaemu: a CPU execution model for the reference's CUDA translation units, so that
 * src/gpujpeg_dct_gpu.cu, gpujpeg_preprocessor.cu and gpujpeg_postprocessor.cu can be compiled
 * by g++ / clang++ straight from /root/reference (never copied; the only rewrite is the launch
 * syntax kernel<<<grid, block, shm, stream>>>(args) -> CUDAEMU_LAUNCH(grid, block, shm, stream, kernel, args),
 * applied by a sed pipe in oracle/Makefile) and RUN on the host:
 *   - a thread block is a set of fibers (one per CUDA thread) on private stacks; __syncthreads() yields
 *     to the next fiber of the block, so barrier semantics are exact;
 *   - __shared__ arrays are function-local static thread_local storage (one block runs per OS thread at a time);
 *   - a kernel that uses __shared__ memory but never reaches a barrier (the warp-synchronous fDCT kernel,
 *     src/gpujpeg_dct_gpu.cu:265-270) is run twice per block: the second pass sees every lane's shared
 *     stores of the first, which equals lock-step execution for kernels whose shared stores do not depend
 *     on shared loads;
 *   - thread blocks are distributed over the host cores.
 * Included as <cuda_runtime.h> by the reference headers when oracle/cudaemu precedes oracle/stub on the
 * include path; the plain-C part (memory, events) is oracle/stub/cuda_runtime.h.
 */
#ifndef GJ_ORACLE_CUDAEMU_H
#define GJ_ORACLE_CUDAEMU_H

#include "../stub/cuda_runtime.h"

#ifdef __cplusplus
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>

struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uchar4 { unsigned char x, y, z, w; };
struct int4 { int x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r = {x, y, z, w}; return r; }

extern thread_local uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

#define __launch_bounds__(...)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __constant__ static
#define __align__(n) __attribute__((aligned(n)))
#define CUDAEMU_CAT2(a, b) a##b
#define CUDAEMU_CAT(a, b) CUDAEMU_CAT2(a, b)
#define __shared__ cudaemu::shared_mark CUDAEMU_CAT(cudaemu_shared_, __LINE__); static thread_local

static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline int __mul24(int a, int b) { return ((a << 8) >> 8) * ((b << 8) >> 8); }
static inline unsigned __byte_perm(unsigned x, unsigned y, unsigned s)
{
    unsigned long long v = ((unsigned long long)y << 32) | x;
    unsigned r = 0;
    for (int i = 0; i < 4; i++) {
        unsigned sel = (s >> (4 * i)) & 0xF;
        unsigned b = (unsigned)(v >> (8 * (sel & 7))) & 0xFF;
        if (sel & 8) b = (b & 0x80) ? 0xFF : 0x00;   /* PTX prmt default mode: msb replication */
        r |= b << (8 * i);
    }
    return r;
}

/* what CUDA's headers give device code besides the above: the integer intrinsics of the two Huffman modules, mixed-sign min / max */
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline unsigned max(unsigned a, int b) { return a > (unsigned)b ? a : (unsigned)b; }
static inline unsigned max(int a, unsigned b) { return (unsigned)a > b ? (unsigned)a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned min(unsigned a, int b) { return a < (unsigned)b ? a : (unsigned)b; }
static inline unsigned min(int a, unsigned b) { return (unsigned)a < b ? (unsigned)a : b; }
enum { cudaFuncCachePreferShared = 1 };
template <typename F> static inline cudaError_t cudaFuncSetCacheConfig(F, int) { return cudaSuccess; }

namespace cudaemu {
struct shared_mark { shared_mark(); };
void sync_threads();
/* runs fn(ctx) once per CUDA thread of every block of the grid. warp = true: WARP MODE (see cudaemu.cpp) -- the 32 threads of a warp run in
 * lock step as far as a program can tell: shared-memory reads wait until every other lane of the warp has stopped (at a read, a vote, a
 * barrier, or its end), votes collect all lanes. Needs the translation unit compiled with -fsanitize=thread: its load / store hooks are how the
 * emulator sees shared memory accesses. */
void run_grid(dim3 grid, dim3 block, void (*fn)(void*), void* ctx, bool warp);
unsigned ballot(int predicate);
unsigned atomic_add(unsigned* p, unsigned v);

template <typename F, typename... A>
void launch(dim3 grid, dim3 block, size_t shm, cudaStream_t stream, F kernel, A... args)
{
    (void)shm; (void)stream;
    auto body = [&]() { kernel(args...); };
#ifdef CUDAEMU_WARP
    const bool warp = true;
#else
    const bool warp = false;
#endif
    run_grid(grid, block, [](void* p) { (*static_cast<decltype(body)*>(p))(); }, &body, warp);
}
} // namespace cudaemu

#define __syncthreads() cudaemu::sync_threads()
#define CUDAEMU_LAUNCH(grid, block, shm, stream, kernel, ...) cudaemu::launch(grid, block, shm, stream, kernel, ##__VA_ARGS__)
/* warp votes (src/gpujpeg_huffman_gpu_encoder.cu:202-262): every lane of the warp that has not ended takes part */
#define __ballot_sync(mask, pred) cudaemu::ballot((pred) != 0)
#define __ballot(pred) cudaemu::ballot((pred) != 0)
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return cudaemu::atomic_add(p, v); }
#define cudaMemcpyToSymbol(sym, src, n, off, kind) (memcpy((char*)(sym) + (off), (src), (n)), cudaSuccess)

#define cudaMemcpyToSymbolAsync(sym, src, n, off, kind, st) (memcpy((char*)(sym) + (off), (src), (n)), cudaSuccess)
static inline cudaError_t cudaMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t w, size_t h,
                                            enum cudaMemcpyKind k, cudaStream_t st)
{
    (void)k; (void)st;
    for (size_t y = 0; y < h; y++) memmove((char*)d + y * dpitch, (const char*)s + y * spitch, w);
    return cudaSuccess;
}
#endif /* __cplusplus */
#endif
