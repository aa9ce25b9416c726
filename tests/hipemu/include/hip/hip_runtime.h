/*
 * TEST INFRASTRUCTURE ONLY -- not part of the product, never linked into libgpujpeg.so.
 *
 * hipemu: a CPU execution model for the product's own gfx950 HIP kernels (the .hip files of gpujpeg_amd/csrc), so that the CPU test
 * tier can run the real kernel code -- wave64 cross-lane operations, LDS, workgroup barriers, atomics -- against the oracle
 * without a GPU (tests/test_emu_parity.py), and so that a kernel can be debugged with host tools (gdb, ASan on LDS arrays).
 * The .hip files are compiled unmodified by clang++ for x86-64 with this directory in front of the include path and
 * -DGJ_HIPEMU (which only replaces the handful of inline-assembly helpers of gj_device.h by their C++ meaning):
 *   - a workgroup is a set of fibers (one per work-item) on private stacks, run on one OS thread; __syncthreads() yields until
 *     every live fiber of the workgroup has arrived;
 *   - a wave is 64 consecutive fibers; every cross-lane operation (ballot, readlane, DPP, bpermute) is a rendezvous of the live
 *     lanes of the wave at the same call site, then evaluated with the ISA's semantics (DPP controls, row / bank masks,
 *     bound_ctrl). A cross-lane operation reached by only some lanes of a wave (divergent control flow around it) is reported
 *     as an error rather than guessed at: the product keeps them in wave-uniform control flow;
 *   - __shared__ arrays are static thread_local storage (one workgroup runs per OS thread at a time);
 *   - workgroups are distributed over the host's cores in launch order (a workgroup may wait for lower-numbered ones, as the
 *     look-back scan of the encoder does); device memory is host memory, streams are synchronous.
 * What it does not model: timing, memory ordering between workgroups beyond acquire/release on the atomics, bank conflicts.
 */
#ifndef GJ_HIPEMU_RUNTIME_H
#define GJ_HIPEMU_RUNTIME_H

#include <stddef.h>
#include <stdint.h>
#include <sched.h>
#include <string.h>

#include <cmath>
#include <functional>

#ifndef GJ_HIPEMU
#define GJ_HIPEMU 1
#endif

/* ------------------------------------------------------------------ language */
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __constant__ static const

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hipemu_idx { unsigned x, y, z; };
extern thread_local hipemu_idx threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

struct __attribute__((aligned(8))) uint2 { uint32_t x, y; };
struct __attribute__((aligned(16))) uint4 { uint32_t x, y, z, w; };
struct __attribute__((aligned(8))) int2 { int x, y; };
struct __attribute__((aligned(8))) float2 { float x, y; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 r = {x, y}; return r; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 r = {x, y, z, w}; return r; }

template <typename A, typename B> static inline auto min(A a, B b) -> decltype(a + b) { return a < b ? a : b; }
template <typename A, typename B> static inline auto max(A a, B b) -> decltype(a + b) { return a > b ? a : b; }

/* ------------------------------------------------------------------ execution model (hipemu.cpp) */
/* identifies the call site of a cross-lane operation: unique number + where it is (for the diagnostics) */
struct hipemu_site { int id; const char* file; int line; };
namespace hipemu {
void sync_threads();
int sync_threads_or(int v);
/* rendezvous of the live lanes of the calling fiber's wave at `site`; returns the wave's value array (indexed by lane) and the mask of
 * the lanes that took part */
const uint64_t* wave_exchange(hipemu_site site, uint64_t v, uint64_t* mask);
int lane_id();
void run_grid(const char* name, dim3 grid, dim3 block, const std::function<void()>& body);
uint32_t dpp(hipemu_site site, uint32_t old, uint32_t src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl);
} // namespace hipemu

#define __syncthreads() hipemu::sync_threads()
#define __syncthreads_or(v) hipemu::sync_threads_or(v)

#define HIPEMU_SITE (hipemu_site{__COUNTER__ + 1, __FILE__, __LINE__})
static inline unsigned long long hipemu_ballot(hipemu_site site, bool p)
{
    uint64_t m;
    const uint64_t* v = hipemu::wave_exchange(site, p ? 1 : 0, &m);
    unsigned long long r = 0;
    for (int i = 0; i < 64; i++)
        if (((m >> i) & 1) && v[i]) r |= 1ull << i;
    return r;
}
static inline int hipemu_readlane(hipemu_site site, int x, int lane)
{
    uint64_t m;
    const uint64_t* v = hipemu::wave_exchange(site, (uint32_t)x, &m);
    return ((m >> (lane & 63)) & 1) ? (int)(uint32_t)v[lane & 63] : 0; /* (an inactive lane's register: whatever it held; 0 here) */
}
static inline int hipemu_readfirstlane(hipemu_site site, int x)
{
    uint64_t m;
    const uint64_t* v = hipemu::wave_exchange(site, (uint32_t)x, &m);
    return (int)(uint32_t)v[__builtin_ctzll(m)];
}
static inline int hipemu_bpermute(hipemu_site site, int addr, int x)
{
    uint64_t m;
    const uint64_t* v = hipemu::wave_exchange(site, ((uint64_t)(uint32_t)addr << 32) | (uint32_t)x, &m);
    const int src = ((v[hipemu::lane_id()] >> 32) >> 2) & 63;
    return ((m >> src) & 1) ? (int)(uint32_t)v[src] : 0;
}
#define __ballot(p) hipemu_ballot(HIPEMU_SITE, (p))
#define __any(p) (hipemu_ballot(HIPEMU_SITE, (p)) != 0ull)
#define __all(p) (hipemu_ballot(HIPEMU_SITE, !(p)) == 0ull)
#define __builtin_amdgcn_readlane(x, l) hipemu_readlane(HIPEMU_SITE, (x), (l))
#define __builtin_amdgcn_readfirstlane(x) hipemu_readfirstlane(HIPEMU_SITE, (x))
#define __builtin_amdgcn_ds_bpermute(a, x) hipemu_bpermute(HIPEMU_SITE, (a), (x))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) ((int)hipemu::dpp(HIPEMU_SITE, (uint32_t)(old), (uint32_t)(src), (ctrl), (rm), (bm), (bc)))
/* the wave barrier is where the product says "lanes of this wave exchange data through LDS here" (gj_wave_sync): in lockstep hardware a
 * compiler fence, for independent fibers a rendezvous */
static inline void hipemu_wave_barrier(hipemu_site site) { uint64_t m; (void)hipemu::wave_exchange(site, 0, &m); }
#define __builtin_amdgcn_wave_barrier() hipemu_wave_barrier(HIPEMU_SITE)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_sleep(n) sched_yield() /* a workgroup waiting for one on another OS thread */
#define __threadfence() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(order)
#define __popc(x) __builtin_popcount(x)
#define __popcll(x) __builtin_popcountll(x)

/* per-lane instructions */
static inline uint32_t hipemu_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
    const uint64_t in = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t s = (sel >> (8 * i)) & 0xFF;
        uint32_t b;
        if (s <= 7) b = (uint32_t)(in >> (8 * s)) & 0xFF;
        else if (s <= 11) b = ((in >> (16 * (s - 8) + 15)) & 1) ? 0xFF : 0x00; /* sign of byte 1, 3, 5, 7 */
        else if (s == 12) b = 0x00;
        else b = 0xFF;
        r |= b << (8 * i);
    }
    return r;
}
static inline uint32_t hipemu_cvt_pk_u8_f32(float f, int byte, uint32_t old)
{
    float r = nearbyintf(f); /* round to nearest even (default rounding mode) */
    uint32_t b = !(r > 0.0f) ? 0u : (r >= 255.0f ? 255u : (uint32_t)r); /* NaN and negatives -> 0 */
    return (old & ~(0xFFu << (8 * (byte & 3)))) | (b << (8 * (byte & 3)));
}
#define __builtin_amdgcn_perm(a, b, s) hipemu_perm((a), (b), (s))
#define __builtin_amdgcn_cvt_pk_u8_f32(f, i, o) hipemu_cvt_pk_u8_f32((f), (i), (o))
#define __builtin_amdgcn_alignbit(hi, lo, s) ((uint32_t)(((((uint64_t)(uint32_t)(hi)) << 32) | (uint32_t)(lo)) >> ((s) & 31)))
#define __builtin_amdgcn_alignbyte(hi, lo, s) ((uint32_t)(((((uint64_t)(uint32_t)(hi)) << 32) | (uint32_t)(lo)) >> (8 * ((s) & 3))))
#define __builtin_amdgcn_ubfe(v, off, w) ((w) == 0 ? 0u : (((uint32_t)(v) >> ((off) & 31)) & ((w) >= 32 ? 0xFFFFFFFFu : ((1u << (w)) - 1u))))

/* atomics: LDS objects are private to the OS thread that runs the workgroup, global ones are shared between OS threads */
template <typename T, typename U> static inline T atomicAdd(T* p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_ACQ_REL); }
template <typename T, typename U> static inline T atomicMax(T* p, U v)
{
    T old = __atomic_load_n(p, __ATOMIC_ACQUIRE);
    while (old < (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {}
    return old;
}
template <typename T, typename U> static inline T atomicMin(T* p, U v)
{
    T old = __atomic_load_n(p, __ATOMIC_ACQUIRE);
    while (old > (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {}
    return old;
}
template <typename T, typename U> static inline T atomicOr(T* p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_ACQ_REL); }
template <typename T, typename U> static inline T atomicExch(T* p, U v) { return __atomic_exchange_n(p, (T)v, __ATOMIC_ACQ_REL); }
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), (order))

/* ------------------------------------------------------------------ runtime API subset (what gj_runtime.hip and the launchers call) */
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
typedef struct hipemu_stream* hipStream_t;
#define hipStreamNonBlocking 1u
typedef struct hipemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipHostMallocDefault = 0, hipHostMallocPortable = 1 };
enum hipMemoryType { hipMemoryTypeHost = 0, hipMemoryTypeDevice = 1, hipMemoryTypeUnregistered = 3 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void* devicePointer; void* hostPointer; };
struct hipDeviceProp_t {
    char name[256];
    size_t totalGlobalMem, sharedMemPerBlock;
    int regsPerBlock, major, minor, multiProcessorCount;
};
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipStreamSynchronize(hipStream_t st);
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned flags);
hipError_t hipGetLastError(void);
const char* hipGetErrorString(hipError_t e);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDevice(int* d);
hipError_t hipSetDevice(int d);
hipError_t hipDeviceReset(void);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipDriverGetVersion(int* v);
hipError_t hipRuntimeGetVersion(int* v);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);

#define hipLaunchKernelGGL(kernel, grid, block, shm, stream, ...) hipemu::run_grid(#kernel, (grid), (block), [=]() { (kernel)(__VA_ARGS__); })

#endif
