/*
 * TEST INFRASTRUCTURE ONLY -- not part of the product.
 *
 * A host-memory stand-in for the handful of CUDA runtime entry points the reference's
 * *host* C files use, so that those files can be compiled straight from /root/reference
 * (never copied) into oracle/_ref/ and serve as the literal oracle for geometry, tables,
 * the JFIF writer/reader and the CPU Huffman coders. "Device" memory is plain malloc'd
 * host memory; streams and events are no-ops.
 */
#ifndef GJ_ORACLE_CUDA_RUNTIME_STUB_H
#define GJ_ORACLE_CUDA_RUNTIME_STUB_H
#define __DRIVER_TYPES_H__ 1
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int cudaError_t;
enum { cudaSuccess = 0, cudaErrorUnknown = 999 };
struct CUstream_st;
typedef struct CUstream_st* cudaStream_t;
struct gj_stub_event { double t; };
typedef struct gj_stub_event* cudaEvent_t;
#define cudaStreamDefault ((cudaStream_t)0)
enum cudaMemcpyKind { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2,
                      cudaMemcpyDeviceToDevice = 3, cudaMemcpyDefault = 4 };
#define cudaHostRegisterDefault 0
#define CUDART_VERSION 12000

struct cudaDeviceProp {
    char name[256];
    int major, minor;
    size_t totalGlobalMem, totalConstMem, sharedMemPerBlock;
    int regsPerBlock, multiProcessorCount;
};
struct cudaPointerAttributes { int type; int device; void* devicePointer; void* hostPointer; };
enum { cudaMemoryTypeUnregistered = 0, cudaMemoryTypeHost = 1, cudaMemoryTypeDevice = 2, cudaMemoryTypeManaged = 3 };

static inline cudaError_t cudaMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? cudaSuccess : cudaErrorUnknown; }
static inline cudaError_t cudaMallocHost(void** p, size_t n) { *p = calloc(1, n + 4096); /* page-granular like real pinned memory */ return *p ? cudaSuccess : cudaErrorUnknown; }
static inline cudaError_t cudaFree(void* p) { free(p); return cudaSuccess; }
static inline cudaError_t cudaFreeHost(void* p) { free(p); return cudaSuccess; }
static inline cudaError_t cudaHostRegister(void* p, size_t n, unsigned f) { (void)p; (void)n; (void)f; return cudaSuccess; }
static inline cudaError_t cudaHostUnregister(void* p) { (void)p; return cudaSuccess; }
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, enum cudaMemcpyKind k) { (void)k; memmove(d, s, n); return cudaSuccess; }
static inline cudaError_t cudaMemcpyAsync(void* d, const void* s, size_t n, enum cudaMemcpyKind k, cudaStream_t st) { (void)k; (void)st; memmove(d, s, n); return cudaSuccess; }
static inline cudaError_t cudaMemset(void* d, int v, size_t n) { memset(d, v, n); return cudaSuccess; }
static inline cudaError_t cudaMemsetAsync(void* d, int v, size_t n, cudaStream_t st) { (void)st; memset(d, v, n); return cudaSuccess; }
static inline cudaError_t cudaStreamSynchronize(cudaStream_t st) { (void)st; return cudaSuccess; }
static inline cudaError_t cudaDeviceSynchronize(void) { return cudaSuccess; }
static inline cudaError_t cudaDeviceReset(void) { return cudaSuccess; }
static inline cudaError_t cudaGetLastError(void) { return cudaSuccess; }
static inline const char* cudaGetErrorString(cudaError_t e) { return e == cudaSuccess ? "no error" : "stub error"; }
static inline cudaError_t cudaGetDevice(int* d) { *d = 0; return cudaSuccess; }
static inline cudaError_t cudaSetDevice(int d) { (void)d; return cudaSuccess; }
static inline cudaError_t cudaGetDeviceCount(int* c) { *c = 1; return cudaSuccess; }
static inline cudaError_t cudaDriverGetVersion(int* v) { *v = 0; return cudaSuccess; }
static inline cudaError_t cudaRuntimeGetVersion(int* v) { *v = 0; return cudaSuccess; }
static inline cudaError_t cudaGetDeviceProperties(struct cudaDeviceProp* p, int d) {
    (void)d; memset(p, 0, sizeof *p); strcpy(p->name, "host-stub"); p->major = 9; return cudaSuccess;
}
static inline cudaError_t cudaPointerGetAttributes(struct cudaPointerAttributes* a, const void* p) {
    memset(a, 0, sizeof *a); a->type = cudaMemoryTypeUnregistered; a->hostPointer = (void*)p; return cudaSuccess;
}
static inline cudaError_t cudaEventCreate(cudaEvent_t* e) { *e = (cudaEvent_t)calloc(1, sizeof **e); return cudaSuccess; }
static inline cudaError_t cudaEventDestroy(cudaEvent_t e) { free(e); return cudaSuccess; }
static inline cudaError_t cudaEventRecord(cudaEvent_t e, cudaStream_t st) { (void)e; (void)st; return cudaSuccess; }
static inline cudaError_t cudaEventSynchronize(cudaEvent_t e) { (void)e; return cudaSuccess; }
static inline cudaError_t cudaEventElapsedTime(float* ms, cudaEvent_t a, cudaEvent_t b) { (void)a; (void)b; *ms = 0.0f; return cudaSuccess; }

#ifdef __cplusplus
}
#endif
#endif
