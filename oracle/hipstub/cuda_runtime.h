/*
 * TEST INFRASTRUCTURE ONLY -- not part of the product.
 *
 * hipstub: lets the reference's own CUDA translation units src/gpujpeg_dct_gpu.cu, gpujpeg_preprocessor.cu and
 * gpujpeg_postprocessor.cu be compiled UNMODIFIED, where they lie, by hipcc for gfx950 (launch syntax, __global__,
 * __shared__, uchar4, __byte_perm ... are native HIP) and run on the MI355X next to the product, so that the
 * product's float arithmetic is compared with the reference's source text under the compiler family (LLVM, aggressive
 * FMA fusion) closest to nvcc's. Built into oracle/_ref/libgpujpeg_refhip.so by oracle/Makefile; used by
 * tests/test_gpu_refhip.py only.
 *
 * "Device" memory is pinned host memory (hipHostMalloc): valid on both sides, so the reference's host C and the
 * CPU Huffman entry points of ref_shim.c work on it directly; every host-side copy synchronises the device first.
 * This header is <cuda_runtime.h> for the reference's host C files too (compiled by gcc, plain C part only).
 */
#ifndef GJ_ORACLE_HIPSTUB_H
#define GJ_ORACLE_HIPSTUB_H
#define __DRIVER_TYPES_H__ 1
#include <stddef.h>
#include <stdio.h>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
typedef hipStream_t cudaStream_t;
#else
struct CUstream_st;
typedef struct CUstream_st* cudaStream_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef int cudaError_t;
enum { cudaSuccess = 0, cudaErrorUnknown = 999 };
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

cudaError_t cudaMalloc(void** p, size_t n);
cudaError_t cudaMallocHost(void** p, size_t n);
cudaError_t cudaFree(void* p);
cudaError_t cudaFreeHost(void* p);
cudaError_t cudaHostRegister(void* p, size_t n, unsigned f);
cudaError_t cudaHostUnregister(void* p);
cudaError_t cudaMemcpy(void* d, const void* s, size_t n, enum cudaMemcpyKind k);
cudaError_t cudaMemcpyAsync(void* d, const void* s, size_t n, enum cudaMemcpyKind k, cudaStream_t st);
cudaError_t cudaMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t w, size_t h, enum cudaMemcpyKind k, cudaStream_t st);
cudaError_t cudaMemset(void* d, int v, size_t n);
cudaError_t cudaMemsetAsync(void* d, int v, size_t n, cudaStream_t st);
cudaError_t cudaStreamSynchronize(cudaStream_t st);
cudaError_t cudaDeviceSynchronize(void);
cudaError_t cudaDeviceReset(void);
cudaError_t cudaGetLastError(void);
const char* cudaGetErrorString(cudaError_t e);
cudaError_t cudaGetDevice(int* d);
cudaError_t cudaSetDevice(int d);
cudaError_t cudaGetDeviceCount(int* c);
cudaError_t cudaDriverGetVersion(int* v);
cudaError_t cudaRuntimeGetVersion(int* v);
cudaError_t cudaGetDeviceProperties(struct cudaDeviceProp* p, int d);
cudaError_t cudaEventCreate(cudaEvent_t* e);
cudaError_t cudaEventDestroy(cudaEvent_t e);
cudaError_t cudaEventRecord(cudaEvent_t e, cudaStream_t st);
cudaError_t cudaEventSynchronize(cudaEvent_t e);
cudaError_t cudaEventElapsedTime(float* ms, cudaEvent_t a, cudaEvent_t b);

#ifdef __cplusplus
}
#endif

#ifdef __HIPCC__
/* the only runtime call the three .cu files make besides the C ones above (src/gpujpeg_dct_gpu.cu:637,703): the source is
 * "device" memory, here pinned host memory. Neither constant table is read by a kernel on this path (the fDCT kernel takes its
 * CC >= 2.0 branch, the IDCT kernel reads the table through its argument, :499), so a failing copy is reported, not fatal. */
template <typename T>
static inline cudaError_t gj_hipstub_to_symbol(const T& sym, const void* src, size_t n, size_t off, hipStream_t st)
{
    hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(sym), src, n, off, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) {
        static bool told = false;
        if (!told) fprintf(stderr, "[hipstub] hipMemcpyToSymbolAsync: %s (ignored: no kernel reads the symbol)\n", hipGetErrorString(e));
        told = true;
        (void)hipGetLastError();
    }
    return cudaSuccess;
}
#define cudaMemcpyToSymbolAsync(sym, src, n, off, kind, st) gj_hipstub_to_symbol(sym, (src), (n), (off), (st))
/* the kernels select their compute-capability >= 2.0 paths (src/gpujpeg_dct_gpu.cu:262-266) */
#define __CUDA_ARCH__ 900
#endif
#endif
