/* TEST INFRASTRUCTURE ONLY -- runtime side of oracle/hipstub/cuda_runtime.h (see there). Compiled by hipcc. */
#include <cuda_runtime.h>

#include <cstdlib>
#include <cstring>

extern "C" {
static cudaError_t rc(hipError_t e) { return e == hipSuccess ? cudaSuccess : cudaErrorUnknown; }
static void sync() { (void)hipDeviceSynchronize(); }

cudaError_t cudaMalloc(void** p, size_t n)
{
    if (hipHostMalloc(p, n ? n : 1, hipHostMallocDefault) != hipSuccess) return cudaErrorUnknown;
    memset(*p, 0, n);
    return cudaSuccess;
}
cudaError_t cudaMallocHost(void** p, size_t n) { return cudaMalloc(p, n + 4096); }
cudaError_t cudaFree(void* p) { sync(); return p ? rc(hipHostFree(p)) : cudaSuccess; }
cudaError_t cudaFreeHost(void* p) { return cudaFree(p); }
cudaError_t cudaHostRegister(void*, size_t, unsigned) { return cudaSuccess; }
cudaError_t cudaHostUnregister(void*) { return cudaSuccess; }
cudaError_t cudaMemcpy(void* d, const void* s, size_t n, enum cudaMemcpyKind) { sync(); memmove(d, s, n); return cudaSuccess; }
cudaError_t cudaMemcpyAsync(void* d, const void* s, size_t n, enum cudaMemcpyKind k, cudaStream_t) { return cudaMemcpy(d, s, n, k); }
cudaError_t cudaMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t w, size_t h, enum cudaMemcpyKind, cudaStream_t)
{
    sync();
    for (size_t y = 0; y < h; y++) memmove((char*)d + y * dpitch, (const char*)s + y * spitch, w);
    return cudaSuccess;
}
cudaError_t cudaMemset(void* d, int v, size_t n) { sync(); memset(d, v, n); return cudaSuccess; }
cudaError_t cudaMemsetAsync(void* d, int v, size_t n, cudaStream_t) { return cudaMemset(d, v, n); }
cudaError_t cudaStreamSynchronize(cudaStream_t) { return rc(hipDeviceSynchronize()); }
cudaError_t cudaDeviceSynchronize(void) { return rc(hipDeviceSynchronize()); }
cudaError_t cudaDeviceReset(void) { return cudaSuccess; }
cudaError_t cudaGetLastError(void) { return rc(hipGetLastError()); }
const char* cudaGetErrorString(cudaError_t e) { return e == cudaSuccess ? "no error" : "HIP error (hipstub)"; }
cudaError_t cudaGetDevice(int* d) { return rc(hipGetDevice(d)); }
cudaError_t cudaSetDevice(int d) { return rc(hipSetDevice(d)); }
cudaError_t cudaGetDeviceCount(int* c) { return rc(hipGetDeviceCount(c)); }
cudaError_t cudaDriverGetVersion(int* v) { return rc(hipDriverGetVersion(v)); }
cudaError_t cudaRuntimeGetVersion(int* v) { return rc(hipRuntimeGetVersion(v)); }
cudaError_t cudaGetDeviceProperties(struct cudaDeviceProp* p, int d)
{
    hipDeviceProp_t h;
    if (hipGetDeviceProperties(&h, d) != hipSuccess) return cudaErrorUnknown;
    memset(p, 0, sizeof *p);
    strncpy(p->name, h.name, sizeof p->name - 1);
    p->major = h.major; p->minor = h.minor;
    p->totalGlobalMem = h.totalGlobalMem; p->totalConstMem = h.totalConstMem; p->sharedMemPerBlock = h.sharedMemPerBlock;
    p->regsPerBlock = h.regsPerBlock; p->multiProcessorCount = h.multiProcessorCount;
    return cudaSuccess;
}
cudaError_t cudaEventCreate(cudaEvent_t* e) { *e = (cudaEvent_t)calloc(1, sizeof **e); return cudaSuccess; }
cudaError_t cudaEventDestroy(cudaEvent_t e) { free(e); return cudaSuccess; }
cudaError_t cudaEventRecord(cudaEvent_t, cudaStream_t) { return cudaSuccess; }
cudaError_t cudaEventSynchronize(cudaEvent_t) { sync(); return cudaSuccess; }
cudaError_t cudaEventElapsedTime(float* ms, cudaEvent_t, cudaEvent_t) { *ms = 0.0f; return cudaSuccess; }
}
