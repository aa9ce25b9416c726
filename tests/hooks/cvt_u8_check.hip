// TEST INFRASTRUCTURE ONLY. v_cvt_pk_u8_f32 against clamp(rintf(x)) for every float the kernels can feed it (all multiples of 1/512
// in [-1024, 1024] plus a sweep of arbitrary values): hipcc --offload-arch=gfx950 cvt_u8_check.hip -o cvt_u8_check && ./cvt_u8_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

__global__ void k(unsigned* bad, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = i < (1 << 20) ? (i - (1 << 19)) / 512.0f : (i - (1 << 20) - (1 << 19)) * 0.0031415926f;
    const unsigned got = __builtin_amdgcn_cvt_pk_u8_f32(x, 0, 0u);
    float r = rintf(x);
    r = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
    if (got != (unsigned)r) atomicAdd(bad, 1u);
}

int main()
{
    unsigned* d;
    unsigned h = 0;
    if (hipMalloc(&d, 4) != hipSuccess) return 2;
    (void)hipMemset(d, 0, 4);
    const int n = 2 << 20;
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, d, n);
    (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("v_cvt_pk_u8_f32 vs clamp(rintf): %u mismatches of %d\n", h, n);
    return h != 0;
}
