// experiment: semantics of v_cvt_pk_u8_f32 against clamp(rintf(x), 0, 255)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdint>
__global__ void k(const float* x, uint32_t* out, int n)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0);
}
int main()
{
    const int n = 1 << 22;
    float* h = new float[n];
    int k0 = 0;
    for (int v = -400; v < 700 && k0 < n - 16; v++)
        for (float f : {0.0f, 0.5f, -0.5f, 0.25f, 0.75f, 0.49999997f, 0.50000006f, -0.49999997f, 0.99999f, 0.00001f, 0.4f, 0.6f})
            h[k0++] = (float)v + f;
    uint32_t s = 12345;
    for (; k0 < n; k0++) { s = s * 1664525u + 1013904223u; h[k0] = ((int)(s >> 8) % 1000000) / 1000.0f - 300.0f; }
    float* d; uint32_t* o; hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, o, n);
    uint32_t* r = new uint32_t[n];
    hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        float t = rintf(h[i]); int w = t < 0 ? 0 : t > 255 ? 255 : (int)t;
        if ((int)(r[i] & 255) != w) { if (bad < 10) printf("x=%.9g got %u want %d\n", h[i], r[i] & 255, w); bad++; }
    }
    printf("cvt_pk_u8_f32 mismatches vs clamp(rintf): %d of %d\n", bad, n);
    return 0;
}
