// experiment: DPP wave scan against a serial prefix sum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../gpujpeg_amd/csrc/gj_device.h"
__global__ void k(const uint32_t* in, uint32_t* out) { out[threadIdx.x + blockIdx.x * 64] = gj_wave_incl_scan(in[threadIdx.x + blockIdx.x * 64]); }
int main()
{
    const int n = 64 * 1000;
    uint32_t *h = new uint32_t[n], *r = new uint32_t[n], *d, *o;
    uint32_t s = 7;
    for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = (i % 640 < 64) ? 1u : (s >> 20); }
    (void)hipMalloc(&d, n * 4); (void)hipMalloc(&o, n * 4);
    (void)hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    k<<<n / 64, 64>>>(d, o);
    (void)hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < n / 64; b++) { uint32_t acc = 0; for (int i = 0; i < 64; i++) { acc += h[b * 64 + i]; if (r[b * 64 + i] != acc) { if (bad < 5) printf("wave %d lane %d got %u want %u\n", b, i, r[b * 64 + i], acc); bad++; } } }
    printf("dpp scan mismatches: %d\n", bad);
    return 0;
}
