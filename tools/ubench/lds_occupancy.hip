// Workgroups per CU for a 256-thread kernel as a function of its LDS bytes: what the runtime's occupancy query reports, and what the
// hardware does (3000 workgroups that each note their start, spin ~30 us and note their end: resident = those that start before the
// first one ends).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void k(unsigned long long* t)
{
    extern __shared__ uint32_t s[];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 3000) {} // 100 MHz: 30 us
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = t0; t[blockIdx.x * 2 + 1] = wall_clock64() + s[255 - threadIdx.x] * 0; }
}
int main()
{
    const int n = 3000;
    unsigned long long* d;
    (void)hipMalloc(&d, n * 16);
    std::vector<unsigned long long> h(n * 2);
    int last = -1;
    for (int b = 16384; b <= 66000; b += 16) {
        int q = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, k, 256, b);
        if (q != last) printf("runtime query: LDS %6d B -> %d workgroups per CU\n", b, q);
        last = q;
    }
    const int sizes[] = {20480, 23392, 26000, 27296, 30000, 31744, 32000, 32256, 32688, 32768, 33000, 36000, 39504, 40960};
    for (int b : sizes) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipLaunchKernelGGL(k, dim3(n), dim3(256), b, 0, d);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
        unsigned long long first_end = ~0ull;
        for (int i = 0; i < n; i++) first_end = std::min(first_end, h[2 * i + 1]);
        int resident = 0;
        for (int i = 0; i < n; i++) resident += h[2 * i] < first_end;
        printf("hardware: LDS %6d B -> %4d workgroups resident at once = %.2f per CU\n", b, resident, resident / 256.0);
    }
    return 0;
}
