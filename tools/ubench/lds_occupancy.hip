// Workgroups per CU the runtime reports for a 256-thread kernel as a function of its LDS bytes (allocation granularity of gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(uint32_t* o) { extern __shared__ uint32_t s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); o[threadIdx.x] = s[255 - threadIdx.x]; }
int main()
{
    int last = -1;
    for (int b = 16384; b <= 66000; b += 16) {
        int n = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, b);
        if (n != last) printf("LDS %6d B -> %d workgroups per CU\n", b, n);
        last = n;
    }
    return 0;
}
