// tools/ubench/host_link_patterns.hip -- how must N synchronous "coders" (a host thread each: upload a frame, run a kernel, download a result, wait)
// issue their copies for the host link to carry both directions at once? Frame = 99.5 MB (8K RGB) up and 99.5 MB down per iteration and thread
// (an encoder's upload and a decoder's download), pinned buffers.
//   P0  every thread copies on its own stream (the library before round 5)
//   P1  one upload stream + one download stream for the process, tied into the thread's stream with events (hipStreamWaitEvent both ways)
//   P2  the same two streams, but the THREAD waits (hipEventSynchronize) between upload, kernel and download: no cross-stream waits on the device
//   P3  every thread has its own upload, compute and download stream, events between them
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/hlp tools/ubench/host_link_patterns.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_touch(uint32_t* p, size_t n) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] += 1; }
static const size_t N = 3840ull * 2160 * 3; // (a 4K frame: the size of BASELINE config 5)
static hipStream_t g_up, g_down, g_up2, g_down2;
struct Coder {
    uint8_t *h_in, *h_out, *d_a, *d_b;
    hipStream_t s, su, sd;
    hipEvent_t e0, e1, e2, e3;
};
static void work(Coder& c, int pattern, int iters)
{
    for (int i = 0; i < iters; i++) {
        switch (pattern) {
        case 0:
            CK(hipMemcpyAsync(c.d_a, c.h_in, N, hipMemcpyHostToDevice, c.s));
            hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, c.s, (uint32_t*)c.d_a, N / 4);
            CK(hipMemcpyAsync(c.h_out, c.d_b, N, hipMemcpyDeviceToHost, c.s));
            CK(hipStreamSynchronize(c.s));
            break;
        case 1:
        case 3: {
            hipStream_t up = pattern == 1 ? g_up : c.su, down = pattern == 1 ? g_down : c.sd;
            CK(hipEventRecord(c.e0, c.s)); CK(hipStreamWaitEvent(up, c.e0, 0));
            CK(hipMemcpyAsync(c.d_a, c.h_in, N, hipMemcpyHostToDevice, up));
            CK(hipEventRecord(c.e1, up)); CK(hipStreamWaitEvent(c.s, c.e1, 0));
            hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, c.s, (uint32_t*)c.d_a, N / 4);
            CK(hipEventRecord(c.e2, c.s)); CK(hipStreamWaitEvent(down, c.e2, 0));
            CK(hipMemcpyAsync(c.h_out, c.d_b, N, hipMemcpyDeviceToHost, down));
            CK(hipEventRecord(c.e3, down)); CK(hipStreamWaitEvent(c.s, c.e3, 0));
            CK(hipStreamSynchronize(c.s));
            break;
        }
        case 2:
            CK(hipMemcpyAsync(c.d_a, c.h_in, N, hipMemcpyHostToDevice, g_up));
            CK(hipEventRecord(c.e1, g_up)); CK(hipEventSynchronize(c.e1));
            hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, c.s, (uint32_t*)c.d_a, N / 4);
            CK(hipStreamSynchronize(c.s));
            CK(hipMemcpyAsync(c.h_out, c.d_b, N, hipMemcpyDeviceToHost, g_down));
            CK(hipEventRecord(c.e3, g_down)); CK(hipEventSynchronize(c.e3));
            break;
        case 4: // P2 + what a coder does between its two images: a compressed stream (1/13 of the image) down and up again on its OWN stream
        case 5: // the same with the small copies through the lanes too
        case 6: { // the same with the small copies on a second pair of process-wide streams
            hipStream_t sd = pattern == 4 ? c.s : pattern == 5 ? g_down : g_down2, su = pattern == 4 ? c.s : pattern == 5 ? g_up : g_up2;
            CK(hipMemcpyAsync(c.d_a, c.h_in, N, hipMemcpyHostToDevice, g_up));
            CK(hipEventRecord(c.e1, g_up)); CK(hipEventSynchronize(c.e1));
            hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, c.s, (uint32_t*)c.d_a, N / 4);
            CK(hipStreamSynchronize(c.s));
            CK(hipMemcpyAsync(c.h_out, c.d_b, N / 13, hipMemcpyDeviceToHost, sd));
            CK(hipEventRecord(c.e0, sd)); CK(hipEventSynchronize(c.e0));
            CK(hipMemcpyAsync(c.d_a, c.h_in, N / 13, hipMemcpyHostToDevice, su));
            CK(hipEventRecord(c.e2, su)); CK(hipEventSynchronize(c.e2));
            hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, c.s, (uint32_t*)c.d_a, N / 4);
            CK(hipStreamSynchronize(c.s));
            CK(hipMemcpyAsync(c.h_out, c.d_b, N, hipMemcpyDeviceToHost, g_down));
            CK(hipEventRecord(c.e3, g_down)); CK(hipEventSynchronize(c.e3));
            break;
        }
        }
    }
}
int main(int argc, char** argv)
{
    CK(hipStreamCreateWithFlags(&g_up, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&g_down, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&g_up2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&g_down2, hipStreamNonBlocking));
    const int T = 4;
    std::vector<Coder> cs(T);
    for (auto& c : cs) {
        CK(hipHostMalloc((void**)&c.h_in, N, hipHostMallocDefault)); CK(hipHostMalloc((void**)&c.h_out, N, hipHostMallocDefault));
        CK(hipMalloc((void**)&c.d_a, N)); CK(hipMalloc((void**)&c.d_b, N));
        CK(hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&c.su, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&c.sd, hipStreamNonBlocking));
        CK(hipEventCreateWithFlags(&c.e0, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&c.e1, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&c.e2, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&c.e3, hipEventDisableTiming));
    }
    const int only = argc > 1 ? atoi(argv[1]) : -1; // (one pattern, four threads, five repetitions: is it the same every time?)
    for (int pattern = 0; pattern < 7; pattern++)
        for (int threads = only >= 0 ? T : 1; threads <= T; threads++) {
            if (only >= 0 && pattern != only) continue;
            for (int rep = 0; rep < (only >= 0 ? 6 : 2); rep++) { // (first repetition: warm-up)
                const int iters = 64;
                const auto t0 = std::chrono::steady_clock::now();
                std::vector<std::thread> th;
                for (int t = 0; t < threads; t++) th.emplace_back(work, std::ref(cs[t]), pattern, iters);
                for (auto& t : th) t.join();
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (rep) printf("P%d  %d thread(s): %6.1f GB/s each way (%5.0f frames/s)\n", pattern, threads, threads * iters * (double)N / dt / 1e9, threads * iters / dt);
            }
        }
    return 0;
}
