/*
 * TEST INFRASTRUCTURE ONLY -- known-answer kernels for cudaemu's WARP MODE (oracle/cudaemu/cudaemu.cpp), written for this repository in the pre-Volta
 * warp-synchronous style the reference's Huffman encoder uses (src/gpujpeg_huffman_gpu_encoder.cu:192-294): votes, and values handed from lane to lane
 * through shared memory with NO barrier between the stores and the loads. Under a model that ran the lanes one after the other (cudaemu's plain mode)
 * every one of them gives wrong results; tests/test_oracle_vs_ref.py::test_cudaemu_warp_mode_known_answers compares with closed forms.
 * Compiled like the reference's modules: the launch syntax rewritten by the perl pipe of oracle/Makefile, -DCUDAEMU_WARP -fsanitize=thread.
 */
#include <cuda_runtime.h>
#include <stdint.h>

/* 1. inclusive prefix sum of a warp by log-step exchange through shared memory, no barrier anywhere (Hillis-Steele on s[]) */
__global__ static void k_scan_no_barrier(const unsigned* in, unsigned* out)
{
    __shared__ unsigned s[4][32];
    const int lane = threadIdx.x & 31, warp = threadIdx.x >> 5;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned v = in[g];
    for (int d = 1; d < 32; d <<= 1) {
        s[warp][lane] = v;               /* every lane stores ... */
        if (lane >= d) v += s[warp][lane - d]; /* ... and reads a NEIGHBOUR's store of this very step: lock step or nothing */
    }
    out[g] = v;
}

/* 2. stream compaction with votes: lanes whose value is odd append it, in lane order, to the warp's output; the count comes from the vote */
__global__ static void k_compact_by_vote(const unsigned* in, unsigned* out, unsigned* count)
{
    const int lane = threadIdx.x & 31, warp = threadIdx.x >> 5;
    const int w = blockIdx.x * (blockDim.x >> 5) + warp;
    const unsigned v = in[w * 32 + lane];
    const unsigned m = __ballot_sync(0xffffffffu, v & 1u);
    if (v & 1u) out[w * 32 + __popc(m & ((1u << lane) - 1u))] = v;
    if (lane == 0) count[w] = __popc(m);
}

/* 3. divergence + early exit + a block barrier: odd warps leave at once, the others rotate their values by one lane through shared memory (no barrier),
 *    then all remaining warps meet at __syncthreads() and lane 0 of warp 0 sums one value per warp; votes after a partial exit see only the live lanes */
__global__ static void k_rotate_exit_barrier(const unsigned* in, unsigned* out, unsigned* sums, unsigned* votes)
{
    __shared__ unsigned s[4][32];
    __shared__ unsigned first[4];
    const int lane = threadIdx.x & 31, warp = threadIdx.x >> 5;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (warp & 1) { out[g] = 0xdeadu; return; }
    if (lane >= 24 && blockIdx.x == 1) { out[g] = 0xbeefu; return; } /* part of a warp leaves */
    const unsigned nlive = (blockIdx.x == 1) ? 24u : 32u;
    s[warp][lane] = in[g];
    const unsigned r = s[warp][(lane + 1) % nlive];
    out[g] = r;
    const unsigned m = __ballot_sync(0xffffffffu, 1);
    if (lane == 0) { first[warp] = r; votes[blockIdx.x * 4 + warp] = m; }
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = first[0] + first[2];
}

/* 4. atomics: every warp reserves room for its odd values in ONE shared output with atomicAdd (as the reference's compaction kernel does, :563-613) */
__global__ static void k_reserve_with_atomics(const unsigned* in, unsigned* out, unsigned* total)
{
    __shared__ unsigned base[4];
    const int lane = threadIdx.x & 31, warp = threadIdx.x >> 5;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned v = in[g];
    const unsigned m = __ballot_sync(0xffffffffu, v & 1u);
    if (lane == 0) base[warp] = atomicAdd(total, (unsigned)__popc(m));
    const unsigned b = base[warp]; /* lane 0's store, read by the other 31 without a barrier */
    if (v & 1u) out[b + __popc(m & ((1u << lane) - 1u))] = v;
}

extern "C" {
__attribute__((visibility("default"))) void cudaemu_selftest_scan(const unsigned* in, unsigned* out, int blocks)
{
    k_scan_no_barrier<<<blocks, 128, 0, 0>>>(in, out);
}
__attribute__((visibility("default"))) void cudaemu_selftest_compact(const unsigned* in, unsigned* out, unsigned* count, int blocks)
{
    k_compact_by_vote<<<blocks, 128, 0, 0>>>(in, out, count);
}
__attribute__((visibility("default"))) void cudaemu_selftest_rotate(const unsigned* in, unsigned* out, unsigned* sums, unsigned* votes, int blocks)
{
    k_rotate_exit_barrier<<<blocks, 128, 0, 0>>>(in, out, sums, votes);
}
__attribute__((visibility("default"))) void cudaemu_selftest_reserve(const unsigned* in, unsigned* out, unsigned* total, int blocks)
{
    k_reserve_with_atomics<<<blocks, 128, 0, 0>>>(in, out, total);
}
}
