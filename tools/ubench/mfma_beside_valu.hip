// Does a small matrix instruction issue BESIDE the vector ALU work of the other waves of its SIMD, or in its place? (round 5: the colour matrix
// of the encoder moved to v_mfma_f32_4x4x1 and the kernel got slower although it issued 9 % fewer vector instructions.)
// Every wave runs ITER x [NV packed FMAs + NM matrix instructions of one kind]; 1024 workgroups of 256, four waves per SIMD like k_encode_rgb444.
// Prints the time of the vector part alone, of the matrix part alone and of both: both ~ max = the pipes run side by side, both ~ sum = they do not.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_beside_valu.hip -o gpujpeg_amd/lib/mfma_beside_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
#define ITER 512
template <int KIND, int NV, int NM>
__global__ __launch_bounds__(256, 4) void k(float* out, float s)
{
    float2 a[8];
    for (int i = 0; i < 8; i++) a[i] = make_float2(s + threadIdx.x + i, s * i);
    f4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = f4{s, s, s, s};
    const float fa = s + (threadIdx.x & 3), fb = s * threadIdx.x;
    const h4 ha = {(_Float16)fa, (_Float16)fb, (_Float16)s, (_Float16)1.0f}, hb = {(_Float16)fb, (_Float16)fa, (_Float16)s, (_Float16)2.0f};
    const int ia = (int)threadIdx.x * 0x01010101, ib = (int)(threadIdx.x + 3) * 0x01020304;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
#pragma unroll
            for (int v = 0; v < NV / 16; v++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[(q + v) & 7]) : "v"(a[(q + v + 1) & 7]));
#pragma unroll
            for (int m = 0; m < NM / 16; m++) {
                if (KIND == 0) acc[(q + m) & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(fa, fb, acc[(q + m) & 3], 0, 0, 0);
                if (KIND == 1) acc[(q + m) & 3] = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, acc[(q + m) & 3], 0, 0, 0);
                if (KIND == 2) {
                    typedef int i4 __attribute__((ext_vector_type(4)));
                    i4 c = __builtin_bit_cast(i4, acc[(q + m) & 3]);
                    c = __builtin_amdgcn_mfma_i32_4x4x4i8(ia, ib, c, 0, 0, 0);
                    acc[(q + m) & 3] = __builtin_bit_cast(f4, c);
                }
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 8; i++) r += a[i].x + a[i].y;
    for (int i = 0; i < 4; i++) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (r == 12345.678f) out[threadIdx.x] = r;
}
template <int KIND, int NV, int NM>
static float run(float* d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NV, NM>), dim3(1024), dim3(256), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<KIND, NV, NM>), dim3(1024), dim3(256), 0, 0, d, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1000.0f;
}
int main()
{
    float* d;
    hipMalloc(&d, 4096);
    const char* names[3] = {"v_mfma_f32_4x4x1_16b_f32", "v_mfma_f32_4x4x4_16b_f16", "v_mfma_i32_4x4x4_16b_i8"};
    const float valu = run<0, 64, 0>(d);
    printf("per wave and iteration: 64 v_pk_fma_f32 alone: %.1f us (= %.2f cycles per instruction and SIMD at 2.4 GHz, 4 waves per SIMD)\n", valu,
           valu * 2400.0 / (ITER * 64.0 * 4.0));
    const float m0 = run<0, 0, 32>(d), m1 = run<1, 0, 32>(d), m2 = run<2, 0, 32>(d);
    const float b0 = run<0, 64, 32>(d), b1 = run<1, 64, 32>(d), b2 = run<2, 64, 32>(d);
    const float ms[3] = {m0, m1, m2}, bs[3] = {b0, b1, b2};
    for (int kd = 0; kd < 3; kd++)
        printf("%-26s 32 alone: %7.1f us (%.2f cycles each)   beside the 64 packed FMAs: %7.1f us   (sum %.1f, max %.1f)\n", names[kd], ms[kd],
               ms[kd] * 2400.0 / (ITER * 32.0 * 4.0), bs[kd], valu + ms[kd], valu > ms[kd] ? valu : ms[kd]);
    const float c0 = run<0, 64, 16>(d), c1 = run<1, 64, 16>(d), c2 = run<2, 64, 16>(d);
    printf("16 matrix instructions beside the 64 packed FMAs: f32 %.1f us, f16 %.1f us, i8 %.1f us\n", c0, c1, c2);
    return 0;
}
