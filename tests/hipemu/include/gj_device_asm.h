// gj_device_asm.h of the CPU execution model: the C++ meaning of the single-instruction helpers that the product writes as gfx950 inline
// assembly (gpujpeg_amd/csrc/gj_device_asm.h). Found first on the include path of tests/hipemu/Makefile; test tier only.
#pragma once
#include <stdint.h>

#define GJ_KEEP(x) ((void)0)
#define GJ_KEEP6(a, b, c, d, e, f) ((void)0)

template <int OFF, int WIDTH> __device__ __forceinline__ uint32_t gj_bfe_u32(uint32_t v) { return (v >> OFF) & ((1u << WIDTH) - 1u); }

__device__ __forceinline__ int gj_ffbh_i32(int v)
{
    const uint32_t x = (uint32_t)(v ^ (v >> 31)); // leading sign bits become leading zeros
    return x ? __builtin_clz(x) : -1;
}

__device__ __forceinline__ uint32_t gj_pk_min_u16(uint32_t a, uint32_t b)
{
    const uint32_t lo = (a & 0xFFFFu) < (b & 0xFFFFu) ? (a & 0xFFFFu) : (b & 0xFFFFu), hi = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16);
    return lo | (hi << 16);
}

typedef float gj_f2 __attribute__((ext_vector_type(2)));

template <int K> __device__ __forceinline__ float gj_ubyte_f(uint32_t w) { return (float)((w >> (8 * K)) & 0xFFu); }
template <int K> __device__ __forceinline__ float gj_ubyte_f_opaque(uint32_t w) { return gj_ubyte_f<K>(w); }

// v_pk_add_f32 ... clamp: the sum clamped to [0, 1]
__device__ __forceinline__ gj_f2 gj_scale256_f(gj_f2 v)
{
    gj_f2 d = v + (gj_f2)-254.0f;
    d.x = d.x < 0.0f ? 0.0f : (d.x > 1.0f ? 1.0f : d.x);
    d.y = d.y < 0.0f ? 0.0f : (d.y > 1.0f ? 1.0f : d.y);
    return v + d;
}

template <int N> __device__ __forceinline__ uint32_t gj_lshl_add_u32(uint32_t a, uint32_t b) { return (a << N) + b; }
