// gj_device_asm.h -- the helpers of gj_device.h that ARE single gfx950 instructions, written as inline assembly because the compiler
// does not select them (or selects them and then rewrites the code around them for the worse). Included by gj_device.h as
// <gj_device_asm.h>: the product build finds this file; the CPU execution model of the test tier (tests/hipemu) puts its own header of the
// same name, with the instructions' C++ meanings, in front of it on the include path -- nothing in the product's sources knows about that.
#pragma once
#include <stdint.h>

#define GJ_KEEP6(a, b, c, d, e, f) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f))
#define GJ_KEEP(x) asm volatile("" : "+v"(x)) // pins a value in its register here: a scheduling fence for the compiler

// bits [OFF, OFF + WIDTH) of v as the instruction itself: written as a shift the compiler folds it into the address arithmetic that
// follows and ends up with shift + mask + add where bit-field extract + shift-add do
template <int OFF, int WIDTH> __device__ __forceinline__ uint32_t gj_bfe_u32(uint32_t v)
{
    uint32_t r;
    asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "n"(OFF), "n"(WIDTH));
    return r;
}

// per-half minimum of two packed u16 pairs (the compiler scalarises the vector form, hence the instruction itself)
// v_ffbh_i32: the number of leading bits that equal the sign bit (31 for 1 and for -2; -1 when all 32 are alike). For t = v - (v < 0)
// of a non-zero v, 32 minus it is the JPEG magnitude category of v -- one instruction where |v| and a count of leading zeros take three
__device__ __forceinline__ int gj_ffbh_i32(int v)
{
    int r;
    asm("v_ffbh_i32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

__device__ __forceinline__ uint32_t gj_pk_min_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

typedef float gj_f2 __attribute__((ext_vector_type(2)));

// byte k of a dword as float: the AMDGPU back end selects v_cvt_f32_ubyte<k> for this pattern
template <int K> __device__ __forceinline__ float gj_ubyte_f(uint32_t w) { return (float)((w >> (8 * K)) & 0xFFu); }

// the same as the instruction itself, for the inputs of the transforms: from the C expression the optimiser learns that the value
// is a small integer and rewrites the first butterfly (float(a) + float(b)) into per-sample integer SDWA adds followed by 72
// conversions per block -- 216 scalar operations where 64 conversions + 32 packed adds do. (Not for the colour transform: there
// the compiler needs to know that the value is no signalling NaN, or every v_max_f32 gets a canonicalising twin.)
template <int K> __device__ __forceinline__ float gj_ubyte_f_opaque(uint32_t w)
{
    float r;
    if (K == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(w));
    else if (K == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(w));
    else if (K == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(w));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(w));
    return r;
}

// c * 256 / 255 for an integer c in [0, 255]: c + (c == 255). The indicator is the clamp-to-[0, 1] output modifier on c - 254 (two
// packed instructions per pixel pair; max(c, 256 c - 65024) costs a packed FMA and two v_max_f32, which have no packed form and
// issue at half the rate of an add, profiles/r2_09_ubench.txt).
__device__ __forceinline__ gj_f2 gj_scale256_f(gj_f2 v)
{
    gj_f2 d;
    asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(d) : "v"(v), "v"((gj_f2)-254.0f));
    return v + d;
}

// (a << N) + b as the one instruction it is (left to itself the compiler splits a table index of two fields into two shifts and a three-operand add)
template <int N> __device__ __forceinline__ uint32_t gj_lshl_add_u32(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(N), "v"(b));
    return r;
}
