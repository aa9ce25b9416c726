// TEST INFRASTRUCTURE ONLY -- not part of libgpujpeg.so.
// The fp32 colour transform of the product's fused kernels (gj_device.h: gj_color_row, header-only) applied to rows of 8 packed pixels, so that
// tests/test_gpu_parity.py::test_exhaustive_colour_transform_fused can push all 2^24 input triples through it. Built by
// gpujpeg_amd/csrc/Makefile into gpujpeg_amd/lib/libgj_testhooks.so from the same header the product kernels include.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../gpujpeg_amd/csrc/gj_device.h"

template <int CS_FROM, int CS_TO>
__global__ __launch_bounds__(256) void k_test_color444(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const uint32_t nrows)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nrows) return;
    uint32_t px[6], o0[2], o1[2], o2[2];
    const uint2* p = reinterpret_cast<const uint2*>(in + (size_t)i * 24);
    const uint2 a = p[0], b = p[1], c = p[2];
    px[0] = a.x; px[1] = a.y; px[2] = b.x; px[3] = b.y; px[4] = c.x; px[5] = c.y;
    gj_color_row<CS_FROM, CS_TO>(px, o0, o1, o2);
    const size_t plane = (size_t)nrows * 8;
    *reinterpret_cast<uint2*>(out + (size_t)i * 8) = make_uint2(o0[0], o0[1]);
    *reinterpret_cast<uint2*>(out + plane + (size_t)i * 8) = make_uint2(o1[0], o1[1]);
    *reinterpret_cast<uint2*>(out + 2 * plane + (size_t)i * 8) = make_uint2(o2[0], o2[1]);
}

extern "C" __attribute__((visibility("default"))) int gj_test_color444(int cs_from, int cs_to, const uint8_t* d_in, uint8_t* d_out, uint32_t nrows, gj_stream_t stream)
{
    void (*k)(const uint8_t*, uint8_t*, uint32_t) = nullptr;
    if (cs_from == GJ_CS_RGB && cs_to == GJ_CS_BT601_256) k = k_test_color444<GJ_CS_RGB, GJ_CS_BT601_256>;
    if (cs_from == GJ_CS_RGB && cs_to == GJ_CS_BT601) k = k_test_color444<GJ_CS_RGB, GJ_CS_BT601>;
    if (cs_from == GJ_CS_RGB && cs_to == GJ_CS_BT709) k = k_test_color444<GJ_CS_RGB, GJ_CS_BT709>;
    if (cs_from == GJ_CS_BT601_256 && cs_to == GJ_CS_RGB) k = k_test_color444<GJ_CS_BT601_256, GJ_CS_RGB>;
    if (cs_from == GJ_CS_BT601 && cs_to == GJ_CS_RGB) k = k_test_color444<GJ_CS_BT601, GJ_CS_RGB>;
    if (cs_from == GJ_CS_BT709 && cs_to == GJ_CS_RGB) k = k_test_color444<GJ_CS_BT709, GJ_CS_RGB>;
    if (cs_from == cs_to) k = k_test_color444<GJ_CS_NONE, GJ_CS_NONE>;
    if (!k) return -1;
    hipLaunchKernelGGL(k, dim3((nrows + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_in, d_out, nrows);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

