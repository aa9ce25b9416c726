// gj_bitreader.h -- what the entropy decoders of gj_dec_entropy_*.hip share on the input side: restart segments are copied into an LDS
// stage with the stuffed zeros removed (big-endian dwords, so that a bit position is a dword index and a shift), and symbols are looked
// up in the two-level tables of gj_hip.h (GJ_DEC2_*).
#pragma once
#include "gj_device.h"

// unstuffs one segment into the stage with one wave; w0 = the lane's dword of the segment's first 256 B (zero behind its end)
__device__ __forceinline__ uint32_t gj_unstuff_segment(const uint8_t* __restrict__ jpeg, const uint32_t* __restrict__ end, const uint32_t pos, const uint32_t len,
                                                       uint32_t* __restrict__ stage, const uint32_t ubase, const int lane, const uint32_t w0)
{
    uint8_t* U8 = reinterpret_cast<uint8_t*>(stage);
    const uintptr_t a = reinterpret_cast<uintptr_t>(jpeg) + pos;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
    const int lead = (int)(a & 3);
    const uint32_t ndw = ((uint32_t)lead + len + 3u) >> 2;
    uint32_t out = 0;
    bool copied = false;
    if (ndw <= 64u) { // no stuffed byte: a shifted, byte-swapped copy (see k_huffman_decode_par)
        const uint32_t pw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w0, 0x138, 0xF, 0xF, false);
        const uint32_t pb = __builtin_amdgcn_alignbit(w0, pw, 24);
        const uint32_t hit = (w0 - 0x01010101u) & ~w0 & (~pb - 0x01010101u) & pb & 0x80808080u;
        if (__ballot(hit != 0u && (uint32_t)lane < ndw) == 0ull) {
            const uint32_t wn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w0, 0x130, 0xF, 0xF, false);
            uint32_t d = __builtin_bswap32(__builtin_amdgcn_alignbyte(wn, w0, (uint32_t)lead));
            const uint32_t full = len >> 2, rest = len & 3u;
            if ((uint32_t)lane == full && rest) d &= 0xFFFFFFFFu << (32u - 8u * rest);
            if ((uint32_t)lane < full + (rest ? 1u : 0u)) stage[(ubase >> 2) + (uint32_t)lane] = d;
            out = len;
            copied = true;
        }
    }
    uint32_t carry = 0;
    for (uint32_t c0 = 0; !copied && c0 < ndw; c0 += 64) {
        const uint32_t idx = c0 + (uint32_t)lane;
        uint32_t w = w0;
        if (c0) {
            w = 0;
            if (idx < ndw && src + idx < end) w = src[idx];
        }
        uint32_t pw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x138, 0xF, 0xF, false); // wave_shr:1
        if (lane == 0) pw = carry;
        uint32_t prev = pw >> 24;
        uint32_t keep = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t b = (w >> (8 * k)) & 0xFFu;
            const int off = (int)(idx * 4u) + k - lead;
            const bool valid = off >= 0 && off < (int)len;
            const bool stuffed = b == 0 && prev == 0xFFu && off > 0;
            if (valid && !stuffed) keep |= 1u << k;
            prev = b;
        }
        const uint32_t cnt = (uint32_t)__popc(keep);
        const uint32_t inc = gj_wave_incl_scan(cnt);
        uint32_t o = ubase + out + inc - cnt;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (keep & (1u << k)) { U8[o ^ 3u] = (uint8_t)(w >> (8 * k)); o++; }
        out += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        carry = (uint32_t)__builtin_amdgcn_readlane((int)w, 63);
    }
    for (uint32_t b = out + (uint32_t)lane; b < ((out + 3u) & ~3u) + 8u; b += 64) U8[(ubase + b) ^ 3u] = 0;
    return out;
}
