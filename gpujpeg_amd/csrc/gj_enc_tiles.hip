// gj_enc_tiles.hip -- MI355X (gfx950, wave64) JPEG encoder: the fully fused encoders, pixels -> entropy-coded tile streams in one kernel
// (part of the encoder's device code, see gj_enc_internal.h for the map of the files)
//   k_encode_rgb444 / k_encode_uyvy422   raw packed pixels -> entropy-coded segments in one kernel (the default for the BASELINE
//                         configurations; no coefficient planes)
//   k_encode_blocks       the same for planar input and for RGB with any chroma sampling: one lane per block in coding order
// k_gather (gj_enc_assemble.hip) turns the tile streams into the file.
#include "gj_enc_internal.h"

// -DGJ_TRACE_PHASES (the `trace` target of the Makefile, tools/encoder_phases.py): the first work-item of every workgroup of the fused encoders
// notes the wall clock (100 MHz) at the phase boundaries in a buffer the tool hands over (16 slots per workgroup); the release build has none of it
#ifdef GJ_TRACE_PHASES
static __device__ unsigned long long* gj_trace_buf_e;
extern "C" GJ_HIP_API int gj_hip_trace_set_encoder(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(gj_trace_buf_e), &p, sizeof p) == hipSuccess ? 0 : -1; }
// (gj_hip_trace_stop_encoder(n): every wave ends at stamp n -- the vector instructions of the phases in front of it are what SQ_INSTS_VALU then
// counts, tools/encoder_valu_budget.py takes the differences; the streams of such a launch are garbage)
static __device__ int gj_trace_stop_e = 1 << 30;
extern "C" GJ_HIP_API int gj_hip_trace_stop_encoder(int n) { return hipMemcpyToSymbol(HIP_SYMBOL(gj_trace_stop_e), &n, sizeof n) == hipSuccess ? 0 : -1; }
#define GJ_TRACE_E(slot) do { if (threadIdx.x == 0 && gj_trace_buf_e) gj_trace_buf_e[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
                              if ((slot) >= gj_trace_stop_e) __builtin_amdgcn_endpgm(); } while (0)
#else
#define GJ_TRACE_E(slot) ((void)0)
#endif

// ================================================================================================
// The coder of the fully fused encoder kernels (k_encode_rgb444, k_encode_uyvy422): one LANE per 8x8 block, 256 block slots per
// workgroup tile, whole restart segments per tile.
//
//   1. the transform stores every quantised coefficient as 16 bits straight to its ZIG-ZAG position in the lane's own LDS column
//      (ds_write_b16 with immediate offsets, layout [z & 31][lane] dwords, half z >> 5): no packing, no reordering pass;
//   2. the lane reads its column back as 32 dwords and forms the 64-bit non-zero mask (v_pk_min_u16 + v_lshl_or_b32 per dword, two
//      v_perm_b32 at the end);
//   3. ONE walk over the non-zero coefficients (mask + ctz) produces the block's bit stream privately: symbols go into a 64-bit
//      register accumulator, every completed dword is stored IN PLACE over the part of the lane's column the walk has already
//      consumed (the halves of dword f go to positions 2f and 2f + 1 once both are behind the walk; true for anything but blocks
//      that average more than 16 bits per coefficient position), the last partial dword stays in a register;
//   4. prefix sums over the block lengths give exact bit positions inside per-segment streams; the rows of the coefficient area
//      above GJ_ENC_PRIV_ROWS become the shared bit window (nothing else lives in LDS: 36 KB per workgroup, four per CU);
//   5. every lane shift-merges its private dwords into the window (ds_or_b32), coalesced copy of the unstuffed segment streams to
//      d_temp with byte and 0xFF counts per segment (k_scan_segments / k_assemble finish the stream).
//
// A block whose stream does not fit in place (noise at q100) continues it in its own slot of d_temp and reads it back for the
// merge; a tile whose streams exceed the window takes several windows.
// Symbol semantics restate src/gpujpeg_huffman_gpu_encoder.cu:139-294 / src/gpujpeg_huffman_cpu_encoder.c:136-246.
// ================================================================================================
#define GJ_ENC_PRIV_ROWS 24                          // rows of the coefficient area whose lower halves may hold private streams (12 dwords per block)
#define GJ_ENC_WIN_DW ((32 - GJ_ENC_PRIV_ROWS) * 256) // shared bit window: the remaining rows, 2048 dwords

// natural (row-major) index -> position in the zig-zag scan (inverse of GJ_ZZ)
__device__ static constexpr uint8_t GJ_IZZ[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
                                                  41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
                                                  46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
// byte offset of zig-zag position z inside a lane's column (column base = lane * 4): dword row z & 31, half z >> 5 -- a walk over
// the lower (upper) 32 positions addresses row * 1024 (+ 2) with one shift-add
#define GJ_COL_OFF(z) (((z) & 31) * 1024 + ((z) >> 5) * 2)

// gj_fdct_quant_pk with the stores of step 1: `col` = this lane's column base in LDS (bytes)
__device__ __forceinline__ void gj_fdct_quant_zz(const uint32_t (&px)[16], const float* __restrict__ q, uint8_t* col)
{
    gj_f2 D[8][4];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t a = px[2 * r], b = px[2 * r + 1];
        D[r][0] = gj_f2{gj_ubyte_f_opaque<0>(a), gj_ubyte_f_opaque<1>(a)};
        D[r][1] = gj_f2{gj_ubyte_f_opaque<2>(a), gj_ubyte_f_opaque<3>(a)};
        D[r][2] = gj_f2{gj_ubyte_f_opaque<0>(b), gj_ubyte_f_opaque<1>(b)};
        D[r][3] = gj_f2{gj_ubyte_f_opaque<2>(b), gj_ubyte_f_opaque<3>(b)};
    }
#pragma unroll
    for (int c = 0; c < 4; c++) gj_fdct8<gj_f2>(D[0][c], D[1][c], D[2][c], D[3][c], D[4][c], D[5][c], D[6][c], D[7][c], -1024.0f);
    __builtin_amdgcn_sched_barrier(0);
    const gj_f2* q2 = reinterpret_cast<const gj_f2*>(q);
#pragma unroll
    for (int rp = 0; rp < 4; rp++) {
        gj_f2 E[8];
#pragma unroll
        for (int cp = 0; cp < 4; cp++) {
            E[2 * cp] = gj_f2{D[2 * rp][cp].x, D[2 * rp + 1][cp].x};
            E[2 * cp + 1] = gj_f2{D[2 * rp][cp].y, D[2 * rp + 1][cp].y};
        }
        gj_fdct8<gj_f2>(E[0], E[1], E[2], E[3], E[4], E[5], E[6], E[7], 0.0f);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            // rintf(coef * q) by adding 1.5 * 2^23: the integer sits in the low mantissa bits, its low 16 bits are the int16
            const gj_f2 u = E[j] * q2[j * 4 + rp] + (gj_f2)12582912.0f;
            const float fx = u.x, fy = u.y;
            *reinterpret_cast<uint16_t*>(col + GJ_COL_OFF(GJ_IZZ[(2 * rp) * 8 + j])) = (uint16_t)__builtin_bit_cast(uint32_t, fx);
            *reinterpret_cast<uint16_t*>(col + GJ_COL_OFF(GJ_IZZ[(2 * rp + 1) * 8 + j])) = (uint16_t)__builtin_bit_cast(uint32_t, fy);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

struct GjCoderLds {
    uint32_t* coef;      // [32][256]; rows GJ_ENC_PRIV_ROWS.. double as the shared bit window once the walks are done
    const uint32_t* lut; // [2][272]: per table type AC[(run << 4) | ((16 - nbits) & 15)] then DC[nbits], entry = (code bits + nbits) << 26 | code << nbits
    uint32_t* wsum;      // [4] block-length totals of the waves
    int* edge;           // [4][16] the last sixteen DC terms of each wave (predecessors of the next wave's first lanes)
    uint32_t *segx, *segend, *segbase, *segbits, *segff; // [64] ([65] segbase)
};

// the private stream of a lane while it walks its block
struct GjWalk {
    uint32_t hi;     // accumulator: 64 - room < 32 bits, left-aligned
    int room;        // 64 - the bits in the accumulator: what a code word is shifted left by, less its own length (kept in this form: one subtraction
                     // per symbol where "fill += n; shift = 64 - fill" takes two)
    int produced;    // completed dwords so far
    int stored;      // once a dword has gone to the block's d_temp slot (lim == GJ_ENC_NO_STORE): the first `stored` dwords sit in the lane's column,
                     // the others in the slot; before that every completed dword is in the column (gj_walk_stored)
    int lim;         // 2 * produced + 1 while every completed dword could be stored in place; GJ_ENC_NO_STORE once one could not
};
#define GJ_ENC_NO_STORE 4096

// append the n <= 26 bits `cw` to a lane's private stream; p = zig-zag position of the coefficient being coded (everything up to
// it has been read). The halves of dword f may be written over positions 2f, 2f + 1 (rows 2f, 2f + 1, lower halves) once both are
// behind the walk and the rows are private ones: lim = 2f + 1 <= min(p, GJ_ENC_PRIV_ROWS - 1). From the first dword that cannot,
// the stream continues in the block's own slot of d_temp (`spill`, GJ_TEMP_BYTES_PER_BLOCK bytes = the largest possible block):
// the segment's final stream, written there by the drain, never reaches a slot whose block it has not passed yet.
__device__ __forceinline__ void gj_put(GjWalk& w, const uint32_t cw, const int n, uint8_t* col, uint32_t* __restrict__ spill, const int p)
{
    w.room -= n;
    const uint64_t t = (uint64_t)cw << w.room; // room was > 32, n <= 26: the shift is >= 6
    w.hi |= (uint32_t)(t >> 32);
    if (w.room <= 32) {
        if (w.lim <= min(p, GJ_ENC_PRIV_ROWS - 1)) {
            *reinterpret_cast<uint16_t*>(col + w.produced * 2048) = (uint16_t)(w.hi >> 16);
            *reinterpret_cast<uint16_t*>(col + w.produced * 2048 + 1024) = (uint16_t)w.hi;
            w.lim += 2;
        } else {
            if (w.lim != GJ_ENC_NO_STORE) w.stored = w.produced; // (the first dword that goes to the slot: the ones in front are in the column)
            spill[w.produced] = w.hi;
            w.lim = GJ_ENC_NO_STORE;
        }
        w.produced++;
        w.hi = (uint32_t)t;
        w.room += 32;
    }
}

// category (bit length) and magnitude bits of a coefficient (ITU T.81 F.1.2.1.1). NONZERO: v != 0 is known (AC walk).
template <bool NONZERO>
__device__ __forceinline__ void gj_value_bits2(const int v, int& nbits, uint32_t& bits)
{
    const int s = v >> 31, t = v + s; // t = v - 1 for negative v
    if (NONZERO) {
        nbits = 32 - gj_ffbh_i32(t); // (t is neither 0 nor -1 for a non-zero v: the first bit that differs from the sign is the top bit of |v|)
    } else {
        const uint32_t a = (uint32_t)(t ^ s) | 1u; // |v| = t ^ s (the 1 keeps clz defined for v == 0)
        nbits = v ? 32 - __builtin_clz(a) : 0;
    }
    bits = __builtin_amdgcn_ubfe((uint32_t)t, 0, (uint32_t)nbits);
}

// the AC part of a walk: non-zero coefficients in zig-zag order, ZRL for runs of 16 zeros, EOB unless the block ends non-zero
__device__ __forceinline__ void gj_walk_ac(uint8_t* col, const uint32_t mlo, const uint32_t mhi, const uint32_t* lut_ac, GjWalk& w,
                                           uint32_t* __restrict__ spill)
{
    int prev = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        uint32_t m = half ? mhi : mlo;
        while (m) {
            const int b = __builtin_ctz(m), p = b + 32 * half;
            m &= m - 1;
            int run = p - prev - 1;
            prev = p;
            const int v = *reinterpret_cast<const int16_t*>(col + b * 1024 + half * 2);
            if (run >= 16) {
                const uint32_t zrl = lut_ac[0xF0];
                do {
                    gj_put(w, zrl & 0x03FFFFFFu, (int)(zrl >> 26), col, spill, p);
                    run -= 16;
                } while (run >= 16);
            }
            // category and magnitude bits of the (non-zero) coefficient (ITU T.81 F.1.2.1.1): t = v - 1 for a negative v; the first bit of t that
            // differs from its sign is the top bit of |v|, so k = v_ffbh_i32(t) = 32 - category. The AC table is indexed by (run << 4) | (k & 15)
            // (gj_huffman_coder_lut): with the table's base moved down by 16 entries that is base[(run << 4) + k], two shift-adds
            const int sg = v >> 31, t = v + sg, k = gj_ffbh_i32(t);
            const uint32_t bits = __builtin_amdgcn_ubfe((uint32_t)t, 0, (uint32_t)(32 - k));
            const uint32_t ent = (lut_ac - 16)[gj_lshl_add_u32<4>((uint32_t)run, (uint32_t)k)];
            gj_put(w, (ent & 0x03FFFFFFu) | bits, (int)(ent >> 26), col, spill, p);
        }
    }
    if (prev != 63) {
        const uint32_t eob = lut_ac[0];
        gj_put(w, eob & 0x03FFFFFFu, (int)(eob >> 26), col, spill, 63);
    }
}

// step 5 of gj_code_tile for one lane and one window [wbase, wend) of the tile stream: the lane's stream -- its completed dwords (column, then, for the
// rare block that outgrew it, the block's d_temp slot), the accumulator, the padding -- lands `sh` bits into dword d0 of the tile stream, every output
// dword is a funnel shift of two neighbours ORed into the window. One loop per KIND of source (round 5: a single loop that picked the source of every
// dword behind four lane-dependent conditions cost 28 vector instructions per dword; these take 8), the window's bounds only where a tile's stream
// needs more than one window (WHOLE = false: noise at high qualities).
template <bool WHOLE>
__device__ __forceinline__ void gj_merge_stream(const GjWalk& w, const uint8_t* col, const uint32_t* __restrict__ spill, const uint64_t tail,
                                                const int ndw, const uint32_t sh, const uint32_t d0, uint32_t* s_bits, const uint32_t wbase,
                                                const uint32_t wend)
{
    uint32_t prevv = 0, d = d0;
    auto emit = [&](const uint32_t cur) {
        const uint32_t out = __builtin_amdgcn_alignbit(prevv, cur, sh);
        if (out && (WHOLE || (d >= wbase && d < wend))) atomicOr(&s_bits[d - wbase], out); // (a dword that is not zero lies inside the lane's segment)
        prevv = cur;
        d++;
    };
    const int stored = w.lim == GJ_ENC_NO_STORE ? w.stored : w.produced; // completed dwords that sit in the column
    for (int f = 0; f < stored; f++)
        emit(((uint32_t)*reinterpret_cast<const uint16_t*>(col + f * 2048) << 16) | *reinterpret_cast<const uint16_t*>(col + f * 2048 + 1024));
    for (int f = stored; f < w.produced; f++) emit(spill[f]);
    // the tail: ndw - produced = 0, 1 or 2 dwords of it carry bits; one more step flushes the last carry
    emit((uint32_t)(tail >> 32));
    if (ndw > w.produced) emit((uint32_t)tail);
    if (ndw > w.produced + 1) emit(0u);
}

// Steps 2-5 for one component of a tile. i = thread, j = local segment of the lane's block, k = block inside its segment,
// nblocks = blocks of that segment, table = 0 luminance / 1 chrominance tables, dc_dist = lanes back to the previous block of the
// same component; region = the tile's area of d_temp (GJ_TEMP_BYTES_PER_BLOCK per block: the tile's UNSTUFFED stream from its start --
// every segment on a dword boundary, in the order of the scan --, block i's spill slot at i * GJ_TEMP_BYTES_PER_BLOCK, which the stream
// reaches only after block i has been merged into it), seg_count_left = segments of the scan from the tile's first one on (the last
// one of a scan gets no restart marker); seg_bytes / seg_ff = unstuffed size and 0xFF count per segment (k_gather stuffs).
// Returns the size of the tile's FINISHED stream (stuffed, restart markers included; the same in every thread).
__device__ __forceinline__ uint32_t gj_code_tile(const GjCoderLds& L, const int i, const int j, const int k, const bool active, const int spt,
                                                 const int nblocks, const int table, const int dc_dist, const int seg_count_left,
                                                 uint8_t* __restrict__ region,
                                                 uint32_t* __restrict__ seg_bytes, uint32_t* __restrict__ seg_ff, const uint32_t first_segment,
                                                 const int trace0 = -1)
{
    (void)trace0;
    const int lane = i & 63, wave = i >> 6;
    uint8_t* const col = reinterpret_cast<uint8_t*>(L.coef) + i * 4;
    uint32_t* const s_bits = L.coef + GJ_ENC_PRIV_ROWS * 256;
    const uint32_t* const lut_ac = L.lut + table * 272;
    const uint32_t* const lut_dc = lut_ac + 256;

    // ---- 2. read the column back: non-zero mask, DC term
    uint32_t mlo = 0, mhi = 0;
    int dc = 0;
    {
        uint32_t elo = 0, ehi = 0;
#pragma unroll
        for (int q = 0; q < 32; q++) {
            const uint32_t d = L.coef[q * 256 + i]; // positions q (lower half) and q + 32
            if (q == 0) dc = (int)(int16_t)(d & 0xFFFFu);
            // both halves clamped to 0 / 1 (v_pk_min_u16); elo collects rows 0..15: bit q = position q, bit 16 + q = position q + 32
            const uint32_t m = gj_pk_min_u16(d, 0x00010001u);
            if (q < 16) elo |= m << q;
            else ehi |= m << (q - 16);
        }
        if (active) {
            mlo = __builtin_amdgcn_perm(ehi, elo, 0x05040100u); // lower halves: positions 0..15 | 16..31
            mhi = __builtin_amdgcn_perm(ehi, elo, 0x07060302u); // upper halves: positions 32..47 | 48..63
        }
    }
    if (lane >= 48) L.edge[wave * 16 + (lane - 48)] = dc;
    if (i < GJ_ENC_MAX_SPT) L.segff[i] = 0;
    __syncthreads(); // B1: edges visible (and, for the first component, the tables)

    // ---- 3. the walk
    GjWalk w = {0, 64, 0, 0, 1};
    uint32_t* const spill = reinterpret_cast<uint32_t*>(region + (size_t)i * GJ_TEMP_BYTES_PER_BLOCK); // (lane i = block i of the tile)
    int dc_diff = 0;
    {
        // DC prediction inside the segment (reset at its first block, src/gpujpeg_huffman_gpu_encoder.cu:339-342)
        const int src = lane - dc_dist;
        int pred = __builtin_amdgcn_ds_bpermute((src & 63) << 2, dc);
        if (src < 0 && wave > 0) pred = L.edge[(wave - 1) * 16 + (16 + src)];
        if (k - dc_dist < 0) pred = 0;
        dc_diff = dc - pred;
    }
    if (active) {
        int nbits;
        uint32_t bits;
        gj_value_bits2<false>(dc_diff, nbits, bits);
        const uint32_t ent = lut_dc[nbits];
        gj_put(w, (ent & 0x03FFFFFFu) | bits, (int)(ent >> 26), col, spill, 0);
        gj_walk_ac(col, mlo & ~1u, mhi, lut_ac, w, spill);
    }
    const int fill = 64 - w.room; // bits in the accumulator (< 32)
    const uint32_t len = (uint32_t)w.produced * 32u + (uint32_t)fill;

    if (trace0 >= 0) GJ_TRACE_E(trace0 + 1); // walk done (this wave)
    // ---- 4. bit positions
    const uint32_t winc = gj_wave_incl_scan(len);
    if (lane == 63) L.wsum[wave] = winc;
    __syncthreads(); // B2: wave totals; every walk is finished, so the window rows are free
    uint32_t excl;
    {
        const uint32_t a = L.wsum[0], b = L.wsum[1], c = L.wsum[2];
        const uint32_t incl = winc + (wave == 0 ? 0u : wave == 1 ? a : wave == 2 ? a + b : a + b + c);
        if (active && k == 0) L.segx[j] = incl - len;
        if (active && k == nblocks - 1) L.segend[j] = incl;
        excl = incl - len;
    }
    {   // clear the first window
        uint4* z = reinterpret_cast<uint4*>(s_bits) + i * 2;
        z[0] = make_uint4(0, 0, 0, 0);
        z[1] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads(); // B3: segment ends visible, window cleared
    if (trace0 >= 0) GJ_TRACE_E(trace0 + 2); // positions known
    // segment books, redundantly in every wave (lane l keeps local segment l): bits with ones-padding to a byte, dword base
    uint32_t sbits = 0, sdw = 0;
    if (lane < spt && lane < seg_count_left) {
        sbits = L.segend[lane] - L.segx[lane];
        sbits += (8u - (sbits & 7u)) & 7u;
        sdw = (sbits + 31u) >> 5;
    }
    const uint32_t sbase_incl = gj_wave_incl_scan(sdw);
    const uint32_t sbase = sbase_incl - sdw;
    const uint32_t total_dw = (uint32_t)__builtin_amdgcn_readlane((int)sbase_incl, 63);
    if (wave == 0) {
        if (lane < spt) { L.segbase[lane] = sbase; L.segbits[lane] = sbits; }
        if (lane == 63) L.segbase[spt] = total_dw;
    }
    uint32_t start_bit = 0;
    int pad_bits = 0;
    {
        const uint32_t my_base = (uint32_t)__builtin_amdgcn_ds_bpermute(j << 2, (int)sbase);
        const uint32_t my_x = active ? L.segx[j] : 0u;
        start_bit = my_base * 32u + (excl - my_x);
        if (active && k == nblocks - 1) pad_bits = (int)((8u - ((start_bit + len) & 7u)) & 7u);
    }

    // ---- 5. merge into the window, drain the window to HBM
    // The lane's stream is its `produced` completed dwords, then the accumulator and the ones-padding of a segment's last block as
    // one 64-bit tail; it lands `start_bit & 31` bits into dword `start_bit >> 5` of the tile stream, so every output dword is
    // one funnel shift (v_alignbit_b32) of two neighbouring stream dwords and one ds_or_b32.
    const uint32_t sh = start_bit & 31u, d0 = start_bit >> 5;
    uint64_t tail = (uint64_t)w.hi << 32;
    if (pad_bits) tail |= (uint64_t)((1u << pad_bits) - 1u) << (64 - fill - pad_bits);
    const int ndw = w.produced + (fill + pad_bits > 32 ? 2 : (fill + pad_bits > 0 ? 1 : 0)); // stream dwords incl. the tail
    const int nseg = min(spt, seg_count_left);
    uint32_t* const dst = reinterpret_cast<uint32_t*>(region); // dword d of the tile stream
    for (uint32_t wbase = 0; wbase < total_dw; wbase += GJ_ENC_WIN_DW) {
        const uint32_t wend = min(total_dw, wbase + (uint32_t)GJ_ENC_WIN_DW);
        if (wbase) {
            __syncthreads(); // previous window drained
            for (uint32_t d = i; d < wend - wbase; d += 256) s_bits[d] = 0;
            __syncthreads();
        }
        if (active && d0 + (uint32_t)ndw + 1u > wbase && d0 < wend) {
            if (total_dw <= (uint32_t)GJ_ENC_WIN_DW) gj_merge_stream<true>(w, col, spill, tail, ndw, sh, d0, s_bits, 0u, total_dw);
            else gj_merge_stream<false>(w, col, spill, tail, ndw, sh, d0, s_bits, wbase, wend);
        }
        __syncthreads(); // B4: window complete
        // every wave drains whole segments: no search for the owner of a dword, the 0xFF count of a segment is one wave reduction
        for (int sl = wave; sl < nseg; sl += 4) {
            const uint32_t sb = L.segbase[sl], nfl = (L.segbits[sl] + 31u) >> 5;
            const uint32_t lo = max(sb, wbase), hi = min(sb + nfl, wend);
            uint32_t ffc = 0;
            for (uint32_t d = lo + (uint32_t)lane; d < hi; d += 64) {
                const uint32_t v = s_bits[d - wbase];
                // 0xFF bytes (the unused low bytes of a segment's last dword are zero)
                ffc += (uint32_t)__builtin_popcount(((v & 0x7F7F7F7Fu) + 0x01010101u) & v & 0x80808080u);
                dst[d] = __builtin_bswap32(v);
            }
            ffc = gj_wave_incl_scan(ffc);
            if (lane == 63 && ffc) L.segff[sl] += ffc;
        }
    }
    __syncthreads(); // B5: 0xFF counts complete; the coefficient area may be overwritten by the next component
    if (trace0 >= 0) GJ_TRACE_E(trace0 + 3); // merged and drained
    // the segments' sizes for k_gather, and what the tile's stream will measure once it is stuffed
    uint32_t out = 0;
    if (lane < nseg) {
        const uint32_t nb = (L.segbits[lane] + 7u) >> 3, ff = L.segff[lane];
        out = nb + ff + (lane != seg_count_left - 1 ? 2u : 0u);
        if (wave == 0) {
            seg_bytes[first_segment + lane] = nb;
            seg_ff[first_segment + lane] = ff;
        }
    }
    return (uint32_t)__builtin_amdgcn_readlane((int)gj_wave_incl_scan(out), 63);
}

__device__ __forceinline__ void gj_load_coder_lut(uint32_t* s_lut, const uint32_t* __restrict__ lut, const int i)
{
    static_assert(GJ_CODER_LUT_WORDS == 2 * 272 && GJ_CODER_LUT_WORDS % 4 == 0, "layout of GjCoderLds::lut");
    if (i < GJ_CODER_LUT_WORDS / 4) reinterpret_cast<uint4*>(s_lut)[i] = reinterpret_cast<const uint4*>(lut + GJ_CODER_LUT_OFFSET)[i];
}


// ================================================================================================
// Fully fused fast path: packed 4:4:4 pixels -> per-segment (unstuffed) Huffman streams, no coefficient planes.
//
// k_fused_rgb444 + k_huffman move 2 x 199 MB of int16 coefficients through HBM for an 8K frame; measured, the store half
// alone costs as much as all arithmetic of the kernel. Both kernels already give one thread one 8x8 block, so the
// quantised block can stay with that thread: a workgroup takes spt = 256 / B whole restart segments (B blocks
// each, e.g. 7 x 36 = 252 block positions) of ALL THREE component scans, colour-converts its pixels once, then for one
// component after the other transforms the block into its LDS column and runs the coder above on it. The output (the tile's unstuffed
// stream in d_temp, byte and 0xFF counts per segment, the stream's size in the file) is what k_gather turns into the file.
// Used for non-interleaved 4:4:4 with restart intervals of 4 .. 256 blocks.
// ================================================================================================
template <int CS_FROM, int CS_TO, bool ONE_COMPONENT = false>
__global__ __launch_bounds__(256, 4) void k_encode_rgb444(const gj_geom g, const uint8_t* __restrict__ raw, const float* __restrict__ q_luma,
                                                          const float* __restrict__ q_chroma, const uint32_t* __restrict__ lut,
                                                          uint8_t* __restrict__ temp, uint32_t* __restrict__ seg_bytes,
                                                          uint32_t* __restrict__ seg_ff, const GjTail T)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_coef[32 * 256];
    __shared__ __attribute__((aligned(8))) float s_q[3][64];
    __shared__ __attribute__((aligned(16))) uint32_t s_lut[2 * 272];
    __shared__ uint32_t s_wsum[4];
    __shared__ int s_edge[64];
    __shared__ uint32_t s_segx[GJ_ENC_MAX_SPT], s_segend[GJ_ENC_MAX_SPT], s_segbase[GJ_ENC_MAX_SPT + 1], s_segbits[GJ_ENC_MAX_SPT], s_segff[GJ_ENC_MAX_SPT];
    const GjCoderLds L = {s_coef, s_lut, s_wsum, s_edge, s_segx, s_segend, s_segbase, s_segbits, s_segff};

    const int i = threadIdx.x;
    GJ_TRACE_E(0);
    gj_load_coder_lut(s_lut, lut, i);
    if (i < 192) s_q[i >> 6][i & 63] = (g.comp[i >> 6].type ? q_chroma : q_luma)[i & 63];
    // frame blockIdx.z of a batch (gj_enc_job::batch; strides zero for a single frame)
    const size_t fz = blockIdx.z;
    raw += fz * T.f_raw;
    temp += fz * T.f_temp;
    seg_bytes += fz * T.f_seg;
    seg_ff += fz * T.f_seg;

    const gj_comp_geom& k0 = g.comp[0];
    const int B = g.seg_blocks;
    const int spt = 256 / B;       // segments per workgroup (per component)
    const int tile_blocks = spt * B;
    const uint32_t recip = (65536u + (uint32_t)B - 1u) / (uint32_t)B; // j = i / B through a 16.16 reciprocal (exact for i < 256, B <= 256)
    const int j = min((int)(((uint32_t)i * recip) >> 16), GJ_ENC_MAX_SPT - 1);
    const int k = i - j * B;       // block inside its segment
    // The tile and, when the workgroup codes ONE component of it, which. Three shapes of launch:
    //   all three components per workgroup (frames from ~half a generation of workgroups up),
    //   ONE_COMPONENT with gridDim.y == 3 (small frames: see below),
    //   and a mixture: the LAST tiles of a frame that has more tiles than the GPU has places (8K: 2058 tiles for 1024 places) as three short workgroups
    //   each, behind the whole ones in the grid. A launch ends with the workgroups that started last; the whole tiles of the last, under-filled
    //   generation run on nearly empty CUs at the latency of one workgroup (~36 us), the split ones start as soon as the first places come free
    //   and take ~40 % of that. 8K alone: 80.7 -> 73.9 us with the last 16 ... 96 tiles split, nothing lost with four pipelines up to 32
    //   (profiles/r5_09_encoder_tail_tiles_split.txt); the verdict's "lone-launch tax".
    unsigned tile = blockIdx.x, ntiles = gridDim.x;
    int only = ONE_COMPONENT ? (int)blockIdx.y : -1;
    if (!ONE_COMPONENT && T.tail_from != 0xFFFFFFFFu) {
        ntiles = T.tiles;
        if (blockIdx.x >= T.tail_from) {
            const unsigned r = blockIdx.x - T.tail_from, q = r / 3u;
            tile = T.tail_from + q;
            only = (int)(r - 3u * q);
        }
    }
    const int seg0 = (int)tile * spt; // first segment (inside each component's scan)
    const unsigned nb = (unsigned)(k0.blocks_x * k0.blocks_y);
    const unsigned lb = tile * (unsigned)tile_blocks + (unsigned)i;
    const bool active = i < tile_blocks && lb < nb; // (every component has the same geometry)
    // the block position: the tile's first block by one division of uniform values, the lane's by carrying over the ends of the block rows (a lane
    // without a block of its own -- tile slack, behind the last block -- takes the frame's last one: nobody looks at what it makes of it)
    unsigned bx, by;
    {
        const unsigned bxn = (unsigned)k0.blocks_x, lbc = min(lb, nb - 1u);
        if (bxn >= 256u) {
            const unsigned lb0 = tile * (unsigned)tile_blocks;
            const unsigned by0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(lb0 / bxn));
            bx = lbc - by0 * bxn;
            by = by0;
            if (bx >= bxn) { bx -= bxn; by++; } // (a tile of 256 blocks crosses the end of a block row once at most)
        } else {
            by = lbc / bxn;
            bx = lbc - by * bxn;
        }
    }

    // ---- pixels -> three byte-packed component blocks
    uint32_t pk[3][16];
    gj_load_color_444<CS_FROM, CS_TO>(g, raw, bx, by, pk);
    __syncthreads(); // tables are in LDS
    GJ_TRACE_E(1); // pixels loaded and converted

    // A small frame has fewer tiles than the GPU has places for workgroups (HD: 135 for 1024): ONE_COMPONENT, launched with gridDim.y == 3, codes
    // ONE component of its tile -- the pixels are loaded and converted three times, by CUs that would otherwise idle, and a tile's components
    // run side by side instead of one after the other.
    // (a template parameter: the check costs the three-component instantiation of the 8K frame 0.7 us when it is made at run time)
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (only >= 0 && c != only) continue;
        const gj_comp_geom& kc = g.comp[c];
        // (pinned: the transform of component c + 1 would otherwise be hoisted over the coder of c)
#pragma unroll
        for (int t = 0; t < 16; t++) GJ_KEEP(pk[c][t]);
        gj_fdct_quant_zz(pk[c], s_q[c], reinterpret_cast<uint8_t*>(s_coef) + i * 4);
        GJ_TRACE_E(2 + 4 * c); // transformed (this wave)
        const uint64_t first_block = kc.data_offset / 64 + (uint64_t)seg0 * B; // coding-order index of the tile's first block of this component
        const uint32_t size = gj_code_tile(L, i, j, k, active, spt, active ? min(B, (int)nb - (seg0 + j) * B) : 0, kc.type, 1, k0.segment_count - seg0,
                                           temp + first_block * GJ_TEMP_BYTES_PER_BLOCK, seg_bytes, seg_ff, (uint32_t)(kc.first_segment + seg0), 2 + 4 * c);
        // file order: the luminance scan's tiles, then the two chrominance scans'
        if (i == 0) gj_piece_put(T, (uint32_t)c * ntiles + tile, size, fz * T.f_tail);
    }
}

// ================================================================================================
// k_encode_rgb444's counterpart for interleaved packed 4:2:2 without colour transform (BASELINE config 4): one lane per
// block in CODING order (Y0 Y1 Cb Cr of MCU 0, of MCU 1, ...), a workgroup takes spt = 256 / B whole restart segments
// (B = 4 x restart interval blocks each). All four lanes of an MCU read its 8 x 32 bytes (the same addresses merge in
// the load unit), pick their own samples with byte permutes, transform, and the coder runs once on the whole
// tile -- no coefficient planes, one pass instead of k_encode_rgb444's three.
// (The DC predecessor distance and the table are per lane here.)
// ================================================================================================
__global__ __launch_bounds__(256, 4) void k_encode_uyvy422(const gj_geom g, const uint8_t* __restrict__ raw, const float* __restrict__ q_luma,
                                                           const float* __restrict__ q_chroma, const uint32_t* __restrict__ lut,
                                                           uint8_t* __restrict__ temp, uint32_t* __restrict__ seg_bytes,
                                                           uint32_t* __restrict__ seg_ff, const GjTail T)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_coef[32 * 256];
    __shared__ __attribute__((aligned(8))) float s_q[2][64];
    __shared__ __attribute__((aligned(16))) uint32_t s_lut[2 * 272];
    __shared__ uint32_t s_wsum[4];
    __shared__ int s_edge[64];
    __shared__ uint32_t s_segx[GJ_ENC_MAX_SPT], s_segend[GJ_ENC_MAX_SPT], s_segbase[GJ_ENC_MAX_SPT + 1], s_segbits[GJ_ENC_MAX_SPT], s_segff[GJ_ENC_MAX_SPT];
    const GjCoderLds L = {s_coef, s_lut, s_wsum, s_edge, s_segx, s_segend, s_segbase, s_segbits, s_segff};

    const int i = threadIdx.x;
    gj_load_coder_lut(s_lut, lut, i);
    if (i < 128) s_q[i >> 6][i & 63] = (i < 64 ? q_luma : q_chroma)[i & 63];
    // frame blockIdx.z of a batch (gj_enc_job::batch; strides zero for a single frame)
    raw += (size_t)blockIdx.z * T.f_raw;
    temp += (size_t)blockIdx.z * T.f_temp;
    seg_bytes += (size_t)blockIdx.z * T.f_seg;
    seg_ff += (size_t)blockIdx.z * T.f_seg;

    const gj_comp_geom& kc = g.comp[1];
    const int ri = g.restart_interval;
    const int B = g.seg_blocks;    // 4 x ri
    const int spt = 256 / B;       // segments per workgroup
    const int tile_blocks = spt * B;
    const uint32_t recip = (65536u + (uint32_t)B - 1u) / (uint32_t)B; // j = i / B through a 16.16 reciprocal (exact for i < 256, B <= 256)
    const int j = min((int)(((uint32_t)i * recip) >> 16), GJ_ENC_MAX_SPT - 1);
    const int k = i - j * B;       // block inside its segment
    const int p = k & 3;           // position inside the MCU: Y0 Y1 Cb Cr
    const int seg0 = blockIdx.x * spt;
    const unsigned m = (unsigned)(seg0 + j) * (unsigned)ri + (unsigned)(k >> 2); // MCU
    const unsigned nm = (unsigned)g.mcu_count;
    const bool active = i < tile_blocks && seg0 + j < g.segment_count && m < nm;
    const unsigned my = m / (unsigned)kc.blocks_x, mx = m - my * (unsigned)kc.blocks_x;

    // ---- pixels -> this lane's byte-packed block
    uint32_t px[16];
    {
        const size_t pitch = (size_t)g.width * 2 + g.width_padding;
        const bool interior = (mx * 16 + 16 <= (unsigned)g.width) && (my * 8 + 8 <= (unsigned)g.height);
        const bool aligned = ((pitch | (size_t)raw) & 15) == 0;
        if (!active) {
#pragma unroll
            for (int t = 0; t < 16; t++) px[t] = 0;
        } else if (interior && aligned) {
            const uint4* src = reinterpret_cast<const uint4*>(raw + (size_t)(my * 8) * pitch + (size_t)mx * 32);
            const size_t pitch4 = pitch >> 4;
            const int first = p == 1; // Y1 lives in the second 16 bytes of the row; chroma needs both halves
            const uint32_t selc = p == 2 ? 0x05040100u : 0x07060302u;
            uint4 lo[8], hi[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                lo[r] = src[r * pitch4 + first];
                hi[r] = src[r * pitch4 + 1];
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t y0 = __builtin_amdgcn_perm(lo[r].y, lo[r].x, 0x07050301u), y1 = __builtin_amdgcn_perm(lo[r].w, lo[r].z, 0x07050301u);
                const uint32_t uv01 = __builtin_amdgcn_perm(lo[r].y, lo[r].x, 0x06020400u), uv23 = __builtin_amdgcn_perm(lo[r].w, lo[r].z, 0x06020400u);
                const uint32_t uv45 = __builtin_amdgcn_perm(hi[r].y, hi[r].x, 0x06020400u), uv67 = __builtin_amdgcn_perm(hi[r].w, hi[r].z, 0x06020400u);
                const uint32_t c0 = __builtin_amdgcn_perm(uv23, uv01, selc), c1 = __builtin_amdgcn_perm(uv67, uv45, selc);
                px[2 * r] = p < 2 ? y0 : c0;
                px[2 * r + 1] = p < 2 ? y1 : c1;
            }
        } else {
            // samples outside the image are zero component values (src/gpujpeg_common.c:941-944); the odd last pixel of an
            // odd-width row shares the chroma of its pair like the generic loader does
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const unsigned y = my * 8 + r;
                uint32_t d[2] = {0, 0};
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    uint32_t v = 0;
                    if (y < (unsigned)g.height) {
                        if (p < 2) {
                            const unsigned x = mx * 16 + p * 8 + t;
                            if (x < (unsigned)g.width) v = raw[(size_t)y * pitch + (size_t)x * 2 + 1];
                        } else {
                            const unsigned cx = mx * 8 + t;
                            if (cx < (unsigned)kc.width) v = raw[(size_t)y * pitch + (size_t)cx * 4 + (p == 2 ? 0 : 2)];
                        }
                    }
                    d[t >> 2] |= v << (8 * (t & 3));
                }
                px[2 * r] = d[0];
                px[2 * r + 1] = d[1];
            }
        }
    }
    __syncthreads(); // tables are in LDS
    {
        const int table = p < 2 ? g.comp[0].type : g.comp[1].type;
#pragma unroll
        for (int t = 0; t < 16; t++) GJ_KEEP(px[t]);
        gj_fdct_quant_zz(px, s_q[table ? 1 : 0], reinterpret_cast<uint8_t*>(s_coef) + i * 4);
        const uint64_t first_block = (uint64_t)seg0 * B;
        const uint32_t size = gj_code_tile(L, i, j, k, active, spt, active ? min(B, ((int)nm - (seg0 + j) * ri) * 4) : 0, table,
                                           p == 0 ? 3 : (p == 1 ? 1 : 4) /* Y1 follows the Y0 of its own MCU */, g.segment_count - seg0,
                                           temp + first_block * GJ_TEMP_BYTES_PER_BLOCK, seg_bytes, seg_ff, (uint32_t)seg0);
        if (i == 0) gj_piece_put(T, blockIdx.x, size, (size_t)blockIdx.z * T.f_tail);
    }
}

// ================================================================================================
// The fully fused encoder for every other layout with restart segments of 4 .. 256 blocks: one lane per block in CODING order, whatever
// the scan structure (gj_segment_block gives the lane its component and block position), a workgroup takes spt = 256 / B whole
// restart segments of one scan, every lane fetches the 64 samples of ITS block, transforms them and the coder of k_encode_rgb444
// runs once on the tile. Replaces k_preprocess (one thread per pixel, byte loads and stores) + k_dct + k_huffman and their planes
// (src/gpujpeg_preprocessor.cu:173-292 has one specialised kernel per sampling; here the sampling is the lane's address arithmetic).
//   PLANAR: planar / grey input whose layout equals the component layout (the reference's copy path, :397-453)
//   !PLANAR: packed 4:4:4 pixels with a colour transform from RGB (or none) and point-sampled chroma (:49-63): the lane computes only
//            its own component, out of the pixels (x * sub_h, y * sub_v)
// ================================================================================================
// row of the colour matrix that produces component c (RGB -> CS_TO), pre-divided by 256 with offset + 0.5 / 256 (see gj_matrix_to_f)
__device__ __forceinline__ void gj_matrix_row(const int cs_to, const int c, float& m0, float& m1, float& m2, float& off)
{
    static constexpr int M[3][9] = {{66, 129, 25, -38, -74, 112, 112, -94, -18},    // BT.601 limited
                                    {77, 150, 29, -43, -85, 128, 128, -107, -21},   // BT.601 full range (JPEG)
                                    {47, 157, 16, -26, -87, 112, 112, -102, -10}};  // BT.709
    static constexpr int BASE[3][3] = {{16, 128, 128}, {0, 128, 128}, {16, 128, 128}};
    const int t = cs_to == GJ_CS_BT601 ? 0 : cs_to == GJ_CS_BT601_256 ? 1 : 2;
    const float s = 1.0f / 256.0f;
    m0 = (float)M[t][c * 3] * s;
    m1 = (float)M[t][c * 3 + 1] * s;
    m2 = (float)M[t][c * 3 + 2] * s;
    off = (float)BASE[t][c] + 0.5f / 256.0f;
}

template <bool PLANAR>
__global__ __launch_bounds__(256, 4) void k_encode_blocks(const gj_geom g, const uint8_t* __restrict__ raw, const float* __restrict__ q_luma,
                                                          const float* __restrict__ q_chroma, const uint32_t* __restrict__ lut,
                                                          uint8_t* __restrict__ temp, uint32_t* __restrict__ seg_bytes,
                                                          uint32_t* __restrict__ seg_ff, const GjTail T)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_coef[32 * 256];
    __shared__ __attribute__((aligned(8))) float s_q[2][64];
    __shared__ __attribute__((aligned(16))) uint32_t s_lut[2 * 272];
    __shared__ uint32_t s_wsum[4];
    __shared__ int s_edge[64];
    __shared__ uint32_t s_segx[GJ_ENC_MAX_SPT], s_segend[GJ_ENC_MAX_SPT], s_segbase[GJ_ENC_MAX_SPT + 1], s_segbits[GJ_ENC_MAX_SPT], s_segff[GJ_ENC_MAX_SPT];
    const GjCoderLds L = {s_coef, s_lut, s_wsum, s_edge, s_segx, s_segend, s_segbase, s_segbits, s_segff};

    const int i = threadIdx.x;
    gj_load_coder_lut(s_lut, lut, i);
    if (i < 128) s_q[i >> 6][i & 63] = (i < 64 ? q_luma : q_chroma)[i & 63];
    // frame blockIdx.z of a batch (gj_enc_job::batch; strides zero for a single frame)
    raw += (size_t)blockIdx.z * T.f_raw;
    temp += (size_t)blockIdx.z * T.f_temp;
    seg_bytes += (size_t)blockIdx.z * T.f_seg;
    seg_ff += (size_t)blockIdx.z * T.f_seg;

    const int B = g.seg_blocks;
    const int spt = 256 / B;       // segments per workgroup
    // tiles never cross a scan: only the last segment of a scan may be short, and it has to be the last one of its tile
    int scan = 0, tile = (int)blockIdx.x, scan_first = 0, scan_segs = g.segment_count;
    if (!g.interleaved) {
        for (int c = 0; c < g.comp_count; c++) {
            const int tiles_c = (g.comp[c].segment_count + spt - 1) / spt;
            if (tile < tiles_c || c == g.comp_count - 1) { scan = c; break; }
            tile -= tiles_c;
        }
        scan_first = g.comp[scan].first_segment;
        scan_segs = g.comp[scan].segment_count;
    }
    const int seg0 = tile * spt;   // first segment of the tile inside its scan
    const uint32_t recip = (65536u + (uint32_t)B - 1u) / (uint32_t)B; // j = i / B through a 16.16 reciprocal (exact for i < 256, B <= 256)
    const int j = min((int)(((uint32_t)i * recip) >> 16), GJ_ENC_MAX_SPT - 1);
    const int k = i - j * B;       // block inside its segment
    GjSeg sg;
    sg.nblocks = 0;
    sg.first_block = 0;
    const bool seg_valid = i < spt * B && seg0 + j < scan_segs;
    if (seg_valid) sg = gj_segment(g, scan_first + seg0 + j);
    const bool active = seg_valid && k < sg.nblocks;
    __shared__ uint64_t s_first_block; // coding-order index of the tile's first block (thread 0: j = k = 0)
    if (i == 0) s_first_block = sg.first_block;

    // ---- the lane's block: component, position, samples
    int comp = 0, mcu_pos = 0;
    unsigned bx = 0, by = 0;
    if (active) {
        const uint64_t off = gj_segment_block(g, sg, k, &comp, &mcu_pos);
        const unsigned blk = (unsigned)((off - g.comp[comp].data_offset) >> 6);
        by = blk / (unsigned)g.comp[comp].blocks_x;
        bx = blk - by * (unsigned)g.comp[comp].blocks_x;
    }
    const gj_comp_geom& kc = g.comp[comp];
    uint32_t px[16];
#pragma unroll
    for (int t = 0; t < 16; t++) px[t] = 0;
    if (active && PLANAR) {
        // raw planes back to back, pitch = component width + padding (src/gpujpeg_preprocessor.cu:414-448); outside: zeros
        size_t src_off = 0;
        for (int c = 0; c < comp; c++) src_off += ((size_t)g.comp[c].width + g.width_padding) * g.comp[c].height;
        const size_t pitch = (size_t)kc.width + g.width_padding;
        const uint8_t* p0 = raw + src_off + (size_t)(by * 8) * pitch + bx * 8;
        const bool interior = bx * 8 + 8 <= (unsigned)kc.width && by * 8 + 8 <= (unsigned)kc.height;
        if (interior && ((pitch | (size_t)p0) & 3) == 0) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t* p = reinterpret_cast<const uint32_t*>(p0 + (size_t)r * pitch);
                px[2 * r] = p[0];
                px[2 * r + 1] = p[1];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                uint32_t d[2] = {0, 0};
                if (by * 8 + r < (unsigned)kc.height) {
#pragma unroll
                    for (int t = 0; t < 8; t++)
                        if (bx * 8 + t < (unsigned)kc.width) d[t >> 2] |= (uint32_t)p0[(size_t)r * pitch + t] << (8 * (t & 3));
                }
                px[2 * r] = d[0];
                px[2 * r + 1] = d[1];
            }
        }
    }
    if (active && !PLANAR) {
        const unsigned sh = (unsigned)kc.sub_h, sv = (unsigned)kc.sub_v;
        const size_t pitch = (size_t)g.width * 3 + g.width_padding;
        const bool transform = g.color_space != g.color_space_internal && g.color_space != GJ_CS_NONE && g.color_space_internal != GJ_CS_NONE;
        float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f, off = 0.5f / 256.0f;
        if (transform) gj_matrix_row(g.color_space_internal, comp, m0, m1, m2, off);
        else { m0 = comp == 0 ? 1.0f : 0.0f; m1 = comp == 1 ? 1.0f : 0.0f; m2 = comp == 2 ? 1.0f : 0.0f; off = 0.25f; } // (identity: the chosen channel + 0.25 rounds to itself)
        const unsigned x0 = bx * 8 * sh; // first pixel of the row
        // rows whose 8 * sub_h pixels all exist are fetched as 6 (sub_h = 1) or 12 (sub_h = 2) aligned dwords; measured against one unaligned
        // dword load per sampled pixel (no divergence between the luminance and chrominance lanes of a wave): twice as fast
        const bool fast = sh <= 2 && ((pitch | (size_t)raw) & 3) == 0 && x0 + 8 * sh <= (unsigned)g.width && (by * 8 + 7) * sv < (unsigned)g.height;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const unsigned y = (by * 8 + r) * sv;
            float v[8];
            if (fast) {
                const uint32_t* p = reinterpret_cast<const uint32_t*>(raw + (size_t)y * pitch + (size_t)x0 * 3);
                float rr[8], gg[8], bb[8];
                if (sh == 1) {
                    uint32_t w[6];
#pragma unroll
                    for (int t = 0; t < 6; t++) w[t] = p[t];
                    rr[0] = gj_row_byte_f<0>(w); rr[1] = gj_row_byte_f<3>(w); rr[2] = gj_row_byte_f<6>(w); rr[3] = gj_row_byte_f<9>(w);
                    rr[4] = gj_row_byte_f<12>(w); rr[5] = gj_row_byte_f<15>(w); rr[6] = gj_row_byte_f<18>(w); rr[7] = gj_row_byte_f<21>(w);
                    gg[0] = gj_row_byte_f<1>(w); gg[1] = gj_row_byte_f<4>(w); gg[2] = gj_row_byte_f<7>(w); gg[3] = gj_row_byte_f<10>(w);
                    gg[4] = gj_row_byte_f<13>(w); gg[5] = gj_row_byte_f<16>(w); gg[6] = gj_row_byte_f<19>(w); gg[7] = gj_row_byte_f<22>(w);
                    bb[0] = gj_row_byte_f<2>(w); bb[1] = gj_row_byte_f<5>(w); bb[2] = gj_row_byte_f<8>(w); bb[3] = gj_row_byte_f<11>(w);
                    bb[4] = gj_row_byte_f<14>(w); bb[5] = gj_row_byte_f<17>(w); bb[6] = gj_row_byte_f<20>(w); bb[7] = gj_row_byte_f<23>(w);
                } else { // every other pixel of 16
                    uint32_t lo[6], hi[6];
#pragma unroll
                    for (int t = 0; t < 6; t++) { lo[t] = p[t]; hi[t] = p[6 + t]; }
                    rr[0] = gj_row_byte_f<0>(lo); rr[1] = gj_row_byte_f<6>(lo); rr[2] = gj_row_byte_f<12>(lo); rr[3] = gj_row_byte_f<18>(lo);
                    rr[4] = gj_row_byte_f<0>(hi); rr[5] = gj_row_byte_f<6>(hi); rr[6] = gj_row_byte_f<12>(hi); rr[7] = gj_row_byte_f<18>(hi);
                    gg[0] = gj_row_byte_f<1>(lo); gg[1] = gj_row_byte_f<7>(lo); gg[2] = gj_row_byte_f<13>(lo); gg[3] = gj_row_byte_f<19>(lo);
                    gg[4] = gj_row_byte_f<1>(hi); gg[5] = gj_row_byte_f<7>(hi); gg[6] = gj_row_byte_f<13>(hi); gg[7] = gj_row_byte_f<19>(hi);
                    bb[0] = gj_row_byte_f<2>(lo); bb[1] = gj_row_byte_f<8>(lo); bb[2] = gj_row_byte_f<14>(lo); bb[3] = gj_row_byte_f<20>(lo);
                    bb[4] = gj_row_byte_f<2>(hi); bb[5] = gj_row_byte_f<8>(hi); bb[6] = gj_row_byte_f<14>(hi); bb[7] = gj_row_byte_f<20>(hi);
                }
#pragma unroll
                for (int t = 0; t < 8; t += 2) {
                    gj_f2 a = gj_f2{rr[t], rr[t + 1]}, b = gj_f2{gg[t], gg[t + 1]}, c = gj_f2{bb[t], bb[t + 1]};
                    if (transform) { a = gj_scale256_f(a); b = gj_scale256_f(b); c = gj_scale256_f(c); }
                    const gj_f2 o = __builtin_elementwise_fma((gj_f2)m0, a, __builtin_elementwise_fma((gj_f2)m1, b, __builtin_elementwise_fma((gj_f2)m2, c, (gj_f2)off)));
                    v[t] = o.x;
                    v[t + 1] = o.y;
                }
            } else {
                // edges, other sampling factors, unaligned rows: pixel by pixel; a sample whose pixel lies outside the image is a zero
                // COMPONENT value (src/gpujpeg_common.c:941-944)
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const unsigned x = x0 + t * sh;
                    v[t] = -1.0f; // (converts to 0)
                    if (x < (unsigned)g.width && y < (unsigned)g.height) {
                        const uint8_t* q = raw + (size_t)y * pitch + (size_t)x * 3;
                        float a = (float)q[0], b = (float)q[1], c = (float)q[2];
                        if (transform) { a = fmaxf(a, __builtin_fmaf(a, 256.0f, -65024.0f)); b = fmaxf(b, __builtin_fmaf(b, 256.0f, -65024.0f)); c = fmaxf(c, __builtin_fmaf(c, 256.0f, -65024.0f)); }
                        v[t] = __builtin_fmaf(m0, a, __builtin_fmaf(m1, b, __builtin_fmaf(m2, c, off)));
                    }
                }
            }
            uint32_t d0 = 0, d1 = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                d0 = __builtin_amdgcn_cvt_pk_u8_f32(v[t], t, d0);
                d1 = __builtin_amdgcn_cvt_pk_u8_f32(v[t + 4], t, d1);
            }
            px[2 * r] = d0;
            px[2 * r + 1] = d1;
        }
    }
    __syncthreads(); // tables are in LDS
    {
        const int table = kc.type;
#pragma unroll
        for (int t = 0; t < 16; t++) GJ_KEEP(px[t]);
        gj_fdct_quant_zz(px, s_q[table ? 1 : 0], reinterpret_cast<uint8_t*>(s_coef) + i * 4);
        const uint64_t first_block = s_first_block;
        const uint32_t size = gj_code_tile(L, i, j, k, active, spt, sg.nblocks, table, g.interleaved ? (int)g.mcu_prev[mcu_pos] : 1, scan_segs - seg0,
                                           temp + first_block * GJ_TEMP_BYTES_PER_BLOCK, seg_bytes, seg_ff, (uint32_t)(scan_first + seg0));
        // (workgroups are numbered in file order: the tiles of scan 0, of scan 1, ...)
        if (i == 0) gj_piece_put(T, blockIdx.x, size, (size_t)blockIdx.z * T.f_tail);
    }
}


// fully fused kernel for this configuration, or nullptr
gj_encode_kernel_t gj_encode_kernel(const gj_geom& g, const bool one_component)
{
    if (g.pixel_format != GJ_PF_444_P012 || g.comp_count != 3 || g.interleaved || g.restart_interval <= 0 || g.seg_blocks > 256 || g.seg_blocks < 256 / GJ_ENC_MAX_SPT) return nullptr;
    for (int c = 0; c < 3; c++)
        if (g.comp[c].samp_h != 1 || g.comp[c].samp_v != 1) return nullptr;
    const int from = g.color_space, to = g.color_space_internal;
    if (from == to || from == GJ_CS_NONE || to == GJ_CS_NONE) return one_component ? k_encode_rgb444<GJ_CS_NONE, GJ_CS_NONE, true> : k_encode_rgb444<GJ_CS_NONE, GJ_CS_NONE>;
    if (from == GJ_CS_RGB && to == GJ_CS_BT601_256) return one_component ? k_encode_rgb444<GJ_CS_RGB, GJ_CS_BT601_256, true> : k_encode_rgb444<GJ_CS_RGB, GJ_CS_BT601_256>;
    if (from == GJ_CS_RGB && to == GJ_CS_BT601) return one_component ? k_encode_rgb444<GJ_CS_RGB, GJ_CS_BT601, true> : k_encode_rgb444<GJ_CS_RGB, GJ_CS_BT601>;
    if (from == GJ_CS_RGB && to == GJ_CS_BT709) return one_component ? k_encode_rgb444<GJ_CS_RGB, GJ_CS_BT709, true> : k_encode_rgb444<GJ_CS_RGB, GJ_CS_BT709>;
    if (from == GJ_CS_BT601_256 && to == GJ_CS_RGB) return one_component ? k_encode_rgb444<GJ_CS_BT601_256, GJ_CS_RGB, true> : k_encode_rgb444<GJ_CS_BT601_256, GJ_CS_RGB>;
    return nullptr;
}


gj_encode_kernel_t gj_encode_uyvy422_kernel() { return k_encode_uyvy422; }
gj_encode_kernel_t gj_encode_blocks_kernel(const bool planar) { return planar ? k_encode_blocks<true> : k_encode_blocks<false>; }
