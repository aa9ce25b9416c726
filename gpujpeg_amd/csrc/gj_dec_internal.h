// gj_dec_internal.h -- what the files of the decoder's device code share.
//
//   gj_decode.hip               gj_hip_decode: picks the kernels of a frame and launches them
//   gj_dec_markers.hip          k_marker_scan, k_marker_segments: segment table built on the device, summary for the host in pinned memory
//   gj_dec_entropy_tok.hip      k_huffman_decode_tok: sub-sequence parallel entropy decoding into 16-bit TOKENS + one record per block
//                               (the default for large non-interleaved frames)
//   gj_dec_entropy_par.hip      k_huffman_decode_par: sub-sequence parallel entropy decoding into the coefficient planes (segments of any length)
//   gj_dec_entropy_seq.hip      one lane per restart segment (interleaved scans with many short segments): k_huffman_decode_win over a ring the
//                               lane refills itself (token mode), k_huffman_decode_seq over an LDS stage (plane mode)
//   gj_dec_entropy_serial.hip   k_huffman_decode: one lane per restart segment, stream windows (Huffman tables that do not fit the two-level layout)
//   gj_dec_idct.hip             k_idct_fused_* (from the planes), k_idct_tok_* (from tokens), k_idct / k_postprocess / k_copy_planes_out (generic)
//   gj_bitreader.h              unstuffing of a restart segment into an LDS stage, two-level table look-up
//
// Restates src/gpujpeg_huffman_gpu_decoder.cu:135-495 (entropy decoding semantics; identical results to
// src/gpujpeg_huffman_cpu_decoder.c:245-372), src/gpujpeg_dct_gpu.cu:312-366,472-618 and src/gpujpeg_postprocessor.cu:49-217.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "gj_device.h"
#include "gj_hip.h"

// developer aid (gj_tuning::debug_sync): waits after every launch and names the stage on stderr (which kernel faulted?)
static inline void gj_debug_stage(const bool on, hipStream_t st, const char* what)
{
    if (!on) return;
    const hipError_t e = hipStreamSynchronize(st);
    fprintf(stderr, "[GPUJPEG] [Debug] %s: %s\n", what, hipGetErrorString(e));
}

// ---- the marker scan's second launch when it was left to gj_hip_decode (gj_dec_job::scan)
void gj_launch_marker_table_deferred(const gj_dec_job* job, hipStream_t st);
// does gj_launch_huffman_tok derive its batches' table entries from the scan's records itself (no k_marker_table launch)?
bool gj_tok_folds_table(const gj_dec_job* job);
bool gj_par_folds_table(const gj_dec_job* job); // the same for gj_launch_huffman_par

// ---- entropy decoders: each launches its kernel for the whole segment table of the job
void gj_launch_huffman_serial(const gj_dec_job* job, hipStream_t st);
void gj_launch_huffman_par(const gj_dec_job* job, hipStream_t st);
void gj_launch_huffman_seq(const gj_dec_job* job, hipStream_t st, bool tokens);
void gj_launch_huffman_tok(const gj_dec_job* job, hipStream_t st);

// Batches of the sub-sequence decoders: consecutive table entries, cut per scan -- the luminance segments of a photograph carry two to
// three times the bytes of the chrominance ones, and a batch is sized to fill the LDS stage (one batch size for the whole stream
// gave luminance batches that had to be decoded as two groups, and chrominance batches that left half of the lanes idle).
struct GjBatchPlan {
    int n;                       // ranges (scans)
    int first[GJ_MAX_COMP];      // first table entry of range c
    int count[GJ_MAX_COMP];      // entries
    int g[GJ_MAX_COMP];          // segments per batch
    int batch0[GJ_MAX_COMP + 1]; // first batch of range c; [n] = number of batches
};
// cap_u: bytes of the kernel's LDS stage, max_blocks / gmax: blocks / segments a batch may have; resident (0: no preference): the workgroups
// of the kernel the GPU holds at once -- fuller batches are preferred when they keep the launch within one such generation
GjBatchPlan gj_plan_batches(const gj_dec_job* job, unsigned cap_u, unsigned max_blocks, unsigned gmax, unsigned resident);

// k_huffman_decode_tok (shared with the launcher's choice of the kernel)
#define GJ_TOK_CAP_U 10752     // bytes of unstuffed stream per group, incl. 8 B of zero padding per segment
#define GJ_TOK_MAX_BLOCKS 2304 // blocks per batch
#define GJ_TOK_RESIDENT ((unsigned)gj_hip_cu_count() * 4u)  // workgroups of the token decoder the GPU holds at once (MI355X: 256 CUs x 4)
#define GJ_TOK_GMAX 64         // segments per batch

// ---- what k_marker_scan leaves per scanning workgroup (gj_dec_markers.hip), read by k_marker_table and -- when the table launch is folded into it --
// by k_huffman_decode_tok
#define GJ_SCAN_LIST 2048    // restart markers a workgroup may hold (beyond that the host walks the stream)
#define GJ_SCAN_REC_WORDS 16 // 32-bit words of a record (layout: gj_dec_markers.hip)
struct GjOther { uint32_t pos, code, after, b03, b47; }; // a marker that is no restart marker: position, code, restart markers of its workgroup behind it, the 8 bytes behind the code
// other marker q (0, 1) of a record whose words 4 .. 15 are r1, r2, r3
__device__ __forceinline__ GjOther gj_rec_other(const uint4& r1, const uint4& r2, const uint4& r3, const uint32_t q)
{
    return q == 0 ? GjOther{r1.x, r1.y, r1.z, r1.w, r2.x} : GjOther{r2.z, r2.w, r3.x, r3.y, r3.z};
}
// the 16 bits behind the code: the length of the marker's segment (big-endian in the stream)
__device__ __forceinline__ uint32_t gj_other_len(const GjOther& o) { return ((o.b03 & 0xFFu) << 8) | ((o.b03 >> 8) & 0xFFu); }

// ---- The segment table without its launch (round 5). k_marker_table turns the scanning workgroups' records and marker lists into table entries --
// 11 us and a launch gap in front of every decoded frame, at every size -- and all a batch needs of that table is the positions of its own G + 1
// markers. On the speculative path (the same header as the previous frame: the only outcome the host accepts is a COMPLETE, REGULAR stream --
// every scan with all its segments, restart markers numbered in sequence, SOS headers between the scans, EOI) the structure is known from the
// geometry: restart marker number r of the stream ends segment r - (scans in front) ... so a workgroup reads the <= 256 records (one per lane, one
// trip), checks that every scan has exactly the restart markers it should have (prefix sums), finds the scanning workgroups that hold its
// markers (binary search in LDS) and reads them from their lists (second trip). Anything else -- a marker too many or too few, a number out of
// sequence, an unexpected marker code -- raises rst_irregular in the host's summary and the workgroup leaves: the host decodes the frame again
// the careful way, through k_marker_table (gj_decoder.c). Workgroup 0 also writes the summary the host validates (scan boundaries, SOS bytes, EOI).
struct GjFold {
    const uint32_t* recs;  // nullptr: the table has been written (seg_pos / seg_len / seg_index)
    const uint32_t* lists;
    uint32_t nwg, part_bytes;
    uint64_t begin;
    gj_scan_summary* hsum; // host memory
    uint32_t* h_maxlen;    // host memory: one word per batch
};


// LDS words gj_fold_batch needs for a kernel whose batches hold up to GMAX segments
#define GJ_FOLD_SCRATCH_WORDS(GMAX) (256 + GJ_SCAN_MAX_OTHER * 6 + 2 * ((GMAX) + 1) + 3)

// The table entries of ONE batch -- segments si0 .. si0 + nseg - 1 of the plan's range pc, nseg <= GMAX <= 255 -- from the scan's records: lane j < nseg
// gets its segment's geometric index, position and length (ld_s, ld_p, ld_l). All 256 threads call; `scratch` = GJ_FOLD_SCRATCH_WORDS(GMAX) words
// of LDS nobody else uses meanwhile, s_tmp = the 4 words of gj_wg256_incl_scan. Returns false (in every thread) when the stream is not the
// regular one the geometry describes: rst_irregular is raised in the host's summary and the workgroup must leave. Workgroup 0 writes the summary.
// (IL: the stream is ONE interleaved scan -- a template parameter: asked at run time it cost the 8K frame's token decoder 1.3 us, profiles/r5_13)
template <int GMAX, bool IL = false>
__device__ __forceinline__ bool gj_fold_batch(const GjFold& F, const gj_geom& g, const uint8_t* __restrict__ jpeg, const uint64_t jpeg_size, const GjBatchPlan& plan,
                                              const int pc, const int si0, const int nseg, uint32_t* const scratch, uint32_t* const s_tmp, uint32_t& ld_s,
                                              uint32_t& ld_p, uint32_t& ld_l)
{
    const int tid = threadIdx.x;
    uint32_t* const f_rinc = scratch;                         // [256] restart markers up to and including scanning workgroup t
    uint32_t* const f_o = f_rinc + 256;                       // [GJ_SCAN_MAX_OTHER][6] the other markers in stream order: position, code, restart markers in front, bytes 0..3, 4..7 behind the code, restart markers of its workgroup behind it
    uint32_t* const f_mpos = f_o + GJ_SCAN_MAX_OTHER * 6;     // [GMAX + 1] the batch's restart markers: entry j ends segment k0 - 1 + j of the scan
    uint32_t* const f_mnum = f_mpos + GMAX + 1;               // ... their numbers (RSTn & 7)
    uint32_t* const f_max = f_mnum + GMAX + 1;                // [0] longest segment of the batch, [1], [2] flags (a workgroup-wide "or" without __syncthreads_or, which takes LDS of its own)
    // scans of the stream: one per component in component order, or ONE interleaved scan that holds every segment (plan range 0)
    const int S = IL ? 1 : g.comp_count;
    auto segs_of = [&](const int c) { return IL ? (uint32_t)g.segment_count : (uint32_t)g.comp[c].segment_count; };
    const uint32_t base0 = (uint32_t)(((reinterpret_cast<uintptr_t>(jpeg) + F.begin) & ~(uintptr_t)15) - reinterpret_cast<uintptr_t>(jpeg)); // where the scanning workgroups' parts begin
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
    if ((uint32_t)tid < F.nwg) {
        const uint4* r4 = reinterpret_cast<const uint4*>(F.recs + (size_t)tid * GJ_SCAN_REC_WORDS);
        r0 = r4[0]; r1 = r4[1]; r2 = r4[2]; r3 = r4[3];
    }
    const uint32_t n = r0.x, no = min(r0.z, 2u);
    const bool bad = r0.w != 0 || n > (uint32_t)GJ_SCAN_LIST || r0.z > 2u;
    if (tid < 3) f_max[tid] = 0; // (visible behind the barriers of the prefix sum)
    uint32_t tot;
    // (two sums in one scan: restart markers in the low 22 bits -- 256 workgroups x GJ_SCAN_LIST = 2^19 at most --, other markers above them: up to 2 per
    // workgroup = 512, which an 8-bit field let wrap around to a small count on a malformed stream)
    const uint32_t inc = gj_wg256_incl_scan((n & 0x3FFFFFu) | (no << 22), s_tmp, &tot);
    const uint32_t rinc = inc & 0x3FFFFFu, oinc = inc >> 22;
    f_rinc[tid] = rinc;
    for (uint32_t q = 0; q < no; q++) {
        const uint32_t slot = oinc - no + q;
        if (slot < (uint32_t)GJ_SCAN_MAX_OTHER) {
            const GjOther o = gj_rec_other(r1, r2, r3, q);
            uint32_t* const fo = f_o + slot * 6;
            fo[0] = o.pos; fo[1] = o.code & 0xFFu; fo[2] = rinc - o.after; fo[3] = o.b03; fo[4] = o.b47; fo[5] = o.after;
        }
    }
    if (bad) f_max[1] = 1;
    __syncthreads(); // (the scratch is written)
    const bool bad_any = f_max[1] != 0;
    const uint32_t total_rst = tot & 0x3FFFFFu, total_other = tot >> 22;
    // is this the stream the geometry describes? scan c: segs_c - 1 restart markers, then the SOS of scan c + 1 (EOI behind the last one)
    bool regular = !bad_any && total_other == (uint32_t)S && plan.n == S;
    uint32_t exp_rst = 0, sstart = (uint32_t)F.begin, my_start = 0, my_end = 0, my_first = 0;
    for (int c = 0; c < S && regular; c++) {
        const uint32_t* const fo = f_o + c * 6;
        const uint32_t first_rank = exp_rst;
        exp_rst += segs_of(c) - 1u;
        regular = fo[1] == (c == S - 1 ? 0xD9u : 0xDAu) && fo[2] == exp_rst && fo[0] >= sstart && (uint64_t)fo[0] + 2u <= jpeg_size;
        if (c == pc) { my_start = sstart; my_end = fo[0]; my_first = first_rank; }
        sstart = fo[0] + 2u + (((fo[3] & 0xFFu) << 8) | ((fo[3] >> 8) & 0xFFu)); // behind the SOS header
    }
    regular = regular && total_rst == exp_rst;
    const uint32_t segs_pc = segs_of(pc);
    const int k0 = si0 - plan.first[pc]; // the batch's first segment inside its scan
    bool irregular = false;
    if (regular && tid <= nseg) { // the restart marker that ends segment k0 - 1 + tid of the scan
        const int kk = k0 - 1 + tid;
        uint32_t pos = 0, num = 0xFFu;
        if (kk >= 0 && (uint32_t)kk + 1u < segs_pc) {
            const uint32_t q = my_first + (uint32_t)kk; // its rank among the stream's restart markers (< total_rst)
            int lo = 0, hi = (int)F.nwg - 1;            // the first scanning workgroup with more than q markers up to and including its own
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (f_rinc[mid] > q) hi = mid; else lo = mid + 1;
            }
            const uint32_t idx = q - (lo ? f_rinc[lo - 1] : 0u);
            if (idx < (uint32_t)GJ_SCAN_LIST) {
                const uint32_t ent = F.lists[(size_t)lo * GJ_SCAN_LIST + idx];
                pos = base0 + (uint32_t)lo * F.part_bytes + (ent & 0xFFFFFFu);
                num = ent >> 24;
            } else {
                irregular = true;
            }
        }
        f_mpos[tid] = pos;
        f_mnum[tid] = num;
    }
    __syncthreads();
    if (regular && tid < nseg) {
        const uint32_t k = (uint32_t)(k0 + tid);
        const bool last = k + 1u == segs_pc;
        const uint32_t from = k == 0 ? my_start : f_mpos[tid] + 2u, to = last ? my_end : f_mpos[tid + 1];
        if (!last && f_mnum[tid + 1] != (k & 7u)) irregular = true; // RSTn out of sequence: the reference reader treats it specially, the host walk reproduces that
        if (last && to <= from && segs_pc > 1u) irregular = true;   // an empty segment in front of the end of a scan
        if (to > jpeg_size) irregular = true;
        ld_s = (IL ? 0u : (uint32_t)g.comp[pc].first_segment) + k;
        ld_p = from;
        ld_l = to > from ? to - from : 0u;
        atomicMax(f_max, ld_l);
    }
    if (!regular || irregular) f_max[2] = 1;
    __syncthreads();
    const bool give_up = f_max[2] != 0;
    if (blockIdx.x == 0) { // what the host validates (the table kernel's summary, gj_dec_markers.hip)
        gj_scan_summary* const hs = F.hsum;
        if (tid < (int)min(total_other, (uint32_t)GJ_SCAN_MAX_OTHER)) {
            const uint32_t* const fo = f_o + tid * 6;
            hs->other_pos[tid] = fo[0];
            hs->other_code[tid] = (uint8_t)fo[1];
            const uint32_t ob[4] = {fo[3], fo[4], 0, 0};
            for (int q = 0; q < 4; q++) reinterpret_cast<uint32_t*>(hs->other_bytes[tid])[q] = ob[q];
            hs->other_after[tid] = fo[5];
        }
        if (tid == 0) {
            hs->rst_count = total_rst;
            hs->other_count = total_other;
            hs->scan_count = regular ? (uint32_t)S : 0u;
            hs->status = regular ? 1u : 2u;
            hs->segment_count = regular ? total_rst + (uint32_t)S : 0u;
        }
        if (regular && tid < S) { // scan boundaries: behind the SOS header in front (scan 0: the start of the data) .. the marker that ends it
            uint32_t st = (uint32_t)F.begin;
            if (tid > 0) {
                const uint32_t* const fp = f_o + (tid - 1) * 6;
                st = fp[0] + 2u + (((fp[3] & 0xFFu) << 8) | ((fp[3] >> 8) & 0xFFu));
            }
            hs->scan_start[tid] = st;
            hs->scan_end[tid] = f_o[tid * 6];
        }
    }
    if (tid == 0) {
        F.h_maxlen[blockIdx.x] = give_up ? 0u : *f_max;
        if (give_up) F.hsum->rst_irregular = 1u;
    }
    __syncthreads(); // (the scratch may be reused)
    return !give_up;
}

// ---- IDCT side
typedef void (*gj_idct_tok_t)(const gj_geom, const int16_t*, const uint2*, const uint16_t*, uint32_t, const float*, uint8_t*);
gj_idct_tok_t gj_idct_tok_for(const gj_geom& g); // the token-fed IDCT kernel for this configuration, or nullptr
bool gj_is_uyvy422(const gj_geom& g);
// dequantisation + IDCT + postprocessing of the frame; ev (may be null): events 2 and 3 of gj_hip_decode
void gj_launch_idct(const gj_dec_job* job, hipStream_t st, gj_idct_tok_t idct_tok, gj_event_t* ev);
