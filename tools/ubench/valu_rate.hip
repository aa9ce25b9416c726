// Issue rate of a few gfx950 vector instructions the entropy coders lean on (cycles per wave64 instruction per SIMD): every wave runs
// ITER x 32 independent copies of one instruction on 8 accumulators; 8 waves per SIMD hide the latency, so time = issue time.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2048
#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define KERNEL(name, INS)                                                                                                   \
    __global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t s)                                                  \
    {                                                                                                                       \
        uint64_t a[8];                                                                                                      \
        uint32_t b = s + threadIdx.x, c = (threadIdx.x & 15) + 1;                                                            \
        for (int i = 0; i < 8; i++) a[i] = ((uint64_t)(threadIdx.x + i) << 32) | (i * 77u + s);                               \
        for (int it = 0; it < ITER; it++) { BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) }                                     \
        uint64_t r = 0;                                                                                                     \
        for (int i = 0; i < 8; i++) r ^= a[i];                                                                               \
        if (r == 0x1234567887654321ull) out[threadIdx.x] = (uint32_t)r + b + c;                                               \
    }
#define I_SHL64(i) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a[i]) : "v"(c));
#define I_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(*(uint32_t*)&a[i]) : "v"(b), "v"(c));
#define I_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(*(uint32_t*)&a[i]) : "v"(b));
#define I_ADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
#define I_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(*(uint32_t*)&a[i]) : "v"(b));
#define I_BFE(i) asm volatile("v_bfe_u32 %0, %0, %1, 5" : "+v"(*(uint32_t*)&a[i]) : "v"(c));
#define I_CNDSDWA(i) asm volatile("v_cndmask_b32_sdwa %0, %1, %1, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "+v"(*(uint32_t*)&a[i]) : "v"(b) : "vcc");
#define I_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(*(uint32_t*)&a[i]) : "v"(b), "v"(c));
#define I_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(*(uint32_t*)&a[i]) : "v"(c), "v"(b));
#define I_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
#define I_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(*(uint32_t*)&a[i]) : "v"(b));
#define I_CVTPK(i) asm volatile("v_cvt_pk_u8_f32 %0, %1, %2, %0" : "+v"(*(uint32_t*)&a[i]) : "v"(b), "v"(c));
#define I_SHL32(i) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(*(uint32_t*)&a[i]) : "v"(c));
#define I_CMPADDC(i) asm volatile("v_cmp_lt_u32 vcc, %1, %0\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(*(uint32_t*)&a[i]) : "v"(b) : "vcc");
KERNEL(k_shl64, I_SHL64) KERNEL(k_align, I_ALIGN) KERNEL(k_add, I_ADD) KERNEL(k_add64, I_ADD64) KERNEL(k_mullo, I_MULLO) KERNEL(k_bfe, I_BFE)
KERNEL(k_cndsdwa, I_CNDSDWA) KERNEL(k_perm, I_PERM) KERNEL(k_lshlor, I_LSHLOR) KERNEL(k_pkfma, I_PKFMA) KERNEL(k_fma, I_FMA) KERNEL(k_cvtpk, I_CVTPK)
KERNEL(k_shl32, I_SHL32) KERNEL(k_cmpaddc, I_CMPADDC)
#define A32(i) (*(uint32_t*)&a[i])
#define I_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(A32(i)) : "v"(b));
#define I_OR(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(A32(i)) : "v"(b));
#define I_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(A32(i)) : "v"(b));
#define I_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(A32(i)) : "v"(b) : "vcc");
#define I_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(A32(i)) : "v"(b));
#define I_MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(A32(i)) : "v"(b));
#define I_MULF(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(A32(i)) : "v"(b));
#define I_ADDF(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(A32(i)) : "v"(b));
#define I_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
#define I_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
#define I_CVTUB(i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(A32(i)));
#define I_CVTI(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(A32(i)));
#define I_CVTI_SDWA(i) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(A32(i)));
#define I_ADD_SDWA(i) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(A32(i)) : "v"(b));
#define I_SHR_SDWA(i) asm volatile("v_lshrrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(A32(i)) : "v"(c));
#define I_CMP_SDWA(i) asm volatile("v_cmp_eq_u32_sdwa vcc, %0, %1 src0_sel:BYTE_1 src1_sel:DWORD" : : "v"(A32(i)), "v"(b) : "vcc");
#define I_CMP(i) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(A32(i)), "v"(b) : "vcc");
#define I_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(A32(i)) : "v"(b), "v"(c));
#define I_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(A32(i)) : "v"(b));
#define I_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(A32(i)) : "v"(b), "v"(c));
#define I_MINU(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(A32(i)) : "v"(b));
#define I_ASHR(i) asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(A32(i)) : "v"(c));
#define I_BFEI(i) asm volatile("v_bfe_i32 %0, %0, 16, 16" : "+v"(A32(i)));
#define I_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(A32(i)) : "v"(b), "v"(c));
#define I_RNDNE(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(A32(i)));
#define I_DPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(A32(i)) : "v"(b));
#define I_ADDDPP(i) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(A32(i)) : "v"(b));
#define I_SUB(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(A32(i)) : "v"(b));
#define I_SHLC(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(A32(i)));
#define I_FFBH(i) asm volatile("v_ffbh_u32 %0, %0" : "+v"(A32(i)));
#define I_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(A32(i)) : "v"(b));
#define I_FMAK(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(A32(i)) : "v"(b), "v"(c));
#define I_SUBREV_F(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(A32(i)) : "v"(b));
KERNEL(k_and, I_AND) KERNEL(k_or, I_OR) KERNEL(k_xor, I_XOR) KERNEL(k_cnd, I_CND) KERNEL(k_mov, I_MOV) KERNEL(k_maxf, I_MAXF) KERNEL(k_mulf, I_MULF) KERNEL(k_addf, I_ADDF)
KERNEL(k_pkmul, I_PKMUL) KERNEL(k_pkadd, I_PKADD) KERNEL(k_cvtub, I_CVTUB) KERNEL(k_cvti, I_CVTI) KERNEL(k_cvti_sdwa, I_CVTI_SDWA) KERNEL(k_add_sdwa, I_ADD_SDWA)
KERNEL(k_shr_sdwa, I_SHR_SDWA) KERNEL(k_cmp_sdwa, I_CMP_SDWA) KERNEL(k_cmp, I_CMP) KERNEL(k_add3, I_ADD3) KERNEL(k_lshladd, I_LSHLADD) KERNEL(k_andor, I_ANDOR)
KERNEL(k_minu, I_MINU) KERNEL(k_ashr, I_ASHR) KERNEL(k_bfei, I_BFEI) KERNEL(k_mad24, I_MAD24) KERNEL(k_rndne, I_RNDNE) KERNEL(k_dpp, I_DPP) KERNEL(k_adddpp, I_ADDDPP)
KERNEL(k_sub, I_SUB) KERNEL(k_shlc, I_SHLC) KERNEL(k_ffbh, I_FFBH) KERNEL(k_bcnt, I_BCNT) KERNEL(k_fmac, I_FMAK) KERNEL(k_subf, I_SUBREV_F)

template <typename K> static void run(const char* name, K k, int per)
{
    uint32_t* d;
    (void)hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int wgs = 256 * 8; // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, 1u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = 8.0 * ITER * 32 * per; // waves per SIMD x instructions per wave
    printf("%-12s %8.3f ms  %6.2f ns per instruction per SIMD  (= %5.2f cycles at 2.4 GHz, %5.2f at 2.1 GHz)\n", name, ms, ms * 1e6 / insts_per_simd,
           ms * 1e6 / insts_per_simd * 2.4, ms * 1e6 / insts_per_simd * 2.1);
    (void)hipFree(d);
}
int main()
{
    run("v_add_u32", k_add, 1); run("v_lshlrev_b32", k_shl32, 1); run("v_lshlrev_b64", k_shl64, 1); run("v_alignbit", k_align, 1); run("v_lshl_add_u64", k_add64, 1);
    run("v_mul_lo_u32", k_mullo, 1); run("v_bfe_u32", k_bfe, 1); run("cndmask_sdwa", k_cndsdwa, 1); run("v_perm_b32", k_perm, 1);
    run("v_lshl_or", k_lshlor, 1); run("v_fma_f32", k_fma, 1); run("v_pk_fma_f32", k_pkfma, 1); run("v_cvt_pk_u8", k_cvtpk, 1); run("cmp+addc", k_cmpaddc, 2);
    run("v_and_b32", k_and, 1); run("v_or_b32", k_or, 1); run("v_xor_b32", k_xor, 1); run("v_cndmask", k_cnd, 1); run("v_mov_b32", k_mov, 1);
    run("v_sub_u32", k_sub, 1); run("v_lshlrev imm", k_shlc, 1); run("v_max_f32", k_maxf, 1); run("v_mul_f32", k_mulf, 1); run("v_add_f32", k_addf, 1); run("v_sub_f32", k_subf, 1);
    run("v_fmac_f32", k_fmac, 1); run("v_pk_mul_f32", k_pkmul, 1); run("v_pk_add_f32", k_pkadd, 1); run("cvt_f32_ubyte1", k_cvtub, 1); run("cvt_f32_i32", k_cvti, 1);
    run("cvt_f32_i32_sdwa", k_cvti_sdwa, 1); run("add_u32_sdwa", k_add_sdwa, 1); run("lshrrev_sdwa", k_shr_sdwa, 1); run("cmp_eq_sdwa", k_cmp_sdwa, 1); run("v_cmp_eq_u32", k_cmp, 1);
    run("v_add3_u32", k_add3, 1); run("v_lshl_add_u32", k_lshladd, 1); run("v_and_or_b32", k_andor, 1); run("v_min_u32", k_minu, 1); run("v_ashrrev_i32", k_ashr, 1);
    run("v_bfe_i32", k_bfei, 1); run("v_mad_u32_u24", k_mad24, 1); run("v_rndne_f32", k_rndne, 1); run("v_mov_dpp", k_dpp, 1); run("v_add_dpp", k_adddpp, 1);
    run("v_ffbh_u32", k_ffbh, 1); run("v_bcnt_u32", k_bcnt, 1);
    return 0;
}
