// valu_ops.hip -- issue cost of the individual VALU opcodes the hit path (the "trips") of k_gram_bitslice is
// made of, on MI355X (gfx950): cycles per wave64 instruction and SIMD for an independent stream of each opcode
// at 4 and 8 wavefronts per SIMD.  tools/valu_peak.hip showed that v_and_or_b32 / v_or3_b32 issue at half the
// rate of v_bitop3_b32; this prices the rest of the integer VOP1/VOP2/VOP3 opcodes the same way.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_ops.hip -o tools/valu_ops && tools/valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// one opcode = a string with %0 (read-write VGPR), %1, %2 (read-only VGPRs), %3 (SGPR)
#define OP8(S)                                                                          \
    asm volatile(S : "+v"(a0) : "v"(b), "v"(c), "s"(sg) : "vcc", "s20", "s21");                      \
    asm volatile(S : "+v"(a1) : "v"(c), "v"(b), "s"(sg) : "vcc", "s20", "s21");                      \
    asm volatile(S : "+v"(a2) : "v"(b), "v"(c), "s"(sg) : "vcc", "s20", "s21");                      \
    asm volatile(S : "+v"(a3) : "v"(c), "v"(b), "s"(sg) : "vcc", "s20", "s21");                      \
    asm volatile(S : "+v"(a4) : "v"(b), "v"(c), "s"(sg) : "vcc", "s20", "s21");                      \
    asm volatile(S : "+v"(a5) : "v"(c), "v"(b), "s"(sg) : "vcc", "s20", "s21");                      \
    asm volatile(S : "+v"(a6) : "v"(b), "v"(c), "s"(sg) : "vcc", "s20", "s21");                      \
    asm volatile(S : "+v"(a7) : "v"(c), "v"(b), "s"(sg) : "vcc", "s20", "s21");

#define KERNEL(NAME, S)                                                                 \
    __global__ __launch_bounds__(64) void NAME(uint32_t *out, int iters)                \
    {                                                                                   \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4,      \
                 a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;                  \
        const uint32_t b = (blockIdx.x * 2654435761u + 12345u) | 1u, c = (b ^ 0x5bd1e995u) & 31u;            \
        const uint32_t sg = __builtin_amdgcn_readfirstlane(b);                          \
        for (int it = 0; it < iters; it++) {                                            \
            _Pragma("unroll") for (int r = 0; r < 8; r++) { OP8(S) }                    \
        }                                                                               \
        out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;     \
    }

KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_or, "v_or_b32 %0, %0, %1")
KERNEL(k_add, "v_add_u32 %0, %0, %1")
KERNEL(k_sub, "v_sub_u32 %0, %0, %1")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 1, %0")
KERNEL(k_lshl, "v_lshlrev_b32 %0, %2, %0")
KERNEL(k_min, "v_min_u32 %0, %0, %1")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_ffbl, "v_ffbl_b32 %0, %0")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL(k_min3, "v_min3_u32 %0, %0, %1, %2")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
KERNEL(k_bfe_u, "v_bfe_u32 %0, %0, 6, 11")
KERNEL(k_bfe_i, "v_bfe_i32 %0, %0, 17, 1")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
KERNEL(k_sad, "v_sad_u32 %0, %0, %1, %2")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 8, %1")
KERNEL(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 2")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 9, %1")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_sdwa, "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
KERNEL(k_mbcnt_lo, "v_mbcnt_lo_u32_b32 %0, %3, %0")
KERNEL(k_mbcnt_hi, "v_mbcnt_hi_u32_b32 %0, %3, %0")
KERNEL(k_cmp, "v_cmp_ne_u32 vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_xor_sgpr, "v_xor_b32 %0, %3, %0")
KERNEL(k_and_lit, "v_and_b32 %0, 0x1fc, %0")
KERNEL(k_or_inl, "v_or_b32 %0, 32, %0")
KERNEL(k_bfm, "v_bfm_b32 %0, %0, %1")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_subrev, "v_subrev_u32 %0, %3, %0")
KERNEL(k_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_lshl_c, "v_lshlrev_b32 %0, 1, %0")
KERNEL(k_lshr_v, "v_lshrrev_b32 %0, %2, %0")
KERNEL(k_ashr_c, "v_ashrrev_i32 %0, 31, %0")
KERNEL(k_ashr_v, "v_ashrrev_i32 %0, %2, %0")
KERNEL(k_not, "v_not_b32 %0, %0")
KERNEL(k_max, "v_max_u32 %0, %0, %1")
KERNEL(k_add_sgpr, "v_add_u32 %0, %3, %0")
KERNEL(k_add_lit, "v_add_u32 %0, 0x12345, %0")
KERNEL(k_sub_inl, "v_subrev_u32 %0, 1, %0")
KERNEL(k_bitop3_inl, "v_bitop3_b32 %0, %0, -8, %1 bitop3:0xf8")
KERNEL(k_bitop3_2src, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x60")
KERNEL(k_xor_e64, "v_xor_b32_e64 %0, %0, %1")
KERNEL(k_cmp_vcc, "v_cmp_ne_u32 vcc, %0, %1")
KERNEL(k_cmp_sgpr, "v_cmp_gt_u32_e64 s[20:21], %0, %1")
KERNEL(k_cndmask_s, "v_cndmask_b32_e64 %0, %0, %1, s[22:23]")
KERNEL(k_addco, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL(k_pk_add, "v_pk_add_u16 %0, %0, %1")
KERNEL(k_pk_lshr, "v_pk_lshrrev_b16 %0, 1, %0")
KERNEL(k_pk_mad, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL(k_pk_min, "v_pk_min_u16 %0, %0, %1")
KERNEL(k_and_sgpr, "v_and_b32 %0, %3, %0")
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL(k_ffbh, "v_ffbh_u32 %0, %0")
KERNEL(k_bfrev, "v_bfrev_b32 %0, %0")
KERNEL(k_cvt, "v_cvt_f32_ubyte0 %0, %0")
KERNEL(k_fadd, "v_add_f32 %0, %0, %1")
KERNEL(k_ffma, "v_fma_f32 %0, %0, %1, %2")

// SGPR-operand instructions in a mix: does it matter whether the two per word sit next to each other (as the kernel's
// mismatch-bit step has them: v_xor with an SGPR, then v_bitop3 with an SGPR) or are spread among VGPR-only ones?
#define MIXK(NAME, BODY)                                                                \
    __global__ __launch_bounds__(64) void NAME(uint32_t *out, int iters)                \
    {                                                                                   \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4,      \
                 a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;                  \
        const uint32_t b = (blockIdx.x * 2654435761u + 12345u) | 1u, c = (b ^ 0x5bd1e995u) & 31u;            \
        const uint32_t sg = __builtin_amdgcn_readfirstlane(b), sh = __builtin_amdgcn_readfirstlane(c + 77u);  \
        for (int it = 0; it < iters; it++) {                                            \
            _Pragma("unroll") for (int r = 0; r < 8; r++) { BODY }                      \
        }                                                                               \
        out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;     \
    }
#define SX(A) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(A) : "s"(sg));
#define SB(A) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(A) : "v"(b), "s"(sh));
#define VB(A) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(A) : "v"(b), "v"(c));
#define VX(A) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(A) : "v"(c));
MIXK(k_mix_adjacent, SX(a0) SB(a1) VB(a2) VB(a3) VX(a4) VB(a5) VX(a6) VB(a7))
MIXK(k_mix_spread, SX(a0) VB(a2) VB(a3) VX(a4) SB(a1) VB(a5) VX(a6) VB(a7))
MIXK(k_mix_same_sgpr, asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a0) : "s"(sg)); asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(a1) : "v"(b), "s"(sg)); VB(a2) VB(a3) VX(a4) VB(a5) VX(a6) VB(a7))
MIXK(k_mix_none, VX(a0) VB(a1) VB(a2) VB(a3) VX(a4) VB(a5) VX(a6) VB(a7))
MIXK(k_mix_one, SX(a0) VB(a1) VB(a2) VB(a3) VX(a4) VB(a5) VX(a6) VB(a7))
MIXK(k_mix_half_adjacent, asm volatile("v_ffbl_b32 %0, %0" : "+v"(a0)); asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a1) : "v"(b)); VB(a2) VB(a3) VX(a4) VB(a5) VX(a6) VB(a7))

// what ONE half-rate instruction costs among seven full-rate ones (the trips are such mixes)
#define HMIX(NAME, HOP) MIXK(NAME, asm volatile(HOP : "+v"(a0) : "v"(b), "v"(c) : "vcc"); VB(a1) VB(a2) VB(a3) VX(a4) VB(a5) VX(a6) VB(a7))
HMIX(k_h1_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
HMIX(k_h1_ffbl, "v_ffbl_b32 %0, %0")
HMIX(k_h1_cmp, "v_cmp_ne_u32 vcc, %0, %1")
HMIX(k_h1_mad, "v_mad_u32_u24 %0, %0, %1, %2")
HMIX(k_h1_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
HMIX(k_h1_sad, "v_sad_u32 %0, %0, %1, %2")
HMIX(k_h1_lshl, "v_lshlrev_b32 %0, 1, %0")
HMIX(k_h1_min3, "v_min3_u32 %0, %0, %1, %2")
HMIX(k_h1_mbcnt, "v_mbcnt_lo_u32_b32 %0, -1, %0")
HMIX(k_h1_lshl_add, "v_lshl_add_u32 %0, %0, 8, %1")
HMIX(k_h1_mul24, "v_mul_u32_u24 %0, %0, %1")
HMIX(k_h1_bfe, "v_bfe_u32 %0, %0, 5, 1")
HMIX(k_h1_add3, "v_add3_u32 %0, %0, %1, %2")
MIXK(k_h4_mixed, asm volatile("v_ffbl_b32 %0, %0" : "+v"(a0)); VB(a1) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a2) : "v"(b)); VB(a3) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c)); VB(a5) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "v"(c)); VB(a7))
MIXK(k_h8_mixed, asm volatile("v_ffbl_b32 %0, %0" : "+v"(a0)); asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c)); asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c)); asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(a5) : "v"(b)); asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "v"(c)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a7) : "v"(b));)

typedef void (*kern_t)(uint32_t *, int);

static int run(const char *name, kern_t k, int per_iter, uint32_t *buf, FILE *js, bool first)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int iters = 10000;
    double cyc[2] = {0, 0};
    int i = 0;
    for (int wps : {4, 8}) {
        const int blocks = 256 * 4 * wps;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, buf, 100);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, buf, iters);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double winst = (double)blocks * iters * per_iter;
        cyc[i++] = 2.4e9 * 1024 / (winst / (ms * 1e-3));
    }
    printf("%-28s cycles per instruction and SIMD (at 2.4 GHz): %5.2f at 4 waves/SIMD, %5.2f at 8\n", name, cyc[0], cyc[1]);
    if (js) fprintf(js, "%s\"%s\": [%.3f, %.3f]", first ? "" : ", ", name, cyc[0], cyc[1]);
    return 0;
}

int main()
{
    uint32_t *buf;
    CHK(hipMalloc(&buf, 256 * 4 * 8 * 64 * sizeof(uint32_t)));
    FILE *js = fopen("gpurun_out/valu_ops.json", "w");
    if (js) fprintf(js, "{");
    bool first = true;
#define RUN(NAME, K, N) if (run(NAME, K, N, buf, js, first)) return 1; first = false;
    RUN("v_xor_b32", k_xor, 64)
    RUN("v_and_b32", k_and, 64)
    RUN("v_or_b32", k_or, 64)
    RUN("v_add_u32", k_add, 64)
    RUN("v_sub_u32", k_sub, 64)
    RUN("v_lshrrev_b32 (const)", k_lshr, 64)
    RUN("v_lshlrev_b32 (vgpr)", k_lshl, 64)
    RUN("v_min_u32", k_min, 64)
    RUN("v_mov_b32", k_mov, 64)
    RUN("v_ffbl_b32", k_ffbl, 64)
    RUN("v_bcnt_u32_b32", k_bcnt, 64)
    RUN("v_bitop3_b32", k_bitop3, 64)
    RUN("v_min3_u32", k_min3, 64)
    RUN("v_mad_u32_u24", k_mad24, 64)
    RUN("v_mul_u32_u24", k_mul24, 64)
    RUN("v_mul_lo_u32", k_mullo, 64)
    RUN("v_mul_hi_u32", k_mulhi, 64)
    RUN("v_bfe_u32", k_bfe_u, 64)
    RUN("v_bfe_i32", k_bfe_i, 64)
    RUN("v_alignbit_b32", k_alignbit, 64)
    RUN("v_sad_u32", k_sad, 64)
    RUN("v_lshl_add_u32", k_lshl_add, 64)
    RUN("v_add_lshl_u32", k_add_lshl, 64)
    RUN("v_lshl_or_b32", k_lshl_or, 64)
    RUN("v_add3_u32", k_add3, 64)
    RUN("v_and_or_b32", k_and_or, 64)
    RUN("v_and_b32 sdwa", k_sdwa, 64)
    RUN("v_mbcnt_lo_u32_b32", k_mbcnt_lo, 64)
    RUN("v_mbcnt_hi_u32_b32", k_mbcnt_hi, 64)
    RUN("v_cmp_ne + v_addc (pair)", k_cmp, 128)
    RUN("v_cndmask_b32", k_cndmask, 64)
    RUN("v_xor_b32 sgpr operand", k_xor_sgpr, 64)
    RUN("v_and_b32 literal", k_and_lit, 64)
    RUN("v_or_b32 inline const", k_or_inl, 64)
    RUN("v_bfm_b32", k_bfm, 64)
    RUN("v_perm_b32", k_perm, 64)
    RUN("v_subrev_u32 sgpr", k_subrev, 64)
    RUN("v_mov_b32 dpp row_shr", k_dpp, 64)
    RUN("v_lshlrev_b32 (const)", k_lshl_c, 64)
    RUN("v_lshrrev_b32 (vgpr)", k_lshr_v, 64)
    RUN("v_ashrrev_i32 (const)", k_ashr_c, 64)
    RUN("v_ashrrev_i32 (vgpr)", k_ashr_v, 64)
    RUN("v_not_b32", k_not, 64)
    RUN("v_max_u32", k_max, 64)
    RUN("v_add_u32 sgpr operand", k_add_sgpr, 64)
    RUN("v_add_u32 literal", k_add_lit, 64)
    RUN("v_subrev_u32 inline const", k_sub_inl, 64)
    RUN("v_bitop3_b32 inline const", k_bitop3_inl, 64)
    RUN("v_bitop3_b32 two sources", k_bitop3_2src, 64)
    RUN("v_xor_b32_e64 (VOP3, vgprs)", k_xor_e64, 64)
    RUN("v_cmp_ne_u32 -> vcc", k_cmp_vcc, 64)
    RUN("v_cmp_gt_u32 -> sgpr pair", k_cmp_sgpr, 64)
    RUN("v_cndmask_b32 sgpr-pair mask", k_cndmask_s, 64)
    RUN("v_add_co_u32", k_addco, 64)
    RUN("v_pk_add_u16", k_pk_add, 64)
    RUN("v_pk_lshrrev_b16", k_pk_lshr, 64)
    RUN("v_pk_mad_u16", k_pk_mad, 64)
    RUN("v_pk_min_u16", k_pk_min, 64)
    RUN("v_and_b32 sgpr operand", k_and_sgpr, 64)
    RUN("v_bfi_b32", k_bfi, 64)
    RUN("v_xad_u32", k_xad, 64)
    RUN("v_or3_b32", k_or3, 64)
    RUN("v_ffbh_u32", k_ffbh, 64)
    RUN("v_bfrev_b32", k_bfrev, 64)
    RUN("v_cvt_f32_ubyte0", k_cvt, 64)
    RUN("v_add_f32", k_fadd, 64)
    RUN("v_fma_f32", k_ffma, 64)
    RUN("mix 8: no SGPR operand", k_mix_none, 64)
    RUN("mix 8: 1 SGPR-operand instr", k_mix_one, 64)
    RUN("mix 8: 2 SGPR-operand, adjacent", k_mix_adjacent, 64)
    RUN("mix 8: 2 SGPR-operand, spread", k_mix_spread, 64)
    RUN("mix 8: 2 adjacent, same SGPR", k_mix_same_sgpr, 64)
    RUN("mix 8: 2 half-rate (ffbl, bcnt)", k_mix_half_adjacent, 64)
    RUN("mix 8: 1 v_bcnt + 7 full", k_h1_bcnt, 64)
    RUN("mix 8: 1 v_ffbl + 7 full", k_h1_ffbl, 64)
    RUN("mix 8: 1 v_cmp + 7 full", k_h1_cmp, 64)
    RUN("mix 8: 1 v_mad_u32_u24 + 7 full", k_h1_mad, 64)
    RUN("mix 8: 1 v_alignbit + 7 full", k_h1_alignbit, 64)
    RUN("mix 8: 1 v_sad_u32 + 7 full", k_h1_sad, 64)
    RUN("mix 8: 1 v_lshlrev + 7 full", k_h1_lshl, 64)
    RUN("mix 8: 1 v_min3_u32 + 7 full", k_h1_min3, 64)
    RUN("mix 8: 1 v_mbcnt_lo + 7 full", k_h1_mbcnt, 64)
    RUN("mix 8: 1 v_lshl_add + 7 full", k_h1_lshl_add, 64)
    RUN("mix 8: 1 v_mul_u32_u24 + 7 full", k_h1_mul24, 64)
    RUN("mix 8: 1 v_bfe_u32 + 7 full", k_h1_bfe, 64)
    RUN("mix 8: 1 v_add3_u32 + 7 full", k_h1_add3, 64)
    RUN("mix 8: 4 half-rate alternating", k_h4_mixed, 64)
    RUN("mix 8: 8 different half-rate", k_h8_mixed, 64)
    if (js) { fprintf(js, "}\n"); fclose(js); }
    return 0;
}
