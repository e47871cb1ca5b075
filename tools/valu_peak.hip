// valu_peak.hip -- empirical integer-VALU issue rate on MI355X (gfx950).
// Measures wave64 instructions per second for the instruction kinds the bit-sliced gkm
// kernel is made of, at 1..8 wavefronts per SIMD.  Used to price roofline.peak.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o tools/valu_peak && tools/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(64) void k_spin(uint32_t *out, int iters)
{
    uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5,
             a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    const uint32_t b = blockIdx.x * 2654435761u + 12345u, c = b ^ 0x5bd1e995u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (KIND == 0) {
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(b));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a1) : "v"(c));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a2) : "v"(b));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a3) : "v"(c));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a4) : "v"(b));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a5) : "v"(c));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a6) : "v"(b));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a7) : "v"(c));
            } else if (KIND == 1) {
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a0) : "v"(b), "v"(c));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8" : "+v"(a1) : "v"(b), "v"(c));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a2) : "v"(b), "v"(c));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8" : "+v"(a3) : "v"(b), "v"(c));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a4) : "v"(b), "v"(c));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8" : "+v"(a5) : "v"(b), "v"(c));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a6) : "v"(b), "v"(c));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8" : "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 2) {
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
                asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
                asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c));
                asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a5) : "v"(b), "v"(c));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "v"(c));
                asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 3) { /* xor with an SGPR operand, as the kernel's Z step */
                asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a0) : "s"(b));
                asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a1) : "s"(c));
                asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a2) : "s"(b));
                asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a3) : "s"(c));
                asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a4) : "s"(b));
                asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a5) : "s"(c));
                asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a6) : "s"(b));
                asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a7) : "s"(c));
            } else if (KIND == 5) { /* 3-input op with an SGPR as third operand, as the kernel's Z step */
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(a0) : "v"(b), "s"(c));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(a1) : "v"(c), "s"(b));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(a2) : "v"(b), "s"(c));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(a3) : "v"(c), "s"(b));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(a4) : "v"(b), "s"(c));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(a5) : "v"(c), "s"(b));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(a6) : "v"(b), "s"(c));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(a7) : "v"(c), "s"(b));
            } else if (KIND == 6) { /* VOP3 encoding of xor, SGPR as second source */
                asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(a0) : "s"(b));
                asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(a1) : "s"(c));
                asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(a2) : "s"(b));
                asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(a3) : "s"(c));
                asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(a4) : "s"(b));
                asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(a5) : "s"(c));
                asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(a6) : "s"(b));
                asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(a7) : "s"(c));
            } else if (KIND == 7) { /* the kernel's mix: 2 SGPR-operand ops among 6 all-VGPR ones */
                asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a0) : "s"(b));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(a1) : "v"(b), "s"(c));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a2) : "v"(b), "v"(c));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8" : "+v"(a3) : "v"(b), "v"(c));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a4) : "v"(b));
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x60" : "+v"(a5) : "v"(b), "v"(c));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a6) : "v"(c));
                asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a7));
            } else if (KIND == 8) { /* xor with an inline constant (no register read for that operand) */
                asm volatile("v_xor_b32 %0, 0x55, %0" : "+v"(a0));
                asm volatile("v_xor_b32 %0, 0x33, %0" : "+v"(a1));
                asm volatile("v_xor_b32 %0, 0x55, %0" : "+v"(a2));
                asm volatile("v_xor_b32 %0, 0x33, %0" : "+v"(a3));
                asm volatile("v_xor_b32 %0, 0x55, %0" : "+v"(a4));
                asm volatile("v_xor_b32 %0, 0x33, %0" : "+v"(a5));
                asm volatile("v_xor_b32 %0, 0x55, %0" : "+v"(a6));
                asm volatile("v_xor_b32 %0, 0x33, %0" : "+v"(a7));
            } else if (KIND == 9) { /* SGPR operand read through DPP-free v_mov once, then all-VGPR: 1 mov per 7 ops */
                uint32_t t;
                asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "s"(b));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a1) : "v"(t));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a2) : "v"(t));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a3) : "v"(t));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a4) : "v"(t));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a5) : "v"(t));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a6) : "v"(t));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a7) : "v"(t));
            } else { /* dependent chain: latency-bound single accumulator */
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(b));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(c));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(b));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(c));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(b));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(c));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(b));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(c));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

static double g_best_full = 0;

template <int KIND>
static int run(const char *name, uint32_t *buf)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int wps : {1, 2, 3, 4, 8}) {
        const int blocks = 256 * 4 * wps; /* one 64-thread block per wave slot */
        hipLaunchKernelGGL(k_spin<KIND>, dim3(blocks), dim3(64), 0, 0, buf, 100);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_spin<KIND>, dim3(blocks), dim3(64), 0, 0, buf, iters);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double winst = (double)blocks * iters * 64.0; /* wave instructions */
        const double rate = winst / (ms * 1e-3);
        if ((KIND == 0 || KIND == 1 || KIND == 4) && rate * 64 / 1e9 > g_best_full) g_best_full = rate * 64 / 1e9;
        printf("%-22s waves/SIMD=%d  %.1f ms  %.3e wave-instr/s  = %.1f Gop/s (x64 lanes)  cycles/instr/SIMD @2.4GHz = %.2f\n",
               name, wps, ms, rate, rate * 64 / 1e9, 2.4e9 * 1024 / rate);
    }
    return 0;
}

int main()
{
    uint32_t *buf;
    CHK(hipMalloc(&buf, 256 * 4 * 8 * 64 * sizeof(uint32_t)));
    if (run<0>("v_xor_b32", buf)) return 1;
    if (run<1>("v_bitop3_b32", buf)) return 1;
    if (run<2>("v_and_or/v_or3", buf)) return 1;
    if (run<3>("v_xor_b32 sgpr-src", buf)) return 1;
    if (run<4>("v_xor_b32 dependent", buf)) return 1;
    if (run<5>("v_bitop3 sgpr-src2", buf)) return 1;
    if (run<6>("v_xor_b32_e64 sgpr", buf)) return 1;
    if (run<7>("kernel mix 2s+6v", buf)) return 1;
    if (run<8>("v_xor_b32 inline-const", buf)) return 1;
    if (run<9>("v_mov sgpr + 7 xor", buf)) return 1;
    /* machine-readable summary for bench.py (roofline.peak_measured): the best sustained rate of the
     * full-rate kinds (all-VGPR v_xor_b32 / v_bitop3_b32 streams) */
    FILE *f = fopen("gpurun_out/valu_peak.json", "w");
    if (f) {
        fprintf(f, "{\"best_full_rate_Gops\": %.1f, \"nominal_Gops\": %.1f, \"what\": \"tools/valu_peak.hip: best of the v_xor_b32 / v_bitop3_b32 all-VGPR streams at 1..8 waves per SIMD, wave-instructions/s x 64 lanes\"}\n",
                g_best_full, 256 * 4 * 32 * 2.4);
        fclose(f);
    }
    return 0;
}
