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
    return 0;
}
