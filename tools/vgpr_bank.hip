// vgpr_bank.hip -- does the VGPR bank (register index mod 4) of the source operands change the issue rate of
// 2- and 3-source VALU instructions on MI355X?   hipcc --offload-arch=gfx950 -O3 tools/vgpr_bank.hip -o tools/vgpr_bank
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// 8 independent accumulators v20..v27 (banks 0,1,2,3,0,1,2,3); sources picked per KIND
#define BODY3(S1, S2)                                              \
    "v_bitop3_b32 v20, v20, " S1 ", " S2 " bitop3:0x96\n"          \
    "v_bitop3_b32 v21, v21, " S1 ", " S2 " bitop3:0xe8\n"          \
    "v_bitop3_b32 v22, v22, " S1 ", " S2 " bitop3:0x96\n"          \
    "v_bitop3_b32 v23, v23, " S1 ", " S2 " bitop3:0xe8\n"          \
    "v_bitop3_b32 v24, v24, " S1 ", " S2 " bitop3:0x96\n"          \
    "v_bitop3_b32 v25, v25, " S1 ", " S2 " bitop3:0xe8\n"          \
    "v_bitop3_b32 v26, v26, " S1 ", " S2 " bitop3:0x96\n"          \
    "v_bitop3_b32 v27, v27, " S1 ", " S2 " bitop3:0xe8\n"
// every instruction with its three sources in three different banks / in one bank
#define DISTINCT                                                   \
    "v_bitop3_b32 v20, v20, v13, v14 bitop3:0x96\n"                \
    "v_bitop3_b32 v21, v21, v14, v15 bitop3:0xe8\n"                \
    "v_bitop3_b32 v22, v22, v15, v12 bitop3:0x96\n"                \
    "v_bitop3_b32 v23, v23, v12, v13 bitop3:0xe8\n"                \
    "v_bitop3_b32 v24, v24, v13, v14 bitop3:0x96\n"                \
    "v_bitop3_b32 v25, v25, v14, v15 bitop3:0xe8\n"                \
    "v_bitop3_b32 v26, v26, v15, v12 bitop3:0x96\n"                \
    "v_bitop3_b32 v27, v27, v12, v13 bitop3:0xe8\n"
#define SAMEBANK                                                   \
    "v_bitop3_b32 v20, v20, v12, v16 bitop3:0x96\n"                \
    "v_bitop3_b32 v21, v21, v13, v17 bitop3:0xe8\n"                \
    "v_bitop3_b32 v22, v22, v14, v18 bitop3:0x96\n"                \
    "v_bitop3_b32 v23, v23, v15, v19 bitop3:0xe8\n"                \
    "v_bitop3_b32 v24, v24, v12, v16 bitop3:0x96\n"                \
    "v_bitop3_b32 v25, v25, v13, v17 bitop3:0xe8\n"                \
    "v_bitop3_b32 v26, v26, v14, v18 bitop3:0x96\n"                \
    "v_bitop3_b32 v27, v27, v15, v19 bitop3:0xe8\n"
#define XOR2_DISTINCT                                              \
    "v_xor_b32 v20, v20, v13\n v_xor_b32 v21, v21, v14\n v_xor_b32 v22, v22, v15\n v_xor_b32 v23, v23, v12\n" \
    "v_xor_b32 v24, v24, v13\n v_xor_b32 v25, v25, v14\n v_xor_b32 v26, v26, v15\n v_xor_b32 v27, v27, v12\n"
#define XOR2_SAME                                                  \
    "v_xor_b32 v20, v20, v12\n v_xor_b32 v21, v21, v13\n v_xor_b32 v22, v22, v14\n v_xor_b32 v23, v23, v15\n" \
    "v_xor_b32 v24, v24, v12\n v_xor_b32 v25, v25, v13\n v_xor_b32 v26, v26, v14\n v_xor_b32 v27, v27, v15\n"
// non-accumulating: destination differs from every source (no forwarding of the previous result)
#define DST_OTHER                                                  \
    "v_bitop3_b32 v28, v20, v13, v14 bitop3:0x96\n"                \
    "v_bitop3_b32 v29, v21, v14, v15 bitop3:0xe8\n"                \
    "v_bitop3_b32 v30, v22, v15, v12 bitop3:0x96\n"                \
    "v_bitop3_b32 v31, v23, v12, v13 bitop3:0xe8\n"                \
    "v_bitop3_b32 v28, v24, v13, v14 bitop3:0x96\n"                \
    "v_bitop3_b32 v29, v25, v14, v15 bitop3:0xe8\n"                \
    "v_bitop3_b32 v30, v26, v15, v12 bitop3:0x96\n"                \
    "v_bitop3_b32 v31, v27, v12, v13 bitop3:0xe8\n"

#define CLOB "v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31"
#define INIT "v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n" \
             "v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n"

template <int KIND>
__global__ __launch_bounds__(64) void k_spin(uint32_t *out, int iters)
{
    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x, r;
    asm volatile(INIT ::"v"(seed) : CLOB);
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) asm volatile(BODY3("v13", "v14") BODY3("v13", "v14") BODY3("v13", "v14") BODY3("v13", "v14") ::: CLOB);
        if (KIND == 1) asm volatile(DISTINCT DISTINCT DISTINCT DISTINCT ::: CLOB);
        if (KIND == 2) asm volatile(SAMEBANK SAMEBANK SAMEBANK SAMEBANK ::: CLOB);
        if (KIND == 3) asm volatile(XOR2_DISTINCT XOR2_DISTINCT XOR2_DISTINCT XOR2_DISTINCT ::: CLOB);
        if (KIND == 4) asm volatile(XOR2_SAME XOR2_SAME XOR2_SAME XOR2_SAME ::: CLOB);
        if (KIND == 5) asm volatile(DST_OTHER DST_OTHER DST_OTHER DST_OTHER ::: CLOB);
    }
    asm volatile("v_xor_b32 %0, v20, v21\n v_xor_b32 %0, %0, v22\n v_xor_b32 %0, %0, v23\n v_xor_b32 %0, %0, v28\n v_xor_b32 %0, %0, v29" : "=v"(r)::CLOB);
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <int KIND>
static int run(const char *name, uint32_t *buf)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int iters = 40000;
    for (int wps : {1, 2, 4, 7, 8}) {
        const int blocks = 256 * 4 * wps;
        hipLaunchKernelGGL(k_spin<KIND>, dim3(blocks), dim3(64), 0, 0, buf, 100);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_spin<KIND>, dim3(blocks), dim3(64), 0, 0, buf, iters);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double winst = (double)blocks * iters * 32.0;
        const double rate = winst / (ms * 1e-3);
        printf("%-44s waves/SIMD=%d  %.3e wave-instr/s = %.1f Tlane-op/s  cycles/instr/SIMD @2.4GHz = %.2f\n", name, wps, rate,
               rate * 64 / 1e12, 2.4e9 * 1024 / rate);
    }
    return 0;
}

int main()
{
    uint32_t *buf;
    CHK(hipMalloc(&buf, 256 * 4 * 8 * 64 * sizeof(uint32_t)));
    if (run<0>("bitop3 acc, fixed v13 v14 (mixed banks)", buf)) return 1;
    if (run<1>("bitop3 acc, three sources in 3 banks", buf)) return 1;
    if (run<2>("bitop3 acc, three sources in ONE bank", buf)) return 1;
    if (run<3>("xor acc, two sources in 2 banks", buf)) return 1;
    if (run<4>("xor acc, two sources in ONE bank", buf)) return 1;
    if (run<5>("bitop3 dst != src, sources in 3 banks", buf)) return 1;
    return 0;
}
