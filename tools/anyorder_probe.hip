// Does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) drop the barrier between two kernels of ONE stream on this GPU?
//   hipcc --offload-arch=gfx950 -O2 -o tools/anyorder_probe tools/anyorder_probe.hip
// Kernel A: one workgroup spinning 20 ms.  Kernel B: one workgroup spinning 1 ms, launched right behind A on the same
// stream, once normally and once with hipExtAnyOrderLaunch.  If B's stop event fires ~19 ms before A's, the flag works.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(long long ticks, unsigned *sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (ticks == 12345) sink[0] = 1;
}
int main()
{
    unsigned *sink;
    CHK(hipMalloc(&sink, 4));
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a0, a1, b0, b1;
    CHK(hipEventCreate(&a0)); CHK(hipEventCreate(&a1)); CHK(hipEventCreate(&b0)); CHK(hipEventCreate(&b1));
    for (int flags = 0; flags <= 1; flags++) {
        long long ta = 2000000, tb = 100000; /* 100 MHz: 20 ms, 1 ms */
        void *argsA[] = {&ta, &sink}, *argsB[] = {&tb, &sink};
        CHK(hipDeviceSynchronize());
        CHK(hipExtLaunchKernel((const void *)spin, dim3(1), dim3(64), argsA, 0, s, a0, a1, 0));
        CHK(hipExtLaunchKernel((const void *)spin, dim3(1), dim3(64), argsB, 0, s, b0, b1, flags));
        CHK(hipStreamSynchronize(s));
        float a = 0, b_after_a0 = 0, b = 0;
        CHK(hipEventElapsedTime(&a, a0, a1));
        CHK(hipEventElapsedTime(&b, b0, b1));
        CHK(hipEventElapsedTime(&b_after_a0, a0, b1));
        printf("flags=%d (%s): A ran %.2f ms, B ran %.2f ms, B finished %.2f ms after A started -> %s\n", flags,
               flags ? "hipExtAnyOrderLaunch" : "in order", a, b, b_after_a0,
               b_after_a0 < a - 5 ? "B OVERTOOK A: no barrier between them" : "B waited for A");
    }
    return 0;
}
