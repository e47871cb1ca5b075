// How many one-wave workgroups fit a CU for a given LDS size?  (hipcc --offload-arch=gfx950 -O2 -o tools/lds_occupancy tools/lds_occupancy.hip)
// Each block spins ~20 us; the launch time of a fixed number of blocks gives the number resident per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ unsigned s_dyn[];
__global__ __launch_bounds__(64) void spin(long long ticks, unsigned *sink)
{
    s_dyn[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (s_dyn[threadIdx.x] == 12345u) sink[0] = 1;
}
int main()
{
    unsigned *sink;
    hipMalloc(&sink, 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const long long ticks = 2000; /* 100 MHz wall clock: 20 us */
    const int per_cu = 32 * 12;   /* blocks per CU in the launch */
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    printf("CUs %d\n", cus);
    for (int bytes = 3584; bytes <= 8192; bytes += 128) {
        hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        spin<<<cus * 32, 64, bytes>>>(ticks, sink);
        hipDeviceSynchronize();
        hipEventRecord(a);
        spin<<<cus * per_cu, 64, bytes>>>(ticks, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        /* rounds = per_cu / resident  ->  resident = per_cu * 0.02 ms / ms */
        printf("LDS %5d B: %.3f ms -> %.1f blocks resident per CU\n", bytes, ms, per_cu * 0.020 / ms);
    }
    return 0;
}
