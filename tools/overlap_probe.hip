// What can run BESIDE a kernel that keeps every wave slot of the GPU full?  (the drop-in call's copy-out pipeline)
//   hipcc --offload-arch=gfx950 -O2 -o tools/overlap_probe tools/overlap_probe.hip
// A "hog" of one-wave workgroups shaped like k_gram_bitslice (72 VGPRs, ~4.7 KB LDS, each alive ~0.5 ms, ~60 ms in all)
// runs on stream A; while it runs, stream B gets ONE of: a linear D2H copy to pinned memory, a pitched (2-D) D2H copy,
// a small kernel of 256-thread workgroups, the same work as 64-thread workgroups.  Printed: when B's work completed,
// relative to the hog's start and end (HIP events + host clock).  "overlapped" = done well before the hog ended.
// Optional argv[1] = number of CUs to leave free for everything else (the hog then runs on a CU-masked stream).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64, 7) void hog(long long ticks, unsigned *sink)
{
    __shared__ unsigned lds[1150];
    unsigned v[56];
#pragma unroll
    for (int i = 0; i < 56; i++) v[i] = threadIdx.x * 2654435761u + i;
    lds[threadIdx.x] = v[3];
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 56; i++) v[i] = v[i] * 1664525u + v[(i + 7) % 56];
    }
    unsigned s = lds[(threadIdx.x + 1) & 63];
#pragma unroll
    for (int i = 0; i < 56; i++) s ^= v[i];
    if (s == 12345u) sink[0] = s;
}

template <int T>
__global__ __launch_bounds__(T) void touch(double *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n; i += (size_t)gridDim.x * T) p[i] = p[i] * 1.0000001 + 1.0;
}

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const int leave = argc > 1 ? atoi(argv[1]) : 0;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned *sink;
    CHK(hipMalloc(&sink, 4));
    const size_t bytes = (size_t)64 << 20, n = bytes / 8;
    double *dev, *host;
    CHK(hipMalloc(&dev, 2 * bytes));
    CHK(hipHostMalloc((void **)&host, bytes, hipHostMallocPortable));
    hipStream_t A, B;
    if (leave > 0) {
        /* one CU less per XCD (the CU index within an XCD is the high part of the bit index on this part: try both
         * interpretations by masking bits spread evenly over the 256) */
        std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0xFFFFFFFFu);
        for (int k = 0; k < leave; k++) {
            const int bit = (int)((long)k * cus / leave);
            mask[(size_t)bit / 32] &= ~(1u << (bit % 32));
        }
        CHK(hipExtStreamCreateWithCUMask(&A, (uint32_t)mask.size(), mask.data()));
    } else {
        CHK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    }
    CHK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    hipEvent_t h0, h1, b1;
    CHK(hipEventCreate(&h0)); CHK(hipEventCreate(&h1)); CHK(hipEventCreate(&b1));
    const long long ticks = 50000;          /* 100 MHz wall clock: 0.5 ms per workgroup */
    const int resident = cus * 4 * 7;        /* 7 waves per SIMD */
    const int rounds = 120;                  /* ~60 ms */
    printf("CUs %d, hog on %s stream (%d CUs left free), %d workgroups\n", cus, leave ? "a CU-MASKED" : "a plain", leave, resident * rounds);
    const char *names[] = {"linear D2H copy, 64 MB", "pitched (2-D) D2H copy, 64 MB", "kernel, 256-thread workgroups", "kernel, 64-thread workgroups", "(nothing)"};
    for (int what = 0; what < 5; what++) {
        CHK(hipDeviceSynchronize());
        const double t0 = now_ms();
        CHK(hipEventRecord(h0, A));
        hipLaunchKernelGGL(hog, dim3((unsigned)(resident * rounds)), dim3(64), 0, A, ticks, sink);
        CHK(hipEventRecord(h1, A));
        /* let the hog fill the machine first */
        while (now_ms() - t0 < 5.0) { }
        const double tb = now_ms();
        if (what == 0) CHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, B));
        if (what == 1) CHK(hipMemcpy2DAsync(host, 4096 * 8, dev, 8192 * 8, 4096 * 8, n / 4096, hipMemcpyDeviceToHost, B));
        if (what == 2) hipLaunchKernelGGL(touch<256>, dim3(2048), dim3(256), 0, B, dev, n);
        if (what == 3) hipLaunchKernelGGL(touch<64>, dim3(8192), dim3(64), 0, B, dev, n);
        CHK(hipEventRecord(b1, B));
        CHK(hipStreamSynchronize(B));
        const double b_done = now_ms();
        CHK(hipStreamSynchronize(A));
        const double a_done = now_ms();
        float hog_ms = 0.f;
        CHK(hipEventElapsedTime(&hog_ms, h0, h1));
        printf("%-32s issued at %5.1f ms, done at %6.1f ms; hog done at %6.1f ms (hog alone on its stream: %.1f ms)  -> %s\n", names[what],
               tb - t0, b_done - t0, a_done - t0, hog_ms, what == 4 ? "-" : (b_done < a_done - 5.0 ? "OVERLAPPED" : "waited for the hog"));
    }
    return 0;
}
