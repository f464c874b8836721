// micro-benchmark: how many kernels from different HIP streams run at the same time on gfx950?
// K streams, each gets R launches of a one-workgroup kernel that spins for ~100 us; wall time / (R * 100 us) = K / concurrency.
// Run with GPU_MAX_HW_QUEUES=4 (default) and =8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
__global__ void k_spin (long long ticks, long long* out)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (out) out[0] = wall_clock64() - t0;
}
int main ()
{
    int rate = 0; (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);      // kHz
    const long long ticks = (long long)rate*100/1000;          // 100 us
    printf("wall clock %d kHz, GPU_MAX_HW_QUEUES=%s\n", rate, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)");
    const int R = 50;
    hipStream_t st[16];
    for (int i = 0; i < 16; ++i) (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    for (int K : {1, 2, 3, 4, 6, 8, 12, 16}) {
        for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], ticks/10, nullptr);
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < R; ++r) for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], ticks, nullptr);
        (void)hipDeviceSynchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("K = %2d streams: %8.0f us for %d x %d launches of 100 us -> %.2f kernels at a time\n", K, us, K, R, K*R*100.0/us);
    }
    return 0;
}
