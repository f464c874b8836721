// micro-benchmark: WHICH pairs of HIP streams of one process run kernels side by side on gfx950, as a function of the order
// (and priority) in which the streams were created?  argv[1] = creation order, one letter per stream: n = normal priority,
// h = high priority (e.g. "nnhhnn": an engine's stream, a second one, the ring's two streams, two more engines).  Every
// stream gets its first command right after it is created (the runtime acquires the hardware queue then).  For every pair:
// R launches of a one-workgroup 100 us kernel on both streams; 1.0 = they took turns, 2.0 = side by side.
// Run with GPU_MAX_HW_QUEUES unset (4) and =8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>
__global__ void k_spin (long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
int main (int argc, char** argv)
{
    const char* order = argc > 1 ? argv[1] : "nnnnnnnn";
    const int N = (int)strlen(order);
    int rate = 0; (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    const long long ticks = (long long)rate*100/1000;
    int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("order %s, GPU_MAX_HW_QUEUES=%s, priority range %d..%d\n", order, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)", lo, hi);
    std::vector<hipStream_t> st((size_t)N);
    for (int i = 0; i < N; ++i) {
        if (order[i] == 'h') (void)hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, hi);
        else (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], ticks/10);
        (void)hipStreamSynchronize(st[i]);
    }
    const int R = 20;
    printf("      ");
    for (int j = 0; j < N; ++j) printf("  %c%-2d", order[j], j);
    printf("\n");
    for (int i = 0; i < N; ++i) {
        printf("  %c%-2d ", order[i], i);
        for (int j = 0; j < N; ++j) {
            if (j <= i) { printf("     "); continue; }
            (void)hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < R; ++r) {
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], ticks);
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[j], ticks);
            }
            (void)hipDeviceSynchronize();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf(" %4.1f", 2.0*R*100.0/us);
        }
        printf("\n");
    }
    return 0;
}
