// micro-benchmark: what does a fork/join of a short kernel onto a second stream cost on gfx950, as a function of the pipe
// the second stream's hardware queue sits on?  Per iteration: A (tA us) on s0; fork; B (tB us) on s1 beside C (tC us) on
// s0; join; the whole thing R times, enqueued without a host wait.  argv[1] = number of streams created (and given a first
// command) between s0 and s1: 0 -> s1 is the process's 2nd queue (another pipe), 3 -> its 5th (the pipe of s0).
// argv[2] = join kind: 0 = events, 1 = hipStreamWriteValue64 / hipStreamWaitValue64, 2 = events, join one iteration late
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
__global__ void k_spin (long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
int main (int argc, char** argv)
{
    const int between = argc > 1 ? atoi(argv[1]) : 0;
    const int kind = argc > 2 ? atoi(argv[2]) : 0;
    const int wgs = argc > 3 ? atoi(argv[3]) : 1;
    int rate = 0; (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    auto ticks = [&] (double us) { return (long long)(rate*us/1000.0); };
    hipStream_t s0, s1; std::vector<hipStream_t> mid((size_t)between);
    (void)hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, ticks(1)); (void)hipStreamSynchronize(s0);
    for (auto& s : mid) { (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, ticks(1)); (void)hipStreamSynchronize(s); }
    (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, ticks(1)); (void)hipStreamSynchronize(s1);
    hipEvent_t ef, ej; (void)hipEventCreateWithFlags(&ef, hipEventDisableTiming | hipEventDisableSystemFence); (void)hipEventCreateWithFlags(&ej, hipEventDisableTiming | hipEventDisableSystemFence);
    unsigned long long* flag = nullptr; (void)hipMalloc(&flag, 2*sizeof(unsigned long long)); (void)hipMemset(flag, 0, 16);
    const double tA = 100, tB = 20, tC = 60;
    const int R = 200;
    for (int mode = 0; mode < 2; ++mode) {        // 0: serial on s0; 1: B beside C
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < R; ++r) {
            hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(64), 0, s0, ticks(tA));
            if (mode == 0) {
                hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(64), 0, s0, ticks(tB));
                hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(64), 0, s0, ticks(tC));
            } else if (kind == 0 || kind == 2) {
                if (kind == 2 && r > 0) (void)hipStreamWaitEvent(s0, ej, 0);
                (void)hipEventRecord(ef, s0); (void)hipStreamWaitEvent(s1, ef, 0);
                hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(64), 0, s1, ticks(tB));
                hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(64), 0, s0, ticks(tC));
                (void)hipEventRecord(ej, s1); if (kind == 0) (void)hipStreamWaitEvent(s0, ej, 0);
            } else {
                const unsigned long long seq = (unsigned long long)r + 1;
                (void)hipStreamWriteValue64(s0, flag, seq, 0); (void)hipStreamWaitValue64(s1, flag, seq, hipStreamWaitValueGte, ~0ULL);
                hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(64), 0, s1, ticks(tB));
                hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(64), 0, s0, ticks(tC));
                (void)hipStreamWriteValue64(s1, flag + 1, seq, 0); (void)hipStreamWaitValue64(s0, flag + 1, seq, hipStreamWaitValueGte, ~0ULL);
            }
        }
        (void)hipDeviceSynchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("between %d kind %d wgs %d  %s: %.1f us per iteration (kernels: %.0f serial, %.0f with B beside C)\n", between, kind, wgs,
               mode == 0 ? "serial on s0    " : "B on s1 beside C", us/R, tA + tB + tC, tA + tC);
    }
    return 0;
}
