// micro-benchmark: does a high-priority HIP stream get its workgroups onto a GPU that a big kernel of another stream fills?
// stream A: kernels of 4096 workgroups x 256 threads, 64 KB of LDS each (2 per CU), every workgroup spins 30 us -> the chip is
// full for ~160 us per launch.  stream B: a chain of 40 dependent small kernels (256 workgroups x 256 threads, 3 us each).
// Chain time alone / beside A with equal priorities / with B on a high-priority stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ __launch_bounds__(256) void k_big (long long ticks, double* out)
{
    extern __shared__ double lds[];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (out && lds[threadIdx.x] < 0) out[0] = 1;
}
__global__ __launch_bounds__(256) void k_small (long long ticks, double* out)
{
    __shared__ double buf[5120];                 // 40 KB: does not fit beside two workgroups of k_big on a CU (2 x 64 KB of 160 KB)
    buf[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (out && buf[threadIdx.x] < 0) out[0] = 1;
}
static double chain (hipStream_t sb, int n, long long t3)
{
    (void)hipStreamSynchronize(sb);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, sb, t3, nullptr);
    (void)hipStreamSynchronize(sb);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
int main ()
{
    int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("stream priority range: least %d, greatest %d\n", lo, hi);
    const long long t30 = 3000, t3 = 300;       // wall clock 100 MHz
    (void)hipFuncSetAttribute((const void*)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipStream_t sa, sb, sbh;
    (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    (void)hipStreamCreateWithPriority(&sbh, hipStreamNonBlocking, hi);
    chain(sb, 40, t3); chain(sbh, 40, t3);
    printf("chain of 40 small kernels alone:                      %8.0f us (normal stream) %8.0f us (high-priority stream)\n", chain(sb, 40, t3), chain(sbh, 40, t3));
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(k_big, dim3(4096), dim3(256), 65536, sa, t30, nullptr);
        const double a = chain(sb, 40, t3);
        (void)hipStreamSynchronize(sa);
        for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(k_big, dim3(4096), dim3(256), 65536, sa, t30, nullptr);
        const double b = chain(sbh, 40, t3);
        (void)hipStreamSynchronize(sa);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(k_big, dim3(4096), dim3(256), 65536, sa, t30, nullptr);
        (void)hipStreamSynchronize(sa);
        const double c = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("beside 12 chip-filling kernels (%.0f us alone): chain %8.0f us (normal stream) %8.0f us (high-priority stream)\n", c, a, b);
    }
    return 0;
}
