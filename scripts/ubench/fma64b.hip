// micro-benchmark: chip-level fp64 FMA throughput vs waves per CU
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_fma (double* out, int iters)
{
    double a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x*1e-3 + i;
    const double b = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = fma(a[i], b, c);
    }
    double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[(size_t)blockIdx.x*blockDim.x + threadIdx.x] = s;
}
int main ()
{
    double* out; (void)hipMalloc(&out, 1 << 26);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int threads : {64, 128, 256, 512, 1024}) for (int blocks : {256, 512}) {
        k_fma<<<blocks, threads>>>(out, 100);
        (void)hipEventRecord(e0); k_fma<<<blocks, threads>>>(out, 20000); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double flops = 2.0*16*20000*(double)threads*blocks;
        printf("blocks %4d x %4d threads: %.3f ms, %.1f TFLOP/s fp64\n", blocks, threads, ms, flops/ms*1e-9);
    }
    return 0;
}
