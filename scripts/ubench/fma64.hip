// micro-benchmark: fp64 FMA issue rate and LDS broadcast read rate per wave on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_fma (double* out, long long* t, int iters, int nchains_mode)
{
    double a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x*1e-3 + i;
    const double b = 1.0000001, c = 1e-9;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = fma(a[i], b, c);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_lds (double* out, long long* t, int iters)
{
    __shared__ double tab[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = i;
    __syncthreads();
    double s0 = 0, s1 = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const double2* p = (const double2*)(tab + ((it*32) & 2047));
#pragma unroll
        for (int i = 0; i < 16; ++i) { double2 v = p[i]; s0 += v.x; s1 += v.y; }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x*blockDim.x + threadIdx.x] = s0 + s1;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
int main ()
{
    double* out; long long* t; long long h;
    hipMalloc(&out, 1 << 24); hipMalloc(&t, 64);
    for (int threads : {64, 256, 512, 1024}) {
        k_fma<<<1, threads>>>(out, t, 1000, 0); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        printf("fma64: %4d threads (1 block): %.2f ticks per wave-FMA (16 independent chains)\n", threads, (double)h/(1000*16));
    }
    for (int threads : {64, 256, 512}) {
        k_lds<<<1, threads>>>(out, t, 1000); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        printf("lds b128 broadcast + 2 adds: %4d threads: %.2f ticks per read\n", threads, (double)h/(1000*16));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k_fma<<<1, 64>>>(out, t, 100000, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("tick rate: %.1f MHz\n", h/(ms*1e3));
    return 0;
}
