// micro-benchmark: LDS fp64 atomic add / store / read throughput per CU on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) double lds_double;
template <int MODE>
__global__ void k (double* out, int iters, int stride)
{
    __shared__ double acc[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) acc[i] = 0;
    __syncthreads();
    lds_double* a = (lds_double*)acc;
    const int base = (threadIdx.x*stride) & 4095;
    double v = threadIdx.x*1e-3, s = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = (base + u*64 + it) & 8191;
            if (MODE == 0) __hip_atomic_fetch_add(a + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 1) a[idx] = v;
            else s += a[idx];
        }
    }
    __syncthreads();
    out[blockIdx.x*blockDim.x + threadIdx.x] = acc[threadIdx.x] + s;
}
template <int MODE> void run (const char* name, double* out)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int stride : {1, 3}) for (int threads : {256, 1024}) for (int blocks : {256, 512}) {
        k<MODE><<<blocks, threads>>>(out, 10, stride);
        (void)hipEventRecord(e0); k<MODE><<<blocks, threads>>>(out, 2000, stride); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double ops = 16.0*2000*threads*blocks;           // lane-ops
        double per_cu_clk = ops/256/(ms*1e-3*2.4e9);      // lane-ops per CU per clock (at 2.4 GHz)
        printf("%-8s stride %d blocks %4d x %4d thr: %.3f ms  %.1f lane-ops/clk/CU  (%.1f clk per wave-instr per CU)\n", name, stride, blocks, threads, ms, per_cu_clk, 64.0/per_cu_clk);
    }
}
int main ()
{
    double* out; (void)hipMalloc(&out, 1 << 24);
    run<0>("ds_add_f64", out); run<1>("ds_write_b64", out); run<2>("ds_read_b64", out);
    return 0;
}
