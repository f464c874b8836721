// micro-benchmark: ds_add_f64 rate per CU on gfx950 against the address pattern of a wave
//   0: lane l -> word l (conflict free)          1: groups of 2 lanes on one word     2: groups of 4 lanes on one word
//   3: a random word per lane                    4: groups of 4 on one word, groups at random words
//   5: lane l -> word l + small random jitter (0..3)   6: groups of 4, consecutive groups 1 word apart but rows of 16 cells at pitch 20
//   7: lane l -> own word, rows of 16 at pitch 20
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) double lds_double;
__device__ inline unsigned hash (unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int MODE>
__global__ void k (double* out, int iters)
{
    __shared__ double acc[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) acc[i] = 0;
    __syncthreads();
    lds_double* a = (lds_double*)acc;
    const int l = threadIdx.x;
    int base;
    if (MODE == 0) base = l;
    else if (MODE == 1) base = l >> 1;
    else if (MODE == 2) base = l >> 2;
    else if (MODE == 3) base = hash(l*7919u + blockIdx.x) & 4095;
    else if (MODE == 4) base = hash((l >> 2)*7919u + blockIdx.x) & 4095;
    else if (MODE == 5) base = l + (hash(l*7919u + blockIdx.x) & 3);
    else if (MODE == 6) { const int c = l >> 2; base = (c >> 4)*20 + (c & 15); }
    else { const int c = l; base = (c >> 4)*20 + (c & 15); }
    double v = threadIdx.x*1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int idx = (base + (u/3)*20 + (u%3) + it*61) & 8191;
            __hip_atomic_fetch_add(a + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    out[blockIdx.x*blockDim.x + threadIdx.x] = acc[threadIdx.x];
}
template <int MODE> void run (const char* name, double* out)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int threads = 256, blocks = 768;
    k<MODE><<<blocks, threads>>>(out, 10);
    (void)hipEventRecord(e0); k<MODE><<<blocks, threads>>>(out, 4000); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double ops = 9.0*4000*threads*blocks;
    double per_cu_clk = ops/256/(ms*1e-3*2.4e9);
    printf("%-44s %.3f ms  %.1f clk per wave-instr per CU\n", name, ms, 64.0/per_cu_clk);
}
int main ()
{
    double* out; (void)hipMalloc(&out, 1 << 24);
    run<0>("0 lane -> own word", out); run<1>("1 pairs of lanes on a word", out); run<2>("2 quads of lanes on a word", out);
    run<3>("3 random word per lane", out); run<4>("4 quads on a word, quads at random", out); run<5>("5 own word + jitter 0..3", out);
    run<6>("6 quads, tile rows at pitch 20", out); run<7>("7 own word, tile rows at pitch 20", out);
    return 0;
}
