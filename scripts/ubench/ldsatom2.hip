// micro-benchmark: ds_add_f64 rate per CU on gfx950 against the address pattern of a wave
//   0: lane l -> word l (conflict free)          1: groups of 2 lanes on one word     2: groups of 4 lanes on one word
//   3: a random word per lane                    4: groups of 4 on one word, groups at random words
//   5: lane l -> word l + small random jitter (0..3)   6: groups of 4, consecutive groups 1 word apart but rows of 16 cells at pitch 20
//   7: lane l -> own word, rows of 16 at pitch 20
//   8: the wave's cells are a 16 x 4 block of a tile (lane l -> cell (l & 15, l >> 4)), stencil base = cell + jitter (0..1 in x and in y)
//   9: cells spread: 16 contiguous lanes hold cells 4 apart in x and y (lane l -> cell (4 (l & 3) + ((l >> 4) & 3), 4 ((l >> 2) & 3))), same jitter
//  10: as 9 without jitter      11: as 8 without jitter
//  12: as 8, but the stencil base is the cell (no jitter) and the stencil 4 x 4 points (16 atomics per particle instead of 9);
//      one lane in eight sits one cell off its place (a particle that has moved since the sort)     13: as 12, one lane in three
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
    else if (MODE == 7) { const int c = l; base = (c >> 4)*20 + (c & 15); }
    else {
        const unsigned h = hash(l*7919u + blockIdx.x);
        const int jx = (MODE == 8 || MODE == 9) ? (h & 1) : 0, jy = (MODE == 8 || MODE == 9) ? ((h >> 1) & 1) : 0;
        int cx, cy;
        if (MODE == 8 || MODE == 11 || MODE >= 12) { cx = l & 15; cy = l >> 4; }
        else { cx = 4*(l & 3) + ((l >> 4) & 3); cy = 4*((l >> 2) & 3); }
        if (MODE == 12 && (h & 0x70) == 0) { cx += ((h >> 8) & 1) ? 1 : -1; }
        if (MODE == 13 && ((h >> 4) % 3) == 0) { if ((h >> 8) & 1) cx += ((h >> 9) & 1) ? 1 : -1; else cy += ((h >> 9) & 1) ? 1 : -1; }
        base = (cy + jy + 1)*20 + cx + jx + 1;
    }
    double v = threadIdx.x*1e-3;
    for (int it = 0; it < iters; ++it) {
        constexpr int NP = (MODE >= 12) ? 16 : 9, W = (MODE >= 12) ? 4 : 3;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int idx = (base + (u/W)*20 + (u%W) + it*61) & 8191;
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
    double ops = (MODE >= 12 ? 16.0 : 9.0)*4000*threads*blocks;
    double per_cu_clk = ops/256/(ms*1e-3*2.4e9);
    printf("%-48s %.3f ms  %.1f clk per wave-instr per CU  %.0f clk per particle round\n", name, ms, 64.0/per_cu_clk, 64.0/per_cu_clk*(MODE >= 12 ? 16 : 9));
}
int main ()
{
    double* out; (void)hipMalloc(&out, 1 << 24);
    run<0>("0 lane -> own word", out); run<1>("1 pairs of lanes on a word", out); run<2>("2 quads of lanes on a word", out);
    run<3>("3 random word per lane", out); run<4>("4 quads on a word, quads at random", out); run<5>("5 own word + jitter 0..3", out);
    run<6>("6 quads, tile rows at pitch 20", out); run<7>("7 own word, tile rows at pitch 20", out);
    run<8>("8 16x4 block of cells + jitter", out); run<9>("9 cells spread 4 apart per 16 lanes + jitter", out);
    run<10>("10 spread, no jitter", out); run<11>("11 16x4 block, no jitter", out);
    run<12>("12 16x4 block, base = cell, 4x4 points, 1/8 off", out); run<13>("13 ... 1/3 of the lanes off by a cell", out);
    return 0;
}
