// launch_floor.hip -- what one dependent launch costs on a stream (MI355X): empty kernels of several grid sizes back to back,
// a kernel that first reads a few words (the multigrid's gate) and returns, and a small load->store kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_empty () {}
__global__ void k_gate (const unsigned long long* n, int* out) { unsigned long long m = 0; for (int q = 0; q < 48; ++q) m = n[q] > m ? n[q] : m; if (m == 12345ULL) *out = 1; }
__global__ void k_copy (const double* a, double* b, int n) { int i = blockIdx.x*blockDim.x + threadIdx.x; if (i < n) b[i] = a[i] + 1.0; }
template <class F> double timeit (hipStream_t st, int reps, F f)
{
    f(); hipStreamSynchronize(st);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < reps; ++r) f();
    hipStreamSynchronize(st);
    return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count()/reps;
}
int main ()
{
    hipStream_t st; hipStreamCreate(&st);
    unsigned long long* n; int* o; double *a, *b;
    hipMalloc(&n, 4096); hipMemset(n, 0, 4096); hipMalloc(&o, 4); hipMalloc(&a, 8<<20); hipMalloc(&b, 8<<20); hipMemset(a, 0, 8<<20);
    const int R = 2000;
    for (int g : {1, 64, 256, 1024, 4096})
        printf("empty grid %5d x 256: %.2f us per launch\n", g, timeit(st, R, [&]{ hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, st); }));
    printf("empty 1 x 1024: %.2f us\n", timeit(st, R, [&]{ hipLaunchKernelGGL(k_empty, dim3(1), dim3(1024), 0, st); }));
    for (int g : {1, 256, 1024})
        printf("gate grid %5d x 256: %.2f us per launch\n", g, timeit(st, R, [&]{ hipLaunchKernelGGL(k_gate, dim3(g), dim3(256), 0, st, n, o); }));
    for (int nn : {1<<14, 1<<17, 1<<20})
        printf("copy %8d doubles: %.2f us per launch\n", nn, timeit(st, R, [&]{ hipLaunchKernelGGL(k_copy, dim3(nn/256), dim3(256), 0, st, a, b, nn); }));
    // alternating two kernels (different code objects entries)
    printf("alternate empty/copy16k: %.2f us per pair\n", timeit(st, R, [&]{ hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st); hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, st, a, b, 1<<14); }));
    // graph of 16 dependent copies
    hipGraph_t gr; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int k = 0; k < 16; ++k) hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, st, a, b, 1<<14);
    hipStreamEndCapture(st, &gr); hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0);
    printf("graph of 16 copy16k: %.2f us per kernel\n", timeit(st, 500, [&]{ hipGraphLaunch(ge, st); })/16);
    return 0;
}
