// layout + rate probe for v_mfma_f64_16x16x4_f64 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k_layout (const double* A /*16x4 row-major*/, const double* B /*4x16*/, double* D /*16x16*/, int* dmap)
{
    const int l = threadIdx.x;
    // hypothesis: a = A[l%16][l/16], b = B[l/16][l%16]
    const double a = A[(l % 16)*4 + l/16], b = B[(l/16)*16 + l % 16];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { D[l*4 + r] = c[r]; }
    (void)dmap;
}
__global__ void k_rate (double* out, int iters)
{
    const int l = threadIdx.x & 63;
    double a = 1.0 + l*1e-9, b = 1.0 - l*1e-9;
    d4 c0 = {0,0,0,0}, c1 = {0,0,0,0}, c2 = {0,0,0,0}, c3 = {0,0,0,0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x*blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main ()
{
    std::vector<double> A(64), B(64), D(256);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i*4 + k] = (i + 1)*100.0 + (k + 1);      // A[i][k]
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k*16 + j] = (k == 0) ? (j + 1)*1.0 : 0.0;  // only k=0 row: D[i][j] = A[i][0]*(j+1)
    double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, nullptr);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    // decode: value = A[i][0]*(j+1) = ((i+1)*100+1)*(j+1)
    for (int l : {0, 1, 15, 16, 17, 32, 48, 63}) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; ++r) {
            const double v = D[l*4 + r]; int fi = -1, fj = -1;
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (v == ((i + 1)*100.0 + 1)*(j + 1)) { fi = i; fj = j; }
            printf("  r%d -> D[%d][%d]", r, fi, fj);
        }
        printf("\n");
    }
    // rate
    double* o; hipMalloc(&o, 256*1024*8*4);
    const int iters = 2000, wg = 256*8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate, dim3(wg), dim3(256), 0, 0, o, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate, dim3(wg), dim3(256), 0, 0, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wg*4 /*waves*/ * iters*4.0 * 16*16*4*2;
    printf("mfma f64 16x16x4: %.1f TFLOP/s (%.3f ms)\n", flops/ms*1e-9, ms);
    return 0;
}
