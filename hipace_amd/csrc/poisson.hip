// poisson.hip -- transverse Poisson solve Lap(F) = S with homogeneous Dirichlet walls, as a
// 2-D type-I discrete sine transform (DST-I) built on rocFFT batched 1-D complex-to-real FFTs.
//
// Replaces FFTPoissonSolverDirichletFast (fields/fft_poisson_solver/
// FFTPoissonSolverDirichletFast.cpp:195-328); eigenvalues / normalisation follow
// FFTPoissonSolverDirichletDirect.cpp:58-83 (FFTW RODFT00 convention), so CPU goldens apply.
//
// DST-I of length n through one real FFT of length N = n+1 (half the odd-extension length):
// with y the odd 2N-periodic extension of the data (y_m = x_{m-1}, y_0 = y_N = 0) feed the
// Hermitian spectrum
//        Z_p = (y_{2p+1} - y_{2p-1}) + i y_{2p},      p = 0 .. N/2
// to an unnormalised C2R transform r = C2R(Z); then for q = 1..n
//        T_{q-1} = 2 sum_m y_m sin(pi m q / N) = (r_{N-q} - r_q)/2 + (r_q + r_{N-q}) / (4 sin(pi q/N)).
// (Derivation: the even-index samples enter through Im Z, the odd-index samples through the
// first difference in Re Z, which pulls out the factor 2 sin(pi q/N).)
//
// Passes per solve: pre(x) | FFTx | post(x)+transpose+pre(y) | FFTy | post(y)*eig+pre(y) | FFTy |
// post(y)+transpose+pre(x) | FFTx | post(x) -> slab component.
#include "common.h"

#include <rocfft/rocfft.h>
#include <vector>
#include <cmath>

namespace hps {

// value of the odd extension at sample m for a row of T values held behind functor get(j),
// j = m-1 in [0, n)
template <class G>
__device__ __forceinline__ double odd_ext (int m, int N, G&& get)
{
    double sgn = 1.0;
    if (m < 0) { m = -m; sgn = -1.0; }
    if (m > N) { m = 2*N - m; sgn = -sgn; }
    if (m == 0 || m == N) return 0.0;
    return sgn*get(m - 1);
}

// post-processing of one C2R output row: r has N = n+1 entries, k in [0, n)
__device__ __forceinline__ double dst_from_r (const double* r, int k, int N, double isin4)
{
    const double a = r[k + 1];
    const double b = r[N - 1 - k];
    return 0.5*(b - a) + (a + b)*isin4;
}

// staging (rows of n reals) -> Z (rows of nh complex)
__global__ __launch_bounds__(256)
void k_pre_rows (const double* __restrict__ src, long src_pitch, double2* __restrict__ z, int n, int nh, int nrows)
{
    const int p = blockIdx.x*blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (p >= nh || row >= nrows) return;
    const double* x = src + (long)row*src_pitch;
    const int N = n + 1;
    auto get = [&] (int j) { return x[j]; };
    const double re = odd_ext(2*p + 1, N, get) - odd_ext(2*p - 1, N, get);
    const double im = odd_ext(2*p, N, get);
    z[(long)row*nh + p] = make_double2(re, im);
}

// r (rows of N reals, one row per batch entry) -> T^transposed -> Z of the other direction.
// in : nrows_in rows (index j), each with n_in DST outputs (index k)
// out: n_in rows (index k), each nh_out complex, transform length N_out = nrows_in + 1
constexpr int TP_K = 32;      // k-extent of a tile
constexpr int TP_P = 16;      // p-extent of a tile -> needs 2*TP_P + 2 values of j
constexpr int TP_J = 2*TP_P + 2;

__global__ __launch_bounds__(256)
void k_post_transpose_pre (const double* __restrict__ r, int n_in, int nrows_in,
                           const double* __restrict__ isin4, double2* __restrict__ z, int nh_out)
{
    __shared__ double tile[TP_J][TP_K + 1];
    const int N_in = n_in + 1;
    const int N_out = nrows_in + 1;
    const int k0 = blockIdx.x*TP_K;
    const int p0 = blockIdx.y*TP_P;
    const int j0 = 2*p0 - 2;

    {   // load + post-process: lanes run along k (contiguous in r)
        const int kk = threadIdx.x % TP_K;
        const int jr = threadIdx.x / TP_K;          // 0..7
        const int k = k0 + kk;
        for (int jj = jr; jj < TP_J; jj += 256/TP_K) {
            const int j = j0 + jj;
            double v = 0.0;
            if (k < n_in && j >= 0 && j < nrows_in) v = dst_from_r(r + (long)j*N_in, k, N_in, isin4[k]);
            tile[jj][kk] = v;
        }
    }
    __syncthreads();
    {   // build Z: lanes run along p (contiguous in z)
        const int pp = threadIdx.x % TP_P;
        const int kr = threadIdx.x / TP_P;          // 0..15
        const int p = p0 + pp;
        for (int kk = kr; kk < TP_K; kk += 256/TP_P) {
            const int k = k0 + kk;
            if (k < n_in && p < nh_out) {
                auto get = [&] (int j) { return tile[j - j0][kk]; };
                const double re = odd_ext(2*p + 1, N_out, get) - odd_ext(2*p - 1, N_out, get);
                const double im = odd_ext(2*p, N_out, get);
                z[(long)k*nh_out + p] = make_double2(re, im);
            }
        }
    }
}

// same orientation: T = post(r) * eig, then pre for the inverse transform along the same axis
__global__ __launch_bounds__(256)
void k_post_mult_pre (const double* __restrict__ r, int n, int nrows, const double* __restrict__ isin4,
                      const double* __restrict__ eig, double2* __restrict__ z, int nh)
{
    const int p = blockIdx.x*blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (p >= nh || row >= nrows) return;
    const int N = n + 1;
    const double* rr = r + (long)row*N;
    const double* ee = eig + (long)row*n;
    auto get = [&] (int l) { return ee[l]*dst_from_r(rr, l, N, isin4[l]); };
    const double re = odd_ext(2*p + 1, N, get) - odd_ext(2*p - 1, N, get);
    const double im = odd_ext(2*p, N, get);
    z[(long)row*nh + p] = make_double2(re, im);
}

// final post-processing straight into the slab component
__global__ __launch_bounds__(256)
void k_post_to_slab (const double* __restrict__ r, int n, int nrows, const double* __restrict__ isin4,
                     double* __restrict__ dst, long dst_pitch)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (k >= n || row >= nrows) return;
    dst[(long)row*dst_pitch + k] = dst_from_r(r + (long)row*(n + 1), k, n + 1, isin4[k]);
}

struct Poisson {
    int nx = 0, ny = 0;
    rocfft_plan plan_x = nullptr, plan_y = nullptr;
    rocfft_execution_info info = nullptr;
    void* work = nullptr; size_t work_bytes = 0;
    double2* zbuf = nullptr; double* rbuf = nullptr;
    double* eig = nullptr; double* isin_x = nullptr; double* isin_y = nullptr;
    hipStream_t bound_stream = nullptr; bool stream_bound = false;

    ~Poisson () {
        if (plan_x) rocfft_plan_destroy(plan_x);
        if (plan_y) rocfft_plan_destroy(plan_y);
        if (info) rocfft_execution_info_destroy(info);
        (void)hipFree(work); (void)hipFree(zbuf); (void)hipFree(rbuf); (void)hipFree(eig);
        (void)hipFree(isin_x); (void)hipFree(isin_y);
    }
};

static bool g_rocfft_setup = false;

static int make_plan (rocfft_plan* plan, int N, int batch)
{
    const size_t len[1] = {(size_t)N};
    if (rocfft_plan_create(plan, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                           rocfft_precision_double, 1, len, (size_t)batch, nullptr) != rocfft_status_success) {
        set_error("rocfft_plan_create failed for length " + std::to_string(N));
        return HPS_ERR_FFT;
    }
    return HPS_OK;
}

int poisson_create (int nx, int ny, double dx, double dy, Poisson** out)
{
    if (!g_rocfft_setup) { rocfft_setup(); g_rocfft_setup = true; }
    Poisson* P = new Poisson;
    P->nx = nx; P->ny = ny;
    const int Nx = nx + 1, Ny = ny + 1;
    const int nhx = Nx/2 + 1, nhy = Ny/2 + 1;
    int e;
    if ((e = make_plan(&P->plan_x, Nx, ny)) || (e = make_plan(&P->plan_y, Ny, nx))) { delete P; return e; }
    size_t wx = 0, wy = 0;
    rocfft_plan_get_work_buffer_size(P->plan_x, &wx);
    rocfft_plan_get_work_buffer_size(P->plan_y, &wy);
    P->work_bytes = std::max(wx, wy);
    rocfft_execution_info_create(&P->info);
    if (P->work_bytes) {
        HPS_HIP_CHECK(hipMalloc(&P->work, P->work_bytes));
        rocfft_execution_info_set_work_buffer(P->info, P->work, P->work_bytes);
    }
    const size_t zc = std::max((size_t)nhx*ny, (size_t)nhy*nx);
    const size_t rc = std::max((size_t)Nx*ny, (size_t)Ny*nx);
    HPS_HIP_CHECK(hipMalloc(&P->zbuf, zc*sizeof(double2)));
    HPS_HIP_CHECK(hipMalloc(&P->rbuf, rc*sizeof(double)));
    HPS_HIP_CHECK(hipMalloc(&P->eig, (size_t)nx*ny*sizeof(double)));
    HPS_HIP_CHECK(hipMalloc(&P->isin_x, nx*sizeof(double)));
    HPS_HIP_CHECK(hipMalloc(&P->isin_y, ny*sizeof(double)));

    // spectral operator in transposed (x-frequency major, y-frequency contiguous) layout
    std::vector<double> h_eig((size_t)nx*ny), hx(nx), hy(ny);
    const double pi = 3.14159265358979323846;
    const double sxf = pi/(2.*(nx + 1)), syf = pi/(2.*(ny + 1));
    const double norm_fac = 0.5/(2*((double)(nx + 1)*(ny + 1)));
    for (int k = 0; k < nx; ++k) {
        const double sxq = std::sin((k + 1)*sxf)*std::sin((k + 1)*sxf);
        for (int l = 0; l < ny; ++l) {
            const double syq = std::sin((l + 1)*syf)*std::sin((l + 1)*syf);
            h_eig[(size_t)k*ny + l] = (sxq != 0 && syq != 0) ? norm_fac/(-4.0*(sxq/(dx*dx) + syq/(dy*dy))) : 0.0;
        }
    }
    for (int k = 0; k < nx; ++k) hx[k] = 1.0/(4.0*std::sin(pi*(k + 1.0)/(nx + 1.0)));
    for (int l = 0; l < ny; ++l) hy[l] = 1.0/(4.0*std::sin(pi*(l + 1.0)/(ny + 1.0)));
    HPS_HIP_CHECK(hipMemcpy(P->eig, h_eig.data(), h_eig.size()*sizeof(double), hipMemcpyHostToDevice));
    HPS_HIP_CHECK(hipMemcpy(P->isin_x, hx.data(), nx*sizeof(double), hipMemcpyHostToDevice));
    HPS_HIP_CHECK(hipMemcpy(P->isin_y, hy.data(), ny*sizeof(double), hipMemcpyHostToDevice));
    *out = P;
    return HPS_OK;
}

static int run_fft (Poisson* P, rocfft_plan plan)
{
    void* in[1] = {P->zbuf};
    void* outp[1] = {P->rbuf};
    if (rocfft_execute(plan, in, outp, P->info) != rocfft_status_success) {
        set_error("rocfft_execute failed");
        return HPS_ERR_FFT;
    }
    return HPS_OK;
}

// src: nx*ny source with row pitch src_pitch; dst: pointer to cell (0,0) of the target plane
int poisson_solve (Poisson* P, const double* src, long src_pitch, double* dst, long dst_pitch, hipStream_t st)
{
    const int nx = P->nx, ny = P->ny;
    const int Nx = nx + 1, Ny = ny + 1;
    const int nhx = Nx/2 + 1, nhy = Ny/2 + 1;
    if (!P->stream_bound || P->bound_stream != st) {
        rocfft_execution_info_set_stream(P->info, st);
        P->bound_stream = st; P->stream_bound = true;
    }
    int e;
    hipLaunchKernelGGL(k_pre_rows, dim3(ceil_div(nhx, 256), ny), dim3(256), 0, st, src, src_pitch, P->zbuf, nx, nhx, ny);
    if ((e = run_fft(P, P->plan_x))) return e;
    hipLaunchKernelGGL(k_post_transpose_pre, dim3(ceil_div(nx, TP_K), ceil_div(nhy, TP_P)), dim3(256), 0, st,
                       P->rbuf, nx, ny, P->isin_x, P->zbuf, nhy);
    if ((e = run_fft(P, P->plan_y))) return e;
    hipLaunchKernelGGL(k_post_mult_pre, dim3(ceil_div(nhy, 256), nx), dim3(256), 0, st,
                       P->rbuf, ny, nx, P->isin_y, P->eig, P->zbuf, nhy);
    if ((e = run_fft(P, P->plan_y))) return e;
    hipLaunchKernelGGL(k_post_transpose_pre, dim3(ceil_div(ny, TP_K), ceil_div(nhx, TP_P)), dim3(256), 0, st,
                       P->rbuf, ny, nx, P->isin_y, P->zbuf, nhx);
    if ((e = run_fft(P, P->plan_x))) return e;
    hipLaunchKernelGGL(k_post_to_slab, dim3(ceil_div(nx, 256), ny), dim3(256), 0, st,
                       P->rbuf, nx, ny, P->isin_x, dst, dst_pitch);
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

} // namespace hps

using namespace hps;

extern "C" int hps_poisson_create (int nx, int ny, double dx, double dy, void** handle)
{
    HPS_REQUIRE(nx >= 2 && ny >= 2 && handle, "hps_poisson_create: bad size");
    Poisson* P = nullptr;
    if (int e = poisson_create(nx, ny, dx, dy, &P)) return e;
    *handle = P;
    return HPS_OK;
}

extern "C" int hps_poisson_solve (void* handle, const double* staging, hps_slab dst, int dst_comp, hps_stream stream)
{
    HPS_REQUIRE(handle && staging && dst.p, "hps_poisson_solve: null argument");
    Poisson* P = static_cast<Poisson*>(handle);
    HPS_REQUIRE(dst.nx == P->nx && dst.ny == P->ny, "hps_poisson_solve: slab size does not match the solver");
    HPS_REQUIRE(dst_comp >= 0 && dst_comp < dst.ncomp, "hps_poisson_solve: bad component");
    double* d = dst.p + (long)dst_comp*dst.nstride + dst.ng + (long)dst.ng*dst.jstride;
    return poisson_solve(P, staging, P->nx, d, dst.jstride, (hipStream_t)stream);
}

extern "C" int hps_poisson_destroy (void* handle)
{
    delete static_cast<Poisson*>(handle);
    return HPS_OK;
}
