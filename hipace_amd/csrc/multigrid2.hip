// multigrid2.hip -- geometric multigrid for the complex Helmholtz-type equation of the laser envelope,
//     -(ar + i ai)(phi_r + i phi_i) + Lap(phi_r + i phi_i) = rhs_r + i rhs_i,     homogeneous Dirichlet walls,
// ar an array, ai a scalar: hpmg::MultiGrid system type 2 (mg_solver/HpMultiGrid.cpp: gs2 :296-334, residual2r/2i
// :192-208, solve2 :1239-1262), called by MultiLaser::AdvanceSliceMG (laser/MultiLaser.cpp:430-608).
//
// Same V-cycle as system type 1 (multigrid.hip): 4 red-black Gauss-Seidel sweeps per level (colour (i+j+s)%2), residual
// behind the sweeps, 4-average restriction, piecewise-constant prolongation, max(16, n) sweeps on the coarsest level,
// cell-centred wall stencil (4/3, -2), stop rule of solve_doit (:1307-1427).  Cell-centred boxes only (even nx, ny; what
// the laser grid of the slice engine is).  The real and the imaginary part are coupled through ai, so a point update
// solves the 2x2 system at once.
//
// This solver is off the headline path (one envelope solve per slice next to 5 field solves): levels with <= 4096 cells run
// whole sweeps sequences in one 1024-thread workgroup (colour after colour behind __syncthreads, arrays in L2), larger
// levels take one launch per colour; the planes are planar [2][ny][nx] without guard cells.
#include "common.h"

#include <vector>
#include <algorithm>
#include <cmath>
#include <cstring>

namespace hps {

struct Lev2 { int nx, ny; long n; double facx, facy; };

__device__ __forceinline__ void gs2_point (const Lev2& l, int i, int j, double* __restrict__ phi, const double* __restrict__ rhs,
                                           const double* __restrict__ acf)
{
    const long o = (long)j*l.nx + i;
    double* pr = phi + o; double* pi = phi + l.n + o;
    double lap0, lap1;
    double c0 = -2.0*(l.facx + l.facy);
    if (i == 0)             { lap0 = l.facx*(4./3.)*pr[1];  lap1 = l.facx*(4./3.)*pi[1];  c0 -= 2.0*l.facx; }
    else if (i == l.nx - 1) { lap0 = l.facx*(4./3.)*pr[-1]; lap1 = l.facx*(4./3.)*pi[-1]; c0 -= 2.0*l.facx; }
    else                    { lap0 = l.facx*(pr[-1] + pr[1]); lap1 = l.facx*(pi[-1] + pi[1]); }
    if (j == 0)             { lap0 += l.facy*(4./3.)*pr[l.nx];  lap1 += l.facy*(4./3.)*pi[l.nx];  c0 -= 2.0*l.facy; }
    else if (j == l.ny - 1) { lap0 += l.facy*(4./3.)*pr[-l.nx]; lap1 += l.facy*(4./3.)*pi[-l.nx]; c0 -= 2.0*l.facy; }
    else                    { lap0 += l.facy*(pr[-l.nx] + pr[l.nx]); lap1 += l.facy*(pi[-l.nx] + pi[l.nx]); }
    double cr = c0 - acf[o], ci = -acf[l.n + o];
    const double cmag = 1.0/(cr*cr + ci*ci);
    cr *= cmag; ci *= cmag;
    const double rr = rhs[o] - lap0, ri = rhs[l.n + o] - lap1;
    *pr = rr*cr + ri*ci;
    *pi = ri*cr - rr*ci;
}

// laplacian (:162-182) of component n at (i, j)
__device__ __forceinline__ double lap2 (const Lev2& l, int i, int j, const double* __restrict__ p)
{
    double lap = -2.0*(l.facx + l.facy)*p[0];
    if (i == 0)             lap += l.facx*((4./3.)*p[1] - 2.0*p[0]);
    else if (i == l.nx - 1) lap += l.facx*((4./3.)*p[-1] - 2.0*p[0]);
    else                    lap += l.facx*(p[-1] + p[1]);
    if (j == 0)             lap += l.facy*((4./3.)*p[l.nx] - 2.0*p[0]);
    else if (j == l.ny - 1) lap += l.facy*((4./3.)*p[-l.nx] - 2.0*p[0]);
    else                    lap += l.facy*(p[-l.nx] + p[l.nx]);
    return lap;
}

// `ncolors` sweeps starting with colour `color0`, then (do_res) res = rhs - L(phi) and its max-norm.  More than one sweep or
// sweeps + residual in one launch need a single workgroup.
__global__ __launch_bounds__(1024)
void k2_sweeps (Lev2 l, double* phi, const double* __restrict__ rhs, const double* __restrict__ acf, int color0, int ncolors,
                int do_res, double* __restrict__ res, unsigned long long* norm)
{
    const long stride = (long)gridDim.x*blockDim.x;
    for (int c = 0; c < ncolors; ++c) {
        for (long o = (long)blockIdx.x*blockDim.x + threadIdx.x; o < l.n; o += stride) {
            const int j = (int)(o / l.nx), i = (int)(o - (long)j*l.nx);
            if (((i + j + color0 + c) & 1) == 0) gs2_point(l, i, j, phi, rhs, acf);
        }
        __syncthreads();
    }
    if (!do_res) return;
    double m = 0.0;
    for (long o = (long)blockIdx.x*blockDim.x + threadIdx.x; o < l.n; o += stride) {
        const int j = (int)(o / l.nx), i = (int)(o - (long)j*l.nx);
        const double pr = phi[o], pi = phi[l.n + o], ar = acf[o], ai = acf[l.n + o];
        const double r0 = rhs[o] - lap2(l, i, j, phi + o) + (ar*pr - ai*pi);              // residual2r (:192-199)
        const double r1 = rhs[l.n + o] - lap2(l, i, j, phi + l.n + o) + (ai*pr + ar*pi);  // residual2i (:201-208)
        res[o] = r0; res[l.n + o] = r1;
        m = fmax(m, fmax(fabs(r0), fabs(r1)));
    }
    if (norm) {
        for (int s = 32; s > 0; s >>= 1) m = fmax(m, __shfl_xor(m, s));
        if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(norm, (unsigned long long)__double_as_longlong(m));
    }
}

// crse = mean of the 4 fine cells (restrict_cc :29-37), ncomp planes
__global__ __launch_bounds__(256)
void k2_restrict (Lev2 c, Lev2 f, double* __restrict__ crse, const double* __restrict__ fine, int ncomp)
{
    const long o = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (o >= c.n*ncomp) return;
    const int n = (int)(o / c.n); const long q = o - (long)n*c.n;
    const int j = (int)(q / c.nx), i = (int)(q - (long)j*c.nx);
    const double* p = fine + (long)n*f.n + (long)(2*j)*f.nx + 2*i;
    crse[o] = 0.25*(p[0] + p[1] + p[f.nx] + p[f.nx + 1]);
}

// fine = fin + crse(i/2, j/2) (interpcpy_cc :88-95), 2 planes
__global__ __launch_bounds__(256)
void k2_interp_add (Lev2 f, Lev2 c, double* fout, const double* fin, const double* __restrict__ crse)      // fout may be fin
{
    const long o = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (o >= 2*f.n) return;
    const int n = (int)(o / f.n); const long q = o - (long)n*f.n;
    const int j = (int)(q / f.nx), i = (int)(q - (long)j*f.nx);
    fout[o] = fin[o] + crse[(long)n*c.n + (long)(j/2)*c.nx + i/2];
}

__global__ __launch_bounds__(256)
void k2_fill_acf (Lev2 l, double* __restrict__ acf, const double* __restrict__ ar, const double* __restrict__ ai_scalar)
{
    const long o = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (o >= l.n) return;
    acf[o] = ar[o]; acf[l.n + o] = *ai_scalar;
}

__global__ __launch_bounds__(256)
void k2_maxabs (const double* __restrict__ p, long n, unsigned long long* norm)
{
    double m = 0.0;
    for (long o = (long)blockIdx.x*blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x*blockDim.x) m = fmax(m, fabs(p[o]));
    for (int s = 32; s > 0; s >>= 1) m = fmax(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(norm, (unsigned long long)__double_as_longlong(m));
}

struct Multigrid2 {
    int nx = 0, ny = 0; double dx = 0, dy = 0;
    std::vector<Lev2> L;
    std::vector<double*> acf, res, cor, rescor;      // per level, 2 planes each
    unsigned long long* d_norm = nullptr;            // [2]: residual, rhs
    long total_vcycles = 0;
    ~Multigrid2 () {
        for (auto& v : {acf, res, cor, rescor}) for (double* p : v) (void)hipFree(p);
        (void)hipFree(d_norm);
    }
};

static const long SINGLE_BLOCK_CELLS = 4096;

// gsrb_4_residual (:742-848) in place on phi
static void sweeps4 (const Lev2& l, double* phi, const double* rhs, const double* acf, bool do_res, double* res,
                     unsigned long long* norm, hipStream_t st)
{
    if (l.n <= SINGLE_BLOCK_CELLS) {
        hipLaunchKernelGGL(k2_sweeps, dim3(1), dim3(1024), 0, st, l, phi, rhs, acf, 0, 4, do_res ? 1 : 0, res, norm);
        return;
    }
    const dim3 grid((unsigned)std::min<long>(ceil_div(l.n, 1024), 1024)), block(1024);
    for (int c = 0; c < 4; ++c) hipLaunchKernelGGL(k2_sweeps, grid, block, 0, st, l, phi, rhs, acf, c, 1, 0, res, norm);
    if (do_res) hipLaunchKernelGGL(k2_sweeps, grid, block, 0, st, l, phi, rhs, acf, 0, 0, 1, res, norm);
}

static int mg2_create (int nx, int ny, double dx, double dy, Multigrid2** out)
{
    Multigrid2* M = new Multigrid2;
    M->nx = nx; M->ny = ny; M->dx = dx; M->dy = dy;
    int lx = nx, ly = ny;
    for (int il = 0; il < 31; ++il) {           // level build (:1043-1072), cell-centred
        const double fx = dx*(double)(1 << il), fy = dy*(double)(1 << il);
        M->L.push_back(Lev2{lx, ly, (long)lx*ly, 1.0/(fx*fx), 1.0/(fy*fy)});
        const bool ok = lx >= 4 && ly >= 4 && lx % 2 == 0 && ly % 2 == 0;
        if (!ok) break;
        lx /= 2; ly /= 2;
    }
    for (const Lev2& l : M->L) {
        double* p[4];
        for (int k = 0; k < 4; ++k) { HPS_HIP_CHECK(hipMalloc(&p[k], (size_t)2*l.n*sizeof(double))); HPS_HIP_CHECK(hipMemset(p[k], 0, (size_t)2*l.n*sizeof(double))); }
        M->acf.push_back(p[0]); M->res.push_back(p[1]); M->cor.push_back(p[2]); M->rescor.push_back(p[3]);
    }
    HPS_HIP_CHECK(hipMalloc(&M->d_norm, 2*sizeof(unsigned long long)));
    *out = M;
    return HPS_OK;
}

static int read_norm (Multigrid2* M, int which, double* v, hipStream_t st)
{
    unsigned long long bits = 0;
    HPS_HIP_CHECK(hipMemcpyAsync(&bits, M->d_norm + which, sizeof(bits), hipMemcpyDeviceToHost, st));
    HPS_HIP_CHECK(hipStreamSynchronize(st));
    long long b = (long long)bits; double d;
    static_assert(sizeof(d) == sizeof(b), "");
    std::memcpy(&d, &b, sizeof(d));
    *v = d;
    return HPS_OK;
}

// vcycle (:1429-1512): on entry rescor[0] holds the level-0 residual, on exit again (of the improved solution in sol)
static void vcycle2 (Multigrid2* M, double* sol, const double* rhs, hipStream_t st)
{
    const int maxl = (int)M->L.size() - 1;
    for (int il = 0; il < maxl; ++il) {
        const Lev2& l = M->L[il];
        if (il > 0) {
            (void)hipMemsetAsync(M->cor[il], 0, (size_t)2*l.n*sizeof(double), st);
            sweeps4(l, M->cor[il], M->res[il], M->acf[il], true, M->rescor[il], nullptr, st);
        }
        const Lev2& c = M->L[il + 1];
        hipLaunchKernelGGL(k2_restrict, dim3(ceil_div(2*c.n, 256)), dim3(256), 0, st, c, l, M->res[il + 1], M->rescor[il], 2);
    }
    {   // bottomsolve, CPU branch (:1583-1593)
        const Lev2& l = M->L[maxl];
        (void)hipMemsetAsync(M->cor[maxl], 0, (size_t)2*l.n*sizeof(double), st);
        const int numsweeps = std::max(16, (std::max(l.nx, l.ny) + 1)/2*2);
        if (l.n <= SINGLE_BLOCK_CELLS)
            hipLaunchKernelGGL(k2_sweeps, dim3(1), dim3(1024), 0, st, l, M->cor[maxl], M->res[maxl], M->acf[maxl], 0, numsweeps, 0, (double*)nullptr, (unsigned long long*)nullptr);
        else {
            const dim3 grid((unsigned)std::min<long>(ceil_div(l.n, 1024), 1024));
            for (int s = 0; s < numsweeps; ++s)
                hipLaunchKernelGGL(k2_sweeps, grid, dim3(1024), 0, st, l, M->cor[maxl], M->res[maxl], M->acf[maxl], s, 1, 0, (double*)nullptr, (unsigned long long*)nullptr);
        }
    }
    for (int il = maxl - 1; il >= 0; --il) {
        const Lev2& l = M->L[il]; const Lev2& c = M->L[il + 1];
        double* phi = (il == 0) ? sol : M->cor[il];
        hipLaunchKernelGGL(k2_interp_add, dim3(ceil_div(2*l.n, 256)), dim3(256), 0, st, l, c, phi, M->cor[il], M->cor[il + 1]);
        sweeps4(l, phi, (il == 0) ? rhs : M->res[il], M->acf[il], false, nullptr, nullptr, st);
    }
    // cor0 = 4 more sweeps of the solution, residual behind them
    const Lev2& l0 = M->L[0];
    (void)hipMemcpyAsync(M->cor[0], sol, (size_t)2*l0.n*sizeof(double), hipMemcpyDeviceToDevice, st);
    (void)hipMemsetAsync(M->d_norm, 0, sizeof(unsigned long long), st);
    sweeps4(l0, M->cor[0], rhs, M->acf[0], true, M->rescor[0], M->d_norm, st);
}

static int mg2_solve2 (Multigrid2* M, double* sol, const double* rhs, const double* acf_real, const double* acf_imag, double tol_rel,
                       double tol_abs, int maxiter, int* iters_host, double* resnorm_host, hipStream_t st)
{
    const Lev2& l0 = M->L[0];
    hipLaunchKernelGGL(k2_fill_acf, dim3(ceil_div(l0.n, 256)), dim3(256), 0, st, l0, M->acf[0], acf_real, acf_imag);
    for (size_t il = 1; il < M->L.size(); ++il)      // average_down_acoef (:1640-1700)
        hipLaunchKernelGGL(k2_restrict, dim3(ceil_div(2*M->L[il].n, 256)), dim3(256), 0, st, M->L[il], M->L[il - 1], M->acf[il], M->acf[il - 1], 2);
    // solve_doit (:1307-1427)
    (void)hipMemcpyAsync(M->cor[0], sol, (size_t)2*l0.n*sizeof(double), hipMemcpyDeviceToDevice, st);
    (void)hipMemsetAsync(M->d_norm, 0, 2*sizeof(unsigned long long), st);
    sweeps4(l0, M->cor[0], rhs, M->acf[0], true, M->rescor[0], M->d_norm, st);
    hipLaunchKernelGGL(k2_maxabs, dim3(64), dim3(256), 0, st, rhs, 2*l0.n, M->d_norm + 1);
    HPS_HIP_CHECK(hipGetLastError());
    double resnorm0 = 0.0, rhsnorm0 = 0.0;
    if (int e = read_norm(M, 0, &resnorm0, st)) return e;
    if (int e = read_norm(M, 1, &rhsnorm0, st)) return e;
    const double max_norm = (rhsnorm0 >= resnorm0) ? rhsnorm0 : resnorm0;
    const double res_target = std::max(tol_abs, std::max(tol_rel, 1.e-16)*max_norm);
    int iters = 0; double norminf = resnorm0;
    if (resnorm0 > res_target) {
        bool converged = false, diverged = false;
        for (int iter = 0; iter < maxiter; ++iter) {
            vcycle2(M, sol, rhs, st);
            HPS_HIP_CHECK(hipGetLastError());
            ++iters;
            if (int e = read_norm(M, 0, &norminf, st)) return e;
            converged = (norminf <= res_target);
            if (converged) break;
            if (!(norminf <= 1.e20*max_norm)) { diverged = true; break; }
        }
        // hpmg aborts here (HpMultiGrid.cpp:1409-1416)
        if (diverged) { set_error("hps_mg2_solve2: diverging"); return HPS_ERR_MG_DIVERGED; }
        if (!converged) { set_error("hps_mg2_solve2: not converged after max_iters V-cycles"); return HPS_ERR_MG_MAXITER; }
    }
    HPS_HIP_CHECK(hipMemcpyAsync(sol, M->cor[0], (size_t)2*l0.n*sizeof(double), hipMemcpyDeviceToDevice, st));
    M->total_vcycles += iters;
    if (iters_host) *iters_host = iters;
    if (resnorm_host) *resnorm_host = norminf;
    return HPS_OK;
}

int mg2_create_internal (int nx, int ny, double dx, double dy, void** h)
{
    Multigrid2* M = nullptr;
    if (int e = mg2_create(nx, ny, dx, dy, &M)) return e;
    *h = M;
    return HPS_OK;
}
int mg2_solve_internal (void* h, double* sol, const double* rhs, const double* ar, const double* ai, double tol_rel, double tol_abs,
                        int maxiter, int* iters, hipStream_t st)
{
    return mg2_solve2(static_cast<Multigrid2*>(h), sol, rhs, ar, ai, tol_rel, tol_abs, maxiter, iters, nullptr, st);
}
void mg2_destroy_internal (void* h) { delete static_cast<Multigrid2*>(h); }

} // namespace hps

using namespace hps;

extern "C" int hps_mg2_create (int nx, int ny, double dx, double dy, void** handle)
{
    HPS_REQUIRE(handle && nx >= 2 && ny >= 2, "hps_mg2_create: bad size");
    HPS_REQUIRE(nx % 2 == 0 && ny % 2 == 0, "hps_mg2_create: cell-centred boxes only (even nx, ny)");
    return mg2_create_internal(nx, ny, dx, dy, handle);
}

extern "C" int hps_mg2_solve2 (void* handle, double* sol2_dev, const double* rhs2_dev, const double* acoef_real_dev,
                               const double* acoef_imag_dev, double tol_rel, double tol_abs, int max_iters, int* iters_host,
                               double* resnorm_host, hps_stream stream)
{
    HPS_REQUIRE(handle && sol2_dev && rhs2_dev && acoef_real_dev && acoef_imag_dev, "hps_mg2_solve2: null argument");
    return mg2_solve2(static_cast<Multigrid2*>(handle), sol2_dev, rhs2_dev, acoef_real_dev, acoef_imag_dev, tol_rel, tol_abs,
                      max_iters, iters_host, resnorm_host, (hipStream_t)stream);
}

extern "C" int hps_mg2_destroy (void* handle)
{
    mg2_destroy_internal(handle);
    return HPS_OK;
}
