// multigrid.hip -- geometric multigrid for  -a*phi + Lap(phi) = rhs  (2 right-hand sides sharing
// one coefficient a: the explicit Bx/By solve), homogeneous Dirichlet walls, on gfx950.
//
// Numerically a restatement of hpmg::MultiGrid system type 1 (mg_solver/HpMultiGrid.cpp):
// V-cycle with 4 red-black Gauss-Seidel sweeps per level (colour (i+j+s)%2, s = 0..3), residual
// fused behind the sweeps, cell-centred 4-average / nodal full-weighting restriction, piecewise
// constant / bilinear prolongation, 16 sweeps on the coarsest level, cell-centred wall stencil
// with the 4/3-2 coefficients (HpMultiGrid.cpp:162-182,265-292), stop rule of solve_doit
// (:1307-1427).  Because Bx/By are only converged to tol_rel = 1e-4 from the previous slice's
// field, parity of the slice engine requires this exact arithmetic.
//
// MI355X mapping: the smoother is one LDS-tiled kernel per level doing all 4 sweeps (+ residual,
// + max-norm) on a 66x34 tile per 256-thread workgroup: phi lives in LDS, the per-cell rhs and
// coefficient stay in registers across sweeps, every cell is read from HBM once and written once.
#include "common.h"

#include <vector>
#include <algorithm>
#include <cmath>
#include <cstring>

namespace hps {

// 2-D multi-component view in level index space: (i,j,n) -> p[(i+oi) + (j+oj)*js + n*ns]
struct FView {
    double* p; long js, ns; int oi, oj;
    __device__ __forceinline__ double& operator() (int i, int j, int n) const { return p[(long)(i + oi) + (long)(j + oj)*js + (long)n*ns]; }
};

struct LevBox { int lox, loy, hix, hiy;      // index bounds of the level box (walls for nodal)
                int vlx, vly, vhx, vhy; };   // unknowns (valid_domain_box)

constexpr int GT_X = 64, GT_Y = 32;          // cells swept per tile
constexpr int GA_X = GT_X + 2, GA_Y = GT_Y + 2;
constexpr int GPAIRS = GT_X*GT_Y/2/256;      // cell pairs per thread = 4

template <bool CC>
__device__ __forceinline__ void gs_update (double* phi /* LDS plane */, int li, int lj, int i, int j,
                                           const LevBox& b, double rhs, double acf, double facx, double facy)
{
    // li, lj: indices into the LDS array (ring included)
    double lap;
    double c0 = -(acf + 2.0*(facx + facy));
    const double* c = phi + lj*GA_X + li;
    if (CC && i == b.lox)      { lap = facx*(4./3.)*c[1];  c0 -= 2.0*facx; }
    else if (CC && i == b.hix) { lap = facx*(4./3.)*c[-1]; c0 -= 2.0*facx; }
    else                       { lap = facx*(c[-1] + c[1]); }
    if (CC && j == b.loy)      { lap += facy*(4./3.)*c[GA_X];  c0 -= 2.0*facy; }
    else if (CC && j == b.hiy) { lap += facy*(4./3.)*c[-GA_X]; c0 -= 2.0*facy; }
    else                       { lap += facy*(c[-GA_X] + c[GA_X]); }
    phi[lj*GA_X + li] = (rhs - lap)*(1.0/c0);
}

__device__ __forceinline__ double residual_at (const double* phi, int li, int lj, int i, int j, const LevBox& b,
                                               double rhs, double acf, double facx, double facy)
{
    const double* c = phi + lj*GA_X + li;
    const double p0 = c[0];
    double lap = -2.0*(facx + facy)*p0;
    if (i == b.lox)      lap += facx*((4./3.)*c[1] - 2.0*p0);
    else if (i == b.hix) lap += facx*((4./3.)*c[-1] - 2.0*p0);
    else                 lap += facx*(c[-1] + c[1]);
    if (j == b.loy)      lap += facy*((4./3.)*c[GA_X] - 2.0*p0);
    else if (j == b.hiy) lap += facy*((4./3.)*c[-GA_X] - 2.0*p0);
    else                 lap += facy*(c[-GA_X] + c[GA_X]);
    return rhs + acf*p0 - lap;
}

__device__ __forceinline__ void atomic_max_abs (unsigned long long* addr, double v)
{
    // non-negative doubles order like their bit patterns
    atomicMax(addr, (unsigned long long)__double_as_longlong(fabs(v)));
}

// phi_out = GSRB^4(phi_in or 0); optionally res = rhs - L(phi_out), max|res|, max|rhs|
template <bool CC, bool ZERO_INIT, bool DO_RES>
__global__ __launch_bounds__(256)
void k_gsrb4 (LevBox b, FView phi_out, FView rhs, FView acf, FView res, FView phi_in,
              double facx, double facy, int ntx, unsigned long long* resnorm, unsigned long long* rhsnorm)
{
    __shared__ double s_phi[2][GA_Y*GA_X];
    constexpr int E = DO_RES ? 4 : 3;                 // rim of the swept tile that is not final
    constexpr int FX = GT_X - 2*E, FY = GT_Y - 2*E;   // cells a tile finalises
    const int bx = blockIdx.x % ntx, by = blockIdx.x / ntx;
    const int gi0 = b.vlx + bx*FX - E;                // global index of swept cell (0,0)
    const int gj0 = b.vly + by*FY - E;
    const int tid = threadIdx.x;

    // fill LDS (ring included): phi_in inside the unknowns' box, 0 elsewhere
    for (int s = tid; s < GA_X*GA_Y; s += 256) {
        const int lj = s / GA_X, li = s - lj*GA_X;
        const int i = gi0 - 1 + li, j = gj0 - 1 + lj;
        double v0 = 0.0, v1 = 0.0;
        if (!ZERO_INIT && i >= b.vlx && i <= b.vhx && j >= b.vly && j <= b.vhy) {
            v0 = phi_in(i, j, 0); v1 = phi_in(i, j, 1);
        }
        s_phi[0][s] = v0; s_phi[1][s] = v1;
    }

    // per-thread cell pairs and their rhs / coefficient registers
    double r0[GPAIRS][2], r1[GPAIRS][2], ac[GPAIRS][2];
    bool in[GPAIRS][2];
    double rmax = 0.0;
#pragma unroll
    for (int m = 0; m < GPAIRS; ++m) {
        const int pi = tid + 256*m;
        const int jj = pi / (GT_X/2), pk = pi - jj*(GT_X/2);
        const int j = gj0 + jj;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = gi0 + 2*pk + h;
            const bool ok = (i >= b.vlx && i <= b.vhx && j >= b.vly && j <= b.vhy);
            in[m][h] = ok;
            r0[m][h] = ok ? rhs(i, j, 0) : 0.0;
            r1[m][h] = ok ? rhs(i, j, 1) : 0.0;
            ac[m][h] = ok ? acf(i, j, 0) : 0.0;
            if (rhsnorm && ok) rmax = fmax(rmax, fmax(fabs(r0[m][h]), fabs(r1[m][h])));
        }
    }
    __syncthreads();

    for (int icolor = 0; icolor < 4; ++icolor) {
#pragma unroll
        for (int m = 0; m < GPAIRS; ++m) {
            const int pi = tid + 256*m;
            const int jj = pi / (GT_X/2), pk = pi - jj*(GT_X/2);
            const int j = gj0 + jj;
            const int ia = gi0 + 2*pk;
            const int h = (ia + j + icolor) & 1;          // which cell of the pair has this colour
            if (in[m][h]) {
                const int i = ia + h;
                gs_update<CC>(s_phi[0], 2*pk + h + 1, jj + 1, i, j, b, r0[m][h], ac[m][h], facx, facy);
                gs_update<CC>(s_phi[1], 2*pk + h + 1, jj + 1, i, j, b, r1[m][h], ac[m][h], facx, facy);
            }
        }
        __syncthreads();
    }

    double resmax = 0.0;
#pragma unroll
    for (int m = 0; m < GPAIRS; ++m) {
        const int pi = tid + 256*m;
        const int jj = pi / (GT_X/2), pk = pi - jj*(GT_X/2);
        const int j = gj0 + jj;
        if (jj < E || jj >= GT_Y - E) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ii = 2*pk + h;
            if (ii < E || ii >= GT_X - E || !in[m][h]) continue;
            const int i = gi0 + ii;
            if (DO_RES) {
                const double q0 = residual_at(s_phi[0], ii + 1, jj + 1, i, j, b, r0[m][h], ac[m][h], facx, facy);
                const double q1 = residual_at(s_phi[1], ii + 1, jj + 1, i, j, b, r1[m][h], ac[m][h], facx, facy);
                res(i, j, 0) = q0; res(i, j, 1) = q1;
                resmax = fmax(resmax, fmax(fabs(q0), fabs(q1)));
            }
            phi_out(i, j, 0) = s_phi[0][(jj + 1)*GA_X + ii + 1];
            phi_out(i, j, 1) = s_phi[1][(jj + 1)*GA_X + ii + 1];
        }
    }
    if (DO_RES && resnorm) {
        for (int o = 32; o > 0; o >>= 1) resmax = fmax(resmax, __shfl_xor(resmax, o));
        if ((tid & 63) == 0 && resmax > 0.0) atomic_max_abs(resnorm, resmax);
    }
    if (rhsnorm) {
        for (int o = 32; o > 0; o >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, o));
        if ((tid & 63) == 0 && rmax > 0.0) atomic_max_abs(rhsnorm, rmax);
    }
}

// coarse = R(fine): 4-average (cell-centred) or 9-point full weighting (nodal)
template <bool CC>
__global__ __launch_bounds__(256)
void k_restrict (LevBox cb, FView crse, FView fine, int ncomp)
{
    const int i = cb.vlx + blockIdx.x*blockDim.x + threadIdx.x;
    const int j = cb.vly + blockIdx.y;
    if (i > cb.vhx || j > cb.vhy) return;
    for (int n = 0; n < ncomp; ++n) {
        if (CC) {
            crse(i, j, n) = 0.25*(fine(2*i, 2*j, n) + fine(2*i+1, 2*j, n) + fine(2*i, 2*j+1, n) + fine(2*i+1, 2*j+1, n));
        } else {
            crse(i, j, n) = (1./16.)*(fine(2*i-1, 2*j-1, n) + 2.*fine(2*i, 2*j-1, n) + fine(2*i+1, 2*j-1, n)
                                    + 2.*fine(2*i-1, 2*j, n) + 4.*fine(2*i, 2*j, n) + 2.*fine(2*i+1, 2*j, n)
                                    + fine(2*i-1, 2*j+1, n) + 2.*fine(2*i, 2*j+1, n) + fine(2*i+1, 2*j+1, n));
        }
    }
}

// fine_out = fine_in + P(coarse)
template <bool CC>
__global__ __launch_bounds__(256)
void k_prolong_add (LevBox fb, FView fin, FView crse, FView fout)
{
    const int i = fb.vlx + blockIdx.x*blockDim.x + threadIdx.x;
    const int j = fb.vly + blockIdx.y;
    if (i > fb.vhx || j > fb.vhy) return;
    const int ic = i >> 1, jc = j >> 1;      // indices are >= 0
    for (int n = 0; n < 2; ++n) {
        double add;
        if (CC) {
            add = crse(ic, jc, n);
        } else {
            const bool io = (i & 1), jo = (j & 1);
            if (io && jo)  add = (crse(ic, jc, n) + crse(ic+1, jc, n) + crse(ic, jc+1, n) + crse(ic+1, jc+1, n))*0.25;
            else if (io)   add = (crse(ic, jc, n) + crse(ic+1, jc, n))*0.5;
            else if (jo)   add = (crse(ic, jc, n) + crse(ic, jc+1, n))*0.5;
            else           add = crse(ic, jc, n);
        }
        fout(i, j, n) = fin(i, j, n) + add;
    }
}

// coarsest level: phi = 0, then nsweeps red-black sweeps, one workgroup, data in place
template <bool CC>
__global__ __launch_bounds__(256)
void k_bottom (LevBox b, FView phi, FView rhs, FView acf, double facx, double facy, int nsweeps)
{
    const int nvx = b.vhx - b.vlx + 1, nvy = b.vhy - b.vly + 1;
    const int nbx = b.hix - b.lox + 1, nby = b.hiy - b.loy + 1;
    for (int s = threadIdx.x; s < nbx*nby; s += blockDim.x) {
        const int jj = s / nbx, ii = s - jj*nbx;
        phi(b.lox + ii, b.loy + jj, 0) = 0.0; phi(b.lox + ii, b.loy + jj, 1) = 0.0;
    }
    __syncthreads();
    for (int is = 0; is < nsweeps; ++is) {
        for (int s = threadIdx.x; s < nvx*nvy; s += blockDim.x) {
            const int jj = s / nvx, ii = s - jj*nvx;
            const int i = b.vlx + ii, j = b.vly + jj;
            if (((i + j + is) & 1) == 0) {
                const double a = acf(i, j, 0);
                for (int n = 0; n < 2; ++n) {
                    double lap, c0 = -(a + 2.0*(facx + facy));
                    if (CC && i == b.lox)      { lap = facx*(4./3.)*phi(i+1, j, n); c0 -= 2.0*facx; }
                    else if (CC && i == b.hix) { lap = facx*(4./3.)*phi(i-1, j, n); c0 -= 2.0*facx; }
                    else                       { lap = facx*(phi(i-1, j, n) + phi(i+1, j, n)); }
                    if (CC && j == b.loy)      { lap += facy*(4./3.)*phi(i, j+1, n); c0 -= 2.0*facy; }
                    else if (CC && j == b.hiy) { lap += facy*(4./3.)*phi(i, j-1, n); c0 -= 2.0*facy; }
                    else                       { lap += facy*(phi(i, j-1, n) + phi(i, j+1, n)); }
                    phi(i, j, n) = (rhs(i, j, n) - lap)*(1.0/c0);
                }
            }
        }
        __syncthreads();
    }
}

__global__ void k_copy2 (LevBox b, FView dst, FView src)
{
    const int i = b.vlx + blockIdx.x*blockDim.x + threadIdx.x;
    const int j = b.vly + blockIdx.y;
    if (i > b.vhx || j > b.vhy) return;
    dst(i, j, 0) = src(i, j, 0); dst(i, j, 1) = src(i, j, 1);
}

struct MGLevelDev { LevBox b; long cells; double *acf, *res, *cor, *rescor; };

struct Multigrid {
    bool cc; int nx, ny; double dx, dy;
    std::vector<MGLevelDev> L;
    unsigned long long* d_norms = nullptr;      // [0] residual, [1] rhs
    unsigned long long* h_norms = nullptr;      // pinned
    // level-0 user views (set per solve)
    FView sol, rhs, acf0;

    ~Multigrid () {
        for (auto& l : L) { (void)hipFree(l.acf); (void)hipFree(l.res); (void)hipFree(l.cor); (void)hipFree(l.rescor); }
        (void)hipFree(d_norms);
        if (h_norms) (void)hipHostFree(h_norms);
    }
    FView lv (int il, double* p) const {
        const MGLevelDev& l = L[il];
        const long nxb = l.b.hix - l.b.lox + 1;
        return FView{p, nxb, l.cells, -l.b.lox, -l.b.loy};
    }
    int nlev () const { return (int)L.size(); }
};

int mg_create (int nx, int ny, double dx, double dy, Multigrid** out)
{
    if (nx % 2 != ny % 2) { set_error("hps_mg_create: nx and ny must have the same parity"); return HPS_ERR_ARG; }
    Multigrid* M = new Multigrid;
    M->cc = (nx % 2 == 0); M->nx = nx; M->ny = ny; M->dx = dx; M->dy = dy;
    int hx = M->cc ? nx - 1 : nx + 1, hy = M->cc ? ny - 1 : ny + 1;
    for (int il = 0; il < 31; ++il) {
        MGLevelDev l{};
        l.b.lox = 0; l.b.loy = 0; l.b.hix = hx; l.b.hiy = hy;
        if (M->cc) { l.b.vlx = 0; l.b.vly = 0; l.b.vhx = hx; l.b.vhy = hy; }
        else       { l.b.vlx = 1; l.b.vly = 1; l.b.vhx = hx - 1; l.b.vhy = hy - 1; }
        l.cells = (long)(hx + 1)*(hy + 1);
        HPS_HIP_CHECK(hipMalloc(&l.acf, l.cells*sizeof(double)));
        HPS_HIP_CHECK(hipMalloc(&l.res, 2*l.cells*sizeof(double)));
        HPS_HIP_CHECK(hipMalloc(&l.cor, 2*l.cells*sizeof(double)));
        HPS_HIP_CHECK(hipMalloc(&l.rescor, 2*l.cells*sizeof(double)));
        HPS_HIP_CHECK(hipMemset(l.acf, 0, l.cells*sizeof(double)));
        HPS_HIP_CHECK(hipMemset(l.res, 0, 2*l.cells*sizeof(double)));
        HPS_HIP_CHECK(hipMemset(l.cor, 0, 2*l.cells*sizeof(double)));
        HPS_HIP_CHECK(hipMemset(l.rescor, 0, 2*l.cells*sizeof(double)));
        M->L.push_back(l);
        bool ok;
        const int nxl = hx + 1, nyl = hy + 1;
        if (M->cc) { ok = nxl >= 4 && nyl >= 4 && nxl % 2 == 0 && nyl % 2 == 0; if (ok) { hx = nxl/2 - 1; hy = nyl/2 - 1; } }
        else       { ok = nxl >= 8 && nyl >= 8 && hx % 2 == 0 && hy % 2 == 0;   if (ok) { hx /= 2; hy /= 2; } }
        if (!ok) break;
    }
    if (M->nlev() < 2) { delete M; set_error("hps_mg_create: grid too small to coarsen"); return HPS_ERR_ARG; }
    HPS_HIP_CHECK(hipMalloc(&M->d_norms, 2*sizeof(unsigned long long)));
    HPS_HIP_CHECK(hipHostMalloc(&M->h_norms, 2*sizeof(unsigned long long)));
    *out = M;
    return HPS_OK;
}

template <bool CC>
static void launch_gsrb4 (Multigrid* M, int il, bool zero_init, bool do_res, FView phi_out, FView rhs, FView acf,
                          FView res, FView phi_in, double ldx, double ldy, unsigned long long* resnorm,
                          unsigned long long* rhsnorm, hipStream_t st)
{
    const LevBox& b = M->L[il].b;
    const int E = do_res ? 4 : 3;
    const int FX = GT_X - 2*E, FY = GT_Y - 2*E;
    const int ntx = ceil_div(b.vhx - b.vlx + 1, FX), nty = ceil_div(b.vhy - b.vly + 1, FY);
    const double facx = 1.0/(ldx*ldx), facy = 1.0/(ldy*ldy);
    const dim3 grid(ntx*nty), block(256);
    if (zero_init)      hipLaunchKernelGGL((k_gsrb4<CC, true, true>),  grid, block, 0, st, b, phi_out, rhs, acf, res, phi_in, facx, facy, ntx, resnorm, rhsnorm);
    else if (do_res)    hipLaunchKernelGGL((k_gsrb4<CC, false, true>), grid, block, 0, st, b, phi_out, rhs, acf, res, phi_in, facx, facy, ntx, resnorm, rhsnorm);
    else                hipLaunchKernelGGL((k_gsrb4<CC, false, false>), grid, block, 0, st, b, phi_out, rhs, acf, res, phi_in, facx, facy, ntx, resnorm, rhsnorm);
}

template <bool CC>
static void vcycle (Multigrid* M, hipStream_t st)
{
    const int maxl = M->nlev() - 1;
    for (int il = 0; il < maxl; ++il) {
        const double fac = (double)(1 << il);
        if (il > 0) {
            launch_gsrb4<CC>(M, il, true, true, M->lv(il, M->L[il].cor), M->lv(il, M->L[il].res), M->lv(il, M->L[il].acf),
                             M->lv(il, M->L[il].rescor), FView{}, M->dx*fac, M->dy*fac, nullptr, nullptr, st);
        }
        const LevBox& cb = M->L[il+1].b;
        hipLaunchKernelGGL(k_restrict<CC>, dim3(ceil_div(cb.vhx - cb.vlx + 1, 64), cb.vhy - cb.vly + 1), dim3(64), 0, st,
                           cb, M->lv(il+1, M->L[il+1].res), M->lv(il, M->L[il].rescor), 2);
    }
    {   // coarsest level (CPU branch of bottomsolve, HpMultiGrid.cpp:1583-1593)
        const double fac = (double)(1 << maxl);
        const LevBox& b = M->L[maxl].b;
        const int nsweeps = std::max(16, (std::max(b.hix - b.lox + 1, b.hiy - b.loy + 1) + 1)/2*2);
        const double ldx = M->dx*fac, ldy = M->dy*fac;
        FView rhsb = (maxl == 0) ? M->rhs : M->lv(maxl, M->L[maxl].res);
        FView acfb = (maxl == 0) ? M->acf0 : M->lv(maxl, M->L[maxl].acf);
        hipLaunchKernelGGL(k_bottom<CC>, dim3(1), dim3(256), 0, st, b, M->lv(maxl, M->L[maxl].cor), rhsb, acfb,
                           1.0/(ldx*ldx), 1.0/(ldy*ldy), nsweeps);
    }
    for (int il = maxl - 1; il >= 0; --il) {
        const double fac = (double)(1 << il);
        const LevBox& fb = M->L[il].b;
        hipLaunchKernelGGL(k_prolong_add<CC>, dim3(ceil_div(fb.vhx - fb.vlx + 1, 64), fb.vhy - fb.vly + 1), dim3(64), 0, st,
                           fb, M->lv(il, M->L[il].cor), M->lv(il+1, M->L[il+1].cor), M->lv(il, M->L[il].rescor));
        if (il == 0) launch_gsrb4<CC>(M, 0, false, false, M->sol, M->rhs, M->acf0, FView{}, M->lv(0, M->L[0].rescor),
                                      M->dx, M->dy, nullptr, nullptr, st);
        else         launch_gsrb4<CC>(M, il, false, false, M->lv(il, M->L[il].cor), M->lv(il, M->L[il].res), M->lv(il, M->L[il].acf),
                                      FView{}, M->lv(il, M->L[il].rescor), M->dx*fac, M->dy*fac, nullptr, nullptr, st);
    }
    launch_gsrb4<CC>(M, 0, false, true, M->lv(0, M->L[0].cor), M->rhs, M->acf0, M->lv(0, M->L[0].rescor), M->sol,
                     M->dx, M->dy, M->d_norms, nullptr, st);
}

static inline double norm_value (unsigned long long bits) { double d; memcpy(&d, &bits, 8); return d; }

template <bool CC>
static int solve1_impl (Multigrid* M, double tol_rel, double tol_abs, int max_iters, int* iters_out, double* resnorm_out,
                        hipStream_t st)
{
    const int nl = M->nlev();
    // coefficient hierarchy (average_down_acoef, HpMultiGrid.cpp:1640-1700); level 0 reads the slab
    for (int il = 1; il < nl; ++il) {
        const LevBox& cb = M->L[il].b;
        FView fine = (il == 1) ? M->acf0 : M->lv(il-1, M->L[il-1].acf);
        hipLaunchKernelGGL(k_restrict<CC>, dim3(ceil_div(cb.vhx - cb.vlx + 1, 64), cb.vhy - cb.vly + 1), dim3(64), 0, st,
                           cb, M->lv(il, M->L[il].acf), fine, 1);
    }
    HPS_HIP_CHECK(hipMemsetAsync(M->d_norms, 0, 2*sizeof(unsigned long long), st));
    launch_gsrb4<CC>(M, 0, false, true, M->lv(0, M->L[0].cor), M->rhs, M->acf0, M->lv(0, M->L[0].rescor), M->sol,
                     M->dx, M->dy, M->d_norms, M->d_norms + 1, st);
    HPS_HIP_CHECK(hipMemcpyAsync(M->h_norms, M->d_norms, 2*sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HPS_HIP_CHECK(hipStreamSynchronize(st));
    const double resnorm0 = norm_value(M->h_norms[0]), rhsnorm0 = norm_value(M->h_norms[1]);
    const double max_norm = (rhsnorm0 >= resnorm0) ? rhsnorm0 : resnorm0;
    const double res_target = std::max(tol_abs, std::max(tol_rel, 1.e-16)*max_norm);
    int iters = 0; double norminf = resnorm0; int status = HPS_OK;
    if (resnorm0 > res_target) {
        bool converged = false;
        for (int iter = 0; iter < max_iters; ++iter) {
            HPS_HIP_CHECK(hipMemsetAsync(M->d_norms, 0, sizeof(unsigned long long), st));
            vcycle<CC>(M, st);
            ++iters;
            HPS_HIP_CHECK(hipMemcpyAsync(M->h_norms, M->d_norms, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            HPS_HIP_CHECK(hipStreamSynchronize(st));
            norminf = norm_value(M->h_norms[0]);
            if (norminf <= res_target) { converged = true; break; }
            if (!(norminf <= 1.e20*max_norm)) { set_error("hps_mg_solve1: diverging"); status = HPS_ERR_MG_DIVERGED; break; }
        }
        if (!converged && status == HPS_OK) { set_error("hps_mg_solve1: not converged after max_iters V-cycles"); status = HPS_ERR_MG_MAXITER; }
    }
    // solution = cor[0] on the unknowns (solve_doit :1419-1426)
    const LevBox& b0 = M->L[0].b;
    hipLaunchKernelGGL(k_copy2, dim3(ceil_div(b0.vhx - b0.vlx + 1, 64), b0.vhy - b0.vly + 1), dim3(64), 0, st,
                       b0, M->sol, M->lv(0, M->L[0].cor));
    HPS_HIP_CHECK(hipGetLastError());
    if (iters_out) *iters_out = iters;
    if (resnorm_out) *resnorm_out = norminf;
    return status;
}

int mg_solve1 (Multigrid* M, hps_slab s, int sol_comp, int rhs_comp, int acf_comp, double tol_rel, double tol_abs,
               int max_iters, int* iters_out, double* resnorm_out, hipStream_t st)
{
    // centre the slab box on the level-0 box (center_box, HpMultiGrid.H:168-175)
    const int sh = M->cc ? 0 : 1;
    const int o = s.ng - sh;
    M->sol  = FView{s.p + (long)sol_comp*s.nstride, s.jstride, s.nstride, o, o};
    M->rhs  = FView{s.p + (long)rhs_comp*s.nstride, s.jstride, s.nstride, o, o};
    M->acf0 = FView{s.p + (long)acf_comp*s.nstride, s.jstride, s.nstride, o, o};
    return M->cc ? solve1_impl<true>(M, tol_rel, tol_abs, max_iters, iters_out, resnorm_out, st)
                 : solve1_impl<false>(M, tol_rel, tol_abs, max_iters, iters_out, resnorm_out, st);
}

} // namespace hps

using namespace hps;

extern "C" int hps_mg_create (int nx, int ny, double dx, double dy, void** handle)
{
    HPS_REQUIRE(nx >= 2 && ny >= 2 && handle, "hps_mg_create: bad size");
    Multigrid* M = nullptr;
    if (int e = mg_create(nx, ny, dx, dy, &M)) return e;
    *handle = M;
    return HPS_OK;
}

extern "C" int hps_mg_solve1 (void* handle, hps_slab slab, int sol_comp, int rhs_comp, int acoef_comp, double tol_rel,
                              double tol_abs, int max_iters, int* iters_host, double* resnorm_host, hps_stream stream)
{
    HPS_REQUIRE(handle && slab.p, "hps_mg_solve1: null argument");
    Multigrid* M = static_cast<Multigrid*>(handle);
    HPS_REQUIRE(slab.nx == M->nx && slab.ny == M->ny, "hps_mg_solve1: slab size does not match the solver");
    HPS_REQUIRE(sol_comp >= 0 && sol_comp + 1 < slab.ncomp && rhs_comp >= 0 && rhs_comp + 1 < slab.ncomp &&
                acoef_comp >= 0 && acoef_comp < slab.ncomp, "hps_mg_solve1: bad component");
    HPS_REQUIRE(M->cc || slab.ng >= 1, "hps_mg_solve1: node-centred solve needs >= 1 guard cell");
    return mg_solve1(M, slab, sol_comp, rhs_comp, acoef_comp, tol_rel, tol_abs, max_iters, iters_host, resnorm_host,
                     (hipStream_t)stream);
}

extern "C" int hps_mg_destroy (void* handle)
{
    delete static_cast<Multigrid*>(handle);
    return HPS_OK;
}
