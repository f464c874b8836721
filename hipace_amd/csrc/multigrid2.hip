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
// Levels with <= 4096 cells run whole sweep sequences in one 1024-thread workgroup (colour after colour behind
// __syncthreads, arrays in L2); larger levels use an LDS-tiled kernel that does the four sweeps, the residual and (up-leg)
// the prolongation in one pass.  The planes are planar [2][ny][nx] without guard cells.
#include "common.h"

#include <vector>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>

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

// max-norm accumulation: one atomic per workgroup into one of the slot's MG2_NSUB words (same-address atomics serialise in
// the L2 at ~20 ns each: one per wave of a 512-workgroup launch cost more than the launch's arithmetic); the value of a slot
// is the maximum over its words (k2_post_norms).  Call from every thread of the workgroup.
constexpr int MG2_NSUB = 16;
__device__ __forceinline__ void block_max_to_slot (unsigned long long* slot, double m)
{
    __shared__ double s_bm[16];
    for (int s = 32; s > 0; s >>= 1) m = fmax(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0) s_bm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (unsigned w = 1; w < (blockDim.x >> 6); ++w) m = fmax(m, s_bm[w]);
        if (m > 0.0) atomicMax(slot + (blockIdx.x & (MG2_NSUB - 1)), (unsigned long long)__double_as_longlong(m));
    }
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
    if (norm) block_max_to_slot(norm, m);
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
    block_max_to_slot(norm, m);      // (one atomic per wave of 2048 workgroups on one word made this reduction of 16 MB take 96 us)
}

// the two norm words -> mapped host memory, sequence number last behind a system-scope fence; the host polls it (as
// k_post_norms of multigrid.hip) instead of a DMA copy + stream synchronise
__global__ void k2_post_norms (const unsigned long long* __restrict__ src, volatile unsigned long long* dst, unsigned long long seq)
{
    unsigned long long a = 0ULL, b = 0ULL;      // non-negative doubles order like their bit patterns
    for (int q = 0; q < MG2_NSUB; ++q) { a = src[q] > a ? src[q] : a; b = src[MG2_NSUB + q] > b ? src[MG2_NSUB + q] : b; }
    dst[0] = a; dst[1] = b;
    HPS_HOST_STORES_ACKNOWLEDGED();
    dst[2] = seq;
}

// ---- LDS-tiled form of gsrb_4_residual for the large levels ----------------------------------------------------------
// A workgroup (512 threads on a 64 x 32 tile, 256 on a 32 x 16 one) owns a tile and carries a 5-cell halo: sweep s (0..3) is applied to the cells within
// 4 - s cells of the tile (what the tile's final values and its residual depend on), so four sweeps + residual cost one
// read of phi / rhs / acf (1.4-1.5x with the halo) and one write instead of five passes.  Out of place (a neighbour's
// halo read must see the old values); the up-leg's prolongation is fused into the load (src = fin + crse(i/2, j/2)).
enum { SRC2_ZERO = 0, SRC2_DIRECT = 1, SRC2_PROLONG = 2 };
// Tile 64 x 32 on the large levels (LDS region 74 x 42, swept region 72 x 40 = 12 cells per thread); 32 x 16 on levels of
// <= 256^2 cells, where the count of workgroups and the latency of a launch matter more than the halo's extra reads.
template <int T2X, int T2Y, int NT>
__global__ __launch_bounds__(NT)
void k2_smooth_tile (Lev2 l, int src_mode, const double* __restrict__ fin, const double* __restrict__ crse, int cnx, long cn,
                     double* __restrict__ phi_out, const double* __restrict__ rhs, const double* __restrict__ acf,
                     int do_res, double* __restrict__ res, unsigned long long* norm, double* __restrict__ crse_res, int crx, long crn)
{
    constexpr int T2H = 5, T2W = T2X + 2*T2H, T2HH = T2Y + 2*T2H;
    constexpr int T2RW = T2X + 8, T2RH = T2Y + 8, T2K = (T2RW*T2RH + NT - 1)/NT;
    __shared__ double sp[2][T2HH][T2W];
    const int ti0 = blockIdx.x*T2X, tj0 = blockIdx.y*T2Y;
    // load phi on the tile + halo (zero outside the domain: never read by the wall stencils)
    for (int q = threadIdx.x; q < T2W*T2HH; q += NT) {
        const int lj = q / T2W, li = q - lj*T2W;
        const int gi = ti0 - T2H + li, gj = tj0 - T2H + lj;
        double vr = 0.0, vi = 0.0;
        if (src_mode != SRC2_ZERO && gi >= 0 && gi < l.nx && gj >= 0 && gj < l.ny) {
            const long o = (long)gj*l.nx + gi;
            vr = fin[o]; vi = fin[l.n + o];
            if (src_mode == SRC2_PROLONG) { const long oc = (long)(gj/2)*cnx + gi/2; vr += crse[oc]; vi += crse[cn + oc]; }
        }
        sp[0][lj][li] = vr; sp[1][lj][li] = vi;
    }
    // per-cell constants of the swept region in registers: rhs and the inverse of the diagonal (gs2 :325-330)
    double rr[T2K], ri[T2K], cr[T2K], ci[T2K];
#pragma unroll
    for (int k = 0; k < T2K; ++k) {
        const int q = threadIdx.x + k*NT;
        const int rj = q / T2RW, rix = q - rj*T2RW;
        const int gi = ti0 - 4 + rix, gj = tj0 - 4 + rj;
        rr[k] = 0.0; ri[k] = 0.0; cr[k] = 0.0; ci[k] = 0.0;
        if (q < T2RW*T2RH && gi >= 0 && gi < l.nx && gj >= 0 && gj < l.ny) {
            const long o = (long)gj*l.nx + gi;
            double c0 = -2.0*(l.facx + l.facy);
            if (gi == 0 || gi == l.nx - 1) c0 -= 2.0*l.facx;
            if (gj == 0 || gj == l.ny - 1) c0 -= 2.0*l.facy;
            double a = c0 - acf[o], b = -acf[l.n + o];
            const double cmag = 1.0/(a*a + b*b);
            cr[k] = a*cmag; ci[k] = b*cmag;
            rr[k] = rhs[o]; ri[k] = rhs[l.n + o];
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int k = 0; k < T2K; ++k) {
            const int q = threadIdx.x + k*NT;
            const int rj = q / T2RW, rix = q - rj*T2RW;
            const int gi = ti0 - 4 + rix, gj = tj0 - 4 + rj;
            // distance to the tile, colour, domain
            const int dxh = max(max(4 - rix, rix - (T2X + 3)), 0), dyh = max(max(4 - rj, rj - (T2Y + 3)), 0);
            if (q >= T2RW*T2RH || max(dxh, dyh) > 4 - s || ((gi + gj + s) & 1) || gi < 0 || gi >= l.nx || gj < 0 || gj >= l.ny) continue;
            const int li = rix + 1, lj = rj + 1;
            double lap0, lap1;
            if (gi == 0)             { lap0 = l.facx*(4./3.)*sp[0][lj][li + 1]; lap1 = l.facx*(4./3.)*sp[1][lj][li + 1]; }
            else if (gi == l.nx - 1) { lap0 = l.facx*(4./3.)*sp[0][lj][li - 1]; lap1 = l.facx*(4./3.)*sp[1][lj][li - 1]; }
            else { lap0 = l.facx*(sp[0][lj][li - 1] + sp[0][lj][li + 1]); lap1 = l.facx*(sp[1][lj][li - 1] + sp[1][lj][li + 1]); }
            if (gj == 0)             { lap0 += l.facy*(4./3.)*sp[0][lj + 1][li]; lap1 += l.facy*(4./3.)*sp[1][lj + 1][li]; }
            else if (gj == l.ny - 1) { lap0 += l.facy*(4./3.)*sp[0][lj - 1][li]; lap1 += l.facy*(4./3.)*sp[1][lj - 1][li]; }
            else { lap0 += l.facy*(sp[0][lj - 1][li] + sp[0][lj + 1][li]); lap1 += l.facy*(sp[1][lj - 1][li] + sp[1][lj + 1][li]); }
            const double a = rr[k] - lap0, b = ri[k] - lap1;
            sp[0][lj][li] = a*cr[k] + b*ci[k];
            sp[1][lj][li] = b*cr[k] - a*ci[k];
        }
        __syncthreads();
    }
    // tile out; residual2r / 2i (:192-208), its max-norm and -- with a coarser level -- its restriction (restrict_cc
    // :29-37) straight from registers: the fine residual then never goes to memory
    double m = 0.0;
    auto residual = [&] (int gi, int gj, int li, int lj, double& r0, double& r1) {
        const long o = (long)gj*l.nx + gi;
        const double pr = sp[0][lj][li], pi = sp[1][lj][li];
        double lp[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const double p0 = sp[n][lj][li];
            double lap = -2.0*(l.facx + l.facy)*p0;
            if (gi == 0)             lap += l.facx*((4./3.)*sp[n][lj][li + 1] - 2.0*p0);
            else if (gi == l.nx - 1) lap += l.facx*((4./3.)*sp[n][lj][li - 1] - 2.0*p0);
            else                     lap += l.facx*(sp[n][lj][li - 1] + sp[n][lj][li + 1]);
            if (gj == 0)             lap += l.facy*((4./3.)*sp[n][lj + 1][li] - 2.0*p0);
            else if (gj == l.ny - 1) lap += l.facy*((4./3.)*sp[n][lj - 1][li] - 2.0*p0);
            else                     lap += l.facy*(sp[n][lj - 1][li] + sp[n][lj + 1][li]);
            lp[n] = lap;
        }
        const double ar = acf[o], ai = acf[l.n + o];
        r0 = rhs[o] - lp[0] + (ar*pr - ai*pi); r1 = rhs[l.n + o] - lp[1] + (ai*pr + ar*pi);
    };
    if (do_res && crse_res) {
        for (int q = threadIdx.x; q < (T2X/2)*(T2Y/2); q += NT) {
            const int cj = q / (T2X/2), cix = q - cj*(T2X/2);
            const int gi0 = ti0 + 2*cix, gj0 = tj0 + 2*cj;
            if (gi0 >= l.nx || gj0 >= l.ny) continue;          // nx, ny even: the 2 x 2 block is inside or outside as a whole
            double s0 = 0.0, s1 = 0.0;
            // order of restrict_cc: (2i,2j) + (2i+1,2j) + (2i,2j+1) + (2i+1,2j+1)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int gi = gi0 + (b & 1), gj = gj0 + (b >> 1);
                const int li = gi - ti0 + T2H, lj = gj - tj0 + T2H;
                const long o = (long)gj*l.nx + gi;
                phi_out[o] = sp[0][lj][li]; phi_out[l.n + o] = sp[1][lj][li];
                double r0, r1;
                residual(gi, gj, li, lj, r0, r1);
                s0 += r0; s1 += r1;
                m = fmax(m, fmax(fabs(r0), fabs(r1)));
            }
            const long oc = (long)(gj0/2)*crx + gi0/2;
            crse_res[oc] = 0.25*s0; crse_res[crn + oc] = 0.25*s1;
        }
    } else
    for (int q = threadIdx.x; q < T2X*T2Y; q += NT) {
        const int tj = q / T2X, tix = q - tj*T2X;
        const int gi = ti0 + tix, gj = tj0 + tj;
        if (gi >= l.nx || gj >= l.ny) continue;
        const int li = tix + T2H, lj = tj + T2H;
        const long o = (long)gj*l.nx + gi;
        phi_out[o] = sp[0][lj][li]; phi_out[l.n + o] = sp[1][lj][li];
        if (!do_res) continue;
        double r0, r1;
        residual(gi, gj, li, lj, r0, r1);
        res[o] = r0; res[l.n + o] = r1;
        m = fmax(m, fmax(fabs(r0), fabs(r1)));
    }
    if (do_res && norm) block_max_to_slot(norm, m);
}

// ---- the lower part of the V-cycle in one workgroup ------------------------------------------------------------------
// All levels from the first one with <= 256 cells down to the coarsest: zero guess, 4 sweeps, residual, restriction on
// the way down, the bottom sweeps, prolongation + 4 sweeps on the way up -- arrays in LDS, one launch instead of ~5 per level.
constexpr int LOW2_MAX_CELLS = 256, LOW2_MAX_LEVELS = 8;
struct Low2 { int nlev; Lev2 l[LOW2_MAX_LEVELS]; int off[LOW2_MAX_LEVELS]; int numsweeps; };

__device__ __forceinline__ void lds_sweeps (const Lev2& l, double* phi, const double* rhs, const double* acf, int nsweeps)
{
    for (int c = 0; c < nsweeps; ++c) {
        for (int o = threadIdx.x; o < (int)l.n; o += blockDim.x) {
            const int j = o / l.nx, i = o - j*l.nx;
            if (((i + j + c) & 1) == 0) gs2_point(l, i, j, phi, rhs, acf);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256)
void k2_lower_v (Low2 lv, const double* __restrict__ rhs_top, const double* const* __restrict__ acf_levels, double* __restrict__ cor_top)
{
    __shared__ double lds[8*(LOW2_MAX_CELLS + LOW2_MAX_CELLS/4 + LOW2_MAX_CELLS/16 + LOW2_MAX_CELLS/64 + 8)];
    auto PHI = [&] (int k) { return lds + lv.off[k]; };
    auto RHS = [&] (int k) { return lds + lv.off[k] + 2*lv.l[k].n; };
    auto ACF = [&] (int k) { return lds + lv.off[k] + 4*lv.l[k].n; };
    auto TMP = [&] (int k) { return lds + lv.off[k] + 6*lv.l[k].n; };
    for (int k = 0; k < lv.nlev; ++k)
        for (int o = threadIdx.x; o < 2*(int)lv.l[k].n; o += blockDim.x) {
            ACF(k)[o] = acf_levels[k][o];
            PHI(k)[o] = 0.0;
            if (k == 0) RHS(0)[o] = rhs_top[o];
        }
    __syncthreads();
    for (int k = 0; k < lv.nlev - 1; ++k) {
        const Lev2& l = lv.l[k]; const Lev2& c = lv.l[k + 1];
        lds_sweeps(l, PHI(k), RHS(k), ACF(k), 4);
        for (int o = threadIdx.x; o < (int)l.n; o += blockDim.x) {
            const int j = o / l.nx, i = o - j*l.nx;
            const double pr = PHI(k)[o], pi = PHI(k)[l.n + o], ar = ACF(k)[o], ai = ACF(k)[l.n + o];
            TMP(k)[o] = RHS(k)[o] - lap2(l, i, j, PHI(k) + o) + (ar*pr - ai*pi);
            TMP(k)[l.n + o] = RHS(k)[l.n + o] - lap2(l, i, j, PHI(k) + l.n + o) + (ai*pr + ar*pi);
        }
        __syncthreads();
        for (int o = threadIdx.x; o < 2*(int)c.n; o += blockDim.x) {
            const int n = o / (int)c.n, q = o - n*(int)c.n;
            const int j = q / c.nx, i = q - j*c.nx;
            const double* p = TMP(k) + (long)n*l.n + (long)(2*j)*l.nx + 2*i;
            RHS(k + 1)[o] = 0.25*(p[0] + p[1] + p[l.nx] + p[l.nx + 1]);
        }
        __syncthreads();
    }
    lds_sweeps(lv.l[lv.nlev - 1], PHI(lv.nlev - 1), RHS(lv.nlev - 1), ACF(lv.nlev - 1), lv.numsweeps);
    for (int k = lv.nlev - 2; k >= 0; --k) {
        const Lev2& l = lv.l[k]; const Lev2& c = lv.l[k + 1];
        for (int o = threadIdx.x; o < 2*(int)l.n; o += blockDim.x) {
            const int n = o / (int)l.n, q = o - n*(int)l.n;
            const int j = q / l.nx, i = q - j*l.nx;
            PHI(k)[o] += PHI(k + 1)[(long)n*c.n + (long)(j/2)*c.nx + i/2];
        }
        __syncthreads();
        lds_sweeps(l, PHI(k), RHS(k), ACF(k), 4);
    }
    for (int o = threadIdx.x; o < 2*(int)lv.l[0].n; o += blockDim.x) cor_top[o] = PHI(0)[o];
}

struct Multigrid2 {
    int nx = 0, ny = 0; double dx = 0, dy = 0;
    std::vector<Lev2> L;
    std::vector<double*> acf, res, cor, rescor;      // per level, 2 planes each
    unsigned long long* d_norm = nullptr;            // [2][MG2_NSUB]: residual, rhs (a slot's value = the maximum over its words)
    unsigned long long *h_post = nullptr, *h_post_dev = nullptr, seq = 0;      // mapped pinned: [0] residual, [1] rhs norm, [2] sequence
    bool res1_ready = false;      // res[1] already holds the restricted level-0 residual (fused into the tile kernel)
    int low_top = -1; Low2 low{}; double** d_low_acf = nullptr;      // levels low_top .. coarsest run in k2_lower_v
    long total_vcycles = 0;
    ~Multigrid2 () {
        for (auto& v : {acf, res, cor, rescor}) for (double* p : v) (void)hipFree(p);
        (void)hipFree(d_norm); (void)hipFree(d_low_acf);
        if (h_post) (void)hipHostFree(h_post);
    }
};

static const long SINGLE_BLOCK_CELLS = LOW2_MAX_CELLS;      // above: the LDS-tiled kernel

// gsrb_4_residual (:742-848): out = 4 sweeps of src (zero | fin | fin + prolonged crse), optionally res = rhs - L(out) and
// its max-norm.  out must not alias fin.  With crse_res (the next level's right-hand side) the tiled path restricts the
// residual itself and returns true (res is then not written); false: the caller restricts res.
static bool smooth4 (const Lev2& l, int src_mode, const double* fin, const double* crse, const Lev2* cl, double* out,
                     const double* rhs, const double* acf, bool do_res, double* res, unsigned long long* norm, hipStream_t st,
                     double* crse_res = nullptr, const Lev2* crl = nullptr)
{
    if (l.n > SINGLE_BLOCK_CELLS) {
        static const long big = getenv("HPS_MG2_BIG") ? atol(getenv("HPS_MG2_BIG")) : 256L*256L;      // measured: 65536 = 262144 > all small
        const bool fuse = do_res && crse_res && crl && l.nx % 2 == 0 && l.ny % 2 == 0;
        if (l.n > big)
            hipLaunchKernelGGL((k2_smooth_tile<64, 32, 512>), dim3(ceil_div(l.nx, 64), ceil_div(l.ny, 32)), dim3(512), 0, st, l, src_mode, fin, crse,
                               cl ? cl->nx : 0, cl ? cl->n : 0L, out, rhs, acf, do_res ? 1 : 0, res, norm, fuse ? crse_res : (double*)nullptr,
                               fuse ? crl->nx : 0, fuse ? crl->n : 0L);
        else
            hipLaunchKernelGGL((k2_smooth_tile<32, 16, 256>), dim3(ceil_div(l.nx, 32), ceil_div(l.ny, 16)), dim3(256), 0, st, l, src_mode, fin, crse,
                               cl ? cl->nx : 0, cl ? cl->n : 0L, out, rhs, acf, do_res ? 1 : 0, res, norm, fuse ? crse_res : (double*)nullptr,
                               fuse ? crl->nx : 0, fuse ? crl->n : 0L);
        return fuse;
    }
    if (src_mode == SRC2_ZERO) (void)hipMemsetAsync(out, 0, (size_t)2*l.n*sizeof(double), st);
    else if (src_mode == SRC2_DIRECT) (void)hipMemcpyAsync(out, fin, (size_t)2*l.n*sizeof(double), hipMemcpyDeviceToDevice, st);
    else hipLaunchKernelGGL(k2_interp_add, dim3(ceil_div(2*l.n, 256)), dim3(256), 0, st, l, *cl, out, fin, crse);
    hipLaunchKernelGGL(k2_sweeps, dim3(1), dim3(1024), 0, st, l, out, rhs, acf, 0, 4, do_res ? 1 : 0, res, norm);
    return false;
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
    HPS_HIP_CHECK(hipMalloc(&M->d_norm, 2*MG2_NSUB*sizeof(unsigned long long)));
    HPS_HIP_CHECK(hipHostMalloc(&M->h_post, 4*sizeof(unsigned long long), hipHostMallocMapped));
    HPS_HIP_CHECK(hipHostGetDevicePointer((void**)&M->h_post_dev, M->h_post, 0));
    M->h_post[0] = M->h_post[1] = M->h_post[2] = 0ULL;
    // the lower V: from the first level (not level 0) with <= LOW2_MAX_CELLS cells
    const int nl = (int)M->L.size();
    for (int il = 1; il < nl; ++il) if (M->L[il].n <= LOW2_MAX_CELLS && nl - il <= LOW2_MAX_LEVELS) { M->low_top = il; break; }
    if (M->low_top >= 1) {
        Low2& lv = M->low; lv.nlev = nl - M->low_top;
        int off = 0; std::vector<double*> ptrs;
        for (int k = 0; k < lv.nlev; ++k) { lv.l[k] = M->L[M->low_top + k]; lv.off[k] = off; off += 8*(int)lv.l[k].n; ptrs.push_back(M->acf[M->low_top + k]); }
        const Lev2& lb = M->L[nl - 1];
        lv.numsweeps = std::max(16, (std::max(lb.nx, lb.ny) + 1)/2*2);
        if (off > 8*(LOW2_MAX_CELLS + LOW2_MAX_CELLS/4 + LOW2_MAX_CELLS/16 + LOW2_MAX_CELLS/64 + 8)) M->low_top = -1;      // odd shapes: generic path
        else {
            HPS_HIP_CHECK(hipMalloc(&M->d_low_acf, ptrs.size()*sizeof(double*)));
            HPS_HIP_CHECK(hipMemcpy(M->d_low_acf, ptrs.data(), ptrs.size()*sizeof(double*), hipMemcpyHostToDevice));
        }
    }
    *out = M;
    return HPS_OK;
}

// both norms (residual, rhs) of the stream's work so far
static int read_norms (Multigrid2* M, double* resnorm, double* rhsnorm, hipStream_t st)
{
    ++M->seq;
    hipLaunchKernelGGL(k2_post_norms, dim3(1), dim3(1), 0, st, M->d_norm, (volatile unsigned long long*)M->h_post_dev, M->seq);
    volatile unsigned long long* hs = M->h_post + 2;
    long spins = 0;
    while (*hs != M->seq) {
        if ((++spins & 0xfffff) == 0 && hipStreamQuery(st) != hipErrorNotReady) {      // a failed launch must not hang us
            if (*hs == M->seq) break;
            HPS_HIP_CHECK(hipStreamSynchronize(st));
            if (*hs != M->seq) { set_error("hps_mg2_solve2: the norm read-back never arrived"); return HPS_ERR_HIP; }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    auto val = [&] (int k) { unsigned long long b = ((volatile unsigned long long*)M->h_post)[k]; double d; std::memcpy(&d, &b, sizeof(d)); return d; };
    *resnorm = val(0);
    if (rhsnorm) *rhsnorm = val(1);
    return HPS_OK;
}

// vcycle (:1429-1512): on entry rescor[0] holds the level-0 residual, on exit again (of the improved solution in sol)
static void vcycle2 (Multigrid2* M, double* sol, const double* rhs, hipStream_t st)
{
    const int maxl = (int)M->L.size() - 1;
    const int top = (M->low_top >= 1) ? M->low_top : maxl;      // the levels [top, maxl] are the bottom part
    for (int il = 0; il < top; ++il) {
        const Lev2& l = M->L[il]; const Lev2& c = M->L[il + 1];
        bool restricted = (il == 0) ? M->res1_ready : false;      // level 0: done by the smooth that produced the residual
        if (il > 0) restricted = smooth4(l, SRC2_ZERO, nullptr, nullptr, nullptr, M->cor[il], M->res[il], M->acf[il], true, M->rescor[il], nullptr, st,
                                         M->res[il + 1], &c);
        if (!restricted) hipLaunchKernelGGL(k2_restrict, dim3(ceil_div(2*c.n, 256)), dim3(256), 0, st, c, l, M->res[il + 1], M->rescor[il], 2);
    }
    if (M->low_top >= 1) {
        hipLaunchKernelGGL(k2_lower_v, dim3(1), dim3(256), 0, st, M->low, M->res[top], (const double* const*)M->d_low_acf, M->cor[top]);
    } else {   // bottomsolve, CPU branch (:1583-1593)
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
    for (int il = top - 1; il >= 0; --il) {
        const Lev2& l = M->L[il]; const Lev2& c = M->L[il + 1];
        // interpolation + 4 sweeps; the result replaces cor[il] (level 0: the solution)
        double* out = (il == 0) ? sol : M->rescor[il];
        smooth4(l, SRC2_PROLONG, M->cor[il], M->cor[il + 1], &c, out, (il == 0) ? rhs : M->res[il], M->acf[il], false, nullptr, nullptr, st);
        if (il > 0) std::swap(M->cor[il], M->rescor[il]);
    }
    // cor0 = 4 more sweeps of the solution, residual behind them
    (void)hipMemsetAsync(M->d_norm, 0, MG2_NSUB*sizeof(unsigned long long), st);
    M->res1_ready = smooth4(M->L[0], SRC2_DIRECT, sol, nullptr, nullptr, M->cor[0], rhs, M->acf[0], true, M->rescor[0], M->d_norm, st,
                            maxl >= 1 ? M->res[1] : nullptr, maxl >= 1 ? &M->L[1] : nullptr);
}

static int mg2_solve2 (Multigrid2* M, double* sol, const double* rhs, const double* acf_real, const double* acf_imag, double tol_rel,
                       double tol_abs, int maxiter, int* iters_host, double* resnorm_host, hipStream_t st)
{
    const Lev2& l0 = M->L[0];
    hipLaunchKernelGGL(k2_fill_acf, dim3(ceil_div(l0.n, 256)), dim3(256), 0, st, l0, M->acf[0], acf_real, acf_imag);
    for (size_t il = 1; il < M->L.size(); ++il)      // average_down_acoef (:1640-1700)
        hipLaunchKernelGGL(k2_restrict, dim3(ceil_div(2*M->L[il].n, 256)), dim3(256), 0, st, M->L[il], M->L[il - 1], M->acf[il], M->acf[il - 1], 2);
    // solve_doit (:1307-1427)
    (void)hipMemsetAsync(M->d_norm, 0, 2*MG2_NSUB*sizeof(unsigned long long), st);
    const bool has1 = M->L.size() >= 2;
    M->res1_ready = smooth4(l0, SRC2_DIRECT, sol, nullptr, nullptr, M->cor[0], rhs, M->acf[0], true, M->rescor[0], M->d_norm, st,
                            has1 ? M->res[1] : nullptr, has1 ? &M->L[1] : nullptr);
    hipLaunchKernelGGL(k2_maxabs, dim3((unsigned)std::min<long>(ceil_div(2*l0.n, 2048), 512)), dim3(256), 0, st, rhs, 2*l0.n, M->d_norm + MG2_NSUB);
    HPS_HIP_CHECK(hipGetLastError());
    double resnorm0 = 0.0, rhsnorm0 = 0.0;
    if (int e = read_norms(M, &resnorm0, &rhsnorm0, st)) return e;
    const double max_norm = (rhsnorm0 >= resnorm0) ? rhsnorm0 : resnorm0;
    const double res_target = std::max(tol_abs, std::max(tol_rel, 1.e-16)*max_norm);
    int iters = 0; double norminf = resnorm0;
    if (resnorm0 > res_target) {
        bool converged = false, diverged = false;
        for (int iter = 0; iter < maxiter; ++iter) {
            vcycle2(M, sol, rhs, st);
            HPS_HIP_CHECK(hipGetLastError());
            ++iters;
            if (int e = read_norms(M, &norminf, nullptr, st)) return e;
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
