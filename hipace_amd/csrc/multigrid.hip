// multigrid.hip -- geometric multigrid for  -a*phi + Lap(phi) = rhs  (2 right-hand sides sharing
// one coefficient a: the explicit Bx/By solve), homogeneous Dirichlet walls, on gfx950.
//
// Numerically a restatement of hpmg::MultiGrid system type 1 (mg_solver/HpMultiGrid.cpp):
// V-cycle with 4 red-black Gauss-Seidel sweeps per level (colour (i+j+s)%2, s = 0..3), residual
// behind the sweeps, cell-centred 4-average / nodal full-weighting restriction, piecewise
// constant / bilinear prolongation, 16 sweeps on the coarsest level, cell-centred wall stencil
// with the 4/3-2 coefficients (HpMultiGrid.cpp:162-182,265-292), stop rule of solve_doit
// (:1307-1427).  Because Bx/By are only converged to tol_rel = 1e-4 from the previous slice's
// field, parity of the slice engine requires this exact per-cell arithmetic.
//
// MI355X mapping (one HBM read + one HBM write of each plane per smoothing step):
//  * k_smooth: one LDS-tiled kernel per level and direction.  A 256-thread workgroup sweeps a
//    64x32-cell tile 4 times in LDS (phi in LDS; rhs, coefficient and 1/diagonal in registers),
//    and fuses what the reference does in separate passes: the prolongation of the coarse
//    correction into the tile load (up-leg), the residual, its max-norm and -- cell-centred --
//    its restriction to the next level (down-leg: the fine residual never touches HBM).
//  * k_lower_v: every level with <= 32x32 unknowns runs inside ONE 1024-thread workgroup with all
//    its arrays in LDS (whole lower V incl. the 16 bottom sweeps), replacing ~4 launches per level.
#include "common.h"
#include <type_traits>
#include "mg_gate.h"
#include "slab_ops.h"

#include <vector>
#include <algorithm>
#include <cmath>
#include <cstring>

namespace hps {

// 2-D multi-component view in level index space: (i,j,n) -> p[(i+oi) + (j+oj)*js + n*ns]
// (HPS_MG_OFF32: the element's byte offset in 32-bit arithmetic from the view's base -- global_load / global_store with the
//  base in SGPRs and one offset VGPR instead of a 64-bit address per access: the level-0 passes spend more instructions on
//  index arithmetic than on fp64; the solver's planes hold at most 2^28 doubles, checked in mg_create)
// 1: the fused 8-sweep level-0 pass evaluates its V-cycle's gate behind its tile's loads, as the 4-sweep kernels do (0: first, a
// dependent trip to memory ahead of the loads).  With the gate behind them all of the tile's operands are live across it: 61
// registers spilled under the 128 the kernel may use for two workgroups per CU (round 4).  Off.
#ifndef HPS_MG_GATE8_BEHIND
#define HPS_MG_GATE8_BEHIND 0
#endif
#ifndef HPS_MG_OFF32
#define HPS_MG_OFF32 1
#endif
// HPS_MG_BLOCKROWS: a thread's GPAIRS cell pairs sit in GPAIRS consecutive rows of the tile (a 2 x GPAIRS block of cells) instead
// of rows NT/PR apart, and the values it has written last stay in registers: of the four neighbours of a cell it updates, the
// partner in its pair and the cells above / below inside its block come from registers, not from LDS -- 5 instead of 12 LDS reads
// per half-sweep and component for GPAIRS = 3.  (The sweeps of the fused level-0 pass ran at 1900 cycles per half-sweep, which is
// what the LDS pipe carries for two workgroups of 123 KB each.)
// Measured (round 4): the kernels with GPAIRS = 2 (64 x 32 tiles: the initial level-0 pass 24.4 -> 22.0 us, the level-1 passes
// 9.1 -> 8.2) gain; the fused 8-sweep level-0 pass (GPAIRS = 3, 128 registers for two workgroups per CU) does not have the 24
// registers for the values and takes 39.2 instead of 32.2 us: block rows up to HPS_MG_BLOCKROWS_MAXG pairs per thread only.
#ifndef HPS_MG_BLOCKROWS
#define HPS_MG_BLOCKROWS 1
#endif
// ... and, HPS_MG_BLOCKROWS_INTERIOR, for any number of pairs in the tiles that touch no wall (their path has no wall multipliers
// and masks in registers)
#ifndef HPS_MG_BLOCKROWS_INTERIOR
#define HPS_MG_BLOCKROWS_INTERIOR 1
#endif
#ifndef HPS_MG_BLOCKROWS_MAXG
#define HPS_MG_BLOCKROWS_MAXG 2
#endif
struct FView {
    double* p; long js, ns; int oi, oj;
    __device__ __forceinline__ double& operator() (int i, int j, int n) const {
#if HPS_MG_OFF32
        const unsigned o = (unsigned)((i + oi) + (j + oj)*(int)js + n*(int)ns)*8u;
        return *reinterpret_cast<double*>(reinterpret_cast<char*>(p) + o);
#else
        return p[(long)(i + oi) + (long)(j + oj)*js + (long)n*ns];
#endif
    }
};

struct LevBox { int lox, loy, hix, hiy;      // index bounds of the level box (walls for nodal)
                int vlx, vly, vhx, vhy; };   // unknowns (valid_domain_box)

enum { SRC_ZERO = 0, SRC_DIRECT = 1, SRC_PROLONG = 2 };

// A value that must be in its register here: keeps the compiler from sinking a load below the gate that follows it.
#define HPS_KEEP(x) asm volatile("" :: "v"(x))

// optional shader-clock stamps of workgroup 0 (set through hps_mg_debug_stamps)
// -- only in the diagnostic build (make stamps: -DHPS_STAMPS): the read of the pointer is one more dependent trip to
// memory at the head of every kernel, ~1 us behind a kernel boundary
#ifdef HPS_STAMPS
__device__ long long* g_mg_dbg = nullptr;
#define MG_STAMP(i) do { if (g_mg_dbg && blockIdx.x == 0 && threadIdx.x == 0) g_mg_dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MG_STAMP(i) do { } while (0)
#endif

// Smoother tile shapes: TX x TY cells swept by NT threads (TX*TY/2/NT cell pairs per thread).  A
// workgroup streams its tile through one CU (~10 B/clk from memory, one instruction stream per
// SIMD), so the levels with few cells get small tiles -- more workgroups on more CUs -- and the
// fine levels large ones (less rim re-computation: the 4 sweeps need a rim of 3..4 cells).
template <int TX_, int TY_, int NT_>
struct TileShape {
    static constexpr int TX = TX_, TY = TY_, NT = NT_;
    static constexpr int AX = TX + 2, AY = TY + 2;          // LDS array with ring
    static constexpr int PR = TX/2;                          // cell pairs per row
    static constexpr int GPAIRS = TX*TY/2/NT;                // cell pairs per thread
    static_assert(TX*TY/2 % NT == 0 && NT % 64 == 0 && (PR == 16 || PR == 32), "tile shape");
};
#ifndef HPS_MG_BIG
#define HPS_MG_BIG 64, 32, 512
#endif
using TileBig = TileShape<HPS_MG_BIG>;
using TileMid = TileShape<32, 32, 256>;
#ifndef HPS_MG_HUGE
#define HPS_MG_HUGE 64, 48, 512
#endif
using TileHuge = TileShape<HPS_MG_HUGE>;      // the fused 8-sweep pass of level 0 (rim 8: 48 x 32 of 64 x 48 final)
// ... of the node-centred hierarchy (2^K - 1 cells per side): on 64 x 48 tiles that pass spills 37 registers under its cap
// (no fused restriction: the residual planes are stored, the prolongation is bilinear) -- 64 x 32 tiles (two pairs per thread)
// do not: Bx/By solve at 1023^2 413 -> 391 us per slice (profiles/r05_nodal_multigrid.txt); the cell-centred pass keeps 64 x 48
#ifndef HPS_MG_HUGE_NODAL
#define HPS_MG_HUGE_NODAL 64, 32, 512
#endif
using TileHugeNodal = TileShape<HPS_MG_HUGE_NODAL>;
template <bool CC> struct HugeTileOf { using type = TileHuge; };
template <> struct HugeTileOf<false> { using type = TileHugeNodal; };
using TileSmall = TileShape<32, 16, 256>;

// diagonal of the operator at (i,j): -(a + 2(fx+fy)) with the wall modification (gs1 :265-292)
template <bool CC>
__device__ __forceinline__ double diag_c0 (int i, int j, const LevBox& b, double acf, double facx, double facy)
{
    double c0 = -(acf + 2.0*(facx + facy));
    if (CC && (i == b.lox || i == b.hix)) c0 -= 2.0*facx;
    if (CC && (j == b.loy || j == b.hiy)) c0 -= 2.0*facy;
    return c0;
}

// Off-diagonal part of the stencil at (i,j); `c` points at the cell, `sy` is the row stride.
// Branch-free: all four neighbours are read unconditionally (the arrays carry a ring / walls), the
// cell-centred wall variants (gs1 :265-292: facx*(4/3)*phi instead of facx*(phi_w+phi_e)) are picked
// with selects, so the result is bit-identical to the branching form while the loads of several
// cells can be in flight together.  INTERIOR = no cell of the tile touches a wall.
template <bool CC, bool INTERIOR = false, class P>
__device__ __forceinline__ double offdiag (P c, int sy, int i, int j, const LevBox& b, double facx, double facy)
{
    const double w = c[-1], e = c[1], s = c[-sy], n = c[sy];
    double lx = facx*(w + e), ly = facy*(s + n);
    if (CC && !INTERIOR) {
        const double fx43 = facx*(4./3.), fy43 = facy*(4./3.);
        lx = (i == b.lox) ? fx43*e : ((i == b.hix) ? fx43*w : lx);
        ly = (j == b.loy) ? fy43*n : ((j == b.hiy) ? fy43*s : ly);
    }
    return lx + ly;
}

// The same with per-cell multipliers, for arrays whose cells outside the unknowns' box hold 0:
// fxm = facx*(4/3) on a cell-centred wall column, facx elsewhere (fym alike), so that
// fxm*(w + e) is facx*(4/3)*e at the low wall (w = 0 and 0 + e == e), facx*(4/3)*w at the high wall
// and facx*(w + e) inside -- the values of gs1 :265-292 without a select or a branch per neighbour.
template <class P>
__device__ __forceinline__ double offdiag_m (P c, int sy, double fxm, double fym)
{
    return fxm*(c[-1] + c[1]) + fym*(c[-sy] + c[sy]);
}
template <bool CC>
__device__ __forceinline__ double wall_mult (int i, int lo, int hi, double fac)
{
    return (CC && (i == lo || i == hi)) ? fac*(4./3.) : fac;
}

// residual rhs - L(phi) at (i,j) (laplacian :162-182, residual1 :184-190), branch-free as above
template <bool INTERIOR = false>
__device__ __forceinline__ double residual_v (double p0, double w, double e, double s, double n, int i, int j, const LevBox& b,
                                              double rhs, double acf, double facx, double facy);

template <bool INTERIOR = false, class P>
__device__ __forceinline__ double residual_at (P c, int sy, int i, int j, const LevBox& b,
                                               double rhs, double acf, double facx, double facy)
{
    return residual_v<INTERIOR>(c[0], c[-1], c[1], c[-sy], c[sy], i, j, b, rhs, acf, facx, facy);
}

template <bool INTERIOR>
__device__ __forceinline__ double residual_v (double p0, double w, double e, double s, double n, int i, int j, const LevBox& b,
                                              double rhs, double acf, double facx, double facy)
{
    double lap = -2.0*(facx + facy)*p0;
    double tx = facx*(w + e), ty = facy*(s + n);
    if (!INTERIOR) {
        tx = (i == b.lox) ? facx*((4./3.)*e - 2.0*p0) : ((i == b.hix) ? facx*((4./3.)*w - 2.0*p0) : tx);
        ty = (j == b.loy) ? facy*((4./3.)*n - 2.0*p0) : ((j == b.hiy) ? facy*((4./3.)*s - 2.0*p0) : ty);
    }
    lap += tx;
    lap += ty;
    return rhs + acf*p0 - lap;
}

// P(coarse) at fine point (i,j): piecewise constant (cell-centred) or bilinear (nodal)
template <bool CC>
__device__ __forceinline__ double prolong_at (const FView& crse, int i, int j, int n)
{
    const int ic = i >> 1, jc = j >> 1;      // indices are >= 0
    if (CC) return crse(ic, jc, n);
    const bool io = (i & 1), jo = (j & 1);
    if (io && jo)  return (crse(ic, jc, n) + crse(ic+1, jc, n) + crse(ic, jc+1, n) + crse(ic+1, jc+1, n))*0.25;
    if (io)        return (crse(ic, jc, n) + crse(ic+1, jc, n))*0.5;
    if (jo)        return (crse(ic, jc, n) + crse(ic, jc+1, n))*0.5;
    return crse(ic, jc, n);
}

// Device-side stopping rule (solve_doit :1352-1398).  norms[0] = initial residual norm, norms[1] =
// rhs norm, norms[2+k] = residual norm after V-cycle k (all zeroed before a solve).  V-cycles are
// enqueued speculatively; every kernel of V-cycle k evaluates the rule itself and returns at once
// if V-cycle k-1 already met the target (an inactive V-cycle leaves its slot at 0, which switches
// off all later ones).  k < 0: unconditional.
constexpr int MG_MAX_VCYCLES = 1024;
// max-norm accumulation: one fire-and-forget atomic per workgroup into one of the slot's words
template <int NT>
__device__ __forceinline__ void block_max_to (unsigned long long* slot, double v, double* s_red)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = 0.0;
        for (int w = 0; w < NT/64; ++w) m = fmax(m, s_red[w]);
        if (m > 0.0) atomicMax(slot + (blockIdx.x & (MG_NSUB - 1)), (unsigned long long)__double_as_longlong(m));
    }
    __syncthreads();
}

// row (within the tile) and pair index of a thread's m-th cell pair (BR: block rows, see HPS_MG_BLOCKROWS)
#define MG_ROWPK(m) const int jj = BR ? (tid / PR)*GPAIRS + (m) : (tid + MG_NT*(m)) / PR, pk = BR ? (tid & (PR - 1)) : (tid + MG_NT*(m)) - jj*PR
// phi_out = GSRB^4(start), start = 0 | phi_in | phi_in + P(crse);
// DO_RES: residual r = rhs - L(phi_out), max|r| (and max|rhs|) -> norms;
//         FUSE_R (cell-centred): cres = R(r) written straight to the next level; else r -> res_out.
// INTERIOR tiles (no swept cell on a wall / outside the box) take a path without masks.
template <bool CC> __device__ __forceinline__ double restrict_at (const FView& fine, int i, int j, int n);
// RPULL (node-centred down-leg): the tile's right-hand side IS the restriction of the finer level's residual -- `crse` then
// is that residual (level above, twice the index), formed while the tile loads (9 reads per cell) and written to `rhs` for
// the up-leg by the tile that finalises the cell: no k_restrict launch between two levels' smoothers
template <class TS, bool CC, int SRC, bool DO_RES, bool FUSE_R, bool INTERIOR, int NSW, bool GATE_IN = (NSW == 4), bool RPULL = false>
__device__ __forceinline__ void smooth_tile (double (&s_phi)[2][TS::AY*TS::AX], double* s_red, double* s_crs, const LevBox& b, const FView& phi_out,
                                             const FView& phi_out2, const FView& rhs, const FView& acf, const FView& phi_in, const FView& crse,
                                             const FView& res_out, const FView& cres_out, double facx, double facy,
                                             int gi0, int gj0, unsigned long long* resnorm, unsigned long long* rhsnorm, const StopRule& sr)
{
    constexpr int E = DO_RES ? NSW : NSW - 1;         // rim of the swept tile that is not final
    constexpr int GT_X = TS::TX, GT_Y = TS::TY, GA_X = TS::AX, GA_Y = TS::AY, MG_NT = TS::NT, GPAIRS = TS::GPAIRS, PR = TS::PR;
    constexpr int AXH = GA_X/2, CH = GA_Y*AXH;        // entries per row / per plane of one colour
    constexpr bool BR = HPS_MG_BLOCKROWS && (GPAIRS <= HPS_MG_BLOCKROWS_MAXG || (INTERIOR && HPS_MG_BLOCKROWS_INTERIOR));
    const int tid = threadIdx.x;
    const int cpar = (gi0 + gj0) & 1;                 // colour of ringed cell (0, 0) is (gi0 - 1 + gj0 - 1) & 1
    MG_STAMP(0);
    // per-thread cell pairs: rhs, coefficient and inverse diagonal stay in registers.  Index p is the
    // sweep parity in which the cell is updated (p = 0: sweeps 0 and 2, p = 1: sweeps 1 and 3), so
    // that every register array below is indexed by compile-time constants only.
    double r0[GPAIRS][2], r1[GPAIRS][2], ac[GPAIRS][2], ci[GPAIRS][2];
    bool in[GPAIRS][2];
    int hx[GPAIRS];                                   // x offset (0/1) within the pair of the p = 0 cell
    double fxm[GPAIRS][2], fym[GPAIRS];               // wall multipliers (boundary tiles only)
    double rmax = 0.0;
#pragma unroll
    for (int m = 0; m < GPAIRS; ++m) {
        MG_ROWPK(m);
        const int j = gj0 + jj;
        hx[m] = (gi0 + 2*pk + j) & 1;                 // colour 0 cell: (i + j) even
        fym[m] = INTERIOR ? facy : wall_mult<CC>(j, b.loy, b.hiy, facy);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int i = gi0 + 2*pk + (hx[m] ^ p);
            fxm[m][p] = INTERIOR ? facx : wall_mult<CC>(i, b.lox, b.hix, facx);
            const int ic = INTERIOR ? i : min(max(i, b.vlx), b.vhx), jc = INTERIOR ? j : min(max(j, b.vly), b.vhy);
            const bool ok = INTERIOR || (ic == i && jc == j);
            in[m][p] = ok;
            double a0, a1;
            if constexpr (RPULL) { a0 = restrict_at<CC>(crse, ic, jc, 0); a1 = restrict_at<CC>(crse, ic, jc, 1); }
            else { a0 = rhs(ic, jc, 0); a1 = rhs(ic, jc, 1); }
            const double a2 = acf(ic, jc, 0);
            r0[m][p] = ok ? a0 : 0.0;
            r1[m][p] = ok ? a1 : 0.0;
            ac[m][p] = ok ? a2 : 0.0;
            ci[m][p] = 1.0/diag_c0<CC>(i, j, b, ac[m][p], facx, facy);
            if (rhsnorm) rmax = fmax(rmax, fmax(fabs(r0[m][p]), fabs(r1[m][p])));
        }
    }
    MG_STAMP(1);
    // fill LDS (ring included): start value inside the unknowns' box, 0 elsewhere.  Loads are
    // unconditional (clamped address + select) and all issued before the first LDS store.
    {
        constexpr int NF = (GA_X*GA_Y + MG_NT - 1)/MG_NT;
        double v0[NF], v1[NF];
        // Node-centred prolongation (bilinear: up to four coarse values per fine cell) THROUGH LDS: asked for from global memory
        // inside the loop below, the 7 cells of a thread had up to 70 loads in flight -- the fused 8-sweep pass of level 0 spilled
        // 39 registers under its 128-register cap (58 us against 32 for the cell-centred pass, n = 1023).  The coarse cells under
        // the ringed tile go to LDS once (one load per coarse cell instead of one per fine neighbour), the interpolation reads them there.
        constexpr bool PLDS = !CC && SRC == SRC_PROLONG;
        constexpr int CXN = GA_X/2 + 2, CYN = GA_Y/2 + 2;
        const int cx0 = (INTERIOR ? gi0 - 1 : min(max(gi0 - 1, b.vlx), b.vhx)) >> 1, cy0 = (INTERIOR ? gj0 - 1 : min(max(gj0 - 1, b.vly), b.vhy)) >> 1;
        if constexpr (PLDS) {
            const int cxl = b.vlx >> 1, cxh = (b.vhx >> 1) + 1, cyl = b.vly >> 1, cyh = (b.vhy >> 1) + 1;      // the coarse level's box, walls included
            constexpr int NC = (CXN*CYN + MG_NT - 1)/MG_NT;
            double c0[NC], c1[NC];
#pragma unroll
            for (int m = 0; m < NC; ++m) {
                const int q = min(tid + MG_NT*m, CXN*CYN - 1);
                const int ly = q / CXN, lx = q - ly*CXN;
                const int X = min(max(cx0 + lx, cxl), cxh), Y = min(max(cy0 + ly, cyl), cyh);
                c0[m] = crse(X, Y, 0); c1[m] = crse(X, Y, 1);
            }
#pragma unroll
            for (int m = 0; m < NC; ++m) {
                const int q = tid + MG_NT*m;
                if (q < CXN*CYN) { s_crs[q] = c0[m]; s_crs[CXN*CYN + q] = c1[m]; }
            }
        }
#pragma unroll
        for (int m = 0; m < NF; ++m) {
            const int s = min(tid + MG_NT*m, GA_X*GA_Y - 1);
            const int lj = s / GA_X, li = s - lj*GA_X;
            const int i = gi0 - 1 + li, j = gj0 - 1 + lj;
            v0[m] = 0.0; v1[m] = 0.0;
            if (SRC != SRC_ZERO) {
                const int ic = INTERIOR ? i : min(max(i, b.vlx), b.vhx), jc = INTERIOR ? j : min(max(j, b.vly), b.vhy);
                double a0 = phi_in(ic, jc, 0), a1 = phi_in(ic, jc, 1);
                if (SRC == SRC_PROLONG && !PLDS) { a0 += prolong_at<CC>(crse, ic, jc, 0); a1 += prolong_at<CC>(crse, ic, jc, 1); }
                const bool inside = INTERIOR || (ic == i && jc == j);
                v0[m] = inside ? a0 : 0.0; v1[m] = inside ? a1 : 0.0;
            }
        }
        if constexpr (PLDS) {
            __syncthreads();
#pragma unroll
            for (int m = 0; m < NF; ++m) {
                const int s = min(tid + MG_NT*m, GA_X*GA_Y - 1);
                const int lj = s / GA_X, li = s - lj*GA_X;
                const int i = gi0 - 1 + li, j = gj0 - 1 + lj;
                const int ic = INTERIOR ? i : min(max(i, b.vlx), b.vhx), jc = INTERIOR ? j : min(max(j, b.vly), b.vhy);
                const bool inside = INTERIOR || (ic == i && jc == j);
                // prolong_at<false>'s four cases with its order of additions, on the LDS copy
                const int o = ((jc >> 1) - cy0)*CXN + (ic >> 1) - cx0;
                const bool io = (ic & 1), jo = (jc & 1);
                double p0, p1;
                const double* q0 = s_crs + o; const double* q1 = s_crs + CXN*CYN + o;
                if (io && jo)  { p0 = (q0[0] + q0[1] + q0[CXN] + q0[CXN + 1])*0.25; p1 = (q1[0] + q1[1] + q1[CXN] + q1[CXN + 1])*0.25; }
                else if (io)   { p0 = (q0[0] + q0[1])*0.5; p1 = (q1[0] + q1[1])*0.5; }
                else if (jo)   { p0 = (q0[0] + q0[CXN])*0.5; p1 = (q1[0] + q1[CXN])*0.5; }
                else           { p0 = q0[0]; p1 = q1[0]; }
                if (inside) { v0[m] += p0; v1[m] += p1; }
            }
        }
        {   // The gate of a speculative V-cycle, read while ALL the tile's loads are in flight (a kernel's first dependent
            // read of global memory costs ~1.5 us behind a kernel boundary; read first, the gate was one such trip ahead
            // of the loads).  HPS_KEEP: the loads must not sink below the branch.
            // Only the 4-sweep kernels of the coarser levels do this (a handful of workgroups, each a chain of latencies);
            // the 8-sweep pass of level 0 is bound by bandwidth and registers (all its operands live at once would cost
            // it a workgroup per CU) and reads its gate first.
            if (GATE_IN) {
                const bool active = vcycle_active(sr);
#pragma unroll
                for (int m = 0; m < GPAIRS; ++m) { HPS_KEEP(r0[m][0]); HPS_KEEP(r0[m][1]); HPS_KEEP(r1[m][0]); HPS_KEEP(r1[m][1]); HPS_KEEP(ac[m][0]); HPS_KEEP(ac[m][1]); }
#pragma unroll
                for (int m = 0; m < NF; ++m) { HPS_KEEP(v0[m]); HPS_KEEP(v1[m]); }
                if (!active) return;
            }
        }
        // LDS layout: the two colours of the red-black ordering in separate planes (cell (li, lj) of the
        // ringed tile -> plane (i + j) & 1, row lj, entry li >> 1): a half-sweep then reads and writes
        // consecutive doubles per lane instead of every other one (which is a 2-way bank conflict).
#pragma unroll
        for (int m = 0; m < NF; ++m) {
            const int s = tid + MG_NT*m;
            if (s < GA_X*GA_Y) {
                const int lj = s / GA_X, li = s - lj*GA_X;
                const int o = ((cpar + li + lj) & 1)*CH + lj*AXH + (li >> 1);
                s_phi[0][o] = v0[m]; s_phi[1][o] = v1[m];
            }
        }
    }
    __syncthreads();
    MG_STAMP(2);

    double l0[GPAIRS][2], l1[GPAIRS][2];              // BR: the thread's own cells, as it wrote them last (pair, sweep parity of the cell)
    if constexpr (BR) {
        // the thread's own cells, as it wrote them last (index: pair, sweep parity of the cell)
#pragma unroll
        for (int m = 0; m < GPAIRS; ++m) {
            MG_ROWPK(m);
            const int row = (jj + 1)*AXH + pk;
#pragma unroll
            for (int p = 0; p < 2; ++p) { const int h = hx[m] ^ p; l0[m][p] = s_phi[0][p*CH + row + h]; l1[m][p] = s_phi[1][p*CH + row + h]; }
        }
#pragma unroll
        for (int icolor = 0; icolor < NSW; ++icolor) {
            const int p = icolor & 1;                     // compile-time after unrolling
#pragma unroll
            for (int m = 0; m < GPAIRS; ++m) {
                MG_ROWPK(m);
                const int h = hx[m] ^ p;
                // self: plane p, entry pk + h.  West / east: the partner (register) and entry pk + h of plane 1-p (the pair's other side);
                // south / north: the block's own rows (registers) or entry pk + h of the rows below / above it
                const int row = (jj + 1)*AXH + pk;
                const double* nb0 = &s_phi[0][(1 - p)*CH + row];
                const double* nb1 = &s_phi[1][(1 - p)*CH + row];
                const double we0 = nb0[h] + l0[m][1 - p], we1 = nb1[h] + l1[m][1 - p];      // (west + east: a + b = b + a to the bit)
                const double s0 = (m > 0) ? l0[m > 0 ? m - 1 : 0][1 - p] : nb0[h - AXH], s1 = (m > 0) ? l1[m > 0 ? m - 1 : 0][1 - p] : nb1[h - AXH];
                const double t0 = (m + 1 < GPAIRS) ? l0[m + 1 < GPAIRS ? m + 1 : m][1 - p] : nb0[h + AXH], t1 = (m + 1 < GPAIRS) ? l1[m + 1 < GPAIRS ? m + 1 : m][1 - p] : nb1[h + AXH];
                const double n0 = (r0[m][p] - (fxm[m][p]*we0 + fym[m]*(s0 + t0)))*ci[m][p];
                const double n1 = (r1[m][p] - (fxm[m][p]*we1 + fym[m]*(s1 + t1)))*ci[m][p];
                if (INTERIOR || in[m][p]) { s_phi[0][p*CH + row + h] = n0; s_phi[1][p*CH + row + h] = n1; l0[m][p] = n0; l1[m][p] = n1; }
            }
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int icolor = 0; icolor < NSW; ++icolor) {
            constexpr int dummy = 0; (void)dummy;
            const int p = icolor & 1;                     // compile-time after unrolling
#pragma unroll
            for (int m = 0; m < GPAIRS; ++m) {
                MG_ROWPK(m);
                const int h = hx[m] ^ p;
                // self: plane p, entry pk + h; west / east: plane 1-p, entries pk, pk + 1; south / north: pk + h
                const int row = (jj + 1)*AXH + pk;
                const double* nb0 = &s_phi[0][(1 - p)*CH + row];
                const double* nb1 = &s_phi[1][(1 - p)*CH + row];
                const double n0 = (r0[m][p] - (fxm[m][p]*(nb0[0] + nb0[1]) + fym[m]*(nb0[h - AXH] + nb0[h + AXH])))*ci[m][p];
                const double n1 = (r1[m][p] - (fxm[m][p]*(nb1[0] + nb1[1]) + fym[m]*(nb1[h - AXH] + nb1[h + AXH])))*ci[m][p];
                if (INTERIOR || in[m][p]) { s_phi[0][p*CH + row + h] = n0; s_phi[1][p*CH + row + h] = n1; }
            }
            __syncthreads();
        }
    }
    MG_STAMP(3);

    double resmax = 0.0;
    if constexpr (BR) {
        double la0[GPAIRS], ra0[GPAIRS], la1[GPAIRS], ra1[GPAIRS];      // FUSE_R: residual of the left / right cell of a pair, per component
        bool finr[GPAIRS];
#pragma unroll
        for (int m = 0; m < GPAIRS; ++m) {
            MG_ROWPK(m);
            const int j = gj0 + jj;
            const bool rowok = (jj >= E && jj < GT_Y - E);
            double q0[2], q1[2];                          // by sweep parity p
            bool fin[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int h = hx[m] ^ p;
                const int ii = 2*pk + h;
                fin[p] = rowok && ii >= E && ii < GT_X - E && in[m][p];
                const int i = gi0 + ii;
                const int row = (jj + 1)*AXH + pk;
                const double f0 = l0[m][p], f1 = l1[m][p];
                q0[p] = 0.0; q1[p] = 0.0;
                if (DO_RES) {
                    const double* nb0 = &s_phi[0][(1 - p)*CH + row];
                    const double* nb1 = &s_phi[1][(1 - p)*CH + row];
                    const double o0 = nb0[h], o1 = nb1[h];                      // the pair's other side; the partner is in a register
                    const double w0 = h ? l0[m][1 - p] : o0, e0 = h ? o0 : l0[m][1 - p];
                    const double w1 = h ? l1[m][1 - p] : o1, e1 = h ? o1 : l1[m][1 - p];
                    const double s0 = (m > 0) ? l0[m > 0 ? m - 1 : 0][1 - p] : nb0[h - AXH], s1 = (m > 0) ? l1[m > 0 ? m - 1 : 0][1 - p] : nb1[h - AXH];
                    const double n0 = (m + 1 < GPAIRS) ? l0[m + 1 < GPAIRS ? m + 1 : m][1 - p] : nb0[h + AXH], n1 = (m + 1 < GPAIRS) ? l1[m + 1 < GPAIRS ? m + 1 : m][1 - p] : nb1[h + AXH];
                    const double t0 = residual_v<INTERIOR>(f0, w0, e0, s0, n0, i, j, b, r0[m][p], ac[m][p], facx, facy);
                    const double t1 = residual_v<INTERIOR>(f1, w1, e1, s1, n1, i, j, b, r1[m][p], ac[m][p], facx, facy);
                    q0[p] = fin[p] ? t0 : 0.0; q1[p] = fin[p] ? t1 : 0.0;
                    resmax = fmax(resmax, fmax(fabs(q0[p]), fabs(q1[p])));
                }
                if (fin[p]) {
                    if constexpr (RPULL) { rhs(i, j, 0) = r0[m][p]; rhs(i, j, 1) = r1[m][p]; }
                    if (DO_RES && !FUSE_R) { res_out(i, j, 0) = q0[p]; res_out(i, j, 1) = q1[p]; }
                    phi_out(i, j, 0) = f0;
                    phi_out(i, j, 1) = f1;
                    if (phi_out2.p) { phi_out2(i, j, 0) = f0; phi_out2(i, j, 1) = f1; }
                }
            }
            const bool sw = (hx[m] != 0);             // the p = 0 cell is the right one
            la0[m] = sw ? q0[1] : q0[0]; ra0[m] = sw ? q0[0] : q0[1];
            la1[m] = sw ? q1[1] : q1[0]; ra1[m] = sw ? q1[0] : q1[1];
            finr[m] = fin[0];
        }
        if (FUSE_R) {
            // restrict_cc (:29-37): 0.25*(((a+b)+c)+d), a,b = left,right cell of an even row, c,d the row above.  Tile origins are
            // even, so a pair is one coarse cell's x-extent; the row above an even row is the thread's next pair, or -- for the last
            // row of its block -- the first pair of the thread PR lanes on (a wave holds 64/PR consecutive blocks, an even number of
            // rows from an even row on: the pairs that straddle two blocks never straddle two waves)
            const double c0 = __shfl_down(la0[0], PR), d0 = __shfl_down(ra0[0], PR);
            const double c1 = __shfl_down(la1[0], PR), d1 = __shfl_down(ra1[0], PR);
#pragma unroll
            for (int m = 0; m < GPAIRS; ++m) {
                MG_ROWPK(m);
                if (finr[m] && ((jj & 1) == 0)) {
                    const bool own = (m + 1 < GPAIRS);
                    const int mu = own ? m + 1 : m;
                    const double ua0 = own ? la0[mu] : c0, ub0 = own ? ra0[mu] : d0, ua1 = own ? la1[mu] : c1, ub1 = own ? ra1[mu] : d1;
                    const int ic = (gi0 + 2*pk) >> 1, jc = (gj0 + jj) >> 1;
                    cres_out(ic, jc, 0) = 0.25*(la0[m] + ra0[m] + ua0 + ub0);
                    cres_out(ic, jc, 1) = 0.25*(la1[m] + ra1[m] + ua1 + ub1);
                }
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < GPAIRS; ++m) {
            MG_ROWPK(m);
            const int j = gj0 + jj;
            const bool rowok = (jj >= E && jj < GT_Y - E);
            double q0[2], q1[2];                          // by sweep parity p
            bool fin[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int h = hx[m] ^ p;
                const int ii = 2*pk + h;
                fin[p] = rowok && ii >= E && ii < GT_X - E && in[m][p];
                const int i = gi0 + ii;
                const int row = (jj + 1)*AXH + pk;
                const double f0 = s_phi[0][p*CH + row + h], f1 = s_phi[1][p*CH + row + h];
                q0[p] = 0.0; q1[p] = 0.0;
                if (DO_RES) {
                    const double* nb0 = &s_phi[0][(1 - p)*CH + row];
                    const double* nb1 = &s_phi[1][(1 - p)*CH + row];
                    const double t0 = residual_v<INTERIOR>(f0, nb0[0], nb0[1], nb0[h - AXH], nb0[h + AXH], i, j, b, r0[m][p], ac[m][p], facx, facy);
                    const double t1 = residual_v<INTERIOR>(f1, nb1[0], nb1[1], nb1[h - AXH], nb1[h + AXH], i, j, b, r1[m][p], ac[m][p], facx, facy);
                    q0[p] = fin[p] ? t0 : 0.0; q1[p] = fin[p] ? t1 : 0.0;
                    resmax = fmax(resmax, fmax(fabs(q0[p]), fabs(q1[p])));
                }
                if (fin[p]) {
                    if constexpr (RPULL) { rhs(i, j, 0) = r0[m][p]; rhs(i, j, 1) = r1[m][p]; }
                    if (DO_RES && !FUSE_R) { res_out(i, j, 0) = q0[p]; res_out(i, j, 1) = q1[p]; }
                    phi_out(i, j, 0) = f0;
                    phi_out(i, j, 1) = f1;
                    if (phi_out2.p) { phi_out2(i, j, 0) = f0; phi_out2(i, j, 1) = f1; }
                }
            }
            if (FUSE_R) {
                // restrict_cc (:29-37): 0.25*(((a+b)+c)+d), a,b = left,right cell of this row (even j),
                // c,d the row above.  Tile origins are even, so a pair is one coarse cell's x-extent and
                // lanes l / l+32 of a wave hold rows jj / jj+1.
                const bool sw = (hx[m] != 0);             // the p = 0 cell is the right one
                const double la0 = sw ? q0[1] : q0[0], ra0 = sw ? q0[0] : q0[1];
                const double la1 = sw ? q1[1] : q1[0], ra1 = sw ? q1[0] : q1[1];
                const double c0 = __shfl_down(la0, PR), d0 = __shfl_down(ra0, PR);
                const double c1 = __shfl_down(la1, PR), d1 = __shfl_down(ra1, PR);
                if (fin[0] && ((tid & PR) == 0)) {
                    const int ic = (gi0 + 2*pk) >> 1, jc = j >> 1;
                    cres_out(ic, jc, 0) = 0.25*(la0 + ra0 + c0 + d0);
                    cres_out(ic, jc, 1) = 0.25*(la1 + ra1 + c1 + d1);
                }
            }
        }
    }
    MG_STAMP(4);
    if (DO_RES && resnorm) block_max_to<MG_NT>(resnorm, resmax, s_red);
    if (rhsnorm) block_max_to<MG_NT>(rhsnorm, rmax, s_red);
    MG_STAMP(5);
}

// NSW red-black half-sweeps per launch: 4 (one GSRB^4 of the reference) or 8 (the two consecutive
// GSRB^4 that end a V-cycle on level 0, fused: one pass over HBM instead of two)
// POST: the launch that ends the last V-cycle enqueued so far also does what k_post_norms does (below): the workgroup that
// finishes last -- a counter -- evaluates the stopping rule for the kernels gated behind the solve and posts the norms to the
// host.  One launch (4.7 us between the last V-cycle and the gated push) less on a slice's chain.
struct PostArgs { const unsigned long long* src; volatile unsigned long long* dst; int nwords; volatile unsigned long long* seq_slot;
                  unsigned long long seq; int* go_word; StopRule after; unsigned int* counter; };

__device__ __forceinline__ void post_epilogue (const PostArgs& pa)      // every workgroup of the launch, whole, at its end
{
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        // This workgroup's norm atomics (thread 0's, block_max_to) must have been performed before it is counted: wait for
        // their acknowledgement.  NOT a __threadfence(): a device-scope release on this GPU writes the XCD's L2 back, once per
        // workgroup -- measured: the launch took 66 us instead of 34.  The norms are device-scope atomics, read back below
        // by device-scope atomic loads: no cache in between.
        HPS_OWN_ATOMICS_ACKNOWLEDGED();
        // two levels of counters (16 + 1): one counter for all workgroups is a chain of ~500 same-address atomics of ~20 ns each
        const unsigned nb = gridDim.x, c = blockIdx.x & 15u, want = (nb - c + 15u) >> 4;
        int last = 0;
        if (atomicAdd(pa.counter + 1 + c, 1u) == want - 1u) {
            __hip_atomic_store(pa.counter + 1 + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (atomicAdd(pa.counter, 1u) == (nb < 16u ? nb : 16u) - 1u) ? 1 : 0;
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(pa.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {   const bool act = vcycle_active<true>(pa.after);
        if (threadIdx.x == 0) *pa.go_word = act ? 0 : 1; }
    for (int w = threadIdx.x; w < pa.nwords; w += blockDim.x)
        pa.dst[w] = __hip_atomic_load(pa.src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    HPS_HOST_STORES_ACKNOWLEDGED();
    __syncthreads();
    if (threadIdx.x == 0) *pa.seq_slot = pa.seq;
}

template <class TS, bool CC, int SRC, bool DO_RES, bool FUSE_R, int NSW = 4, bool POST = false, bool RPULL = false>
#ifndef HPS_MG_NODAL_WAVES
#define HPS_MG_NODAL_WAVES 4
#endif
__global__ __launch_bounds__(TS::NT, (RPULL && TS::GPAIRS > 1) ? 2 : (CC || NSW == 4) ? 4 : HPS_MG_NODAL_WAVES)      // at most 128 VGPRs: two 512-thread workgroups per CU (several variants sit at 113-130); the pulling smoother with two pairs per thread (36 + 36 loads in flight) gets 256
void k_smooth (LevBox b, FView phi_out, FView phi_out2, FView rhs, FView acf, FView phi_in, FView crse, FView res_out,
               FView cres_out, double facx, double facy, int ntx, unsigned long long* resnorm,
               unsigned long long* rhsnorm, StopRule sr, PostArgs pa)
{
    static_assert(!FUSE_R || (CC && DO_RES), "fused restriction is cell-centred only");
    static_assert(!POST || NSW != 4, "the post rides on the fused level-0 pass");
    if (NSW != 4 && !(HPS_MG_GATE8_BEHIND && !POST) && !vcycle_active(sr)) { if (POST) post_epilogue(pa); return; }       // (the 4-sweep kernels read the gate behind their loads, see smooth_tile)
    constexpr int GT_X = TS::TX, GT_Y = TS::TY;
    __shared__ double s_phi[2][TS::AY*TS::AX];
    __shared__ double s_red[TS::NT/64];
    // node-centred prolongation goes through LDS (smooth_tile): the coarse cells under the ringed tile, two components
    constexpr bool PLDS = !CC && SRC == SRC_PROLONG;
    __shared__ double s_crs[PLDS ? 2*(TS::AX/2 + 2)*(TS::AY/2 + 2) : 1];
    constexpr int E = DO_RES ? NSW : NSW - 1;
    constexpr int FX = GT_X - 2*E, FY = GT_Y - 2*E;   // cells a tile finalises (even numbers)
    // workgroups go round-robin to the 8 XCDs: give each XCD a contiguous run of tiles, so that the
    // rims shared by neighbouring tiles hit in that XCD's L2
    const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = blockIdx.x & 7;
    const int tile = xcd*q8 + min(xcd, r8) + (blockIdx.x >> 3);
    const int bx = tile % ntx, by = tile / ntx;
    const int gi0 = b.vlx + bx*FX - E;                // global index of swept cell (0,0)
    const int gj0 = b.vly + by*FY - E;
    // every swept cell and its ring strictly inside the unknowns' box and off the walls
    const bool interior = (gi0 - 1 >= b.vlx) && (gi0 + GT_X <= b.vhx) && (gj0 - 1 >= b.vly) && (gj0 + GT_Y <= b.vhy)
                       && (gi0 > b.lox) && (gi0 + GT_X - 1 < b.hix) && (gj0 > b.loy) && (gj0 + GT_Y - 1 < b.hiy);
    constexpr bool GATE_IN = (NSW == 4) || (HPS_MG_GATE8_BEHIND && !POST);
    if (interior) smooth_tile<TS, CC, SRC, DO_RES, FUSE_R, true, NSW, GATE_IN, RPULL>(s_phi, s_red, s_crs, b, phi_out, phi_out2, rhs, acf, phi_in, crse, res_out, cres_out,
                                                             facx, facy, gi0, gj0, resnorm, rhsnorm, sr);
    else          smooth_tile<TS, CC, SRC, DO_RES, FUSE_R, false, NSW, GATE_IN, RPULL>(s_phi, s_red, s_crs, b, phi_out, phi_out2, rhs, acf, phi_in, crse, res_out, cres_out,
                                                              facx, facy, gi0, gj0, resnorm, rhsnorm, sr);
    if (POST) post_epilogue(pa);
}

// coarse = R(fine): 4-average (cell-centred) or 9-point full weighting (nodal)
template <bool CC>
__device__ __forceinline__ double restrict_at (const FView& fine, int i, int j, int n)
{
    if (CC) return 0.25*(fine(2*i, 2*j, n) + fine(2*i+1, 2*j, n) + fine(2*i, 2*j+1, n) + fine(2*i+1, 2*j+1, n));
    return (1./16.)*(fine(2*i-1, 2*j-1, n) + 2.*fine(2*i, 2*j-1, n) + fine(2*i+1, 2*j-1, n)
                   + 2.*fine(2*i-1, 2*j, n) + 4.*fine(2*i, 2*j, n) + 2.*fine(2*i+1, 2*j, n)
                   + fine(2*i-1, 2*j+1, n) + 2.*fine(2*i, 2*j+1, n) + fine(2*i+1, 2*j+1, n));
}

template <bool CC>
__global__ __launch_bounds__(256)
void k_restrict (LevBox cb, FView crse, FView fine, int ncomp, StopRule sr)
{
    if (!vcycle_active(sr)) return;
    const int i = cb.vlx + blockIdx.x*blockDim.x + threadIdx.x;
    const int j = cb.vly + blockIdx.y;
    if (i > cb.vhx || j > cb.vhy) return;
    for (int n = 0; n < ncomp; ++n) crse(i, j, n) = restrict_at<CC>(fine, i, j, n);
}

// ---- all small levels in one workgroup, LDS resident --------------------------------------------
typedef __attribute__((address_space(3))) double lds_double;

// per small level: box, row length, points, offset (in doubles) of its 8-plane block in LDS:
// [acf | res0 res1 | cor0 cor1 | rescor0 rescor1 | 1/diagonal]
struct LowLev { LevBox b; int nxb; int cells; int off; };

struct LView {      // one component plane of a small level in LDS, level index space
    lds_double* p; int nxb; int lox, loy;
    __device__ __forceinline__ lds_double& operator() (int i, int j) const { return p[(i - lox) + (j - loy)*nxb]; }
};
__device__ __forceinline__ LView lplane (lds_double* base, const LowLev& l, int plane)
{
    return LView{base + l.off + plane*l.cells, l.nxb, l.b.lox, l.b.loy};
}

template <bool CC>
__device__ __forceinline__ double lrestrict (const LView& f, int i, int j)
{
    if (CC) return 0.25*(f(2*i, 2*j) + f(2*i+1, 2*j) + f(2*i, 2*j+1) + f(2*i+1, 2*j+1));
    return (1./16.)*(f(2*i-1, 2*j-1) + 2.*f(2*i, 2*j-1) + f(2*i+1, 2*j-1)
                   + 2.*f(2*i-1, 2*j) + 4.*f(2*i, 2*j) + 2.*f(2*i+1, 2*j)
                   + f(2*i-1, 2*j+1) + 2.*f(2*i, 2*j+1) + f(2*i+1, 2*j+1));
}

template <bool CC>
__device__ __forceinline__ double lprolong (const LView& c, int i, int j)
{
    const int ic = i >> 1, jc = j >> 1;
    if (CC) return c(ic, jc);
    const bool io = (i & 1), jo = (j & 1);
    if (io && jo)  return (c(ic, jc) + c(ic+1, jc) + c(ic, jc+1) + c(ic+1, jc+1))*0.25;
    if (io)        return (c(ic, jc) + c(ic+1, jc))*0.5;
    if (jo)        return (c(ic, jc) + c(ic, jc+1))*0.5;
    return c(ic, jc);
}

// A level is worked either by the whole workgroup -- its threads as a 32-wide patch swept over the level (no integer divisions in
// the loops), phases separated by __syncthreads -- or, round 6, by WAVE 0 ALONE: levels of at most LOWV_WAVE_CELLS points (15^2
// unknowns and below) as a 16 x 4 patch of the 64 lanes, phases separated by a fence only.  A phase of the generic lower V costs
// ~900 clocks whatever the level holds (the barrier of 16 waves + an LDS round trip: shader-clock stamps, scripts/diag_mg.py,
// 84 k clocks per V-cycle at 1023^2 of which the 16 sweeps of the 3 x 3 bottom level alone 14.8 k); in one wave it is the LDS
// round trip.  Same per-point expressions, same values: only who computes them changes.
// MEASURED (round 6, profiles/r06_lowv_wave_ab.txt, 1023^2): Bx/By solve 371.8 us per slice with every level on the whole workgroup,
// 396.3 with the levels of <= 17^2 points on wave 0 alone, 488 with the 33^2 level there too: one wave has nothing to hide its own
// LDS round trips behind (a 15^2 level is four dependent trips per phase), sixteen waves overlap theirs.  Kept as a switch
// (HPS_MG_LOWV_WAVE=1), parity-tested by the same multigrid tests, OFF by default.
struct LowMap { int ti, tj, si, sj, lin, nlin; };
template <bool WAVE>
__device__ __forceinline__ LowMap low_map ()
{
    if (WAVE) return LowMap{(int)(threadIdx.x & 15), (int)((threadIdx.x & 63) >> 4), 16, 4, (int)(threadIdx.x & 63), 64};
    return LowMap{(int)(threadIdx.x & 31), (int)(threadIdx.x >> 5), 32, (int)(blockDim.x >> 5), (int)threadIdx.x, (int)blockDim.x};
}
template <bool WAVE>
__device__ __forceinline__ void low_sync ()
{
    if (WAVE) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
}
#define HPS_LOW_FOR_MAP(mp, l, i, j)                                                        \
    for (int j = (l).b.vly + (mp).tj; j <= (l).b.vhy; j += (mp).sj)                         \
        for (int i = (l).b.vlx + (mp).ti; i <= (l).b.vhx; i += (mp).si)
// (whole workgroup, outside the helpers below)
#define HPS_LOW_FOR_VALID(l, i, j)                                                          \
    for (int j = (l).b.vly + (int)(threadIdx.x >> 5); j <= (l).b.vhy; j += (int)(blockDim.x >> 5))   \
        for (int i = (l).b.vlx + (int)(threadIdx.x & 31); i <= (l).b.vhx; i += 32)

__device__ __forceinline__ double low_fac (double f0, int l) { for (int k = 0; k < l; ++k) f0 *= 0.25; return f0; }      // (exact: powers of two)

template <bool CC, bool WAVE>
__device__ void low_sweeps (lds_double* base, const LowLev& l, double facx, double facy, int nsweeps, int n0 = 0, int n1 = 2)
{
    // A half-sweep touches the points of one colour only: the threads are mapped onto THOSE -- a patch 16 half-columns wide
    // (8 in a lone wave), so a 31^2 level is one trip of 496 threads (8 waves: two per SIMD) instead of 961 threads of which
    // every other one sits out (16 waves: four per SIMD, all of them issuing the loop's ~150 instructions)
    const int ti = WAVE ? (int)(threadIdx.x & 7) : (int)(threadIdx.x & 15), tj = WAVE ? (int)((threadIdx.x & 63) >> 3) : (int)(threadIdx.x >> 4);
    const int si = WAVE ? 8 : 16, sj = WAVE ? 8 : (int)(blockDim.x >> 4);
    const LView cinv = lplane(base, l, 7);
    if (!WAVE && !CC && l.b.vhy - l.b.vly < sj && l.b.vhx - l.b.vlx < 2*si) {
        // One trip per thread (every level the node-centred lower V holds): the thread's row is fixed and its point alternates
        // between two columns with the colour -- both points' offsets, coefficients and right-hand sides are set up ONCE, ahead
        // of the half-sweeps, whose bodies are then four neighbour reads, six fp64 operations and a store.  (The stamps had
        // shown the phases bound by their instruction streams: the generic views' index arithmetic and the loops' exec-mask
        // control were most of a phase's 130-250 instructions.)  No wall variants on node-centred levels: walls hold zeros.
        const int j = l.b.vly + tj;
        const bool rowok = j <= l.b.vhy;
        const int jc = rowok ? j : l.b.vly;
        int off[2]; bool ok[2]; double ci[2], r0[2], r1[2];
        const bool two = (n1 - n0) == 2;
        lds_double* ph0 = base + l.off + (3 + n0)*l.cells;
        lds_double* ph1 = base + l.off + (2 + n1)*l.cells;
#pragma unroll
        for (int par = 0; par < 2; ++par) {            // par = is & 1
            const int i = l.b.vlx + ((l.b.vlx + jc + par) & 1) + 2*ti;
            ok[par] = rowok && i <= l.b.vhx;
            const int ic = ok[par] ? i : l.b.vlx;
            off[par] = (ic - l.b.lox) + (jc - l.b.loy)*l.nxb;
            ci[par] = cinv(ic, jc);
            r0[par] = base[l.off + (1 + n0)*l.cells + off[par]];
            r1[par] = base[l.off + n1*l.cells + off[par]];
        }
        const int sy = l.nxb;
        for (int is = 0; is < nsweeps; ++is) {
            const int par = is & 1;
            const int o = par ? off[1] : off[0];
            const bool k = par ? ok[1] : ok[0];
            const double c = par ? ci[1] : ci[0], ra = par ? r0[1] : r0[0], rb = par ? r1[1] : r1[0];
            if (k) {
                {   const lds_double* q = ph0 + o;
                    const double w = q[-1], e = q[1], so = q[-sy], no = q[sy];
                    ph0[o] = (ra - (facx*(w + e) + facy*(so + no)))*c; }
                if (two) {
                    const lds_double* q = ph1 + o;
                    const double w = q[-1], e = q[1], so = q[-sy], no = q[sy];
                    ph1[o] = (rb - (facx*(w + e) + facy*(so + no)))*c; }
            }
            __syncthreads();
        }
        return;
    }
    for (int is = 0; is < nsweeps; ++is) {
        for (int j = l.b.vly + tj; j <= l.b.vhy; j += sj) {
            for (int i = l.b.vlx + ((l.b.vlx + j + is) & 1) + 2*ti; i <= l.b.vhx; i += 2*si) {      // (i + j + is) even
                const double ci = cinv(i, j);
                for (int n = n0; n < n1; ++n) {
                    const LView rhs = lplane(base, l, 1 + n), phi = lplane(base, l, 3 + n);
                    phi(i, j) = (rhs(i, j) - offdiag<CC, false>((const lds_double*)&phi(i, j), l.nxb, i, j, l.b, facx, facy))*ci;
                }
            }
        }
        low_sync<WAVE>();
    }
}

// coefficient of level c from level f (average_down_acoef) and the inverse diagonals of a level
template <bool CC, bool WAVE>
__device__ void low_hier_level (lds_double* base, const LowLev& f, const LowLev& c)
{
    const LowMap mp = low_map<WAVE>();
    const LView fine = lplane(base, f, 0), crse = lplane(base, c, 0);
    // (the walls of every plane hold the zeros the kernel starts from: only unknowns are ever written)
    HPS_LOW_FOR_MAP(mp, c, i, j) crse(i, j) = lrestrict<CC>(fine, i, j);
    low_sync<WAVE>();
}
template <bool CC, bool WAVE>
__device__ void low_cinv_level (lds_double* base, const LowLev& l, double fx, double fy)
{
    const LowMap mp = low_map<WAVE>();
    const LView acf = lplane(base, l, 0), cinv = lplane(base, l, 7);
    HPS_LOW_FOR_MAP(mp, l, i, j) cinv(i, j) = 1.0/diag_c0<CC>(i, j, l.b, acf(i, j), fx, fy);
}

// down-leg of level l: cor = 0, four half-sweeps, residual, its restriction = right-hand side of level c
template <bool CC, bool WAVE>
__device__ void low_down_level (lds_double* base, const LowLev& l, const LowLev& c, double facx, double facy, int n0, int n1)
{
    const LowMap mp = low_map<WAVE>();
    // cor = 0 and the walls of rescor = 0 (the nodal restriction reads them): both still hold the zeros the kernel starts from --
    // a level is visited once per launch -- so neither costs a phase of its own (round 6: 13 of the ~105 barrier-separated
    // phases of a V-cycle at 1023^2)
    low_sweeps<CC, WAVE>(base, l, facx, facy, 4, n0, n1);
    {   // residual -> rescor
        const LView acf = lplane(base, l, 0);
        HPS_LOW_FOR_MAP(mp, l, i, j) {
            const double a = acf(i, j);
            for (int n = n0; n < n1; ++n) {
                const LView rhs = lplane(base, l, 1 + n), phi = lplane(base, l, 3 + n), rc = lplane(base, l, 5 + n);
                rc(i, j) = residual_at<false>((const lds_double*)&phi(i, j), l.nxb, i, j, l.b, rhs(i, j), a, facx, facy);
            }
        }
        low_sync<WAVE>();
    }
    {   // restriction -> res of the next level
        HPS_LOW_FOR_MAP(mp, c, i, j) {
            for (int n = n0; n < n1; ++n) lplane(base, c, 1 + n)(i, j) = lrestrict<CC>(lplane(base, l, 5 + n), i, j);
        }
        low_sync<WAVE>();
    }
}
// up-leg of level l: cor += P(cor of level c), four half-sweeps
template <bool CC, bool WAVE>
__device__ void low_up_level (lds_double* base, const LowLev& l, const LowLev& c, double facx, double facy, int n0, int n1)
{
    const LowMap mp = low_map<WAVE>();
    HPS_LOW_FOR_MAP(mp, l, i, j) {
        for (int n = n0; n < n1; ++n) {
            const LView fine = lplane(base, l, 3 + n);
            fine(i, j) = fine(i, j) + lprolong<CC>(lplane(base, c, 3 + n), i, j);
        }
    }
    low_sync<WAVE>();
    low_sweeps<CC, WAVE>(base, l, facx, facy, 4, n0, n1);
}

// The bottom level of a 2^K - 1 grid holds 3 x 3 unknowns and takes 16+ half-sweeps (HpMultiGrid.cpp:854-1033): as phases of the
// whole workgroup that was 16 x ~1100 clocks, a quarter of the kernel, for nine numbers.  One lane per component does them in
// registers instead: the same expression per point ((rhs - offdiag) * 1/diag, offdiag<CC, false>'s), the same colour order; points
// of one colour do not read one another, so doing them one after the other changes nothing.
template <bool CC, int P0>          // P0 = (vlx + vly) & 1: the colour of the level's first unknown, so that every point's colour is a constant
__device__ __forceinline__ void low_bottom_regs_p (lds_double* base, const LowLev& l, double facx, double facy, int nsweeps, int n0, int n1)
{
    const int nvx = l.b.vhx - l.b.vlx + 1, nvy = l.b.vhy - l.b.vly + 1;      // <= 3 each (caller)
    for (int n = n0 + (int)threadIdx.x; n < n1; n += (int)blockDim.x) {       // (thread 0: component n0; thread 1: the other one without the split)
        const LView rhs = lplane(base, l, 1 + n), phi = lplane(base, l, 3 + n), cinv = lplane(base, l, 7);
        double p[5][5], r[3][3], c[3][3];
        bool ok[3][3];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) p[a][b] = 0.0;                          // ring = walls (0), unknowns start from 0
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                ok[a][b] = a < nvy && b < nvx;
                r[a][b] = ok[a][b] ? rhs(l.b.vlx + b, l.b.vly + a) : 0.0;
                c[a][b] = ok[a][b] ? cinv(l.b.vlx + b, l.b.vly + a) : 0.0;
            }
        // one half-sweep: the points of colour `col` (compile-time), straight-line
        auto half = [&] (auto colc) {
            constexpr int col = decltype(colc)::value;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    if (((a + b + P0 + col) & 1) != 0) continue;                // (i + j + is) even  <=>  (a + b + P0 + is) even
                    const int i = l.b.vlx + b, j = l.b.vly + a;
                    const double w = p[a + 1][b], e = p[a + 1][b + 2], so = p[a][b + 1], no = p[a + 2][b + 1];
                    double lx = facx*(w + e), ly = facy*(so + no);
                    if (CC) {
                        const double fx43 = facx*(4./3.), fy43 = facy*(4./3.);
                        lx = (i == l.b.lox) ? fx43*e : ((i == l.b.hix) ? fx43*w : lx);
                        ly = (j == l.b.loy) ? fy43*no : ((j == l.b.hiy) ? fy43*so : ly);
                    }
                    const double v = (r[a][b] - (lx + ly))*c[a][b];
                    p[a + 1][b + 1] = ok[a][b] ? v : p[a + 1][b + 1];
                }
        };
        int is = 0;
        for (; is + 1 < nsweeps; is += 2) { half(std::integral_constant<int, 0>{}); half(std::integral_constant<int, 1>{}); }
        if (is < nsweeps) half(std::integral_constant<int, 0>{});
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                if (ok[a][b]) phi(l.b.vlx + b, l.b.vly + a) = p[a + 1][b + 1];
    }
}
template <bool CC>
__device__ void low_bottom_regs (lds_double* base, const LowLev& l, double facx, double facy, int nsweeps, int n0, int n1)
{
    if (((l.b.vlx + l.b.vly) & 1) == 0) low_bottom_regs_p<CC, 0>(base, l, facx, facy, nsweeps, n0, n1);
    else low_bottom_regs_p<CC, 1>(base, l, facx, facy, nsweeps, n0, n1);
}
template <bool CC>
__device__ void low_bottom_lane (lds_double* base, const LowLev& l, double facx, double facy, int nsweeps, int n0, int n1)
{
    low_bottom_regs<CC>(base, l, facx, facy, nsweeps, n0, n1);
    __syncthreads();
}

// ---- one wave, batched phases -------------------------------------------------------------------------------------------------
// The lone-wave phases above lose because a phase of several trips is several dependent LDS round trips: the compiler cannot move
// a trip's reads above the previous trip's writes (it does not know that a phase's outputs and inputs are disjoint planes / colours).
// Here every phase has a compile-time number of trips (TJ x TI, predicates for the level's real extent), computes ALL of a lane's
// points into registers first and stores them afterwards: one round trip per phase, as with k_lower_v3's register blocks.
// Patches: 16 x 4 lanes over all points (TI trips of 16 columns, TJ of 4 rows); 8 x 8 lanes over the half-columns of one colour.
template <int TJ, int TI, class F>
__device__ __forceinline__ void wv_all (const LevBox& b, F&& f)            // f(i, j, slot): slot = compile-time-indexable trip number
{
    const int lane = (int)(threadIdx.x & 63), ti = lane & 15, tj = lane >> 4;
#pragma unroll
    for (int m = 0; m < TJ; ++m)
#pragma unroll
        for (int q = 0; q < TI; ++q) {
            const int j = b.vly + tj + 4*m, i = b.vlx + ti + 16*q;
            f(i, j, m*TI + q, j <= b.vhy && i <= b.vhx);
        }
}
template <bool CC, int TJ, int TI>          // TJ: trips of 8 rows, TI: trips of 16 columns (8 half-columns)
__device__ void wv_sweeps (lds_double* base, const LowLev& l, double facx, double facy, int nsweeps, int n0, int n1)
{
    const int lane = (int)(threadIdx.x & 63), ti = lane & 7, tj = lane >> 3;
    const LView cinv = lplane(base, l, 7);
    const bool two = (n1 - n0) == 2;
    const LView rhs0 = lplane(base, l, 1 + n0), phi0 = lplane(base, l, 3 + n0), rhs1 = lplane(base, l, n1), phi1 = lplane(base, l, 2 + n1);
    for (int is = 0; is < nsweeps; ++is) {
        double o0[TJ*TI], o1[TJ*TI];
#pragma unroll
        for (int m = 0; m < TJ; ++m)
#pragma unroll
            for (int q = 0; q < TI; ++q) {
                const int j = l.b.vly + tj + 8*m, i = l.b.vlx + ((l.b.vlx + j + is) & 1) + 2*(ti + 8*q);
                const bool ok = j <= l.b.vhy && i <= l.b.vhx;
                const int ic = ok ? i : l.b.vlx, jc = ok ? j : l.b.vly;
                const double ci = cinv(ic, jc);
                o0[m*TI + q] = (rhs0(ic, jc) - offdiag<CC, false>((const lds_double*)&phi0(ic, jc), l.nxb, ic, jc, l.b, facx, facy))*ci;
                o1[m*TI + q] = two ? (rhs1(ic, jc) - offdiag<CC, false>((const lds_double*)&phi1(ic, jc), l.nxb, ic, jc, l.b, facx, facy))*ci : 0.0;
            }
#pragma unroll
        for (int m = 0; m < TJ; ++m)
#pragma unroll
            for (int q = 0; q < TI; ++q) {
                const int j = l.b.vly + tj + 8*m, i = l.b.vlx + ((l.b.vlx + j + is) & 1) + 2*(ti + 8*q);
                if (j <= l.b.vhy && i <= l.b.vhx) { phi0(i, j) = o0[m*TI + q]; if (two) phi1(i, j) = o1[m*TI + q]; }
            }
        low_sync<true>();
    }
}
template <bool CC, int TJ, int TI>
__device__ void wv_hier_level (lds_double* base, const LowLev& f, const LowLev& c)
{
    const LView fine = lplane(base, f, 0), crse = lplane(base, c, 0);
    double o[TJ*TI];
    wv_all<TJ, TI>(c.b, [&] (int i, int j, int k, bool ok) { o[k] = lrestrict<CC>(fine, ok ? i : c.b.vlx, ok ? j : c.b.vly); });
    wv_all<TJ, TI>(c.b, [&] (int i, int j, int k, bool ok) { if (ok) crse(i, j) = o[k]; });
    low_sync<true>();
}
template <bool CC, int TJ, int TI>
__device__ void wv_cinv_level (lds_double* base, const LowLev& l, double fx, double fy)
{
    const LView acf = lplane(base, l, 0), cinv = lplane(base, l, 7);
    double o[TJ*TI];
    wv_all<TJ, TI>(l.b, [&] (int i, int j, int k, bool ok) { const int ic = ok ? i : l.b.vlx, jc = ok ? j : l.b.vly; o[k] = 1.0/diag_c0<CC>(ic, jc, l.b, acf(ic, jc), fx, fy); });
    wv_all<TJ, TI>(l.b, [&] (int i, int j, int k, bool ok) { if (ok) cinv(i, j) = o[k]; });
}
template <bool CC, int TJ, int TI, int SJ, int SI>
__device__ void wv_down_level (lds_double* base, const LowLev& l, const LowLev& c, double facx, double facy, int n0, int n1)
{
    wv_sweeps<CC, SJ, SI>(base, l, facx, facy, 4, n0, n1);
    const bool two = (n1 - n0) == 2;
    {   // residual -> rescor
        const LView acf = lplane(base, l, 0);
        const LView rhs0 = lplane(base, l, 1 + n0), phi0 = lplane(base, l, 3 + n0), rc0 = lplane(base, l, 5 + n0);
        const LView rhs1 = lplane(base, l, n1), phi1 = lplane(base, l, 2 + n1), rc1 = lplane(base, l, 4 + n1);
        double o0[TJ*TI], o1[TJ*TI];
        wv_all<TJ, TI>(l.b, [&] (int i, int j, int k, bool ok) {
            const int ic = ok ? i : l.b.vlx, jc = ok ? j : l.b.vly;
            const double a = acf(ic, jc);
            o0[k] = residual_at<false>((const lds_double*)&phi0(ic, jc), l.nxb, ic, jc, l.b, rhs0(ic, jc), a, facx, facy);
            o1[k] = two ? residual_at<false>((const lds_double*)&phi1(ic, jc), l.nxb, ic, jc, l.b, rhs1(ic, jc), a, facx, facy) : 0.0; });
        wv_all<TJ, TI>(l.b, [&] (int i, int j, int k, bool ok) { if (ok) { rc0(i, j) = o0[k]; if (two) rc1(i, j) = o1[k]; } });
        low_sync<true>();
    }
    {   // restriction -> res of the next level
        const LView rc0 = lplane(base, l, 5 + n0), rc1 = lplane(base, l, 4 + n1), r0 = lplane(base, c, 1 + n0), r1 = lplane(base, c, n1);
        double o0[TJ*TI], o1[TJ*TI];
        wv_all<TJ, TI>(c.b, [&] (int i, int j, int k, bool ok) {
            const int ic = ok ? i : c.b.vlx, jc = ok ? j : c.b.vly;
            o0[k] = lrestrict<CC>(rc0, ic, jc); o1[k] = two ? lrestrict<CC>(rc1, ic, jc) : 0.0; });
        wv_all<TJ, TI>(c.b, [&] (int i, int j, int k, bool ok) { if (ok) { r0(i, j) = o0[k]; if (two) r1(i, j) = o1[k]; } });
        low_sync<true>();
    }
}
template <bool CC, int TJ, int TI, int SJ, int SI>
__device__ void wv_up_level (lds_double* base, const LowLev& l, const LowLev& c, double facx, double facy, int n0, int n1)
{
    const bool two = (n1 - n0) == 2;
    const LView f0 = lplane(base, l, 3 + n0), f1 = lplane(base, l, 2 + n1), c0 = lplane(base, c, 3 + n0), c1 = lplane(base, c, 2 + n1);
    double o0[TJ*TI], o1[TJ*TI];
    wv_all<TJ, TI>(l.b, [&] (int i, int j, int k, bool ok) {
        const int ic = ok ? i : l.b.vlx, jc = ok ? j : l.b.vly;
        o0[k] = f0(ic, jc) + lprolong<CC>(c0, ic, jc); o1[k] = two ? f1(ic, jc) + lprolong<CC>(c1, ic, jc) : 0.0; });
    wv_all<TJ, TI>(l.b, [&] (int i, int j, int k, bool ok) { if (ok) { f0(i, j) = o0[k]; if (two) f1(i, j) = o1[k]; } });
    low_sync<true>();
    wv_sweeps<CC, SJ, SI>(base, l, facx, facy, 4, n0, n1);
}
// by the level's extent: up to 16 x 16 unknowns batched (trips 4 x 1 over all points, 2 x 1 over a colour); larger ones through the
// unbatched helpers (batched with 8 x 2 trips the kernel spilled 757 registers under its 1024-thread cap)
#define HPS_WV_DISPATCH(l, CALL_SMALL, CALL_BIG) do { if ((l).b.vhx - (l).b.vlx < 16 && (l).b.vhy - (l).b.vly < 16) { CALL_SMALL; } else { CALL_BIG; } } while (0)

constexpr int LOWV_WAVE_CELLS = 17*17;      // levels of at most this many points (walls included) are wave 0's alone

// levels lv[0..nl-1] (finest first): cor[0] = lower-V(res[0]); mirrors the single-block bottom
// solver of the reference (HpMultiGrid.cpp:854-1033), 16+ sweeps on the last level.  Level 0's
// coefficient and residual come from HBM (acf_g, res_g), its correction goes back (cor_g); the
// coefficients of the lower levels are re-derived in LDS (average_down_acoef).
template <bool CC>
__global__ __launch_bounds__(1024)
void k_lower_v (const LowLev* __restrict__ lv, int nl, const double* __restrict__ acf_g, const double* __restrict__ res_g,
                double* __restrict__ cor_g, double facx0, double facy0, int nsweeps_bottom, StopRule sr, FView fine_res = FView{},
                int wave_cells = 0, int bottom_lane = 0)
{
    if (!vcycle_active(sr)) return;
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    lds_double* base = (lds_double*)lds_raw;
    // grid = 2 (HPS_MG_LOWV_SPLIT, round 6): one field component per workgroup -- the components only meet in norms this kernel
    // does not take, so two CUs need no synchronisation and every barrier-separated phase carries half the LDS traffic
    const int n0 = gridDim.x == 2 ? (int)blockIdx.x : 0, n1 = gridDim.x == 2 ? n0 + 1 : 2;
    // nw: first level that wave 0 works alone (nl: none)
    int nw = nl;
    for (int il = nl - 1; il >= 0; --il) if (lv[il].cells <= wave_cells) nw = il;
    MG_STAMP(8);
    {   // every plane of every level starts from zero (walls, corrections, the coefficient planes' rims)
        const LowLev last = lv[nl - 1];
        const int total = last.off + 8*last.cells;
        for (int s = threadIdx.x; s < total; s += blockDim.x) base[s] = 0.0;
        __syncthreads();
    }
    {
        const LowLev l = lv[0];
        if (fine_res.p) {
            // the top level's right-hand side = R(residual of the level above), formed here (no k_restrict launch ahead of this
            // kernel); cells outside the unknowns' box read as 0
            for (int s = threadIdx.x; s < l.cells; s += blockDim.x) base[l.off + s] = acf_g[s];
            HPS_LOW_FOR_VALID(l, i, j) {
                for (int n = n0; n < n1; ++n) lplane(base, l, 1 + n)(i, j) = restrict_at<CC>(fine_res, i, j, n);
            }
        } else
        for (int s = threadIdx.x; s < l.cells; s += blockDim.x) {
            base[l.off + s] = acf_g[s];
            base[l.off + l.cells + s] = res_g[s];
            base[l.off + 2*l.cells + s] = res_g[l.cells + s];
        }
        __syncthreads();
    }
    // (every level's descriptor is copied out of global memory ONCE per use below: handed on as a reference to lv[il] itself, each
    //  phase behind a barrier began with three to four dependent scalar loads of its fields -- a third of a phase's ~900 clocks)
    // coefficient hierarchy and inverse diagonals of the levels the whole workgroup works (one division per point and V-cycle)
    for (int il = 1; il < nw; ++il) { const LowLev f = lv[il - 1], c = lv[il]; low_hier_level<CC, false>(base, f, c); }
    MG_STAMP(9);
    for (int il = 0; il < nw; ++il) { const LowLev l = lv[il]; low_cinv_level<CC, false>(base, l, low_fac(facx0, il), low_fac(facy0, il)); }
    __syncthreads();
    MG_STAMP(10);
    const int nbig_down = nw < nl - 1 ? nw : nl - 1;          // levels 0 .. nbig_down-1 go down (and later up) with the whole workgroup
    for (int il = 0; il < nbig_down; ++il) { const LowLev l = lv[il], c = lv[il + 1]; low_down_level<CC, false>(base, l, c, low_fac(facx0, il), low_fac(facy0, il), n0, n1); }
    MG_STAMP(11);
    if (nw >= nl) {
        const LowLev l = lv[nl - 1];
        if (bottom_lane && l.b.vhx - l.b.vlx < 3 && l.b.vhy - l.b.vly < 3)
            low_bottom_lane<CC>(base, l, low_fac(facx0, nl - 1), low_fac(facy0, nl - 1), nsweeps_bottom, n0, n1);
        else
        low_sweeps<CC, false>(base, l, low_fac(facx0, nl - 1), low_fac(facy0, nl - 1), nsweeps_bottom, n0, n1);
    } else {
        if (threadIdx.x < 64) {
            // wave 0: what is left of the coefficient hierarchy, the down-legs below the workgroup's levels, the bottom, the up-legs
            for (int il = nw > 0 ? nw : 1; il < nl; ++il) { const LowLev f = lv[il - 1], c = lv[il];
                HPS_WV_DISPATCH(c, (wv_hier_level<CC, 4, 1>(base, f, c)), (low_hier_level<CC, true>(base, f, c))); }
            for (int il = nw; il < nl; ++il) { const LowLev l = lv[il];
                HPS_WV_DISPATCH(l, (wv_cinv_level<CC, 4, 1>(base, l, low_fac(facx0, il), low_fac(facy0, il))), (low_cinv_level<CC, true>(base, l, low_fac(facx0, il), low_fac(facy0, il)))); }
            low_sync<true>();
            for (int il = nw; il < nl - 1; ++il) { const LowLev l = lv[il], c = lv[il + 1];
                HPS_WV_DISPATCH(l, (wv_down_level<CC, 4, 1, 2, 1>(base, l, c, low_fac(facx0, il), low_fac(facy0, il), n0, n1)),
                                   (low_down_level<CC, true>(base, l, c, low_fac(facx0, il), low_fac(facy0, il), n0, n1))); }
            {   const LowLev l = lv[nl - 1];
                if (bottom_lane && l.b.vhx - l.b.vlx < 3 && l.b.vhy - l.b.vly < 3) {
                    // (one lane per component: thread 0, and thread 1 without the split; the helper's barrier is the workgroup's: not here)
                    low_bottom_regs<CC>(base, l, low_fac(facx0, nl - 1), low_fac(facy0, nl - 1), nsweeps_bottom, n0, n1);
                    low_sync<true>();
                } else
                HPS_WV_DISPATCH(l, (wv_sweeps<CC, 2, 1>(base, l, low_fac(facx0, nl - 1), low_fac(facy0, nl - 1), nsweeps_bottom, n0, n1)),
                                   (low_sweeps<CC, true>(base, l, low_fac(facx0, nl - 1), low_fac(facy0, nl - 1), nsweeps_bottom, n0, n1))); }
            for (int il = nl - 2; il >= nw; --il) { const LowLev l = lv[il], c = lv[il + 1];
                HPS_WV_DISPATCH(l, (wv_up_level<CC, 4, 1, 2, 1>(base, l, c, low_fac(facx0, il), low_fac(facy0, il), n0, n1)),
                                   (low_up_level<CC, true>(base, l, c, low_fac(facx0, il), low_fac(facy0, il), n0, n1))); }
        }
        __syncthreads();
    }
    MG_STAMP(12);
    for (int il = nbig_down - 1; il >= 0; --il) { const LowLev l = lv[il], c = lv[il + 1]; low_up_level<CC, false>(base, l, c, low_fac(facx0, il), low_fac(facy0, il), n0, n1); }
    MG_STAMP(13);
    {
        const LowLev l = lv[0];
        for (int s = n0*l.cells + threadIdx.x; s < n1*l.cells; s += blockDim.x) cor_g[s] = base[l.off + 3*l.cells + s];
    }
    MG_STAMP(14);
}

// ---------------------------------------------------------------------------------------------
// Cell-centred lower V, register/LDS resident, from the first level with at most 64 x 64 cells.
// Every level is worked in 2 x 2 cell blocks, one thread per block: the block's correction, rhs and
// inverse diagonal sit in registers, so a red-black half-sweep updates two cells with two LDS reads
// each (the other two neighbours are the thread's own cells), the residual's 4-average for the
// restriction and the piecewise-constant prolongation are thread-local, and the 2 x 2 bottom level is
// one lane's registers.  LDS: the correction of every level (two ringed planes, for the neighbours);
// below level A also rhs (2), coefficient and inverse diagonal planes (level A reloads its rhs from
// HBM for the up-leg).  Levels of at most 64 blocks run in wave 0 alone without workgroup barriers.
// The first sweep of every down-leg level starts from cor = 0, where the update is rhs/diag exactly.
constexpr int LOW2_MAXLEV = 8;
struct Low2 { int nl; int total; int nx[LOW2_MAXLEV], ny[LOW2_MAXLEV], off[LOW2_MAXLEV];
              int coff[LOW2_MAXLEV], cbase, ctot; };      // k_lower_v3: the levels' [acf | 1/diag] planes, one contiguous image from cbase

__device__ __forceinline__ LevBox cc_box (int nx, int ny) { return LevBox{0, 0, nx - 1, ny - 1, 0, 0, nx - 1, ny - 1}; }

template <bool WAVE>
__device__ __forceinline__ void lvl_sync ()
{
    // one wave: its DS instructions reach the LDS in program order, so a read issued behind a write sees it without the
    // wave waiting for the write to retire -- only the compiler has to be kept in order (the s_waitcnt that stood here
    // stalled every one of the ~60 phases of the one-wave levels for the write's round trip)
    if (WAVE) asm volatile("" ::: "memory");
    else __syncthreads();
}

// a thread's 2 x 2 block: cells k = 0:(i,j) 1:(i+1,j) 2:(i,j+1) 3:(i+1,j+1); 0 and 3 have colour 0
struct Blk {
    lds_double *c0, *c1;      // the block's cell 0 in the two correction planes
    int pitch;
    bool act;                 // this thread has a block on this level
    bool ok[4];               // cell inside the level (odd sizes)
    double fxm[2], fym[2];    // wall multipliers of the block's two columns / rows
    double v0[4], v1[4];      // correction (both components)
    double r0[4], r1[4], ci[4];
};

// TWO = false: the functions below work on ONE component (planes selected by `cmp` in blk_geometry / the callers):
// the levels that fit a wave are run by two waves, one per component, with no synchronisation between them (the
// two components only meet in the norm) -- a lone wave is bound by its own instruction latencies, so halving the
// stream per phase nearly halves the phase.

// half-sweep of parity P (0: cells 0 and 3, 1: cells 1 and 2) + publication of the new values
template <int P, bool WAVE, bool TWO = true>
__device__ __forceinline__ void blk_sweep (Blk& B)
{
    if (B.act) {
        const int pt = B.pitch;
        if (P == 0) {
            {   const double w0 = B.c0[-1], s0 = B.c0[-pt];
                const double n0 = (B.r0[0] - (B.fxm[0]*(w0 + B.v0[1]) + B.fym[0]*(s0 + B.v0[2])))*B.ci[0];
                if (B.ok[0]) { B.v0[0] = n0; B.c0[0] = n0; }
                if (TWO) {
                    const double w1 = B.c1[-1], s1 = B.c1[-pt];
                    const double n1 = (B.r1[0] - (B.fxm[0]*(w1 + B.v1[1]) + B.fym[0]*(s1 + B.v1[2])))*B.ci[0];
                    if (B.ok[0]) { B.v1[0] = n1; B.c1[0] = n1; }
                } }
            {   const double e0 = B.c0[pt + 2], t0 = B.c0[2*pt + 1];
                const double n0 = (B.r0[3] - (B.fxm[1]*(B.v0[2] + e0) + B.fym[1]*(B.v0[1] + t0)))*B.ci[3];
                if (B.ok[3]) { B.v0[3] = n0; B.c0[pt + 1] = n0; }
                if (TWO) {
                    const double e1 = B.c1[pt + 2], t1 = B.c1[2*pt + 1];
                    const double n1 = (B.r1[3] - (B.fxm[1]*(B.v1[2] + e1) + B.fym[1]*(B.v1[1] + t1)))*B.ci[3];
                    if (B.ok[3]) { B.v1[3] = n1; B.c1[pt + 1] = n1; }
                } }
        } else {
            {   const double e0 = B.c0[2], s0 = B.c0[1 - pt];
                const double n0 = (B.r0[1] - (B.fxm[1]*(B.v0[0] + e0) + B.fym[0]*(s0 + B.v0[3])))*B.ci[1];
                if (B.ok[1]) { B.v0[1] = n0; B.c0[1] = n0; }
                if (TWO) {
                    const double e1 = B.c1[2], s1 = B.c1[1 - pt];
                    const double n1 = (B.r1[1] - (B.fxm[1]*(B.v1[0] + e1) + B.fym[0]*(s1 + B.v1[3])))*B.ci[1];
                    if (B.ok[1]) { B.v1[1] = n1; B.c1[1] = n1; }
                } }
            {   const double w0 = B.c0[pt - 1], t0 = B.c0[2*pt];
                const double n0 = (B.r0[2] - (B.fxm[0]*(w0 + B.v0[3]) + B.fym[1]*(B.v0[0] + t0)))*B.ci[2];
                if (B.ok[2]) { B.v0[2] = n0; B.c0[pt] = n0; }
                if (TWO) {
                    const double w1 = B.c1[pt - 1], t1 = B.c1[2*pt];
                    const double n1 = (B.r1[2] - (B.fxm[0]*(w1 + B.v1[3]) + B.fym[1]*(B.v1[0] + t1)))*B.ci[2];
                    if (B.ok[2]) { B.v1[2] = n1; B.c1[pt] = n1; }
                } }
        }
    }
    lvl_sync<WAVE>();
}

// down-leg sweeps from cor = 0: sweep 0 is rhs/diag on the colour-0 cells, then sweeps 1 .. nsw-1 (nsw even)
template <bool WAVE, bool TWO = true>
__device__ __forceinline__ void blk_down_sweeps (Blk& B, int nsw)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) { B.v0[k] = 0.0; B.v1[k] = 0.0; }
    if (B.act) {
        if (B.ok[0]) { B.v0[0] = B.r0[0]*B.ci[0]; B.c0[0] = B.v0[0]; if (TWO) { B.v1[0] = B.r1[0]*B.ci[0]; B.c1[0] = B.v1[0]; } }
        if (B.ok[3]) { B.v0[3] = B.r0[3]*B.ci[3]; B.c0[B.pitch + 1] = B.v0[3]; if (TWO) { B.v1[3] = B.r1[3]*B.ci[3]; B.c1[B.pitch + 1] = B.v1[3]; } }
    }
    lvl_sync<WAVE>();
    blk_sweep<1, WAVE, TWO>(B);
    for (int s = 2; s < nsw; s += 2) { blk_sweep<0, WAVE, TWO>(B); blk_sweep<1, WAVE, TWO>(B); }
}

// The whole level is this one block (2 x 2 cells or fewer): every neighbour outside the block is the
// zero ring, so all nsw sweeps from cor = 0 run in the lane's registers; published once at the end.
template <bool TWO = true>
__device__ __forceinline__ void blk_single_sweeps (Blk& B, int nsw)
{
    // (every neighbour outside the block is the zero ring: 0 + v == v exactly, so the sums with it are left out -- this is a
    //  chain of 2 nsw dependent half-sweeps in one lane, every instruction of it is on the V-cycle's critical path)
#pragma unroll
    for (int k = 0; k < 4; ++k) { B.v0[k] = 0.0; B.v1[k] = 0.0; }
    if (B.act) {
        for (int s = 0; s < nsw; s += 2) {
            {   const double a0 = (B.r0[0] - (B.fxm[0]*B.v0[1] + B.fym[0]*B.v0[2]))*B.ci[0];
                const double b0 = (B.r0[3] - (B.fxm[1]*B.v0[2] + B.fym[1]*B.v0[1]))*B.ci[3];
                if (B.ok[0]) B.v0[0] = a0;
                if (B.ok[3]) B.v0[3] = b0;
                if (TWO) {
                    const double a1 = (B.r1[0] - (B.fxm[0]*B.v1[1] + B.fym[0]*B.v1[2]))*B.ci[0];
                    const double b1 = (B.r1[3] - (B.fxm[1]*B.v1[2] + B.fym[1]*B.v1[1]))*B.ci[3];
                    if (B.ok[0]) B.v1[0] = a1;
                    if (B.ok[3]) B.v1[3] = b1;
                } }
            {   const double a0 = (B.r0[1] - (B.fxm[1]*B.v0[0] + B.fym[0]*B.v0[3]))*B.ci[1];
                const double b0 = (B.r0[2] - (B.fxm[0]*B.v0[3] + B.fym[1]*B.v0[0]))*B.ci[2];
                if (B.ok[1]) B.v0[1] = a0;
                if (B.ok[2]) B.v0[2] = b0;
                if (TWO) {
                    const double a1 = (B.r1[1] - (B.fxm[1]*B.v1[0] + B.fym[0]*B.v1[3]))*B.ci[1];
                    const double b1 = (B.r1[2] - (B.fxm[0]*B.v1[3] + B.fym[1]*B.v1[0]))*B.ci[2];
                    if (B.ok[1]) B.v1[1] = a1;
                    if (B.ok[2]) B.v1[2] = b1;
                } }
        }
        const int ok[4] = {0, 1, B.pitch, B.pitch + 1};
#pragma unroll
        for (int k = 0; k < 4; ++k) if (B.ok[k]) { B.c0[ok[k]] = B.v0[k]; if (TWO) B.c1[ok[k]] = B.v1[k]; }
    }
}

// residuals of the block's four cells (0 for cells outside the level)
template <bool TWO = true>
__device__ __forceinline__ void blk_residual (const Blk& B, int i, int j, const LevBox& b, const double (&a)[4],
                                              double fx, double fy, double (&q0)[4], double (&q1)[4])
{
    const int pt = B.pitch;
    {   const double w00 = B.c0[-1], s00 = B.c0[-pt], e01 = B.c0[2], s01 = B.c0[1 - pt];
        const double w02 = B.c0[pt - 1], n02 = B.c0[2*pt], e03 = B.c0[pt + 2], n03 = B.c0[2*pt + 1];
        q0[0] = residual_v<false>(B.v0[0], w00, B.v0[1], s00, B.v0[2], i, j, b, B.r0[0], a[0], fx, fy);
        q0[1] = residual_v<false>(B.v0[1], B.v0[0], e01, s01, B.v0[3], i + 1, j, b, B.r0[1], a[1], fx, fy);
        q0[2] = residual_v<false>(B.v0[2], w02, B.v0[3], B.v0[0], n02, i, j + 1, b, B.r0[2], a[2], fx, fy);
        q0[3] = residual_v<false>(B.v0[3], B.v0[2], e03, B.v0[1], n03, i + 1, j + 1, b, B.r0[3], a[3], fx, fy); }
    if (TWO) {
        const double w10 = B.c1[-1], s10 = B.c1[-pt], e11 = B.c1[2], s11 = B.c1[1 - pt];
        const double w12 = B.c1[pt - 1], n12 = B.c1[2*pt], e13 = B.c1[pt + 2], n13 = B.c1[2*pt + 1];
        q1[0] = residual_v<false>(B.v1[0], w10, B.v1[1], s10, B.v1[2], i, j, b, B.r1[0], a[0], fx, fy);
        q1[1] = residual_v<false>(B.v1[1], B.v1[0], e11, s11, B.v1[3], i + 1, j, b, B.r1[1], a[1], fx, fy);
        q1[2] = residual_v<false>(B.v1[2], w12, B.v1[3], B.v1[0], n12, i, j + 1, b, B.r1[2], a[2], fx, fy);
        q1[3] = residual_v<false>(B.v1[3], B.v1[2], e13, B.v1[1], n13, i + 1, j + 1, b, B.r1[3], a[3], fx, fy);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (!B.ok[k]) { q0[k] = 0.0; q1[k] = 0.0; } else if (!TWO) q1[k] = 0.0; }
}

// geometry of thread t's block on a level of nx x ny cells whose planes start at `lev` (pitch nx + 2);
// cmp: the component whose planes become c0 (one-component mode), 0 otherwise
__device__ __forceinline__ void blk_geometry (Blk& B, lds_double* lev, int nx, int ny, int t, double fx, double fy, int& i, int& j,
                                              int cmp = 0)
{
    const int nbx = (nx + 1) >> 1, nby = (ny + 1) >> 1;
    const int lg = (nbx <= 1) ? 0 : 32 - __clz(nbx - 1);       // blocks per row rounded up to a power of two
    const int bi = t & ((1 << lg) - 1), bj = t >> lg;
    B.act = (bi < nbx) && (bj < nby);
    i = 2*bi; j = 2*bj;
    B.pitch = nx + 2;
    const int ps = B.pitch*(ny + 2);
    const int o = B.act ? (j + 1)*B.pitch + i + 1 : B.pitch + 1;
    B.c0 = lev + cmp*ps + o; B.c1 = B.c0 + ps;
    B.ok[0] = B.act; B.ok[1] = B.act && (i + 1 < nx); B.ok[2] = B.act && (j + 1 < ny); B.ok[3] = B.ok[1] && B.ok[2];
    B.fxm[0] = wall_mult<true>(i, 0, nx - 1, fx); B.fxm[1] = wall_mult<true>(i + 1, 0, nx - 1, fx);
    B.fym[0] = wall_mult<true>(j, 0, ny - 1, fy); B.fym[1] = wall_mult<true>(j + 1, 0, ny - 1, fy);
}

__device__ __forceinline__ double lvl_fac (double f0, int l) { for (int k = 0; k < l; ++k) f0 *= 0.25; return f0; }

// levels >= 1 keep six LDS planes: cor0 cor1 | res0 res1 | acf | 1/diag (all ringed, pitch nx + 2).  With c0 = the
// correction plane of component cmp: rhs of cmp = c0 + 2 ps, coefficient = c0 + (4 - cmp) ps, 1/diag = c0 + (5 - cmp) ps.

// down-leg of level l >= 1: cor = GSRB^nsw(0), then (unless last) the restricted residual -> rhs of level l+1
template <bool WAVE, bool TWO = true>
__device__ __forceinline__ void low_down (lds_double* base, const Low2& d, int l, int t, double fx, double fy, int nsw, bool last,
                                          int cmp = 0)
{
    const int nx = d.nx[l], ny = d.ny[l];
    Blk B; int i, j;
    blk_geometry(B, base + d.off[l], nx, ny, t, fx, fy, i, j, cmp);
    const int ps = B.pitch*(ny + 2);
    const lds_double* r = B.c0 + 2*ps;
    const int ok[4] = {0, 1, B.pitch, B.pitch + 1};
    double a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int o = B.ok[k] ? ok[k] : 0;
        B.r0[k] = r[o]; B.r1[k] = TWO ? r[ps + o] : 0.0; a[k] = r[(2 - cmp)*ps + o]; B.ci[k] = r[(3 - cmp)*ps + o];
    }
    if (last && nx <= 2 && ny <= 2) { blk_single_sweeps<TWO>(B, nsw); lvl_sync<WAVE>(); }
    else blk_down_sweeps<WAVE, TWO>(B, nsw);
    if (!last) {
        double q0[4], q1[4];
        blk_residual<TWO>(B, i, j, cc_box(nx, ny), a, fx, fy, q0, q1);
        if (B.act) {
            const int pn = d.nx[l+1] + 2, psn = pn*(d.ny[l+1] + 2);
            lds_double* rn = base + d.off[l+1] + (2 + cmp)*psn + ((j >> 1) + 1)*pn + (i >> 1) + 1;
            rn[0] = 0.25*(q0[0] + q0[1] + q0[2] + q0[3]);
            if (TWO) rn[psn] = 0.25*(q1[0] + q1[1] + q1[2] + q1[3]);
        }
        lvl_sync<WAVE>();
    }
}

// up-leg of level l >= 1: cor += P(cor of level l+1), GSRB^4
template <bool WAVE, bool TWO = true>
__device__ __forceinline__ void low_up (lds_double* base, const Low2& d, int l, int t, double fx, double fy, int cmp = 0)
{
    const int nx = d.nx[l], ny = d.ny[l];
    Blk B; int i, j;
    blk_geometry(B, base + d.off[l], nx, ny, t, fx, fy, i, j, cmp);
    const int ps = B.pitch*(ny + 2);
    const lds_double* r = B.c0 + 2*ps;
    const int ok[4] = {0, 1, B.pitch, B.pitch + 1};
    const int pn = d.nx[l+1] + 2, psn = pn*(d.ny[l+1] + 2);
    const lds_double* kc = base + d.off[l+1] + cmp*psn + ((j >> 1) + 1)*pn + (i >> 1) + 1;
    const double k0 = B.act ? kc[0] : 0.0, k1 = (TWO && B.act) ? kc[psn] : 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int o = B.ok[k] ? ok[k] : 0;
        B.r0[k] = r[o]; B.r1[k] = TWO ? r[ps + o] : 0.0; B.ci[k] = r[(3 - cmp)*ps + o];
        const double c0 = B.c0[o];
        B.v0[k] = B.ok[k] ? c0 + k0 : 0.0;
        if (B.ok[k]) B.c0[o] = B.v0[k];
        B.v1[k] = 0.0;
        if (TWO) {
            const double c1 = B.c1[o];
            B.v1[k] = B.ok[k] ? c1 + k1 : 0.0;
            if (B.ok[k]) B.c1[o] = B.v1[k];
        }
    }
    lvl_sync<WAVE>();
    blk_sweep<0, WAVE, TWO>(B); blk_sweep<1, WAVE, TWO>(B); blk_sweep<0, WAVE, TWO>(B); blk_sweep<1, WAVE, TWO>(B);
}

// Levels of at most 8 x 8 cells inside one wave, one lane per cell (lane t <-> cell (t & 7, t >> 3)), one component
// (cmp) per wave: a lone wave is bound by its own instruction latencies, and a cell per lane is the shortest
// stream per sweep.
__device__ __forceinline__ void tiny_down (lds_double* base, const Low2& d, int l, int t, double fx, double fy, int cmp)
{
    const int nx = d.nx[l], ny = d.ny[l], pitch = nx + 2, ps = pitch*(ny + 2);
    const int i = t & 7, j = t >> 3;
    const bool ok = (i < nx) && (j < ny);
    const int o = ok ? (j + 1)*pitch + i + 1 : pitch + 1;
    lds_double* c0 = base + d.off[l] + cmp*ps + o;
    const double r0 = c0[2*ps], a = c0[(4 - cmp)*ps], ci = c0[(5 - cmp)*ps];
    const double fxm = wall_mult<true>(i, 0, nx - 1, fx), fym = wall_mult<true>(j, 0, ny - 1, fy);
    if (ok && (((i + j) & 1) == 0)) c0[0] = r0*ci;                          // sweep 0 from cor = 0
    lvl_sync<true>();
    for (int s = 1; s < 4; ++s) {
        if (ok && (((i + j + s) & 1) == 0)) c0[0] = (r0 - offdiag_m((const lds_double*)c0, pitch, fxm, fym))*ci;
        lvl_sync<true>();
    }
    const LevBox b = cc_box(nx, ny);
    const double u0 = residual_at<false>((const lds_double*)c0, pitch, i, j, b, r0, a, fx, fy);
    const double q0 = ok ? u0 : 0.0;
    const double b0 = __shfl_down(q0, 1), g0 = __shfl_down(q0, 8), e0 = __shfl_down(q0, 9);
    if (ok && !(i & 1) && !(j & 1)) {
        const int pn = d.nx[l+1] + 2, psn = pn*(d.ny[l+1] + 2);
        base[d.off[l+1] + (2 + cmp)*psn + ((j >> 1) + 1)*pn + (i >> 1) + 1] = 0.25*(q0 + b0 + g0 + e0);
    }
    lvl_sync<true>();
}

__device__ __forceinline__ void tiny_up (lds_double* base, const Low2& d, int l, int t, double fx, double fy, int cmp)
{
    const int nx = d.nx[l], ny = d.ny[l], pitch = nx + 2, ps = pitch*(ny + 2);
    const int i = t & 7, j = t >> 3;
    const bool ok = (i < nx) && (j < ny);
    const int o = ok ? (j + 1)*pitch + i + 1 : pitch + 1;
    lds_double* c0 = base + d.off[l] + cmp*ps + o;
    const double r0 = c0[2*ps], ci = c0[(5 - cmp)*ps];
    const double fxm = wall_mult<true>(i, 0, nx - 1, fx), fym = wall_mult<true>(j, 0, ny - 1, fy);
    if (ok) {
        const int pn = d.nx[l+1] + 2, psn = pn*(d.ny[l+1] + 2);
        c0[0] = c0[0] + base[d.off[l+1] + cmp*psn + ((j >> 1) + 1)*pn + (i >> 1) + 1];
    }
    lvl_sync<true>();
    for (int s = 0; s < 4; ++s) {
        if (ok && (((i + j + s) & 1) == 0)) c0[0] = (r0 - offdiag_m((const lds_double*)c0, pitch, fxm, fym))*ci;
        lvl_sync<true>();
    }
}

__global__ __launch_bounds__(1024)
void k_lower_v2 (const Low2* __restrict__ dp, const double* __restrict__ acf_g, const double* __restrict__ res_g, double* __restrict__ cor_g,
                 double facx0, double facy0, int nsweeps_bottom, StopRule sr)
{
    if (!vcycle_active(sr)) return;
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    lds_double* base = (lds_double*)lds_raw;
    const int t = threadIdx.x;
    const Low2& d = *dp;          // uniform (scalar) loads; a by-value copy indexed by level would live in scratch
    const int nl = d.nl;
    MG_STAMP(8);
    // ---- level A: rhs and coefficient of the thread's block from HBM, inverse diagonals
    const int nxA = d.nx[0], nyA = d.ny[0], cellsA = nxA*nyA;
    const LevBox bA = cc_box(nxA, nyA);
    Blk A; int iA, jA;
    blk_geometry(A, base + d.off[0], nxA, nyA, t, facx0, facy0, iA, jA);
    double aA[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = iA + (k & 1), j = jA + (k >> 1);
        const int g = min(i, nxA - 1) + min(j, nyA - 1)*nxA;
        const double v0 = res_g[g], v1 = res_g[cellsA + g], v2 = acf_g[g];
        A.r0[k] = A.ok[k] ? v0 : 0.0; A.r1[k] = A.ok[k] ? v1 : 0.0; aA[k] = A.ok[k] ? v2 : 0.0;
        A.ci[k] = 1.0/diag_c0<true>(i, j, bA, aA[k], facx0, facy0);
    }
    // only the correction planes need zeros (ring + the cells of the colour the first half-sweep skips); the
    // rhs / coefficient / inverse-diagonal planes are written before they are read
    for (int l = 0; l < nl; ++l) {
        const int n2 = 2*(d.nx[l] + 2)*(d.ny[l] + 2);
        lds_double* c = base + d.off[l];
        for (int s = t; s < n2; s += 1024) c[s] = 0.0;
    }
    __syncthreads();
    // ---- coefficient hierarchy (average_down_acoef) and inverse diagonals of the levels below
    if (nl > 1 && A.act) {
        const int p1 = d.nx[1] + 2, ps1 = p1*(d.ny[1] + 2);
        base[d.off[1] + 4*ps1 + ((jA >> 1) + 1)*p1 + (iA >> 1) + 1] = 0.25*(aA[0] + aA[1] + aA[2] + aA[3]);
    }
    __syncthreads();
    for (int l = 1; l < nl; ++l) {
        const double fx = lvl_fac(facx0, l), fy = lvl_fac(facy0, l);
        const int nx = d.nx[l], ny = d.ny[l];
        Blk B; int i, j;
        blk_geometry(B, base + d.off[l], nx, ny, t, fx, fy, i, j);
        const int ps = B.pitch*(ny + 2);
        lds_double* acf = B.c0 + 4*ps;
        const int ok[4] = {0, 1, B.pitch, B.pitch + 1};
        double a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = B.ok[k] ? acf[ok[k]] : 0.0;
            if (B.ok[k]) acf[ps + ok[k]] = 1.0/diag_c0<true>(i + (k & 1), j + (k >> 1), cc_box(nx, ny), a[k], fx, fy);
        }
        if (l + 1 < nl && B.act) {
            const int pn = d.nx[l+1] + 2, psn = pn*(d.ny[l+1] + 2);
            base[d.off[l+1] + 4*psn + ((j >> 1) + 1)*pn + (i >> 1) + 1] = 0.25*(a[0] + a[1] + a[2] + a[3]);
        }
        __syncthreads();
    }
    MG_STAMP(9);
    // ---- level A down-leg
    const bool bottomA = (nl == 1);
    blk_down_sweeps<false>(A, bottomA ? nsweeps_bottom : 4);
    if (!bottomA) {
        double q0[4], q1[4];
        blk_residual(A, iA, jA, bA, aA, facx0, facy0, q0, q1);
        if (A.act) {
            const int p1 = d.nx[1] + 2, ps1 = p1*(d.ny[1] + 2);
            lds_double* rn = base + d.off[1] + 2*ps1 + ((jA >> 1) + 1)*p1 + (iA >> 1) + 1;
            rn[0] = 0.25*(q0[0] + q0[1] + q0[2] + q0[3]);
            rn[ps1] = 0.25*(q1[0] + q1[1] + q1[2] + q1[3]);
        }
        __syncthreads();
    }
    MG_STAMP(10);
    if (!bottomA) {
        // lw = first level whose blocks (and those of every level below it) fit wave 0
        int lw = 1;
        for (; lw < nl; ++lw) {
            const int nbx = (d.nx[lw] + 1) >> 1, nby = (d.ny[lw] + 1) >> 1;
            int lg = 0; while ((1 << lg) < nbx) ++lg;
            if ((nby << lg) <= 64) break;
        }
        for (int l = 1; l < lw; ++l)
            low_down<false>(base, d, l, t, lvl_fac(facx0, l), lvl_fac(facy0, l), (l == nl - 1) ? nsweeps_bottom : 4, l == nl - 1);
        MG_STAMP(11);
        if (lw < nl) {
            if (t < 128) {
                // waves 0 and 1: one component each, no synchronisation between them
                const int cmp = t >> 6, tl = t & 63;
                for (int l = lw; l < nl; ++l) {
                    MG_STAMP(16 + l);
                    const bool tiny = (d.nx[l] <= 8 && d.ny[l] <= 8 && l < nl - 1);
                    if (tiny) tiny_down(base, d, l, tl, lvl_fac(facx0, l), lvl_fac(facy0, l), cmp);
                    else low_down<true, false>(base, d, l, tl, lvl_fac(facx0, l), lvl_fac(facy0, l), (l == nl - 1) ? nsweeps_bottom : 4, l == nl - 1, cmp);
                }
                for (int l = nl - 2; l >= lw; --l) {
                    MG_STAMP(32 + l);
                    if (d.nx[l] <= 8 && d.ny[l] <= 8) tiny_up(base, d, l, tl, lvl_fac(facx0, l), lvl_fac(facy0, l), cmp);
                    else low_up<true, false>(base, d, l, tl, lvl_fac(facx0, l), lvl_fac(facy0, l), cmp);
                }
            }
            __syncthreads();
        }
        MG_STAMP(12);
        // level A's rhs left the registers after its residual: fetch it again (L2) behind the up-leg below
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = min(iA + (k & 1), nxA - 1) + min(jA + (k >> 1), nyA - 1)*nxA;
            const double v0 = __builtin_nontemporal_load(res_g + g), v1 = __builtin_nontemporal_load(res_g + cellsA + g);
            A.r0[k] = A.ok[k] ? v0 : 0.0; A.r1[k] = A.ok[k] ? v1 : 0.0;
        }
        for (int l = min(lw - 1, nl - 2); l >= 1; --l)
            low_up<false>(base, d, l, t, lvl_fac(facx0, l), lvl_fac(facy0, l));
        // ---- level A up-leg: correction back from LDS, prolongation, 4 sweeps
        const int p1 = d.nx[1] + 2, ps1 = p1*(d.ny[1] + 2);
        const lds_double* kc = base + d.off[1] + ((jA >> 1) + 1)*p1 + (iA >> 1) + 1;
        const double k0 = A.act ? kc[0] : 0.0, k1 = A.act ? kc[ps1] : 0.0;
        const int ok[4] = {0, 1, A.pitch, A.pitch + 1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = A.ok[k] ? ok[k] : 0;
            const double c0 = A.c0[o], c1 = A.c1[o];
            A.v0[k] = A.ok[k] ? c0 + k0 : 0.0; A.v1[k] = A.ok[k] ? c1 + k1 : 0.0;
            if (A.ok[k]) { A.c0[o] = A.v0[k]; A.c1[o] = A.v1[k]; }
        }
        __syncthreads();
        blk_sweep<0, false>(A); blk_sweep<1, false>(A); blk_sweep<0, false>(A); blk_sweep<1, false>(A);
    }
    MG_STAMP(13);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int g = min(iA + (k & 1), nxA - 1) + min(jA + (k >> 1), nyA - 1)*nxA;
        if (A.ok[k]) { cor_g[g] = A.v0[k]; cor_g[cellsA + g] = A.v1[k]; }
    }
    MG_STAMP(14);
}

// ---------------------------------------------------------------------------------------------
// k_lower_v3: the same lower V with ONE field component per workgroup (grid = 2).  The two components of the solve only
// meet in the norm, which the lower V does not take, so two workgroups on two CUs need no synchronisation at all, and
// every barrier phase of the 64 x 64 and 32 x 32 levels carries half the LDS traffic and half the fp64 instruction
// stream of k_lower_v2 (one CU's LDS pipe and VALUs were what bounded those phases: ~1800 cycles per half-sweep).
// LDS per level, single component: level A one ringed plane (cor); below it four: cor | res | acf | 1/diag.
// Level A's inverse diagonals come from k_acf_pyramid (cinv_g: 4 divisions per thread and V-cycle less).  The gate of
// the speculative V-cycle is evaluated AFTER the level-A loads have been issued: a kernel's first dependent read of
// global memory costs ~1.5 us behind a kernel boundary (scripts/ubench/launch_floor.hip), and the two now overlap.

template <bool WAVE>
__device__ __forceinline__ void low_down_s (lds_double* base, const Low2& d, int l, int t, double fx, double fy, int nsw, bool last)
{
    const int nx = d.nx[l], ny = d.ny[l];
    Blk B; int i, j;
    blk_geometry(B, base + d.off[l], nx, ny, t, fx, fy, i, j);
    const int ps = B.pitch*(ny + 2);
    const lds_double* r = B.c0 + ps;
    const lds_double* cf = base + d.coff[l] + (B.c0 - (base + d.off[l]));      // the block's cell 0 in the acf plane
    const int ok[4] = {0, 1, B.pitch, B.pitch + 1};
    double a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int o = B.ok[k] ? ok[k] : 0;
        B.r0[k] = r[o]; B.r1[k] = 0.0; a[k] = cf[o]; B.ci[k] = cf[ps + o];
    }
    if (last && nx <= 2 && ny <= 2) { blk_single_sweeps<false>(B, nsw); lvl_sync<WAVE>(); }
    else blk_down_sweeps<WAVE, false>(B, nsw);
    if (!last) {
        double q0[4], q1[4];
        blk_residual<false>(B, i, j, cc_box(nx, ny), a, fx, fy, q0, q1);
        if (B.act) {
            const int pn = d.nx[l+1] + 2, psn = pn*(d.ny[l+1] + 2);
            base[d.off[l+1] + psn + ((j >> 1) + 1)*pn + (i >> 1) + 1] = 0.25*(q0[0] + q0[1] + q0[2] + q0[3]);
        }
        lvl_sync<WAVE>();
    }
}

template <bool WAVE>
__device__ __forceinline__ void low_up_s (lds_double* base, const Low2& d, int l, int t, double fx, double fy)
{
    const int nx = d.nx[l], ny = d.ny[l];
    Blk B; int i, j;
    blk_geometry(B, base + d.off[l], nx, ny, t, fx, fy, i, j);
    const int ps = B.pitch*(ny + 2);
    const lds_double* r = B.c0 + ps;
    const lds_double* cf = base + d.coff[l] + (B.c0 - (base + d.off[l]));
    const int ok[4] = {0, 1, B.pitch, B.pitch + 1};
    const int pn = d.nx[l+1] + 2;
    const lds_double* kc = base + d.off[l+1] + ((j >> 1) + 1)*pn + (i >> 1) + 1;
    const double k0 = B.act ? kc[0] : 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int o = B.ok[k] ? ok[k] : 0;
        B.r0[k] = r[o]; B.r1[k] = 0.0; B.ci[k] = cf[ps + o];
        const double c0 = B.c0[o];
        B.v0[k] = B.ok[k] ? c0 + k0 : 0.0;
        if (B.ok[k]) B.c0[o] = B.v0[k];
        B.v1[k] = 0.0;
    }
    lvl_sync<WAVE>();
    blk_sweep<0, WAVE, false>(B); blk_sweep<1, WAVE, false>(B); blk_sweep<0, WAVE, false>(B); blk_sweep<1, WAVE, false>(B);
}

// levels of at most 8 x 8 cells in one wave, one lane per cell
__device__ __forceinline__ void tiny_down_s (lds_double* base, const Low2& d, int l, int t, double fx, double fy)
{
    const int nx = d.nx[l], ny = d.ny[l], pitch = nx + 2, ps = pitch*(ny + 2);
    const int i = t & 7, j = t >> 3;
    const bool ok = (i < nx) && (j < ny);
    const int o = ok ? (j + 1)*pitch + i + 1 : pitch + 1;
    lds_double* c0 = base + d.off[l] + o;
    const lds_double* cf = base + d.coff[l] + o;
    const double r0 = c0[ps], a = cf[0], ci = cf[ps];
    const double fxm = wall_mult<true>(i, 0, nx - 1, fx), fym = wall_mult<true>(j, 0, ny - 1, fy);
    if (ok && (((i + j) & 1) == 0)) c0[0] = r0*ci;                          // sweep 0 from cor = 0
    lvl_sync<true>();
    for (int s = 1; s < 4; ++s) {
        if (ok && (((i + j + s) & 1) == 0)) c0[0] = (r0 - offdiag_m((const lds_double*)c0, pitch, fxm, fym))*ci;
        lvl_sync<true>();
    }
    const LevBox b = cc_box(nx, ny);
    const double u0 = residual_at<false>((const lds_double*)c0, pitch, i, j, b, r0, a, fx, fy);
    const double q0 = ok ? u0 : 0.0;
    const double b0 = __shfl_down(q0, 1), g0 = __shfl_down(q0, 8), e0 = __shfl_down(q0, 9);
    if (ok && !(i & 1) && !(j & 1)) {
        const int pn = d.nx[l+1] + 2, psn = pn*(d.ny[l+1] + 2);
        base[d.off[l+1] + psn + ((j >> 1) + 1)*pn + (i >> 1) + 1] = 0.25*(q0 + b0 + g0 + e0);
    }
    lvl_sync<true>();
}

__device__ __forceinline__ void tiny_up_s (lds_double* base, const Low2& d, int l, int t, double fx, double fy)
{
    const int nx = d.nx[l], ny = d.ny[l], pitch = nx + 2, ps = pitch*(ny + 2);
    const int i = t & 7, j = t >> 3;
    const bool ok = (i < nx) && (j < ny);
    const int o = ok ? (j + 1)*pitch + i + 1 : pitch + 1;
    lds_double* c0 = base + d.off[l] + o;
    const double r0 = c0[ps], ci = base[d.coff[l] + ps + o];
    const double fxm = wall_mult<true>(i, 0, nx - 1, fx), fym = wall_mult<true>(j, 0, ny - 1, fy);
    if (ok) {
        const int pn = d.nx[l+1] + 2;
        c0[0] = c0[0] + base[d.off[l+1] + ((j >> 1) + 1)*pn + (i >> 1) + 1];
    }
    lvl_sync<true>();
    for (int s = 0; s < 4; ++s) {
        if (ok && (((i + j + s) & 1) == 0)) c0[0] = (r0 - offdiag_m((const lds_double*)c0, pitch, fxm, fym))*ci;
        lvl_sync<true>();
    }
}

__global__ __launch_bounds__(1024)
void k_lower_v3 (const Low2* __restrict__ dp, const double* __restrict__ acf_g, const double* __restrict__ cinv_g,
                 const double* __restrict__ res_g, double* __restrict__ cor_g, double* __restrict__ coef_g, int nxA, int nyA, int ctot,
                 double facx0, double facy0, int nsweeps_bottom, StopRule sr)
{
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    lds_double* base = (lds_double*)lds_raw;
    const int t = threadIdx.x;
    const Low2& d = *dp;          // (level A's size and the image's length come by value: the first loads must not wait for this read)
    const int nl = d.nl;
    const int cellsA = nxA*nyA;
    res_g += (long)blockIdx.x*cellsA; cor_g += (long)blockIdx.x*cellsA;      // this workgroup's component
    const LevBox bA = cc_box(nxA, nyA);
    Blk A; int iA, jA;
    blk_geometry(A, base, nxA, nyA, t, facx0, facy0, iA, jA);       // (d.off[0] = 0)
    // ---- level A: rhs, coefficient and inverse diagonal of the thread's block from HBM, issued before the gate is read
    double aA[4], rA[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = iA + (k & 1), j = jA + (k >> 1);
        const int g = min(i, nxA - 1) + min(j, nyA - 1)*nxA;
        const double v0 = res_g[g], v2 = acf_g[g], v3 = cinv_g[g];
        rA[k] = A.ok[k] ? v0 : 0.0; aA[k] = A.ok[k] ? v2 : 0.0; A.ci[k] = A.ok[k] ? v3 : 0.0;
        A.r0[k] = rA[k]; A.r1[k] = 0.0;
    }
    // The [acf | 1/diag] planes of the levels below A depend on the coefficient only: the first V-cycle of a solve derives
    // them (average_down_acoef + one division per cell) and leaves the image in coef_g, the later ones copy it in.
    const bool derive = (sr.k <= 0);
    constexpr int NIMG = 4;      // image words per thread (LOW2 levels below 64 x 64: 3264 doubles)
    double img[NIMG];
#pragma unroll
    for (int m = 0; m < NIMG; ++m) { const int q = min(t + 1024*m, ctot - 1); img[m] = derive ? 0.0 : coef_g[q]; }
    const bool active = vcycle_active(sr);
#pragma unroll
    for (int k = 0; k < 4; ++k) { HPS_KEEP(rA[k]); HPS_KEEP(aA[k]); HPS_KEEP(A.ci[k]); }
#pragma unroll
    for (int m = 0; m < NIMG; ++m) HPS_KEEP(img[m]);
    if (!active) return;
    MG_STAMP(8);
    // only the correction planes need zeros (ring + the cells of the colour the first half-sweep skips)
    for (int l = 0; l < nl; ++l) {
        const int n1 = (d.nx[l] + 2)*(d.ny[l] + 2);
        lds_double* c = base + d.off[l];
        for (int s = t; s < n1; s += 1024) c[s] = 0.0;
    }
    if (!derive) {
#pragma unroll
        for (int m = 0; m < NIMG; ++m) if (t + 1024*m < ctot) base[d.cbase + t + 1024*m] = img[m];
        for (int q = t + 1024*NIMG; q < ctot; q += 1024) base[d.cbase + q] = coef_g[q];      // (larger level sets)
    }
    __syncthreads();
    if (derive) {
        // ---- coefficient hierarchy (average_down_acoef) and inverse diagonals of the levels below
        if (nl > 1 && A.act) {
            const int p1 = d.nx[1] + 2;
            base[d.coff[1] + ((jA >> 1) + 1)*p1 + (iA >> 1) + 1] = 0.25*(aA[0] + aA[1] + aA[2] + aA[3]);
        }
        __syncthreads();
        for (int l = 1; l < nl; ++l) {
            const double fx = lvl_fac(facx0, l), fy = lvl_fac(facy0, l);
            const int nx = d.nx[l], ny = d.ny[l];
            Blk B; int i, j;
            blk_geometry(B, base + d.coff[l], nx, ny, t, fx, fy, i, j);     // c0 = the block's cell 0 in the acf plane
            const int ps = B.pitch*(ny + 2);
            lds_double* acf = B.c0;
            const int ok[4] = {0, 1, B.pitch, B.pitch + 1};
            if (B.act) {          // (whole waves have no block on the small levels: they skip the divisions)
                double a[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = B.ok[k] ? acf[ok[k]] : 0.0;
                if (l + 1 < nl) {     // the next level's coefficient first: the divisions below are off the critical path
                    const int pn = d.nx[l+1] + 2;
                    base[d.coff[l+1] + ((j >> 1) + 1)*pn + (i >> 1) + 1] = 0.25*(a[0] + a[1] + a[2] + a[3]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (B.ok[k]) acf[ps + ok[k]] = 1.0/diag_c0<true>(i + (k & 1), j + (k >> 1), cc_box(nx, ny), a[k], fx, fy);
            }
            __syncthreads();
        }
        if (blockIdx.x == 0) for (int q = t; q < ctot; q += 1024) coef_g[q] = base[d.cbase + q];
    }
    MG_STAMP(9);
    // ---- level A down-leg
    const bool bottomA = (nl == 1);
    blk_down_sweeps<false, false>(A, bottomA ? nsweeps_bottom : 4);
    if (!bottomA) {
        double q0[4], q1[4];
        blk_residual<false>(A, iA, jA, bA, aA, facx0, facy0, q0, q1);
        if (A.act) {
            const int p1 = d.nx[1] + 2, ps1 = p1*(d.ny[1] + 2);
            base[d.off[1] + ps1 + ((jA >> 1) + 1)*p1 + (iA >> 1) + 1] = 0.25*(q0[0] + q0[1] + q0[2] + q0[3]);
        }
        __syncthreads();
    }
    MG_STAMP(10);
    if (!bottomA) {
        // lw = first level whose blocks (and those of every level below it) fit wave 0
        int lw = 1;
        for (; lw < nl; ++lw) {
            const int nbx = (d.nx[lw] + 1) >> 1, nby = (d.ny[lw] + 1) >> 1;
            int lg = 0; while ((1 << lg) < nbx) ++lg;
            if ((nby << lg) <= 64) break;
        }
        for (int l = 1; l < lw; ++l)
            low_down_s<false>(base, d, l, t, lvl_fac(facx0, l), lvl_fac(facy0, l), (l == nl - 1) ? nsweeps_bottom : 4, l == nl - 1);
        MG_STAMP(11);
        if (lw < nl) {
            if (t < 64) {
                for (int l = lw; l < nl; ++l) {
                    MG_STAMP(16 + l);
                    const bool tiny = (d.nx[l] <= 8 && d.ny[l] <= 8 && l < nl - 1);
                    if (tiny) tiny_down_s(base, d, l, t, lvl_fac(facx0, l), lvl_fac(facy0, l));
                    else low_down_s<true>(base, d, l, t, lvl_fac(facx0, l), lvl_fac(facy0, l), (l == nl - 1) ? nsweeps_bottom : 4, l == nl - 1);
                }
                for (int l = nl - 2; l >= lw; --l) {
                    MG_STAMP(32 + l);
                    if (d.nx[l] <= 8 && d.ny[l] <= 8) tiny_up_s(base, d, l, t, lvl_fac(facx0, l), lvl_fac(facy0, l));
                    else low_up_s<true>(base, d, l, t, lvl_fac(facx0, l), lvl_fac(facy0, l));
                }
            }
            __syncthreads();
        }
        MG_STAMP(12);
        for (int l = min(lw - 1, nl - 2); l >= 1; --l)
            low_up_s<false>(base, d, l, t, lvl_fac(facx0, l), lvl_fac(facy0, l));
        // ---- level A up-leg: correction back from LDS, prolongation, 4 sweeps
        const int p1 = d.nx[1] + 2;
        const lds_double* kc = base + d.off[1] + ((jA >> 1) + 1)*p1 + (iA >> 1) + 1;
        const double k0 = A.act ? kc[0] : 0.0;
        const int ok[4] = {0, 1, A.pitch, A.pitch + 1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = A.ok[k] ? ok[k] : 0;
            const double c0 = A.c0[o];
            A.v0[k] = A.ok[k] ? c0 + k0 : 0.0;
            if (A.ok[k]) A.c0[o] = A.v0[k];
            A.r0[k] = rA[k];
        }
        __syncthreads();
        blk_sweep<0, false, false>(A); blk_sweep<1, false, false>(A); blk_sweep<0, false, false>(A); blk_sweep<1, false, false>(A);
    }
    MG_STAMP(13);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int g = min(iA + (k & 1), nxA - 1) + min(jA + (k >> 1), nyA - 1)*nxA;
        if (A.ok[k]) cor_g[g] = A.v0[k];
    }
    MG_STAMP(14);
}

// Coefficient pyramid, cell-centred (average_down_acoef, HpMultiGrid.cpp:1640-1700): one launch
// derives levels 1..nlev_out from level 0.  A workgroup owns a 32 x 32 block of level-0 cells;
// thread t holds the level-1 cell (t & 15, t >> 4) of the block and levels 2.. go through LDS.
// The nested 0.25*(((a+b)+c)+d) order of restrict_cc is kept at every level.
struct PyrOut { double* p[5]; int nx[5], ny[5];      // levels 1..5, row pitch = nx
                double* cinv; int cinv_l; double cfx, cfy; };      // inverse diagonals of level 1 + cinv_l (k_lower_v3's level A), or null

__device__ __forceinline__ void acf_pyramid_block (const FView& acf0, int nx0, int ny0, const PyrOut& out, int nlev_out, unsigned long long* zero, int nzero,
                                                   int bx, int by)
{
    // the first launch of a solve: it also clears the norm slots (saves a memset launch)
    if (bx == 0 && by == 0) for (int q = threadIdx.x; q < nzero; q += 256) zero[q] = 0ULL;
    __shared__ double s_a[2][16*16];
    const int t = threadIdx.x;
    const int bi = bx*32, bj = by*32;       // level-0 origin of the block
    int li = t & 15, lj = t >> 4;
    double v = 0.0;
    {
        const int i = bi + 2*li, j = bj + 2*lj;
        if (i + 1 < nx0 && j + 1 < ny0) v = 0.25*(acf0(i, j, 0) + acf0(i+1, j, 0) + acf0(i, j+1, 0) + acf0(i+1, j+1, 0));
        const int ic = (bi >> 1) + li, jc = (bj >> 1) + lj;
        if (ic < out.nx[0] && jc < out.ny[0]) {
            out.p[0][ic + (long)jc*out.nx[0]] = v;
            if (out.cinv && out.cinv_l == 0) out.cinv[ic + (long)jc*out.nx[0]] = 1.0/diag_c0<true>(ic, jc, cc_box(out.nx[0], out.ny[0]), v, out.cfx, out.cfy);
        }
        s_a[0][t] = v;
    }
    int n = 16, cur = 0;
    for (int l = 1; l < nlev_out; ++l) {
        __syncthreads();
        n >>= 1;
        if (t < n*n) {
            li = t % n; lj = t / n;
            const double* f = s_a[cur];
            const int w = 2*n;
            v = 0.25*(f[2*li + 2*lj*w] + f[2*li + 1 + 2*lj*w] + f[2*li + (2*lj + 1)*w] + f[2*li + 1 + (2*lj + 1)*w]);
            const int ic = (bi >> (l + 1)) + li, jc = (bj >> (l + 1)) + lj;
            if (ic < out.nx[l] && jc < out.ny[l]) {
                out.p[l][ic + (long)jc*out.nx[l]] = v;
                if (out.cinv && out.cinv_l == l) out.cinv[ic + (long)jc*out.nx[l]] = 1.0/diag_c0<true>(ic, jc, cc_box(out.nx[l], out.ny[l]), v, out.cfx, out.cfy);
            }
            s_a[cur ^ 1][t] = v;
        }
        cur ^= 1;
    }
}

__global__ __launch_bounds__(256)
void k_acf_pyramid (FView acf0, int nx0, int ny0, PyrOut out, int nlev_out, unsigned long long* zero, int nzero)
{
    acf_pyramid_block(acf0, nx0, ny0, out, nlev_out, zero, nzero, (int)blockIdx.x, (int)blockIdx.y);
}

// the pyramid's blocks (the first gx*gy workgroups) and, behind them, a pass of the engine's slab that is due at the same point of
// a slice: -grad Psi and the Sx / Sy initialisation (slab_ops.h), 256 cells of a padded row per workgroup -- the coefficient
// hierarchy costs the slice no launch of its own
__global__ __launch_bounds__(256)
void k_hierarchy_gradpsi (FView acf0, int nx0, int ny0, PyrOut out, int nlev_out, unsigned long long* zero, int nzero, int gx, int gy,
                          SlabView f, GradPsiSxSy ga, int nbx)
{
    const int b = (int)blockIdx.x;
    if (b < gx*gy) { acf_pyramid_block(acf0, nx0, ny0, out, nlev_out, zero, nzero, b % gx, b / gx); return; }
    const int q = b - gx*gy;
    const int row = q / nbx, bxs = q - row*nbx;
    gradpsi_sxsy_cell(f, ga, bxs*256 + (int)threadIdx.x - f.ng, row - f.ng);
}

// ---- node-centred coefficient hierarchy in one launch (round 6) ------------------------------------------------------------
// average_down_acoef on the 2^K - 1 grids (HpMultiGrid.cpp:1640-1700 with the 9-point full weighting of the nodal levels) was one
// k_restrict launch per level: five dependent 5-us launches per solve at 1023^2 (26 us of the slice).  Here a workgroup owns
// NB x NB nodes of the coarsest level wanted (level np) and everything below them: with s = np - l, level l's nodes
// [T0 2^s, (T0 + NB) 2^s) are its to write, and it needs [T0 2^s - c_l, (T0 + NB) 2^s) with c_l = 2^s - 1 of them (the 3 x 3
// stencils of the nodes above reach one node further down-left per level): a (NB 2^np + 2^np - 1)^2 window of level 0 goes to
// LDS (95^2 for np = 5, NB = 2: 2.2 x the plane's 8 MB, once), every level is formed there with the same expression and
// operand order as restrict_at<false> -- the nodes outside a level's unknowns are never needed by a coarser unknown -- and
// the owned unknowns go to their level's plane.  Fine window index of node 2i - 1 is 2 (i - lo_l): no index arithmetic.
constexpr int NPYR_MAX = 5, NPYR_NB = 2;
struct NodalPyr { FView lev[NPYR_MAX]; int vhx[NPYR_MAX + 1], vhy[NPYR_MAX + 1]; int np; };      // vh*[l]: last unknown of level l (first: 1)

template <int NP>       // (compile-time level count: the window widths are constants, the index divisions multiplications)
__global__ __launch_bounds__(256)
void k_nodal_acf_pyramid (FView f0, NodalPyr o, unsigned long long* zero, int nzero)
{
    extern __shared__ double npyr_lds[];
    constexpr int np = NP;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && blockIdx.y == 0) for (int w = tid; w < nzero; w += 256) zero[w] = 0ULL;      // the solve's norm slots
    const int T0x = blockIdx.x*NPYR_NB, T0y = blockIdx.y*NPYR_NB;
    // level 0 window
    int w = NPYR_NB*(1 << np) + (1 << np) - 1;
    int lox = T0x*(1 << np) - ((1 << np) - 1), loy = T0y*(1 << np) - ((1 << np) - 1);
    double* cur = npyr_lds;
    // (eight loads of a thread in flight before the first LDS store: one load per trip was 35 dependent round trips per workgroup
    //  -- the launch took what the five k_restrict launches it replaces had taken)
    for (int e0 = tid; e0 < w*w; e0 += 8*256) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = min(e0 + 256*u, w*w - 1);
            const int lj = e / w, li = e - lj*w;
            const int i = lox + li, j = loy + lj;
            const bool in = (i >= 1 && i <= o.vhx[0] && j >= 1 && j <= o.vhy[0]);
            const double x = f0(min(max(i, 1), o.vhx[0]), min(max(j, 1), o.vhy[0]), 0);
            v[u] = in ? x : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + 256*u; if (e < w*w) cur[e] = v[u]; }
    }
    __syncthreads();
#pragma unroll
    for (int l = 1; l <= np; ++l) {
        const int s = np - l;
        const int wl = NPYR_NB*(1 << s) + (1 << s) - 1;
        const int lxl = T0x*(1 << s) - ((1 << s) - 1), lyl = T0y*(1 << s) - ((1 << s) - 1);
        const int ox = T0x*(1 << s), oy = T0y*(1 << s);          // first owned node
        double* nxt = cur + w*w;
        const FView& out = o.lev[l - 1];
        for (int e = tid; e < wl*wl; e += 256) {
            const int lj = e / wl, li = e - lj*wl;
            const int i = lxl + li, j = lyl + lj;
            double v = 0.0;
            if (i >= 1 && i <= o.vhx[l] && j >= 1 && j <= o.vhy[l]) {
                const double* f = cur + (2*lj)*w + 2*li;             // node (2i - 1, 2j - 1) of the finer window
                v = (1./16.)*(f[0] + 2.*f[1] + f[2]
                            + 2.*f[w] + 4.*f[w + 1] + 2.*f[w + 2]
                            + f[2*w] + 2.*f[2*w + 1] + f[2*w + 2]);
                if (i >= ox && j >= oy) out(i, j, 0) = v;
            }
            if (l < np) nxt[e] = v;
        }
        __syncthreads();
        cur = nxt; w = wl;
    }
}
static size_t nodal_pyramid_lds (int np)
{
    size_t n = 0;
    for (int l = 0; l < np; ++l) { const int s = np - l; const size_t w = (size_t)NPYR_NB*(1 << s) + (1 << s) - 1; n += w*w; }
    return n*sizeof(double);
}

// inverse diagonals of a cell-centred level (grids whose level A lies below the pyramid kernel's five levels)
__global__ void k_level_cinv (const double* __restrict__ acf, double* __restrict__ cinv, int nx, int ny, double fx, double fy)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i < nx && j < ny) cinv[i + (long)j*nx] = 1.0/diag_c0<true>(i, j, cc_box(nx, ny), acf[i + (long)j*nx], fx, fy);
}

__global__ void k_copy2 (LevBox b, FView dst, FView src)
{
    const int i = b.vlx + blockIdx.x*blockDim.x + threadIdx.x;
    const int j = b.vly + blockIdx.y;
    if (i > b.vhx || j > b.vhy) return;
    dst(i, j, 0) = src(i, j, 0); dst(i, j, 1) = src(i, j, 1);
}

struct MGLevelDev { LevBox b; long cells; double *acf, *res, *cor, *rescor; };

constexpr long LOWV_MAX_CELLS = 34*34;     // levels with at most ~32x32 unknowns run in k_lower_v (LDS resident)

constexpr int MG_GO_WORD = 8;      // int word of the header slot that k_post_norms sets: 1 = the solve is over
struct SolveRun { int enq, nspec, nzeroed, max_iters; double tol_rel, tol_abs; bool cc; };

struct Multigrid {
    bool cc; int nx, ny; double dx, dy;
    bool post_fold = false; unsigned int* d_post_counter = nullptr;     // k_post_norms' work in the last V-cycle's level-0 launch (HPS_MG_POST_FOLD=1; measured: 1474 against 1481 slices/s, off)
    bool hierarchy_ready = false;               // mg_solve1_prepare has enqueued the coefficient hierarchy of the next solve
    bool lowv_bottom_lane = true;                           // k_lower_v: a bottom level of at most 3 x 3 unknowns in one lane's registers (HPS_MG_LOWV_BOTTOM_LANE=0: as phases of the workgroup)
    int lowv_wave_cells = 0;                                // k_lower_v: levels of at most so many points are worked by wave 0 alone (HPS_MG_LOWV_WAVE=1: 17^2; measured SLOWER, see low_map: off)
    bool lowv_split = false; int lowv_threads = 1024;      // k_lower_v: one component per workgroup / threads per workgroup (HPS_MG_LOWV_SPLIT, HPS_MG_LOWV_THREADS)
    bool nodal_pull1 = false;                   // node-centred: level 1's down-leg smoother forms its right-hand side from level 0's residual itself (HPS_MG_NODAL_PULL1=0: a k_restrict launch)
    bool nodal_pyramid = false;                 // node-centred grids: the coefficient hierarchy in one launch (k_nodal_acf_pyramid; HPS_MG_NODAL_PYRAMID=0: off)
    std::vector<MGLevelDev> L;
    int lowv_begin = 1;                         // first level handled by k_lower_v
    LowLev* d_low = nullptr; size_t low_lds = 0;
    // device / pinned buffers: one header slot (MG_NSUB words: a caller's counters ride along with the norm
    // read-back, see hps_mg_rider) followed by the norm slots; d_norms / h_norms point at slot 0
    unsigned long long *d_buf = nullptr, *h_buf = nullptr;
    unsigned long long* h_buf_dev = nullptr;    // device address of the (mapped) pinned buffer: k_post_norms writes it
    long dbg_trips = 0, dbg_solves = 0, dbg_hist[8] = {0,0,0,0,0,0,0,0};
    unsigned long long* h_seq = nullptr; unsigned long long* h_seq_dev = nullptr; unsigned long long seq = 0;
    unsigned long long* d_norms = nullptr;      // [0] residual, [1] rhs
    unsigned long long* h_norms = nullptr;      // pinned copy of d_norms (2 + MG_MAX_VCYCLES slots)
    int last_iters = 1;                         // V-cycles of the previous solve = speculation depth
    SolveRun run{};                             // the solve between mg_solve1_begin and mg_solve1_finish
    bool defer_post = false; bool deferred_valid = false; MgPost deferred{};      // mg_defer_post / mg_take_deferred_post
    bool use_low2 = false; Low2 low2{}; Low2* d_low2 = nullptr; size_t low2_lds = 0;   // cell-centred register/LDS lower V
    bool use_low3 = false; Low2 low3{}; Low2* d_low3 = nullptr; size_t low3_lds = 0;   // ... one component per workgroup (k_lower_v3)
    double* cinvA = nullptr;                    // inverse diagonals of level lowv_begin (written with the coefficient pyramid)
    double* coef_img = nullptr;                 // [acf | 1/diag] planes of the levels below it (left by the first V-cycle of a solve)
    double* tmp0 = nullptr;                     // level-0 scratch: smoothed solution before the last GSRB^4
    bool init_huge = false;                     // HPS_MG_INIT_HUGE=1: the initial level-0 pass on 64 x 48 tiles (measured: 235 against 230 us per solve)
    bool fuse_level0 = true, cor_in_tmp = false; // fused 8-sweep end of the V-cycle; which buffer holds cor[0]
    long small_tile_cells = 300L*300L;          // levels up to this many cells use TileSmall
    long mid_tile_cells = 0;                    // ... up to this many TileMid
    FView sol, rhs, acf0;                       // level-0 user views (set per solve)

    ~Multigrid () {
        for (auto& l : L) { (void)hipFree(l.acf); (void)hipFree(l.res); (void)hipFree(l.cor); (void)hipFree(l.rescor); }
        if (getenv("HPS_MG_DEBUG")) fprintf(stderr, "mg: solves %ld trips %ld hist %ld %ld %ld %ld %ld %ld\n", dbg_solves, dbg_trips, dbg_hist[0], dbg_hist[1], dbg_hist[2], dbg_hist[3], dbg_hist[4], dbg_hist[5]);
        (void)hipFree(d_buf); (void)hipFree(d_post_counter); (void)hipFree(d_low); (void)hipFree(tmp0); (void)hipFree(d_low2); (void)hipFree(d_low3); (void)hipFree(cinvA); (void)hipFree(coef_img);
        if (h_buf) (void)hipHostFree(h_buf);
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
    HPS_REQUIRE((long)(nx + 16)*(ny + 16) < (1L << 27), "hps_mg_create: the kernels address a two-component plane pair by 32-bit byte offsets (at most 2^27 cells per plane)");
    Multigrid* M = new Multigrid;
    M->cc = (nx % 2 == 0); M->nx = nx; M->ny = ny; M->dx = dx; M->dy = dy;
    // level boxes (ctor, HpMultiGrid.cpp:1043-1072)
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
    // (node-centred grids: the split is the default -- Bx/By solve at 1023^2 375 -> 361 us per slice, 512 threads the same, 256 slower:
    //  profiles/r06_lowv_ab.txt; the cell-centred grids run k_lower_v3, which is built that way)
    {   const char* v = getenv("HPS_MG_LOWV_SPLIT"); M->lowv_split = v ? atoi(v) != 0 : !M->cc;
        if (const char* b = getenv("HPS_MG_LOWV_BOTTOM_LANE")) M->lowv_bottom_lane = atoi(b) != 0;
        if (const char* w = getenv("HPS_MG_LOWV_WAVE")) M->lowv_wave_cells = atoi(w) != 0 ? (atoi(w) == 1 ? LOWV_WAVE_CELLS : atoi(w)) : 0;
        const char* t = getenv("HPS_MG_LOWV_THREADS"); if (t) { const int n = atoi(t); if (n == 256 || n == 512 || n == 1024) M->lowv_threads = n; } }
    if (!M->cc) {
        const char* v = getenv("HPS_MG_NODAL_PYRAMID");
        M->nodal_pyramid = !(v && atoi(v) == 0);
        if (M->nodal_pyramid)
            HPS_HIP_CHECK(hipFuncSetAttribute((const void*)k_nodal_acf_pyramid<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nodal_pyramid_lds(NPYR_MAX)));
    }
    if (const char* e = getenv("HPS_MG_SMALL_CELLS")) M->small_tile_cells = atol(e);
    if (const char* e = getenv("HPS_MG_INIT_HUGE")) M->init_huge = atoi(e) != 0;
    if (const char* e = getenv("HPS_MG_POST_FOLD")) M->post_fold = atoi(e) != 0;
    if (const char* e = getenv("HPS_MG_MID_CELLS")) M->mid_tile_cells = atol(e);
    const int nl = M->nlev();
    M->lowv_begin = nl - 1;
    for (int il = nl - 1; il >= 1; --il) if (M->L[il].cells <= LOWV_MAX_CELLS) M->lowv_begin = il;
    if (M->cc && !getenv("HPS_MG_OLD_LOWV")) {
        // first level with at most 64 x 64 cells whose coarser levels all fit 32 x 32
        const int amax = getenv("HPS_MG_LOW2_MAX") ? atoi(getenv("HPS_MG_LOW2_MAX")) : 64;
        for (int il = 1; il < nl; ++il) {
            const LevBox& b = M->L[il].b;
            const bool fits = (b.hix + 1 <= amax) && (b.hiy + 1 <= amax) && (nl - il <= LOW2_MAXLEV)
                           && (il + 1 >= nl || (M->L[il+1].b.hix + 1 <= 32 && M->L[il+1].b.hiy + 1 <= 32));
            if (!fits) continue;
            M->use_low2 = true; M->lowv_begin = il;
            Low2& d = M->low2;
            d.nl = nl - il;
            int off2 = 0;
            for (int k = 0; k < d.nl; ++k) {
                const LevBox& bk = M->L[il + k].b;
                d.nx[k] = bk.hix + 1; d.ny[k] = bk.hiy + 1; d.off[k] = off2;
                off2 += (k == 0 ? 2 : 6)*(d.nx[k] + 2)*(d.ny[k] + 2);
            }
            d.total = off2;
            M->low2_lds = (size_t)off2*sizeof(double);
            if (M->low2_lds > 64*1024)
                HPS_HIP_CHECK(hipFuncSetAttribute((const void*)k_lower_v2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)M->low2_lds));
            HPS_HIP_CHECK(hipMalloc(&M->d_low2, sizeof(Low2)));
            HPS_HIP_CHECK(hipMemcpy(M->d_low2, &d, sizeof(Low2), hipMemcpyHostToDevice));
            if (!getenv("HPS_MG_LOWV2")) {
                // the same levels, one component per workgroup: level A one plane, the others cor | res | acf | 1/diag
                M->use_low3 = true;
                Low2& e = M->low3;
                e = d;
                int off3 = 0;
                for (int k = 0; k < e.nl; ++k) { e.off[k] = off3; off3 += (k == 0 ? 1 : 2)*(e.nx[k] + 2)*(e.ny[k] + 2); }      // cor [| res]
                e.cbase = off3; e.coff[0] = off3;
                for (int k = 1; k < e.nl; ++k) { e.coff[k] = off3; off3 += 2*(e.nx[k] + 2)*(e.ny[k] + 2); }                        // acf | 1/diag
                e.ctot = std::max(off3 - e.cbase, 1);
                e.total = off3;
                HPS_HIP_CHECK(hipMalloc(&M->coef_img, (size_t)e.ctot*sizeof(double)));
                HPS_HIP_CHECK(hipMemset(M->coef_img, 0, (size_t)e.ctot*sizeof(double)));
                M->low3_lds = (size_t)off3*sizeof(double);
                if (M->low3_lds > 64*1024)
                    HPS_HIP_CHECK(hipFuncSetAttribute((const void*)k_lower_v3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)M->low3_lds));
                HPS_HIP_CHECK(hipMalloc(&M->d_low3, sizeof(Low2)));
                HPS_HIP_CHECK(hipMemcpy(M->d_low3, &e, sizeof(Low2), hipMemcpyHostToDevice));
                HPS_HIP_CHECK(hipMalloc(&M->cinvA, (size_t)d.nx[0]*d.ny[0]*sizeof(double)));
            }
            break;
        }
    }
    if (!M->cc) {
        // level 1 pulls (vcycle: pulls) when it is a smoother level of its own (not the lower V's top) and the general switch is on
        const char* g0 = getenv("HPS_MG_NODAL_PULL"); const char* g1 = getenv("HPS_MG_NODAL_PULL1");
        M->nodal_pull1 = !(g0 && atoi(g0) == 0) && !(g1 && atoi(g1) == 0) && M->lowv_begin > 1;
    }
    std::vector<LowLev> low;
    int off = 0;
    for (int il = M->lowv_begin; il < nl; ++il) {
        const MGLevelDev& l = M->L[il];
        low.push_back(LowLev{l.b, l.b.hix - l.b.lox + 1, (int)l.cells, off});
        off += 8*(int)l.cells;
    }
    M->low_lds = (size_t)off*sizeof(double);
    if (!M->use_low2 && M->low_lds > 64*1024) {
        HPS_HIP_CHECK(hipFuncSetAttribute((const void*)k_lower_v<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)M->low_lds));
        HPS_HIP_CHECK(hipFuncSetAttribute((const void*)k_lower_v<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)M->low_lds));
    }
    HPS_HIP_CHECK(hipMalloc(&M->d_low, low.size()*sizeof(LowLev)));
    HPS_HIP_CHECK(hipMemcpy(M->d_low, low.data(), low.size()*sizeof(LowLev), hipMemcpyHostToDevice));
    HPS_HIP_CHECK(hipMalloc(&M->d_buf, (3 + MG_MAX_VCYCLES)*MG_NSUB*sizeof(unsigned long long)));
    HPS_HIP_CHECK(hipMalloc(&M->d_post_counter, 17*sizeof(unsigned int)));
    HPS_HIP_CHECK(hipMemset(M->d_post_counter, 0, 17*sizeof(unsigned int)));
    HPS_HIP_CHECK(hipHostMalloc(&M->h_buf, ((3 + MG_MAX_VCYCLES)*MG_NSUB + 8)*sizeof(unsigned long long), hipHostMallocMapped));
    HPS_HIP_CHECK(hipHostGetDevicePointer((void**)&M->h_buf_dev, M->h_buf, 0));
    M->h_seq = M->h_buf + (3 + MG_MAX_VCYCLES)*MG_NSUB; M->h_seq_dev = M->h_buf_dev + (3 + MG_MAX_VCYCLES)*MG_NSUB;
    *M->h_seq = 0ULL;
    HPS_HIP_CHECK(hipMemset(M->d_buf, 0, (3 + MG_MAX_VCYCLES)*MG_NSUB*sizeof(unsigned long long)));
    memset(M->h_buf, 0, (3 + MG_MAX_VCYCLES)*MG_NSUB*sizeof(unsigned long long));
    M->d_norms = M->d_buf + MG_NSUB; M->h_norms = M->h_buf + MG_NSUB;
    HPS_HIP_CHECK(hipMalloc(&M->tmp0, 2*M->L[0].cells*sizeof(double)));
    HPS_HIP_CHECK(hipMemset(M->tmp0, 0, 2*M->L[0].cells*sizeof(double)));
    *out = M;
    return HPS_OK;
}

template <class TS, bool CC, int SRC, bool DO_RES, int NSW = 4, bool RPULL = false>
static void launch_smooth_ts (Multigrid* M, int il, FView phi_out, FView phi_out2, FView rhs, FView acf, FView phi_in, FView crse,
                              FView res_out, FView cres_out, unsigned long long* resnorm, unsigned long long* rhsnorm,
                              const StopRule& sr, hipStream_t st, const PostArgs* post = nullptr)
{
    const LevBox& b = M->L[il].b;
    constexpr int E = DO_RES ? NSW : NSW - 1;
    constexpr int FX = TS::TX - 2*E, FY = TS::TY - 2*E;
    const int ntx = ceil_div(b.vhx - b.vlx + 1, FX), nty = ceil_div(b.vhy - b.vly + 1, FY);
    const double fac = (double)(1 << il);
    const double ldx = M->dx*fac, ldy = M->dy*fac;
    const double facx = 1.0/(ldx*ldx), facy = 1.0/(ldy*ldy);
    constexpr bool FUSE = CC && DO_RES;
    if constexpr (NSW != 4) {
        if (post) {
            hipLaunchKernelGGL((k_smooth<TS, CC, SRC, DO_RES, FUSE, NSW, true>), dim3(ntx*nty), dim3(TS::NT), 0, st, b, phi_out, phi_out2, rhs, acf,
                               phi_in, crse, res_out, cres_out, facx, facy, ntx, resnorm, rhsnorm, sr, *post);
            return;
        }
    }
    hipLaunchKernelGGL((k_smooth<TS, CC, SRC, DO_RES, FUSE, NSW, false, RPULL>), dim3(ntx*nty), dim3(TS::NT), 0, st, b, phi_out, phi_out2, rhs, acf,
                       phi_in, crse, res_out, cres_out, facx, facy, ntx, resnorm, rhsnorm, sr, PostArgs{});
}

template <bool CC, int SRC, bool DO_RES>
static void launch_smooth (Multigrid* M, int il, FView phi_out, FView phi_out2, FView rhs, FView acf, FView phi_in, FView crse,
                           FView res_out, FView cres_out, unsigned long long* resnorm, unsigned long long* rhsnorm,
                           const StopRule& sr, hipStream_t st)
{
    if (M->L[il].cells <= M->small_tile_cells)
        launch_smooth_ts<TileSmall, CC, SRC, DO_RES>(M, il, phi_out, phi_out2, rhs, acf, phi_in, crse, res_out, cres_out, resnorm, rhsnorm, sr, st);
    else if (M->L[il].cells <= M->mid_tile_cells)
        launch_smooth_ts<TileMid, CC, SRC, DO_RES>(M, il, phi_out, phi_out2, rhs, acf, phi_in, crse, res_out, cres_out, resnorm, rhsnorm, sr, st);
    else
        launch_smooth_ts<TileBig, CC, SRC, DO_RES>(M, il, phi_out, phi_out2, rhs, acf, phi_in, crse, res_out, cres_out, resnorm, rhsnorm, sr, st);
}

template <bool CC>
static void restrict_residual_if_nodal (Multigrid* M, int il, const StopRule& sr, hipStream_t st)
{
    if (CC) return;     // fused into k_smooth
    const LevBox& cb = M->L[il+1].b;
    hipLaunchKernelGGL(k_restrict<CC>, dim3(ceil_div(cb.vhx - cb.vlx + 1, 64), cb.vhy - cb.vly + 1), dim3(64), 0, st,
                       cb, M->lv(il+1, M->L[il+1].res), M->lv(il, M->L[il].rescor), 2, sr);
}

// V-cycle k (vcycle :1429-1512).  On entry res[1] = R(rhs - L(cor[0])); on exit again, plus
// tmp0 = smoothed solution, cor[0] = sol = GSRB^4(tmp0) and the residual norm in d_norms[2+k].
template <bool CC>
static bool vcycle (Multigrid* M, int k, double tol_rel, double tol_abs, hipStream_t st, const PostArgs* post = nullptr)      // returns: the post has ridden along
{
    const int nl = M->nlev();
    const int lb = M->lowv_begin;
    const FView none{};
    const StopRule sr{M->d_norms, k, tol_rel, tol_abs};
    // node-centred levels: the smoother of level il >= 2 forms its right-hand side from level il - 1's residual itself (RPULL);
    // the launch of k_restrict stays where the next consumer is not a smoother (level 0 -> 1 behind the fused pass, the lower V's input)
    static const bool pull = [] { const char* v = std::getenv("HPS_MG_NODAL_PULL"); return !(v && std::atoi(v) == 0); }();
    // (levels on 32 x 16 tiles only: with the nine reads per cell in flight the 64 x 32 variant spills 108 registers under its cap)
    // (round 6: level 1 too -- on 32 x 32 tiles of 256 threads where it is too large for the 32 x 16 ones: the 64 x 32 variant's 512
    //  threads are what capped its registers -- so the level 0 -> 1 restriction launch is gone as well: M->nodal_pull1)
    auto pulls = [&] (int il) { return !CC && pull && ((il >= 2 && ((il < lb && M->L[il].cells <= M->small_tile_cells) ||
                                                                   (il == lb && !M->use_low3 && !M->use_low2))) ||      // (il == lb: k_lower_v's own load)
                                                      (il == 1 && M->nodal_pull1)); };
    for (int il = 1; il < lb; ++il) {
        if (pulls(il) && M->L[il].cells > M->small_tile_cells)
            launch_smooth_ts<TileMid, CC, SRC_ZERO, true, 4, !CC>(M, il, M->lv(il, M->L[il].cor), none, M->lv(il, M->L[il].res), M->lv(il, M->L[il].acf), none,
                                                                  M->lv(il-1, M->L[il-1].rescor), M->lv(il, M->L[il].rescor), M->lv(il+1, M->L[il+1].res), nullptr, nullptr, sr, st);
        else if (pulls(il))
            launch_smooth_ts<TileSmall, CC, SRC_ZERO, true, 4, !CC>(M, il, M->lv(il, M->L[il].cor), none, M->lv(il, M->L[il].res), M->lv(il, M->L[il].acf), none,
                                                                    M->lv(il-1, M->L[il-1].rescor), M->lv(il, M->L[il].rescor), M->lv(il+1, M->L[il+1].res), nullptr, nullptr, sr, st);
        else
        launch_smooth<CC, SRC_ZERO, true>(M, il, M->lv(il, M->L[il].cor), none, M->lv(il, M->L[il].res), M->lv(il, M->L[il].acf), none,
                                          none, M->lv(il, M->L[il].rescor), M->lv(il+1, M->L[il+1].res), nullptr, nullptr, sr, st);
        if (!pulls(il + 1)) restrict_residual_if_nodal<CC>(M, il, sr, st);
    }
    {
        const double fac = (double)(1 << lb);
        const double ldx = M->dx*fac, ldy = M->dy*fac;
        const LevBox& bb = M->L[nl-1].b;
        const int nsweeps = std::max(16, (std::max(bb.hix - bb.lox + 1, bb.hiy - bb.loy + 1) + 1)/2*2);
        if (M->use_low3)
            hipLaunchKernelGGL(k_lower_v3, dim3(2), dim3(1024), M->low3_lds, st, M->d_low3, M->L[lb].acf, M->cinvA, M->L[lb].res, M->L[lb].cor,
                               M->coef_img, M->low3.nx[0], M->low3.ny[0], M->low3.ctot, 1.0/(ldx*ldx), 1.0/(ldy*ldy), nsweeps, sr);
        else if (M->use_low2)
            hipLaunchKernelGGL(k_lower_v2, dim3(1), dim3(1024), M->low2_lds, st, M->d_low2, M->L[lb].acf, M->L[lb].res, M->L[lb].cor,
                               1.0/(ldx*ldx), 1.0/(ldy*ldy), nsweeps, sr);
        else
            hipLaunchKernelGGL(k_lower_v<CC>, dim3(M->lowv_split ? 2 : 1), dim3(M->lowv_threads), M->low_lds, st, M->d_low, nl - lb, M->L[lb].acf, M->L[lb].res,
                               M->L[lb].cor, 1.0/(ldx*ldx), 1.0/(ldy*ldy), nsweeps, sr, pulls(lb) ? M->lv(lb-1, M->L[lb-1].rescor) : FView{}, M->lowv_wave_cells, M->lowv_bottom_lane ? 1 : 0);
    }
    // up-leg: the smoothed correction of level il lands in rescor[il] (out of place)
    for (int il = lb - 1; il >= 1; --il) {
        double* crse = (il + 1 == lb) ? M->L[il+1].cor : M->L[il+1].rescor;
        launch_smooth<CC, SRC_PROLONG, false>(M, il, M->lv(il, M->L[il].rescor), none, M->lv(il, M->L[il].res), M->lv(il, M->L[il].acf),
                                              M->lv(il, M->L[il].cor), M->lv(il+1, crse), none, none, nullptr, nullptr, sr, st);
    }
    {
        // level 0: cor[0] + P(correction), GSRB^4 (the smoothed solution), GSRB^4 + residual (solve_doit's
        // iterate) in one pass: out of place between cor[0] and tmp0 (other tiles read the rims), the
        // iterate also goes to the caller's slab
        double* crse = (1 == lb) ? M->L[1].cor : M->L[1].rescor;
        double* in = M->cor_in_tmp ? M->tmp0 : M->L[0].cor;
        double* out = M->cor_in_tmp ? M->L[0].cor : M->tmp0;
        if (M->fuse_level0) {
            launch_smooth_ts<typename HugeTileOf<CC>::type, CC, SRC_PROLONG, true, 8>(M, 0, M->lv(0, out), M->sol, M->rhs, M->acf0, M->lv(0, in), M->lv(1, crse),
                                                                 M->lv(0, M->L[0].rescor), M->lv(1, M->L[1].res), M->d_norms + (2 + k)*MG_NSUB,
                                                                 nullptr, sr, st, CC ? post : nullptr);
            M->cor_in_tmp = !M->cor_in_tmp;
            if (!M->nodal_pull1) restrict_residual_if_nodal<CC>(M, 0, sr, st);
            return CC && post != nullptr;
        } else {
            launch_smooth<CC, SRC_PROLONG, false>(M, 0, M->lv(0, M->tmp0), none, M->rhs, M->acf0, M->lv(0, M->L[0].cor), M->lv(1, crse),
                                                  none, none, nullptr, nullptr, sr, st);
            launch_smooth<CC, SRC_DIRECT, true>(M, 0, M->lv(0, M->L[0].cor), M->sol, M->rhs, M->acf0, M->lv(0, M->tmp0), none,
                                                M->lv(0, M->L[0].rescor), M->lv(1, M->L[1].res), M->d_norms + (2 + k)*MG_NSUB, nullptr, sr, st);
        }
    }
    if (!M->nodal_pull1) restrict_residual_if_nodal<CC>(M, 0, sr, st);
    return false;
}

// norm slots (and the rider words) -> mapped host memory, sequence number last behind a system-scope fence: the host
// polls it instead of a DMA copy + stream synchronise (one round trip per solve, two when the speculation fell short)
__global__ __launch_bounds__(256)
void k_post_norms (const unsigned long long* __restrict__ src, volatile unsigned long long* dst, int nwords,
                   volatile unsigned long long* seq_slot, unsigned long long seq, int* go_word, StopRule after)
{
    // one word for the kernels enqueued behind this solve before the host has seen its norms (the gated plasma push):
    // is the solve over after the V-cycles enqueued so far?
    {   const bool act = vcycle_active(after);      // (whole waves: the rule is read lane-parallel)
        if (threadIdx.x == 0) *go_word = act ? 0 : 1; }
    for (int w = threadIdx.x; w < nwords; w += blockDim.x) dst[w] = src[w];
    HPS_HOST_STORES_ACKNOWLEDGED();
    __syncthreads();
    if (threadIdx.x == 0) { *seq_slot = seq; }
}

// one batch of (gated) V-cycles and the post of all norms so far to the host
template <bool CC>
static void enqueue_cycles (Multigrid* M, hipStream_t st)
{
    SolveRun& r = M->run;
    bool posted = false;
    for (int v = 0; v < r.nspec && r.enq < r.max_iters; ++v, ++r.enq) {
        if (r.enq >= r.nzeroed) {        // more slots than foreseen: zero the next batch (rare)
            const int more = std::min(r.max_iters - r.nzeroed, 64);
            (void)hipMemsetAsync(M->d_norms + (2 + r.nzeroed)*MG_NSUB, 0, more*MG_NSUB*sizeof(unsigned long long), st);
            r.nzeroed += more;
        }
        const bool last = !(v + 1 < r.nspec && r.enq + 1 < r.max_iters);
        if (last && M->post_fold) {
            ++M->seq; ++M->dbg_trips;
            const PostArgs pa{M->d_buf, (volatile unsigned long long*)M->h_buf_dev, (3 + r.enq + 1)*MG_NSUB, (volatile unsigned long long*)M->h_seq_dev, M->seq,
                              reinterpret_cast<int*>(M->d_buf) + MG_GO_WORD, StopRule{M->d_norms, r.enq + 1, r.tol_rel, r.tol_abs}, M->d_post_counter};
            posted = vcycle<CC>(M, r.enq, r.tol_rel, r.tol_abs, st, &pa);
            if (!posted) { --M->seq; --M->dbg_trips; }
        } else
        vcycle<CC>(M, r.enq, r.tol_rel, r.tol_abs, st);
    }
    if (posted) return;
    ++M->seq; ++M->dbg_trips;
    if (M->defer_post) {
        // the caller's next kernel on this stream posts (mg_take_deferred_post): no launch of its own between the last V-cycle and it
        M->defer_post = false; M->deferred_valid = true;
        M->deferred = MgPost{M->d_buf, (volatile unsigned long long*)M->h_buf_dev, (3 + r.enq)*MG_NSUB, (volatile unsigned long long*)M->h_seq_dev, M->seq,
                             StopRule{M->d_norms, r.enq, r.tol_rel, r.tol_abs}};
        return;
    }
    hipLaunchKernelGGL(k_post_norms, dim3(1), dim3(256), 0, st, M->d_buf, (volatile unsigned long long*)M->h_buf_dev,
                       (3 + r.enq)*MG_NSUB, (volatile unsigned long long*)M->h_seq_dev, M->seq,
                       reinterpret_cast<int*>(M->d_buf) + MG_GO_WORD, StopRule{M->d_norms, r.enq, r.tol_rel, r.tol_abs});
}

// The coefficient hierarchy of a solve (average_down_acoef, HpMultiGrid.cpp:1640-1700; level 0 reads the slab) and the zeroing
// of its norm slots.  It needs the coefficient (chi) only, not the right-hand side: mg_solve1_prepare lets the caller
// enqueue it early, on a stream of its own, beside whatever produces the right-hand side.
struct GuestPass { bool on = false; SlabView f{hps_slab{}}; GradPsiSxSy ga{}; };

template <bool CC>
static int solve1_hierarchy (Multigrid* M, int max_iters, hipStream_t st, const GuestPass* guest = nullptr)
{
    const int lb = M->lowv_begin;
    max_iters = std::min(max_iters, MG_MAX_VCYCLES);
    const StopRule always{nullptr, -1, 0.0, 0.0};
    // speculate as many V-cycles as the previous solve needed; each one is a no-op once converged
    int nspec = std::min(std::max(M->last_iters, 1), std::max(max_iters, 1));
    int nzeroed = std::min(max_iters, nspec + 8);
    const int nzero_words = (2 + nzeroed)*MG_NSUB;
    int first = 1;
    if (CC) {
        PyrOut po{};
        const int np = std::min(lb, 5);
        for (int il = 1; il <= np; ++il) { po.p[il-1] = M->L[il].acf; po.nx[il-1] = M->L[il].b.hix + 1; po.ny[il-1] = M->L[il].b.hiy + 1; }
        const double lfac = (double)(1 << lb), lfx = 1.0/(M->dx*lfac*M->dx*lfac), lfy = 1.0/(M->dy*lfac*M->dy*lfac);
        if (M->use_low3 && lb <= np) { po.cinv = M->cinvA; po.cinv_l = lb - 1; po.cfx = lfx; po.cfy = lfy; }
        if (guest) {
            const int gx = ceil_div(M->nx, 32), gy = ceil_div(M->ny, 32);
            const SlabView& f = guest->f;
            const int nbx = ceil_div(f.js, 256), rows = f.ny + 2*f.ng;
            hipLaunchKernelGGL(k_hierarchy_gradpsi, dim3(gx*gy + nbx*rows), dim3(256), 0, st, M->acf0, M->nx, M->ny, po, np, M->d_norms, nzero_words,
                               gx, gy, f, guest->ga, nbx);
        } else
        hipLaunchKernelGGL(k_acf_pyramid, dim3(ceil_div(M->nx, 32), ceil_div(M->ny, 32)), dim3(256), 0, st, M->acf0, M->nx, M->ny, po, np,
                           M->d_norms, nzero_words);
        first = np + 1;
    } else if (M->nodal_pyramid && lb >= 1) {
        // node-centred: levels 1..min(lb, 5) in one launch, the norm slots zeroed by it too
        NodalPyr o{};
        const int np = std::min(lb, NPYR_MAX);
        o.np = np;
        o.vhx[0] = M->L[0].b.vhx; o.vhy[0] = M->L[0].b.vhy;
        for (int il = 1; il <= np; ++il) { o.lev[il-1] = M->lv(il, M->L[il].acf); o.vhx[il] = M->L[il].b.vhx; o.vhy[il] = M->L[il].b.vhy; }
        const int tx = ceil_div(M->L[np].b.hix, NPYR_NB), ty = ceil_div(M->L[np].b.hiy, NPYR_NB);      // top-level nodes 0..hi - 1 (0: wall)
        switch (np) {
            case 1: hipLaunchKernelGGL(k_nodal_acf_pyramid<1>, dim3(tx, ty), dim3(256), nodal_pyramid_lds(np), st, M->acf0, o, M->d_norms, nzero_words); break;
            case 2: hipLaunchKernelGGL(k_nodal_acf_pyramid<2>, dim3(tx, ty), dim3(256), nodal_pyramid_lds(np), st, M->acf0, o, M->d_norms, nzero_words); break;
            case 3: hipLaunchKernelGGL(k_nodal_acf_pyramid<3>, dim3(tx, ty), dim3(256), nodal_pyramid_lds(np), st, M->acf0, o, M->d_norms, nzero_words); break;
            case 4: hipLaunchKernelGGL(k_nodal_acf_pyramid<4>, dim3(tx, ty), dim3(256), nodal_pyramid_lds(np), st, M->acf0, o, M->d_norms, nzero_words); break;
            default: hipLaunchKernelGGL(k_nodal_acf_pyramid<5>, dim3(tx, ty), dim3(256), nodal_pyramid_lds(np), st, M->acf0, o, M->d_norms, nzero_words); break;
        }
        first = np + 1;
    } else {
        HPS_HIP_CHECK(hipMemsetAsync(M->d_norms, 0, nzero_words*sizeof(unsigned long long), st));
    }
    for (int il = first; il <= lb; ++il) {
        const LevBox& cb = M->L[il].b;
        FView fine = (il == 1) ? M->acf0 : M->lv(il-1, M->L[il-1].acf);
        hipLaunchKernelGGL(k_restrict<CC>, dim3(ceil_div(cb.vhx - cb.vlx + 1, 64), cb.vhy - cb.vly + 1), dim3(64), 0, st,
                           cb, M->lv(il, M->L[il].acf), fine, 1, always);
    }
    if (CC && M->use_low3 && lb > 5) {
        const double lfac = (double)(1 << lb);
        const int lnx = M->L[lb].b.hix + 1, lny = M->L[lb].b.hiy + 1;
        hipLaunchKernelGGL(k_level_cinv, dim3(ceil_div(lnx, 64), lny), dim3(64), 0, st, M->L[lb].acf, M->cinvA, lnx, lny,
                           1.0/(M->dx*lfac*M->dx*lfac), 1.0/(M->dy*lfac*M->dy*lfac));
    }
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

template <bool CC>
static int solve1_begin (Multigrid* M, double tol_rel, double tol_abs, int max_iters, hipStream_t st)
{
    max_iters = std::min(max_iters, MG_MAX_VCYCLES);
    M->cor_in_tmp = false;
    const StopRule always{nullptr, -1, 0.0, 0.0};
    int nspec = std::min(std::max(M->last_iters, 1), std::max(max_iters, 1));
    int nzeroed = std::min(max_iters, nspec + 8);
    if (M->hierarchy_ready) M->hierarchy_ready = false;       // (mg_solve1_prepare has enqueued it; the caller has ordered the streams)
    else if (int e = solve1_hierarchy<CC>(M, max_iters, st)) return e;
    // cor[0] = GSRB^4(sol), residual norm, rhs norm, res[1] = R(residual)  (solve_doit :1319-1346)
    // (optional 64 x 48 tiles: 494 workgroups at 1024^2 instead of 817 -- measured slower, see init_huge)
    if (M->init_huge && M->L[0].cells > M->small_tile_cells)
        launch_smooth_ts<TileHuge, CC, SRC_DIRECT, true>(M, 0, M->lv(0, M->L[0].cor), FView{}, M->rhs, M->acf0, M->sol, FView{}, M->lv(0, M->L[0].rescor),
                                                         M->lv(1, M->L[1].res), M->d_norms, M->d_norms + MG_NSUB, always, st);
    else
    launch_smooth<CC, SRC_DIRECT, true>(M, 0, M->lv(0, M->L[0].cor), FView{}, M->rhs, M->acf0, M->sol, FView{}, M->lv(0, M->L[0].rescor),
                                        M->lv(1, M->L[1].res), M->d_norms, M->d_norms + MG_NSUB, always, st);
    if (!M->nodal_pull1) restrict_residual_if_nodal<CC>(M, 0, always, st);
    M->run = SolveRun{0, nspec, nzeroed, max_iters, tol_rel, tol_abs, CC};
    enqueue_cycles<CC>(M, st);
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

// read the norms back, replay the stopping rule on the host (solve_doit :1352-1398), enqueue one more V-cycle at a time
// while it says so.  *extra: did that happen (then kernels gated on gate_after_enqueued() have not run)?
template <bool CC>
static int solve1_finish (Multigrid* M, int* iters_out, double* resnorm_out, int* extra, hipStream_t st)
{
    SolveRun& r = M->run;
    int status = HPS_OK;
    int iters = 0;
    double last_norm = 0.0;
    bool converged = false, diverged = false, first_trip = true;
    if (extra) *extra = 0;
    auto slot_value = [M] (int slot) {
        unsigned long long m = 0ULL;
        for (int q = 0; q < MG_NSUB; ++q) m = std::max(m, M->h_norms[slot*MG_NSUB + q]);
        double dd; memcpy(&dd, &m, 8); return dd;
    };
    while (true) {
        {   volatile unsigned long long* hs = M->h_seq;
            long spins = 0;
            while (*hs != M->seq) {
                if ((++spins & 0xfffff) == 0 && hipStreamQuery(st) != hipErrorNotReady) {      // a failed launch must not hang us
                    if (*hs == M->seq) break;
                    HPS_HIP_CHECK(hipStreamSynchronize(st));
                    if (*hs != M->seq) { set_error("hps_mg_solve1: the norm read-back never arrived"); return HPS_ERR_HIP; }
                }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE); }
        const double res0 = slot_value(0), rhs0 = slot_value(1);
        const double max_norm = (rhs0 >= res0) ? rhs0 : res0;
        const double target = std::max(r.tol_abs, std::max(r.tol_rel, 1.e-16)*max_norm);
        last_norm = res0; iters = 0;
        converged = (res0 <= target); diverged = false;
        for (int k = 0; k < r.enq && !converged && !diverged; ++k) {
            last_norm = slot_value(2 + k); ++iters;
            if (last_norm <= target) converged = true;
            else if (!(last_norm <= 1.e20*max_norm)) diverged = true;
        }
        if (first_trip) {
            // what the device-side gate of the kernels enqueued behind the first batch has seen (vcycle_active, k = enq)
            const double prev = (r.enq == 0) ? res0 : slot_value(2 + r.enq - 1);
            const bool gate_active = prev > target && prev <= 1.e20*max_norm;
            if (gate_active != !(converged || diverged)) { set_error("hps_mg_solve1: host and device stopping rules disagree"); return HPS_ERR_HIP; }
            if (extra) *extra = gate_active ? 1 : 0;
            first_trip = false;
        }
        if (converged || diverged) break;
        if (r.enq >= r.max_iters) { set_error("hps_mg_solve1: not converged after max_iters V-cycles"); status = HPS_ERR_MG_MAXITER; break; }
        r.nspec = 1;
        enqueue_cycles<CC>(M, st);
    }
    if (diverged) { set_error("hps_mg_solve1: diverging"); status = HPS_ERR_MG_DIVERGED; }
    M->last_iters = std::max(1, iters); ++M->dbg_solves; ++M->dbg_hist[std::min(iters, 7)];
    if (iters == 0) {
        // converged on entry: solution = cor[0] of the initial smoothing (solve_doit :1419-1426)
        const LevBox& b0 = M->L[0].b;
        hipLaunchKernelGGL(k_copy2, dim3(ceil_div(b0.vhx - b0.vlx + 1, 64), b0.vhy - b0.vly + 1), dim3(64), 0, st,
                           b0, M->sol, M->lv(0, M->L[0].cor));
    }
    HPS_HIP_CHECK(hipGetLastError());
    if (iters_out) *iters_out = iters;
    if (resnorm_out) *resnorm_out = last_norm;
    return status;
}

static void set_views (Multigrid* M, const hps_slab& s, int sol_comp, int rhs_comp, int acf_comp)
{
    // centre the slab box on the level-0 box (center_box, HpMultiGrid.H:168-175)
    const int sh = M->cc ? 0 : 1;
    const int o = s.ng - sh;
    M->sol  = FView{s.p + (long)sol_comp*s.nstride, s.jstride, s.nstride, o, o};
    M->rhs  = FView{s.p + (long)rhs_comp*s.nstride, s.jstride, s.nstride, o, o};
    M->acf0 = FView{s.p + (long)acf_comp*s.nstride, s.jstride, s.nstride, o, o};
}

// the solve in two halves: begin enqueues the speculated V-cycles and the post of their norms; finish waits for the
// norms.  Kernels enqueued in between must be gated on the word at mg_gate_after_enqueued (they run iff it is 1: the solve is over).
int mg_solve1_begin (void* handle, hps_slab s, int sol_comp, int rhs_comp, int acf_comp, double tol_rel, double tol_abs,
                     int max_iters, hipStream_t st)
{
    Multigrid* M = static_cast<Multigrid*>(handle);
    HPS_REQUIRE(s.nstride < (1L << 27) && s.ng <= 8, "mg_solve1_begin: planes of at most 2^27 doubles and at most 8 guard cells (32-bit byte offsets in the kernels)");
    set_views(M, s, sol_comp, rhs_comp, acf_comp);
    return M->cc ? solve1_begin<true>(M, tol_rel, tol_abs, max_iters, st) : solve1_begin<false>(M, tol_rel, tol_abs, max_iters, st);
}
// the coefficient hierarchy of the NEXT mg_solve1_begin (same slab, same components, same max_iters) on stream `st`, which
// the caller orders: behind whatever writes the coefficient, ahead of mg_solve1_begin's stream
int mg_solve1_prepare (void* handle, hps_slab s, int sol_comp, int rhs_comp, int acf_comp, int max_iters, hipStream_t st)
{
    Multigrid* M = static_cast<Multigrid*>(handle);
    set_views(M, s, sol_comp, rhs_comp, acf_comp);
    if (int e = M->cc ? solve1_hierarchy<true>(M, max_iters, st) : solve1_hierarchy<false>(M, max_iters, st)) return e;
    M->hierarchy_ready = true;
    return HPS_OK;
}
int mg_solve1_prepare_with (void* handle, hps_slab s, int sol_comp, int rhs_comp, int acf_comp, int max_iters, SlabView f, GradPsiSxSy ga,
                            hipStream_t st, bool* done)
{
    Multigrid* M = static_cast<Multigrid*>(handle);
    *done = false;
    if (!M->cc) return HPS_OK;                       // (the nodal hierarchy has no pyramid kernel)
    set_views(M, s, sol_comp, rhs_comp, acf_comp);
    GuestPass g; g.on = true; g.f = f; g.ga = ga;
    if (int e = solve1_hierarchy<true>(M, max_iters, st, &g)) return e;
    M->hierarchy_ready = true;
    *done = true;
    return HPS_OK;
}
// a hierarchy enqueued by mg_solve1_prepare* that no mg_solve1_begin followed (the caller failed in between) is dropped
void mg_solve1_forget_hierarchy (void* handle) { if (handle) static_cast<Multigrid*>(handle)->hierarchy_ready = false; }
// mg_defer_post ahead of mg_solve1_begin: the post of the first batch's norms is not launched but handed out by
// mg_take_deferred_post -- the caller MUST then enqueue, next on the same stream, a kernel that performs it (k_advance_tiled's
// MgPost argument), or the solve's finish never sees its norms.  (With HPS_MG_POST_FOLD the last V-cycle has posted already:
// mg_take_deferred_post returns false and the caller gates on mg_gate_after_enqueued as before.)
void mg_defer_post (void* handle) { Multigrid* M = static_cast<Multigrid*>(handle); M->defer_post = true; M->deferred_valid = false; }
bool mg_take_deferred_post (void* handle, MgPost* out)
{
    Multigrid* M = static_cast<Multigrid*>(handle);
    M->defer_post = false;
    if (!M->deferred_valid) return false;
    M->deferred_valid = false; *out = M->deferred;
    return true;
}
const int* mg_gate_after_enqueued (void* handle)
{
    return reinterpret_cast<const int*>(static_cast<Multigrid*>(handle)->d_buf) + MG_GO_WORD;
}
// have the norms of the batch enqueued by mg_solve1_begin arrived (mg_solve1_finish would not wait)?
bool mg_solve1_ready (void* handle)
{
    Multigrid* M = static_cast<Multigrid*>(handle);
    return *(volatile unsigned long long*)M->h_seq == M->seq;
}
int mg_solve1_finish (void* handle, int* iters_out, double* resnorm_out, int* extra, hipStream_t st)
{
    Multigrid* M = static_cast<Multigrid*>(handle);
    return M->cc ? solve1_finish<true>(M, iters_out, resnorm_out, extra, st) : solve1_finish<false>(M, iters_out, resnorm_out, extra, st);
}

int mg_solve1 (Multigrid* M, hps_slab s, int sol_comp, int rhs_comp, int acf_comp, double tol_rel, double tol_abs,
               int max_iters, int* iters_out, double* resnorm_out, hipStream_t st)
{
    if (int e = mg_solve1_begin(M, s, sol_comp, rhs_comp, acf_comp, tol_rel, tol_abs, max_iters, st)) return e;
    return mg_solve1_finish(M, iters_out, resnorm_out, nullptr, st);
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
    HPS_REQUIRE(slab.nstride < (1L << 27) && slab.ng <= 8, "hps_mg_solve1: planes of at most 2^27 doubles and at most 8 guard cells (32-bit byte offsets in the kernels)");
    return mg_solve1(M, slab, sol_comp, rhs_comp, acoef_comp, tol_rel, tol_abs, max_iters, iters_host, resnorm_host,
                     (hipStream_t)stream);
}

// hpmg::MultiGrid::solve1 with its own argument list (HpMultiGrid.H:64-66: FArrayBox& sol, FArrayBox const& rhs,
// FArrayBox const& acoef): three separate views -- sol and rhs with two components each, acoef with one -- that need
// not live in one slab.  Each view is centred on the solver's box as center_box does (HpMultiGrid.H:168-175).
extern "C" int hps_mg_solve1_fabs (void* handle, hps_slab sol2, hps_slab rhs2, hps_slab acoef1, double tol_rel, double tol_abs,
                                   int max_iters, int* iters_host, double* resnorm_host, hps_stream stream)
{
    HPS_REQUIRE(handle && sol2.p && rhs2.p && acoef1.p, "hps_mg_solve1_fabs: null argument");
    Multigrid* M = static_cast<Multigrid*>(handle);
    for (const hps_slab* s : {&sol2, &rhs2, &acoef1}) {
        HPS_REQUIRE(s->nx == M->nx && s->ny == M->ny, "hps_mg_solve1_fabs: view size does not match the solver");
        HPS_REQUIRE(M->cc || s->ng >= 1, "hps_mg_solve1_fabs: node-centred solve needs >= 1 guard cell in every view");
    }
    HPS_REQUIRE(sol2.ncomp >= 2 && rhs2.ncomp >= 2 && acoef1.ncomp >= 1, "hps_mg_solve1_fabs: sol and rhs need two components, acoef one");
    const int sh = M->cc ? 0 : 1;
    HPS_REQUIRE(sol2.nstride < (1L << 27) && rhs2.nstride < (1L << 27) && acoef1.nstride < (1L << 27), "hps_mg_solve1_fabs: planes of at most 2^27 doubles (32-bit byte offsets in the kernels)");
    M->sol  = FView{sol2.p, sol2.jstride, sol2.nstride, sol2.ng - sh, sol2.ng - sh};
    M->rhs  = FView{rhs2.p, rhs2.jstride, rhs2.nstride, rhs2.ng - sh, rhs2.ng - sh};
    M->acf0 = FView{acoef1.p, acoef1.jstride, acoef1.nstride, acoef1.ng - sh, acoef1.ng - sh};
    hipStream_t st = (hipStream_t)stream;
    if (int e = M->cc ? solve1_begin<true>(M, tol_rel, tol_abs, max_iters, st) : solve1_begin<false>(M, tol_rel, tol_abs, max_iters, st)) return e;
    return mg_solve1_finish(M, iters_host, resnorm_host, nullptr, st);
}

// debug: first call arms the stamps, later calls read the 16 slots back
// A header slot of the norm buffer for the caller's own device counters: they reach the host with the read-back
// every solve does anyway (the engine's halo-fallback counter uses word 0).
void mg_rider (void* handle, int** dev_words, const int** host_words)
{
    Multigrid* M = static_cast<Multigrid*>(handle);
    *dev_words = reinterpret_cast<int*>(M->d_buf);
    *host_words = reinterpret_cast<const int*>(M->h_buf);
}

extern "C" int hps_mg_debug_stamps (long long* stamps16_host)
{
#ifndef HPS_STAMPS
    (void)stamps16_host; set_error("hps_mg_debug_stamps: the library was built without -DHPS_STAMPS (make stamps)"); return HPS_ERR_UNSUPPORTED;
#else
    static long long* d = nullptr;
    if (!d) {
        HPS_HIP_CHECK(hipMalloc(&d, 48*sizeof(long long)));
        HPS_HIP_CHECK(hipMemset(d, 0, 48*sizeof(long long)));
        HPS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_mg_dbg), &d, sizeof(d)));
        return HPS_OK;
    }
    HPS_HIP_CHECK(hipDeviceSynchronize());
    HPS_HIP_CHECK(hipMemcpy(stamps16_host, d, 48*sizeof(long long), hipMemcpyDeviceToHost));
    return HPS_OK;
#endif
}

extern "C" int hps_mg_destroy (void* handle)
{
    delete static_cast<Multigrid*>(handle);
    return HPS_OK;
}
