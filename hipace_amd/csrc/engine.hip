// engine.hip -- the slice engine: slab bookkeeping kernels + the per-slice schedule of the explicit
// solver, i.e. the device-resident equivalent of Hipace::Evolve / Hipace::SolveOneSlice
// (Hipace.cpp:393-728) restricted to the hot path (one plasma species, fixed-ppc driver beam,
// Dirichlet fields, normalised units, hipace.dt = 0 so the beam is static).
#include "common.h"
#include "engine.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <atomic>
#include <algorithm>
#include <map>
#include <mutex>

// the engines' stream pool (see create_engine_stream)
namespace {
std::mutex g_pool_mutex;
std::map<int, std::vector<hipStream_t>> g_stream_pool;      // device -> streams not in use
std::map<int, bool> g_pool_made;
}

namespace hps {

int deposit_current_tiled (const hps_slab& slab, const hps_plasma& pl, const hps_geom& g, const int comp[6], double charge,
                           double mass, int order, double max_qsa, int can_ionize, int* n_qsa, Tiling* T, int* n_fallback,
                           hipStream_t st, int aabs_comp = -1, const int* tile_flag = nullptr, TailWork tw = TailWork{},
                           const BeamPairWork* beam = nullptr, const int* go = nullptr, bool valid_by_w = false);
int explicit_deposit_tiled (const hps_slab& slab, const hps_plasma& pl, const hps_geom& g, const int cache[4], const int depos[2],
                            double charge, double mass, int order, int dtype, int can_ionize, Tiling* T, int* n_fallback,
                            hipStream_t st, int aabs_comp = -1, const int* tile_flag = nullptr, TailWork tw = TailWork{}, bool valid_by_w = false);
int advance_plasma_tiled (const hps_slab& slab, const hps_plasma& pl, const hps_geom& g, const int comp[5], double charge,
                          double mass, int order, int temp_slice, int n_subcycles, int can_ionize, Tiling* T,
                          int* n_fallback, hipStream_t st, int aabs_comp = -1, const IonArgs* ion = nullptr, const int* go = nullptr,
                          TailWork tw = TailWork{}, const MgPost* post = nullptr);
int advance_deposit_tiled (const hps_slab& slab, const hps_plasma& pl, const hps_geom& g, const int comp[5], const int dep_comp[6],
                           double charge, double mass, int order, int n_subcycles, double max_qsa, int* n_qsa, Tiling* T,
                           int* n_fallback, hipStream_t st);

static thread_local std::string g_err;
void set_error (const std::string& msg) { g_err = msg; }

// ------------------------------------------------------------------------------------------
// slab kernels (all over whole planes incl. guards unless noted; x fastest, fully coalesced)
// ------------------------------------------------------------------------------------------
struct CompList { int n; int c[12]; };


// zero the components of `full` everywhere and those of `boxed` inside the box
__global__ __launch_bounds__(256)
void k_zero_comps (double* p, long ns, long plane, int js, CompList full, CompList boxed, CellBox bb, CompList from_ion = CompList{0, {}}, int c_ion = -1)
{
    const long s = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (s >= plane) return;
    for (int k = 0; k < full.n; ++k) p[full.c[k]*ns + s] = 0.0;
    // AddRhoIons (fields/Fields.cpp:606-615) ahead of the deposition instead of behind it: the charge planes start from the
    // neutralising background and the plasma is added on top (one pass over them less per slice)
    if (c_ion >= 0) { const double ion = p[c_ion*ns + s]; for (int k = 0; k < from_ion.n; ++k) p[from_ion.c[k]*ns + s] = ion; }
    const int j = (int)(s / js), i = (int)(s - (long)j*js);
    if (i >= bb.ilo && i <= bb.ihi && j >= bb.jlo && j <= bb.jhi)
        for (int k = 0; k < boxed.n; ++k) p[boxed.c[k]*ns + s] = 0.0;
}

__global__ __launch_bounds__(256)
void k_copy_comps (double* p, long ns, long plane, CompList dst, CompList src)
{
    const long s = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (s >= plane) return;
    double v[12];
    for (int k = 0; k < src.n; ++k) v[k] = p[src.c[k]*ns + s];
    for (int k = 0; k < dst.n; ++k) p[dst.c[k]*ns + s] = v[k];
}

// ShiftSlices (fields/Fields.cpp:588-604): Previous <- This <- Next for the beam currents, jx, jy <- Next.
// The beam planes are zero outside the box: there only jx = jy = 0 is written.
__global__ __launch_bounds__(256)
void k_shift_slices (double* p, long ns, long plane, int js, CellBox bb)
{
    const long s = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (s >= plane) return;
    const int j = (int)(s / js), i = (int)(s - (long)j*js);
    double njx = 0.0, njy = 0.0;
    if (i >= bb.ilo && i <= bb.ihi && j >= bb.jlo && j <= bb.jhi) {
        const double tjx = p[HPS_C_JXB*ns + s], tjy = p[HPS_C_JYB*ns + s];
        njx = p[HPS_C_N_JXB*ns + s]; njy = p[HPS_C_N_JYB*ns + s];
        p[HPS_C_P_JXB*ns + s] = tjx; p[HPS_C_P_JYB*ns + s] = tjy;
        p[HPS_C_JXB*ns + s] = njx;   p[HPS_C_JYB*ns + s] = njy;
    }
    p[HPS_C_JX*ns + s] = njx; p[HPS_C_JY*ns + s] = njy;
}

// ShiftSlices (fields/Fields.cpp:588-604) of this slice AND InitializeSlices (:535-586) of the next one in one pass: what
// the fused push + deposition (k_advance_deposit_tiled) needs done before it deposits into the next slice's jx jy chi
// rhomjz [rho].  Inside the beam's box also jz_beam and the Next beam currents are cleared.
__global__ __launch_bounds__(256)
void k_shift_zero (double* p, long ns, long plane, int js, CellBox bb, int c_rho, int c_ion = -1)
{
    const long s = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (s >= plane) return;
    const int j = (int)(s / js), i = (int)(s - (long)j*js);
    double njx = 0.0, njy = 0.0;
    if (i >= bb.ilo && i <= bb.ihi && j >= bb.jlo && j <= bb.jhi) {
        const double tjx = p[HPS_C_JXB*ns + s], tjy = p[HPS_C_JYB*ns + s];
        njx = p[HPS_C_N_JXB*ns + s]; njy = p[HPS_C_N_JYB*ns + s];
        p[HPS_C_P_JXB*ns + s] = tjx; p[HPS_C_P_JYB*ns + s] = tjy;
        p[HPS_C_JXB*ns + s] = njx;   p[HPS_C_JYB*ns + s] = njy;
        p[HPS_C_N_JXB*ns + s] = 0.0; p[HPS_C_N_JYB*ns + s] = 0.0; p[HPS_C_JZB*ns + s] = 0.0;
    }
    p[HPS_C_JX*ns + s] = njx; p[HPS_C_JY*ns + s] = njy;
    const double ion = c_ion >= 0 ? p[c_ion*ns + s] : 0.0;      // AddRhoIons ahead of the deposition (see k_zero_comps)
    p[HPS_C_CHI*ns + s] = 0.0; p[HPS_C_RHOMJZ*ns + s] = ion;
    if (c_rho >= 0) p[c_rho*ns + s] = ion;
}

// AddRhoIons (fields/Fields.cpp:606-615) fused with the Psi source  -rhomjz/ep0  (:887-888)
__global__ __launch_bounds__(256)
void k_add_ions_psi_rhs (SlabView f, int c_rhomjz, int c_ion, int c_rho, double inv_ep0, double* staging)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x - f.ng;
    const int j = blockIdx.y - f.ng;
    if (i >= f.nx + f.ng) return;
    const long o = f.off(i, j);
    const double ion = f.p[c_ion*f.ns + o];
    const double r = f.p[c_rhomjz*f.ns + o] + ion;
    f.p[c_rhomjz*f.ns + o] = r;
    if (c_rho >= 0) f.p[c_rho*f.ns + o] += ion;
    if (i >= 0 && i < f.nx && j >= 0 && j < f.ny) staging[(long)j*f.nx + i] = -inv_ep0*r;
}

// AddRhoIons + the three Poisson sources of a slice in one pass (fields/Fields.cpp:606-615,
// 887-912): st[0] = -rhomjz/ep0 (Psi), st[1] = (d_x jx + d_y jy)/(ep0 c) (Ez),
// st[2] = mu0 (d_y jx - d_x jy) (Bz); `plane` = nx*ny
__global__ __launch_bounds__(256)
void k_rhs_all (SlabView f, int c_rhomjz, int c_ion, int c_rho, int c_jx, int c_jy, double inv_ep0,
                double fez_x, double fez_y, double fbz_y, double fbz_x, double* staging, long plane)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x - f.ng;
    const int j = blockIdx.y - f.ng;
    if (i >= f.nx + f.ng) return;
    const long o = f.off(i, j);
    double r;
    if (c_ion >= 0) {
        const double ion = f.p[c_ion*f.ns + o];
        r = f.p[c_rhomjz*f.ns + o] + ion;
        f.p[c_rhomjz*f.ns + o] = r;
        if (c_rho >= 0) f.p[c_rho*f.ns + o] += ion;
    } else {      // the background is in already (k_zero_comps / k_shift_zero): valid cells only
        if (!(i >= 0 && i < f.nx && j >= 0 && j < f.ny)) return;
        r = f.p[c_rhomjz*f.ns + o];
    }
    if (i >= 0 && i < f.nx && j >= 0 && j < f.ny) {
        const long so = (long)j*f.nx + i;
        const double* X = f.p + c_jx*f.ns + o;
        const double* Y = f.p + c_jy*f.ns + o;
        staging[so] = -inv_ep0*r;
        staging[plane + so] = fez_x*(X[1] - X[-1]) + fez_y*(Y[f.js] - Y[-f.js]);
        staging[2*plane + so] = fbz_y*(X[f.js] - X[-f.js]) + fbz_x*(Y[1] - Y[-1]);
    }
}

// staging = fa * d(A)/d(da) + fb * d(B)/d(db), centred differences (LinCombination + derivative,
// fields/Fields.cpp:223-249,368-387); dir 0 = x, 1 = y
__global__ __launch_bounds__(256)
void k_rhs_lincomb (SlabView f, int cA, int dirA, double fa, int cB, int dirB, double fb, double* staging)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= f.nx) return;
    const long o = f.off(i, j);
    const long sa = dirA == 0 ? 1 : f.js, sb = dirB == 0 ? 1 : f.js;
    const double* A = f.p + cA*f.ns + o;
    const double* B = f.p + cB*f.ns + o;
    staging[(long)j*f.nx + i] = fa*(A[sa] - A[-sa]) + fb*(B[sb] - B[-sb]);
}

// GridCurrent::DepositCurrentSlice (utils/GridCurrent.cpp:25-71): valid cells; amp = peak * exp(-dz^2/2) of the slice
__global__ __launch_bounds__(256)
void k_grid_current (SlabView f, int cjz, double xlo, double ylo, double dx, double dy, double mx, double my, double sx, double sy,
                     double amp)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= f.nx) return;
    const double delta_x = (xlo + (i + 0.5)*dx - mx) / sx;
    const double delta_y = (ylo + (j + 0.5)*dy - my) / sy;
    f(i, j, cjz) += amp*exp(-0.5*(delta_x*delta_x + delta_y*delta_y));
}

// ExmBy = -dPsi/dx, EypBx = -dPsi/dy on the box grown by (guards-1) (fields/Fields.cpp:931-956)
__global__ __launch_bounds__(256)
void k_grad_psi (SlabView f, int cPsi, int cExmBy, int cEypBx, double hdx_inv, double hdy_inv)
{
    const int gg = f.ng - 1;
    const int i = blockIdx.x*blockDim.x + threadIdx.x - gg;
    const int j = blockIdx.y - gg;
    if (i >= f.nx + gg) return;
    const long o = f.off(i, j);
    const double* P = f.p + cPsi*f.ns + o;
    f.p[cExmBy*f.ns + o] = -(P[1] - P[-1])*hdx_inv;
    f.p[cEypBx*f.ns + o] = -(P[f.js] - P[-f.js])*hdy_inv;
}

// beam contribution to the Bx/By sources (Hipace::InitializeSxSyWithBeam, Hipace.cpp:744-790)
__global__ __launch_bounds__(256)
void k_sxsy_beam (SlabView f, int cSx, int cSy, int cJzb, int cNx, int cNy, int cPx, int cPy,
                  double mu0, double dx2, double dy2, double dz2, CellBox bb)
{
    // whole plane: the guard cells are set to 0 here (they are not part of InitializeSlices' zero list any more)
    const int i = blockIdx.x*blockDim.x + threadIdx.x - f.ng;
    const int j = blockIdx.y - f.ng;
    if (i >= f.nx + f.ng) return;
    const long o = f.off(i, j);
    // no beam current within reach (bb is in padded-array cells): the sources are 0 without a load
    const int ia = i + f.ng, ja = j + f.ng;
    if (i < 0 || i >= f.nx || j < 0 || j >= f.ny || ia < bb.ilo || ia > bb.ihi || ja < bb.jlo || ja > bb.jhi) {
        f.p[cSy*f.ns + o] = 0.0; f.p[cSx*f.ns + o] = 0.0; return;
    }
    const double* J = f.p + cJzb*f.ns + o;
    const double dx_jzb = (J[1] - J[-1])/dx2;
    const double dy_jzb = (J[f.js] - J[-f.js])/dy2;
    const double dz_jxb = (f.p[cPx*f.ns + o] - f.p[cNx*f.ns + o])/dz2;
    const double dz_jyb = (f.p[cPy*f.ns + o] - f.p[cNy*f.ns + o])/dz2;
    f.p[cSy*f.ns + o] =  mu0*(-dy_jzb + dz_jyb);
    f.p[cSx*f.ns + o] = -mu0*(-dx_jzb + dz_jxb);
}

// -grad Psi (k_grad_psi) and the beam part of Sx, Sy (k_sxsy_beam) in one pass over the plane: one launch less per slice
__global__ __launch_bounds__(256)
void k_gradpsi_sxsy (SlabView f, GradPsiSxSy a)
{
    gradpsi_sxsy_cell(f, a, (int)(blockIdx.x*blockDim.x + threadIdx.x) - f.ng, (int)blockIdx.y - f.ng);
}

// per-component sum |Q| over the valid cells, accumulated into acc[n]
__global__ __launch_bounds__(256)
void k_checksum (SlabView f, int ncomp, double* acc)
{
    const int n = blockIdx.y;
    double s = 0.0;
    const long cells = (long)f.nx*f.ny;
    for (long c = (long)blockIdx.x*blockDim.x + threadIdx.x; c < cells; c += (long)gridDim.x*blockDim.x) {
        const int j = (int)(c / f.nx), i = (int)(c - (long)j*f.nx);
        s += fabs(f(i, j, n));
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomic_add_f64(acc + n, part[0] + part[1] + part[2] + part[3]);
    (void)ncomp;
}

// piecewise-linear table, constant beyond its ends; n = 0: the factor is 1
__host__ __device__ inline double table_value (const double* x, const double* f, int n, double v)
{
    if (n <= 0) return 1.0;
    if (v <= x[0]) return f[0];
    if (v >= x[n - 1]) return f[n - 1];
    int k = 1;
    while (x[k] < v) ++k;
    const double t = (v - x[k - 1])/(x[k] - x[k - 1]);
    return f[k - 1] + t*(f[k] - f[k - 1]);
}

// weight = density(x, y, c t) * scale_fac (PlasmaParticleContainerInit.cpp:246-313) with the tabulated profile
// density * f_r(r) * f_t: prof = [r[nr] | f_r[nr]] on the device, ft the time factor of this step; a lattice point
// whose density is <= 0 holds no particle (the reference does not create it: here its slot is invalid)
// plasma sheet on the fixed-ppc lattice, ppc index outermost so that consecutive lanes own
// consecutive cells (PlasmaParticleContainerInit.cpp:192-313, ParticleUtil.H:72-83)
__global__ __launch_bounds__(256)
void k_init_plasma (hps_plasma pl, long n, int nx, int ny, int ppcx, int ppcy, double lox, double loy, double dx, double dy,
                    double weight, int level, int keyed, const double* __restrict__ prof, int nr, double ft, double radius_sq)
{
    const long k = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    const long cells = (long)nx*ny;
    const int ip = (int)(k / cells);
    const long c = k - (long)ip*cells;
    const int j = (int)(c / nx), i = (int)(c - (long)j*nx);
    const int ixp = ip % ppcx, iyp = ip / ppcx;
    const double x = lox + (i + (0.5 + ixp)/ppcx)*dx;
    const double y = loy + (j + (0.5 + iyp)/ppcy)*dy;
    // <plasma>.radius: no particle beyond it (PlasmaParticleContainerInit.cpp:262-266)
    const double fac = (x*x + y*y > radius_sq) ? 0.0 : ft*table_value(prof, prof + nr, nr, sqrt(x*x + y*y));
    pl.x[k] = x; pl.y[k] = y; pl.w[k] = fac > 0.0 ? weight*fac : 0.0;
    pl.ux[k] = 0.0; pl.uy[k] = 0.0; pl.psi[k] = 1.0;
    pl.x_prev[k] = x; pl.y_prev[k] = y;
    pl.ux_half[k] = 0.0; pl.uy_half[k] = 0.0; pl.psi_half[k] = fac > 0.0 ? 1.0 : 0.0;      // (0: not a particle, Tiling::valid_by_psi)
    // id = 1, cpu (level) = 0.  A species that can ionise carries its lattice index + 1 in the id bits: the key of its
    // random draws (ionization.hip); the reference only ever reads the sign of the id
    pl.idcpu[k] = (fac > 0.0 ? HPS_ID_VALID : 0ULL) | ((keyed ? (unsigned long long)(k + 1) : 1ULL) << 24);
    pl.ion_lev[k] = level;
}

// beam slice deposit (particles/deposition/BeamDepositCurrent.cpp:21-195), depos_order_z = 0
template <int ORDER>
__global__ __launch_bounds__(256)
void k_beam_deposit (SlabView f, BeamView b, long first, long count, int cjx, int cjy, int cjz,
                     double q_invvol, double clightsq_inv, double dx_inv, double dy_inv, double xoff, double yoff, const int* go,
                     int* disturbed = nullptr)
{
    if (go && *go == 0) return;      // (an iteration of the predictor-corrector loop enqueued past the loop's end)
    const long t = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (t >= count) return;
    const long ip = first + t;
    const double ux = b.ux[ip], uy = b.uy[ip], uz = b.uz[ip];
    const double gaminv = 1.0/sqrt(1.0 + ux*ux*clightsq_inv + uy*uy*clightsq_inv + uz*uz*clightsq_inv);
    const double wq = q_invvol*b.w[ip];
    // predictor-corrector loop: "something other than the cold plasma's rounding residue has reached the current planes"
    // (Engine::d_pc_dist) -- a term that is exactly zero (a cold beam's jx, jy on the Next slice) leaves the planes, and the
    // word, as they are
    if (disturbed && ((cjx >= 0 && (wq*(ux*gaminv) != 0.0 || wq*(uy*gaminv) != 0.0)) || (cjz >= 0 && wq*(uz*gaminv) != 0.0)))
        __hip_atomic_store(disturbed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double sx[ORDER + 1], sy[ORDER + 1];
    const int i0 = shape_weights<ORDER>((b.x[ip] - xoff)*dx_inv, sx);
    const int j0 = shape_weights<ORDER>((b.y[ip] - yoff)*dy_inv, sy);
#pragma unroll
    for (int iy = 0; iy <= ORDER; ++iy) {
#pragma unroll
        for (int ix = 0; ix <= ORDER; ++ix) {
            double* p = f.p + f.off(i0 + ix, j0 + iy);
            const double s = sx[ix]*sy[iy];
            if (cjx >= 0) { atomic_add_f64(p + cjx*f.ns, s*(wq*(ux*gaminv))); atomic_add_f64(p + cjy*f.ns, s*(wq*(uy*gaminv))); }
            if (cjz >= 0) atomic_add_f64(p + cjz*f.ns, s*(wq*(uz*gaminv)));
        }
    }
}

// the two beam deposits of a slice of the explicit schedule in one launch: jz of this slice's block (workgroups
// [0, nbA)) and jx, jy of the next slice's block (the rest) -- different components, no ordering between them
template <int ORDER>
__global__ __launch_bounds__(256)
void k_beam_deposit_pair (SlabView f, BeamPairWork w, double dx_inv, double dy_inv, double xoff, double yoff)
{
    beam_pair_block<ORDER>(f, w, (int)blockIdx.x, dx_inv, dy_inv, xoff, yoff);
}

__global__ __launch_bounds__(256)
void k_beam_new_step (int* nsub, long n)
{
    const long t = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (t < n && nsub[t] > 0) nsub[t] = 0;
}

// ------------------------------------------------------------------------------------------
// Engine
// ------------------------------------------------------------------------------------------
Engine::~Engine ()
{
    if (ps) hps_poisson_destroy(ps);
    if (mg) hps_mg_destroy(mg);
    (void)hipFree(slab.p); (void)hipFree(pl_real); (void)hipFree(pl.idcpu); (void)hipFree(pl.ion_lev);
    delete tiling;
    (void)hipFree(pl_real_alt); (void)hipFree(pl_alt.idcpu); (void)hipFree(pl_alt.ion_lev); 
    (void)hipFree(staging); (void)hipFree(d_open_mom); (void)hipFree(beam_data); (void)hipFree(beam_init);
    (void)hipFree(bm_store); (void)hipFree(bm_nsub); (void)hipFree(bm_nsub_scr); (void)hipFree(d_B); (void)hipFree(d_nfront);
    (void)hipFree(d_Bimp); (void)hipFree(d_beam_overflow); (void)hipFree(d_nqsa); (void)hipFree(d_checksum);
    (void)hipFree(d_pc); (void)hipFree(d_pc_aux); (void)hipFree(d_pc_go); if (h_pc) (void)hipHostFree(h_pc);
    (void)hipFree(d_laser_sum);
    if (laser) laser_destroy(*this);
    ion_destroy(*this);
    (void)hipFree(d_prof_r);
    (void)hipFree(d_fd); (void)hipFree(d_fd_comps); (void)hipFree(d_insitu); (void)hipFree(d_insitu_pl); (void)hipFree(d_insitu_bm);
    for (auto e : ev) (void)hipEventDestroy(e);
    for (auto e : hand_ev) if (e) (void)hipEventDestroy(e);
    if (st_aux) (void)hipStreamDestroy(st_aux);
    for (hipEvent_t ev : ev_aux) if (ev) (void)hipEventDestroy(ev);
    if (st_laser) (void)hipStreamDestroy(st_laser);
    if (ev_lfork) (void)hipEventDestroy(ev_lfork);
    if (ev_ldone) (void)hipEventDestroy(ev_ldone);
    if (st && st_pooled) {
        (void)hipStreamSynchronize(st);
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        g_stream_pool[st_device].push_back(st);
    } else if (st) (void)hipStreamDestroy(st);
}

static double beam_density_at (const hps_deck& d, double x, double y, double z)
{
    if (d.beam_profile == 0) {
        const double a = (x - d.beam_pos_mean[0])/d.beam_pos_std[0];
        const double b = (y - d.beam_pos_mean[1])/d.beam_pos_std[1];
        const double c = (z - d.beam_pos_mean[2])/d.beam_pos_std[2];
        return d.beam_density*std::exp(-0.5*a*a)*std::exp(-0.5*b*b)*std::exp(-0.5*c*c);
    }
    return d.beam_density;
}

// fixed-ppc beam, generated once on the host, stored slice-major (head slice first); restates
// InitBeamFixedPPCSlice (beam/BeamParticleContainerInit.cpp:198-346)
int Engine::init_beam ()
{
    std::vector<double> h[7];
    beam_off.assign(d.nz + 1, 0);
    if (d.beam_profile >= 0) {
        const int nppc = d.beam_ppc[0]*d.beam_ppc[1]*d.beam_ppc[2];
        const int nyp = d.beam_ppc[1], nzp = d.beam_ppc[2];
        for (int isl = d.nz - 1; isl >= 0; --isl) {
            beam_off[d.nz - 1 - isl] = (long)h[0].size();
            for (int j = 0; j < d.ny; ++j) for (int i = 0; i < d.nx; ++i) for (int ip = 0; ip < nppc; ++ip) {
                const int ixp = ip/(nyp*nzp), iyp = (ip % (nyp*nzp)) % nyp, izp = (ip % (nyp*nzp))/nyp;
                const double x = d.lo[0] + (i + (0.5 + ixp)/d.beam_ppc[0])*gm.dx;
                const double y = d.lo[1] + (j + (0.5 + iyp)/d.beam_ppc[1])*gm.dy;
                const double z = d.lo[2] + (isl + (0.5 + izp)/d.beam_ppc[2])*gm.dz;
                const double rx = x - d.beam_pos_mean[0], ry = y - d.beam_pos_mean[1];
                if (z >= d.beam_zmax || z < d.beam_zmin || (rx*rx + ry*ry) > d.beam_radius*d.beam_radius) continue;
                const double dens = beam_density_at(d, x, y, z);
                if (dens <= 0.0) continue;
                h[0].push_back(x); h[1].push_back(y); h[2].push_back(z);
                h[3].push_back(d.beam_umean[0]*gm.c); h[4].push_back(d.beam_umean[1]*gm.c); h[5].push_back(d.beam_umean[2]*gm.c);
                h[6].push_back(std::fabs(dens*(d.si_units ? gm.dx*gm.dy*gm.dz/nppc : 1.0/nppc)));      // scale_fac, BeamParticleContainerInit.cpp:215-216
            }
        }
    }
    beam_off[d.nz] = (long)h[0].size();
    return install_beam(h);
}

// A beam the HOST has initialised (any of the reference's injection types: fixed_weight, fixed_weight_pdf, from_file --
// beam/BeamParticleContainerInit.cpp:348-695 -- draw from amrex::Random or read a file; the slice operators do not care):
// n particles as [7][n] x y z ux uy uz w in the engine's units (u = gamma beta c, w as the deposition takes it), in any
// order.  They are binned into the box's slices (BoxSorter's rule, sorting/BoxSort.cpp:34-43: slice = int((z - lo_z)/dz)), head slice first, input
// order kept inside a slice; particles outside the box in z are counted and left out.  Replaces the deck's beam.
int Engine::set_beam_particles (long n, const double* soa, long* n_outside)
{
    HPS_REQUIRE(n >= 0 && (soa || n == 0), "hps_engine_set_beam_particles: null argument");
    HPS_REQUIRE(steps_begun == 0 && step_index < 0, "hps_engine_set_beam_particles: call before the first hps_engine_begin_step");
    std::vector<long> count((size_t)d.nz + 1, 0);
    std::vector<int> where((size_t)n);
    long outside = 0;
    const double inv_dz = 1.0/gm.dz;
    for (long i = 0; i < n; ++i) {
        const double z = soa[2*n + i];
        const double t = (z - d.lo[2])*inv_dz;
        const int q = (t > -1.0e9 && t < 1.0e9) ? static_cast<int>(t) : -1;      // (the reference's cast: truncation)
        if (q < 0 || q >= d.nz) { where[(size_t)i] = -1; ++outside; continue; }
        const int p = d.nz - 1 - q;                      // p-th slice from the head
        where[(size_t)i] = p; ++count[(size_t)p + 1];
    }
    if (n_outside) *n_outside = outside;
    HPS_REQUIRE(outside == 0 || n_outside, "hps_engine_set_beam_particles: particles outside the box in z (pass n_outside to have them counted and left out)");
    beam_off.assign(d.nz + 1, 0);
    for (int p = 0; p < d.nz; ++p) beam_off[p + 1] = beam_off[p] + count[(size_t)p + 1];
    std::vector<long> next(beam_off.begin(), beam_off.end() - 1);
    std::vector<double> h[7];
    for (int k = 0; k < 7; ++k) h[k].resize((size_t)(n - outside));
    for (long i = 0; i < n; ++i) {
        const int p = where[(size_t)i];
        if (p < 0) continue;
        const long j = next[(size_t)p]++;
        for (int k = 0; k < 7; ++k) h[k][(size_t)j] = soa[(size_t)k*n + i];
    }
    HPS_HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFree(beam_data); (void)hipFree(beam_init); (void)hipFree(bm_store); (void)hipFree(bm_nsub); (void)hipFree(bm_nsub_scr);
    (void)hipFree(d_B); (void)hipFree(d_nfront); (void)hipFree(d_Bimp); (void)hipFree(d_beam_overflow);
    beam_data = beam_init = beam_cur = nullptr; bm_store = nullptr; bm_nsub = bm_nsub_scr = nullptr;
    d_B = nullptr; d_nfront = nullptr; d_Bimp = nullptr; d_beam_overflow = nullptr;
    return install_beam(h);
}

// the beam's device storage from the host arrays h[7] (head slice first, beam_off filled in)
int Engine::install_beam (std::vector<double> (&h)[7])
{
    nbeam = (long)h[0].size();
    {
        const int js = d.nx + 2*g, jn = d.ny + 2*g;
        full_box = Box{0, js - 1, 0, jn - 1};
        beam_box_init = Box{0, -1, 0, -1};          // empty
        if (nbeam > 0) {
            double xlo = h[0][0], xhi = h[0][0], ylo = h[1][0], yhi = h[1][0];
            for (long k = 1; k < nbeam; ++k) { xlo = std::min(xlo, h[0][k]); xhi = std::max(xhi, h[0][k]); ylo = std::min(ylo, h[1][k]); yhi = std::max(yhi, h[1][k]); }
            // nearest cell -+ (deposit footprint 2 + one cell for the centred differences + 1 spare)
            const int m = 4;
            beam_box_init.ilo = std::max(0,      (int)std::floor((xlo - gm.xoff)/gm.dx + 0.5) - m + g);
            beam_box_init.ihi = std::min(js - 1, (int)std::floor((xhi - gm.xoff)/gm.dx + 0.5) + m + g);
            beam_box_init.jlo = std::max(0,      (int)std::floor((ylo - gm.yoff)/gm.dy + 0.5) - m + g);
            beam_box_init.jhi = std::min(jn - 1, (int)std::floor((yhi - gm.yoff)/gm.dy + 0.5) + m + g);
        }
        if (d.grid_current_on) beam_box_init = full_box;      // the grid current fills jz_beam of every cell
        beam_box = beam_box_init;
    }
    if (nbeam > 0) {
        // slice-major blocks: block p (p-th slice from the head) = [7][count_p] at 7*beam_off[p]
        std::vector<double> blk((size_t)7*nbeam);
        for (int p = 0; p < d.nz; ++p) {
            const long first = beam_off[p], cnt = beam_off[p + 1] - first;
            for (int k = 0; k < 7; ++k)
                for (long i = 0; i < cnt; ++i) blk[(size_t)7*first + (size_t)k*cnt + i] = h[k][first + i];
        }
        HPS_HIP_CHECK(hipMalloc(&beam_data, 7*nbeam*sizeof(double)));
        HPS_HIP_CHECK(hipMalloc(&beam_init, 7*nbeam*sizeof(double)));
        HPS_HIP_CHECK(hipMemcpy(beam_init, blk.data(), 7*nbeam*sizeof(double), hipMemcpyHostToDevice));
        HPS_HIP_CHECK(hipMemcpy(beam_data, beam_init, 7*nbeam*sizeof(double), hipMemcpyDeviceToDevice));
        beam_cur = beam_data;
    }
    moving = (d.dt != 0.0);
    if (moving) {
        // global SoA in head-first order + device boundaries (beam.hip)
        const long nb = std::max(nbeam, 1L);
        const bool spin = d.beam_spin_tracking != 0;
        beam_rows = spin ? 10 : 7;
        HPS_HIP_CHECK(hipMalloc(&bm_store, (size_t)2*beam_rows*nb*sizeof(double)));
        HPS_HIP_CHECK(hipMalloc(&bm_nsub, (size_t)nb*sizeof(int)));
        HPS_HIP_CHECK(hipMalloc(&bm_nsub_scr, (size_t)nb*sizeof(int)));
        double* a = bm_store;
        bm = BeamSoA{a, a + nb, a + 2*nb, a + 3*nb, a + 4*nb, a + 5*nb, a + 6*nb, bm_nsub,
                     spin ? a + 7*nb : nullptr, spin ? a + 8*nb : nullptr, spin ? a + 9*nb : nullptr};
        a += (size_t)beam_rows*nb;
        bm_scr = BeamSoA{a, a + nb, a + 2*nb, a + 3*nb, a + 4*nb, a + 5*nb, a + 6*nb, bm_nsub_scr,
                         spin ? a + 7*nb : nullptr, spin ? a + 8*nb : nullptr, spin ? a + 9*nb : nullptr};
        if (spin && nbeam > 0) {       // initial_spin, normalised, for every particle (BeamParticleContainer.cpp:390-402)
            const double* s0 = d.beam_initial_spin;
            const double nrm = std::sqrt(s0[0]*s0[0] + s0[1]*s0[1] + s0[2]*s0[2]);
            HPS_REQUIRE(nrm > 0.0, "hps_engine_create: beam initial_spin must not vanish");
            for (int q = 0; q < 3; ++q) {
                std::vector<double> v((size_t)nbeam, s0[q]/nrm);
                HPS_HIP_CHECK(hipMemcpy(bm_store + (size_t)(7 + q)*nb, v.data(), nbeam*sizeof(double), hipMemcpyHostToDevice));
            }
        }
        for (int k = 0; k < 7 && nbeam > 0; ++k)
            HPS_HIP_CHECK(hipMemcpy(bm_store + (size_t)k*nb, h[k].data(), nbeam*sizeof(double), hipMemcpyHostToDevice));
        HPS_HIP_CHECK(hipMemset(bm_nsub, 0, (size_t)nb*sizeof(int)));
        h_B.assign(beam_off.begin(), beam_off.end());
        HPS_HIP_CHECK(hipMalloc(&d_B, (size_t)(d.nz + 1)*sizeof(long)));
        HPS_HIP_CHECK(hipMemcpy(d_B, h_B.data(), (size_t)(d.nz + 1)*sizeof(long), hipMemcpyHostToDevice));
        HPS_HIP_CHECK(hipMalloc(&d_nfront, (size_t)(d.nz + 2)*sizeof(int)));
        HPS_HIP_CHECK(hipMemset(d_nfront, 0, (size_t)(d.nz + 2)*sizeof(int)));
        HPS_HIP_CHECK(hipMalloc(&d_Bimp, (size_t)(d.nz + 1)*sizeof(long)));
        HPS_HIP_CHECK(hipMemset(d_Bimp, 0, (size_t)(d.nz + 1)*sizeof(long)));
        HPS_HIP_CHECK(hipMalloc(&d_beam_overflow, sizeof(int)));
        HPS_HIP_CHECK(hipMemset(d_beam_overflow, 0, sizeof(int)));
        long mx = 0;
        for (int p = 0; p < d.nz; ++p) mx = std::max(mx, beam_off[p + 1] - beam_off[p]);
        beam_cap = std::max(2*mx, 1L);                 // particles a slice may hold in a hand-off message
        if (nbeam > 0) beam_box = beam_box_init = full_box;          // a moving beam may go anywhere (no beam at all, e.g. a laser driver: the box stays empty)
    }
    return HPS_OK;
}

// HPS_CU_MASKS="lo-hi[+lo-hi...],..." (diagnostic): the j-th engine created in this process runs its stream on the compute
// units of entry j % n only (hipExtStreamCreateWithCUMask).  Bit i of a mask is compute unit i / 8 of XCD i % 8 on this GPU (the
// driver deals the bits round the XCDs first), so ranges that are multiples of 8 wide take the same number of CUs from every XCD.
//
// Default: the engines' streams come from a small per-device POOL whose streams are created back to back the first time an
// engine is created on the device.  Why: the runtime gives every stream a hardware queue when it is first used, hardware
// queues go to the 4 pipes of the compute front end in the order they are created (queue k of the PROCESS on pipe k mod 4),
// and two queues on one pipe do not overlap chains of short dependent kernels (the multigrid's ~30 launches of 5-10 us): two
// stages whose queues are 4 apart take turns.  Measured on MI355X (scripts/ubench/queue_pairs.hip; profiles/r05_queue_pairs.txt,
// r05_stage_queues.txt): three stages whose engines were created AFTER the ring's two streams sat on queues 1, 5, 6 and made
// 1700 slices/s, created before them (queues 1, 3, 4) 2186.  With the pool the stages' queues are consecutive whatever else
// the host creates in between.  An engine returns its stream to the pool when it is destroyed.  HPS_STREAM_POOL=<n> sets the
// pool's size (0 = a fresh stream per engine, as before).  Default 3: with the process's null stream that is the runtime's
// default of 4 hardware queues per priority (GPU_MAX_HW_QUEUES) -- a 4th pooled stream has to share one of those queues,
// and two processes on one device with pools of 4 made 1249 slices/s together against 1916-1928 with pools of 0-3
// (profiles/r05_stream_pool_two_processes.txt).  Further engines get streams of their own.
// Self-check of the engines' stream pool (HPS_STREAM_POOL_CHECK=0: off).  The pool's size is a property of the runtime read
// off one box (four front-end pipes per priority less the null stream's: streams created back to back land on different
// pipes up to 3; a fourth shares one -- profiles/r05_queue_pairs.txt, r05_stream_pool_two_processes.txt).  Another runtime, or
// a process that had created streams of its own first, may map two pool streams to one pipe, and three stages in flight
// then take turns without anything saying so.  So once per device, when the pool is made: a one-workgroup kernel that spins
// for 100 us, R times on both streams of every pair; side by side takes R x 100 us, taking turns 2 R x 100 us.
__global__ void k_pool_spin (long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
namespace { std::map<int, int> g_pool_shared_pairs; }
static void check_stream_pool (int device, const std::vector<hipStream_t>& pool)
{
    g_pool_shared_pairs[device] = -1;
    if (const char* v = std::getenv("HPS_STREAM_POOL_CHECK")) if (std::atoi(v) == 0) return;
    if (pool.size() < 2) { g_pool_shared_pairs[device] = 0; return; }
    int rate = 0;
    if (hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, device) != hipSuccess || rate <= 0) return;
    const long long ticks = (long long)rate/10;                 // (kHz: 100 us)
    const int R = 4;
    int shared = 0;
    std::string which;
    for (size_t i = 0; i < pool.size(); ++i) for (size_t j = i + 1; j < pool.size(); ++j) {
        hipLaunchKernelGGL(k_pool_spin, dim3(1), dim3(64), 0, pool[i], ticks/20);      // (code object loaded, queues awake)
        hipLaunchKernelGGL(k_pool_spin, dim3(1), dim3(64), 0, pool[j], ticks/20);
        (void)hipStreamSynchronize(pool[i]); (void)hipStreamSynchronize(pool[j]);
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < R; ++r) {
            hipLaunchKernelGGL(k_pool_spin, dim3(1), dim3(64), 0, pool[i], ticks);
            hipLaunchKernelGGL(k_pool_spin, dim3(1), dim3(64), 0, pool[j], ticks);
        }
        (void)hipStreamSynchronize(pool[i]); (void)hipStreamSynchronize(pool[j]);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us > 1.6*R*100.0) { ++shared; which += " (" + std::to_string(i) + "," + std::to_string(j) + "): " + std::to_string((int)us) + " us"; }
    }
    g_pool_shared_pairs[device] = shared;
    if (shared)
        std::fprintf(stderr, "libhpslice: device %d: %d pair(s) of the %zu engine streams of the pool take turns instead of running side by side "
                             "(%d x 100 us on both streams of a pair, expected %d us:%s) -- several stages in flight on this device will share a "
                             "front-end pipe; try HPS_STREAM_POOL=%zu or GPU_MAX_HW_QUEUES\n", device, shared, pool.size(), R, R*100, which.c_str(), pool.size() - 1);
}
extern "C" int hps_stream_pool_shared_pairs (int device)      // -1: not checked (no pool yet, or HPS_STREAM_POOL_CHECK=0)
{
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    auto it = g_pool_shared_pairs.find(device);
    return it == g_pool_shared_pairs.end() ? -1 : it->second;
}

static hipError_t create_engine_stream (hipStream_t* s, int device, bool* pooled)
{
    *pooled = false;
    const char* v = std::getenv("HPS_CU_MASKS");
    if (!v || !std::strchr(v, '-')) {                       // (unset, or no range in it)
        int want = 3;
        if (const char* p = std::getenv("HPS_STREAM_POOL")) want = std::atoi(p);
        if (want <= 0) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        std::vector<hipStream_t>& pool = g_stream_pool[device];
        if (!g_pool_made[device]) {
            g_pool_made[device] = true;
            void* scratch = nullptr;
            if (hipMalloc(&scratch, 256) != hipSuccess) scratch = nullptr;
            for (int k = 0; k < want; ++k) {
                hipStream_t t = nullptr;
                if (hipStreamCreateWithFlags(&t, hipStreamNonBlocking) != hipSuccess) break;
                // the first command makes the runtime acquire the stream's hardware queue: now, in this order
                if (scratch) { (void)hipMemsetAsync(scratch, 0, 256, t); (void)hipStreamSynchronize(t); }
                pool.push_back(t);
            }
            if (scratch) (void)hipFree(scratch);
            check_stream_pool(device, pool);
            std::reverse(pool.begin(), pool.end());          // (handed out from the back: first created first)
        }
        if (pool.empty()) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);       // more engines than the pool holds
        *s = pool.back(); pool.pop_back(); *pooled = true;
        return hipSuccess;
    }
    static std::atomic<int> created{0};
    std::vector<std::string> entries;
    {   std::string cur; for (const char* p = v; ; ++p) { if (*p == ',' || !*p) { entries.push_back(cur); cur.clear(); if (!*p) break; } else cur += *p; } }
    const std::string& e = entries[created++ % entries.size()];
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t pos = 0;
    while (pos < e.size()) {
        size_t end = e.find('+', pos); if (end == std::string::npos) end = e.size();
        int lo = 0, hi = -1;
        if (std::sscanf(e.substr(pos, end - pos).c_str(), "%d-%d", &lo, &hi) == 2)
            for (int b = std::max(lo, 0); b <= std::min(hi, 255); ++b) mask[b >> 5] |= 1u << (b & 31);
        pos = end + 1;
    }
    return hipExtStreamCreateWithCUMask(s, 8, mask);
}

int Engine::create (const hps_deck& deck, int device)
{
    d = deck;
    HPS_REQUIRE(d.nx >= 4 && d.ny >= 4 && d.nz >= 1, "hps_engine_create: bad grid");
    HPS_REQUIRE(d.order >= 0 && d.order <= 3, "hps_engine_create: depos_order must be 0..3");
    HPS_REQUIRE(d.n_subcycles >= 0, "hps_engine_create: plasma n_subcycles must be >= 1 (0 = default 1)");
    if (d.n_subcycles == 0) d.n_subcycles = 1;       // <plasma>.n_subcycles default (particles/plasma/PlasmaParticleContainer.H:182)
    if (d.field_bc != 0 && d.field_bc != 1) { set_error("hps_engine_create: boundary.field must be 0 (Dirichlet) or 1 (Open)"); return HPS_ERR_UNSUPPORTED; }
    if (d.field_bc == 1) {
        // (the expansion is about x = y = 0, which must lie inside the box: fields/Fields.cpp:703-705)
        const double radius = std::min(std::min(std::fabs(d.lo[0]), std::fabs(d.hi[0])), std::min(std::fabs(d.lo[1]), std::fabs(d.hi[1])));
        HPS_REQUIRE(radius > 0.0 && d.lo[0] < 0.0 && d.hi[0] > 0.0 && d.lo[1] < 0.0 && d.hi[1] > 0.0, "hps_engine_create: boundary.field = Open needs x = y = 0 inside the box");
        HPS_HIP_CHECK(hipMalloc(&d_open_mom, (size_t)4*256*2*19*sizeof(double)));      // [source][OPEN_PARTS][2 (OPEN_ORDER + 1)]
    }
    pc = (d.bxby_solver != 0);
    if (const char* v = std::getenv("HPS_GATED_PUSH")) gate_push = std::atoi(v) != 0;
    if (const char* v = std::getenv("HPS_AUX_STREAM")) aux_on = std::atoi(v) != 0;
    if (aux_on) {
        HPS_HIP_CHECK(hipStreamCreateWithFlags(&st_aux, hipStreamNonBlocking));
        for (hipEvent_t& ev : ev_aux) HPS_HIP_CHECK(hipEventCreateWithFlags(&ev, event_flags(false)));
    }
    if (const char* v = std::getenv("HPS_FOLD_TAIL")) fold_tail = std::atoi(v) != 0;
    if (const char* v = std::getenv("HPS_POST_IN_PUSH")) post_in_push = std::atoi(v) != 0;
    if (const char* v = std::getenv("HPS_FOLD_BEAM")) fold_beam = std::atoi(v) != 0;
    if (const char* v = std::getenv("HPS_FOLD_HIERARCHY")) fold_hierarchy = std::atoi(v) != 0;
    if (const char* v = std::getenv("HPS_VALID_BY_W")) valid_by_w = std::atoi(v) != 0;
    if (const char* v = std::getenv("HPS_GATED_ION_PUSH")) gate_ion_push = std::atoi(v) != 0;
    if (const char* v = std::getenv("HPS_LAZY_SHIFT")) lazy_shift = std::atoi(v) != 0;
    if (const char* v = std::getenv("HPS_FUSE_SOURCES")) fuse_sources = std::atoi(v) != 0;
    if (const char* v = std::getenv("HPS_SORT_FALLBACK_DIV")) { const long q = std::atol(v); if (q >= 1) fallback_div = q; }
    HPS_REQUIRE(!(d.beam_spin_tracking && d.dt == 0.0), "hps_engine_create: spin tracking needs a moving beam (hipace.dt != 0)");
    if (d.predcorr_tol > 0.0) pc_tol = d.predcorr_tol;
    if (d.predcorr_max_iter > 0) pc_max_iter = d.predcorr_max_iter;
    if (d.predcorr_mix > 0.0) pc_mix = d.predcorr_mix;
    HPS_HIP_CHECK(hipSetDevice(device));
    HPS_HIP_CHECK(create_engine_stream(&st, device, &st_pooled));
    st_device = device;
    if (const char* v = std::getenv("HPS_LASER_ASYNC")) laser_async = std::atoi(v) != 0;
    if (laser_async && d.laser_on && d.laser_solver >= 1 && d.dt != 0.0) {
        // below the engine's stream: its kernels fill what the slice leaves idle, they are not to win a CU from it
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        HPS_HIP_CHECK(hipStreamCreateWithPriority(&st_laser, hipStreamNonBlocking, lo));
        HPS_HIP_CHECK(hipEventCreateWithFlags(&ev_lfork, event_flags(false)));
        HPS_HIP_CHECK(hipEventCreateWithFlags(&ev_ldone, event_flags(false)));
    }
    g = (d.order + 1)/2 + 1;                       // Fields::AllocData (fields/Fields.cpp:63-64)
    ncomp = pc ? (d.deposit_rho ? HPS_PC_RHO + 1 : HPS_PC_RHO) : (d.deposit_rho ? HPS_C_RHO + 1 : HPS_C_RHO);
    if (d.beam_radiation_reaction && !d.si_units && !(d.background_density_SI > 0.0)) {      // BeamParticleAdvance.cpp:39-43
        set_error("hps_engine_create: radiation reaction in normalised units needs background_density_SI"); return HPS_ERR_ARG; }
    if (d.laser_on) {
        if (pc) { set_error("hps_engine_create: the laser needs the explicit solver"); return HPS_ERR_UNSUPPORTED; }
        HPS_REQUIRE(d.laser_w0 > 0.0 && d.laser_L0 > 0.0 && d.laser_lambda0 > 0.0, "hps_engine_create: laser w0, L0, lambda0 must be positive");
        if (d.laser_solver < 0 || d.laser_solver > 2) { set_error("hps_engine_create: lasers.solver_type must be 0 (static), 1 (fft) or 2 (multigrid)"); return HPS_ERR_UNSUPPORTED; }
        c_aabs = ncomp++;                // appended last
    }
    gm.dx = (d.hi[0] - d.lo[0])/d.nx; gm.dy = (d.hi[1] - d.lo[1])/d.ny; gm.dz = (d.hi[2] - d.lo[2])/d.nz;
    gm.xoff = 0.5*(d.lo[0] + d.hi[0] - gm.dx*(d.nx - 1));
    gm.yoff = 0.5*(d.lo[1] + d.hi[1] - gm.dy*(d.ny - 1));
    gm.c = gm.ep0 = gm.mu0 = gm.q_e = gm.m_e = 1.0;
    if (d.si_units) {       // make_constants_SI (utils/Constants.H:15-24, 2018 CODATA)
        gm.c = 299792458.0; gm.ep0 = 8.8541878128e-12; gm.mu0 = 1.25663706212e-06; gm.q_e = 1.602176634e-19; gm.m_e = 9.1093837015e-31;
    }
    gm.plo[0] = d.lo[0]; gm.plo[1] = d.lo[1]; gm.phi[0] = d.hi[0]; gm.phi[1] = d.hi[1];
    gm.bc = d.bc; gm.normalized = d.si_units ? 0 : 1;

    slab.nx = d.nx; slab.ny = d.ny; slab.ng = g; slab.ncomp = ncomp;
    slab.jstride = d.nx + 2*g; slab.nstride = slab.jstride*(d.ny + 2*g);
    HPS_HIP_CHECK(hipMalloc(&slab.p, (size_t)slab.nstride*ncomp*sizeof(double)));
    HPS_HIP_CHECK(hipMemset(slab.p, 0, (size_t)slab.nstride*ncomp*sizeof(double)));
    HPS_HIP_CHECK(hipMalloc(&staging, (size_t)3*d.nx*d.ny*sizeof(double)));

    const int nppc = d.plasma_ppc[0]*d.plasma_ppc[1];
    np = (d.plasma_density > 0.0) ? (long)nppc*d.nx*d.ny : 0;
    np_init = np; np_cap = np;
    if (d.ion_on) {
        // (predictor-corrector: every species is pushed to the temporary slice and deposited in turn inside the loop; the ADK
        //  decisions are taken once per slice, ahead of the committing push, as Hipace.cpp:693-701 has them)
        HPS_REQUIRE(d.ion_ppc[0] >= 1 && d.ion_ppc[1] >= 1 && d.ion_density > 0.0 && d.ion_mass > 0.0 && d.ion_charge != 0.0,
                    "hps_engine_create: the ion species needs ppc, density, mass and charge");
        ion.n = (long)d.ion_ppc[0]*d.ion_ppc[1]*d.nx*d.ny;
        if (int e = ion_create(*this)) return e;
        // every ion can release Z - initial level electrons into the first species
        np_cap = np_init + ion.n*(long)(d.ion_Z - d.ion_init_level);
    }
    std::memset(&pl, 0, sizeof(pl));
    pl.n = np;
    auto alloc_sheet = [&] (hps_plasma& p, double*& real, long cap, bool alias) -> int {
        HPS_HIP_CHECK(hipMalloc(&real, (size_t)cap*11*sizeof(double)));
        double** arr[11] = {&p.x, &p.y, &p.w, &p.ux, &p.uy, &p.psi, &p.x_prev, &p.y_prev, &p.ux_half, &p.uy_half, &p.psi_half};
        for (int k = 0; k < 11; ++k) *arr[k] = real + (size_t)k*cap;
        // explicit solver: every push commits its state (temp_slice = false), so x_prev == x and y_prev == y at
        // all times (PlasmaParticleAdvance.cpp:176-188): alias them -- two arrays less to write per push and to
        // move per re-sort.  The C ABI keeps 11 pointers; the kernels skip the second store when two coincide.
        // (the predictor-corrector pushes to temporary slices from x_prev: separate arrays there)
        if (alias) { p.x_prev = p.x; p.y_prev = p.y; }
        HPS_HIP_CHECK(hipMalloc(&p.idcpu, (size_t)cap*sizeof(uint64_t)));
        HPS_HIP_CHECK(hipMalloc(&p.ion_lev, (size_t)cap*sizeof(int32_t)));
        return HPS_OK;
    };
    if (np_cap > 0) { if (int e = alloc_sheet(pl, pl_real, np_cap, !pc)) return e; }
    if (ion.n > 0) { ion.pl.n = ion.n; if (int e = alloc_sheet(ion.pl, ion.real, ion.n, !pc)) return e; }
    HPS_HIP_CHECK(hipMalloc(&d_laser_sum, sizeof(double)));
    HPS_HIP_CHECK(hipMemset(d_laser_sum, 0, sizeof(double)));
    HPS_HIP_CHECK(hipMalloc(&d_nqsa, sizeof(int)));
    HPS_HIP_CHECK(hipMemset(d_nqsa, 0, sizeof(int)));
    HPS_HIP_CHECK(hipMalloc(&d_checksum, HPS_PC_NCOMP_MAX*sizeof(double)));
    HPS_HIP_CHECK(hipMemset(d_checksum, 0, HPS_PC_NCOMP_MAX*sizeof(double)));
    if (int e = hps_poisson_create(d.nx, d.ny, gm.dx, gm.dy, &ps)) return e;
    if (int e = hps_mg_create(d.nx, d.ny, gm.dx, gm.dy, &mg)) return e;
    // the halo-fallback counter lives in the header slot of the multigrid's norm buffer: it reaches the host
    // with the read-back every solve does anyway (no copy of its own)
    {   int* dw = nullptr; const int* hw = nullptr;
        mg_rider(mg, &dw, &hw);
        d_nfallback = dw; h_nfallback = hw; }
    if (pc) {
        // no multigrid solves in this mode: the counter travels with the per-iteration read-back of the B error
        HPS_HIP_CHECK(hipMalloc(&d_pc, 4*sizeof(double)));
        HPS_HIP_CHECK(hipMemset(d_pc, 0, 4*sizeof(double)));
        HPS_HIP_CHECK(hipMalloc(&d_pc_aux, 4*sizeof(double)));      // error of the last two iterations (k_pc_mix)
        HPS_HIP_CHECK(hipMemset(d_pc_aux, 0, 4*sizeof(double)));
        HPS_HIP_CHECK(hipHostMalloc(&h_pc, 8*PC_MAX_SPEC*sizeof(double), hipHostMallocMapped));      // slot 0 + one slot per loop iteration
        std::memset(h_pc, 0, 8*PC_MAX_SPEC*sizeof(double));
        HPS_HIP_CHECK(hipHostGetDevicePointer((void**)&h_pc_dev, h_pc, 0));
        HPS_HIP_CHECK(hipMalloc(&d_pc_go, (PC_MAX_SPEC + 1)*sizeof(int)));
        HPS_HIP_CHECK(hipMemset(d_pc_go, 0, (PC_MAX_SPEC + 1)*sizeof(int)));
        // The exact zeros of the serial path (round 6; replaces the magnitude floor below, which stays as a diagnostic).  Ahead
        // of the first beam particle the serial CPU path holds EXACT zeros in every field -- electron and ion charge cancel term
        // by term, a cold plasma at rest deposits no current -- so ComputeRelBFieldError returns 0 (norm_B > 0 ? diff/norm_B : 0,
        // fields/Fields.cpp:1283), the loop leaves after one pass there AND on the first slice that holds beam (the guess it
        // compares with is that zero).  A scatter with atomics leaves 1e-16 residue of the background charge instead, and the
        // literal rule iterates to max_iterations on it.  d_pc_dist is one device word, cleared by begin_step: "something
        // other than that residue has been deposited on this slice or one ahead" -- set by the beam's deposition when a term
        // is not exactly zero (k_beam_deposit, k_beam_deposit_dyn) and from the start with a grid current.  While it is clear
        // the slice's loop leaves after its first pass by the LITERAL rule (sum |B| of the guess IS 0: see next sentence) and
        // k_pc_mix stores the zero the serial path holds into Bx, By instead of a mix of residues; once it is set the
        // literal rule applies to whatever sum |B| is -- a moving beam's thin head iterates as it does on the CPU.
        // HPS_PC_EXACT_ZERO=0: off (the literal rule on residue: 2400+ instead of 1631 iterations on config 2's box).
        {   const char* v = std::getenv("HPS_PC_EXACT_ZERO");
            if (!(v && std::atoi(v) == 0)) d_pc_dist = d_pc_go + PC_MAX_SPEC; }
        // Rounding floor of sum |B| (k_rel_b_error).  Ahead of the driver the serial CPU path has EXACT zeros -- electron and ion
        // charge cancel term by term -- so ComputeRelBFieldError returns 0 (fields/Fields.cpp:1283) and the loop leaves after
        // one pass, also on the first slice that holds beam.  A scatter with atomics leaves 1e-16 residue of the background
        // charge; without a floor the loop runs to max_iterations on that noise on every slice ahead of the driver (config 2:
        // 54 slices x 30 iterations = 40 % of the box's time) and enters the driver on a different path, per cent away from
        // the CPU's.  What can be resolved of B is eps x mu0 c sum|rho_background| L: 1e-12 of that is "zero".
        // HPS_PC_NOISE_FLOOR=<relative floor>, 0: the literal rule.
        {   double rel = 0.0;           // (round 6: no floor by default -- d_pc_dist above; rounds 4-5 ran with 1e-12)
            if (const char* v = std::getenv("HPS_PC_NOISE_FLOOR")) rel = std::atof(v);
            pc_floor = rel*gm.mu0*gm.c*std::fabs(d.plasma_charge*d.plasma_density)*(double)d.nx*d.ny*(d.nx*gm.dx); }
        // HPS_PC_SPECULATE=0: the host decides after every iteration, as in rounds 1-3
        {   const char* v = std::getenv("HPS_PC_SPECULATE");
            pc_speculate = !(v && std::atoi(v) == 0) && pc_max_iter <= PC_MAX_SPEC - 2 && poisson_gateable(ps)
                           && d.field_bc == 0; }      // (the open boundary's two launches per solve are not gated: host-controlled loop)
        d_nfallback = reinterpret_cast<int*>(d_pc + 2); h_nfallback = reinterpret_cast<const int*>(h_pc + 4);
    }
    if (c_aabs >= 0) { if (int e = laser_create(*this)) return e; }
    return init_beam();
}

int Engine::setup_tiling ()
{
    if (tile_size == 0 || np_cap == 0 || tiling) return HPS_OK;
    auto second = [&] (hps_plasma& alt, double*& real, long cap, bool alias) -> int {
        std::memset(&alt, 0, sizeof(alt));
        HPS_HIP_CHECK(hipMalloc(&real, (size_t)cap*11*sizeof(double)));
        double** arr[11] = {&alt.x, &alt.y, &alt.w, &alt.ux, &alt.uy, &alt.psi, &alt.x_prev, &alt.y_prev,
                            &alt.ux_half, &alt.uy_half, &alt.psi_half};
        for (int k = 0; k < 11; ++k) *arr[k] = real + (size_t)k*cap;
        if (alias) { alt.x_prev = alt.x; alt.y_prev = alt.y; }
        HPS_HIP_CHECK(hipMalloc(&alt.idcpu, (size_t)cap*sizeof(uint64_t)));
        HPS_HIP_CHECK(hipMalloc(&alt.ion_lev, (size_t)cap*sizeof(int32_t)));
        return HPS_OK;
    };
    if (int e = tiling_create(d.nx, d.ny, tile_size, np_cap, &tiling)) return e;
    {   const char* v = std::getenv("HPS_VALID_BY_PSI");       // the electrons' push without its idcpu read (tiling.h)
        tiling->valid_by_psi = (v && std::atoi(v) != 0); }       // (measured: the push takes the same time with and without, off)
    if (int e = second(pl_alt, pl_real_alt, np_cap, !pc)) return e;
    pl_alt.n = np;
    if (ion.n > 0) {
        // the ionisable species on the engine's tile size.  (Round 2 gave a species with at most one particle per cell 32 x 32-cell
        // tiles -- 256 particles in a 16 x 16 tile, one per thread, against the tile's fixed cost.  With the tile skip most of
        // its tiles cost nothing, and its kernels last as long as the few workgroups around the laser's axis that do have
        // charged ions: four times as many of them, a quarter as long -- config 5 989 -> 1004 slices/s.  HPS_ION_TILE=32)
        int ion_ts = tile_size;
        if (const char* v = std::getenv("HPS_ION_TILE")) { const int t = std::atoi(v); if (t == 16 || t == 32) ion_ts = t; }
        if (int e = tiling_create(d.nx, d.ny, ion_ts, ion.n, &ion.tiling)) return e;
        if (int e = second(ion.pl_alt, ion.real_alt, ion.n, !pc)) return e;
        ion.pl_alt.n = ion.n;
        HPS_HIP_CHECK(hipMalloc(&ion.d_tile_flag, (size_t)ion.tiling->g.ntiles*sizeof(int)));
        {   const char* v = std::getenv("HPS_ION_TILE_SKIP");
            if (!(v && std::atoi(v) == 0)) HPS_HIP_CHECK(hipMalloc(&ion.d_fbound, (size_t)5*ion.tiling->g.ntiles*sizeof(double))); }
    }
    return HPS_OK;
}

int Engine::resort ()
{
    pl_alt.n = pl.n;
    if (int e = tiling_sort(tiling, pl, pl_alt, gm, st)) return e;
    std::swap(pl, pl_alt);
    std::swap(pl_real, pl_real_alt);
    since_sort = 0; ++n_sorts;
    fb_at_sort = h_nfallback ? *h_nfallback : 0;
    return HPS_OK;
}

int Engine::begin_step ()
{
    if (int e = setup_tiling()) return e;
    if (moving && steps_begun > 0) {
        // a slice that outgrew the hand-off capacity during the previous step lost particles: refuse to go on
        int ov = 0;
        HPS_HIP_CHECK(hipMemcpyAsync(&ov, d_beam_overflow, sizeof(int), hipMemcpyDeviceToHost, st));
        HPS_HIP_CHECK(hipStreamSynchronize(st));
        HPS_REQUIRE(ov == 0, "moving beam: a slice outgrew the hand-off capacity (twice the fullest injected slice) in the previous step");
    }
    if (moving && beam_import) {
        // the slices of this step arrive through hps_engine_import_beam_slice: every range empty, imports start at 0
        HPS_HIP_CHECK(hipMemsetAsync(d_B, 0, (size_t)(d.nz + 1)*sizeof(long), st));
        HPS_HIP_CHECK(hipMemsetAsync(d_Bimp, 0, (size_t)(d.nz + 1)*sizeof(long), st));
        HPS_HIP_CHECK(hipMemsetAsync(d_nfront, 0, (size_t)(d.nz + 2)*sizeof(int), st));
        ++steps_begun;
    } else if (moving) {
        if (steps_begun > 0) {
            // the hand-off between steps does not carry the sub-cycle counters (BeamParticleContainer.H:35-37);
            // whatever sits on a slice now is regular (MultiBuffer.cpp:809).  Absorbed particles keep nsub < 0.
            hipLaunchKernelGGL(k_beam_new_step, dim3(ceil_div(std::max(nbeam, 1L), 256)), dim3(256), 0, st, bm_nsub, nbeam);
            HPS_HIP_CHECK(hipMemsetAsync(d_nfront, 0, (size_t)(d.nz + 2)*sizeof(int), st));
        }
        HPS_HIP_CHECK(hipMemcpyAsync(h_B.data(), d_B, (size_t)(d.nz + 1)*sizeof(long), hipMemcpyDeviceToHost, st));
        HPS_HIP_CHECK(hipStreamSynchronize(st));
        ++steps_begun;
    }
    shift_pending = false;      // (the slab is cleared whole below)
    // ResetAllQuantities (Hipace.cpp:730-742)
    HPS_HIP_CHECK(hipMemsetAsync(slab.p, 0, (size_t)slab.nstride*ncomp*sizeof(double), st));
    // predictor-corrector: nothing but the cold plasma has been deposited in this sweep yet.  Not so from the first slice on
    // with a grid current, and with a second species as particles (or no neutralising background): there the serial path's
    // charges cancel to rounding only, it iterates on that residue itself, and the literal rule on the engine's residue is its
    // match (any non-zero byte pattern reads as "disturbed")
    if (d_pc_dist) HPS_HIP_CHECK(hipMemsetAsync(d_pc_dist, (d.grid_current_on || d.ion_on || d.plasma_no_neutralize) ? 1 : 0, sizeof(int), st));
    HPS_HIP_CHECK(hipMemsetAsync(d_checksum, 0, HPS_PC_NCOMP_MAX*sizeof(double), st));
    HPS_HIP_CHECK(hipMemsetAsync(d_laser_sum, 0, sizeof(double), st));
    if (int e = join_laser()) return e;        // the time levels rotate: the last slice's envelope must be in
    if (laser) { if (int e = laser_begin_step(*this)) return e; }
    if (d_insitu_pl) HPS_HIP_CHECK(hipMemsetAsync(d_insitu_pl, 0, (size_t)15*d.nz*sizeof(double), st));
    if (d_insitu_bm) HPS_HIP_CHECK(hipMemsetAsync(d_insitu_bm, 0, (size_t)23*d.nz*sizeof(double), st));
    if (d_insitu) HPS_HIP_CHECK(hipMemsetAsync(d_insitu, 0, (size_t)10*d.nz*sizeof(double), st));
    if (d_fd) HPS_HIP_CHECK(hipMemsetAsync(d_fd, 0, fd_comps.size()*fd_cells()*sizeof(double), st));
    step_index = (next_step >= 0) ? next_step : step_index + 1;      // Hipace::m_physical_time (PlasmaParticleContainerInit.cpp:90)
    next_step = -1;
    ahead_for = -2;
    // time factor of the density profile at z = c t of this step (UpdateDensityFunction, PlasmaParticleContainer.cpp:211-217)
    prof_ft = table_value(prof_t.data(), prof_f_t.data(), (int)prof_t.size(), gm.c*d.dt*step_index);
    if (int e = ionize_collect()) return e;
    np = np_init; pl.n = np; pl_alt.n = np;
    const double radius_sq = d.plasma_radius > 0.0 ? d.plasma_radius*d.plasma_radius : std::numeric_limits<double>::infinity();
    if (np > 0) {
        const int nppc = d.plasma_ppc[0]*d.plasma_ppc[1];
        hipLaunchKernelGGL(k_init_plasma, dim3(ceil_div(np, 256)), dim3(256), 0, st, pl, np, d.nx, d.ny,
                           d.plasma_ppc[0], d.plasma_ppc[1], d.lo[0], d.lo[1], gm.dx, gm.dy,
                           d.plasma_density*(d.si_units ? gm.dx*gm.dy*gm.dz/nppc : 1.0/nppc), 0, 0,     // scale_fac, PlasmaParticleContainerInit.cpp:40-41
                           d_prof_r, (int)prof_r.size(), prof_ft, radius_sq);
    }
    if (tiling) { if (int e = resort()) return e; }
    if (np > 0 && !d.plasma_no_neutralize) {
        // neutralising ion background, deposited once per step with charge -q (MultiPlasma.cpp:106-118)
        const int comp[6] = {-1, -1, -1, -1, -1, pc ? (int)HPS_PC_ION_RHOMJZ : (int)HPS_C_ION_RHOMJZ};
        if (tiling) {
            if (int e = deposit_current_tiled(slab, pl, gm, comp, -d.plasma_charge, d.plasma_mass, d.order, d.max_qsa, 0, d_nqsa, tiling, d_nfallback, st)) return e;
        } else {
            if (int e = hps_deposit_current(slab, pl, gm, comp, -d.plasma_charge, d.plasma_mass, d.order, d.max_qsa, 0, d_nqsa, st)) return e;
        }
    }
    if (ion.n > 0) {
        // the species "ion": every macro-ion back on its lattice point at its initial level; the product species is back
        // to its own InitParticles count
        const int inppc = d.ion_ppc[0]*d.ion_ppc[1];
        hipLaunchKernelGGL(k_init_plasma, dim3(ceil_div(ion.n, 256)), dim3(256), 0, st, ion.pl, ion.n, d.nx, d.ny,
                           d.ion_ppc[0], d.ion_ppc[1], d.lo[0], d.lo[1], gm.dx, gm.dy,
                           d.ion_density*(d.si_units ? gm.dx*gm.dy*gm.dz/inppc : 1.0/inppc), d.ion_init_level, 1,
                           d_prof_r, (int)prof_r.size(), prof_ft, radius_sq);
        if (ion.tiling) {
            ion.pl_alt.n = ion.n;
            if (int e = tiling_sort(ion.tiling, ion.pl, ion.pl_alt, gm, st)) return e;
            std::swap(ion.pl, ion.pl_alt); std::swap(ion.real, ion.real_alt);
            // every tile may hold charged ions if the species starts ionised; else none does until the push says so
            HPS_HIP_CHECK(hipMemsetAsync(ion.d_tile_flag, d.ion_init_level > 0 ? 0x01 : 0, (size_t)ion.tiling->g.ntiles*sizeof(int), st));
        }
        const unsigned long long c0[4] = {(unsigned long long)np_init, 0ULL, 0ULL, (unsigned long long)ion.n_ionized};
        HPS_HIP_CHECK(hipMemcpyAsync(ion.d_cnt, c0, sizeof(c0), hipMemcpyHostToDevice, st));
        HPS_HIP_CHECK(hipStreamSynchronize(st));      // c0 is on the stack
    }
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

// m_grid_current.DepositCurrentSlice (Hipace.cpp:629); z of the slice is plo[2] + islice*dz (GridCurrent.cpp:44)
void Engine::deposit_grid_current (int islice, int cjz)
{
    if (!d.grid_current_on) return;
    const double delta_z = (d.lo[2] + islice*gm.dz - d.grid_current_mean[2]) / d.grid_current_std[2];
    hipLaunchKernelGGL(k_grid_current, dim3(ceil_div(d.nx, 256), d.ny), dim3(256), 0, st, SlabView(slab), cjz, d.lo[0], d.lo[1], gm.dx,
                       gm.dy, d.grid_current_mean[0], d.grid_current_mean[1], d.grid_current_std[0], d.grid_current_std[1],
                       d.grid_current_peak*std::exp(-0.5*(delta_z*delta_z)));
}

int Engine::deposit_beam_slice (int islice, int cjx, int cjy, int cjz, const int* go)
{
    if (nbeam == 0 || islice < 0 || islice >= d.nz) return HPS_OK;
    if (moving) return beam_deposit_moving(*this, d.nz - 1 - islice, cjx, cjy, cjz);
    const long first0 = beam_off[d.nz - 1 - islice], count = beam_off[d.nz - islice] - first0;
    if (count <= 0) return HPS_OK;
    double* blk = beam_cur + 7*first0;
    const BeamView beam{blk, blk + count, blk + 2*count, blk + 3*count, blk + 4*count, blk + 5*count, blk + 6*count};
    const long first = 0;
    const double q_invvol = d.beam_charge*(d.si_units ? 1.0/(gm.dx*gm.dy*gm.dz) : 1.0);      // BeamDepositCurrent.cpp:70-83, level 0
    const double csq_inv = 1.0/(gm.c*gm.c);
    const dim3 grid(ceil_div(count, 256)), block(256);
    SlabView f(slab);
    switch (d.order) {
        case 0: hipLaunchKernelGGL(k_beam_deposit<0>, grid, block, 0, st, f, beam, first, count, cjx, cjy, cjz, q_invvol, csq_inv, 1.0/gm.dx, 1.0/gm.dy, gm.xoff, gm.yoff, go, d_pc_dist); break;
        case 1: hipLaunchKernelGGL(k_beam_deposit<1>, grid, block, 0, st, f, beam, first, count, cjx, cjy, cjz, q_invvol, csq_inv, 1.0/gm.dx, 1.0/gm.dy, gm.xoff, gm.yoff, go, d_pc_dist); break;
        case 2: hipLaunchKernelGGL(k_beam_deposit<2>, grid, block, 0, st, f, beam, first, count, cjx, cjy, cjz, q_invvol, csq_inv, 1.0/gm.dx, 1.0/gm.dy, gm.xoff, gm.yoff, go, d_pc_dist); break;
        default: hipLaunchKernelGGL(k_beam_deposit<3>, grid, block, 0, st, f, beam, first, count, cjx, cjy, cjz, q_invvol, csq_inv, 1.0/gm.dx, 1.0/gm.dy, gm.xoff, gm.yoff, go, d_pc_dist); break;
    }
    return HPS_OK;
}

hps_plasma Engine::tail_of (const hps_plasma& p, long first, long n) const
{
    hps_plasma t = p;
    double** arr[11] = {&t.x, &t.y, &t.w, &t.ux, &t.uy, &t.psi, &t.x_prev, &t.y_prev, &t.ux_half, &t.uy_half, &t.psi_half};
    for (int k = 0; k < 11; ++k) *arr[k] += first;      // (x_prev / y_prev stay aliased to x / y where they were)
    t.idcpu += first; t.ion_lev += first; t.n = n;
    return t;
}

// The three particle operators over one species: LDS-tile kernels over the tile-sorted body of the sheet; what has been
// appended behind it since the last sort (electrons released by the species "ion": some hundred particles) rides in the same
// launch on up to 64 extra workgroups (TailWork; HPS_FOLD_TAIL=0: per-particle kernels of their own, 10-40 us of latency
// chains per launch), anything beyond those goes through the per-particle kernels.
TailWork Engine::fold_tail_of (const hps_plasma& p, const Tiling* T, long margin, long* covered) const
{
    TailWork tw;
    *covered = T->sorted_n;
    if (!fold_tail || T != tiling || T->sorted_n <= 0) return tw;
    const long nt = p.n - T->sorted_n + margin;
    if (nt <= 0) return tw;
    tw.first = (int)T->sorted_n; tw.nwg = (int)std::min<long>(ceil_div(nt, 256), 64);
    *covered = T->sorted_n + 256L*tw.nwg;
    return tw;
}
int Engine::species_deposit (const hps_plasma& p, Tiling* T, const int comp[6], double charge, double mass, int can_ionize, const BeamPairWork* beam)
{
    if (p.n == 0) return HPS_OK;
    if (!T) return hps_deposit_current_laser(slab, p, gm, comp, c_aabs, charge, mass, d.order, d.max_qsa, can_ionize, d_nqsa, st);
    long covered; const TailWork tw = fold_tail_of(p, T, 0, &covered);
    if (T->sorted_n > 0) { if (int e = deposit_current_tiled(slab, p, gm, comp, charge, mass, d.order, d.max_qsa, can_ionize, d_nqsa, T, d_nfallback, st, c_aabs, T == ion.tiling ? ion.d_tile_flag : nullptr, tw, beam, nullptr, valid_by_w && !can_ionize)) return e; }
    if (p.n > covered) return hps_deposit_current_laser(slab, tail_of(p, covered, p.n - covered), gm, comp, c_aabs, charge, mass, d.order, d.max_qsa, can_ionize, d_nqsa, st);
    return HPS_OK;
}
int Engine::species_explicit (const hps_plasma& p, Tiling* T, const int cache[4], const int depos[2], double charge, double mass, int can_ionize)
{
    if (p.n == 0) return HPS_OK;
    if (!T) return hps_explicit_deposit_laser(slab, p, gm, cache, c_aabs, depos, charge, mass, d.order, d.deriv_type, can_ionize, st);
    long covered; const TailWork tw = fold_tail_of(p, T, 0, &covered);
    if (T->sorted_n > 0) { if (int e = explicit_deposit_tiled(slab, p, gm, cache, depos, charge, mass, d.order, d.deriv_type, can_ionize, T, d_nfallback, st, c_aabs, T == ion.tiling ? ion.d_tile_flag : nullptr, tw, valid_by_w && !can_ionize)) return e; }
    if (p.n > covered) return hps_explicit_deposit_laser(slab, tail_of(p, covered, p.n - covered), gm, cache, c_aabs, depos, charge, mass, d.order, d.deriv_type, can_ionize, st);
    return HPS_OK;
}
int Engine::species_advance (const hps_plasma& p, Tiling* T, const int comp[5], double charge, double mass, int temp_slice, int can_ionize)
{
    if (p.n == 0) return HPS_OK;
    if (!T) return hps_advance_plasma_laser(slab, p, gm, comp, c_aabs, charge, mass, d.order, temp_slice, d.n_subcycles, can_ionize, st);
    long covered; const TailWork tw = fold_tail_of(p, T, 0, &covered);
    if (T->sorted_n > 0) { if (int e = advance_plasma_tiled(slab, p, gm, comp, charge, mass, d.order, temp_slice, d.n_subcycles, can_ionize, T, d_nfallback, st, c_aabs, nullptr, nullptr, tw)) return e; }
    if (p.n > covered) return hps_advance_plasma_laser(slab, tail_of(p, covered, p.n - covered), gm, comp, c_aabs, charge, mass, d.order, temp_slice, d.n_subcycles, can_ionize, st);
    return HPS_OK;
}

// HPS_EVENT_FENCE: 0 = HIP's default system-scope release at every event, 1 (default) = none at the engine's own
// events (their consumers are streams of this device; measured: ring hand-off 14 us per slice cheaper),
// 2 = none at the ring's events either (no further gain)
unsigned event_flags (bool timing)
{
    static const int mode = [] { const char* v = std::getenv("HPS_EVENT_FENCE"); return v ? std::atoi(v) : 1; }();
    unsigned f = timing ? 0u : (unsigned)hipEventDisableTiming;
    if (mode >= 1) f |= timing ? (unsigned)hipEventReleaseToDevice : (unsigned)hipEventDisableSystemFence;
    return f;
}

int Engine::fork_laser ()
{
    if (laser_stream() == st) return HPS_OK;
    HPS_HIP_CHECK(hipEventRecord(ev_lfork, st));
    HPS_HIP_CHECK(hipStreamWaitEvent(st_laser, ev_lfork, 0));
    return HPS_OK;
}
int Engine::laser_done ()
{
    if (laser_stream() == st) return HPS_OK;
    HPS_HIP_CHECK(hipEventRecord(ev_ldone, st_laser));
    laser_pending = true;
    return HPS_OK;
}
int Engine::join_laser ()
{
    if (!laser_pending) return HPS_OK;
    HPS_HIP_CHECK(hipStreamWaitEvent(st, ev_ldone, 0));
    laser_pending = false;
    return HPS_OK;
}

void Engine::mark ()
{
    if (!prof_now) return;
    const int k = (int)(ev_used % 11);                  // which of the 11 marks of a slice this is
    if (ev_used == ev.size()) { hipEvent_t e; (void)hipEventCreateWithFlags(&e, event_flags(true)); ev.push_back(e); }
    // light mode: only the marks around the deposition kernel (2, 3) and the kernel-free pair (4, 5)
    if (!prof_light || (k >= 2 && k <= 5)) (void)hipEventRecord(ev[ev_used], st);
    ++ev_used;
}

// ---- field diagnostics (Fields::Copy, fields/Fields.cpp:413-533) ---------------------------------------------------
// F(i,j,k,n) += rel_z * sum_{iy,ix} sy[iy] sx[ix] slab(i_cell+ix, j_cell+iy, comp n), order-1 shape factors at the
// diagnostic cell centre, slab zero-extended beyond its guard cells (guarded_field_xy, Fields.cpp:331-358)
__global__ __launch_bounds__(256)
void k_diag_copy (SlabView f, int ncomp_slab, const int* __restrict__ comps, int ncd, double* __restrict__ F,
                  int nxc, int nyc, long kplane_off, long comp_stride, double rel_z,
                  double dxc, double dyc, double poff_dx, double poff_dy, double poff_cx, double poff_cy,
                  double dx_inv, double dy_inv)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= nxc) return;
    const double x = i*dxc + poff_dx, y = j*dyc + poff_dy;
    const double xmid = (x - poff_cx)*dx_inv, ymid = (y - poff_cy)*dy_inv;
    const int ic = (int)floor(xmid), jc = (int)floor(ymid);
    const double tx = xmid - ic, ty = ymid - jc;
    const double sx[2] = {1.0 - tx, tx}, sy[2] = {1.0 - ty, ty};
    for (int n = 0; n < ncd; ++n) {
        const int m = comps[n];
        double v = 0.0;
        for (int iy = 0; iy < 2; ++iy) for (int ix = 0; ix < 2; ++ix) {
            const int ii = ic + ix, jj = jc + iy;
            const bool in = ii >= -f.ng && ii < f.nx + f.ng && jj >= -f.ng && jj < f.ny + f.ng;
            v += sx[ix]*sy[iy]*(in ? f(ii, jj, m) : 0.0);
        }
        F[n*comp_stride + kplane_off + (long)j*nxc + i] += rel_z*v;
    }
    (void)ncomp_slab;
}

// Fields::InSituComputeDiags (fields/Fields.cpp:1288-1347): ten sums over the valid cells of one slice
__global__ __launch_bounds__(256)
void k_insitu_fields (SlabView f, double clight, double dxdydz, double* out, int nz, int islice)
{
    double s[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long cells = (long)f.nx*f.ny;
    for (long c = (long)blockIdx.x*blockDim.x + threadIdx.x; c < cells; c += (long)gridDim.x*blockDim.x) {
        const int j = (int)(c / f.nx), i = (int)(c - (long)j*f.nx);
        const long o = f.off(i, j);
        const double exmby = f.p[HPS_C_EXMBY*f.ns + o], eypbx = f.p[HPS_C_EYPBX*f.ns + o], ez = f.p[HPS_C_EZ*f.ns + o];
        const double bx = f.p[HPS_C_BX*f.ns + o], by = f.p[HPS_C_BY*f.ns + o], bz = f.p[HPS_C_BZ*f.ns + o];
        const double jzb = f.p[HPS_C_JZB*f.ns + o];
        const double ex = exmby + by*clight, ey = eypbx - bx*clight;
        s[0] += ex*ex; s[1] += ey*ey; s[2] += ez*ez; s[3] += bx*bx; s[4] += by*by; s[5] += bz*bz;
        s[6] += exmby*exmby; s[7] += eypbx*eypbx; s[8] += jzb; s[9] += ez*jzb;
    }
    __shared__ double part[4][10];
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        double v = s[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 10)
        atomic_add_f64(out + (long)threadIdx.x*nz + islice,
                       (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x])*dxdydz);
}

// PlasmaParticleContainer::InSituComputeDiags (particles/plasma/PlasmaParticleContainer.cpp:443-530): raw sums
__global__ __launch_bounds__(256)
void k_insitu_plasma (hps_plasma pl, double clight_inv, double radius_sq, double* out, int nz, int islice)
{
    double s[15];
#pragma unroll
    for (int q = 0; q < 15; ++q) s[q] = 0.0;
    for (long ip = (long)blockIdx.x*blockDim.x + threadIdx.x; ip < pl.n; ip += (long)gridDim.x*blockDim.x) {
        const double x = pl.x[ip], y = pl.y[ip];
        if (!(pl.idcpu[ip] & HPS_ID_VALID) || x*x + y*y > radius_sq) continue;
        const double ux = pl.ux[ip]*clight_inv, uy = pl.uy[ip]*clight_inv, psi = pl.psi[ip];
        const double gamma = (1.0 + ux*ux + uy*uy + psi*psi)/(2.0*psi);
        const double uz = gamma - psi;
        const double w = pl.w[ip]*gamma/psi;
        const double energy = pl.w[ip]*(gamma - 1.0);
        s[0] += w; s[1] += w*x; s[2] += w*x*x; s[3] += w*y; s[4] += w*y*y; s[5] += w*ux; s[6] += w*ux*ux;
        s[7] += w*uy; s[8] += w*uy*uy; s[9] += w*uz; s[10] += w*uz*uz; s[11] += w*gamma; s[12] += w*gamma*gamma;
        s[13] += energy; s[14] += 1.0;
    }
    __shared__ double part[4][15];
#pragma unroll
    for (int q = 0; q < 15; ++q) {
        double v = s[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 15)
        atomic_add_f64(out + (long)threadIdx.x*nz + islice,
                       part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// BeamParticleContainer::InSituComputeDiags (particles/beam/BeamParticleContainer.cpp:476-556): raw sums over the
// particles of the slice (slipped-in ones excluded, :494).  Static beam: count_static particles behind b; moving beam:
// the block [B[p] + nfront[p], B[p+1]) of the SoA, nsub < 0 = absorbed.
__global__ __launch_bounds__(256)
void k_insitu_beam (BeamView b, const int* __restrict__ nsub, const long* __restrict__ B, const int* __restrict__ nfront, int p,
                    long count_static, double clight_inv, double radius_sq, double* out, int nz, int islice)
{
    long first = 0, count = count_static;
    if (B) { first = B[p] + nfront[p]; count = B[p + 1] - first; }
    double s[23];
#pragma unroll
    for (int q = 0; q < 23; ++q) s[q] = 0.0;
    for (long t = (long)blockIdx.x*blockDim.x + threadIdx.x; t < count; t += (long)gridDim.x*blockDim.x) {
        const long ip = first + t;
        const double x = b.x[ip], y = b.y[ip], z = b.z[ip];
        if ((nsub && nsub[ip] < 0) || x*x + y*y > radius_sq) continue;
        const double ux = b.ux[ip]*clight_inv, uy = b.uy[ip]*clight_inv, uz = b.uz[ip]*clight_inv, w = b.w[ip];
        const double uz_inv = uz == 0.0 ? 0.0 : 1.0/uz;
        const double gamma = sqrt(1.0 + ux*ux + uy*uy + uz*uz);
        s[0] += w; s[1] += w*x; s[2] += w*x*x; s[3] += w*y; s[4] += w*y*y; s[5] += w*z; s[6] += w*z*z; s[7] += w*ux; s[8] += w*ux*ux;
        s[9] += w*uy; s[10] += w*uy*uy; s[11] += w*uz; s[12] += w*uz*uz; s[13] += w*x*ux; s[14] += w*y*uy; s[15] += w*z*uz;
        s[16] += w*x*uy; s[17] += w*y*ux; s[18] += w*ux*uz_inv; s[19] += w*uy*uz_inv; s[20] += w*gamma; s[21] += w*gamma*gamma;
        s[22] += 1.0;
    }
    __shared__ double part[4][23];
#pragma unroll
    for (int q = 0; q < 23; ++q) {
        double v = s[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 23) {
        const double v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (v != 0.0) atomic_add_f64(out + (long)threadIdx.x*nz + islice, v);
    }
}

// m_multi_beam.InSituComputeDiags (Hipace.cpp:681): after the field solves, before the beam push
void Engine::insitu_beam (int islice)
{
    if (!d_insitu_bm || nbeam == 0) return;
    const double r2 = insitu_bm_radius*insitu_bm_radius;
    if (moving) {
        const int p = d.nz - 1 - islice;
        const long bound = beam_bound(p);
        if (bound <= 0) return;
        const BeamView b{bm.x, bm.y, bm.z, bm.ux, bm.uy, bm.uz, bm.w};
        hipLaunchKernelGGL(k_insitu_beam, dim3((unsigned)std::min<long>(ceil_div(bound, 256), 64)), dim3(256), 0, st, b, bm.nsub, d_B, d_nfront, p,
                           0L, 1.0/gm.c, r2, d_insitu_bm, d.nz, islice);
    } else {
        const long first0 = beam_off[d.nz - 1 - islice], count = beam_off[d.nz - islice] - first0;
        if (count <= 0) return;
        double* blk = beam_cur + 7*first0;
        const BeamView b{blk, blk + count, blk + 2*count, blk + 3*count, blk + 4*count, blk + 5*count, blk + 6*count};
        hipLaunchKernelGGL(k_insitu_beam, dim3((unsigned)std::min<long>(ceil_div(count, 256), 64)), dim3(256), 0, st, b, (const int*)nullptr,
                           (const long*)nullptr, (const int*)nullptr, 0, count, 1.0/gm.c, r2, d_insitu_bm, d.nz, islice);
    }
}

int Engine::fill_field_diagnostic (int islice)
{
    if (fd_comps.empty()) return HPS_OK;
    const int nxc = fd_n[0], nyc = fd_n[1], nzc = fd_n[2];
    const double dzc = fd_h[2], dxc = fd_h[0], dyc = fd_h[1];
    // GetPosOffset (fields/Fields.H:71-77) of the calculation geometry; of the diagnostic one: fd_pos0 (position of its first cell)
    auto poff = [] (double lo, double hi, double h, int n) { return 0.5*(lo + hi - h*(n - 1)); };
    const double poff_cz = poff(d.lo[2], d.hi[2], gm.dz, d.nz), poff_dz = fd_pos0[2];
    const double poff_dx = fd_pos0[0], poff_dy = fd_pos0[1];
    SlabView f(slab);
    const long plane = (long)nxc*nyc, cstride = plane*nzc;
    if (fd_slice_dir == 2) {
        // diag_type xy: one plane that takes every slice inside the diagnostic's z range with the weight dz (Fields.cpp:469-479)
        const double pos_z = islice*gm.dz + poff_cz;
        if (!(fd_lo[2] <= pos_z && pos_z <= fd_hi[2])) return HPS_OK;
        hipLaunchKernelGGL(k_diag_copy, dim3(ceil_div(nxc, 256), nyc), dim3(256), 0, st, f, ncomp, d_fd_comps, (int)fd_comps.size(),
                           d_fd, nxc, nyc, 0L, cstride, gm.dz, dxc, dyc, poff_dx, poff_dy, gm.xoff, gm.yoff, 1.0/gm.dx, 1.0/gm.dy);
        return HPS_OK;
    }
    // which diagnostic planes this slice contributes to (order 1 in z, :428-468)
    const double pos_min = (islice - 1)*gm.dz + poff_cz, pos_max = (islice + 1)*gm.dz + poff_cz;
    const int k_min = (int)std::round((pos_min - poff_dz)*(1.0/dzc)), k_max = (int)std::round((pos_max - poff_dz)*(1.0/dzc));
    for (int k = std::max(k_min, 0); k <= std::min(k_max, nzc - 1); ++k) {
        const double pos = k*dzc + poff_dz;
        const double mid = (pos - poff_cz)*(1.0/gm.dz);
        const int kc = (int)std::floor(mid);
        const double t = mid - kc;
        double rel = 0.0;
        if (kc == islice) rel = 1.0 - t;
        if (kc + 1 == islice) rel = t;
        if (rel == 0.0) continue;
        hipLaunchKernelGGL(k_diag_copy, dim3(ceil_div(nxc, 256), nyc), dim3(256), 0, st, f, ncomp, d_fd_comps, (int)fd_comps.size(),
                           d_fd, nxc, nyc, (long)k*plane, cstride, rel, dxc, dyc, poff_dx, poff_dy, gm.xoff, gm.yoff,
                           1.0/gm.dx, 1.0/gm.dy);
    }
    return HPS_OK;
}

// ---- predictor-corrector Bx/By (Hipace::PredictorCorrectorLoopToSolveBxBy, Hipace.cpp:935-1031) ----------------

// Fields::ComputeRelBFieldError (fields/Fields.cpp:1233-1286): out[0] += sum |B|, out[1] += sum |B - B_iter| over
// the valid cells
// With `host` (mapped pinned memory) the last workgroup to finish posts {sum |B|, seq} {sum |B - B_iter|, seq} {fallback
// counter, seq} there as three 16-byte stores: the host polls the tags instead of a copy + stream synchronise.
// Loop control on the device (speculative iterations): `go` points at the word of iteration `it` in a per-slice array of
// flags (k_pc_guess: flag[1] = 1, the others 0); every kernel of an iteration does nothing when its word is 0.  The last
// workgroup here decides whether the loop goes on -- go[1] = (err > tol && it < max_it), the condition of Hipace.cpp:957 --
// and posts to slot `it` of the host array (4 doubles per slot; the halo-fallback counter also to slot 0, where the
// engine's re-sort rule reads it).
__global__ __launch_bounds__(256)
void k_rel_b_error (SlabView f, int cB, int cBit, double* out, volatile double* host, double seq,
                    int* go = nullptr, double tol = 0.0, int it = 0, int max_it = 0, double floor_b = 0.0)
{
    if (go && *go == 0) return;
    double sb = 0.0, sd = 0.0;
    const long cells = (long)f.nx*f.ny;
    for (long c = (long)blockIdx.x*blockDim.x + threadIdx.x; c < cells; c += (long)gridDim.x*blockDim.x) {
        const int j = (int)(c / f.nx), i = (int)(c - (long)j*f.nx);
        const long o = f.off(i, j);
        const double bx = f.p[cB*f.ns + o], by = f.p[(cB + 1)*f.ns + o];
        const double ex = bx - f.p[cBit*f.ns + o], ey = by - f.p[(cBit + 1)*f.ns + o];
        sb += sqrt(bx*bx + by*by);
        sd += sqrt(ex*ex + ey*ey);
    }
    for (int o = 32; o > 0; o >>= 1) { sb += __shfl_xor(sb, o); sd += __shfl_xor(sd, o); }
    __shared__ double part[8];
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = sb; part[4 + (threadIdx.x >> 6)] = sd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomic_add_f64(out, part[0] + part[1] + part[2] + part[3]);
        atomic_add_f64(out + 1, part[4] + part[5] + part[6] + part[7]);
        if (host) {
            HPS_OWN_ATOMICS_ACKNOWLEDGED();      // the two atomics above are acknowledged (no __threadfence(): see adk_post)
            unsigned int* done = reinterpret_cast<unsigned int*>(out + 3);
            if (atomicAdd(done, 1u) == gridDim.x - 1) {
                __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                double tb = __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double td = __hip_atomic_load(out + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // sum |B| at the rounding floor of a scatter with atomics counts as the exact zero the serial path has there
                // (Engine::pc_floor): the rule "relative error = 0 when sum |B| = 0" (fields/Fields.cpp:1283) then applies as on
                // the CPU; k_pc_mix reads the same word
                if (!(tb > floor_b)) { tb = 0.0; __hip_atomic_store(out, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                const double fbk = __hip_atomic_load(out + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // The post: three self-validating 16-byte stores {value, seq} -- no fence.  (A system-scope release fence ahead
                // of a lone sequence word writes the XCD's L2 back: the kernel took 9.4 us with it against 3.4 without the
                // post, once per loop iteration.)  The host waits until all three tags carry its sequence number.
                typedef double dbl2 __attribute__((ext_vector_type(2)));
                volatile dbl2* hs = reinterpret_cast<volatile dbl2*>(const_cast<double*>(host));
                if (go) {
                    const double err = tb > 0.0 ? td/tb : 0.0;
                    __hip_atomic_store(go + 1, (err > tol && it < max_it) ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hs[2] = dbl2{fbk, seq};       // slot 0: the fallback counter where the host's re-sort rule looks for it
                    hs += 4*it;
                }
                static_assert(sizeof(dbl2) == 16, "the post is one 16-byte store per {value, seq} pair");
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
                __threadfence_system();      // (no claim about single-copy atomicity of a 16-byte store to host memory elsewhere)
#endif
                hs[0] = dbl2{tb, seq}; hs[1] = dbl2{td, seq}; hs[2] = dbl2{fbk, seq};
            }
        }
    }
}

// InitialBfieldGuess (fields/Fields.cpp:1151-1173) + the set-up of the loop (Hipace.cpp:950-954): This = (1+m) Previous
// - m PCPrevIter with m = exp(-0.5 (err/(2.5 tol))^2), err = sums[1]/sums[0] of (Previous, PCPrevIter);
// PCIter = 0; PCPrevIter = This.  Whole planes incl. guards, both components.
__global__ __launch_bounds__(256)
void k_pc_guess (double* p, long ns, long plane, const double* sums, double tol, int* go, double floor_b)
{
    const long s = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (s < PC_MAX_SPEC && go) go[s] = (s == 1);      // the loop of this slice starts: its first iteration always runs (Hipace.cpp:956)
    if (s >= plane) return;
    const double err = sums[0] > floor_b ? sums[1]/sums[0] : 0.0;
    const double q = err/(2.5*tol);
    const double m = exp(-0.5*(q*q));
    for (int c = 0; c < 2; ++c) {
        const double b = (1.0 + m)*p[(HPS_PC_P_BX + c)*ns + s] + (-m)*p[(HPS_PC_PIT_BX + c)*ns + s];
        p[(HPS_PC_BX + c)*ns + s] = b;
        p[(HPS_PC_IT_BX + c)*ns + s] = 0.0;
        p[(HPS_PC_PIT_BX + c)*ns + s] = b;
    }
}

// sources of Fields::SolvePoissonBxBy (fields/Fields.cpp:1043-1064): st[0] = mu0 (-d_y jz + d_z jy),
// st[1] = mu0 (d_x jz - d_z jx), d_z = (Previous - Next)/(2 dz)
__global__ __launch_bounds__(256)
void k_rhs_bxby (SlabView f, double mu0, double hdx_inv, double hdy_inv, double hdz_inv, double* staging, long plane,
                 double* sums, const int* go)
{
    if (go && *go == 0) return;
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i == 0 && j == 0) { sums[0] = 0.0; sums[1] = 0.0; }      // for the k_rel_b_error of this pass
    if (i >= f.nx) return;
    const long o = f.off(i, j), so = (long)j*f.nx + i;
    const double* Z = f.p + HPS_PC_JZ*f.ns + o;
    staging[so] = (-mu0)*((Z[f.js] - Z[-f.js])*hdy_inv) + mu0*((f.p[HPS_PC_P_JY*f.ns + o] - f.p[HPS_PC_N_JY*f.ns + o])*hdz_inv);
    staging[plane + so] = mu0*((Z[1] - Z[-1])*hdx_inv) + (-mu0)*((f.p[HPS_PC_P_JX*f.ns + o] - f.p[HPS_PC_N_JX*f.ns + o])*hdz_inv);
}

// MixAndShiftBfields (fields/Fields.cpp:1175-1231) + the reset of the temporary currents (Hipace.cpp:1000-1003)
__global__ __launch_bounds__(256)
void k_pc_mix (double* p, long ns, long plane, const double* sums, double* err_slots, int it, double mix, const int* go,
               const int* disturbed = nullptr)
{
    if (go && *go == 0) return;      // (this iteration's own flag: its k_rel_b_error has written the NEXT one's)
    const long s = (long)blockIdx.x*blockDim.x + threadIdx.x;
    // nothing but the cold plasma's rounding residue has been deposited so far in this sweep: the serial path holds exact zeros
    const bool exact_zero = disturbed && __hip_atomic_load(disturbed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
    // the weights of Hipace.cpp:1005-1014 from this iteration's error and the previous one's, on the device: the kernel is
    // enqueued before the host has read the error.  err_slots[it & 1] <- err for the next iteration (nobody reads that slot now)
    const double err = sums[0] > 0.0 ? sums[1]/sums[0] : 0.0;
    const double err_prev = (it == 1) ? err : err_slots[(it - 1) & 1];
    double w_it = 0.5, w_prev = 0.5;
    if (err != 0.0 || err_prev != 0.0) { w_it = err_prev/(err + err_prev); w_prev = err/(err + err_prev); }
    if (s == 0) err_slots[it & 1] = err;
    if (s >= plane) return;
    for (int c = 0; c < 2; ++c) {
        const double it = p[(HPS_PC_IT_BX + c)*ns + s];
        const double mixed = w_it*it + w_prev*p[(HPS_PC_PIT_BX + c)*ns + s];
        p[(HPS_PC_BX + c)*ns + s] = exact_zero ? 0.0 : (1.0 - mix)*p[(HPS_PC_BX + c)*ns + s] + mix*mixed;
        p[(HPS_PC_PIT_BX + c)*ns + s] = it;
    }
    p[HPS_PC_N_JX*ns + s] = 0.0; p[HPS_PC_N_JY*ns + s] = 0.0;
}

// boundary.field = Open (Fields::SetBoundaryCondition, fields/Fields.cpp:678-735): the free-space Green's function
// ln|r - r'|^2 / (4 pi) expanded to order 18 about the origin (fields/OpenBoundary.H: 37 real moments; here in complex form,
// z = x + i y:  ln|z - z'|^2 = ln|z|^2 - sum_n (2/n) Re((z'/z)^n), so with M_n = sum_src s z'^n the potential outside the
// sources is dx dy/(4 pi) [M_0 ln|z|^2 - sum_n (2/n) Re(M_n z^-n)]).  Coordinates scaled by 3/|diagonal|; sources beyond 95 % of
// the distance to the nearest wall are left out.  k_multipole_moments: mom[b][workgroup][2 n], [2 n + 1] = that workgroup's share of
// Re, Im M_n of plane b.
constexpr int OPEN_ORDER = 18, OPEN_PARTS = 256;      // order of the expansion; workgroups (= partial sums) per source
__global__ __launch_bounds__(256)
void k_multipole_moments (const double* __restrict__ staging, long nval, int nx, double dx, double dy, double xoff, double yoff,
                          double scale, double cutoff_sq, double* mom)
{
    const double* s = staging + (long)blockIdx.y*nval;
    double re[OPEN_ORDER + 1], im[OPEN_ORDER + 1];
#pragma unroll
    for (int n = 0; n <= OPEN_ORDER; ++n) { re[n] = 0.0; im[n] = 0.0; }
    // (eight loads in flight per thread: with two or three waves per SIMD one load per trip leaves the memory latency bare)
    const long stride = (long)gridDim.x*blockDim.x;
    for (long c0 = (long)blockIdx.x*blockDim.x + threadIdx.x; c0 < nval; c0 += 8*stride) {
        double sv8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const long c = c0 + u*stride; sv8[u] = c < nval ? s[c] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long c = c0 + u*stride;
            if (c >= nval) break;
            const int j = (int)(c / nx), i = (int)(c - (long)j*nx);
            const double x = (i*dx + xoff)*scale, y = (j*dy + yoff)*scale;
            if (x*x + y*y > cutoff_sq) continue;
            const double sv = sv8[u];
            double zr = 1.0, zi = 0.0;
#pragma unroll
            for (int n = 0; n <= OPEN_ORDER; ++n) {
                re[n] += sv*zr; im[n] += sv*zi;
                const double t = zr*x - zi*y; zi = zr*y + zi*x; zr = t;
            }
        }
    }
    // the workgroup's sum through LDS (38 values x 256 threads; 38 x 6 wave-shuffle steps of doubles took longer than the pass):
    // six threads per value add 43 entries each, one of them the six
    constexpr int NM = 2*(OPEN_ORDER + 1), LD = 257;
    __shared__ double part[NM*LD];
#pragma unroll
    for (int n = 0; n <= OPEN_ORDER; ++n) { part[(2*n)*LD + threadIdx.x] = re[n]; part[(2*n + 1)*LD + threadIdx.x] = im[n]; }
    __syncthreads();
    const int v = threadIdx.x / 6, q = threadIdx.x - 6*v;
    double a = 0.0;
    if (v < NM) for (int k = q; k < 256; k += 6) a += part[v*LD + k];
    __syncthreads();
    if (v < NM) part[v*LD + q] = a;
    __syncthreads();
    // (one partial sum per workgroup, added up by the consumer in a fixed order)
    if (threadIdx.x < NM) {
        const double* r = part + threadIdx.x*LD;
        mom[((long)blockIdx.y*gridDim.x + blockIdx.x)*NM + threadIdx.x] = ((r[0] + r[1]) + (r[2] + r[3])) + (r[4] + r[5]);
    }
}

// SetDirichletBoundaries (fields/Fields.cpp:627-669) with offset = factor = 1: the outermost rows and columns of the source get
// -phi(one cell outside)/h^2, corners from both sides.  One thread per boundary point: 2 nx + 2 ny of them per plane.
__global__ __launch_bounds__(256)
void k_open_boundary_apply (double* staging, long nval, int nx, int ny, double dx, double dy, double xoff, double yoff,
                            double scale, double pref, const double* __restrict__ mom, int nparts, unsigned no_monopole_mask)
{
    const int b = blockIdx.y;
    constexpr int NM = 2*(OPEN_ORDER + 1);
    __shared__ double Msh[6][NM];
    {   // the workgroups' partial moments, in a fixed order
        const int v = threadIdx.x % NM, q = threadIdx.x / NM;
        if (q < 6) {
            double a = 0.0;
            for (int k = q; k < nparts; k += 6) a += mom[((long)b*nparts + k)*NM + v];
            Msh[q][v] = a;
        }
        __syncthreads();
        if (threadIdx.x < NM) Msh[0][threadIdx.x] = ((Msh[0][threadIdx.x] + Msh[1][threadIdx.x]) + (Msh[2][threadIdx.x] + Msh[3][threadIdx.x]))
                                                    + (Msh[4][threadIdx.x] + Msh[5][threadIdx.x]);
        __syncthreads();
    }
    const int t = blockIdx.x*blockDim.x + threadIdx.x;
    if (t >= 2*nx + 2*ny) return;
    double xd, yd, hh; long target;
    if (t < nx)               { xd = t*dx + xoff;            yd = -dy + yoff;              hh = dy*dy; target = t; }
    else if (t < 2*nx)        { xd = (t - nx)*dx + xoff;     yd = ny*dy + yoff;            hh = dy*dy; target = (long)(ny - 1)*nx + (t - nx); }
    else if (t < 2*nx + ny)   { xd = -dx + xoff;             yd = (t - 2*nx)*dy + yoff;    hh = dx*dx; target = (long)(t - 2*nx)*nx; }
    else                      { xd = nx*dx + xoff;           yd = (t - 2*nx - ny)*dy + yoff; hh = dx*dx; target = (long)(t - 2*nx - ny)*nx + (nx - 1); }
    const double* M = Msh[0];
    const double zx = xd*scale, zy = yd*scale, nrm = zx*zx + zy*zy;
    const double ix = zx/nrm, iy = -zy/nrm;                   // 1/z
    double v = ((no_monopole_mask >> b) & 1u) ? 0.0 : M[0]*log(nrm);
    double pr = ix, pi_ = iy;
    for (int n = 1; n <= OPEN_ORDER; ++n) {
        v -= (2.0/n)*(M[2*n]*pr - M[2*n + 1]*pi_);
        const double q = pr*ix - pi_*iy; pi_ = pr*iy + pi_*ix; pr = q;
    }
    atomic_add_f64(staging + (long)b*nval + target, -(pref*v)/hh);
}

int Engine::open_boundary (int nbatch, unsigned no_monopole_mask)
{
    if (d.field_bc != 1) return HPS_OK;
    const long nval = (long)d.nx*d.ny;
    const double Lx = d.hi[0] - d.lo[0], Ly = d.hi[1] - d.lo[1];
    const double scale = 3.0/std::sqrt(Lx*Lx + Ly*Ly);
    const double radius = std::min(std::min(std::fabs(d.lo[0]), std::fabs(d.hi[0])), std::min(std::fabs(d.lo[1]), std::fabs(d.hi[1])));
    const double cutoff_sq = (0.95*radius*scale)*(0.95*radius*scale);
    hipLaunchKernelGGL(k_multipole_moments, dim3(OPEN_PARTS, nbatch), dim3(256), 0, st, staging, nval, d.nx, gm.dx, gm.dy, gm.xoff, gm.yoff,
                       scale, cutoff_sq, d_open_mom);
    hipLaunchKernelGGL(k_open_boundary_apply, dim3(ceil_div(2*d.nx + 2*d.ny, 256), nbatch), dim3(256), 0, st, staging, nval, d.nx, d.ny,
                       gm.dx, gm.dy, gm.xoff, gm.yoff, scale, gm.dx*gm.dy/(4.0*3.14159265358979323846), d_open_mom, OPEN_PARTS, no_monopole_mask);
    return HPS_OK;
}

// SolveOneSlice with hipace.bxby_solver = predictor-corrector (Hipace.cpp:556-728; the explicit branch is
// Engine::solve_slice below).  Event marks keep the 10 intervals of the explicit schedule: the loop is booked under
// the Bx/By-solve interval.
int Engine::solve_slice_pc_begin (int islice)
{
    SlabView f(slab);
    const long plane = slab.nstride;
    const dim3 b256(256);
    const dim3 gplane(ceil_div(plane, 256));
    const long nval = (long)d.nx*d.ny;
    int e;

    prof_now = profiling && (slices_done % prof_stride == 0);
    mark();   // b0
    if (d_insitu_pl && np > 0)
        hipLaunchKernelGGL(k_insitu_plasma, dim3(256), b256, 0, st, pl, 1.0/gm.c, insitu_pl_radius*insitu_pl_radius, d_insitu_pl, d.nz, islice);
    // InitializeSlices (fields/Fields.cpp:565-570); ExmBy, EypBx are rewritten by k_grad_psi up to the outermost
    // guard ring, which stays zero from begin_step
    {   CompList z{0, {}}, zb{0, {}};
        for (int c : {HPS_PC_JX, HPS_PC_JY, HPS_PC_JZ, HPS_PC_RHOMJZ}) z.c[z.n++] = c;
        if (d.deposit_rho) z.c[z.n++] = HPS_PC_RHO;
        hipLaunchKernelGGL(k_zero_comps, gplane, b256, 0, st, slab.p, slab.nstride, plane, (int)slab.jstride, z, zb, CellBox{0, -1, 0, -1}); }
    mark();   // b1
    if (tiling && (since_sort >= sort_period || (since_sort >= 2 && *h_nfallback - fb_at_sort > np/fallback_div) ||
                   (since_sort >= 1 && np - tiling->sorted_n > std::max(np/32, 16384L)))) { if ((e = resort())) return e; }
    ++since_sort;
    mark();   // b1b
    // plasma: jx jy jz [rho] rhomjz (Hipace.cpp:616-618); beams deposit into the same jx jy jz (:620-623)
    {   const int comp[6] = {HPS_PC_JX, HPS_PC_JY, HPS_PC_JZ, d.deposit_rho ? HPS_PC_RHO : -1, -1, HPS_PC_RHOMJZ};
        // (with an ionisable species the sheet has electrons behind its tile-sorted body: the helper that also covers that tail)
        if (ion.n > 0) { if ((e = species_deposit(pl, tiling, comp, d.plasma_charge, d.plasma_mass, 0))) return e; }
        else if (tiling) { if ((e = deposit_current_tiled(slab, pl, gm, comp, d.plasma_charge, d.plasma_mass, d.order, d.max_qsa, 0, d_nqsa, tiling, d_nfallback, st, -1, nullptr, TailWork{}, nullptr, nullptr, valid_by_w))) return e; }
        else        { if ((e = hps_deposit_current(slab, pl, gm, comp, d.plasma_charge, d.plasma_mass, d.order, d.max_qsa, 0, d_nqsa, st))) return e; }
        // MultiPlasma::DepositCurrent: every species in turn (MultiPlasma.cpp:78-87)
        if (ion.n > 0) { if ((e = species_deposit(ion.pl, ion.tiling, comp, d.ion_charge, d.ion_mass, 1))) return e; } }
    mark();   // b2
    if ((e = deposit_beam_slice(islice, HPS_PC_JX, HPS_PC_JY, HPS_PC_JZ))) return e;
    deposit_grid_current(islice, HPS_PC_JZ);
    {   const double fa = 1.0/(gm.ep0*gm.c);
        hipLaunchKernelGGL(k_rhs_all, dim3(ceil_div(slab.jstride, 256), d.ny + 2*g), b256, 0, st, f, HPS_PC_RHOMJZ,
                           HPS_PC_ION_RHOMJZ, d.deposit_rho ? HPS_PC_RHO : -1, HPS_PC_JX, HPS_PC_JY, 1.0/gm.ep0,
                           fa*0.5*(1.0/gm.dx), fa*0.5*(1.0/gm.dy), gm.mu0*0.5*(1.0/gm.dy), -gm.mu0*0.5*(1.0/gm.dx),
                           staging, nval);
        const int comps[3] = {HPS_PC_PSI, HPS_PC_EZ, HPS_PC_BZ};
        if ((e = open_boundary(3, 0x6u))) return e;      // (Ez, Bz: no physical monopole, fields/Fields.cpp:727-731)
        if ((e = hps_poisson_solve_batch(ps, 3, staging, slab, comps, st))) return e; }
    hipLaunchKernelGGL(k_grad_psi, dim3(ceil_div(d.nx + 2*(g - 1), 256), d.ny + 2*(g - 1)), b256, 0, st, f, HPS_PC_PSI,
                       HPS_PC_EXMBY, HPS_PC_EYPBX, 0.5*(1.0/gm.dx), 0.5*(1.0/gm.dy));
    mark();   // b3
    mark();   // b4
    mark();   // b5
    // the loop (Hipace.cpp:935-1031)
    HPS_HIP_CHECK(hipMemsetAsync(d_pc, 0, 2*sizeof(double), st));
    hipLaunchKernelGGL(k_rel_b_error, dim3(128), b256, 0, st, f, HPS_PC_P_BX, HPS_PC_PIT_BX, d_pc, (volatile double*)nullptr, 0.0);
    const bool spec = pc_speculate && tiling && !moving && ion.n == 0;      // (a second species: the host-controlled loop)
    hipLaunchKernelGGL(k_pc_guess, gplane, b256, 0, st, slab.p, slab.nstride, plane, d_pc, pc_tol, spec ? d_pc_go : (int*)nullptr, pc_floor);
    pc_islice = islice;
    if (spec) {
        // Device-side loop control: as many iterations as the previous slice took are enqueued at once, every kernel of
        // iteration `it` looking at the flag its predecessor's error kernel has written (k_rel_b_error): the queue never
        // runs dry while an error travels to the host and the next iteration's ten launches travel back.  The host reads
        // the slots in solve_slice_pc_finish and adds iterations one by one only if the speculated ones did not suffice.
        pc_base_seq = pc_seq;
        pc_enqueued = 0;
        const int K = std::max(1, std::min(pc_spec_iters, pc_max_iter));
        for (int it = 1; it <= K; ++it) {
            if ((e = pc_enqueue_iteration(it))) {
                // a failed launch with the go flags armed: nothing of this slice's loop is to be mistaken for the next slice's --
                // sequence numbers move past whatever the iterations enqueued so far may still post
                pc_seq = pc_base_seq + pc_max_iter + 2; pc_enqueued = 0; pc_islice = -1;
                return e;
            }
        }
        return HPS_OK;
    }
    double err = 1.0;
    int it = 0;
    // (host-controlled loop.)  The mixing weights of an iteration are computed on the device (k_pc_mix) and the mixing is
    // enqueued before the host waits for the iteration's error: it runs while the error travels (config 2: +5 % iterations
    // per second).
    while (err > pc_tol && it < pc_max_iter) {
        ++it; ++pc_iterations;
        pc_enqueued = it - 1;
        if ((e = pc_enqueue_iteration(it))) return e;
        if ((e = pc_wait_slot(0, pc_seq))) return e;
        err = h_pc[0] > 0.0 ? h_pc[2]/h_pc[0] : 0.0;
    }
    pc_last_err = err;
    return HPS_OK;
}

// one iteration of the loop on the engine's stream (Hipace.cpp:958-1030); with device-side control every launch is gated
// on the iteration's flag
int Engine::pc_enqueue_iteration (int it)
{
    SlabView f(slab);
    const long plane = slab.nstride;
    const dim3 b256(256);
    const dim3 gplane(ceil_div(plane, 256));
    const long nval = (long)d.nx*d.ny;
    const int islice = pc_islice;
    const bool spec = pc_speculate && tiling && !moving && ion.n == 0;
    int* go = spec ? d_pc_go + it : nullptr;
    const int comp_push[5] = {HPS_PC_PSI, HPS_PC_EZ, HPS_PC_BX, HPS_PC_BY, HPS_PC_BZ};
    int e;
    // plasma to the temporary next slice, its jx jy (+ the beam's) there
    if (ion.n > 0) { if ((e = species_advance(pl, tiling, comp_push, d.plasma_charge, d.plasma_mass, 1, 0))) return e; }
    else if (tiling) { if ((e = advance_plasma_tiled(slab, pl, gm, comp_push, d.plasma_charge, d.plasma_mass, d.order, 1, d.n_subcycles, 0, tiling, d_nfallback, st, -1, nullptr, go))) return e; }
    else        { if ((e = hps_advance_plasma(slab, pl, gm, comp_push, d.plasma_charge, d.plasma_mass, d.order, 1, d.n_subcycles, 0, st))) return e; }
    if (ion.n > 0) { if ((e = species_advance(ion.pl, ion.tiling, comp_push, d.ion_charge, d.ion_mass, 1, 1))) return e; }      // (go == nullptr with a second species)
    {   const int comp[6] = {HPS_PC_N_JX, HPS_PC_N_JY, -1, -1, -1, -1};
        if (ion.n > 0) { if ((e = species_deposit(pl, tiling, comp, d.plasma_charge, d.plasma_mass, 0))) return e; }
        else if (tiling) { if ((e = deposit_current_tiled(slab, pl, gm, comp, d.plasma_charge, d.plasma_mass, d.order, d.max_qsa, 0, d_nqsa, tiling, d_nfallback, st, -1, nullptr, TailWork{}, nullptr, go, valid_by_w))) return e; }
        else        { if ((e = hps_deposit_current(slab, pl, gm, comp, d.plasma_charge, d.plasma_mass, d.order, d.max_qsa, 0, d_nqsa, st))) return e; }
        if (ion.n > 0) { if ((e = species_deposit(ion.pl, ion.tiling, comp, d.ion_charge, d.ion_mass, 1))) return e; } }
    if ((e = deposit_beam_slice(islice - 1, HPS_PC_N_JX, HPS_PC_N_JY, -1, go))) return e;
    hipLaunchKernelGGL(k_rhs_bxby, dim3(ceil_div(d.nx, 256), d.ny), b256, 0, st, f, gm.mu0, 0.5*(1.0/gm.dx), 0.5*(1.0/gm.dy),
                       0.5*(1.0/gm.dz), staging, nval, d_pc, (const int*)go);
    {   const int comps[2] = {HPS_PC_IT_BX, HPS_PC_IT_BY};
        if ((e = open_boundary(2, 0u))) return e;
        poisson_set_gate(ps, go);
        e = hps_poisson_solve_batch(ps, 2, staging, slab, comps, st);
        poisson_set_gate(ps, nullptr);
        if (e) return e; }
    pc_seq += 1.0;
    hipLaunchKernelGGL(k_rel_b_error, dim3(128), b256, 0, st, f, HPS_PC_BX, HPS_PC_IT_BX, d_pc, (volatile double*)h_pc_dev, pc_seq,
                       go, pc_tol, it, pc_max_iter, pc_floor);
    hipLaunchKernelGGL(k_pc_mix, gplane, b256, 0, st, slab.p, slab.nstride, plane, d_pc, d_pc_aux, it, pc_mix, (const int*)go, (const int*)d_pc_dist);
    pc_enqueued = it;
    return HPS_OK;
}

// wait for the post of sequence number `seq` in host slot `slot`; fall back to the stream's status every so often so that a
// failed launch cannot hang us
int Engine::pc_wait_slot (int slot, double seq)
{
    volatile double* hp = h_pc + 8*slot;          // {sum |B|, seq} {sum |B - B_iter|, seq} {fallback counter, seq} (k_rel_b_error)
    auto there = [&] () { return hp[1] == seq && hp[3] == seq && hp[5] == seq; };
    long spins = 0;
    while (!there()) {
        if ((++spins & 0xfffff) == 0 && hipStreamQuery(st) != hipErrorNotReady) {
            if (there()) break;
            HPS_HIP_CHECK(hipStreamSynchronize(st));
            if (!there()) { set_error("predictor-corrector: the error read-back never arrived"); return HPS_ERR_HIP; }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return HPS_OK;
}

// second half of a predictor-corrector slice: the loop's end (device-side control: read the posted errors, add iterations if
// the speculated ones did not suffice), diagnostics, the committing push, ShiftSlices
int Engine::solve_slice_pc_finish (int islice)
{
    SlabView f(slab);
    const long plane = slab.nstride;
    const dim3 b256(256);
    const dim3 gplane(ceil_div(plane, 256));
    const int comp_push[5] = {HPS_PC_PSI, HPS_PC_EZ, HPS_PC_BX, HPS_PC_BY, HPS_PC_BZ};
    int e;
    double err = pc_last_err;
    if (pc_speculate && tiling && !moving && ion.n == 0) {
        int it = 0;
        err = 1.0;
        while (err > pc_tol && it < pc_max_iter) {
            ++it; ++pc_iterations;
            if (it > pc_enqueued) { if ((e = pc_enqueue_iteration(it))) return e; }
            if ((e = pc_wait_slot(it, pc_base_seq + it))) return e;
            const volatile double* hp = h_pc + 8*it;
            err = hp[0] > 0.0 ? hp[2]/hp[0] : 0.0;
            if (!(hp[0] > 0.0) && it == 1) ++pc_zero_b_slices;      // sum |B| exactly 0 or under Engine::pc_floor: the loop leaves after this pass
        }
        // (iterations enqueued beyond `it` find their flag at 0 and do nothing; their sequence numbers are never posted)
        pc_spec_iters = it;
        pc_seq = pc_base_seq + std::max(it, pc_enqueued);
    }
    pc_err_sum += err;
    mark();   // b6
    if (diagnostics)
        hipLaunchKernelGGL(k_checksum, dim3(64, ncomp), b256, 0, st, f, ncomp, d_checksum);
    if ((e = fill_field_diagnostic(islice))) return e;
    mark();   // b7
    if (ion.n > 0) {
        // DoFieldIonization (Hipace.cpp:693-696) and the committing pushes of both species (:699-701): the decisions on the tiles'
        // gathered fields inside the ions' push (or k_ionize ahead of the per-particle push), the electron count back on the host,
        // then every electron -- the ones this slice has released included
        if (ion.tiling) {
            if ((e = ion_field_bounds())) return e;
            const IonArgs ia = ion_args(islice);
            if ((e = advance_plasma_tiled(slab, ion.pl, gm, comp_push, d.ion_charge, d.ion_mass, d.order, 0, d.n_subcycles, 1, ion.tiling, d_nfallback, st, -1, &ia))) return e;
        } else {
            if ((e = ionize_slice(islice))) return e;
            if ((e = species_advance(ion.pl, ion.tiling, comp_push, d.ion_charge, d.ion_mass, 0, 1))) return e;
        }
        if ((e = ionize_collect())) return e;
        if ((e = species_advance(pl, tiling, comp_push, d.plasma_charge, d.plasma_mass, 0, 0))) return e;
    }
    else if (tiling) { if ((e = advance_plasma_tiled(slab, pl, gm, comp_push, d.plasma_charge, d.plasma_mass, d.order, 0, d.n_subcycles, 0, tiling, d_nfallback, st))) return e; }
    else        { if ((e = hps_advance_plasma(slab, pl, gm, comp_push, d.plasma_charge, d.plasma_mass, d.order, 0, d.n_subcycles, 0, st))) return e; }
    insitu_beam(islice);
    if (moving && nbeam > 0) { if ((e = beam_push_moving(*this, islice))) return e; }
    mark();   // b8
    // ShiftSlices (fields/Fields.cpp:600-603): PCPrevIter <- Previous <- This for Bx By, Previous <- This for jx jy
    {   CompList dst{6, {HPS_PC_PIT_BX, HPS_PC_PIT_BY, HPS_PC_P_BX, HPS_PC_P_BY, HPS_PC_P_JX, HPS_PC_P_JY}};
        CompList src{6, {HPS_PC_P_BX, HPS_PC_P_BY, HPS_PC_BX, HPS_PC_BY, HPS_PC_JX, HPS_PC_JY}};
        hipLaunchKernelGGL(k_copy_comps, gplane, b256, 0, st, slab.p, slab.nstride, plane, dst, src); }
    mark();   // b9
    HPS_HIP_CHECK(hipGetLastError());
    ++slices_done;
    return HPS_OK;
}

// The slice in two halves for a host that drives several engines from one thread (several time steps in flight on one
// device, hps_engine_solve_slice_begin / _finish): begin enqueues everything up to and including the speculated V-cycles of
// the Bx/By solve (and the gated push behind them) and returns without waiting for the device; finish waits for the
// solve's norms -- the one host wait of a slice -- and enqueues the rest.  solve_slice = begin + finish.
void Engine::flush_shift ()
{
    if (!shift_pending) return;
    shift_pending = false;
    const CellBox bb{beam_box.ilo, beam_box.ihi, beam_box.jlo, beam_box.jhi};
    hipLaunchKernelGGL(k_shift_slices, dim3(ceil_div(slab.nstride, 256)), dim3(256), 0, st, slab.p, slab.nstride, slab.nstride, (int)slab.jstride, bb);
}

int Engine::solve_slice (int islice)
{
    if (int e = solve_slice_begin(islice)) return e;
    return solve_slice_finish(islice);
}

int Engine::solve_slice_begin (int islice)
{
    HPS_REQUIRE(pending_slice == -1, "hps_engine_solve_slice_begin: the previous slice has not been finished");
    if (pc) { if (int e = solve_slice_pc_begin(islice)) return e; pending_slice = islice; return HPS_OK; }
    SlabView f(slab);
    const long plane = slab.nstride;
    const dim3 b256(256);
    const dim3 gplane(ceil_div(plane, 256));
    const dim3 gvalid(ceil_div(d.nx, 256), d.ny);
    int e;

    prof_now = profiling && (slices_done % prof_stride == 0);
    mg_solve1_forget_hierarchy(mg);            // (left set only if the previous slice failed between prepare and begin)
    if ((e = join_laser())) return e;          // the envelope solver of the previous slice has read its chi
    mark();   // b0
    if (d_insitu_pl && np > 0)        // m_multi_plasma.InSituComputeDiags (Hipace.cpp:590)
        hipLaunchKernelGGL(k_insitu_plasma, dim3(256), b256, 0, st, pl, 1.0/gm.c, insitu_pl_radius*insitu_pl_radius, d_insitu_pl, d.nz, islice);
    // InitializeSlices (fields/Fields.cpp:535-586)
    const CellBox bb{beam_box.ilo, beam_box.ihi, beam_box.jlo, beam_box.jhi};
    // the previous slice's fused push + deposition has already shifted / zeroed the slab and deposited this slice's
    // plasma currents (k_shift_zero + k_advance_deposit_tiled)
    const bool ahead = (ahead_for == islice);
    ahead_for = -2;
    // ShiftSlices of the slice before, left pending at its end (lazy_shift), and this slice's InitializeSlices in one pass
    bool zeroed = false;
    if (shift_pending) {
        shift_pending = false;
        if (!ahead) { hipLaunchKernelGGL(k_shift_zero, gplane, b256, 0, st, slab.p, slab.nstride, plane, (int)slab.jstride, bb, d.deposit_rho ? (int)HPS_C_RHO : -1, (int)HPS_C_ION_RHOMJZ); zeroed = true; }
        else hipLaunchKernelGGL(k_shift_slices, gplane, b256, 0, st, slab.p, slab.nstride, plane, (int)slab.jstride, bb);
    }
    if (!ahead && !zeroed)
    {   CompList z{0, {}}, zb{0, {}};
        // Sx, Sy are written as whole planes by k_sxsy_beam; ExmBy, EypBx by k_grad_psi up to the outermost
        // guard ring, which nothing ever writes (it keeps the zeros of begin_step); the beam planes only
        // ever hold values inside the beam's box
        for (int c : {HPS_C_CHI, HPS_C_RHOMJZ}) z.c[z.n++] = c;
        if (d.deposit_rho) z.c[z.n++] = HPS_C_RHO;
        for (int c : {HPS_C_JZB, HPS_C_N_JXB, HPS_C_N_JYB}) zb.c[zb.n++] = c;
        CompList fi{0, {}};
        fi.c[fi.n++] = HPS_C_RHOMJZ; if (d.deposit_rho) fi.c[fi.n++] = HPS_C_RHO;
        hipLaunchKernelGGL(k_zero_comps, gplane, b256, 0, st, slab.p, slab.nstride, plane, (int)slab.jstride, z, zb, bb, fi, (int)HPS_C_ION_RHOMJZ); }
    if (c_aabs >= 0) {
        // UpdateLaserAabs (Hipace.cpp:603)
        if ((e = laser_update_aabs(*this, islice, diagnostics ? d_laser_sum : nullptr))) return e;
    }
    // the auxiliary stream may start on the beam's planes (zeroed above)
    const bool aux = aux_on && st_aux && !pc && !ahead && !prof_now;
    if (aux) { HPS_HIP_CHECK(hipEventRecord(ev_aux[0], st)); HPS_HIP_CHECK(hipStreamWaitEvent(st_aux, ev_aux[0], 0)); }

    mark();   // b1
    // plasma: jx, jy, [rho], chi, rhomjz (Hipace.cpp:609-610); beam: jz_beam on This (:613-614)
    // re-sort after sort_period slices at the latest, earlier once more than 1/256 of the sheet has left
    // the halo of its tile (h_nfallback is as of the previous slice's multigrid sync)
    // (with a species "ion": also once the electrons appended behind the sorted body since the last sort -- they run
    // through the per-particle kernels -- are more than 1/32 of the sheet)
    if (tiling && (since_sort >= sort_period || (since_sort >= 2 && *h_nfallback - fb_at_sort > np/fallback_div) ||
                   (since_sort >= 1 && np - tiling->sorted_n > std::max(np/32, 16384L)))) { if ((e = resort())) return e; }
    ++since_sort;
    mark();   // b1b
    // static beam: jz of this slice and jx, jy of the next one in one go (Hipace.cpp:613-614, 656-657), as extra workgroups of
    // the plasma's deposition where that runs on tiles; a moving beam keeps the two calls (the next slice's block is only
    // final once this slice's push has handed its slipped particles on -- it is deposited after the solves below as before)
    const bool pair = !moving && nbeam > 0;
    BeamPairWork bw;
    if (pair) {
        const long fA = beam_off[d.nz - 1 - islice], cA = beam_off[d.nz - islice] - fA;
        const long fB = islice >= 1 ? beam_off[d.nz - islice] : 0, cB = islice >= 1 ? beam_off[d.nz - islice + 1] - fB : 0;
        double* pa = beam_cur + 7*fA; double* pb = beam_cur + 7*fB;
        bw.a = BeamView{pa, pa + cA, pa + 2*cA, pa + 3*cA, pa + 4*cA, pa + 5*cA, pa + 6*cA};
        bw.b = BeamView{pb, pb + cB, pb + 2*cB, pb + 3*cB, pb + 4*cB, pb + 5*cB, pb + 6*cB};
        bw.ca = cA; bw.cb = cB; bw.nba = (int)ceil_div(cA, 256); bw.nwg = bw.nba + (int)ceil_div(cB, 256);
        bw.cjz = HPS_C_JZB; bw.cjxn = HPS_C_N_JXB; bw.cjyn = HPS_C_N_JYB;
        bw.q_invvol = d.beam_charge*(d.si_units ? 1.0/(gm.dx*gm.dy*gm.dz) : 1.0); bw.csq_inv = 1.0/(gm.c*gm.c);
    }
    const bool beam_folded = pair && fold_beam && bw.nwg > 0 && !ahead && np > 0 && tiling && tiling->sorted_n > 0;
    if (!ahead)
    {   const int comp[6] = {HPS_C_JX, HPS_C_JY, -1, d.deposit_rho ? HPS_C_RHO : -1, HPS_C_CHI, HPS_C_RHOMJZ};
        if ((e = species_deposit(pl, tiling, comp, d.plasma_charge, d.plasma_mass, 0, beam_folded ? &bw : nullptr))) return e;
        // MultiPlasma::DepositCurrent: every species in turn (MultiPlasma.cpp:78-87); an ion weighs in with its level
        if (ion.n > 0) { if ((e = species_deposit(ion.pl, ion.tiling, comp, d.ion_charge, d.ion_mass, 1))) return e; } }
    mark();   // b2
    if (pair) {
        if (!beam_folded && bw.nwg > 0) {
            // (beside the plasma's deposition and the Poisson solves: nothing ahead of the Sx/Sy initialisation reads the beam's planes)
            hipStream_t sb = (aux && !d.grid_current_on) ? st_aux : st;
#define HPS_PAIR(O) hipLaunchKernelGGL(k_beam_deposit_pair<O>, dim3(bw.nwg), b256, 0, sb, f, bw, 1.0/gm.dx, 1.0/gm.dy, gm.xoff, gm.yoff)
            switch (d.order) { case 0: HPS_PAIR(0); break; case 1: HPS_PAIR(1); break; case 2: HPS_PAIR(2); break; default: HPS_PAIR(3); break; }
#undef HPS_PAIR
        }
    } else if ((e = deposit_beam_slice(islice, -1, -1, HPS_C_JZB))) return e;
    deposit_grid_current(islice, HPS_C_JZB);
    if (aux) {
        // chi is final: the multigrid's coefficient hierarchy on the auxiliary stream, joined ahead of the Sx/Sy initialisation
        HPS_HIP_CHECK(hipEventRecord(ev_aux[1], st)); HPS_HIP_CHECK(hipStreamWaitEvent(st_aux, ev_aux[1], 0));
        if ((e = mg_solve1_prepare(mg, slab, HPS_C_BX, HPS_C_SY, HPS_C_CHI, 200, st_aux))) return e;
        HPS_HIP_CHECK(hipEventRecord(ev_aux[2], st_aux));
        aux_pending = true;
    }

    // AddRhoIons + Psi, Ez, Bz solves + -grad Psi (fields/Fields.cpp:840-957)
    {   const double fa = 1.0/(gm.ep0*gm.c);
        const double fez_x = fa*0.5*(1.0/gm.dx), fez_y = fa*0.5*(1.0/gm.dy), fbz_y = gm.mu0*0.5*(1.0/gm.dy), fbz_x = -gm.mu0*0.5*(1.0/gm.dx);
        const int comps[3] = {HPS_C_PSI, HPS_C_EZ, HPS_C_BZ};
        if (fuse_sources && poisson_sources_fusable(ps) && d.field_bc == 0) {
            // the three sources (fields/Fields.cpp:887-912) are formed by the first transform pass while it loads its rows:
            // -rhomjz/ep0;  (d_x jx + d_y jy)/(ep0 c);  mu0 (d_y jx - d_x jy) -- centred differences, guard cells read as they are
            const long js = slab.jstride;
            const double* R = slab.p + (long)HPS_C_RHOMJZ*slab.nstride + g + (long)g*js;       // cell (0, 0) of the planes
            const double* X = slab.p + (long)HPS_C_JX*slab.nstride + g + (long)g*js;
            const double* Y = slab.p + (long)HPS_C_JY*slab.nstride + g + (long)g*js;
            const PoissonSrc spec[3] = {
                {1, {R, nullptr}, {nullptr, nullptr}, {-1.0/gm.ep0, 0.0}},
                {2, {X + 1, Y + js}, {X - 1, Y - js}, {fez_x, fez_y}},
                {2, {X + js, Y + 1}, {X - js, Y - 1}, {fbz_y, fbz_x}}};
            if ((e = poisson_solve_batch_src(ps, 3, spec, js, slab, comps, st))) return e;
        } else {
            hipLaunchKernelGGL(k_rhs_all, dim3(ceil_div(slab.jstride, 256), d.ny + 2*g), b256, 0, st, f, HPS_C_RHOMJZ,
                               -1 /* AddRhoIons: done by the slice's zeroing pass */, d.deposit_rho ? HPS_C_RHO : -1, HPS_C_JX, HPS_C_JY, 1.0/gm.ep0,
                               fez_x, fez_y, fbz_y, fbz_x, staging, (long)d.nx*d.ny);
            if ((e = open_boundary(3, 0x6u))) return e;
            if ((e = hps_poisson_solve_batch(ps, 3, staging, slab, comps, st))) return e;
        } }
    // m_multi_laser.AdvanceSlice (Hipace.cpp:637): a_{n+1} of this slice from chi and the neighbouring slices
    // On the laser's own stream when there is one, forked here (chi is final; nothing else of this slice needs a_{n+1}).
    // FFT solver (0.11 ms of bandwidth-bound passes at 1024^2): enqueued at once; beside the explicit deposition or beside
    // the Bx/By multigrid it gets half of its time back either way (measured: 898 -> 949 slices/s on config 5 without the
    // dopant; stream priority makes no difference).  Multigrid solver (0.7 ms, latency-bound, holds the host once per
    // V-cycle): enqueued further down, when the engine's stream has the explicit deposition and the Bx/By V-cycles
    // queued (545 -> 706 slices/s).
    const bool laser_now = c_aabs >= 0 && d.laser_solver >= 1 && d.dt != 0.0;
    const bool laser_split = laser_now && laser_stream() != st;
    if (laser_now && !laser_split) { if ((e = laser_advance_slice(*this, islice))) return e; }
    if (laser_split) { if ((e = fork_laser())) return e; }
    if (laser_split && d.laser_solver == 1) { if ((e = laser_advance_slice(*this, islice)) || (e = laser_done())) return e; }
    if (aux_pending) { HPS_HIP_CHECK(hipStreamWaitEvent(st, ev_aux[2], 0)); aux_pending = false; }
    if (pair || nbeam == 0) {
        // -grad Psi and the beam part of Sx, Sy (Hipace.cpp:659-660) in one pass (without a beam there is no deposition between them either)
        const GradPsiSxSy ga{HPS_C_PSI, HPS_C_EXMBY, HPS_C_EYPBX, 0.5*(1.0/gm.dx), 0.5*(1.0/gm.dy), HPS_C_SX, HPS_C_SY, HPS_C_JZB, HPS_C_N_JXB, HPS_C_N_JYB,
                             HPS_C_P_JXB, HPS_C_P_JYB, gm.mu0, 2.0*gm.dx, 2.0*gm.dy, 2.0*gm.dz, bb};
        // ... and, in the same launch, the coefficient hierarchy of the Bx/By multigrid solve (it needs chi only): one launch less
        // on the chain between the explicit deposition and the first smoothing pass
        bool with_hierarchy = false;
        if (fold_hierarchy && !pc && !aux) {
            if ((e = mg_solve1_prepare_with(mg, slab, HPS_C_BX, HPS_C_SY, HPS_C_CHI, 200, f, ga, st, &with_hierarchy))) return e;
        }
        if (!with_hierarchy)
            hipLaunchKernelGGL(k_gradpsi_sxsy, dim3(ceil_div(slab.jstride, 256), d.ny + 2*g), b256, 0, st, f, ga);
        mark();   // b3
    } else {
        hipLaunchKernelGGL(k_grad_psi, dim3(ceil_div(d.nx + 2*(g - 1), 256), d.ny + 2*(g - 1)), b256, 0, st, f, HPS_C_PSI,
                           HPS_C_EXMBY, HPS_C_EYPBX, 0.5*(1.0/gm.dx), 0.5*(1.0/gm.dy));
        mark();   // b3
        // beam jx, jy of the next slice; beam part of Sx, Sy (Hipace.cpp:656-660)
        if ((e = deposit_beam_slice(islice - 1, HPS_C_N_JXB, HPS_C_N_JYB, -1))) return e;
        hipLaunchKernelGGL(k_sxsy_beam, dim3(ceil_div(slab.jstride, 256), d.ny + 2*g), b256, 0, st, f, HPS_C_SX, HPS_C_SY, HPS_C_JZB, HPS_C_N_JXB, HPS_C_N_JYB,
                           HPS_C_P_JXB, HPS_C_P_JYB, gm.mu0, 2.0*gm.dx, 2.0*gm.dy, 2.0*gm.dz, bb);
    }
    mark();   // b4
    {   const int cache[4] = {HPS_C_BZ, HPS_C_EZ, HPS_C_EXMBY, HPS_C_EYPBX};
        const int depos[2] = {HPS_C_SY, HPS_C_SX};
        if ((e = species_explicit(pl, tiling, cache, depos, d.plasma_charge, d.plasma_mass, 0))) return e;
        if (ion.n > 0) { if ((e = species_explicit(ion.pl, ion.tiling, cache, depos, d.ion_charge, d.ion_mass, 1))) return e; } }

    mark();   // b5
    // Bx, By: Helmholtz multigrid from the previous slice's field (Hipace.cpp:793-933).  The plain case (one tile-sorted
    // species, nothing that reads the fields between the solve and the push) enqueues the push BEHIND the speculated
    // V-cycles, gated on the solve's own stopping rule, and only then waits for the norms: the device goes from the last
    // V-cycle straight into the push instead of idling until the host has seen the norms and launched it.
    const bool fuse = fuse_push_deposit && tiling && islice > 0 && !moving && c_aabs < 0 && ion.n == 0 && np > 0 && tiling->sorted_n == np;
    const bool gated = gate_push && tiling && !fuse && ion.n == 0 && np > 0 && tiling->sorted_n == np && !diagnostics && !d_fd && !d_insitu;
    // ... and with an ionisable species on tiles: its field bounds, its push (which takes the ADK decisions and appends the
    // electrons) and the electrons' push, all behind the speculated V-cycles; the two pushes are gated (HPS_GATED_ION_PUSH=0: off)
    const bool gated_ion = gate_push && gate_ion_push && tiling && !fuse && ion.n > 0 && ion.tiling && np > 0 && tiling->sorted_n > 0 && fold_tail
                           && !diagnostics && !d_fd && !d_insitu;
    const int comp_push[5] = {HPS_C_PSI, HPS_C_EZ, HPS_C_BX, HPS_C_BY, HPS_C_BZ};
    {
        // the plain gated push also posts the solve's norms to the host (k_advance_tiled's MgPost): no k_post_norms launch between
        // the last V-cycle and the push (HPS_POST_IN_PUSH=0: the launch of its own)
        if (gated && post_in_push) mg_defer_post(mg);
        if ((e = mg_solve1_begin(mg, slab, HPS_C_BX, HPS_C_SY, HPS_C_CHI, d.mg_tol_rel, d.mg_tol_abs, 200, st))) return e;
        if (gated_ion) {
            mark();   // b6
            mark();   // b7
            if ((e = push_with_ionization(islice, comp_push, mg_gate_after_enqueued(mg), true))) return e;
        }
        if (gated) {
            mark();   // b6
            mark();   // b7
            MgPost mp{};
            const bool posting = mg_take_deferred_post(mg, &mp);
            if ((e = advance_plasma_tiled(slab, pl, gm, comp_push, d.plasma_charge, d.plasma_mass, d.order, 0, d.n_subcycles, 0, tiling, d_nfallback, st, c_aabs, nullptr,
                                          posting ? nullptr : mg_gate_after_enqueued(mg), TailWork{}, posting ? &mp : nullptr))) return e;
        }
        if (laser_split && d.laser_solver == 2) { if ((e = laser_advance_slice(*this, islice)) || (e = laser_done())) return e; }
        pending_slice = islice; pend_fuse = fuse; pend_gated = gated; pend_gated_ion = gated_ion;
        HPS_HIP_CHECK(hipGetLastError());
        return HPS_OK;
    }
}

// DoFieldIonization + the ions' push + the electrons' push (Hipace.cpp:693-701) of a slice on tiles, enqueued without a word from
// the host in between: the ions' field bounds, the ions' push (ADK decisions on the fields it gathers, electrons appended on the
// device), the electrons' tile-sorted body with the tail's workgroups reading the live particle count.  `go`: the multigrid
// solve's gate word (both pushes do nothing unless the solve is over); first = false: the same launches again, ungated, after
// the host has added V-cycles (same ADK sequence number: the count is posted once).
int Engine::push_with_ionization (int islice, const int comp[5], const int* go, bool first)
{
    int e;
    if ((e = ion_field_bounds())) return e;
    if (first) pend_ia = ion_args(islice);
    if ((e = advance_plasma_tiled(slab, ion.pl, gm, comp, d.ion_charge, d.ion_mass, d.order, 0, d.n_subcycles, 1, ion.tiling, d_nfallback, st, c_aabs, &pend_ia, go))) return e;
    TailWork tw = fold_tail_of(pl, tiling, 256, &pend_covered);
    if (tw.nwg) tw.live_n = ion.d_cnt;
    return advance_plasma_tiled(slab, pl, gm, comp, d.plasma_charge, d.plasma_mass, d.order, 0, d.n_subcycles, 0, tiling, d_nfallback, st, c_aabs, nullptr, go, tw);
}

int Engine::solve_slice_finish (int islice)
{
    HPS_REQUIRE(pending_slice == islice, "hps_engine_solve_slice_finish: not the slice that hps_engine_solve_slice_begin started");
    pending_slice = -1;
    if (pc) return solve_slice_pc_finish(islice);
    SlabView f(slab);
    const long plane = slab.nstride;
    const dim3 b256(256);
    const dim3 gplane(ceil_div(plane, 256));
    const CellBox bb{beam_box.ilo, beam_box.ihi, beam_box.jlo, beam_box.jhi};
    const bool fuse = pend_fuse, gated = pend_gated, gated_ion = pend_gated_ion;
    const int comp_push[5] = {HPS_C_PSI, HPS_C_EZ, HPS_C_BX, HPS_C_BY, HPS_C_BZ};
    int e;
    {   int iters = 0, extra = 0;
        if ((e = mg_solve1_finish(mg, &iters, nullptr, &extra, st))) return e;
        total_vcycles += iters;
        // the speculated V-cycles were not enough (the gated push has not run): the host has added the rest, push now
        if (gated && extra) { if ((e = species_advance(pl, tiling, comp_push, d.plasma_charge, d.plasma_mass, 0, 0))) return e; }
        if (gated_ion) {
            if (extra) { if ((e = push_with_ionization(islice, comp_push, nullptr, false))) return e; }
            // the electrons the slice has released beyond the room the body's launch had for them
            if ((e = ionize_collect())) return e;
            if (pl.n > pend_covered) { if ((e = hps_advance_plasma_laser(slab, tail_of(pl, pend_covered, pl.n - pend_covered), gm, comp_push, c_aabs, d.plasma_charge, d.plasma_mass, d.order, 0, d.n_subcycles, 0, st))) return e; }
        } }

    if (!gated && !gated_ion) {
    mark();   // b6
    if (diagnostics)
        hipLaunchKernelGGL(k_checksum, dim3(64, ncomp), b256, 0, st, f, ncomp, d_checksum);
    if ((e = fill_field_diagnostic(islice))) return e;      // FillFieldDiagnostics (Hipace.cpp:691)
    if (d_insitu)                                           // Fields::InSituComputeDiags (Hipace.cpp:686)
        hipLaunchKernelGGL(k_insitu_fields, dim3(64), b256, 0, st, f, gm.c, gm.dx*gm.dy*gm.dz, d_insitu, d.nz, islice);

    mark();   // b7
    // gather + push (Hipace.cpp:699-701)
    {   const int* comp = comp_push;
        bool body_done = false; long body_covered = 0;
        if (ion.n > 0) {
            // DoFieldIonization (Hipace.cpp:693-696), then the ions' own push; the host learns how many electrons the
            // slice has released while that push runs
            if (ion.tiling) {       // decided inside the ions' LDS-tile push, on the fields it gathers anyway
                if ((e = ion_field_bounds())) return e;
                const IonArgs ia = ion_args(islice);
                if ((e = advance_plasma_tiled(slab, ion.pl, gm, comp, d.ion_charge, d.ion_mass, d.order, 0, d.n_subcycles, 1, ion.tiling, d_nfallback, st, c_aabs, &ia))) return e;
            } else {
                if ((e = ionize_slice(islice))) return e;
                if ((e = species_advance(ion.pl, ion.tiling, comp, d.ion_charge, d.ion_mass, 0, 1))) return e;
            }
            // the electrons' tile-sorted body does not depend on how many electrons the slice has released: push it while the
            // count travels to the host, then the tail with the new count
            // (the tail's workgroups in that launch read the count on the device -- the ions' push is ahead of them on the
            //  stream --, with room for 256 electrons more than the host knows of; whoever is beyond that is pushed below)
            if (tiling && tiling->sorted_n > 0 && !fuse) {
                TailWork tw = fold_tail_of(pl, tiling, 256, &body_covered);
                if (tw.nwg) tw.live_n = ion.d_cnt;
                if ((e = advance_plasma_tiled(slab, pl, gm, comp, d.plasma_charge, d.plasma_mass, d.order, 0, d.n_subcycles, 0, tiling, d_nfallback, st, c_aabs, nullptr, nullptr, tw))) return e;
                body_done = true;
            }
            if ((e = ionize_collect())) return e;
        }
        // push of this slice and deposition of the next one in one pass over the sheet (static beam, no laser, one
        // plasma species, the whole sheet inside the tile-sorted body)
        if (fuse) {
            hipLaunchKernelGGL(k_shift_zero, gplane, b256, 0, st, slab.p, slab.nstride, plane, (int)slab.jstride, bb, d.deposit_rho ? (int)HPS_C_RHO : -1, (int)HPS_C_ION_RHOMJZ);
            const int dep[6] = {HPS_C_JX, HPS_C_JY, -1, d.deposit_rho ? HPS_C_RHO : -1, HPS_C_CHI, HPS_C_RHOMJZ};
            if ((e = advance_deposit_tiled(slab, pl, gm, comp, dep, d.plasma_charge, d.plasma_mass, d.order, d.n_subcycles, d.max_qsa, d_nqsa, tiling, d_nfallback, st))) return e;
            ahead_for = islice - 1;
        } else if (body_done) {
            if (pl.n > body_covered) { if ((e = hps_advance_plasma_laser(slab, tail_of(pl, body_covered, pl.n - body_covered), gm, comp, c_aabs, d.plasma_charge, d.plasma_mass, d.order, 0, d.n_subcycles, 0, st))) return e; }
        } else {
            if ((e = species_advance(pl, tiling, comp, d.plasma_charge, d.plasma_mass, 0, 0))) return e;
        } }
    }

    // beam push and hand-off of the slipped particles (Hipace.cpp:704-706)
    insitu_beam(islice);
    if (moving && nbeam > 0) { if ((e = beam_push_moving(*this, islice))) return e; }
    mark();   // b8
    // ShiftSlices (fields/Fields.cpp:588-604)
    // (lazy_shift: left to the start of the next slice, where it shares a pass with InitializeSlices; anything that looks at
    //  the slab in between -- hps_engine_slab, _sync, the diagnostics' accessors -- runs it first: flush_shift)
    if (ahead_for != islice - 1) {
        if (lazy_shift && islice > 0) shift_pending = true;
        else hipLaunchKernelGGL(k_shift_slices, gplane, b256, 0, st, slab.p, slab.nstride, plane, (int)slab.jstride, bb);
    }
    mark();   // b9
    HPS_HIP_CHECK(hipGetLastError());
    ++slices_done;
    return HPS_OK;
}

int Engine::run_step ()
{
    if (int e = begin_step()) return e;
    for (int isl = d.nz - 1; isl >= 0; --isl)
        if (int e = solve_slice(isl)) return e;
    return HPS_OK;
}

} // namespace hps

using namespace hps;

extern "C" const char* hps_last_error (void) { return g_err.c_str(); }
extern "C" const char* hps_version (void) { return "hpslice 0.1 (gfx950)"; }

extern "C" int hps_engine_create (const hps_deck* deck, int device, void** handle)
{
    HPS_REQUIRE(deck && handle, "hps_engine_create: null argument");
    Engine* E = new Engine;
    if (int e = E->create(*deck, device)) { delete E; return e; }
    *handle = E;
    return HPS_OK;
}
extern "C" int hps_engine_destroy (void* h) { delete static_cast<Engine*>(h); return HPS_OK; }
extern "C" int hps_engine_begin_step (void* h) { return static_cast<Engine*>(h)->begin_step(); }
extern "C" int hps_engine_solve_slice (void* h, int islice) { return static_cast<Engine*>(h)->solve_slice(islice); }
extern "C" int hps_engine_solve_slice_begin (void* h, int islice) { return static_cast<Engine*>(h)->solve_slice_begin(islice); }
extern "C" int hps_engine_slice_ready (void* h)
{
    Engine* E = static_cast<Engine*>(h);
    if (E->pc) {
        // device-controlled loop: ready when the host's walk over the posted errors (solve_slice_pc_finish) would not wait --
        // an iteration that met the tolerance has been posted, or every iteration enqueued so far has
        if (!(E->pc_speculate && E->tiling && !E->moving && E->ion.n == 0) || E->pc_islice < 0 || E->pc_enqueued <= 0) return 1;
        for (int it = 1; it <= E->pc_enqueued; ++it) {
            const volatile double* hp = E->h_pc + 8*it;
            const double seq = E->pc_base_seq + it;
            if (!(hp[1] == seq && hp[3] == seq && hp[5] == seq)) return 0;
            const double err = hp[0] > 0.0 ? hp[2]/hp[0] : 0.0;
            if (!(err > E->pc_tol) || it >= E->pc_max_iter) return 1;
        }
        return 1;
    }
    if (E->pending_slice < 0) return 1;
    return mg_solve1_ready(E->mg) ? 1 : 0;
}
extern "C" int hps_engine_solve_slice_finish (void* h, int islice) { return static_cast<Engine*>(h)->solve_slice_finish(islice); }
extern "C" int hps_engine_run_step (void* h) { return static_cast<Engine*>(h)->run_step(); }
extern "C" int hps_engine_sync (void* h)
{
    Engine* E = static_cast<Engine*>(h);
    E->flush_shift();
    if (int e = E->join_laser()) return e;
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    if (E->moving && E->d_beam_overflow) {
        // a slice that outgrew the hand-off capacity lost particles: say so at the end of the step that did it (also the last
        // step of a run, and a stage that runs one step only), not at the next begin_step of this engine
        int ov = 0;
        HPS_HIP_CHECK(hipMemcpy(&ov, E->d_beam_overflow, sizeof(int), hipMemcpyDeviceToHost));
        HPS_REQUIRE(ov == 0, "moving beam: a slice outgrew the hand-off capacity (twice the fullest injected slice): particles were lost");
    }
    return HPS_OK;
}
extern "C" int hps_engine_info (void* h, int* ncomp, int* ng, long* np)
{
    Engine* E = static_cast<Engine*>(h);
    if (ncomp) *ncomp = E->ncomp;
    if (ng) *ng = E->g;
    if (np) *np = E->np;
    return HPS_OK;
}
extern "C" hps_slab hps_engine_slab (void* h) { Engine* E = static_cast<Engine*>(h); E->flush_shift(); return E->slab; }
extern "C" hps_plasma hps_engine_plasma (void* h) { return static_cast<Engine*>(h)->pl; }
extern "C" int hps_engine_tiling (void* h, void** tiling)
{
    HPS_REQUIRE(h && tiling, "hps_engine_tiling: null argument");
    *tiling = static_cast<Engine*>(h)->tiling;
    return HPS_OK;
}
extern "C" hps_plasma hps_engine_ions (void* h)
{
    Engine* E = static_cast<Engine*>(h);
    (void)hipStreamSynchronize(E->st);
    hps_plasma p = E->ion.pl;
    p.n = E->ion.n;
    return p;
}
extern "C" int hps_engine_ion_stats (void* h, long* n_ionized, long* n_product)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    if (int e = E->ionize_collect()) return e;
    if (n_ionized) *n_ionized = E->ion.n_ionized;
    if (n_product) *n_product = E->np;
    return HPS_OK;
}
extern "C" hps_stream hps_engine_stream (void* h) { return static_cast<Engine*>(h)->st; }
extern "C" int hps_engine_checksums (void* h, double* out)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    HPS_HIP_CHECK(hipMemcpy(out, E->d_checksum, E->ncomp*sizeof(double), hipMemcpyDeviceToHost));
    return HPS_OK;
}
extern "C" int hps_engine_stats (void* h, long* vc, long* sl)
{
    Engine* E = static_cast<Engine*>(h);
    if (vc) *vc = E->total_vcycles;
    if (sl) *sl = E->slices_done;
    return HPS_OK;
}
extern "C" int hps_engine_set_insitu_plasma (void* h, double radius)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    (void)hipFree(E->d_insitu_pl); E->d_insitu_pl = nullptr; E->insitu_pl_radius = radius;
    if (!(radius > 0.0)) return HPS_OK;
    HPS_HIP_CHECK(hipMalloc(&E->d_insitu_pl, (size_t)15*E->d.nz*sizeof(double)));
    HPS_HIP_CHECK(hipMemset(E->d_insitu_pl, 0, (size_t)15*E->d.nz*sizeof(double)));
    return HPS_OK;
}
extern "C" int hps_engine_set_insitu_beam (void* h, double radius)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    (void)hipFree(E->d_insitu_bm); E->d_insitu_bm = nullptr; E->insitu_bm_radius = radius;
    if (!(radius > 0.0)) return HPS_OK;
    HPS_HIP_CHECK(hipMalloc(&E->d_insitu_bm, (size_t)23*E->d.nz*sizeof(double)));
    HPS_HIP_CHECK(hipMemset(E->d_insitu_bm, 0, (size_t)23*E->d.nz*sizeof(double)));
    return HPS_OK;
}
extern "C" int hps_engine_insitu_beam (void* h, double* out)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->d_insitu_bm && out, "hps_engine_insitu_beam: not switched on");
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    const int nz = E->d.nz;
    HPS_HIP_CHECK(hipMemcpy(out, E->d_insitu_bm, (size_t)23*nz*sizeof(double), hipMemcpyDeviceToHost));
    // averages: everything but sum(w) and the count is divided by sum(w) (BeamParticleContainer.cpp:542-548)
    for (int k = 0; k < nz; ++k) {
        const double sw = out[k], inv = sw <= 0.0 ? 0.0 : 1.0/sw;
        for (int q = 1; q <= 21; ++q) out[(size_t)q*nz + k] *= inv;
    }
    return HPS_OK;
}
extern "C" int hps_engine_insitu_plasma (void* h, double* out)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->d_insitu_pl && out, "hps_engine_insitu_plasma: not switched on");
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    const int nz = E->d.nz;
    HPS_HIP_CHECK(hipMemcpy(out, E->d_insitu_pl, (size_t)15*nz*sizeof(double), hipMemcpyDeviceToHost));
    // averages: everything but sum(w), the energy and the count is divided by sum(w) (:511-516)
    for (int k = 0; k < nz; ++k) {
        const double sw = out[k], inv = sw <= 0.0 ? 0.0 : 1.0/sw;
        for (int q = 1; q <= 12; ++q) out[(size_t)q*nz + k] *= inv;
    }
    return HPS_OK;
}
extern "C" int hps_engine_set_insitu_fields (void* h, int on)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    (void)hipFree(E->d_insitu); E->d_insitu = nullptr;
    if (!on) return HPS_OK;
    if (E->pc) { set_error("hps_engine_set_insitu_fields: explicit solver only (needs jz_beam), as the reference"); return HPS_ERR_UNSUPPORTED; }
    HPS_HIP_CHECK(hipMalloc(&E->d_insitu, (size_t)10*E->d.nz*sizeof(double)));
    HPS_HIP_CHECK(hipMemset(E->d_insitu, 0, (size_t)10*E->d.nz*sizeof(double)));
    return HPS_OK;
}
extern "C" int hps_engine_insitu_fields (void* h, double* out)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->d_insitu && out, "hps_engine_insitu_fields: not switched on");
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    HPS_HIP_CHECK(hipMemcpy(out, E->d_insitu, (size_t)10*E->d.nz*sizeof(double), hipMemcpyDeviceToHost));
    return HPS_OK;
}
// diagnostic.diag_type (xyz: slice_dir -1, yz: 0, xz: 1, xy: 2), diagnostic.coarsening, diagnostic.patch_lo / patch_hi:
// the box of Diagnostic::ResizeFDiagFAB (diagnostics/Diagnostic.cpp:300-390) and TrimIOBox (:393-410)
extern "C" int hps_engine_set_field_diagnostic_box (void* h, int ncomps, const int* comps, const int coarsening[3], int slice_dir,
                                                    const double* patch_lo, const double* patch_hi)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    (void)hipFree(E->d_fd); (void)hipFree(E->d_fd_comps); E->d_fd = nullptr; E->d_fd_comps = nullptr; E->fd_comps.clear();
    if (ncomps <= 0) return HPS_OK;
    HPS_REQUIRE(comps && coarsening, "hps_engine_set_field_diagnostic: null argument");
    HPS_REQUIRE(slice_dir >= -1 && slice_dir <= 2, "hps_engine_set_field_diagnostic: slice_dir must be -1 (xyz), 0 (yz), 1 (xz) or 2 (xy)");
    for (int k = 0; k < ncomps; ++k) HPS_REQUIRE(comps[k] >= 0 && comps[k] < E->ncomp, "hps_engine_set_field_diagnostic: bad component");
    const int n3[3] = {E->d.nx, E->d.ny, E->d.nz};
    const double h3[3] = {E->gm.dx, E->gm.dy, E->gm.dz};
    auto floor_div = [] (int a, int b) { return a >= 0 ? a/b : -((-a + b - 1)/b); };
    for (int q = 0; q < 3; ++q) {
        const int c = (q == slice_dir) ? 1 : coarsening[q];               // (the slice direction is never coarsened, Diagnostic.cpp:71-73)
        HPS_REQUIRE(c >= 1, "hps_engine_set_field_diagnostic: coarsening must be >= 1");
        if (!patch_lo && !patch_hi && q != slice_dir)
            HPS_REQUIRE(n3[q] % c == 0, "hps_engine_set_field_diagnostic: sizes must be divisible by the coarsening");
        E->fd_c[q] = c;
        const double plo = E->d.lo[q], phi = E->d.hi[q], hh = h3[q];
        const double poff_sim = 0.5*(plo + phi - hh*(n3[q] - 1));
        int small = 0, big = n3[q] - 1;
        if (patch_lo) small = std::max(small, (int)std::round((patch_lo[q] - poff_sim)/hh));
        if (patch_hi) big = std::min(big, (int)std::round((patch_hi[q] - poff_sim)/hh));
        HPS_REQUIRE(big >= small, "hps_engine_set_field_diagnostic: the patch does not meet the box");
        double lo_d = plo + small*hh, hi_d = phi + (big - (n3[q] - 1))*hh;
        if (q == slice_dir) {
            const double half = (hi_d - lo_d)/(2.0*(big - small + 1)), mid = 0.5*(lo_d + hi_d);
            small = big = 0;
            if (q < 2) { lo_d = mid - half; hi_d = mid + half; }
        }
        const int sc = floor_div(small, c), bc = floor_div(big, c);
        E->fd_n[q] = bc - sc + 1;
        E->fd_h[q] = (hi_d - lo_d)/E->fd_n[q];
        E->fd_pos0[q] = 0.5*(lo_d + hi_d - E->fd_h[q]*(sc + bc)) + sc*E->fd_h[q];
        E->fd_lo[q] = lo_d; E->fd_hi[q] = hi_d;
    }
    E->fd_slice_dir = slice_dir;
    E->fd_comps.assign(comps, comps + ncomps);
    const size_t cells = E->fd_cells();
    HPS_HIP_CHECK(hipMalloc(&E->d_fd, ncomps*cells*sizeof(double)));
    HPS_HIP_CHECK(hipMemset(E->d_fd, 0, ncomps*cells*sizeof(double)));
    HPS_HIP_CHECK(hipMalloc(&E->d_fd_comps, ncomps*sizeof(int)));
    HPS_HIP_CHECK(hipMemcpy(E->d_fd_comps, comps, ncomps*sizeof(int), hipMemcpyHostToDevice));
    return HPS_OK;
}
extern "C" int hps_engine_set_field_diagnostic (void* h, int ncomps, const int* comps, const int coarsening[3])
{
    return hps_engine_set_field_diagnostic_box(h, ncomps, comps, coarsening, -1, nullptr, nullptr);
}
extern "C" int hps_engine_field_diagnostic_geometry (void* h, int* n3, double* lo3, double* hi3)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->d_fd, "hps_engine_field_diagnostic_geometry: no diagnostic set");
    for (int q = 0; q < 3; ++q) { if (n3) n3[q] = E->fd_n[q]; if (lo3) lo3[q] = E->fd_lo[q]; if (hi3) hi3[q] = E->fd_hi[q]; }
    return HPS_OK;
}
extern "C" int hps_engine_field_diagnostic (void* h, double* out)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->d_fd && out, "hps_engine_field_diagnostic: no diagnostic set");
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    HPS_HIP_CHECK(hipMemcpy(out, E->d_fd, E->fd_comps.size()*E->fd_cells()*sizeof(double), hipMemcpyDeviceToHost));
    return HPS_OK;
}
extern "C" int hps_engine_record_event (void* h, int slot, void** out)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(slot >= 0 && slot < (1 << 20) && out, "hps_engine_record_event: bad slot");
    if ((size_t)slot >= E->hand_ev.size()) E->hand_ev.resize((size_t)slot + 1, nullptr);
    if (!E->hand_ev[slot]) HPS_HIP_CHECK(hipEventCreateWithFlags(&E->hand_ev[slot], event_flags(false)));
    if (int e = E->join_laser()) return e;     // "everything the engine has been given so far" includes the laser stream's slice
    HPS_HIP_CHECK(hipEventRecord(E->hand_ev[slot], E->st));
    *out = E->hand_ev[slot];
    return HPS_OK;
}
extern "C" int hps_engine_wait_event (void* h, void* event)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(event, "hps_engine_wait_event: null event");
    HPS_REQUIRE(!ring_is_ticket(event), "hps_engine_wait_event: this is the receive handle of an ipc ring edge, not a hipEvent_t -- order the engine behind "
                                        "it with hps_ring_engine_wait(ring, engine, handle), which works for both kinds of edge");
    HPS_HIP_CHECK(hipStreamWaitEvent(E->st, static_cast<hipEvent_t>(event), 0));
    return HPS_OK;
}
extern "C" int hps_engine_copy_async (void* h, void* dst, const void* src, long bytes)
{
    Engine* E = static_cast<Engine*>(h);
    if (bytes <= 0) return HPS_OK;
    HPS_REQUIRE(dst && src, "hps_engine_copy_async: null pointer");
    HPS_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, E->st));
    return HPS_OK;
}
extern "C" int hps_engine_set_step (void* h, int step)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E && step >= 0, "hps_engine_set_step: bad argument");
    E->next_step = step;
    return HPS_OK;
}
extern "C" int hps_engine_set_laser_import (void* h, int on, int step)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->laser, "hps_engine_set_laser_import: no laser");
    return laser_set_import(*E, on, step);
}
extern "C" int hps_engine_export_laser_slice (void* h, int islice, double* msg_dev)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->laser && msg_dev && islice >= 0 && islice < E->d.nz, "hps_engine_export_laser_slice: bad argument");
    return laser_export_slice(*E, islice, msg_dev);
}
extern "C" int hps_engine_import_laser_slice (void* h, int islice, const double* msg_dev)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->laser && msg_dev && islice >= 0 && islice < E->d.nz, "hps_engine_import_laser_slice: bad argument");
    return laser_import_slice(*E, islice, msg_dev);
}
extern "C" int hps_engine_import_laser_from (void* h, int islice, void* src)
{
    Engine* E = static_cast<Engine*>(h); Engine* S = static_cast<Engine*>(src);
    HPS_REQUIRE(E->laser && S && S->laser && islice >= 0 && islice < E->d.nz && S->d.nx == E->d.nx && S->d.ny == E->d.ny && S->d.nz == E->d.nz,
                "hps_engine_import_laser_from: bad argument");
    return laser_import_from(*E, islice, *S);
}
extern "C" int hps_engine_laser_envelope (void* h, double* out_host)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->laser && out_host, "hps_engine_laser_envelope: no laser");
    return laser_copy_envelope(*E, out_host);
}
extern "C" int hps_engine_laser_info (void* h, int* aabs_comp, double* sum_host)
{
    Engine* E = static_cast<Engine*>(h);
    if (aabs_comp) *aabs_comp = E->c_aabs;
    if (sum_host) {
        HPS_HIP_CHECK(hipStreamSynchronize(E->st));
        HPS_HIP_CHECK(hipMemcpy(sum_host, E->d_laser_sum, sizeof(double), hipMemcpyDeviceToHost));
    }
    return HPS_OK;
}
extern "C" int hps_engine_laser_vcycles (void* h, long* vcycles)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(vcycles, "hps_engine_laser_vcycles: null argument");
    *vcycles = E->laser ? laser_mg_vcycles(*E) : 0;
    return HPS_OK;
}
extern "C" int hps_engine_pc_stats (void* h, long* its, double* err_sum)
{
    Engine* E = static_cast<Engine*>(h);
    if (its) *its = E->pc_iterations;
    if (err_sum) *err_sum = E->pc_err_sum;
    return HPS_OK;
}
// slices on which the predictor-corrector loop saw sum |B| = 0 in its first pass -- exactly, or below the engine's rounding
// floor (Engine::pc_floor, HPS_PC_NOISE_FLOOR) -- and left after it (fields/Fields.cpp:1283)
extern "C" int hps_engine_pc_zero_b_slices (void* h, long* n)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E && n, "hps_engine_pc_zero_b_slices: null argument");
    *n = E->pc_zero_b_slices;
    return HPS_OK;
}
extern "C" int hps_engine_set_profiling (void* h, int on)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    E->profiling = (on != 0); E->prof_light = (on == 2); E->ev_used = 0; E->prof_now = false;
    return HPS_OK;
}
extern "C" int hps_engine_set_profiling_stride (void* h, int stride)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(stride >= 1, "hps_engine_set_profiling_stride: stride >= 1");
    E->prof_stride = stride;
    return HPS_OK;
}
extern "C" int hps_engine_phase_times (void* h, double* ms, long* nsl)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    // interval -> phase: 0 deposit, 1 poisson, 2 explicit, 3 mg, 4 push, 5 other
    static const int phase_of[10] = {5, 6, 0, 1, 5, 2, 3, 5, 4, 5};
    for (int k = 0; k < 8; ++k) ms[k] = 0.0;
    const size_t ns = E->ev_used/11;
    // interval 4 (b3 -> b4) holds no kernel when the two beam deposits share a launch (static beam, explicit solver): two
    // event records back to back, i.e. what an interval costs by itself
    const bool empty4 = !E->pc && !E->moving && E->nbeam > 0;
    for (size_t s = 0; s < ns; ++s)
        for (int k = 0; k < 10; ++k) {
            if (E->prof_light && k != 2 && k != 4) continue;     // light mode recorded marks 2..5 only
            float t = 0.f;
            HPS_HIP_CHECK(hipEventElapsedTime(&t, E->ev[s*11 + k], E->ev[s*11 + k + 1]));
            ms[phase_of[k]] += t;
            if (k == 4 && empty4) ms[7] += t;
        }
    if (nsl) *nsl = (long)ns;
    E->ev_used = 0;
    return HPS_OK;
}
extern "C" int hps_engine_beam_info (void* h, long* nbeam, long* offsets_host)
{
    Engine* E = static_cast<Engine*>(h);
    if (nbeam) *nbeam = E->nbeam;
    if (offsets_host) for (int p = 0; p <= E->d.nz; ++p) offsets_host[p] = E->beam_off[p];
    return HPS_OK;
}
extern "C" int hps_engine_set_beam_storage (void* h, double* storage_dev)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(!(E->moving && storage_dev), "hps_engine_set_beam_storage: caller-owned static beam blocks need hipace.dt = 0; a moving beam is handed on with hps_engine_set_beam_import / export_beam_slice / import_beam_slice");
    E->beam_cur = storage_dev ? storage_dev : E->beam_data;
    // caller-owned particles may sit anywhere: treat the whole plane as beam support until told otherwise
    E->beam_box = storage_dev ? E->full_box : E->beam_box_init;
    return HPS_OK;
}
extern "C" int hps_engine_beam_capacity (void* h, long* cap)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->moving, "hps_engine_beam_capacity: the engine's beam is static (hipace.dt = 0)");
    *cap = E->beam_cap;
    return HPS_OK;
}
extern "C" int hps_engine_set_beam_capacity (void* h, long cap)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->moving, "hps_engine_set_beam_capacity: the engine's beam is static (hipace.dt = 0)");
    HPS_REQUIRE(E->steps_begun == 0, "hps_engine_set_beam_capacity: call before the first hps_engine_begin_step");
    HPS_REQUIRE(cap >= 1 && cap < (1L << 31), "hps_engine_set_beam_capacity: bad capacity");
    E->beam_cap = cap;
    return HPS_OK;
}
extern "C" int hps_engine_beam_message_rows (void* h, int* rows)
{
    *rows = static_cast<Engine*>(h)->beam_rows;
    return HPS_OK;
}
extern "C" int hps_engine_beam_spin (void* h, double* soa_host)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->moving && E->bm.sx && soa_host, "hps_engine_beam_spin: the beam has no spin (do_spin_tracking with hipace.dt != 0)");
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    const double* a[3] = {E->bm.sx, E->bm.sy, E->bm.sz};
    for (int q = 0; q < 3; ++q)
        if (E->nbeam > 0) HPS_HIP_CHECK(hipMemcpy(soa_host + (size_t)q*E->nbeam, a[q], E->nbeam*sizeof(double), hipMemcpyDeviceToHost));
    return HPS_OK;
}
extern "C" int hps_engine_set_beam_import (void* h, int on)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->moving, "hps_engine_set_beam_import: the engine's beam is static (hipace.dt = 0)");
    E->beam_import = (on != 0);
    return HPS_OK;
}
extern "C" int hps_engine_export_beam_slice (void* h, int islice, double* msg_dev)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->moving && msg_dev && islice >= 0 && islice < E->d.nz, "hps_engine_export_beam_slice: bad argument");
    return beam_export_slice(*E, islice, msg_dev, E->beam_cap);
}
extern "C" int hps_engine_import_beam_slice (void* h, int islice, const double* msg_dev)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->moving && E->beam_import && msg_dev && islice >= 0 && islice < E->d.nz, "hps_engine_import_beam_slice: bad argument (import mode off?)");
    return beam_import_slice(*E, islice, msg_dev, E->beam_cap);
}
extern "C" int hps_engine_beam_state (void* h, long* boundaries_host, double* soa_host)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(E->moving, "hps_engine_beam_state: the engine's beam is static (hipace.dt = 0)");
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    {   int ov = 0;
        HPS_HIP_CHECK(hipMemcpy(&ov, E->d_beam_overflow, sizeof(int), hipMemcpyDeviceToHost));
        HPS_REQUIRE(ov == 0, "moving beam: a slice outgrew the hand-off capacity (twice the fullest injected slice)"); }
    if (boundaries_host) HPS_HIP_CHECK(hipMemcpy(boundaries_host, E->d_B, (size_t)(E->d.nz + 1)*sizeof(long), hipMemcpyDeviceToHost));
    if (soa_host && E->nbeam > 0)
        for (int k = 0; k < 7; ++k)
            HPS_HIP_CHECK(hipMemcpy(soa_host + (size_t)k*E->nbeam, E->bm_store + (size_t)k*std::max(E->nbeam, 1L), E->nbeam*sizeof(double), hipMemcpyDeviceToHost));
    return HPS_OK;
}
extern "C" int hps_engine_assume_initial_beam_support (void* h)
{
    Engine* E = static_cast<Engine*>(h);
    E->beam_box = E->beam_box_init;
    return HPS_OK;
}
extern "C" int hps_engine_initial_beam (void* h, double* dst_dev)
{
    Engine* E = static_cast<Engine*>(h);
    if (E->nbeam > 0) HPS_HIP_CHECK(hipMemcpy(dst_dev, E->beam_init, 7*E->nbeam*sizeof(double), hipMemcpyDeviceToDevice));
    return HPS_OK;
}
extern "C" int hps_engine_set_beam_particles (void* h, long n, const double* soa_host, long* n_outside)
{
    HPS_REQUIRE(h, "hps_engine_set_beam_particles: null engine");
    return static_cast<Engine*>(h)->set_beam_particles(n, soa_host, n_outside);
}
extern "C" int hps_engine_set_tiling (void* h, int tile_size, int sort_period)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(tile_size == 0 || tile_size == 16 || tile_size == 32, "hps_engine_set_tiling: tile_size must be 0, 16 or 32");
    HPS_REQUIRE(sort_period >= 1, "hps_engine_set_tiling: sort_period must be >= 1");
    HPS_REQUIRE(E->tiling == nullptr, "hps_engine_set_tiling: call before the first hps_engine_begin_step");
    E->tile_size = tile_size; E->sort_period = sort_period;
    return HPS_OK;
}
extern "C" int hps_engine_set_density_profile (void* h, int nr, const double* r_host, const double* fr_host, int nt, const double* ct_host,
                                              const double* ft_host)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_REQUIRE(nr >= 0 && nt >= 0 && (nr == 0 || (r_host && fr_host)) && (nt == 0 || (ct_host && ft_host)), "hps_engine_set_density_profile: bad argument");
    for (int k = 1; k < nr; ++k) HPS_REQUIRE(r_host[k] > r_host[k - 1], "hps_engine_set_density_profile: r must increase");
    for (int k = 1; k < nt; ++k) HPS_REQUIRE(ct_host[k] > ct_host[k - 1], "hps_engine_set_density_profile: ct must increase");
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    (void)hipFree(E->d_prof_r); E->d_prof_r = nullptr;
    E->prof_r.assign(r_host, r_host + nr);
    E->prof_t.assign(ct_host, ct_host + nt); E->prof_f_t.assign(ft_host, ft_host + nt);
    if (nr > 0) {
        std::vector<double> both(r_host, r_host + nr);
        both.insert(both.end(), fr_host, fr_host + nr);
        HPS_HIP_CHECK(hipMalloc(&E->d_prof_r, both.size()*sizeof(double)));
        HPS_HIP_CHECK(hipMemcpy(E->d_prof_r, both.data(), both.size()*sizeof(double), hipMemcpyHostToDevice));
    }
    return HPS_OK;
}
extern "C" int hps_engine_set_fusion (void* h, int on)
{
    static_cast<Engine*>(h)->fuse_push_deposit = (on != 0);
    return HPS_OK;
}
extern "C" int hps_engine_fallbacks (void* h, long* n)
{
    Engine* E = static_cast<Engine*>(h);
    HPS_HIP_CHECK(hipStreamSynchronize(E->st));
    int v = 0;
    HPS_HIP_CHECK(hipMemcpy(&v, E->d_nfallback, sizeof(int), hipMemcpyDeviceToHost));
    *n = v;
    return HPS_OK;
}
extern "C" int hps_engine_sorts (void* h, long* n) { *n = static_cast<Engine*>(h)->n_sorts; return HPS_OK; }
extern "C" int hps_engine_set_diagnostics (void* h, int on) { static_cast<Engine*>(h)->diagnostics = (on != 0); return HPS_OK; }

extern "C" int hps_memcpy_d2h (void* dst, const void* src, long bytes) { HPS_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return HPS_OK; }
extern "C" int hps_memcpy_h2d (void* dst, const void* src, long bytes) { HPS_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return HPS_OK; }
extern "C" int hps_device_count (int* n)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    *n = c;
    return HPS_OK;
}
