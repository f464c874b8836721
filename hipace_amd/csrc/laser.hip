// laser.hip -- laser envelope on the device: the Gaussian initial envelope, |a|^2 for the plasma operators and the
// FFT envelope solver that advances the pulse by one time step, slice by slice.
//
// Reference: laser/MultiLaser.cpp -- InitLaserSlice :803-921 (Gaussian branch :881-919, Laser.H defaults: no
// carrier-envelope phase, no propagation angle, no pulse-front tilt), UpdateLaserAabs :214-291, AdvanceSliceFFT
// :609-801 (Benedetti et al. 2017 scheme with the on-axis phase terms of lasers.use_phase), and the hand-over of the
// time levels between two steps, utils/MultiBuffer.cpp:840-852, 913-925.
//
// Layout: three complex arrays [nz][ny][nx] (a at time steps n-1, n, n+1), 16 B per cell, resident in HBM for the whole
// run -- 1024^2 x 2048 slices are 3 x 34 GB, which is what 288 GB per GPU are for; the reference streams them through
// host buffers.  The laser grid is the field grid (lasers.n_cell / patch_* not given), lasers.interp_order = 1, so chi
// and |a|^2 pass between the two grids cell by cell.
#include "common.h"
#include "engine.h"

#include <rocfft/rocfft.h>

namespace hps {

struct LaserPhase { double2 exp1, exp2; double djn, pad; };

int mg2_create_internal (int nx, int ny, double dx, double dy, void** h);       // multigrid2.hip
int mg2_solve_internal (void* h, double* sol, const double* rhs, const double* ar, const double* ai, double tol_rel, double tol_abs,
                        int maxiter, int* iters, hipStream_t st);
void mg2_destroy_internal (void* h);

struct LaserState {
    int nx = 0, ny = 0, nz = 0;
    double2 *nm1 = nullptr, *n00 = nullptr, *np1 = nullptr;      // [nz][ny][nx]
    double2* work = nullptr;                                       // [ny][nx] right-hand side / solution
    LaserPhase* phase = nullptr;
    rocfft_plan fwd = nullptr, bwd = nullptr; rocfft_execution_info info = nullptr; void* fft_work = nullptr;
    // lasers.solver_type = multigrid (multigrid2.hip): planar [2][ny][nx] solution (kept from slice to slice: the next
    // solve's initial guess, as np1j00 in the reference) and right-hand side, Re of the coefficient, its Im (one double)
    void* mg = nullptr; double *mg_sol = nullptr, *mg_rhs = nullptr, *mg_acf_real = nullptr, *mg_acf_imag = nullptr; long mg_vcycles = 0;
    int steps = 0; bool initialised = false;
    bool import_mode = false;        // ring pipeline: a_n, a_{n-1} of the coming step arrive through laser_import_slice
    ~LaserState () {
        if (fwd) rocfft_plan_destroy(fwd);
        if (bwd) rocfft_plan_destroy(bwd);
        if (info) rocfft_execution_info_destroy(info);
        (void)hipFree(nm1); (void)hipFree(n00); (void)hipFree(np1); (void)hipFree(work); (void)hipFree(phase); (void)hipFree(fft_work);
        if (mg) mg2_destroy_internal(mg);
        (void)hipFree(mg_sol); (void)hipFree(mg_rhs); (void)hipFree(mg_acf_real); (void)hipFree(mg_acf_imag);
    }
};

__device__ __forceinline__ double2 cmul (double2 a, double2 b) { return make_double2(a.x*b.x - a.y*b.y, a.x*b.y + a.y*b.x); }
__device__ __forceinline__ double2 cadd (double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub (double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cscale (double s, double2 a) { return make_double2(s*a.x, s*a.y); }

struct LaserPars { double a0, w0, L0, k0, x0, y0, z0, zfoc; };

// Gaussian envelope of every slice (InitLaserSlice :881-919)
__global__ __launch_bounds__(256)
void k_laser_init (double2* a, int nx, int ny, int nz, LaserPars L, double dx, double dy, double dz, double xoff, double yoff, double zoff)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    const int j = blockIdx.y, k = blockIdx.z;
    if (i >= nx) return;
    const double x = i*dx + xoff - L.x0, y = j*dy + yoff - L.y0, zp = k*dz + zoff - L.z0;
    // diffract_factor D = 1 + i q, q = (zp - zfoc + z0) 2/(k0 w0^2)
    const double q = (zp - L.zfoc + L.z0)*2.0/(L.k0*L.w0*L.w0);
    const double den = 1.0 + q*q;
    const double dr = 1.0/den, di = -q/den;                     // 1/D
    const double wr = dr/(L.w0*L.w0), wi = di/(L.w0*L.w0);       // 1/(w0^2 D)
    const double r2 = x*x + y*y;
    const double er = -r2*wr - zp*zp/(L.L0*L.L0), ei = -r2*wi;   // exponent
    const double m = exp(er);
    double sn, cs; sincos(ei, &sn, &cs);
    const double ar = L.a0*dr, ai = L.a0*di;                     // prefactor a0/D
    a[((long)k*ny + j)*nx + i] = make_double2(m*(ar*cs - ai*sn), m*(ar*sn + ai*cs));
}

// UpdateLaserAabs (:214-291): aabs = |a_n|^2 on the valid cells, 0 in the guard cells; optionally sum |a_n|
__global__ __launch_bounds__(256)
void k_laser_aabs (SlabView f, int c_aabs, const double2* __restrict__ a, double* sum_abs)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x - f.ng;
    const int j = blockIdx.y - f.ng;
    double mag = 0.0;
    if (i < f.nx + f.ng) {
        double v = 0.0;
        if (i >= 0 && i < f.nx && j >= 0 && j < f.ny) {
            const double2 e = a[(long)j*f.nx + i];
            v = e.x*e.x + e.y*e.y;
            mag = sqrt(v);
        }
        f.p[c_aabs*f.ns + f.off(i, j)] = v;
    }
    if (sum_abs) {
        for (int o = 32; o > 0; o >>= 1) mag += __shfl_xor(mag, o);
        __shared__ double part[4];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mag;
        __syncthreads();
        if (threadIdx.x == 0) atomic_add_f64(sum_abs, part[0] + part[1] + part[2] + part[3]);
    }
}

// on-axis phases of a_n on slices j, j+1, j+2 (the sum of the 1, 2 or 4 cells nearest the axis, as Wake-T; :651-697)
// -> exp(i(t_j - t_j+1)), exp(i(t_j - t_j+2)) and D_j^n.  One lane.
__global__ void k_laser_phase (const double2* j00, const double2* jp1, const double2* jp2, int nx, int ny, int use_phase, double dz,
                               LaserPhase* out)
{
    const int imid = (nx + 1)/2, jmid = (ny + 1)/2;
    double t[3] = {0.0, 0.0, 0.0};
    if (use_phase) {
        const double2* src[3] = {j00, jp1, jp2};
        for (int q = 0; q < 3; ++q) {
            double sr = 0.0, si = 0.0;
            if (src[q]) {
                for (int j = (ny % 2 == 0 ? jmid - 1 : jmid); j <= jmid; ++j)
                    for (int i = (nx % 2 == 0 ? imid - 1 : imid); i <= imid; ++i) { const double2 v = src[q][(long)j*nx + i]; sr += v.x; si += v.y; }
            }
            t[q] = atan2(si, sr);
        }
    }
    const double pi = 3.14159265358979323846;
    double dt1 = t[0] - t[1], dt2 = t[1] - t[2];
    if (dt1 < -1.5*pi) dt1 += 2.0*pi;
    if (dt1 >  1.5*pi) dt1 -= 2.0*pi;
    if (dt2 < -1.5*pi) dt2 += 2.0*pi;
    if (dt2 >  1.5*pi) dt2 -= 2.0*pi;
    LaserPhase p;
    double s, c;
    sincos(t[0] - t[1], &s, &c); p.exp1 = make_double2(c, s);
    sincos(t[0] - t[2], &s, &c); p.exp2 = make_double2(c, s);
    p.djn = (-3.0*dt1 + dt2)/(2.0*dz); p.pad = 0.0;
    *out = p;
}

struct LaserSlices { const double2 *n00j00, *n00jp1, *n00jp2, *nm1j00, *nm1jp1, *nm1jp2, *np1jp1, *np1jp2; };   // null = beyond the head: 0

// right-hand side of the envelope equation (:700-749)
__global__ __launch_bounds__(256)
void k_laser_rhs (LaserSlices S, SlabView f, int c_chi, double chi0, int gshrink, const LaserPhase* ph, int step,
                  double dx, double dy, double dz, double c, double dt, double k0, double2* rhs,
                  double* mg_rhs, double* mg_acf_real, double* mg_acf_imag)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    const int nx = f.nx, ny = f.ny;
    if (i >= nx) return;
    const long o = (long)j*nx + i;
    auto ld = [o] (const double2* p) { return p ? p[o] : make_double2(0.0, 0.0); };
    const double2 exp1 = ph->exp1, exp2 = ph->exp2; const double djn = ph->djn;
    const double2* ls = (step == 0) ? S.n00j00 : S.nm1j00;
    double2 lap = make_double2(0.0, 0.0);
    if (i > 0 && i < nx - 1 && j > 0 && j < ny - 1) {
        const double2 c0 = ls[o], xp = ls[o + 1], xm = ls[o - 1], yp = ls[o + nx], ym = ls[o - nx];
        lap.x = (xp.x + xm.x - 2.0*c0.x)/(dx*dx) + (yp.x + ym.x - 2.0*c0.x)/(dy*dy);
        lap.y = (xp.y + xm.y - 2.0*c0.y)/(dx*dx) + (yp.y + ym.y - 2.0*c0.y)/(dy*dy);
    }
    // InterpolateChi (:334-407): chi of the slab inside the box shrunk by the guard width, the initial chi outside
    const bool inside = i >= gshrink && i < nx - gshrink && j >= gshrink && j < ny - gshrink;
    const double chi = inside ? f.p[c_chi*f.ns + f.off(i, j)] : chi0;
    const double2 an00j00 = ld(S.n00j00), anp1jp1 = ld(S.np1jp1), anp1jp2 = ld(S.np1jp2);
    const double cdz = 1.0/(c*dt*dz), cdt = 1.0/(c*dt);
    // chi term: 2 chi a_n on the right-hand side (fft, :713-740); multigrid with MG_average_rhs = 1: chi a_n (first step)
    // or chi a_{n-1}, the other half in the operator (:575-598)
    const bool mgs = (mg_rhs != nullptr);
    double2 r;
    if (step == 0) {
        const double2 an00jp1 = ld(S.n00jp1), an00jp2 = ld(S.n00jp2);
        r = cscale(8.0*cdz, cmul(csub(an00jp1, anp1jp1), exp1));
        r = cadd(r, cscale(2.0*cdz, cmul(csub(anp1jp2, an00jp2), exp2)));
        r = cadd(r, cscale((mgs ? 1.0 : 2.0)*chi, an00j00));
        r = csub(r, lap);
        r = cadd(r, cmul(make_double2(-6.0*cdz, 4.0*djn*cdt + 4.0*k0*cdt), an00j00));
    } else {
        const double2 anm1jp1 = ld(S.nm1jp1), anm1jp2 = ld(S.nm1jp2), anm1j00 = ld(S.nm1j00);
        r = cscale(4.0*cdz, cmul(csub(anm1jp1, anp1jp1), exp1));
        r = cadd(r, cscale(1.0*cdz, cmul(csub(anp1jp2, anm1jp2), exp2)));
        r = csub(r, cscale(4.0*cdt*cdt, an00j00));
        r = cadd(r, mgs ? cscale(chi, anm1j00) : cscale(2.0*chi, an00j00));
        r = csub(r, lap);
        r = cadd(r, cmul(make_double2(-3.0*cdz + 2.0*cdt*cdt, 2.0*djn*cdt + 2.0*k0*cdt), anm1j00));
    }
    if (!mgs) { rhs[o] = r; return; }
    const long plane = (long)nx*ny;
    mg_rhs[o] = r.x; mg_rhs[plane + o] = r.y;
    mg_acf_real[o] = ((step == 0) ? 6.0*cdz : 3.0*cdz + 2.0*cdt*cdt) + chi;                 // :520-523, 571-572
    if (o == 0) *mg_acf_imag = (step == 0) ? -4.0*(k0 + djn)*cdt : -2.0*(k0 + djn)*cdt;    // :524-525
}

// planar (Re plane, Im plane) -> complex
__global__ __launch_bounds__(256)
void k_laser_from_planar (const double* __restrict__ p, double2* __restrict__ out, long plane)
{
    const long o = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (o < plane) out[o] = make_double2(p[o], p[plane + o]);
}

// divide by -(k^2 + a) in Fourier space (:754-772)
__global__ __launch_bounds__(256)
void k_laser_divide (double2* rhs, int nx, int ny, double dkx, double dky, const LaserPhase* ph, int step,
                     double dz, double c, double dt, double k0)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= nx) return;
    const int imid = (nx + 1)/2, jmid = (ny + 1)/2;
    const double djn = ph->djn;
    const double cdz = 1.0/(c*dt*dz), cdt = 1.0/(c*dt);
    const double2 ac = (step == 0) ? make_double2(6.0*cdz, -4.0*(k0 + djn)*cdt)
                                   : make_double2(3.0*cdz + 2.0*cdt*cdt, -2.0*(k0 + djn)*cdt);
    const double kx = (i < imid) ? dkx*i : dkx*(i - nx);
    const double ky = (j < jmid) ? dky*j : dky*(j - ny);
    const double dr = kx*kx + ky*ky + ac.x, di = ac.y;
    const double m2 = dr*dr + di*di;
    const double2 inv = (m2 > 0.0) ? make_double2(dr/m2, -di/m2) : make_double2(0.0, 0.0);
    const long o = (long)j*nx + i;
    rhs[o] = cscale(-1.0, cmul(rhs[o], inv));
}

__global__ __launch_bounds__(256)
void k_laser_store (const double2* sol, double2* np1, long n, double inv_n)
{
    const long o = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (o < n) np1[o] = cscale(inv_n, sol[o]);
}

static LaserPars laser_pars (const hps_deck& d)
{
    return LaserPars{d.laser_a0, d.laser_w0, d.laser_L0, 2.0*3.14159265358979323846/d.laser_lambda0, d.laser_pos[0], d.laser_pos[1],
                     d.laser_pos[2], d.laser_zfoc};
}

int laser_create (Engine& E)
{
    LaserState* L = new LaserState;
    E.laser = L;
    const hps_deck& d = E.d;
    L->nx = d.nx; L->ny = d.ny; L->nz = d.nz;
    const size_t plane = (size_t)d.nx*d.ny, tot = plane*d.nz;
    HPS_HIP_CHECK(hipMalloc(&L->n00, tot*sizeof(double2)));
    HPS_HIP_CHECK(hipMalloc(&L->phase, sizeof(LaserPhase)));
    if (d.laser_solver >= 1 && d.dt != 0.0) {
        HPS_HIP_CHECK(hipMalloc(&L->nm1, tot*sizeof(double2)));
        HPS_HIP_CHECK(hipMalloc(&L->np1, tot*sizeof(double2)));
        HPS_HIP_CHECK(hipMemset(L->nm1, 0, tot*sizeof(double2)));
        HPS_HIP_CHECK(hipMemset(L->np1, 0, tot*sizeof(double2)));
    }
    if (d.laser_solver == 2 && d.dt != 0.0) {
        if (d.nx % 2 || d.ny % 2) { set_error("laser: the multigrid envelope solver needs even nx, ny (cell-centred hpmg box)"); return HPS_ERR_UNSUPPORTED; }
        if (int e = mg2_create_internal(d.nx, d.ny, E.gm.dx, E.gm.dy, &L->mg)) return e;
        HPS_HIP_CHECK(hipMalloc(&L->mg_sol, 2*plane*sizeof(double)));
        HPS_HIP_CHECK(hipMemset(L->mg_sol, 0, 2*plane*sizeof(double)));
        HPS_HIP_CHECK(hipMalloc(&L->mg_rhs, 2*plane*sizeof(double)));
        HPS_HIP_CHECK(hipMalloc(&L->mg_acf_real, plane*sizeof(double)));
        HPS_HIP_CHECK(hipMalloc(&L->mg_acf_imag, sizeof(double)));
    }
    if (d.laser_solver == 1 && d.dt != 0.0) {
        HPS_HIP_CHECK(hipMalloc(&L->work, plane*sizeof(double2)));
        static bool setup = false;
        if (!setup) { rocfft_setup(); setup = true; }
        const size_t len[2] = {(size_t)d.nx, (size_t)d.ny};
        if (rocfft_plan_create(&L->fwd, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_double, 2, len, 1, nullptr) != rocfft_status_success ||
            rocfft_plan_create(&L->bwd, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_double, 2, len, 1, nullptr) != rocfft_status_success) {
            set_error("laser: rocfft_plan_create failed"); return HPS_ERR_FFT;
        }
        size_t wf = 0, wb = 0;
        rocfft_plan_get_work_buffer_size(L->fwd, &wf);
        rocfft_plan_get_work_buffer_size(L->bwd, &wb);
        rocfft_execution_info_create(&L->info);
        if (std::max(wf, wb) > 0) {
            HPS_HIP_CHECK(hipMalloc(&L->fft_work, std::max(wf, wb)));
            rocfft_execution_info_set_work_buffer(L->info, L->fft_work, std::max(wf, wb));
        }
        rocfft_execution_info_set_stream(L->info, E.laser_stream());
    }
    return HPS_OK;
}

void laser_destroy (Engine& E) { delete E.laser; E.laser = nullptr; }

// start of a time step: the first one evaluates the Gaussian on every slice, later ones hand the time levels on
// (a_{n+1} -> a_n -> a_{n-1}; MultiBuffer.cpp:840-852, 913-925)
int laser_begin_step (Engine& E)
{
    LaserState* L = E.laser;
    const hps_deck& d = E.d;
    if (L->import_mode) {
        L->initialised = true;       // levels are filled slice by slice; the driver has set the step index
    } else if (!L->initialised) {
        const double zoff = 0.5*(d.lo[2] + d.hi[2] - E.gm.dz*(d.nz - 1));
        hipLaunchKernelGGL(k_laser_init, dim3(ceil_div(d.nx, 256), d.ny, d.nz), dim3(256), 0, E.st, L->n00, d.nx, d.ny, d.nz, laser_pars(d),
                           E.gm.dx, E.gm.dy, E.gm.dz, E.gm.xoff, E.gm.yoff, zoff);
        L->initialised = true; L->steps = 0;
    } else if (L->np1) {
        double2* old = L->nm1; L->nm1 = L->n00; L->n00 = L->np1; L->np1 = old;
        ++L->steps;
    }
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

int laser_update_aabs (Engine& E, int islice, double* sum_abs)
{
    LaserState* L = E.laser;
    SlabView f(E.slab);
    hipLaunchKernelGGL(k_laser_aabs, dim3(ceil_div(E.slab.jstride, 256), E.d.ny + 2*E.g), dim3(256), 0, E.st, f, E.c_aabs,
                       L->n00 + (size_t)islice*L->nx*L->ny, sum_abs);
    return HPS_OK;
}

// MultiLaser::AdvanceSlice: lasers.solver_type = fft (AdvanceSliceFFT) or multigrid (AdvanceSliceMG)
int laser_advance_slice (Engine& E, int islice)
{
    LaserState* L = E.laser;
    if (!L->np1) return HPS_OK;
    const hipStream_t ls = E.laser_stream();       // ordered against the engine's stream by Engine::fork_laser / join_laser
    const hps_deck& d = E.d;
    const size_t plane = (size_t)d.nx*d.ny;
    auto at = [&] (const double2* base, int sl) -> const double2* { return sl < d.nz ? base + (size_t)sl*plane : nullptr; };
    const LaserSlices S{at(L->n00, islice), at(L->n00, islice + 1), at(L->n00, islice + 2),
                        at(L->nm1, islice), at(L->nm1, islice + 1), at(L->nm1, islice + 2),
                        at(L->np1, islice + 1), at(L->np1, islice + 2)};
    const double k0 = 2.0*3.14159265358979323846/d.laser_lambda0;
    hipLaunchKernelGGL(k_laser_phase, dim3(1), dim3(1), 0, ls, S.n00j00, S.n00jp1, S.n00jp2, d.nx, d.ny, d.laser_use_phase, E.gm.dz, L->phase);
    const dim3 grid(ceil_div(d.nx, 256), d.ny), block(256);
    const double chi0 = d.plasma_density > 0.0 ? d.plasma_density*d.plasma_charge*d.plasma_charge*E.gm.mu0/d.plasma_mass : 0.0;
    hipLaunchKernelGGL(k_laser_rhs, grid, block, 0, ls, S, SlabView(E.slab), (int)HPS_C_CHI, chi0, E.g, L->phase, L->steps,
                       E.gm.dx, E.gm.dy, E.gm.dz, E.gm.c, d.dt, k0, L->work, L->mg_rhs, L->mg_acf_real, L->mg_acf_imag);
    if (L->mg) {
        // MultiLaser::AdvanceSliceMG (:430-608): hpmg system type 2, at most 200 V-cycles; the initial guess is the solution
        // of the slice solved before this one (np1j00 is left in place by ShiftLaserSlices, :208)
        int iters = 0;
        if (int e = mg2_solve_internal(L->mg, L->mg_sol, L->mg_rhs, L->mg_acf_real, L->mg_acf_imag,
                                       d.laser_mg_tol_rel > 0.0 ? d.laser_mg_tol_rel : 1.0e-4, d.laser_mg_tol_abs, 200, &iters, ls)) return e;
        L->mg_vcycles += iters;
        hipLaunchKernelGGL(k_laser_from_planar, dim3(ceil_div((long)plane, 256)), block, 0, ls, L->mg_sol, L->np1 + (size_t)islice*plane, (long)plane);
        HPS_HIP_CHECK(hipGetLastError());
        return HPS_OK;
    }
    void* buf[1] = {L->work};
    if (rocfft_execute(L->fwd, buf, nullptr, L->info) != rocfft_status_success) { set_error("laser: forward FFT failed"); return HPS_ERR_FFT; }
    hipLaunchKernelGGL(k_laser_divide, grid, block, 0, ls, L->work, d.nx, d.ny, 2.0*3.14159265358979323846/(d.hi[0] - d.lo[0]),
                       2.0*3.14159265358979323846/(d.hi[1] - d.lo[1]), L->phase, L->steps, E.gm.dz, E.gm.c, d.dt, k0);
    if (rocfft_execute(L->bwd, buf, nullptr, L->info) != rocfft_status_success) { set_error("laser: backward FFT failed"); return HPS_ERR_FFT; }
    hipLaunchKernelGGL(k_laser_store, dim3(ceil_div((long)plane, 256)), block, 0, ls, L->work, L->np1 + (size_t)islice*plane, (long)plane,
                       1.0/(double)plane);
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

// ring hand-off of the envelope (MultiBuffer.cpp:840-852, 913-925): a stage passes {a_{n+1}, a_n} of a slice on, the next
// one stores them as its {a_n, a_{n-1}}.  msg = [2][ny][nx] complex on the device; both asynchronous on the stream.
int laser_set_import (Engine& E, int on, int step)
{
    E.laser->import_mode = (on != 0); E.laser->steps = step;
    return HPS_OK;
}
int laser_export_slice (Engine& E, int islice, double* msg_dev)
{
    if (int e = E.join_laser()) return e;
    LaserState* L = E.laser;
    const size_t plane = (size_t)L->nx*L->ny;
    const double2* newest = L->np1 ? L->np1 : L->n00;
    HPS_HIP_CHECK(hipMemcpyAsync(msg_dev, newest + (size_t)islice*plane, plane*sizeof(double2), hipMemcpyDeviceToDevice, E.st));
    HPS_HIP_CHECK(hipMemcpyAsync(msg_dev + 2*plane, L->n00 + (size_t)islice*plane, plane*sizeof(double2), hipMemcpyDeviceToDevice, E.st));
    return HPS_OK;
}
int laser_import_slice (Engine& E, int islice, const double* msg_dev)
{
    LaserState* L = E.laser;
    const size_t plane = (size_t)L->nx*L->ny;
    HPS_HIP_CHECK(hipMemcpyAsync(L->n00 + (size_t)islice*plane, msg_dev, plane*sizeof(double2), hipMemcpyDeviceToDevice, E.st));
    if (L->nm1) HPS_HIP_CHECK(hipMemcpyAsync(L->nm1 + (size_t)islice*plane, msg_dev + 2*plane, plane*sizeof(double2), hipMemcpyDeviceToDevice, E.st));
    return HPS_OK;
}

// in-process hand-off (several steps in flight on one device): the same two planes straight from the engine that ran
// the previous step, on the receiving engine's stream (the caller has made it wait for the sender's slice event)
int laser_import_from (Engine& E, int islice, Engine& src)
{
    LaserState* L = E.laser; LaserState* P = src.laser;
    const size_t plane = (size_t)L->nx*L->ny;
    const double2* newest = P->np1 ? P->np1 : P->n00;
    // a_n -> a_{n-1} first: with one stage in flight source and destination are the same engine
    if (L->nm1) HPS_HIP_CHECK(hipMemcpyAsync(L->nm1 + (size_t)islice*plane, P->n00 + (size_t)islice*plane, plane*sizeof(double2), hipMemcpyDeviceToDevice, E.st));
    if (newest != L->n00) HPS_HIP_CHECK(hipMemcpyAsync(L->n00 + (size_t)islice*plane, newest + (size_t)islice*plane, plane*sizeof(double2), hipMemcpyDeviceToDevice, E.st));
    return HPS_OK;
}

long laser_mg_vcycles (Engine& E) { return E.laser ? E.laser->mg_vcycles : 0; }

int laser_copy_envelope (Engine& E, double* out_host)
{
    if (int e = E.join_laser()) return e;
    LaserState* L = E.laser;
    HPS_HIP_CHECK(hipStreamSynchronize(E.st));
    HPS_HIP_CHECK(hipMemcpy(out_host, L->n00, (size_t)L->nx*L->ny*L->nz*sizeof(double2), hipMemcpyDeviceToHost));
    return HPS_OK;
}

} // namespace hps
