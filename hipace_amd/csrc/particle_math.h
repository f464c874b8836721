// particle_math.h -- shared device math and host helpers of the particle kernels.
#ifndef HPS_PARTICLE_MATH_H_
#define HPS_PARTICLE_MATH_H_
#include "common.h"

namespace hps {

struct DepComps { int jx, jy, jz, rho, chi, rhomjz; };

struct PartConsts {
    double dx_inv, dy_inv, xoff, yoff;
    double c, c_inv;
    double a, b;          // kernel-specific prefactors
    double max_qsa;
    double plo0, plo1, phi0, phi1;
    double dz;
    int bc, can_ionize, temp_slice, n_subcycles;
    // depositions of the engine's own sheets: a particle is valid iff its weight is not 0 (every path that clears the valid
    // bit of idcpu -- QSA drop, absorbing boundary, lattice point without density -- also zeroes the weight, and a weight of
    // 0 deposits nothing): the kernels then do not read idcpu (8 of 56 bytes per particle).  0 for sheets of a caller.
    int valid_by_w;
    // laser envelope (use_laser of the reference's operators): slab component of |a|^2 (-1 = no laser) and the
    // operator's normalisation of it (laser_norm / laser_fac)
    int aabs; double laser_fac;
};

// ------------------------------------------------------------------------------------------
// current / charge deposition
// ------------------------------------------------------------------------------------------
struct D2 { double v, e; };   // dual number (value, first-order part)
__device__ __forceinline__ D2 operator+ (D2 a, D2 b) { return {a.v + b.v, a.e + b.e}; }
__device__ __forceinline__ D2 operator- (D2 a, D2 b) { return {a.v - b.v, a.e - b.e}; }
__device__ __forceinline__ D2 operator* (D2 a, D2 b) { return {a.v*b.v, a.e*b.v + a.v*b.e}; }
__device__ __forceinline__ D2 operator* (D2 a, double b) { return {a.v*b, a.e*b}; }
__device__ __forceinline__ D2 operator+ (D2 a, double b) { return {a.v + b, a.e}; }
__device__ __forceinline__ D2 operator- (D2 a, double b) { return {a.v - b, a.e}; }

// 1/x for the particle kernels: v_rcp_f64 + one Newton step (3 instructions, error <= 1 ulp) instead of the IEEE division
// sequence (v_div_scale x2, v_rcp, 5 fma, v_div_fmas, v_div_fixup).  The push takes 7 reciprocals of psi per particle
// and sub-cycle: a quarter of its fp64 instruction stream.  x is a positive normal number here (psi, checked by the
// deposition's QSA test).
__device__ __forceinline__ double fast_rcp (double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

struct Fld { double ExmBy, EypBx, Ez, Bxc, Byc, Bz; };

// d/dzeta of (ux, uy, psi) in the quasi-static frame; T = double or D2.
// gamma/psi = 1/2 psi^-2 (1 + u^2/c^2) + 1/2  (particles/pusher/PushPlasmaParticles.H:59-72)
template <class T>
__device__ __forceinline__ void zeta_derivs (const T& ux, const T& uy, const T& psi_inv, const Fld& F,
                                             double c_inv, double qmc, T& dux, T& duy, T& dpsi)
{
    const double ci2 = c_inv*c_inv;
    const T gamma_psi = (psi_inv*psi_inv)*0.5*( (ux*ux)*ci2 + (uy*uy)*ci2 + 1.0 ) + 0.5;
    dux = (gamma_psi*F.ExmBy + F.Byc + (uy*F.Bz)*psi_inv)*qmc;
    duy = (gamma_psi*F.EypBx - F.Bxc - (ux*F.Bz)*psi_inv)*qmc;
    dpsi = (((ux*F.ExmBy + uy*F.EypBx)*c_inv)*psi_inv - F.Ez)*(qmc*c_inv);
}

// |a|^2 at a particle with the plain deposition-order shape, read from the slab in global memory (doLaserGatherShapeN,
// particles/particles_utils/FieldGather.H:236-331); the second form also returns the centred x, y differences
template <int ORDER>
__device__ __forceinline__ double laser_gather (const SlabView& f, int aabs, double xmid, double ymid)
{
    double lx[ORDER + 1], ly[ORDER + 1];
    const int li = shape_weights<ORDER>(xmid, lx), lj = shape_weights<ORDER>(ymid, ly);
    double A = 0.0;
#pragma unroll
    for (int iy = 0; iy <= ORDER; ++iy)
#pragma unroll
        for (int ix = 0; ix <= ORDER; ++ix) A += lx[ix]*ly[iy]*f.p[aabs*f.ns + f.off(li + ix, lj + iy)];
    return A;
}
template <int ORDER>
__device__ __forceinline__ void laser_gather_grad (const SlabView& f, int aabs, double xmid, double ymid, double dx_inv, double dy_inv,
                                                   double& A, double& ADx, double& ADy)
{
    double lx[ORDER + 1], ly[ORDER + 1];
    const int li = shape_weights<ORDER>(xmid, lx), lj = shape_weights<ORDER>(ymid, ly);
    A = 0.0; ADx = 0.0; ADy = 0.0;
#pragma unroll
    for (int iy = 0; iy <= ORDER; ++iy)
#pragma unroll
        for (int ix = 0; ix <= ORDER; ++ix) {
            const double* a = f.p + aabs*f.ns + f.off(li + ix, lj + iy);
            const double w = lx[ix]*ly[iy];
            A += w*a[0];
            ADx += w*0.5*dx_inv*(a[1] - a[-1]);
            ADy += w*0.5*dy_inv*(a[f.js] - a[-f.js]);
        }
}

// the same with the ponderomotive terms of a laser envelope (PushPlasmaParticles.H:59-72): A = Aabssq_norm,
// ADx, ADy = AabssqD{x,y}_norm
struct LaserFld { double A, ADx, ADy; };
template <class T>
__device__ __forceinline__ void zeta_derivs_laser (const T& ux, const T& uy, const T& psi_inv, const Fld& F, const LaserFld& Lf,
                                                   double c_inv, double qmc, T& dux, T& duy, T& dpsi)
{
    const double ci2 = c_inv*c_inv;
    const T gamma_psi = (psi_inv*psi_inv)*0.5*( (ux*ux)*ci2 + (uy*uy)*ci2 + (1.0 + Lf.A) ) + 0.5;
    dux = (gamma_psi*F.ExmBy + F.Byc + (uy*F.Bz)*psi_inv)*qmc - psi_inv*Lf.ADx;
    duy = (gamma_psi*F.EypBx - F.Bxc - (ux*F.Bz)*psi_inv)*qmc - psi_inv*Lf.ADy;
    dpsi = (((ux*F.ExmBy + uy*F.EypBx)*c_inv)*psi_inv - F.Ez)*(qmc*c_inv);
}
__device__ __forceinline__ void taylor2_substep_laser (double& ux, double& uy, double& psi, const Fld& F, const LaserFld& Lf,
                                                       double c_inv, double qmc, double sdz)
{
    const double psi_inv = fast_rcp(psi);
    double dux, duy, dpsi;
    zeta_derivs_laser<double>(ux, uy, psi_inv, F, Lf, c_inv, qmc, dux, duy, dpsi);
    const D2 uxd{ux, dux}, uyd{uy, duy}, pid{psi_inv, -psi_inv*psi_inv*dpsi};
    D2 ddux, dduy, ddpsi;
    zeta_derivs_laser<D2>(uxd, uyd, pid, F, Lf, c_inv, qmc, ddux, dduy, ddpsi);
    const double h2 = 0.5*sdz*sdz;
    ux += sdz*dux + h2*ddux.e;
    uy += sdz*duy + h2*dduy.e;
    psi += sdz*dpsi + h2*ddpsi.e;
}

__device__ __forceinline__ void taylor2_substep (double& ux, double& uy, double& psi, const Fld& F,
                                                 double c_inv, double qmc, double sdz)
{
    const double psi_inv = fast_rcp(psi);
    double dux, duy, dpsi;
    zeta_derivs<double>(ux, uy, psi_inv, F, c_inv, qmc, dux, duy, dpsi);
    const D2 uxd{ux, dux}, uyd{uy, duy}, pid{psi_inv, -psi_inv*psi_inv*dpsi};
    D2 ddux, dduy, ddpsi;
    zeta_derivs<D2>(uxd, uyd, pid, F, c_inv, qmc, ddux, dduy, ddpsi);
    const double h2 = 0.5*sdz*sdz;
    ux += sdz*dux + h2*ddux.e;
    uy += sdz*duy + h2*dduy.e;
    psi += sdz*dpsi + h2*ddpsi.e;
}

// The same sub-step with everything that does not change over a particle's six sub-steps taken out of them (the gathered
// fields are fixed while the momenta advance): the field products with q/(m c) once per particle, the dual-number pass
// written out by hand -- d/dzeta of every factor once, no second evaluation of the value parts.  Same formulas
// (PushPlasmaParticles.H:59-72 and their zeta derivative), 40 fp64 instructions per sub-step instead of the 62 the compiler
// makes of taylor2_substep (the tile push spends most of its time issuing them: 6 sub-steps per particle); the results differ
// from taylor2_substep's by the rounding of the re-associated products (1e-16 relative per operation).
struct PushForce { double A1, A2, B1, B2, BZ, E1, E2, EZ, ci2, onepA, ADx, ADy; };
__device__ __forceinline__ PushForce push_force (const Fld& F, const LaserFld& Lf, double c_inv, double qmc)
{
    const double qc = qmc*c_inv;
    return PushForce{qmc*F.ExmBy, qmc*F.EypBx, qmc*F.Byc, -(qmc*F.Bxc), qmc*F.Bz, (F.ExmBy*c_inv)*qc, (F.EypBx*c_inv)*qc, -(F.Ez*qc),
                     c_inv*c_inv, 1.0 + Lf.A, Lf.ADx, Lf.ADy};
}
template <bool LASER>
__device__ __forceinline__ void taylor2_substep_pre (double& ux, double& uy, double& psi, const PushForce& P, double sdz, double h2)
{
    const double p = fast_rcp(psi);
    const double uxc = ux*P.ci2, uyc = uy*P.ci2;
    const double s = fma(uxc, ux, fma(uyc, uy, LASER ? P.onepA : 1.0));       // 1 + u^2/c^2 [+ |a|^2/2]
    const double p2 = p*p, hp2 = 0.5*p2;
    const double g = fma(hp2, s, 0.5);                                          // gamma/psi
    const double uxp = ux*p, uyp = uy*p;
    double dux = fma(g, P.A1, fma(uyp, P.BZ, LASER ? fma(-p, P.ADx, P.B1) : P.B1));
    double duy = fma(g, P.A2, fma(-uxp, P.BZ, LASER ? fma(-p, P.ADy, P.B2) : P.B2));
    const double dpsi = fma(uxp, P.E1, fma(uyp, P.E2, P.EZ));
    // zeta derivatives of the factors
    const double dp = -(p2*dpsi);                                               // d(1/psi)
    const double t = fma(uxc, dux, uyc*duy);                                    // 1/2 d(s)
    const double dg = fma(p2, t, s*(p*dp));                                     // d(gamma/psi) = hp2 ds + s d(hp2)
    const double duxp = fma(dux, p, ux*dp), duyp = fma(duy, p, uy*dp);
    double ddux = fma(dg, P.A1, duyp*P.BZ);
    double dduy = fma(dg, P.A2, -(duxp*P.BZ));
    const double ddpsi = fma(duxp, P.E1, duyp*P.E2);
    if (LASER) { ddux = fma(-dp, P.ADx, ddux); dduy = fma(-dp, P.ADy, dduy); }
    ux = fma(h2, ddux, fma(sdz, dux, ux));
    uy = fma(h2, dduy, fma(sdz, duy, uy));
    psi = fma(h2, ddpsi, fma(sdz, dpsi, psi));
}

// particle boundary; returns true if the particle was absorbed
__device__ __forceinline__ bool apply_particle_bc (const PartConsts& k, double& x, double& y,
                                                   double& ux, double& uy)
{
    if (x < k.plo0 || y < k.plo1 || x > k.phi0 || y > k.phi1) {
        const double lx = k.phi0 - k.plo0, ly = k.phi1 - k.plo1;
        if (k.bc == HPS_BC_REFLECTING) {
            x = fmod(x - k.plo0, 2*lx); if (x < 0) x += 2*lx; x += k.plo0;
            if (x > k.phi0) { x = 2*k.phi0 - x; ux = -ux; }
            y = fmod(y - k.plo1, 2*ly); if (y < 0) y += 2*ly; y += k.plo1;
            if (y > k.phi1) { y = 2*k.phi1 - y; uy = -uy; }
        } else if (k.bc == HPS_BC_PERIODIC) {
            x = fmod(x - k.plo0, lx); if (x < 0) x += lx; x += k.plo0;
            y = fmod(y - k.plo1, ly); if (y < 0) y += ly; y += k.plo1;
        } else {
            return true;
        }
    }
    return false;
}

// ---- ADK field ionisation (ionization.hip; fused into the ions' LDS-tile push in particles_tiled.hip) -----------------
// uniform deviate in [0, 1) of (seed, ion, time step, slice): counter based (two rounds of the splitmix64 finaliser), same
// integer arithmetic as the oracle's ion_uniform
__device__ __forceinline__ double ion_uniform (unsigned long long seed, unsigned long long uid, unsigned long long step,
                                               unsigned long long islice)
{
    unsigned long long z = seed + 0x9E3779B97F4A7C15ULL*(uid + 1) + 0xBF58476D1CE4E5B9ULL*(step + 1) + 0x94D049BB133111EBULL*(islice + 1);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
        z ^= z >> 27; z *= 0x94D049BB133111EBULL;
        z ^= z >> 31;
    }
    return (double)(z >> 11)*(1.0/9007199254740992.0);
}

struct IonArgs {
    hps_plasma el;                       // product species: electrons are appended behind cnt[0] particles
    const double* adk;                   // [prefactor[Z] | exp_prefactor[Z] | power[Z] | field below which nothing can ionise[Z]]
    unsigned long long* cnt;             // {electrons in el, overflow flag, workgroups done, ionisations so far}
    volatile long long* host; long long seq;      // mapped host memory the last workgroup posts {cnt[0], cnt[1], cnt[3], seq} to
    double E0, clightsq_inv; int Z;
    unsigned long long seed, step, islice;
    long cap;                            // capacity of el's arrays
    int product_init_lev;                // ion_lev of a released electron: the product species' m_init_ion_lev (PlasmaParticleContainer.cpp:430)
    int* tile_flag;                      // [tiles of the ion tiling] 1 = the tile holds a charged ion (written by the ions' tile push)
    const double* fbound;                // [5][tiles] max over the tile's cells of |d_x psi|, |d_y psi| (staggered), |Bx|, |By|, |Ez|
                                         // (k_ion_field_bounds), or null; ntx, nty = tiles per direction
    int fb_ntx, fb_nty; double fb_dx_inv, fb_dy_inv, fb_c;
};

// One ion's decision (PlasmaParticleContainer.cpp:352-372) from the fields gathered at (x_prev, y_prev): Ex, Ey, Ez in the
// engine's units.  Returns true if the ion loses an electron on this slice.
__device__ __forceinline__ bool adk_decide (const IonArgs& a, double Ex, double Ey, double Ez, double ux, double uy, double psi,
                                            int lev, uint64_t id)
{
    const double Ep = sqrt(Ex*Ex + Ey*Ey + Ez*Ez)*a.E0;
    const double gammap = (1.0 + ux*ux*a.clightsq_inv + uy*uy*a.clightsq_inv + psi*psi)/(2.0*psi);
    // Below adk[3 Z + lev] the rate is so small (w_dtau < 1e-20 for gamma/psi = 1) that 1 - exp(-w_dtau) is exactly 0 in
    // double precision -- no draw can be below it: skip the pow and the two exp (most atoms of a slice sit in such a field).
    // The bound leaves a factor 1e3 for gamma/psi; beyond that the full expression is evaluated.
    if (Ep < a.adk[3*a.Z + lev] && gammap < 1.0e3*psi) return false;
    // gamma / psi completes dt for the quasi-static frame (:362-366)
    const double w_dtau = gammap/psi*a.adk[lev]*pow(Ep, a.adk[2*a.Z + lev])*exp(a.adk[a.Z + lev]/Ep);
    const double p = 1.0 - exp(-w_dtau);
    const unsigned long long uid = ((id >> 24) & ((1ULL << 39) - 1)) - 1;
    return ion_uniform(a.seed, uid, a.step, a.islice) < p;
}

// Upper bound of the field an atom of tile (tx, ty) can see, from the block maxima of the 3 x 3 tiles around it (the
// stencil of a particle of the tile stays inside the tile's halo < one tile).  The gathered ExmBy is a convex combination
// of staggered differences of psi (the derivative of the quadratic shape is the difference of two linear ones), Ez, Bx, By
// are convex combinations of cell values: |E| <= sqrt((max|d_x psi|/dx + c max|By|)^2 + (max|d_y psi|/dy + c max|Bx|)^2 +
// max|Ez|^2).  True = no neutral atom at rest in the tile can ionise (adk_decide's own early-out would take each of them).
__device__ __forceinline__ bool adk_tile_below_threshold (const IonArgs& a, int tx, int ty)
{
    double m[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    const long nt = (long)a.fb_ntx*a.fb_nty;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int bx = tx + dx, by = ty + dy;
            if (bx < 0 || by < 0 || bx >= a.fb_ntx || by >= a.fb_nty) continue;
            const long b = (long)by*a.fb_ntx + bx;
#pragma unroll
            for (int q = 0; q < 5; ++q) m[q] = fmax(m[q], a.fbound[q*nt + b]);
        }
    const double ex = m[0]*a.fb_dx_inv + a.fb_c*m[3], ey = m[1]*a.fb_dy_inv + a.fb_c*m[2];
    const double Ep = sqrt(ex*ex + ey*ey + m[4]*m[4])*a.E0;
    return Ep*(1.0 + 1.0e-9) < a.adk[3*a.Z];
}

// The lanes of a wave that ionise take a block of electron slots with ONE atomic (ballot + popcount) and write their
// electrons: at rest on the ion, with its weight (PlasmaParticleContainer.cpp:404-433); id 2, level 0 of the mesh.
// Call with `ionize` false from lanes that do not ionise; the lanes of the wave that have left the loop do not matter.
__device__ __forceinline__ void adk_emit (const IonArgs& a, bool ionize, double x, double y, double xprev, double yprev, double w)
{
    const unsigned long long mask = __ballot(ionize);
    if (mask == 0ULL) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned long long base = 0;
    if (lane == leader) {
        base = atomicAdd(a.cnt, (unsigned long long)__popcll(mask));
        atomicAdd(a.cnt + 3, (unsigned long long)__popcll(mask));
    }
    base = __shfl(base, leader);
    if (ionize) {
        const long q = (long)base + __popcll(mask & ((1ULL << lane) - 1ULL));
        if (q < a.cap) {
            const hps_plasma& el = a.el;
            el.x[q] = x; el.y[q] = y; el.w[q] = w;
            el.ux[q] = 0.0; el.uy[q] = 0.0; el.psi[q] = 1.0;
            if (el.x_prev != el.x) el.x_prev[q] = xprev;
            if (el.y_prev != el.y) el.y_prev[q] = yprev;
            el.ux_half[q] = 0.0; el.uy_half[q] = 0.0; el.psi_half[q] = 1.0;
            el.idcpu[q] = HPS_ID_VALID | (2ULL << 24);
            el.ion_lev[q] = a.product_init_lev;
        } else {
            atomicExch(a.cnt + 1, 1ULL);
        }
    }
}

// the last workgroup of the launch posts {electrons, overflow, ionisations, seq} to the host (seq last, behind a
// system-scope fence); call from every workgroup after its last adk_emit
__device__ __forceinline__ void adk_post (const IonArgs& a)
{
    // every wave waits for the acknowledgement of its own counter atomics (adk_emit) before the workgroup is counted as done.
    // (Not a __threadfence(): a device-scope release fence on this GPU writes the XCD's L2 back -- once per workgroup that
    // was most of this kernel's time on tiles that have nothing to ionise.  The counters are device-scope atomics, read
    // back below by device-scope atomic loads; the electrons' arrays are read by the NEXT kernel.)
    HPS_OWN_ATOMICS_ACKNOWLEDGED();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(a.cnt + 2, 1ULL) == (unsigned long long)gridDim.x - 1ULL) {
            __hip_atomic_store(a.cnt + 2, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.host[0] = (long long)__hip_atomic_load(a.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.host[1] = (long long)__hip_atomic_load(a.cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.host[2] = (long long)__hip_atomic_load(a.cnt + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            a.host[3] = a.seq;
        }
    }
}

inline PartConsts base_consts (const hps_geom& g)
{
    PartConsts k{};
    k.dx_inv = 1.0/g.dx; k.dy_inv = 1.0/g.dy; k.xoff = g.xoff; k.yoff = g.yoff;
    k.c = g.c; k.c_inv = 1.0/g.c;
    k.plo0 = g.plo[0]; k.plo1 = g.plo[1]; k.phi0 = g.phi[0]; k.phi1 = g.phi[1];
    k.bc = g.bc;
    k.aabs = -1; k.laser_fac = 0.0;
    return k;
}

inline double invvol_of (const hps_geom& g)
{
    // normalised units: 1 on level 0; SI: charge -> charge density (PlasmaDepositCurrent.cpp:71-73)
    const double dxi = 1.0/g.dx, dyi = 1.0/g.dy, dzi = 1.0/g.dz;
    return g.normalized ? g.dx*g.dy*dxi*dyi : dxi*dyi*dzi;
}

inline int check_stencil (const hps_slab& s, int need_guards, const char* what)
{
    if (s.p == nullptr || s.nx <= 0 || s.ny <= 0 || s.ng < need_guards) {
        set_error(std::string(what) + ": slab needs at least " + std::to_string(need_guards) + " guard cells");
        return HPS_ERR_ARG;
    }
    return HPS_OK;
}


} // namespace hps
#endif
