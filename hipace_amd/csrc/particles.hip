// particles.hip -- plasma-sheet particle kernels for gfx950:
//   hps_deposit_current   (scatter of jx,jy,jz,rho,chi,rhomjz)
//   hps_explicit_deposit  (scatter of the explicit-solver sources Sy,Sx)
//   hps_advance_plasma    (field gather + leapfrog push with dual-number 2nd-order sub-steps)
// One particle per lane, SoA particle arrays read fully coalesced, 1-D shape factors hoisted out
// of the stencil loops, native fp64 global atomics (global_atomic_add_f64) for the scatter.
// Reference semantics: particles/deposition/PlasmaDepositCurrent.cpp:155-246,
// particles/deposition/ExplicitDeposition.cpp:140-261,
// particles/pusher/PlasmaParticleAdvance.cpp:92-217.
#include "common.h"
#include "particle_math.h"

namespace hps {

template <int ORDER>
__global__ __launch_bounds__(256)
void k_deposit_current (SlabView f, hps_plasma pl, DepComps cm, PartConsts k, int* n_qsa)
{
    const long ip = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (ip >= pl.n) return;
    const uint64_t id = pl.idcpu[ip];
    if (!(id & HPS_ID_VALID)) return;

    const double psi_inv = 1.0/pl.psi[ip];
    const double vx_c = pl.ux[ip]*psi_inv;
    const double vy_c = pl.uy[ip]*psi_inv;
    double q_invvol = k.a*pl.w[ip];           // charge * invvol * w
    double q_mu0_mass = k.b;                  // charge * mu0 / mass
    if (k.can_ionize) { const double il = (double)pl.ion_lev[ip]; q_invvol *= il; q_mu0_mass *= il; }

    // laser: |a|^2 at the particle with the plain shape (doLaserGatherShapeN, FieldGather.H:298-331), times laser_norm
    double Aabssq = 0.0;
    if (k.aabs >= 0) {
        double lx[ORDER + 1], ly[ORDER + 1];
        const int li = shape_weights<ORDER>((pl.x[ip] - k.xoff)*k.dx_inv, lx);
        const int lj = shape_weights<ORDER>((pl.y[ip] - k.yoff)*k.dy_inv, ly);
#pragma unroll
        for (int iy = 0; iy <= ORDER; ++iy)
#pragma unroll
            for (int ix = 0; ix <= ORDER; ++ix) Aabssq += lx[ix]*ly[iy]*f.p[k.aabs*f.ns + f.off(li + ix, lj + iy)];
        Aabssq *= k.laser_fac*(k.can_ionize ? (double)pl.ion_lev[ip]*(double)pl.ion_lev[ip] : 1.0);
    }
    const double gamma_psi = 0.5*((1.0 + 0.5*Aabssq)*psi_inv*psi_inv + vx_c*vx_c*k.c_inv*k.c_inv
                                  + vy_c*vy_c*k.c_inv*k.c_inv + 1.0);
    if (gamma_psi < 0.0 || gamma_psi > k.max_qsa || psi_inv < 0.0) {
        // particle violates the quasi-static approximation: drop it
        if (n_qsa) atomicAdd(n_qsa, 1);
        pl.w[ip] = 0.0;
        pl.idcpu[ip] = id & ~HPS_ID_VALID;
        if (pl.psi_half) pl.psi_half[ip] = 0.0;      // (Tiling::valid_by_psi: the tile push of the engine's sheet reads validity from here)
        return;
    }

    double sx[ORDER + 1], sy[ORDER + 1];
    const int i0 = shape_weights<ORDER>((pl.x[ip] - k.xoff)*k.dx_inv, sx);
    const int j0 = shape_weights<ORDER>((pl.y[ip] - k.yoff)*k.dy_inv, sy);

    const double wz = (gamma_psi - 1.0)*k.c;
    const double wchi = q_mu0_mass*psi_inv;
#pragma unroll
    for (int iy = 0; iy <= ORDER; ++iy) {
        const long row = f.off(i0, j0 + iy);
#pragma unroll
        for (int ix = 0; ix <= ORDER; ++ix) {
            const double cd = q_invvol*sx[ix]*sy[iy];
            double* p = f.p + row + ix;
            if (cm.jx >= 0) {
                atomic_add_f64(p + cm.jx*f.ns, cd*vx_c);
                atomic_add_f64(p + cm.jy*f.ns, cd*vy_c);
            }
            if (cm.jz >= 0)     atomic_add_f64(p + cm.jz*f.ns, cd*wz);
            if (cm.rho >= 0)    atomic_add_f64(p + cm.rho*f.ns, cd*gamma_psi);
            if (cm.chi >= 0)    atomic_add_f64(p + cm.chi*f.ns, cd*wchi);
            if (cm.rhomjz >= 0) atomic_add_f64(p + cm.rhomjz*f.ns, cd);
        }
    }
}

// ------------------------------------------------------------------------------------------
// explicit-solver source deposition (Sy, Sx)
// ------------------------------------------------------------------------------------------
template <int ORDER, int DT>
__global__ __launch_bounds__(256)
void k_explicit_deposit (SlabView f, hps_plasma pl, int cBz, int cEz, int cExmBy, int cEypBx,
                         int cSy, int cSx, PartConsts k)
{
    const long ip = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (ip >= pl.n) return;
    if (!(pl.idcpu[ip] & HPS_ID_VALID)) return;

    constexpr int NS = ORDER + DT + 1;
    const double psi_inv = 1.0/pl.psi[ip];
    const double vx = pl.ux[ip]*psi_inv*k.c_inv;
    const double vy = pl.uy[ip]*psi_inv*k.c_inv;
    double q_invvol_mu0 = k.a;      // charge * invvol * mu0
    double q_mass = k.b;            // charge / mass
    if (k.can_ionize) { const double il = (double)pl.ion_lev[ip]; q_invvol_mu0 *= il; q_mass *= il; }
    const double cdm = q_invvol_mu0*pl.w[ip];
    const double xmid = (pl.x[ip] - k.xoff)*k.dx_inv;
    const double ymid = (pl.y[ip] - k.yoff)*k.dy_inv;
    // laser: |a|^2 gathered first with the plain shape (ExplicitDeposition.cpp:167-174), its gradient per stencil cell
    double Aabssq = 0.0;
    if (k.aabs >= 0) {
        double lx[ORDER + 1], ly[ORDER + 1];
        const int li = shape_weights<ORDER>(xmid, lx), lj = shape_weights<ORDER>(ymid, ly);
#pragma unroll
        for (int iy = 0; iy <= ORDER; ++iy)
#pragma unroll
            for (int ix = 0; ix <= ORDER; ++ix) Aabssq += lx[ix]*ly[iy]*f.p[k.aabs*f.ns + f.off(li + ix, lj + iy)];
        Aabssq *= k.laser_fac*q_mass*q_mass;
    }
    const double gp = 0.5*((1.0 + 0.5*Aabssq)*psi_inv*psi_inv + vx*vx + vy*vy + 1.0);

    double sx[NS], dsx[NS], sy[NS], dsy[NS];
    int i0, j0;
    if constexpr (DT == 2) { i0 = centred_weights<ORDER>(xmid, sx, dsx); j0 = centred_weights<ORDER>(ymid, sy, dsy); }
    else                   { i0 = nodal_weights<ORDER>(xmid, sx, dsx);   j0 = nodal_weights<ORDER>(ymid, sy, dsy); }

    const double qp = q_mass*psi_inv;
    // field-independent coefficient combinations
    const double vxvy = vx*vy;
    const double gy = gp - vy*vy;      // multiplies EypBx in Sy
    const double gx = gp - vx*vx;      // multiplies ExmBy in Sx
#pragma unroll
    for (int iy = 0; iy < NS; ++iy) {
        const long row = f.off(i0, j0 + iy);
#pragma unroll
        for (int ix = 0; ix < NS; ++ix) {
            if (DT == 2 && (ix == 0 || ix == NS - 1) && (iy == 0 || iy == NS - 1)) continue;
            double* p = f.p + row + ix;
            const double Bz = p[cBz*f.ns];
            const double Ez = p[cEz*f.ns];
            const double ExmBy = p[cExmBy*f.ns];
            const double EypBx = p[cEypBx*f.ns];
            const double ss = sx[ix]*sy[iy];
            const double dxs = dsx[ix]*sy[iy]*k.dx_inv;
            const double sdy = sx[ix]*dsy[iy]*k.dy_inv;
            double ADx = 0.0, ADy = 0.0;
            if (k.aabs >= 0 && ss != 0.0) {      // (:215-226)
                const double* a = p + k.aabs*f.ns;
                ADx = (a[1] - a[-1])*0.5*k.dx_inv*k.laser_fac*k.c;
                ADy = (a[f.js] - a[-f.js])*0.5*k.dy_inv*k.laser_fac*k.c;
            }
            const double sy_add = cdm*(
                - ss*( -Bz*vx + (Ez*vy + ExmBy*(-vxvy) + EypBx*gy)*k.c_inv - 0.25*ADy*qp )*qp
                + ( -dxs*(-vxvy) - sdy*(gy - 1.0) )*k.c);
            const double sx_add = cdm*(
                + ss*( Bz*vy + (Ez*vx + ExmBy*gx + EypBx*(-vxvy))*k.c_inv - 0.25*ADx*qp )*qp
                + ( dxs*(gx - 1.0) + sdy*(-vxvy) )*k.c);
            atomic_add_f64(p + cSy*f.ns, sy_add);
            atomic_add_f64(p + cSx*f.ns, sx_add);
        }
    }
}

template <int ORDER>
__global__ __launch_bounds__(256)
void k_advance_plasma (SlabView f, hps_plasma pl, int cPsi, int cEz, int cBx, int cBy, int cBz,
                       PartConsts k)
{
    const long ip = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (ip >= pl.n) return;
    const uint64_t id = pl.idcpu[ip];
    if (!(id & HPS_ID_VALID)) return;

    constexpr int NS = ORDER + 2;
    double qmc = k.a;    // charge / (mass c)
    if (k.can_ionize) qmc *= (double)pl.ion_lev[ip];

    for (int isc = 0; isc < k.n_subcycles; ++isc) {
        double xp = pl.x_prev[ip];
        double yp = pl.y_prev[ip];

        // gather at (x_prev, y_prev): E-like fields from -grad(Psi) via the nodal-derivative set
        double sx[NS], dsx[NS], sy[NS], dsy[NS];
        const int i0 = nodal_weights<ORDER>((xp - k.xoff)*k.dx_inv, sx, dsx);
        const int j0 = nodal_weights<ORDER>((yp - k.yoff)*k.dy_inv, sy, dsy);
        Fld F{0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int iy = 0; iy < NS; ++iy) {
            const long row = f.off(i0, j0 + iy);
#pragma unroll
            for (int ix = 0; ix < NS; ++ix) {
                const double* p = f.p + row + ix;
                const double psi_c = p[cPsi*f.ns];
                const double ss = sx[ix]*sy[iy];
                F.ExmBy += (dsx[ix]*sy[iy])*psi_c*k.dx_inv;
                F.EypBx += (sx[ix]*dsy[iy])*psi_c*k.dy_inv;
                F.Ez  += ss*p[cEz*f.ns];
                F.Bxc += ss*p[cBx*f.ns];
                F.Byc += ss*p[cBy*f.ns];
                F.Bz  += ss*p[cBz*f.ns];
            }
        }
        F.Bxc *= k.c;
        F.Byc *= k.c;
        // laser: |a|^2 and its centred gradient with the plain shape (doLaserGatherShapeN, FieldGather.H:236-280)
        LaserFld Lf{0.0, 0.0, 0.0};
        if (k.aabs >= 0) {
            double lx[ORDER + 1], ly[ORDER + 1];
            const int li = shape_weights<ORDER>((xp - k.xoff)*k.dx_inv, lx);
            const int lj = shape_weights<ORDER>((yp - k.yoff)*k.dy_inv, ly);
#pragma unroll
            for (int iy = 0; iy <= ORDER; ++iy)
#pragma unroll
                for (int ix = 0; ix <= ORDER; ++ix) {
                    const double* a = f.p + k.aabs*f.ns + f.off(li + ix, lj + iy);
                    const double w = lx[ix]*ly[iy];
                    Lf.A += w*a[0];
                    Lf.ADx += w*0.5*k.dx_inv*(a[1] - a[-1]);
                    Lf.ADy += w*0.5*k.dy_inv*(a[f.js] - a[-f.js]);
                }
            const double ln = k.laser_fac*(k.can_ionize ? (double)pl.ion_lev[ip]*(double)pl.ion_lev[ip] : 1.0);
            Lf.A *= 0.5*ln; Lf.ADx *= 0.25*k.c*ln; Lf.ADy *= 0.25*k.c*ln;
        }

        const double dz = k.dz;
        const double sdz = dz*0.25;
        double ux = pl.ux_half[ip], uy = pl.uy_half[ip], psi = pl.psi_half[ip];

        // momenta t-1/2 -> t+1/2 with the fields at t (4 second-order Taylor sub-steps)
        if (k.aabs >= 0) {
#pragma unroll 1
            for (int s = 0; s < 4; ++s) taylor2_substep_laser(ux, uy, psi, F, Lf, k.c_inv, qmc, sdz);
        } else {
#pragma unroll 1
            for (int s = 0; s < 4; ++s) taylor2_substep(ux, uy, psi, F, k.c_inv, qmc, sdz);
        }

        // positions t -> t+1 with the momenta at t+1/2
        const double pinv = 1.0/psi;
        xp += dz*k.c_inv*(ux*pinv);
        yp += dz*k.c_inv*(uy*pinv);

        if (apply_particle_bc(k, xp, yp, ux, uy)) {
            pl.w[ip] = 0.0;
            pl.idcpu[ip] = id & ~HPS_ID_VALID;
        pl.psi_half[ip] = 0.0;      // (Tiling::valid_by_psi: the tile push of the engine's sheet reads validity from here)
            return;
        }
        pl.x[ip] = xp;
        pl.y[ip] = yp;
        if (!k.temp_slice) {
            pl.ux_half[ip] = ux; pl.uy_half[ip] = uy; pl.psi_half[ip] = psi;
            pl.x_prev[ip] = xp;  pl.y_prev[ip] = yp;
        }

        // extra half push t+1/2 -> t+1: time-centred state used by the deposition only
        if (k.aabs >= 0) {
#pragma unroll 1
            for (int s = 0; s < 2; ++s) taylor2_substep_laser(ux, uy, psi, F, Lf, k.c_inv, qmc, sdz);
        } else {
#pragma unroll 1
            for (int s = 0; s < 2; ++s) taylor2_substep(ux, uy, psi, F, k.c_inv, qmc, sdz);
        }
        pl.ux[ip] = ux; pl.uy[ip] = uy; pl.psi[ip] = psi;
    }
}

} // namespace hps

using namespace hps;

static int deposit_current_impl (hps_slab slab, hps_plasma pl, hps_geom g, const int comp[6],
                                 double charge, double mass, int order, double max_qsa,
                                 int can_ionize, int* n_qsa, hps_stream stream, int aabs_comp)
{
    HPS_REQUIRE(aabs_comp >= -1 && aabs_comp < slab.ncomp, "hps_deposit_current: bad aabs component");
    HPS_REQUIRE(order >= 0 && order <= 3, "hps_deposit_current: depos_order must be 0..3");
    if (int e = check_stencil(slab, (order + 1)/2, "hps_deposit_current")) return e;
    for (int c = 0; c < 6; ++c) HPS_REQUIRE(comp[c] >= -1 && comp[c] < slab.ncomp, "hps_deposit_current: bad component");
    HPS_REQUIRE((comp[0] < 0) == (comp[1] < 0), "hps_deposit_current: jx and jy are deposited together");
    if (pl.n == 0) return HPS_OK;
    PartConsts k = base_consts(g);
    k.a = charge*invvol_of(g);
    k.b = charge*g.mu0/mass;
    k.max_qsa = max_qsa; k.can_ionize = can_ionize;
    k.aabs = aabs_comp; k.laser_fac = (charge/g.q_e)*(g.m_e/mass)*(charge/g.q_e)*(g.m_e/mass);     // laser_norm
    DepComps cm{comp[0], comp[1], comp[2], comp[3], comp[4], comp[5]};
    const dim3 grid(ceil_div(pl.n, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    SlabView f(slab);
    switch (order) {
        case 0: hipLaunchKernelGGL(k_deposit_current<0>, grid, block, 0, st, f, pl, cm, k, n_qsa); break;
        case 1: hipLaunchKernelGGL(k_deposit_current<1>, grid, block, 0, st, f, pl, cm, k, n_qsa); break;
        case 2: hipLaunchKernelGGL(k_deposit_current<2>, grid, block, 0, st, f, pl, cm, k, n_qsa); break;
        default: hipLaunchKernelGGL(k_deposit_current<3>, grid, block, 0, st, f, pl, cm, k, n_qsa); break;
    }
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

extern "C" int hps_deposit_current (hps_slab slab, hps_plasma pl, hps_geom g, const int comp[6],
                                    double charge, double mass, int order, double max_qsa,
                                    int can_ionize, int* n_qsa, hps_stream stream)
{
    return deposit_current_impl(slab, pl, g, comp, charge, mass, order, max_qsa, can_ionize, n_qsa, stream, -1);
}
extern "C" int hps_deposit_current_laser (hps_slab slab, hps_plasma pl, hps_geom g, const int comp[6], int aabs_comp,
                                          double charge, double mass, int order, double max_qsa,
                                          int can_ionize, int* n_qsa, hps_stream stream)
{
    return deposit_current_impl(slab, pl, g, comp, charge, mass, order, max_qsa, can_ionize, n_qsa, stream, aabs_comp);
}

template <int DT>
static void launch_explicit (int order, dim3 grid, dim3 block, hipStream_t st, SlabView f, hps_plasma pl,
                             const int* ca, const int* de, PartConsts k)
{
    switch (order) {
        case 0: hipLaunchKernelGGL((k_explicit_deposit<0, DT>), grid, block, 0, st, f, pl, ca[0], ca[1], ca[2], ca[3], de[0], de[1], k); break;
        case 1: hipLaunchKernelGGL((k_explicit_deposit<1, DT>), grid, block, 0, st, f, pl, ca[0], ca[1], ca[2], ca[3], de[0], de[1], k); break;
        case 2: hipLaunchKernelGGL((k_explicit_deposit<2, DT>), grid, block, 0, st, f, pl, ca[0], ca[1], ca[2], ca[3], de[0], de[1], k); break;
        default: hipLaunchKernelGGL((k_explicit_deposit<3, DT>), grid, block, 0, st, f, pl, ca[0], ca[1], ca[2], ca[3], de[0], de[1], k); break;
    }
}

static int explicit_deposit_impl (hps_slab slab, hps_plasma pl, hps_geom g, const int cache[4],
                                  const int depos[2], double charge, double mass, int order,
                                  int dtype, int can_ionize, hps_stream stream, int aabs_comp)
{
    HPS_REQUIRE(aabs_comp >= -1 && aabs_comp < slab.ncomp, "hps_explicit_deposit: bad aabs component");
    HPS_REQUIRE(order >= 0 && order <= 3, "hps_explicit_deposit: depos_order must be 0..3");
    if (dtype != 1 && dtype != 2) {
        set_error("hps_explicit_deposit: derivative_type 1 (nodal) or 2 (centred) only");
        return HPS_ERR_UNSUPPORTED;
    }
    if (int e = check_stencil(slab, (order + 1)/2 + 1, "hps_explicit_deposit")) return e;
    for (int c = 0; c < 4; ++c) HPS_REQUIRE(cache[c] >= 0 && cache[c] < slab.ncomp, "hps_explicit_deposit: bad cache component");
    for (int c = 0; c < 2; ++c) HPS_REQUIRE(depos[c] >= 0 && depos[c] < slab.ncomp, "hps_explicit_deposit: bad depos component");
    if (pl.n == 0) return HPS_OK;
    PartConsts k = base_consts(g);
    k.a = charge*invvol_of(g)*g.mu0;
    k.b = charge/mass;
    k.can_ionize = can_ionize;
    k.aabs = aabs_comp; k.laser_fac = (g.m_e/g.q_e)*(g.m_e/g.q_e);        // laser_fac: a0 is always normalised
    const dim3 grid(ceil_div(pl.n, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 2) launch_explicit<2>(order, grid, block, st, SlabView(slab), pl, cache, depos, k);
    else            launch_explicit<1>(order, grid, block, st, SlabView(slab), pl, cache, depos, k);
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}
extern "C" int hps_explicit_deposit (hps_slab slab, hps_plasma pl, hps_geom g, const int cache[4],
                                     const int depos[2], double charge, double mass, int order,
                                     int dtype, int can_ionize, hps_stream stream)
{
    return explicit_deposit_impl(slab, pl, g, cache, depos, charge, mass, order, dtype, can_ionize, stream, -1);
}
extern "C" int hps_explicit_deposit_laser (hps_slab slab, hps_plasma pl, hps_geom g, const int cache[4], int aabs_comp,
                                           const int depos[2], double charge, double mass, int order,
                                           int dtype, int can_ionize, hps_stream stream)
{
    return explicit_deposit_impl(slab, pl, g, cache, depos, charge, mass, order, dtype, can_ionize, stream, aabs_comp);
}

static int advance_plasma_impl (hps_slab slab, hps_plasma pl, hps_geom g, const int comp[5],
                                double charge, double mass, int order, int temp_slice,
                                int n_subcycles, int can_ionize, hps_stream stream, int aabs_comp)
{
    HPS_REQUIRE(aabs_comp >= -1 && aabs_comp < slab.ncomp, "hps_advance_plasma: bad aabs component");
    HPS_REQUIRE(order >= 0 && order <= 3, "hps_advance_plasma: depos_order must be 0..3");
    HPS_REQUIRE(n_subcycles >= 1, "hps_advance_plasma: n_subcycles must be >= 1");
    if (int e = check_stencil(slab, (order + 1)/2 + 1, "hps_advance_plasma")) return e;
    for (int c = 0; c < 5; ++c) HPS_REQUIRE(comp[c] >= 0 && comp[c] < slab.ncomp, "hps_advance_plasma: bad component");
    if (pl.n == 0) return HPS_OK;
    PartConsts k = base_consts(g);
    k.a = charge/(mass*g.c);
    k.dz = g.dz/n_subcycles;
    k.temp_slice = temp_slice; k.n_subcycles = n_subcycles; k.can_ionize = can_ionize;
    k.aabs = aabs_comp; k.laser_fac = (charge/g.q_e)*(g.m_e/mass)*(charge/g.q_e)*(g.m_e/mass);     // laser_norm
    const dim3 grid(ceil_div(pl.n, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    SlabView f(slab);
    switch (order) {
        case 0: hipLaunchKernelGGL(k_advance_plasma<0>, grid, block, 0, st, f, pl, comp[0], comp[1], comp[2], comp[3], comp[4], k); break;
        case 1: hipLaunchKernelGGL(k_advance_plasma<1>, grid, block, 0, st, f, pl, comp[0], comp[1], comp[2], comp[3], comp[4], k); break;
        case 2: hipLaunchKernelGGL(k_advance_plasma<2>, grid, block, 0, st, f, pl, comp[0], comp[1], comp[2], comp[3], comp[4], k); break;
        default: hipLaunchKernelGGL(k_advance_plasma<3>, grid, block, 0, st, f, pl, comp[0], comp[1], comp[2], comp[3], comp[4], k); break;
    }
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}
extern "C" int hps_advance_plasma (hps_slab slab, hps_plasma pl, hps_geom g, const int comp[5],
                                   double charge, double mass, int order, int temp_slice,
                                   int n_subcycles, int can_ionize, hps_stream stream)
{
    return advance_plasma_impl(slab, pl, g, comp, charge, mass, order, temp_slice, n_subcycles, can_ionize, stream, -1);
}
extern "C" int hps_advance_plasma_laser (hps_slab slab, hps_plasma pl, hps_geom g, const int comp[5], int aabs_comp,
                                         double charge, double mass, int order, int temp_slice,
                                         int n_subcycles, int can_ionize, hps_stream stream)
{
    return advance_plasma_impl(slab, pl, g, comp, charge, mass, order, temp_slice, n_subcycles, can_ionize, stream, aabs_comp);
}
