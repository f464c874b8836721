// ionization.hip -- ADK field ionisation of the plasma species "ion" (SURVEY 8f-2):
// PlasmaParticleContainer::InitIonizationModule (particles/plasma/PlasmaParticleContainerInit.cpp:382-464) and
// PlasmaParticleContainer::IonizationModule (particles/plasma/PlasmaParticleContainer.cpp:261-440), called once per slice
// after the field solves and before the plasma push (Hipace.cpp:693-696).
//
// The reference runs two passes (decide + mask with an atomic count, then a second kernel that hands out electron
// slots with another atomic per new electron, with a stream synchronisation in between to resize the product
// species).  Here it is ONE kernel: every wave decides for its 64 ions, takes its block of electron slots with one
// atomic per wave (ballot + popcount), and writes the electrons straight behind the product species' sheet, whose
// arrays are allocated once for the most electrons the ions can release.  The new count reaches the host through
// mapped host memory (posted by the last workgroup to finish), which the host polls after it has enqueued the push
// of the ions -- no stream synchronisation.
//
// Random numbers: one uniform deviate per (ion, slice, time step) from a counter-based generator keyed by the ion's
// lattice index, which travels in the id bits of idcpu (the reference only ever reads the sign of that field), so
// that the draw does not depend on the order the tile sort has put the ions in.  Same integer arithmetic as the
// oracle's ion_uniform.
#include "common.h"
#include "particle_math.h"
#include "engine.h"

#include <cmath>
#include <cstdlib>

namespace hps {

// per-particle form (tile_size = 0, or a sheet without a tiling); the LDS-tile form is fused into the ions' push
// (particles_tiled.hip: k_advance_tiled<.., IONIZE>), which gathers the same fields anyway
template <int ORDER>
__global__ __launch_bounds__(256)
void k_ionize (SlabView f, hps_plasma ion, int cPsi, int cEz, int cBx, int cBy, PartConsts k, IonArgs a)
{
    constexpr int NS = ORDER + 2;
    const long ip = (long)blockIdx.x*blockDim.x + threadIdx.x;
    bool ionize = false;
    if (ip < ion.n) {
        const uint64_t id = ion.idcpu[ip];
        const int lev = ion.ion_lev[ip];
        // an ion that has lost all Z electrons cannot ionise (the reference reads past the end of its tables there)
        if ((id & HPS_ID_VALID) && lev < a.Z) {
            // doGatherShapeN at (x_prev, y_prev) (PlasmaParticleContainer.cpp:341-350)
            const double xp = ion.x_prev[ip], yp = ion.y_prev[ip];
            double sx[NS], dsx[NS], sy[NS], dsy[NS];
            const int i0 = nodal_weights<ORDER>((xp - k.xoff)*k.dx_inv, sx, dsx);
            const int j0 = nodal_weights<ORDER>((yp - k.yoff)*k.dy_inv, sy, dsy);
            double ExmBy = 0.0, EypBx = 0.0, Ez = 0.0, Bx = 0.0, By = 0.0;
#pragma unroll
            for (int iy = 0; iy < NS; ++iy) {
                const long row = f.off(i0, j0 + iy);
#pragma unroll
                for (int ix = 0; ix < NS; ++ix) {
                    const double* p = f.p + row + ix;
                    const double psi_c = p[cPsi*f.ns];
                    const double ss = sx[ix]*sy[iy];
                    ExmBy += (dsx[ix]*sy[iy])*psi_c*k.dx_inv;
                    EypBx += (sx[ix]*dsy[iy])*psi_c*k.dy_inv;
                    Ez += ss*p[cEz*f.ns];
                    Bx += ss*p[cBx*f.ns];
                    By += ss*p[cBy*f.ns];
                }
            }
            ionize = adk_decide(a, ExmBy + By*k.c, EypBx - Bx*k.c, Ez, ion.ux_half[ip], ion.uy_half[ip], ion.psi_half[ip], lev, id);
            if (ionize) ion.ion_lev[ip] = lev + 1;
        }
    }
    const long q = ip < ion.n ? ip : 0;
    adk_emit(a, ionize, ion.x[q], ion.y[q], ion.x_prev[q], ion.y_prev[q], ion.w[q]);
    adk_post(a);
}

// Block maxima for adk_tile_below_threshold: one workgroup per tile of the ion tiling scans the tile's cells (the edge
// tiles take the guard cells with them) -- 4 planes read once, 32 MB at 1024^2.
__global__ __launch_bounds__(256)
void k_ion_field_bounds (SlabView f, int cPsi, int cEz, int cBx, int cBy, int ts, int ntx, int nty, double* out)
{
    const int tx = blockIdx.x % ntx, ty = blockIdx.x / ntx;
    const int i0 = tx == 0 ? -f.ng : tx*ts, i1 = tx == ntx - 1 ? f.nx + f.ng : (tx + 1)*ts;
    const int j0 = ty == 0 ? -f.ng : ty*ts, j1 = ty == nty - 1 ? f.ny + f.ng : (ty + 1)*ts;
    const int w = i1 - i0, n = w*(j1 - j0);
    double m[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int s = threadIdx.x; s < n; s += 256) {
        const int lj = s / w, i = i0 + (s - lj*w), j = j0 + lj;
        const long o = f.off(i, j);
        const double psi = f.p[cPsi*f.ns + o];
        if (i + 1 < f.nx + f.ng) m[0] = fmax(m[0], fabs(f.p[cPsi*f.ns + o + 1] - psi));
        if (j + 1 < f.ny + f.ng) m[1] = fmax(m[1], fabs(f.p[cPsi*f.ns + o + f.js] - psi));
        m[2] = fmax(m[2], fabs(f.p[cBx*f.ns + o]));
        m[3] = fmax(m[3], fabs(f.p[cBy*f.ns + o]));
        m[4] = fmax(m[4], fabs(f.p[cEz*f.ns + o]));
    }
    __shared__ double red[4][5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        double v = m[q];
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const int q = threadIdx.x;
        out[(long)q*ntx*nty + blockIdx.x] = fmax(fmax(red[0][q], red[1][q]), fmax(red[2][q], red[3][q]));
    }
}

int Engine::ion_field_bounds ()
{
    if (!ion.tiling || !ion.d_fbound) return HPS_OK;
    const TileGeom& g = ion.tiling->g;
    const int cP = pc ? (int)HPS_PC_PSI : (int)HPS_C_PSI, cE = pc ? (int)HPS_PC_EZ : (int)HPS_C_EZ, cX = pc ? (int)HPS_PC_BX : (int)HPS_C_BX, cY = pc ? (int)HPS_PC_BY : (int)HPS_C_BY;
    hipLaunchKernelGGL(k_ion_field_bounds, dim3(g.ntiles), dim3(256), 0, st, SlabView(slab), cP, cE, cX, cY, g.ts, g.ntx, g.nty,
                       ion.d_fbound);
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

// InitIonizationModule: ADK prefactors (Chen et al., JCP 236 (2013), eq. (2); l = m = 0, the approximate expressions
// without the Gamma function of the angular part); adk = [prefactor[Z] | exp_prefactor[Z] | power[Z]] on the device
int ion_create (Engine& E)
{
    const hps_deck& d = E.d;
    HPS_REQUIRE(d.ion_Z >= 1 && d.ion_Z <= HPS_MAX_ION_LEVELS, "hps_engine_create: ion_Z out of range");
    HPS_REQUIRE(d.ion_init_level >= 0 && d.ion_init_level <= d.ion_Z, "hps_engine_create: the initial ion level must be specified (0 .. Z)");
    for (int i = 0; i < d.ion_Z; ++i) HPS_REQUIRE(d.ion_energies[i] > 0.0, "hps_engine_create: ionisation energies must be positive");
    HPS_REQUIRE(d.si_units || d.background_density_SI > 0.0,
                "hps_engine_create: ionisation in normalised units needs hipace.background_density_SI");      // (:391-396)
    HPS_REQUIRE(std::fabs(d.plasma_charge/d.ion_charge + 1.0) < 1e-3, "hps_engine_create: ion and ionisation product charges have to be opposite");   // (:404-405)
    const double cSI = 299792458.0, qeSI = 1.602176634e-19, meSI = 9.1093837015e-31, ep0SI = 8.8541878128e-12;
    const double alpha = 0.0072973525693, r_e = 2.8179403227e-15;
    const double a3 = alpha*alpha*alpha, a4 = a3*alpha;
    const double wa = a3*cSI/r_e;
    const double Ea = meSI*cSI*cSI/qeSI*a4/r_e;
    const double UH = 13.59843449;                    // ionisation energy of hydrogen, entry 0 of the reference's table
    const double l_eff = std::sqrt(UH/d.ion_energies[0]) - 1.0;
    const double wp = std::sqrt(d.background_density_SI*qeSI*qeSI/(ep0SI*meSI));
    const double dt = d.si_units ? E.gm.dz/cSI : E.gm.dz/wp;
    std::vector<double> adk((size_t)4*d.ion_Z);
    for (int i = 0; i < d.ion_Z; ++i) {
        const double Uion = d.ion_energies[i];
        const double n_eff = (i + 1)*std::sqrt(UH/Uion);
        const double C2 = std::pow(2, 2*n_eff)/(n_eff*std::tgamma(n_eff + l_eff + 1)*std::tgamma(n_eff - l_eff));
        adk[i] = dt*wa*C2*(Uion/(2*UH))*std::pow(2*std::pow((Uion/UH), 3./2)*Ea, 2*n_eff - 1);
        adk[d.ion_Z + i] = -2./3*std::pow(Uion/UH, 3./2)*Ea;
        adk[2*d.ion_Z + i] = -(2*n_eff - 1);
        // field below which w = prefactor E^power exp(exp_prefactor / E) < 1e-20 (w grows with E there): bisection in ln E
        auto lnw = [&] (double E) { return std::log(adk[i]) + adk[2*d.ion_Z + i]*std::log(E) + adk[d.ion_Z + i]/E; };
        double lo = 1.0, hi = -adk[d.ion_Z + i];          // ln w(hi) = ln(prefactor) + power ln(hi) - 1: the far side of the bend
        adk[3*d.ion_Z + i] = 0.0;
        if (lnw(lo) < std::log(1e-20) && lnw(hi) > std::log(1e-20)) {
            for (int it = 0; it < 200; ++it) { const double mid = std::sqrt(lo*hi); (lnw(mid) < std::log(1e-20) ? lo : hi) = mid; }
            adk[3*d.ion_Z + i] = lo;
        }
    }
    HPS_HIP_CHECK(hipMalloc(&E.ion.d_adk, adk.size()*sizeof(double)));
    HPS_HIP_CHECK(hipMemcpy(E.ion.d_adk, adk.data(), adk.size()*sizeof(double), hipMemcpyHostToDevice));
    HPS_HIP_CHECK(hipMalloc(&E.ion.d_cnt, 4*sizeof(unsigned long long)));
    HPS_HIP_CHECK(hipMemset(E.ion.d_cnt, 0, 4*sizeof(unsigned long long)));
    HPS_HIP_CHECK(hipHostMalloc(&E.ion.h_cnt, 4*sizeof(long long), hipHostMallocMapped));
    for (int i = 0; i < 4; ++i) E.ion.h_cnt[i] = 0;
    HPS_HIP_CHECK(hipHostGetDevicePointer((void**)&E.ion.h_cnt_dev, E.ion.h_cnt, 0));
    return HPS_OK;
}

void ion_destroy (Engine& E)
{
    (void)hipFree(E.ion.d_adk); (void)hipFree(E.ion.d_cnt);
    if (E.ion.h_cnt) (void)hipHostFree(E.ion.h_cnt);
    (void)hipFree(E.ion.real); (void)hipFree(E.ion.pl.idcpu); (void)hipFree(E.ion.pl.ion_lev);
    (void)hipFree(E.ion.real_alt); (void)hipFree(E.ion.pl_alt.idcpu); (void)hipFree(E.ion.pl_alt.ion_lev);
    delete E.ion.tiling; (void)hipFree(E.ion.d_tile_flag); (void)hipFree(E.ion.d_fbound);
}

IonArgs Engine::ion_args (int islice)
{
    IonArgs a{};
    const double cSI = 299792458.0, qeSI = 1.602176634e-19, meSI = 9.1093837015e-31, ep0SI = 8.8541878128e-12;
    const double wp = std::sqrt(d.background_density_SI*qeSI*qeSI/(ep0SI*meSI));
    a.el = pl; a.adk = ion.d_adk; a.cnt = ion.d_cnt; a.host = (volatile long long*)ion.h_cnt_dev;
    a.E0 = d.si_units ? 1.0 : wp*meSI*cSI/qeSI;
    a.clightsq_inv = 1.0/(gm.c*gm.c);
    a.Z = d.ion_Z; a.seed = d.ion_seed; a.step = (unsigned long long)step_index; a.islice = (unsigned long long)islice;
    a.cap = np_cap; a.tile_flag = ion.d_tile_flag;
    a.product_init_lev = 0;       // the product species is the first (electron) species: not ionisable here, its particles carry level 0
    a.fbound = (ion.tiling && ion.d_fbound) ? ion.d_fbound : nullptr;
    if (ion.tiling) { a.fb_ntx = ion.tiling->g.ntx; a.fb_nty = ion.tiling->g.nty; }
    a.fb_dx_inv = 1.0/gm.dx; a.fb_dy_inv = 1.0/gm.dy; a.fb_c = gm.c;
    a.seq = ++ion.seq;
    ion.pending = true;
    return a;
}

int Engine::ionize_slice (int islice)
{
    if (!d.ion_on || ion.n == 0) return HPS_OK;
    const IonArgs a = ion_args(islice);
    const PartConsts k = base_consts(gm);
    const SlabView f(slab);
    const dim3 grid(ceil_div(ion.n, 256)), block(256);
    const int cP = pc ? (int)HPS_PC_PSI : (int)HPS_C_PSI, cE = pc ? (int)HPS_PC_EZ : (int)HPS_C_EZ, cX = pc ? (int)HPS_PC_BX : (int)HPS_C_BX, cY = pc ? (int)HPS_PC_BY : (int)HPS_C_BY;
#define CALL(O) hipLaunchKernelGGL(k_ionize<O>, grid, block, 0, st, f, ion.pl, cP, cE, cX, cY, k, a)
    switch (d.order) { case 0: CALL(0); break; case 1: CALL(1); break; case 2: CALL(2); break; default: CALL(3); break; }
#undef CALL
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

// wait for the post of the last k_ionize; fall back to the stream's status every so often so that a failed launch
// cannot hang the host
int Engine::ionize_collect ()
{
    if (!ion.pending) return HPS_OK;
    ion.pending = false;
    volatile long long* hp = ion.h_cnt;
    long spins = 0;
    while (hp[3] != ion.seq) {
        if ((++spins & 0xfffff) == 0 && hipStreamQuery(st) != hipErrorNotReady) {
            if (hp[3] == ion.seq) break;
            HPS_HIP_CHECK(hipStreamSynchronize(st));
            if (hp[3] != ion.seq) { set_error("ionisation: the electron count never arrived"); return HPS_ERR_HIP; }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (hp[1] != 0) { set_error("ionisation: the product species' arrays are full"); return HPS_ERR_ARG; }
    np = (long)hp[0]; pl.n = np; pl_alt.n = np;
    ion.n_ionized = (long)hp[2];
    return HPS_OK;
}

} // namespace hps
