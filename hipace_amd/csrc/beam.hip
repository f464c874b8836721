// beam.hip -- driver-beam slice operators for hipace.dt != 0: the beam slice push
// (AdvanceBeamParticlesSlice, particles/pusher/BeamParticleAdvance.cpp:20-336), the hand-off of
// slipped particles to the next slice (shiftSlippedParticles, particles/sorting/SliceSort.cpp:12-64)
// and the deposit over the part of a slice that has not slipped in during this step
// (particles/deposition/BeamDepositCurrent.cpp:100).
//
// Layout: one global SoA over all beam particles, head slice first.  Slice p (counted from the head)
// is the index range [B[p], B[p+1]) of DEVICE-resident boundaries: a particle that leaves slice p
// (z < lower end of the slice) is moved to the end of its range by the partition kernel, and B[p+1]
// is lowered past it -- it has become the front of slice p+1 without being copied anywhere, and the
// host never has to learn the counts inside a step (nfront[p+1] = how many such particles lead
// slice p+1: they are pushed there, to finish their sub-cycles, but not deposited).
#include "common.h"
#include "particle_math.h"
#include "engine.h"

namespace hps {

struct BeamPushConsts {
    double dt;                 // per sub-cycle
    double c, inv_c2, qm, min_z;
    double ex_slope, ey_slope; // beams.external_E = (ex_slope*x, ey_slope*y, 0)
    int nsc;
    PartConsts pc;             // geometry, boundary
    // radiation reaction (BeamParticleAdvance.cpp:101-113, 244-297): rr != 0 switches it on
    int rr, normalized, no_z_push;
    double RRcoeff, E0, wp_inv, c_SI;
    // spin tracking (:218-238): spin != 0 switches it on
    int spin; double spin_anom;
};

template <int ORDER>
__global__ __launch_bounds__(256)
void k_beam_deposit_dyn (SlabView f, BeamSoA b, const long* __restrict__ B, const int* __restrict__ nfront, int p,
                         int cjx, int cjy, int cjz, double q_invvol, double clightsq_inv, PartConsts k, int* disturbed)
{
    const long first = B[p] + nfront[p], count = B[p + 1] - first;
    for (long t = (long)blockIdx.x*blockDim.x + threadIdx.x; t < count; t += (long)gridDim.x*blockDim.x) {
    const long ip = first + t;
    if (b.nsub[ip] < 0) continue;               // absorbed at the boundary
    const double ux = b.ux[ip], uy = b.uy[ip], uz = b.uz[ip];
    const double gaminv = 1.0/sqrt(1.0 + ux*ux*clightsq_inv + uy*uy*clightsq_inv + uz*uz*clightsq_inv);
    const double wq = q_invvol*b.w[ip];
    // (predictor-corrector loop: Engine::d_pc_dist, as k_beam_deposit)
    if (disturbed && ((cjx >= 0 && (wq*(ux*gaminv) != 0.0 || wq*(uy*gaminv) != 0.0)) || (cjz >= 0 && wq*(uz*gaminv) != 0.0)))
        __hip_atomic_store(disturbed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double sx[ORDER + 1], sy[ORDER + 1];
    const int i0 = shape_weights<ORDER>((b.x[ip] - k.xoff)*k.dx_inv, sx);
    const int j0 = shape_weights<ORDER>((b.y[ip] - k.yoff)*k.dy_inv, sy);
#pragma unroll
    for (int iy = 0; iy <= ORDER; ++iy) {
#pragma unroll
        for (int ix = 0; ix <= ORDER; ++ix) {
            double* q = f.p + f.off(i0 + ix, j0 + iy);
            const double s = sx[ix]*sy[iy];
            if (cjx >= 0) { atomic_add_f64(q + cjx*f.ns, s*(wq*(ux*gaminv))); atomic_add_f64(q + cjy*f.ns, s*(wq*(uy*gaminv))); }
            if (cjz >= 0) atomic_add_f64(q + cjz*f.ns, s*(wq*(uz*gaminv)));
        }
    }
    }
}

template <int ORDER>
__global__ __launch_bounds__(256)
void k_beam_push (SlabView f, BeamSoA b, const long* __restrict__ B, int p, int cPsi, int cEz, int cBx, int cBy, int cBz,
                  BeamPushConsts k)
{
    constexpr int NS = ORDER + 2;
    const long first = B[p], count = B[p + 1] - first;       // slipped-in particles included (:131)
    for (long t = (long)blockIdx.x*blockDim.x + threadIdx.x; t < count; t += (long)gridDim.x*blockDim.x) {
    const long ip = first + t;
    int i = b.nsub[ip];
    if (i < 0) continue;
    double xp = b.x[ip], yp = b.y[ip], zp = b.z[ip], ux = b.ux[ip], uy = b.uy[ip], uz = b.uz[ip];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (k.spin) { s0 = b.sx[ip]; s1 = b.sy[ip]; s2 = b.sz[ip]; }
    bool absorbed = false;
    for (; i < k.nsc; ++i) {
        if (zp < k.min_z) break;                              // not on this slice any more (:150-153)
        const double gi = 1.0/sqrt(1.0 + (ux*ux + uy*uy + uz*uz)*k.inv_c2);
        xp += k.dt*0.5*ux*gi;
        yp += k.dt*0.5*uy*gi;
        if (apply_particle_bc(k.pc, xp, yp, ux, uy)) { absorbed = true; break; }
        // doGatherShapeN (particles/particles_utils/FieldGather.H:45-96)
        double sx[NS], dsx[NS], sy[NS], dsy[NS];
        const int i0 = nodal_weights<ORDER>((xp - k.pc.xoff)*k.pc.dx_inv, sx, dsx);
        const int j0 = nodal_weights<ORDER>((yp - k.pc.yoff)*k.pc.dy_inv, sy, dsy);
        double ExmBy = 0.0, EypBx = 0.0, Ez = 0.0, Bx = 0.0, By = 0.0, Bz = 0.0;
#pragma unroll
        for (int iy = 0; iy < NS; ++iy) {
#pragma unroll
            for (int ix = 0; ix < NS; ++ix) {
                const double* q = f.p + f.off(i0 + ix, j0 + iy);
                const double psi_c = q[cPsi*f.ns];
                const double ss = sx[ix]*sy[iy];
                ExmBy += (dsx[ix]*sy[iy])*psi_c*k.pc.dx_inv;
                EypBx += (sx[ix]*dsy[iy])*psi_c*k.pc.dy_inv;
                Ez += ss*q[cEz*f.ns];
                Bx += ss*q[cBx*f.ns];
                By += ss*q[cBy*f.ns];
                Bz += ss*q[cBz*f.ns];
            }
        }
        // ApplyExternalField (particles/pusher/ExternalFields.H:29-56), E = (ex_slope x, ey_slope y, 0), B = 0
        ExmBy += k.ex_slope*xp;
        EypBx += k.ey_slope*yp;
        double ux_next = ux + k.dt*k.qm*(ExmBy + (k.c - uz*gi)*By + uy*gi*Bz);
        double uy_next = uy + k.dt*k.qm*(EypBx + (uz*gi - k.c)*Bx - ux*gi*Bz);
        const double ux_i = (ux_next + ux)*0.5, uy_i = (uy_next + uy)*0.5;
        const double uz_i = uz + k.dt*0.5*k.qm*Ez;
        const double gii = 1.0/sqrt(1.0 + (ux_i*ux_i + uy_i*uy_i + uz_i*uz_i)*k.inv_c2);
        if (k.spin) {      // Thomas-BMT precession as a Boris-type rotation (:218-238)
            const double ic = 1.0/k.c;
            const double E0_ = ExmBy + k.c*By, E1_ = EypBx - k.c*Bx, E2_ = Ez;
            const double u0 = ux_i*ic, u1 = uy_i*ic, u2 = uz_i*ic;
            const double be0 = u0*gii, be1 = u1*gii, be2 = u2*gii;
            const double gp1 = gii/(1.0 + gii);
            const double x0 = be1*E2_ - be2*E1_, x1 = be2*E0_ - be0*E2_, x2 = be0*E1_ - be1*E0_;      // beta x E
            const double bdB = be0*Bx + be1*By + be2*Bz;
            const double aq = fabs(k.qm);
            const double h0 = aq*(Bx*gii - x0*ic*gp1 + k.spin_anom*(Bx - gp1*u0*bdB - x0*ic))*k.dt*0.5;
            const double h1 = aq*(By*gii - x1*ic*gp1 + k.spin_anom*(By - gp1*u1*bdB - x1*ic))*k.dt*0.5;
            const double h2 = aq*(Bz*gii - x2*ic*gp1 + k.spin_anom*(Bz - gp1*u2*bdB - x2*ic))*k.dt*0.5;
            const double p0 = s0 + (h1*s2 - h2*s1), p1 = s1 + (h2*s0 - h0*s2), p2 = s2 + (h0*s1 - h1*s0);   // s' = s + h x s
            const double o = 1.0/(1.0 + (h0*h0 + h1*h1 + h2*h2));
            const double hd = h0*p0 + h1*p1 + h2*p2;
            s0 = o*(p0 + (hd*h0 + (h1*p2 - h2*p1)));
            s1 = o*(p1 + (hd*h1 + (h2*p0 - h0*p2)));
            s2 = o*(p2 + (hd*h2 + (h0*p1 - h1*p0)));
        }
        double uz_next = uz + k.dt*k.qm*(Ez + (ux_i*By - uy_i*Bx)*gii);
        if (k.rr) {      // classical radiation reaction in SI quantities (:244-297)
            const double icSI = 1.0/k.c_SI, ic = 1.0/k.c;
            double Ex = ExmBy + k.c*By, Ey = EypBx - k.c*Bx, Ezs = Ez, Bxs = Bx, Bys = By, Bzs = Bz;
            if (k.normalized) { Ex *= k.E0; Ey *= k.E0; Ezs *= k.E0; Bxs *= k.E0*icSI; Bys *= k.E0*icSI; Bzs *= k.E0*icSI; }
            const double gam = sqrt(1.0 + (ux_i*ux_i + uy_i*uy_i + uz_i*uz_i)*k.inv_c2);
            const double vx = ux_i*gii*k.c_SI*ic, vy = uy_i*gii*k.c_SI*ic, vz = uz_i*gii*k.c_SI*ic;
            const double bx = vx*icSI, by = vy*icSI, bz = vz*icSI;
            const double flx = (Ex + vy*Bzs - vz*Bys), fly = (Ey + vz*Bxs - vx*Bzs), flz = (Ezs + vx*Bys - vy*Bxs);
            const double fl2 = flx*flx + fly*fly + flz*flz;
            const double bE = (bx*Ex + by*Ey + bz*Ezs);
            const double coeff = gam*gam*(fl2 - bE*bE);
            const double frx = k.RRcoeff*(k.c_SI*(fly*Bzs - flz*Bys) + bE*Ex - coeff*bx);
            const double fry = k.RRcoeff*(k.c_SI*(flz*Bxs - flx*Bzs) + bE*Ey - coeff*by);
            const double frz = k.RRcoeff*(k.c_SI*(flx*Bys - fly*Bxs) + bE*Ezs - coeff*bz);
            const double sc = k.dt*k.wp_inv*k.c*icSI;
            ux_next += frx*sc; uy_next += fry*sc; uz_next += frz*sc;
        }
        const double gni = 1.0/sqrt(1.0 + (ux_next*ux_next + uy_next*uy_next + uz_next*uz_next)*k.inv_c2);
        xp += k.dt*0.5*ux_next*gni;
        yp += k.dt*0.5*uy_next*gni;
        if (!k.no_z_push) zp += k.dt*(uz_next*gni - k.c);     // do_z_push (:316)
        ux = ux_next; uy = uy_next; uz = uz_next;
    }
    if (absorbed || apply_particle_bc(k.pc, xp, yp, ux, uy)) { b.w[ip] = 0.0; b.nsub[ip] = -1; continue; }
    b.x[ip] = xp; b.y[ip] = yp; b.z[ip] = zp; b.nsub[ip] = i;
    b.ux[ip] = ux; b.uy[ip] = uy; b.uz[ip] = uz;
    if (k.spin) { b.sx[ip] = s0; b.sy[ip] = s1; b.sz[ip] = s2; }
    }
}

// One workgroup: particles of slice p with z < min_z go to the end of the slice's range, the boundary to
// slice p+1 is lowered past them.  Nothing moves when nothing slipped (the usual case).
__global__ __launch_bounds__(1024)
void k_beam_partition (BeamSoA b, BeamSoA scr, long* B, int* nfront, int p, double min_z)
{
    __shared__ int s_cnt[2];
    __shared__ int s_red[16];
    const long first = B[p], count = B[p + 1] - first;
    const int t = threadIdx.x;
    int mine = 0;
    for (long q = t; q < count; q += 1024) mine += (b.z[first + q] < min_z) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((t & 63) == 0) s_red[t >> 6] = mine;
    if (t < 2) s_cnt[t] = 0;
    __syncthreads();
    int nslip = 0;
    for (int w = 0; w < 16; ++w) nslip += s_red[w];
    if (nslip == 0) { if (t == 0) nfront[p + 1] = 0; return; }
    for (long q = t; q < count; q += 1024) {
        const long ip = first + q;
        const bool slip = b.z[ip] < min_z;
        const long dst = slip ? count - 1 - atomicAdd(&s_cnt[1], 1) : atomicAdd(&s_cnt[0], 1);
        scr.x[dst] = b.x[ip]; scr.y[dst] = b.y[ip]; scr.z[dst] = b.z[ip]; scr.ux[dst] = b.ux[ip];
        scr.uy[dst] = b.uy[ip]; scr.uz[dst] = b.uz[ip]; scr.w[dst] = b.w[ip]; scr.nsub[dst] = b.nsub[ip];
        if (b.sx) { scr.sx[dst] = b.sx[ip]; scr.sy[dst] = b.sy[ip]; scr.sz[dst] = b.sz[ip]; }
    }
    __threadfence_block();
    __syncthreads();
    for (long q = t; q < count; q += 1024) {
        const long ip = first + q;
        b.x[ip] = scr.x[q]; b.y[ip] = scr.y[q]; b.z[ip] = scr.z[q]; b.ux[ip] = scr.ux[q];
        b.uy[ip] = scr.uy[q]; b.uz[ip] = scr.uz[q]; b.w[ip] = scr.w[q]; b.nsub[ip] = scr.nsub[q];
        if (b.sx) { b.sx[ip] = scr.sx[q]; b.sy[ip] = scr.sy[q]; b.sz[ip] = scr.sz[q]; }
    }
    if (t == 0) { B[p + 1] = first + count - nslip; nfront[p + 1] = nslip; }
}

// ---- ring hand-off (MultiBuffer::put_data / get_data, utils/MultiBuffer.cpp:444-609) ----------------------------
// message = [count | x[cap] y[cap] z[cap] ux[cap] uy[cap] uz[cap] w[cap]]; everything stays on the device
__global__ __launch_bounds__(256)
void k_beam_export (BeamSoA b, const long* __restrict__ B, int p, double* __restrict__ msg, long cap, int* overflow)
{
    const long first = B[p], count = B[p + 1] - first;
    if (blockIdx.x == 0 && threadIdx.x == 0) { msg[0] = (double)min(count, cap); if (count > cap) atomicAdd(overflow, 1); }
    const double* a[10] = {b.x, b.y, b.z, b.ux, b.uy, b.uz, b.w, b.sx, b.sy, b.sz};
    const int rows = b.sx ? 10 : 7;
    for (long q = (long)blockIdx.x*blockDim.x + threadIdx.x; q < min(count, cap); q += (long)gridDim.x*blockDim.x)
        for (int k = 0; k < rows; ++k) msg[1 + k*cap + q] = a[k][first + q];
}

// block p of the coming step: placed behind the blocks imported so far (imp[p] = where it starts); every later
// boundary moves with it, so the slices not imported yet are empty ranges
__global__ __launch_bounds__(256)
void k_beam_import (BeamSoA b, long* B, long* imp, int p, int nz, const double* __restrict__ msg, long cap, long capacity, int* overflow)
{
    const long start = imp[p];
    long count = (long)msg[0];
    if (count > cap) count = cap;
    if (start + count > capacity) { count = capacity - start; if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(overflow, 1); }
    double* a[10] = {b.x, b.y, b.z, b.ux, b.uy, b.uz, b.w, b.sx, b.sy, b.sz};
    const int rows = b.sx ? 10 : 7;
    for (long q = (long)blockIdx.x*blockDim.x + threadIdx.x; q < count; q += (long)gridDim.x*blockDim.x) {
        for (int k = 0; k < rows; ++k) a[k][start + q] = msg[1 + k*cap + q];
        b.nsub[start + q] = 0;
    }
}
__global__ void k_beam_import_bounds (long* B, long* imp, int p, int nz, const double* __restrict__ msg, long cap, long capacity)
{
    const long start = imp[p];
    long count = (long)msg[0];
    if (count > cap) count = cap;
    if (start + count > capacity) count = capacity - start;
    if (threadIdx.x == 0) imp[p + 1] = start + count;
    for (int q = p + 1 + threadIdx.x; q <= nz; q += blockDim.x) B[q] = start + count;
}

int beam_export_slice (Engine& E, int islice, double* msg_dev, long cap)
{
    const int p = E.d.nz - 1 - islice;
    hipLaunchKernelGGL(k_beam_export, dim3((unsigned)std::max<long>(1, std::min<long>(ceil_div(cap, 256), 256))), dim3(256), 0, E.st,
                       E.bm, E.d_B, p, msg_dev, cap, E.d_beam_overflow);
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

int beam_import_slice (Engine& E, int islice, const double* msg_dev, long cap)
{
    const int p = E.d.nz - 1 - islice;
    const long capacity = std::max(E.nbeam, 1L);
    hipLaunchKernelGGL(k_beam_import, dim3((unsigned)std::max<long>(1, std::min<long>(ceil_div(cap, 256), 256))), dim3(256), 0, E.st,
                       E.bm, E.d_B, E.d_Bimp, p, E.d.nz, msg_dev, cap, capacity, E.d_beam_overflow);
    hipLaunchKernelGGL(k_beam_import_bounds, dim3(1), dim3(256), 0, E.st, E.d_B, E.d_Bimp, p, E.d.nz, msg_dev, cap, capacity);
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

static BeamPushConsts push_consts (const Engine& E, int islice)
{
    BeamPushConsts k{};
    const hps_deck& d = E.d;
    k.nsc = d.beam_n_subcycles > 0 ? d.beam_n_subcycles : 10;
    k.dt = d.dt/k.nsc;
    k.c = E.gm.c; k.inv_c2 = 1.0/(E.gm.c*E.gm.c);
    k.qm = d.beam_charge/(d.beam_mass != 0.0 ? d.beam_mass : 1.0);
    k.min_z = d.lo[2] + islice*E.gm.dz;
    k.ex_slope = d.ext_E_slope[0]; k.ey_slope = d.ext_E_slope[1];
    k.pc = base_consts(E.gm);
    // radiation reaction constants (:101-113), PhysConstSI of utils/Constants.H:15-24
    const double cSI = 299792458.0, qeSI = 1.602176634e-19, meSI = 9.1093837015e-31, ep0SI = 8.8541878128e-12, reSI = 2.817940326204929e-15;
    k.rr = d.beam_radiation_reaction; k.normalized = d.si_units ? 0 : 1; k.no_z_push = d.beam_no_z_push; k.c_SI = cSI;
    const double q_over_mc = k.normalized ? k.qm/cSI*qeSI/meSI : k.qm/cSI;
    k.RRcoeff = (2.0/3.0)*reSI*q_over_mc*q_over_mc;
    k.wp_inv = (k.normalized && d.background_density_SI > 0.0) ? std::sqrt(ep0SI*meSI/(d.background_density_SI*qeSI*qeSI)) : 1.0;
    k.E0 = k.normalized ? meSI*cSI/k.wp_inv/qeSI : 1.0;
    k.spin = d.beam_spin_tracking; k.spin_anom = d.beam_spin_anom;      // (0 = pure Thomas precession; hosts that want the electron pass 0.00115965218128, as decks.py does)
    return k;
}

#define HPS_BEAM_ORDER(order, CALL) switch (order) { case 0: { CALL(0); } break; case 1: { CALL(1); } break; \
                                                     case 2: { CALL(2); } break; default: { CALL(3); } break; }

int beam_deposit_moving (Engine& E, int p, int cjx, int cjy, int cjz)
{
    if (p < 0 || p >= E.d.nz) return HPS_OK;
    const long bound = E.beam_bound(p);
    if (bound <= 0) return HPS_OK;
    const SlabView f(E.slab);
    const PartConsts k = base_consts(E.gm);
    const double q_invvol = E.d.beam_charge*(E.d.si_units ? 1.0/(E.gm.dx*E.gm.dy*E.gm.dz) : 1.0), csq_inv = 1.0/(E.gm.c*E.gm.c);
    const dim3 grid((unsigned)std::min<long>(ceil_div(bound, 256), 2048)), block(256);
#define CALL(O) hipLaunchKernelGGL(k_beam_deposit_dyn<O>, grid, block, 0, E.st, f, E.bm, E.d_B, E.d_nfront, p, cjx, cjy, cjz, q_invvol, csq_inv, k, E.d_pc_dist)
    HPS_BEAM_ORDER(E.d.order, CALL)
#undef CALL
    return HPS_OK;
}

int beam_push_moving (Engine& E, int islice)
{
    const int p = E.d.nz - 1 - islice;
    const long bound = E.beam_bound(p);
    if (bound <= 0) return HPS_OK;
    const SlabView f(E.slab);
    const BeamPushConsts k = push_consts(E, islice);
    const dim3 grid((unsigned)std::min<long>(ceil_div(bound, 256), 2048)), block(256);
    // the fields of This slice sit at other components of the predictor-corrector's slab (fields/Fields.cpp:128-164)
    const int cP = E.pc ? (int)HPS_PC_PSI : (int)HPS_C_PSI, cE = E.pc ? (int)HPS_PC_EZ : (int)HPS_C_EZ, cX = E.pc ? (int)HPS_PC_BX : (int)HPS_C_BX,
              cY = E.pc ? (int)HPS_PC_BY : (int)HPS_C_BY, cZ = E.pc ? (int)HPS_PC_BZ : (int)HPS_C_BZ;
#define CALL(O) hipLaunchKernelGGL(k_beam_push<O>, grid, block, 0, E.st, f, E.bm, E.d_B, p, cP, cE, cX, cY, cZ, k)
    HPS_BEAM_ORDER(E.d.order, CALL)
#undef CALL
    hipLaunchKernelGGL(k_beam_partition, dim3(1), dim3(1024), 0, E.st, E.bm, E.bm_scr, E.d_B, E.d_nfront, p, k.min_z);
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

} // namespace hps
