// slab_ops.h -- slab passes that more than one translation unit launches.
#ifndef HPS_SLAB_OPS_H_
#define HPS_SLAB_OPS_H_

#include "common.h"

namespace hps {

struct CellBox { int ilo, ihi, jlo, jhi; };     // padded-array cell range, inclusive

// -grad Psi (Fields.cpp:938-955) and the beam part of Sx, Sy (Hipace::InitializeSxSyWithBeam, Hipace.cpp:744-790) of one
// cell; (i, j) over the whole padded plane: i in [-ng, nx + ng), j likewise
struct GradPsiSxSy { int cPsi, cExmBy, cEypBx; double hdx_inv, hdy_inv; int cSx, cSy, cJzb, cNx, cNy, cPx, cPy;
                     double mu0, dx2, dy2, dz2; CellBox bb; };

__device__ __forceinline__ void gradpsi_sxsy_cell (const SlabView& f, const GradPsiSxSy& a, int i, int j)
{
    if (i >= f.nx + f.ng) return;
    const long o = f.off(i, j);
    const int gg = f.ng - 1;
    if (i >= -gg && i < f.nx + gg && j >= -gg && j < f.ny + gg) {
        const double* P = f.p + a.cPsi*f.ns + o;
        f.p[a.cExmBy*f.ns + o] = -(P[1] - P[-1])*a.hdx_inv;
        f.p[a.cEypBx*f.ns + o] = -(P[f.js] - P[-f.js])*a.hdy_inv;
    }
    // no beam current within reach (bb is in padded-array cells): the sources are 0 without a load
    const int ia = i + f.ng, ja = j + f.ng;
    if (i < 0 || i >= f.nx || j < 0 || j >= f.ny || ia < a.bb.ilo || ia > a.bb.ihi || ja < a.bb.jlo || ja > a.bb.jhi) {
        f.p[a.cSy*f.ns + o] = 0.0; f.p[a.cSx*f.ns + o] = 0.0; return;
    }
    const double* J = f.p + a.cJzb*f.ns + o;
    const double dx_jzb = (J[1] - J[-1])/a.dx2;
    const double dy_jzb = (J[f.js] - J[-f.js])/a.dy2;
    const double dz_jxb = (f.p[a.cPx*f.ns + o] - f.p[a.cNx*f.ns + o])/a.dz2;
    const double dz_jyb = (f.p[a.cPy*f.ns + o] - f.p[a.cNy*f.ns + o])/a.dz2;
    f.p[a.cSy*f.ns + o] =  a.mu0*(-dy_jzb + dz_jyb);
    f.p[a.cSx*f.ns + o] = -a.mu0*(-dx_jzb + dz_jxb);
}

} // namespace hps
#endif
