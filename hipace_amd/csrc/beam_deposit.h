// beam_deposit.h -- the static beam's two deposits of a slice (jz of this slice's block, jx / jy of the next one's:
// BeamDepositCurrent.cpp:25-140, Hipace.cpp:613-614, 656-657) as a device function: the kernel of its own in engine.hip
// and the extra workgroups at the head of the plasma deposition's grid (particles_tiled.hip) both run it.
#ifndef HPS_BEAM_DEPOSIT_H_
#define HPS_BEAM_DEPOSIT_H_

#include "common.h"
#include "particle_math.h"

namespace hps {

struct BeamView { double *x, *y, *z, *ux, *uy, *uz, *w; };

// workgroups [0, nba) of the `nwg`: block A (jz into component cjz), the rest: block B (jx, jy into cjxn, cjyn)
struct BeamPairWork { BeamView a{}, b{}; long ca = 0, cb = 0; int nba = 0, nwg = 0; int cjz = -1, cjxn = -1, cjyn = -1;
                      double q_invvol = 0.0, csq_inv = 0.0; };

template <int ORDER>
__device__ __forceinline__ void beam_pair_block (const SlabView& f, const BeamPairWork& w, int block, double dx_inv, double dy_inv,
                                                 double xoff, double yoff)
{
    const bool second = block >= w.nba;
    const BeamView& b = second ? w.b : w.a;
    const long ip = (long)(second ? block - w.nba : block)*blockDim.x + threadIdx.x;
    if (ip >= (second ? w.cb : w.ca)) return;
    const double ux = b.ux[ip], uy = b.uy[ip], uz = b.uz[ip];
    const double gaminv = 1.0/sqrt(1.0 + ux*ux*w.csq_inv + uy*uy*w.csq_inv + uz*uz*w.csq_inv);
    const double wq = w.q_invvol*b.w[ip];
    double sx[ORDER + 1], sy[ORDER + 1];
    const int i0 = shape_weights<ORDER>((b.x[ip] - xoff)*dx_inv, sx);
    const int j0 = shape_weights<ORDER>((b.y[ip] - yoff)*dy_inv, sy);
#pragma unroll
    for (int iy = 0; iy <= ORDER; ++iy) {
#pragma unroll
        for (int ix = 0; ix <= ORDER; ++ix) {
            double* p = f.p + f.off(i0 + ix, j0 + iy);
            const double s = sx[ix]*sy[iy];
            // (a term that is exactly zero leaves the plane as it is: a cold beam's jx, jy -- 18 of a particle's 27 atomics)
            if (second) { const double vx = s*(wq*(ux*gaminv)), vy = s*(wq*(uy*gaminv));
                          if (vx != 0.0) atomic_add_f64(p + w.cjxn*f.ns, vx);
                          if (vy != 0.0) atomic_add_f64(p + w.cjyn*f.ns, vy); }
            else atomic_add_f64(p + w.cjz*f.ns, s*(wq*(uz*gaminv)));
        }
    }
}

} // namespace hps
#endif
