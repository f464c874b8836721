// poisson_src.h -- a batch of Poisson solves whose sources are formed from other planes while the first transform pass loads
// its rows (poisson.hip; used by the slice engine): source of solve b at (row j, column i) = sum over its 1-2 pairs of
// c * (p[j*pitch + i] - q[j*pitch + i]) (q may be null).
#ifndef HPS_POISSON_SRC_H_
#define HPS_POISSON_SRC_H_
#include "common.h"
namespace hps {
struct PoissonSrc { int npairs; const double* p[2]; const double* q[2]; double c[2]; };
bool poisson_sources_fusable (void* poisson_handle);
int poisson_solve_batch_src (void* poisson_handle, int nb, const PoissonSrc* spec, long src_pitch, hps_slab dst, const int* dst_comps, hipStream_t st);
}
#endif
