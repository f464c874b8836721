// mg_gate.h -- the multigrid's device-side stopping rule (solve_doit, HpMultiGrid.cpp:1352-1398), shared with the kernels
// that are enqueued behind a solve before the host has read its norms back (the plasma push of the slice engine)
#pragma once
#include <hip/hip_runtime.h>

namespace hps {

constexpr int MG_NSUB = 16;       // a norm slot is MG_NSUB words (workgroups spread their atomics: same-address
                                  // L2 atomics serialise at ~20 ns each); its value is the maximum over them
struct StopRule { const unsigned long long* norms; int k; double tol_rel, tol_abs; };

__device__ __forceinline__ double norm_slot (const unsigned long long* norms, int slot)
{
    unsigned long long m = 0ULL;      // non-negative doubles order like their bit patterns
#pragma unroll
    for (int q = 0; q < MG_NSUB; ++q) { const unsigned long long v = norms[slot*MG_NSUB + q]; m = v > m ? v : m; }
    return __longlong_as_double((long long)m);
}

__device__ __forceinline__ bool vcycle_active (const StopRule& sr)
{
    if (sr.k < 0) return true;
    const double res0 = norm_slot(sr.norms, 0), rhs0 = norm_slot(sr.norms, 1);
    const double prev = (sr.k == 0) ? res0 : norm_slot(sr.norms, 2 + sr.k - 1);
    const double max_norm = (rhs0 >= res0) ? rhs0 : res0;
    const double target = fmax(sr.tol_abs, fmax(sr.tol_rel, 1.e-16)*max_norm);
    return prev > target && prev <= 1.e20*max_norm;
}


} // namespace hps
