// mg_gate.h -- the multigrid's device-side stopping rule (solve_doit, HpMultiGrid.cpp:1352-1398), shared with the kernels
// that are enqueued behind a solve before the host has read its norms back (the plasma push of the slice engine)
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace hps {

constexpr int MG_NSUB = 16;       // a norm slot is MG_NSUB words (workgroups spread their atomics: same-address
                                  // L2 atomics serialise at ~20 ns each); its value is the maximum over them
struct StopRule { const unsigned long long* norms; int k; double tol_rel, tol_abs; };
// What k_post_norms does behind the V-cycles enqueued so far, handed to the kernel that is enqueued behind them instead (the gated
// plasma push): evaluate the rule `after` itself -- every workgroup, for its own gate -- and, workgroup 0, copy the solve's
// `nwords` words to the host's mapped buffer and the sequence number behind them.  src == nullptr: nothing to do.
struct MgPost { const unsigned long long* src; volatile unsigned long long* dst; int nwords; volatile unsigned long long* seq_slot;
                unsigned long long seq; StopRule after; };

__device__ __forceinline__ double norm_slot (const unsigned long long* norms, int slot)
{
    unsigned long long m = 0ULL;      // non-negative doubles order like their bit patterns
#pragma unroll
    for (int q = 0; q < MG_NSUB; ++q) { const unsigned long long v = norms[slot*MG_NSUB + q]; m = v > m ? v : m; }
    return __longlong_as_double((long long)m);
}

// The rule, evaluated by a whole wave: lane q reads sub-word q of the three slots it needs (three coalesced 128-byte
// loads in one batch -- the previous-slot index depends on k by arithmetic, not by a branch, so there is no second,
// dependent trip to memory), a butterfly over the 16 lanes of a group takes the maxima, and the verdict comes back
// wave-uniform (read-first-lane), so that the `return` behind it is a scalar branch.  (Written as 48 reads per lane the
// compiler emitted 17 vector loads per wave and batch: every wave of every workgroup pushed them through its CU's
// address unit at the head of the kernel.)  All lanes of the wave must be active.
// FRESH: the three reads go to the L2 (device-scope atomic loads): for a workgroup that evaluates the rule on norms that other
// workgroups of the SAME launch have just added to -- its CU's L1 may hold the slots' lines from the gate at the kernel's head
template <bool FRESH = false>
__device__ __forceinline__ bool vcycle_active (const StopRule& sr)
{
    if (sr.k < 0) return true;
    const int q = threadIdx.x & (MG_NSUB - 1);
    const int pslot = (sr.k == 0) ? 0 : 1 + sr.k;
    unsigned long long a, b, c;
    if (FRESH) {
        a = __hip_atomic_load(sr.norms + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b = __hip_atomic_load(sr.norms + MG_NSUB + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c = __hip_atomic_load(sr.norms + pslot*MG_NSUB + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else { a = sr.norms[q]; b = sr.norms[MG_NSUB + q]; c = sr.norms[pslot*MG_NSUB + q]; }
    // maxima over the 16 lanes of a row by DPP rotations (row_ror 8, 4, 2, 1: every lane ends with the row's maximum) -- as
    // __shfl_xor butterflies these were 24 ds_bpermute round trips through the LDS crossbar at the head of every kernel of the chain
    static_assert(MG_NSUB == 16, "the reduction below is over one DPP row");
    auto ror = [] (unsigned long long v, auto C) __attribute__((always_inline)) {
        constexpr int ctrl = 0x120 + decltype(C)::value;      // DPP row_ror:n
        const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)v, ctrl, 0xf, 0xf, false);
        const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(v >> 32), ctrl, 0xf, 0xf, false);
        return ((unsigned long long)hi << 32) | lo;
    };
#define HPS_GATE_STEP(N) { const unsigned long long a2 = ror(a, std::integral_constant<int, N>{}), b2 = ror(b, std::integral_constant<int, N>{}), \
                                                     c2 = ror(c, std::integral_constant<int, N>{}); \
                           a = a2 > a ? a2 : a; b = b2 > b ? b2 : b; c = c2 > c ? c2 : c; }
    HPS_GATE_STEP(8) HPS_GATE_STEP(4) HPS_GATE_STEP(2) HPS_GATE_STEP(1)
#undef HPS_GATE_STEP
    const double res0 = __longlong_as_double((long long)a), rhs0 = __longlong_as_double((long long)b), prev = __longlong_as_double((long long)c);
    const double max_norm = (rhs0 >= res0) ? rhs0 : res0;
    const double target = fmax(sr.tol_abs, fmax(sr.tol_rel, 1.e-16)*max_norm);
    const int act = (prev > target && prev <= 1.e20*max_norm) ? 1 : 0;
    return __builtin_amdgcn_readfirstlane(act) != 0;
}


} // namespace hps
