// common.h -- shared host/device helpers of libhpslice (gfx950 only).
#ifndef HPS_COMMON_H_
#define HPS_COMMON_H_

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/hpslice.h"

namespace hps {

void set_error (const std::string& msg);

#define HPS_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            hps::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));           \
            return HPS_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

#define HPS_REQUIRE(cond, msg)                                                           \
    do { if (!(cond)) { hps::set_error(msg); return HPS_ERR_ARG; } } while (0)

// Slab view usable in device code.  (i,j) are cell indices, guards at negative indices.
struct SlabView {
    double* p; int nx, ny, ng; long js, ns;
    __host__ __device__ SlabView () : p(nullptr), nx(0), ny(0), ng(0), js(0), ns(0) {}
    __host__ __device__ explicit SlabView (const hps_slab& s)
        : p(s.p), nx(s.nx), ny(s.ny), ng(s.ng), js(s.jstride), ns(s.nstride) {}
    __device__ __forceinline__ long off (int i, int j) const { return (long)(i + ng) + (long)(j + ng)*js; }
    __device__ __forceinline__ double* comp (int n) const { return p + (long)n*ns; }
    __device__ __forceinline__ double& operator() (int i, int j, int n) const { return p[off(i, j) + (long)n*ns]; }
};

// ---- B-spline shape factors (cell-centred grid; xmid = (x - offset)/dx in index space) ------
// Standard weights W_order; returns left-most cell.  Same polynomials as the reference's
// compute_shape_factor (particles/particles_utils/ShapeFactors.H:27-108).
template <int ORDER>
__device__ __forceinline__ int shape_weights (double xmid, double* s)
{
    if constexpr (ORDER == 0) {
        s[0] = 1.0;
        return (int)floor(xmid + 0.5);
    } else if constexpr (ORDER == 1) {
        const double xf = floor(xmid);
        const double t = xmid - xf;
        s[0] = 1.0 - t; s[1] = t;
        return (int)xf;
    } else if constexpr (ORDER == 2) {
        const double xr = floor(xmid + 0.5);
        const double t = xmid - xr;            // in [-1/2, 1/2)
        s[0] = 0.5*(0.5 - t)*(0.5 - t);
        s[1] = 0.75 - t*t;
        s[2] = 0.5*(0.5 + t)*(0.5 + t);
        return (int)xr - 1;
    } else {
        const double xf = floor(xmid);
        const double t = xmid - xf;
        const double u = 1.0 - t;
        s[0] = (1.0/6.0)*u*u*u;
        s[1] = 2.0/3.0 - t*t*(1.0 - 0.5*t);
        s[2] = 2.0/3.0 - u*u*(1.0 - 0.5*u);
        s[3] = (1.0/6.0)*t*t*t;
        return (int)xf - 1;
    }
}

// Order 2, the same weights computed ONCE: the plain set lives on three of the four cells -- (w0, w1, w2) at cells 0..2 for
// t < 1/2, the same three polynomials of u = t - 1 at cells 1..3 otherwise (0.5 t^2 - 1.5 t + 1.125 = w0(t - 1), ...) -- where
// nodal_weights<2> evaluates both polynomials of every cell and selects.  w / hi are what the push's 3 x 3 gather of the plain
// fields needs (no second selection from s), s the four-cell set for the Psi gather.
__device__ __forceinline__ int nodal_weights2_w3 (double xmid, double* s, double* d, double* w, bool& hi)
{
    const double xf = floor(xmid);
    const double t = xmid - xf;
    const double t2 = t*t;
    const bool lo = t < 0.5;
    const double u = lo ? t : t - 1.0, u2 = u*u;
    w[0] = 0.5*u2 - 0.5*u + 0.125;
    w[1] = 0.75 - u2;
    w[2] = 0.5*u2 + 0.5*u + 0.125;
    s[0] = lo ? w[0] : 0.0;
    s[1] = lo ? w[1] : w[0];
    s[2] = lo ? w[2] : w[1];
    s[3] = lo ? 0.0 : w[2];
    d[0] = -(-0.5*t2 + t - 0.5);
    d[1] = -(1.5*t2 - 2.0*t);
    d[2] = -(-1.5*t2 + t + 0.5);
    d[3] = -(0.5*t2);
    hi = !lo;
    return (int)xf - 1;
}

// Gather stencil: weights of the "nodal derivative" set (derivative_type 1 of
// ShapeFactors.H:275-367) on ORDER+2 cells: s[k] interpolates the field, d[k] (= -dS/dx in
// cells) differentiates it on the fly.  Returns left-most cell.
template <int ORDER>
__device__ __forceinline__ int nodal_weights (double xmid, double* s, double* d)
{
    if constexpr (ORDER == 0) {
        const double xf = floor(xmid);
        const double t = xmid - xf;
        s[0] = (t < 0.5) ? 1.0 : 0.0;  d[0] = 1.0;
        s[1] = (t < 0.5) ? 0.0 : 1.0;  d[1] = -1.0;
        return (int)xf;
    } else if constexpr (ORDER == 1) {
        const double xf = floor(xmid + 0.5);
        const double t = xmid + 0.5 - xf;
        s[0] = (t < 0.5) ? 0.5 - t : 0.0;          d[0] = -(t - 1.0);
        s[1] = (t < 0.5) ? t + 0.5 : 1.5 - t;      d[1] = -(1.0 - 2.0*t);
        s[2] = (t < 0.5) ? 0.0 : t - 0.5;          d[2] = -t;
        return (int)xf - 1;
    } else if constexpr (ORDER == 2) {
        const double xf = floor(xmid);
        const double t = xmid - xf;
        const double t2 = t*t;
        const bool lo = t < 0.5;
        s[0] = lo ? 0.5*t2 - 0.5*t + 0.125 : 0.0;
        s[1] = lo ? 0.75 - t2 : 0.5*t2 - 1.5*t + 1.125;
        s[2] = lo ? 0.5*t2 + 0.5*t + 0.125 : -t2 + 2.0*t - 0.25;
        s[3] = lo ? 0.0 : 0.5*t2 - 0.5*t + 0.125;
        d[0] = -(-0.5*t2 + t - 0.5);
        d[1] = -(1.5*t2 - 2.0*t);
        d[2] = -(-1.5*t2 + t + 0.5);
        d[3] = -(0.5*t2);
        return (int)xf - 1;
    } else {
        const double xf = floor(xmid + 0.5);
        const double t = xmid + 0.5 - xf;
        const double t2 = t*t, t3 = t2*t;
        const bool lo = t < 0.5;
        s[0] = lo ? -1.0/6.0*t3 + 0.25*t2 - 0.125*t + 1.0/48.0 : 0.0;
        s[1] = lo ? 0.5*t3 - 0.25*t2 - 0.625*t + 23.0/48.0 : -1.0/6.0*t3 + 0.75*t2 - 1.125*t + 9.0/16.0;
        s[2] = lo ? -0.5*t3 - 0.25*t2 + 0.625*t + 23.0/48.0 : 0.5*t3 - 1.75*t2 + 1.375*t + 17.0/48.0;
        s[3] = lo ? 1.0/6.0*t3 + 0.25*t2 + 0.125*t + 1.0/48.0 : -0.5*t3 + 1.25*t2 - 0.375*t + 5.0/48.0;
        s[4] = lo ? 0.0 : 1.0/6.0*t3 - 0.25*t2 + 0.125*t - 1.0/48.0;
        d[0] = -(1.0/6.0*t3 - 0.5*t2 + 0.5*t - 1.0/6.0);
        d[1] = -(-2.0/3.0*t3 + 1.5*t2 - 0.5*t - 0.5);
        d[2] = -(t3 - 1.5*t2 - 0.5*t + 0.5);
        d[3] = -(-2.0/3.0*t3 + 0.5*t2 + 0.5*t + 1.0/6.0);
        d[4] = -(1.0/6.0*t3);
        return (int)xf - 2;
    }
}

// Explicit-deposition stencil: "centred derivative" set (derivative_type 2 of
// ShapeFactors.H:368-461) on ORDER+3 cells.  Returns left-most cell.
template <int ORDER>
__device__ __forceinline__ int centred_weights (double xmid, double* s, double* d)
{
    if constexpr (ORDER == 0) {
        const double xf = floor(xmid + 0.5);
        s[0] = 0.0; d[0] = 0.5;
        s[1] = 1.0; d[1] = 0.0;
        s[2] = 0.0; d[2] = -0.5;
        return (int)xf - 1;
    } else if constexpr (ORDER == 1) {
        const double xf = floor(xmid);
        const double t = xmid - xf;
        s[0] = 0.0;     d[0] = -(0.5*t - 0.5);
        s[1] = 1.0 - t; d[1] = 0.5*t;
        s[2] = t;       d[2] = -(0.5 - 0.5*t);
        s[3] = 0.0;     d[3] = -(0.5*t);
        return (int)xf - 1;
    } else if constexpr (ORDER == 2) {
        const double xf = floor(xmid + 0.5);
        const double t = xmid + 0.5 - xf;
        const double t2 = t*t;
        s[0] = 0.0;                    d[0] = -(-0.25*t2 + 0.5*t - 0.25);
        s[1] = 0.5*t2 - t + 0.5;       d[1] = -(0.5*t2 - 0.5*t - 0.25);
        s[2] = -t2 + t + 0.5;          d[2] = -(0.25 - 0.5*t);
        s[3] = 0.5*t2;                 d[3] = -(-0.5*t2 + 0.5*t + 0.25);
        s[4] = 0.0;                    d[4] = -(0.25*t2);
        return (int)xf - 2;
    } else {
        const double xf = floor(xmid);
        const double t = xmid - xf;
        const double t2 = t*t, t3 = t2*t;
        s[0] = 0.0;                                         d[0] = -(1.0/12.0*t3 - 0.25*t2 + 0.25*t - 1.0/12.0);
        s[1] = -1.0/6.0*t3 + 0.5*t2 - 0.5*t + 1.0/6.0;      d[1] = -(-0.25*t3 + 0.5*t2 - 1.0/3.0);
        s[2] = 0.5*t3 - t2 + 2.0/3.0;                       d[2] = -(1.0/6.0*t3 - 0.5*t);
        s[3] = -0.5*t3 + 0.5*t2 + 0.5*t + 1.0/6.0;          d[3] = -(1.0/6.0*t3 - 0.5*t2 + 1.0/3.0);
        s[4] = 1.0/6.0*t3;                                  d[4] = -(-0.25*t3 + 0.25*t2 + 0.25*t + 1.0/12.0);
        s[5] = 0.0;                                         d[5] = -(1.0/12.0*t3);
        return (int)xf - 2;
    }
}

// native global fp64 atomic add on gfx950 (global_atomic_add_f64, no CAS loop)
__device__ __forceinline__ void atomic_add_f64 (double* addr, double v)
{
    unsafeAtomicAdd(addr, v);
}

inline int ceil_div (long a, long b) { return (int)((a + b - 1)/b); }

} // namespace hps


// A wave waits until its own outstanding global atomics / stores are acknowledged, ahead of a "this workgroup is done" count.
// gfx942 / gfx950 (the Makefile's ARCH): no-return atomics and stores are counted in vmcnt and device-scope atomics are
// performed at the memory side of the XCDs' L2s, so `s_waitcnt vmcnt(0)` + device-scope atomic loads of the counters is
// enough -- and a __threadfence() there writes the XCD's L2 back, once per workgroup (DESIGN.md section 4 "tried").  Any
// other target (gfx10+ count stores in vscnt / storecnt) gets the formal device-scope fence.
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(__gfx942__) || defined(__gfx950__))
#define HPS_OWN_ATOMICS_ACKNOWLEDGED() __threadfence()
#else
#define HPS_OWN_ATOMICS_ACKNOWLEDGED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

// Ahead of the sequence word of a post to mapped HOST memory: the payload's stores (volatile: system-coherent write-through,
// `sc0 sc1`, never dirty in an L2) have been acknowledged.  A __threadfence_system() there also writes the XCD's L2 back --
// microseconds on the one chain of launches a slice is (k_post_norms: 4.9 us per slice for 80 words).  Other targets keep
// the formal fence.
#if defined(HPS_KEEP_SYSTEM_FENCE) || (defined(__HIP_DEVICE_COMPILE__) && !(defined(__gfx942__) || defined(__gfx950__)))
#define HPS_HOST_STORES_ACKNOWLEDGED() __threadfence_system()
#else
#define HPS_HOST_STORES_ACKNOWLEDGED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

#endif
