// depvar.hip -- prototypes of the plasma current deposition (k_deposit_tiled<2,16,51>, hipace_amd/csrc/particles_tiled.hip)
// that try to LOWER THE NUMBER of LDS fp64 atomics per particle, timed on a real sheath slice: the
// driver (scripts/deposit_variants.py) runs the engine to a slice of the headline deck and hands its tile-sorted sheet,
// its tile launch records and a scratch slab to depvar_run.
//
//   mode 0  the shipped kernel's inner loop (36 ds_add_f64 per particle), as the harness' own baseline
//   mode 1  neighbouring lanes that hit the same accumulator words (same stencil base) are merged over DPP before the
//           atomics: the follower lane hands its 36 products to the leader (wave_shl:1), only the leader issues
//   mode 2  per-launch binning inside the workgroup: particles are counted into bins by stencil base (one ds_add_rtn_u32
//           each), their payload (t_x, t_y, four products) goes to LDS in bin order, then ONE thread per bin sums the
//           bin's particles into a 3 x 3 x 4 register patch and issues 36 atomics per BIN (lanes = consecutive bins:
//           conflict-free addresses) instead of 36 per particle
//   mode 3  mode 0 without its LDS atomics and without flush traffic (loads and arithmetic only)
//   mode 5  mode 0 without the flush (loads, arithmetic, LDS atomics)
//   mode 6  mode 3 with every cell of the region flushed (loads, arithmetic, 4 x 28 x 28 global atomics per tile)
//   mode 7/8/9  persistent workgroups (768 / 1024 / 512 of them), the next tile's particles requested ahead of the current tile's atomics
//
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC depvar.hip -o libdepvar.so   (scripts/deposit_variants.py does it)
#include "../../hipace_amd/csrc/common.h"
#include "../../hipace_amd/csrc/particle_math.h"
#include "../../hipace_amd/csrc/tiling.h"

using namespace hps;

namespace {

typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(3))) unsigned lds_uint;
__device__ __forceinline__ void lds_add (double* p, double v) { __hip_atomic_fetch_add((lds_double*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned lds_inc (unsigned* p) { return __hip_atomic_fetch_add((lds_uint*)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void gadd (double* p, double v) { unsafeAtomicAdd(p, v); }

constexpr int TS = 16;
#ifndef DV_HALO
#define DV_HALO 6
#endif
constexpr int HALO = DV_HALO;
constexpr int R = TS + 2*HALO;

struct Consts { double dx_inv, dy_inv, xoff, yoff, c, c_inv, a, b, max_qsa; };
struct Comps { int c[4]; };     // jx, jy, chi, rhomjz

struct Rec { double x, y, w, ux, uy, psi; uint64_t id; };
__device__ __forceinline__ Rec fetch (const hps_plasma& pl, int ip)
{
    Rec r; r.id = pl.idcpu[ip]; r.x = pl.x[ip]; r.y = pl.y[ip]; r.w = pl.w[ip]; r.ux = pl.ux[ip]; r.uy = pl.uy[ip]; r.psi = pl.psi[ip];
    return r;
}

// what a particle deposits: stencil base (i0, j0), t_x, t_y of the order-2 shape and the four products q w {vx, vy, q mu0/(m psi), 1}
struct Dep { int i0, j0; double tx, ty, v[4]; bool on; };
__device__ __forceinline__ Dep prepare (const Rec& r, const Consts& k)
{
    Dep d; d.on = false; d.i0 = d.j0 = 0; d.tx = d.ty = 0.0; d.v[0] = d.v[1] = d.v[2] = d.v[3] = 0.0;
    if (!(r.id & HPS_ID_VALID)) return d;
    const double psi_inv = 1.0/r.psi;
    const double vx = r.ux*psi_inv, vy = r.uy*psi_inv;
    const double gp = 0.5*(psi_inv*psi_inv + vx*vx*k.c_inv*k.c_inv + vy*vy*k.c_inv*k.c_inv + 1.0);
    if (gp < 0.0 || gp > k.max_qsa || psi_inv < 0.0) return d;      // (the engine's own deposition has dropped these already)
    const double q = k.a*r.w;
    const double xm = (r.x - k.xoff)*k.dx_inv, ym = (r.y - k.yoff)*k.dy_inv;
    const double xr = floor(xm + 0.5), yr = floor(ym + 0.5);
    d.tx = xm - xr; d.ty = ym - yr; d.i0 = (int)xr - 1; d.j0 = (int)yr - 1;
    d.v[0] = q*vx; d.v[1] = q*vy; d.v[2] = q*(k.b*psi_inv); d.v[3] = q;
    d.on = true;
    return d;
}
__device__ __forceinline__ void weights (double t, double* s) { s[0] = 0.5*(0.5 - t)*(0.5 - t); s[1] = 0.75 - t*t; s[2] = 0.5*(0.5 + t)*(0.5 + t); }

__device__ __forceinline__ void global_path (const SlabView& f, const Comps& cm, const Dep& d)
{
    double sx[3], sy[3]; weights(d.tx, sx); weights(d.ty, sy);
#pragma unroll
    for (int iy = 0; iy < 3; ++iy)
#pragma unroll
        for (int ix = 0; ix < 3; ++ix) {
            double* p = f.p + f.off(d.i0 + ix, d.j0 + iy);
            const double ss = sx[ix]*sy[iy];
#pragma unroll
            for (int c = 0; c < 4; ++c) gadd(p + cm.c[c]*f.ns, ss*d.v[c]);
        }
}

__device__ __forceinline__ void flush (const SlabView& f, const Comps& cm, const double* acc, int ox, int oy, int tid)
{
    for (int s = tid; s < R*R; s += 256) {
        const int lj = s / R, li = s - lj*R;
        const int i = ox + li, j = oy + lj;
        if (i < -f.ng || i >= f.nx + f.ng || j < -f.ng || j >= f.ny + f.ng) continue;
        double* p = f.p + f.off(i, j);
#pragma unroll
        for (int c = 0; c < 4; ++c) { const double v = acc[c*R*R + s]; if (v != 0.0) gadd(p + cm.c[c]*f.ns, v); }
    }
}

__device__ __forceinline__ double dpp_wave_shl1 (double v)
{
    // lane l receives lane l+1's value (wave_shl:1 = 0x130; gfx9 DPP), lane 63 keeps its own
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int l2 = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    const int h2 = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(h2, l2);
}

// ---- modes 0, 1, 3 --------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256)
void k_dep_plain (SlabView f, hps_plasma pl, const int* __restrict__ offsets, int ntiles, int ntx, Comps cm, Consts k, double* sink)
{
    extern __shared__ __attribute__((aligned(16))) double acc[];      // [4][R*R]
    const int4 lrec = reinterpret_cast<const int4*>(offsets + tile_launch_offset(ntiles))[blockIdx.x];
    const int tile = lrec.x, pend = lrec.z;
    const int ox = (tile % ntx)*TS - HALO, oy = (tile / ntx)*TS - HALO;
    const int tid = threadIdx.x;
    constexpr int NB = 4;
    Rec rec[NB];
    const int ipb = lrec.y + tid;
    if (ipb < pend) {
#pragma unroll
        for (int u = 0; u < NB; ++u) rec[u] = fetch(pl, min(ipb + 256*u, pend - 1));
    }
    { double2* z = (double2*)acc; const double z0 = MODE == 6 ? 1.0e-300 : 0.0; for (int s = tid; s < 4*R*R/2; s += 256) z[s] = make_double2(z0, z0); }
    __syncthreads();
    double junk = 0.0;
    const int npad = ((pend - lrec.y + 255)/256)*256;      // MODE 1: every lane of a wave walks the same rounds (DPP needs them all)
    for (int ip0 = ipb; ip0 < (MODE == 1 ? lrec.y + npad : pend); ip0 += 256*NB) {
        if (ip0 != ipb) {
#pragma unroll
            for (int u = 0; u < NB; ++u) rec[u] = fetch(pl, max(min(ip0 + 256*u, pend - 1), 0));
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int ip = ip0 + 256*u;
            if (MODE != 1 && ip >= pend) break;
            if (MODE == 1 && ip - tid >= pend) break;            // (uniform over the workgroup's waves' rounds)
            Dep d = prepare(rec[u], k);
            if (MODE == 1 && ip >= pend) d.on = false;
            const int li = d.i0 - ox, lj = d.j0 - oy;
            const bool local = d.on && li >= 0 && li + 2 < R && lj >= 0 && lj + 2 < R;
            if (d.on && !local) { global_path(f, cm, d); d.on = false; }
            double sx[3], sy[3]; weights(d.tx, sx); weights(d.ty, sy);
            if constexpr (MODE == 1) {
                // key of the lane's stencil base; a lane whose predecessor has the same key (and is not itself a follower of
                // ITS predecessor) hands its products over
                const int key = d.on ? lj*R + li : -1 - (int)(threadIdx.x & 63);
                const int kprev = __builtin_amdgcn_update_dpp(-1000, key, 0x138, 0xf, 0xf, false);       // wave_shr:1: lane l gets lane l-1
                const int kpp = __builtin_amdgcn_update_dpp(-1001, kprev, 0x138, 0xf, 0xf, false);
                const bool follower = d.on && key == kprev && kprev != kpp;
                const int fnext = __builtin_amdgcn_update_dpp(0, (int)follower, 0x130, 0xf, 0xf, false);   // is lane l+1 my follower?
                double* p0 = acc + lj*R + li;
#pragma unroll
                for (int iy = 0; iy < 3; ++iy)
#pragma unroll
                    for (int ix = 0; ix < 3; ++ix) {
                        const double ss = sx[ix]*sy[iy];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            double val = ss*d.v[c];
                            const double nb = dpp_wave_shl1(val);
                            if (fnext) val += nb;
                            if (d.on && !follower) lds_add(p0 + iy*R + ix + c*R*R, val);
                        }
                    }
            } else {
                if (!d.on) continue;
                double* p0 = acc + lj*R + li;
#pragma unroll
                for (int iy = 0; iy < 3; ++iy)
#pragma unroll
                    for (int ix = 0; ix < 3; ++ix) {
                        const double ss = sx[ix]*sy[iy];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if constexpr (MODE == 0 || MODE == 5) lds_add(p0 + iy*R + ix + c*R*R, ss*d.v[c]);
                            else junk += ss*d.v[c];
                        }
                    }
            }
        }
    }
    if ((MODE == 3 || MODE == 6) && junk == 1.2345e-300) sink[tid] = junk;
    __syncthreads();
    if (MODE == 5) { if (acc[tid] == 1.2345e-300) sink[tid] = acc[tid]; return; }
    flush(f, cm, acc, ox, oy, tid);
}

// ---- mode 2: bins by stencil base, one thread per bin ---------------------------------------------------------------------
template <int NBC>      // particles per thread and chunk (chunk = 256*NBC particles)
__global__ __launch_bounds__(256)
void k_dep_bins (SlabView f, hps_plasma pl, const int* __restrict__ offsets, int ntiles, int ntx, Comps cm, Consts k)
{
    constexpr int CH = 256*NBC;
    constexpr int NBIN = R*R;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* acc = lds;                                  // [4][R*R]
    double2* pay = (double2*)(lds + 4*R*R);             // [3][CH] double2: (tx, ty), (v0, v1), (v2, v3)
    unsigned* start = (unsigned*)(pay + 3*CH);          // [NBIN + 1] counts, then exclusive starts
    unsigned* wsum = start + NBIN + 1 + 3;              // [4] per-wave totals of the scan
    const int4 lrec = reinterpret_cast<const int4*>(offsets + tile_launch_offset(ntiles))[blockIdx.x];
    const int tile = lrec.x, pend = lrec.z;
    const int ox = (tile % ntx)*TS - HALO, oy = (tile / ntx)*TS - HALO;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    Rec rec[NBC];
    if (lrec.y + tid < pend) {
#pragma unroll
        for (int u = 0; u < NBC; ++u) rec[u] = fetch(pl, min(lrec.y + tid + 256*u, pend - 1));
    }
    { double2* z = (double2*)acc; for (int s = tid; s < 4*R*R/2; s += 256) z[s] = make_double2(0.0, 0.0); }
    for (int base = lrec.y; base < pend; base += CH) {
        for (int s = tid; s <= NBIN; s += 256) start[s] = 0u;
        if (base != lrec.y) {
#pragma unroll
            for (int u = 0; u < NBC; ++u) rec[u] = fetch(pl, min(base + tid + 256*u, pend - 1));
        }
        __syncthreads();
        // phase A: bin counts, rank of each particle in its bin
        Dep d[NBC]; int key[NBC]; unsigned rank[NBC];
#pragma unroll
        for (int u = 0; u < NBC; ++u) {
            const int ip = base + tid + 256*u;
            d[u] = prepare(rec[u], k);
            if (ip >= pend) d[u].on = false;
            const int li = d[u].i0 - ox, lj = d[u].j0 - oy;
            const bool local = d[u].on && li >= 0 && li + 2 < R && lj >= 0 && lj + 2 < R;
            if (d[u].on && !local) { global_path(f, cm, d[u]); d[u].on = false; }
            key[u] = lj*R + li; rank[u] = 0u;
            if (d[u].on) rank[u] = lds_inc(start + key[u]);
        }
        __syncthreads();
        // phase B: exclusive scan of the NBIN counts (4 bins per thread, wave scan over DPP-free shuffles, 4 wave totals)
        unsigned c4[4]; unsigned tsum = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int b = 4*tid + q; c4[q] = b < NBIN ? start[b] : 0u; tsum += c4[q]; }
        unsigned incl = tsum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned n = __shfl_up(incl, o, 64); if (lane >= o) incl += n; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        unsigned woff = 0u;
        for (int q = 0; q < wv; ++q) woff += wsum[q];
        unsigned run = woff + incl - tsum;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int b = 4*tid + q; if (b < NBIN) start[b] = run; run += c4[q]; }
        if (tid == 255) start[NBIN] = run;
        __syncthreads();
        // phase C: payload in bin order
#pragma unroll
        for (int u = 0; u < NBC; ++u) {
            if (!d[u].on) continue;
            const unsigned s = start[key[u]] + rank[u];
            pay[s] = make_double2(d[u].tx, d[u].ty);
            pay[CH + s] = make_double2(d[u].v[0], d[u].v[1]);
            pay[2*CH + s] = make_double2(d[u].v[2], d[u].v[3]);
        }
        __syncthreads();
        // phase D: one thread per bin
        for (int b = tid; b < NBIN; b += 256) {
            const unsigned s0 = start[b], s1 = start[b + 1];
            if (s1 == s0) continue;
            double patch[4][9];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int m = 0; m < 9; ++m) patch[c][m] = 0.0;
            for (unsigned s = s0; s < s1; ++s) {
                const double2 t = pay[s], va = pay[CH + s], vb = pay[2*CH + s];
                double sx[3], sy[3]; weights(t.x, sx); weights(t.y, sy);
#pragma unroll
                for (int iy = 0; iy < 3; ++iy)
#pragma unroll
                    for (int ix = 0; ix < 3; ++ix) {
                        const double ss = sx[ix]*sy[iy];
                        patch[0][iy*3 + ix] = fma(ss, va.x, patch[0][iy*3 + ix]);
                        patch[1][iy*3 + ix] = fma(ss, va.y, patch[1][iy*3 + ix]);
                        patch[2][iy*3 + ix] = fma(ss, vb.x, patch[2][iy*3 + ix]);
                        patch[3][iy*3 + ix] = fma(ss, vb.y, patch[3][iy*3 + ix]);
                    }
            }
            double* p0 = acc + b;
#pragma unroll
            for (int iy = 0; iy < 3; ++iy)
#pragma unroll
                for (int ix = 0; ix < 3; ++ix)
#pragma unroll
                    for (int c = 0; c < 4; ++c) lds_add(p0 + iy*R + ix + c*R*R, patch[c][iy*3 + ix]);
        }
        __syncthreads();
    }
    __syncthreads();
    flush(f, cm, acc, ox, oy, tid);
}

// ---- mode 7: persistent workgroups, the NEXT tile's particles requested while the current tile's atomics and flush run -------
// (the shipped kernel loads a whole tile in one batch, works on it and leaves: its own prefetch switch has nothing to prefetch,
// and loads + arithmetic, LDS atomics and flush add up -- 38 + 17 + 10.5 us on a sheath slice)
__device__ __forceinline__ void persist_fetch (const hps_plasma& pl, const int4 rec, int tid, Rec (&r)[4])
{
    const int ipb = rec.y + tid;
    if (ipb < rec.z) {
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = fetch(pl, min(ipb + 256*u, rec.z - 1));
    }
}
__device__ __forceinline__ void persist_tile (const SlabView& f, const hps_plasma& pl, const Comps& cm, const Consts& k, double* acc,
                                              const int4 lrec, int ntx, int tid, Rec (&first)[4])
{
    const int tile = lrec.x, pend = lrec.z;
    const int ox = (tile % ntx)*TS - HALO, oy = (tile / ntx)*TS - HALO;
    { double2* z = (double2*)acc; for (int s = tid; s < 4*R*R/2; s += 256) z[s] = make_double2(0.0, 0.0); }
    __syncthreads();
    const int ipb = lrec.y + tid;
    for (int ip0 = ipb; ip0 < pend; ip0 += 1024) {
        Rec rec[4];
        if (ip0 != ipb) {
#pragma unroll
            for (int u = 0; u < 4; ++u) rec[u] = fetch(pl, min(ip0 + 256*u, pend - 1));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ip = ip0 + 256*u;
            if (ip >= pend) break;
            const Dep d = prepare(ip0 == ipb ? first[u] : rec[u], k);
            if (!d.on) continue;
            const int li = d.i0 - ox, lj = d.j0 - oy;
            if (!(li >= 0 && li + 2 < R && lj >= 0 && lj + 2 < R)) { global_path(f, cm, d); continue; }
            double sx[3], sy[3]; weights(d.tx, sx); weights(d.ty, sy);
            double* p0 = acc + lj*R + li;
#pragma unroll
            for (int iy = 0; iy < 3; ++iy)
#pragma unroll
                for (int ix = 0; ix < 3; ++ix) {
                    const double ss = sx[ix]*sy[iy];
#pragma unroll
                    for (int c = 0; c < 4; ++c) lds_add(p0 + iy*R + ix + c*R*R, ss*d.v[c]);
                }
        }
    }
    __syncthreads();
    flush(f, cm, acc, ox, oy, tid);
    __syncthreads();
}
__global__ __launch_bounds__(256)
void k_dep_persist (SlabView f, hps_plasma pl, const int* __restrict__ offsets, int ntiles, int ntx, Comps cm, Consts k)
{
    extern __shared__ __attribute__((aligned(16))) double acc[];      // [4][R*R]
    const int4* recs = reinterpret_cast<const int4*>(offsets + tile_launch_offset(ntiles));
    const int tid = threadIdx.x, G = gridDim.x;
    Rec a[4], b[4];
    int t = blockIdx.x;
    if (t >= ntiles) return;
    int4 ra = recs[t], rb = ra;
    persist_fetch(pl, ra, tid, a);
    while (true) {
        const bool more_b = t + G < ntiles;
        if (more_b) { rb = recs[t + G]; persist_fetch(pl, rb, tid, b); }
        persist_tile(f, pl, cm, k, acc, ra, ntx, tid, a);
        if (!more_b) break;
        t += G;
        const bool more_a = t + G < ntiles;
        if (more_a) { ra = recs[t + G]; persist_fetch(pl, ra, tid, a); }
        persist_tile(f, pl, cm, k, acc, rb, ntx, tid, b);
        if (!more_a) break;
        t += G;
    }
}

} // namespace

extern "C" int depvar_run (int mode, hps_slab slab, hps_plasma pl, hps_geom g, const int* offsets_dev, int ntiles, int ntx,
                           double charge, double mass, int reps, float* ms_per_launch)
{
    SlabView f(slab);
    Consts k{1.0/g.dx, 1.0/g.dy, g.xoff, g.yoff, g.c, 1.0/g.c, charge*invvol_of(g), charge*g.mu0/mass, 35.0};
    Comps cm{{HPS_C_JX, HPS_C_JY, HPS_C_CHI, HPS_C_RHOMJZ}};
    double* sink = nullptr;
    if (hipMalloc(&sink, 256*sizeof(double)) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds_plain = 4*R*R*sizeof(double);
    auto launch = [&] () {
        switch (mode) {
        case 0: hipLaunchKernelGGL((k_dep_plain<0>), dim3(ntiles), dim3(256), lds_plain, 0, f, pl, offsets_dev, ntiles, ntx, cm, k, sink); break;
        case 1: hipLaunchKernelGGL((k_dep_plain<1>), dim3(ntiles), dim3(256), lds_plain, 0, f, pl, offsets_dev, ntiles, ntx, cm, k, sink); break;
        case 5: hipLaunchKernelGGL((k_dep_plain<5>), dim3(ntiles), dim3(256), lds_plain, 0, f, pl, offsets_dev, ntiles, ntx, cm, k, sink); break;
        case 6: hipLaunchKernelGGL((k_dep_plain<6>), dim3(ntiles), dim3(256), lds_plain, 0, f, pl, offsets_dev, ntiles, ntx, cm, k, sink); break;
        case 3: hipLaunchKernelGGL((k_dep_plain<3>), dim3(ntiles), dim3(256), lds_plain, 0, f, pl, offsets_dev, ntiles, ntx, cm, k, sink); break;
        case 2: { constexpr int NBC = 4; const size_t lds = lds_plain + 3*256*NBC*sizeof(double2) + (R*R + 8)*sizeof(unsigned);
                  (void)hipFuncSetAttribute((const void*)k_dep_bins<NBC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                  hipLaunchKernelGGL((k_dep_bins<NBC>), dim3(ntiles), dim3(256), lds, 0, f, pl, offsets_dev, ntiles, ntx, cm, k); } break;
        case 4: { constexpr int NBC = 2; const size_t lds = lds_plain + 3*256*NBC*sizeof(double2) + (R*R + 8)*sizeof(unsigned);
                  (void)hipFuncSetAttribute((const void*)k_dep_bins<NBC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                  hipLaunchKernelGGL((k_dep_bins<NBC>), dim3(ntiles), dim3(256), lds, 0, f, pl, offsets_dev, ntiles, ntx, cm, k); } break;
        case 7: hipLaunchKernelGGL(k_dep_persist, dim3(ntiles < 768 ? ntiles : 768), dim3(256), lds_plain, 0, f, pl, offsets_dev, ntiles, ntx, cm, k); break;
        case 8: hipLaunchKernelGGL(k_dep_persist, dim3(ntiles < 1024 ? ntiles : 1024), dim3(256), lds_plain, 0, f, pl, offsets_dev, ntiles, ntx, cm, k); break;
        case 9: hipLaunchKernelGGL(k_dep_persist, dim3(ntiles < 512 ? ntiles : 512), dim3(256), lds_plain, 0, f, pl, offsets_dev, ntiles, ntx, cm, k); break;
        default: break;
        }
    };
    launch();                                   // warm-up (and the launch whose output the driver compares)
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return 3;
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    if (ms_per_launch) *ms_per_launch = reps > 0 ? ms/reps : 0.f;
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(sink);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
