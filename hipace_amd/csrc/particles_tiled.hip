// particles_tiled.hip -- LDS-tile variants of the particle kernels for a tile-sorted plasma sheet
// (see sort.hip).  One 256-thread workgroup owns one TS x TS-cell tile: it keeps an LDS image of
// the tile plus a 6-cell halo (R = TS+12 cells per side; 8 cells cost the explicit deposition one workgroup per CU of occupancy), streams the tile's particles (contiguous
// in the sorted SoA, fully coalesced) and
//   * deposit:   accumulates all stencil contributions with LDS fp64 atomics (ds_add_f64) and
//                flushes the non-zero cells once with global atomics (halo cells overlap
//                neighbouring tiles) -- ~1/10 of the global atomics of the per-particle scatter;
//   * explicit deposit: additionally serves the 4 per-cell field reads from the LDS image;
//   * gather+push: serves the 5x16 gathered values per particle from the LDS image.
// Particles that drifted out of their home tile's halo since the last sort take the global-memory
// path of the same arithmetic, so results never depend on how stale the sort is.
// Arithmetic identical to particles.hip (reference file:line cited there).
#include "common.h"
#include <type_traits>
#include "particle_math.h"
#include "tiling.h"
#include "mg_gate.h"
#include "beam_deposit.h"

#include <cstdlib>

namespace hps {

// LDS row pitch of the explicit deposition's images: R + PAD doubles.  PAD 2 is the best for the row-by-row numbering of
// a tile's cells (measured -2.5 % against 0; PAD 8 with it: 136.5 against 124.8 us); PAD 8 (pitch 36 = +4 mod 32) goes
// with the 4 x 8-block numbering (sort.hip: cell_in_tile, HPS_CELL_BLOCK_W=4), which is not the default.
static int expl_pad ()
{
    static int pad = -1;
    if (pad < 0) {
        pad = 2;
        if (const char* e = std::getenv("HPS_EXPL_PAD")) { const int v = std::atoi(e); if (v == 2 || v == 8) pad = v; }
    }
    return pad;
}
#ifndef HPS_TILE_HALO
#define HPS_TILE_HALO 6
#endif
constexpr int TILE_HALO = HPS_TILE_HALO;

// optional shader-clock stamps of one workgroup (hps_particles_debug_stamps)
#ifdef HPS_STAMPS      // diagnostic build only (make stamps): reading the pointer is a dependent trip to memory at the head of the kernel
__device__ long long* g_pt_dbg = nullptr;
#define PT_STAMP(i) do { if (g_pt_dbg && blockIdx.x == 2000 && threadIdx.x == 0) g_pt_dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PT_STAMP(i) do { } while (0)
#endif

struct CompSlots { int n; int comp[6]; };   // active deposition components, in DepComps order

typedef __attribute__((address_space(3))) double lds_double;

// ds_add_f64: the explicit LDS address space keeps the compiler from emitting a flat atomic
__device__ __forceinline__ void lds_add (double* p, double v)
{
    __hip_atomic_fetch_add((lds_double*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ double lds_get (const double* p)
{
    return *(const lds_double*)p;
}

// |a|^2 of the laser envelope from one more plane of the tile's LDS image (the LASER variants of the three kernels): the
// plain-shape gather of doLaserGatherShapeN (FieldGather.H:236-331) with the same weights and summation order as
// laser_gather / laser_gather_grad (particle_math.h), so the result does not depend on which path a particle takes.
// (li, lj) = left-most stencil cell relative to the image, pitch = row pitch of the plane.
template <int ORDER>
__device__ __forceinline__ double laser_gather_lds (const double* a, int pitch, int li, int lj, const double* lx, const double* ly)
{
    double A = 0.0;
#pragma unroll
    for (int iy = 0; iy <= ORDER; ++iy)
#pragma unroll
        for (int ix = 0; ix <= ORDER; ++ix) A += lx[ix]*ly[iy]*lds_get(a + (lj + iy)*pitch + li + ix);
    return A;
}
template <int ORDER>
__device__ __forceinline__ void laser_gather_grad_lds (const double* a, int pitch, int li, int lj, const double* lx, const double* ly,
                                                       double dx_inv, double dy_inv, double& A, double& ADx, double& ADy)
{
    A = 0.0; ADx = 0.0; ADy = 0.0;
#pragma unroll
    for (int iy = 0; iy <= ORDER; ++iy)
#pragma unroll
        for (int ix = 0; ix <= ORDER; ++ix) {
            const double* c = a + (lj + iy)*pitch + li + ix;
            const double w = lx[ix]*ly[iy];
            A += w*lds_get(c);
            ADx += w*0.5*dx_inv*(lds_get(c + 1) - lds_get(c - 1));
            ADy += w*0.5*dy_inv*(lds_get(c + pitch) - lds_get(c - pitch));
        }
}

// What a workgroup of a tile kernel works on: {tile, first particle, end} from the launch records of the sort (heaviest
// tile first), or -- the first tw.nwg workgroups of the grid -- 256 of the particles appended BEHIND the tile-sorted body
// (electrons released by an ionisable species since the last sort).  Those take the kernels' path for a particle outside
// its tile's halo (slab gathers / slab atomics): a few hundred particles whose per-particle kernels were chains of
// latencies of 10-40 us per launch, three launches per slice, now hidden beside the tiles' work.
// live_n: the species' particle count on the device (the push of a slice is enqueued before the host knows how many
// electrons the slice has released).
__device__ __forceinline__ int4 tile_record (const int* __restrict__ offsets, const TailWork& tw, long n, bool& tail)
{
    const int b0 = (int)blockIdx.x - tw.extra;       // (tw.extra workgroups at the head of the grid are not ours: the beam's deposition)
    const int b = b0 - tw.nwg;
    tail = b < 0;
    if (tail) {
        const long nn = tw.live_n ? (long)*tw.live_n : n;
        const long first = (long)tw.first + 256L*b0;
        return make_int4(0, (int)min(first, nn), (int)min(first + 256, nn), 0);
    }
    return reinterpret_cast<const int4*>(offsets + tile_launch_offset(gridDim.x - tw.nwg - tw.extra))[b];
}
constexpr int TAIL_ORIGIN = -(1 << 24);      // "tile origin" of a tail workgroup: no particle is local to it

template <int R, int RP = R, int NT = 256, int NC = 6>
__device__ __forceinline__ void load_region (double* img, const SlabView& f, const int* comps, int nc, int ox, int oy, int tid);

// Row pitch of the deposition's LDS accumulators = R + HPS_DEP_PAD doubles.  The lanes of a wave work on particles of 64
// consecutive cells (four tile rows at the sort's (rank, cell) order): with pitch 20 the four rows fall on the 32 eight-byte
// bank slots at offsets 0, 20, 8, 28 -- three rows deep on some slots, one on others; with pitch 48 (HPS_DEP_PAD=28) rows
// alternate between the two halves of the banks.
// the explicit deposition in two passes (image particles straight-line, slab particles behind them; 0: one loop with the choice at every stencil cell)
#ifndef HPS_EXPL_TWO_PASS
#define HPS_EXPL_TWO_PASS 1
#endif
#ifndef HPS_EXPL_ROW_BARRIER
#define HPS_EXPL_ROW_BARRIER 1
#endif
#ifndef HPS_DEP_PAD
#define HPS_DEP_PAD 0
#endif
#ifndef HPS_DEP_FAST_RCP
#define HPS_DEP_FAST_RCP 0
#endif
#ifndef HPS_DEP_NB
#define HPS_DEP_NB 4
#endif
#ifndef HPS_DEP_PIPE
#define HPS_DEP_PIPE 0
#endif
// (An XCD-chunked tile order -- contiguous tile runs per XCD -- was measured slower here: 916 vs 953 slices/s.)
// MASK: compile-time set of deposited components (bit c = DepComps entry c), -1 = decide at run time.
// With a compile-time set the 9x4 accumulations are straight-line ds_add_f64 with immediate offsets.
// VBW: PartConsts::valid_by_w at compile time (as a run-time branch in the fetch it split the batch of loads: 76 against 70 us)
template <int ORDER, int TS, int MASK, bool LASER = false, bool VBW = false>
__global__ __launch_bounds__(256)
void k_deposit_tiled (SlabView f, hps_plasma pl, const int* __restrict__ offsets, int ntx, DepComps cm,
                      PartConsts k, int* n_qsa, int* n_fallback, const int* __restrict__ tile_flag, TailWork tw, BeamPairWork bw,
                      const int* go)
{
    // an iteration of the predictor-corrector loop enqueued past the loop's end (engine.hip: pc_enqueue_iteration)
    if (go && *go == 0) return;
    // the first bw.nwg workgroups: the static beam's deposits of this slice (beam_deposit.h) -- a 7.9 us launch of a few
    // thousand particles that nothing ahead of the Sx/Sy initialisation waits for, off the slice's chain of launches.  (At the
    // head of the grid: at its end they were a tail of 6 us behind the last tile.)
    if ((int)blockIdx.x < bw.nwg) {
        beam_pair_block<ORDER>(f, bw, (int)blockIdx.x, k.dx_inv, k.dy_inv, k.xoff, k.yoff);
        return;
    }
    constexpr int R = TS + 2*TILE_HALO;
    constexpr int RP = R + HPS_DEP_PAD, PL = RP*R;      // row pitch and plane size of the accumulators (HPS_DEP_PAD, below)
    // an ionisable species: tiles that hold no charged ion have nothing to deposit (flag written by the species' push)
    if (tile_flag && !tile_flag[offsets[gridDim.x + 2 + blockIdx.x]]) return;
    extern __shared__ __attribute__((aligned(16))) double acc[];     // [active comps][R][RP]
    const int gc[6] = {cm.jx, cm.jy, cm.jz, cm.rho, cm.chi, cm.rhomjz};
    // slot of each component (compacted)
    int slot[6]; int na = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const bool on = (MASK >= 0) ? ((MASK >> c) & 1) : (gc[c] >= 0);
        slot[c] = on ? na++ : -1;
    }

    bool tail;
    const int4 lrec = tile_record(offsets, tw, pl.n, tail);     // {tile, first, end}: launch order = heaviest tile first (sort.hip)
    const int tile = lrec.x;
    const int ox = tail ? TAIL_ORIGIN : (tile % ntx)*TS - TILE_HALO, oy = tail ? TAIL_ORIGIN : (tile / ntx)*TS - TILE_HALO;
    const int tid = threadIdx.x;
    PT_STAMP(0);
    const int pend = lrec.z;
    int nfb = 0;
    struct Rec { double x, y, w, ux, uy, psi; uint64_t id; int ion; };
    auto fetch = [&] (int ip) {
        Rec r;
        r.x = pl.x[ip]; r.y = pl.y[ip]; r.w = pl.w[ip];
        r.ux = pl.ux[ip]; r.uy = pl.uy[ip]; r.psi = pl.psi[ip];
        if constexpr (VBW) r.id = (r.w != 0.0) ? HPS_ID_VALID : 0ULL;      // (PartConsts::valid_by_w: idcpu is not read)
        else r.id = pl.idcpu[ip];
        r.ion = k.can_ionize ? pl.ion_lev[ip] : 1;
        return r;
    };
    // NB particles per thread are fetched back to back (NB*7 loads in flight per lane) before any
    // of them is processed: the kernel is bound by memory-level parallelism, not by issue slots.
    // The first batch (the whole tile at nominal density) is requested before the accumulators are
    // zeroed, so its HBM latency hides behind the zeroing.
    // HPS_DEP_PIPE: the next batch is requested while the current one is worked on (the loads of a workgroup then overlap
    // its own LDS atomics, not only those of the other workgroups of the CU).  Measured with HPS_DEP_NB = 2 and 1 (76 VGPRs,
    // 6 waves per SIMD instead of 4): 71.0 / 74.8 us against 71.4 -- no gain: on a lattice sheet the kernel moves its 235 MB at
    // 3.9 TB/s (59.9 us, scripts/diag_deposit.py 0.0; a device copy reaches 5.4), behind the driver at 3.3
    constexpr int NB = HPS_DEP_NB;
    Rec rec[NB];
#if HPS_DEP_PIPE
    Rec nxt[NB];
#endif
    const int ipb = lrec.y + tid;
    if (ipb < pend) {
#pragma unroll
        for (int u = 0; u < NB; ++u) rec[u] = fetch(min(ipb + 256*u, pend - 1));
    }
    {
        double2* z = (double2*)acc;
        for (int s = tid; s < na*PL/2; s += 256) z[s] = make_double2(0.0, 0.0);
    }
    double* aimg = acc + na*PL;          // LASER: |a|^2 over the tile region
    if constexpr (LASER) { const int ca[1] = {k.aabs}; load_region<R, R, 256, 1>(aimg, f, ca, 1, ox, oy, tid); }
    __syncthreads();
    PT_STAMP(1);

    for (int ip0 = ipb; ip0 < pend; ip0 += 256*NB) {
#if HPS_DEP_PIPE
      if (ip0 != ipb) {
#pragma unroll
        for (int u = 0; u < NB; ++u) rec[u] = nxt[u];
      }
      if (ip0 + 256*NB < pend) {
#pragma unroll
        for (int u = 0; u < NB; ++u) nxt[u] = fetch(min(ip0 + 256*NB + 256*u, pend - 1));
      }
#else
      if (ip0 != ipb) {
#pragma unroll
        for (int u = 0; u < NB; ++u) rec[u] = fetch(min(ip0 + 256*u, pend - 1));
      }
#endif
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int ip = ip0 + 256*u;
        if (ip >= pend) break;
        const Rec cur = rec[u];
        const uint64_t id = cur.id;
        if (!(id & HPS_ID_VALID)) continue;
        if (k.can_ionize && cur.ion == 0) continue;      // a neutral atom deposits nothing (every term carries its level)
#if HPS_DEP_FAST_RCP
        // v_rcp_f64 + one Newton step (as the push: particle_math.h fast_rcp) instead of the IEEE division's eleven
        // instructions; psi = 0 must still give the inf the QSA test below drops the particle on (Newton would make it a NaN)
        const double psi_inv = cur.psi != 0.0 ? fast_rcp(cur.psi) : __builtin_huge_val();
#else
        const double psi_inv = 1.0/cur.psi;      // (IEEE: the QSA test below must see inf for psi = 0)
#endif
        const double vx_c = cur.ux*psi_inv;
        const double vy_c = cur.uy*psi_inv;
        double q_invvol = k.a*cur.w;
        double q_mu0_mass = k.b;
        if (k.can_ionize) { const double il = (double)cur.ion; q_invvol *= il; q_mu0_mass *= il; }
        double sx[ORDER + 1], sy[ORDER + 1];
        const int i0 = shape_weights<ORDER>((cur.x - k.xoff)*k.dx_inv, sx);
        const int j0 = shape_weights<ORDER>((cur.y - k.yoff)*k.dy_inv, sy);
        const int li = i0 - ox, lj = j0 - oy;
        const bool local = (li >= 0 && li + ORDER < R && lj >= 0 && lj + ORDER < R);
        double gamma_psi;
        if constexpr (LASER) {
            // |a|^2 with the deposition's own shape: from the LDS image, from the slab for a particle outside the halo
            double A = (local ? laser_gather_lds<ORDER>(aimg, R, li, lj, sx, sy)
                              : laser_gather<ORDER>(f, k.aabs, (cur.x - k.xoff)*k.dx_inv, (cur.y - k.yoff)*k.dy_inv))*k.laser_fac;
            if (k.can_ionize) A *= (double)cur.ion*(double)cur.ion;
            gamma_psi = 0.5*((1.0 + 0.5*A)*psi_inv*psi_inv + vx_c*vx_c*k.c_inv*k.c_inv + vy_c*vy_c*k.c_inv*k.c_inv + 1.0);
        } else {
            gamma_psi = 0.5*(psi_inv*psi_inv + vx_c*vx_c*k.c_inv*k.c_inv + vy_c*vy_c*k.c_inv*k.c_inv + 1.0);
        }
        if (gamma_psi < 0.0 || gamma_psi > k.max_qsa || psi_inv < 0.0) {
            if (n_qsa) atomicAdd(n_qsa, 1);
            pl.w[ip] = 0.0;
            pl.idcpu[ip] = (VBW ? pl.idcpu[ip] : id) & ~HPS_ID_VALID;
            if (pl.psi_half) pl.psi_half[ip] = 0.0;      // (Tiling::valid_by_psi)
            continue;
        }
        // per-component weights in DepComps order
        const double wv[6] = {vx_c, vy_c, (gamma_psi - 1.0)*k.c, gamma_psi, q_mu0_mass*psi_inv, 1.0};
        if (local) {
#pragma unroll
            for (int iy = 0; iy <= ORDER; ++iy) {
#pragma unroll
                for (int ix = 0; ix <= ORDER; ++ix) {
                    const double cd = q_invvol*sx[ix]*sy[iy];
                    double* p = acc + (lj + iy)*RP + li + ix;
#pragma unroll
                    for (int c = 0; c < 6; ++c) if (slot[c] >= 0) lds_add(p + slot[c]*PL, cd*wv[c]);
                }
            }
        } else {
            nfb += !tail;
#pragma unroll
            for (int iy = 0; iy <= ORDER; ++iy) {
#pragma unroll
                for (int ix = 0; ix <= ORDER; ++ix) {
                    const double cd = q_invvol*sx[ix]*sy[iy];
                    double* p = f.p + f.off(i0 + ix, j0 + iy);
#pragma unroll
                    for (int c = 0; c < 6; ++c) if (slot[c] >= 0) atomic_add_f64(p + gc[c]*f.ns, cd*wv[c]);
                }
            }
        }
      }
    }
    if (n_fallback && nfb) atomicAdd(n_fallback, nfb);
    PT_STAMP(2);
    __syncthreads();
    PT_STAMP(3);

    // flush the touched cells; halo cells are shared with neighbouring tiles -> atomics
    for (int s = tid; s < R*R; s += 256) {
        const int lj = s / R, li = s - lj*R;
        const int i = ox + li, j = oy + lj;
        if (i < -f.ng || i >= f.nx + f.ng || j < -f.ng || j >= f.ny + f.ng) continue;
        double* p = f.p + f.off(i, j);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            if (slot[c] >= 0) {
                const double v = acc[slot[c]*PL + lj*RP + li];
                if (v != 0.0) atomic_add_f64(p + gc[c]*f.ns, v);
            }
        }
    }
    PT_STAMP(4);
}

// image of `nc` slab components over the tile region into LDS (0 outside the slab box).  All loads of a thread are issued
// before its first LDS store (the loop over the thread's cells is unrolled by hand: its trip count depends on the thread, and
// left to the compiler every round of loads waited for the one before -- four trips to memory at the head of each tile)
template <int R, int RP, int NT, int NC>
__device__ __forceinline__ void load_region (double* img, const SlabView& f, const int* comps, int nc, int ox, int oy, int tid)
{
    constexpr int NIT = (R*R + NT - 1)/NT;
    double v[NIT][NC];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int s = min(tid + it*NT, R*R - 1);
        const int lj = s / R, li = s - lj*R;
        const int i = ox + li, j = oy + lj;
        const bool in = (i >= -f.ng && i < f.nx + f.ng && j >= -f.ng && j < f.ny + f.ng);
        const long o = in ? f.off(i, j) : 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) { v[it][c] = 0.0; if (c < nc) { const double q = f.p[comps[c]*f.ns + o]; v[it][c] = in ? q : 0.0; } }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int s = tid + it*NT;
        if (s < R*R) {
            const int lj = s / R, li = s - lj*R;
#pragma unroll
            for (int c = 0; c < NC; ++c) if (c < nc) img[c*RP*R + lj*RP + li] = v[it][c];
        }
    }
}

template <int ORDER, int DT, int TS, bool LASER = false, int PAD = 2, bool VBW = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))      // 128 VGPRs: 4 workgroups per CU
void k_explicit_tiled (SlabView f, hps_plasma pl, const int* __restrict__ offsets, int ntx,
                       int cBz, int cEz, int cExmBy, int cEypBx, int cSy, int cSx, PartConsts k, int* n_fallback,
                       const int* __restrict__ tile_flag, TailWork tw)
{
    constexpr int R = TS + 2*TILE_HALO;
    constexpr int RP = R + PAD, PL = RP*R;     // row pitch and plane size of the LDS images
    if (tile_flag && !tile_flag[offsets[gridDim.x + 2 + blockIdx.x]]) return;      // no charged ion in this tile (see k_deposit_tiled)
    constexpr int NS = ORDER + DT + 1;
    extern __shared__ __attribute__((aligned(16))) double lds[];     // [4 cached][R*R] + [2 accum][R*R] (+ |a|^2 with a laser)
    double* img = lds;
    double* acc = lds + 4*PL;
    double* aimg = lds + 6*PL;
    bool tail;
    const int4 lrec = tile_record(offsets, tw, pl.n, tail);     // {tile, first, end} of this workgroup (sort.hip)
    const int tile = lrec.x;
    const int ox = tail ? TAIL_ORIGIN : (tile % ntx)*TS - TILE_HALO, oy = tail ? TAIL_ORIGIN : (tile / ntx)*TS - TILE_HALO;
    const int tid = threadIdx.x;
    const int cc[4] = {cBz, cEz, cExmBy, cEypBx};
    // software pipeline over the tile's particles: the next particle's record is in flight while the
    // current one is deposited; the first one is requested ahead of the field-image load
    const int pend = lrec.z;
    struct Rec { double x, y, w, ux, uy, psi; uint64_t id; int ion; };
    auto fetch = [&] (int ip) {
        Rec r;
        r.x = pl.x[ip]; r.y = pl.y[ip]; r.w = pl.w[ip];
        r.ux = pl.ux[ip]; r.uy = pl.uy[ip]; r.psi = pl.psi[ip];
        if constexpr (VBW) r.id = (r.w != 0.0) ? HPS_ID_VALID : 0ULL;      // (PartConsts::valid_by_w: idcpu is not read)
        else r.id = pl.idcpu[ip];
        r.ion = k.can_ionize ? pl.ion_lev[ip] : 1;
        return r;
    };
    const int ipb = lrec.y + tid;
    Rec nxt{};
    if (ipb < pend) nxt = fetch(ipb);
    load_region<R, RP, 256, 4>(img, f, cc, 4, ox, oy, tid);
    if constexpr (LASER) { const int ca[1] = {k.aabs}; load_region<R, RP, 256, 1>(aimg, f, ca, 1, ox, oy, tid); }
    {
        double2* z = (double2*)acc;
        for (int s = tid; s < PL; s += 256) z[s] = make_double2(0.0, 0.0);
    }
    __syncthreads();

#if HPS_EXPL_TWO_PASS
    // Two passes over the workgroup's particles.  The first deposits every particle whose stencil lies inside the tile's image --
    // all but one in ten thousand -- through LDS, as straight-line code: the choice "image or slab" is made once per particle, not
    // at each of its 21 stencil cells (84 exec-mask branches per particle before; hoisting them was tried in round 2 and spilled,
    // because the scheduler then pulls all 36 field reads ahead -- here a compiler barrier per stencil row keeps one row of reads
    // in flight, as in the push).  The second pass, entered only by a workgroup that met such a particle (or a tail workgroup),
    // fetches the sheet again and deposits the others through the slab.  Per row the products with sy / dsy are formed once:
    //   Sy += sx (sy ty + a6 dsy) + (a5 sy) dsx     (14 instead of 17 fp64 operations per inner cell, 2 instead of 3 on the ring)
    int nfb = 0;
    bool slow = false;
    auto one = [&] (const Rec& cur, auto LC) __attribute__((always_inline)) -> bool {
        constexpr bool LOCAL = decltype(LC)::value;
        if (!(cur.id & HPS_ID_VALID)) return true;
        if (k.can_ionize && cur.ion == 0) return true;      // a neutral atom deposits nothing (every term carries its level)
        const double xmid = (cur.x - k.xoff)*k.dx_inv;
        const double ymid = (cur.y - k.yoff)*k.dy_inv;
        double sx[NS], dsx[NS], sy[NS], dsy[NS];
        int i0, j0;
        if constexpr (DT == 2) { i0 = centred_weights<ORDER>(xmid, sx, dsx); j0 = centred_weights<ORDER>(ymid, sy, dsy); }
        else                   { i0 = nodal_weights<ORDER>(xmid, sx, dsx);   j0 = nodal_weights<ORDER>(ymid, sy, dsy); }
        const int li = i0 - ox, lj = j0 - oy;
        const bool local = LASER ? (li >= 1 && li + NS + 1 <= R && lj >= 1 && lj + NS + 1 <= R)
                                 : (li >= 0 && li + NS <= R && lj >= 0 && lj + NS <= R);
        if (local != LOCAL) return false;
        const double psi_inv = fast_rcp(cur.psi);
        const double vx = cur.ux*psi_inv*k.c_inv;
        const double vy = cur.uy*psi_inv*k.c_inv;
        double q_invvol_mu0 = k.a, q_mass = k.b;
        if (k.can_ionize) { const double il = (double)cur.ion; q_invvol_mu0 *= il; q_mass *= il; }
        const double cdm = q_invvol_mu0*cur.w;
        double gp;
        if constexpr (LASER) {
            double lx[ORDER + 1], ly[ORDER + 1];
            const int ai = shape_weights<ORDER>(xmid, lx), aj = shape_weights<ORDER>(ymid, ly);
            const double A = (LOCAL ? laser_gather_lds<ORDER>(aimg, RP, ai - ox, aj - oy, lx, ly)
                                    : laser_gather<ORDER>(f, k.aabs, xmid, ymid))*k.laser_fac*q_mass*q_mass;
            gp = 0.5*((1.0 + 0.5*A)*psi_inv*psi_inv + vx*vx + vy*vy + 1.0);
        } else {
            gp = 0.5*(psi_inv*psi_inv + vx*vx + vy*vy + 1.0);
        }
        const double qp = q_mass*psi_inv;
        const double vxvy = vx*vy, gy = gp - vy*vy, gx = gp - vx*vx;
        const double cq = cdm*qp, cqc = cq*k.c_inv, cc = cdm*k.c;
        const double a1 = cq*vx, a2 = -cqc*vy, a3 = cqc*vxvy, a4 = -cqc*gy, a5 = cc*vxvy, a6 = -cc*(gy - 1.0);
        const double b1 = cq*vy, b2 = cqc*vx, b3 = cqc*gx, b4 = -cqc*vxvy, b5 = cc*(gx - 1.0), b6 = -cc*vxvy;
#pragma unroll
        for (int m = 0; m < NS; ++m) { dsx[m] *= k.dx_inv; dsy[m] *= k.dy_inv; }
#pragma unroll
        for (int iy = 0; iy < NS; ++iy) {
#if HPS_EXPL_ROW_BARRIER
            if constexpr (LOCAL) asm volatile("" ::: "memory");      // one stencil row of LDS reads in flight at a time
#endif
            const bool yedge = (DT == 2) && (iy == 0 || iy == NS - 1);
            const double a5s = a5*sy[iy], b5s = b5*sy[iy], ay = a6*dsy[iy], by = b6*dsy[iy];
#pragma unroll
            for (int ix = 0; ix < NS; ++ix) {
                const bool xedge = (DT == 2) && (ix == 0 || ix == NS - 1);
                if (xedge && yedge) continue;
                double* gp_ = nullptr; int ls = 0;
                if constexpr (LOCAL) ls = (lj + iy)*RP + li + ix;
                else                 gp_ = f.p + f.off(i0 + ix, j0 + iy);
                double sy_add, sx_add;
                if (xedge)      { sy_add = a5s*dsx[ix]; sx_add = b5s*dsx[ix]; }      // sx = 0: only dsx*sy survives
                else if (yedge) { sy_add = ay*sx[ix];   sx_add = by*sx[ix]; }        // sy = 0: only sx*dsy
                else {
                    double Bz, Ez, ExmBy, EypBx;
                    if constexpr (LOCAL) { Bz = lds_get(img + ls); Ez = lds_get(img + PL + ls); ExmBy = lds_get(img + 2*PL + ls); EypBx = lds_get(img + 3*PL + ls); }
                    else                 { Bz = gp_[cBz*f.ns]; Ez = gp_[cEz*f.ns]; ExmBy = gp_[cExmBy*f.ns]; EypBx = gp_[cEypBx*f.ns]; }
                    double ty = fma(a1, Bz, fma(a2, Ez, fma(a3, ExmBy, a4*EypBx)));
                    double tx = fma(b1, Bz, fma(b2, Ez, fma(b3, ExmBy, b4*EypBx)));
                    if constexpr (LASER) {
                        // gradient of |a|^2 at this stencil cell (ExplicitDeposition.cpp:211-226)
                        if (sx[ix]*sy[iy] != 0.0) {
                            const double lf = 0.25*cq*qp*k.laser_fac*k.c;
                            double ady, adx;
                            if constexpr (LOCAL) {
                                const double* a = aimg + ls;
                                ady = lds_get(a + RP) - lds_get(a - RP); adx = lds_get(a + 1) - lds_get(a - 1);
                            } else {
                                const double* a = f.p + k.aabs*f.ns + f.off(i0 + ix, j0 + iy);
                                ady = a[f.js] - a[-f.js]; adx = a[1] - a[-1];
                            }
                            ty = fma(lf*0.5*k.dy_inv, ady, ty);
                            tx = fma(-lf*0.5*k.dx_inv, adx, tx);
                        }
                    }
                    sy_add = fma(sx[ix], fma(sy[iy], ty, ay), a5s*dsx[ix]);
                    sx_add = fma(sx[ix], fma(sy[iy], tx, by), b5s*dsx[ix]);
                }
                if constexpr (LOCAL) { lds_add(acc + ls, sy_add); lds_add(acc + PL + ls, sx_add); }
                else                 { atomic_add_f64(gp_ + cSy*f.ns, sy_add); atomic_add_f64(gp_ + cSx*f.ns, sx_add); }
            }
        }
        return true;
    };
    for (int ip = ipb; ip < pend; ip += 256) {
        const Rec cur = nxt;
        if (ip + 256 < pend) nxt = fetch(ip + 256);
        if (!one(cur, std::true_type{})) { slow = true; nfb += !tail; }
    }
    if (slow) {
#pragma unroll 1
        for (int ip = ipb; ip < pend; ip += 256) { const Rec cur = fetch(ip); (void)one(cur, std::false_type{}); }
    }
#else
    int nfb = 0;
#ifdef HPS_DIAG_EXPL_NO_ATOMICS
    double diag_sink = 0.0;
#endif
    for (int ip = ipb; ip < pend; ip += 256) {
        const Rec cur = nxt;
        if (ip + 256 < pend) nxt = fetch(ip + 256);
        if (!(cur.id & HPS_ID_VALID)) continue;
        if (k.can_ionize && cur.ion == 0) continue;      // a neutral atom deposits nothing (every term carries its level)
        const double psi_inv = fast_rcp(cur.psi);
        const double vx = cur.ux*psi_inv*k.c_inv;
        const double vy = cur.uy*psi_inv*k.c_inv;
        double q_invvol_mu0 = k.a, q_mass = k.b;
        if (k.can_ionize) { const double il = (double)cur.ion; q_invvol_mu0 *= il; q_mass *= il; }
        const double cdm = q_invvol_mu0*cur.w;
        const double xmid = (cur.x - k.xoff)*k.dx_inv;
        const double ymid = (cur.y - k.yoff)*k.dy_inv;
        double sx[NS], dsx[NS], sy[NS], dsy[NS];
        int i0, j0;
        if constexpr (DT == 2) { i0 = centred_weights<ORDER>(xmid, sx, dsx); j0 = centred_weights<ORDER>(ymid, sy, dsy); }
        else                   { i0 = nodal_weights<ORDER>(xmid, sx, dsx);   j0 = nodal_weights<ORDER>(ymid, sy, dsy); }
        const int li = i0 - ox, lj = j0 - oy;
        // the whole stencil, and with a laser also the ring around its inner cells (the gradient of |a|^2), inside the image
        const bool local = LASER ? (li >= 1 && li + NS + 1 <= R && lj >= 1 && lj + NS + 1 <= R)
                                 : (li >= 0 && li + NS <= R && lj >= 0 && lj + NS <= R);
        double gp;
        if constexpr (LASER) {
            double lx[ORDER + 1], ly[ORDER + 1];
            const int ai = shape_weights<ORDER>(xmid, lx), aj = shape_weights<ORDER>(ymid, ly);
            const double A = (local ? laser_gather_lds<ORDER>(aimg, RP, ai - ox, aj - oy, lx, ly)
                                    : laser_gather<ORDER>(f, k.aabs, xmid, ymid))*k.laser_fac*q_mass*q_mass;
            gp = 0.5*((1.0 + 0.5*A)*psi_inv*psi_inv + vx*vx + vy*vy + 1.0);
        } else {
            gp = 0.5*(psi_inv*psi_inv + vx*vx + vy*vy + 1.0);
        }
        const double qp = q_mass*psi_inv;
        const double vxvy = vx*vy, gy = gp - vy*vy, gx = gp - vx*vx;
        // the source terms of ExplicitDeposition.cpp:225-252 are linear in the cached fields and in the
        // (derivative) shapes: collect the per-particle coefficients once,
        //   Sy += ss*(a1 Bz + a2 Ez + a3 ExmBy + a4 EypBx) + a5 dxs + a6 sdy,   Sx likewise with b1..b6,
        // 17 fp64 operations per stencil cell instead of 30
        const double cq = cdm*qp, cqc = cq*k.c_inv, cc = cdm*k.c;
        const double a1 = cq*vx, a2 = -cqc*vy, a3 = cqc*vxvy, a4 = -cqc*gy, a5 = cc*vxvy, a6 = -cc*(gy - 1.0);
        const double b1 = cq*vy, b2 = cqc*vx, b3 = cqc*gx, b4 = -cqc*vxvy, b5 = cc*(gx - 1.0), b6 = -cc*vxvy;
#pragma unroll
        for (int m = 0; m < NS; ++m) { dsx[m] *= k.dx_inv; dsy[m] *= k.dy_inv; }
        if (!local && !tail) ++nfb;
#pragma unroll
        for (int iy = 0; iy < NS; ++iy) {
#pragma unroll
            for (int ix = 0; ix < NS; ++ix) {
                // centred-derivative shapes (DT == 2): the plain shape vanishes on the outer ring of the stencil
                // (s[0] = s[NS-1] = 0 for every order), so there the cached fields do not enter at all -- only one of the
                // two derivative terms does: no LDS reads for 12 of the 21 cells (order 2)
                const bool xedge = (DT == 2) && (ix == 0 || ix == NS - 1), yedge = (DT == 2) && (iy == 0 || iy == NS - 1);
                if (xedge && yedge) continue;
                const bool ring = xedge || yedge;
                double* gp_ = nullptr; int ls = 0;
                if (local) ls = (lj + iy)*RP + li + ix;
                else       gp_ = f.p + f.off(i0 + ix, j0 + iy);
                double sy_add, sx_add;
                if (ring) {
                    // xedge: sx = 0 -> only dsx*sy survives; yedge: sy = 0 -> only sx*dsy
                    const double dd = xedge ? dsx[ix]*sy[iy] : sx[ix]*dsy[iy];
                    sy_add = (xedge ? a5 : a6)*dd;
                    sx_add = (xedge ? b5 : b6)*dd;
                } else {
                    double Bz, Ez, ExmBy, EypBx;
#ifdef HPS_DIAG_EXPL_NO_READS      // (diagnostic build: what the kernel costs without its field reads; results are wrong)
                    if (local) { Bz = 1.0 + ls; Ez = 2.0; ExmBy = 3.0; EypBx = 4.0; }
#else
                    if (local) { Bz = lds_get(img + ls); Ez = lds_get(img + PL + ls); ExmBy = lds_get(img + 2*PL + ls); EypBx = lds_get(img + 3*PL + ls); }
#endif
                    else       { Bz = gp_[cBz*f.ns]; Ez = gp_[cEz*f.ns]; ExmBy = gp_[cExmBy*f.ns]; EypBx = gp_[cEypBx*f.ns]; }
                    const double ss = sx[ix]*sy[iy];
                    const double dxs = dsx[ix]*sy[iy];
                    const double sdy = sx[ix]*dsy[iy];
                    double ty = fma(a1, Bz, fma(a2, Ez, fma(a3, ExmBy, a4*EypBx)));
                    double tx = fma(b1, Bz, fma(b2, Ez, fma(b3, ExmBy, b4*EypBx)));
                    if constexpr (LASER) {
                        // gradient of |a|^2 at this stencil cell (ExplicitDeposition.cpp:211-226), from the slab
                        if (ss != 0.0) {
                            const double lf = 0.25*cq*qp*k.laser_fac*k.c;
                            double ady, adx;
                            if (local) {
                                const double* a = aimg + ls;
                                ady = lds_get(a + RP) - lds_get(a - RP); adx = lds_get(a + 1) - lds_get(a - 1);
                            } else {
                                const double* a = f.p + k.aabs*f.ns + f.off(i0 + ix, j0 + iy);
                                ady = a[f.js] - a[-f.js]; adx = a[1] - a[-1];
                            }
                            ty = fma(lf*0.5*k.dy_inv, ady, ty);
                            tx = fma(-lf*0.5*k.dx_inv, adx, tx);
                        }
                    }
                    sy_add = fma(ss, ty, fma(a5, dxs, a6*sdy));
                    sx_add = fma(ss, tx, fma(b5, dxs, b6*sdy));
                }
#ifdef HPS_DIAG_EXPL_NO_ATOMICS    // (diagnostic build: without the LDS atomics; results are wrong)
                if (local) { diag_sink += sy_add + sx_add; }
#else
                if (local) { lds_add(acc + ls, sy_add); lds_add(acc + PL + ls, sx_add); }
#endif
                else       { atomic_add_f64(gp_ + cSy*f.ns, sy_add); atomic_add_f64(gp_ + cSx*f.ns, sx_add); }
            }
        }
    }
#endif
    if (n_fallback && nfb) atomicAdd(n_fallback, nfb);
#ifdef HPS_DIAG_EXPL_NO_ATOMICS
    if (diag_sink == 1.2345e-300 && n_fallback) atomicAdd(n_fallback, 1);
#endif
    __syncthreads();
#ifdef HPS_DIAG_EXPL_NO_FLUSH
    if (acc[tid] == 1.2345e-300 && n_fallback) atomicAdd(n_fallback, 1);
    return;
#endif
    for (int s = tid; s < R*R; s += 256) {
        const int lj = s / R, li = s - lj*R;
        const int i = ox + li, j = oy + lj;
        if (i < -f.ng || i >= f.nx + f.ng || j < -f.ng || j >= f.ny + f.ng) continue;
        double* p = f.p + f.off(i, j);
        const double a = acc[lj*RP + li], b = acc[PL + lj*RP + li];
        if (a != 0.0) atomic_add_f64(p + cSy*f.ns, a);
        if (b != 0.0) atomic_add_f64(p + cSx*f.ns, b);
    }
}

// array element at a 32-bit byte offset from a uniform base: global_load / global_store with an SGPR base and one offset VGPR
template <class T> __device__ __forceinline__ T ldo (const T* base, unsigned o) { return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + o); }
template <class T> __device__ __forceinline__ void sto (T* base, unsigned o, T v) { *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + o) = v; }
// the same with the non-temporal hint (global_load / global_store ... nt): for arrays only this kernel touches -- the
// half-step momenta -- so that they do not push what the next deposition reads (x, y, w, ux, uy, psi) out of the caches
#ifndef HPS_PUSH_NT
#define HPS_PUSH_NT 1
#endif
template <class T> __device__ __forceinline__ T ldo_nt (const T* base, unsigned o)
{
#if HPS_PUSH_NT
    return __builtin_nontemporal_load(reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + o));
#else
    return ldo(base, o);
#endif
}
template <class T> __device__ __forceinline__ void sto_nt (T* base, unsigned o, T v)
{
#if HPS_PUSH_NT
    __builtin_nontemporal_store(v, reinterpret_cast<T*>(reinterpret_cast<char*>(base) + o));
#else
    sto(base, o, v);
#endif
}

// IONIZE: the species can be field-ionised (ADK, ionization.hip).  The decision needs exactly the fields the push gathers,
// so it is taken here, between the gather and the push (the reference ionises, then pushes: Hipace.cpp:693-701): the ion's
// level goes up, its electron is appended to the product species, and the push runs with the new charge.  A neutral
// atom at rest is not pushed at all (zero charge: the push would leave every quantity as it is).
// the six sub-steps of a particle through taylor2_substep_pre (particle_math.h; 0: taylor2_substep, the dual-number form)
#ifndef HPS_PUSH_ALGEBRA
#define HPS_PUSH_ALGEBRA 1
#endif
#ifndef HPS_PUSH_GATHER_PIPE
#define HPS_PUSH_GATHER_PIPE 0
#endif
#ifndef HPS_PUSH_SPLIT_GATHER
#define HPS_PUSH_SPLIT_GATHER 1
#endif
#ifndef HPS_PUSH_W3
#define HPS_PUSH_W3 1
#endif
#ifndef HPS_PUSH_WAVES
#define HPS_PUSH_WAVES 3
#endif
#ifndef HPS_PUSH_WAVES_ION
#define HPS_PUSH_WAVES_ION 2
#endif
#ifndef HPS_PUSH_PF_LASER
#define HPS_PUSH_PF_LASER 1
#endif
// VBP: the engine's own electron sheet -- a particle is valid iff its psi_half is not 0 (Tiling::valid_by_psi: every path that
// clears the valid bit of idcpu also zeroes psi_half, which only the push reads): idcpu is not read, 40 instead of 48 B in per
// particle in the one particle kernel that is bound by its bytes.
// DEP != 0 (mask of DepComps, 51 or 59): the workgroup goes on to DEPOSIT its tile's pushed particles into the next slice's
// jx jy chi rhomjz [rho] (which the engine has shifted / zeroed ahead of the launch) -- PlasmaDepositCurrent.cpp:155-246 with the
// arithmetic of k_deposit_tiled -- once all of them are pushed: the accumulators take the place of the field image in LDS (the
// push's occupancy is unchanged: round 2's fused kernel held both, 56 KB, two workgroups per CU), every thread reads back the
// six values it has just stored itself (from the L2, not from HBM: the 48 B per particle the stand-alone deposition of the
// next slice would fetch), and the next slice needs no deposition launch.  Static beam, no laser, no ionisable species.
struct DepTail { DepComps cm; double a, b, max_qsa; int* n_qsa; };
template <int ORDER, int TS, bool LASER = false, bool IONIZE = false, bool VBP = false, int DEP = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(IONIZE ? HPS_PUSH_WAVES_ION : HPS_PUSH_WAVES)))
void k_advance_tiled (SlabView f, hps_plasma pl, const int* __restrict__ offsets, int ntx,
                      int cPsi, int cEz, int cBx, int cBy, int cBz, PartConsts k, int* n_fallback, IonArgs ia, const int* go, TailWork tw, MgPost mp,
                      DepTail dt)
{
    static_assert(DEP == 0 || (!LASER && !IONIZE), "the deposition tail exists for the plain push only");
    constexpr int R = TS + 2*TILE_HALO;
    constexpr int NS = ORDER + 2;
    // enqueued behind a multigrid solve whose norms the host has not seen yet: run only if that solve is over (*go == 1,
    // set by the solve's k_post_norms; else the host adds V-cycles and launches the push again).  The word is loaded
    // here and looked at behind the barrier that waits for the field image anyway.
    static_assert(!(VBP && IONIZE), "the ADK draw is keyed by the id");
    // mp.src: this launch stands where k_post_norms stood, directly behind the multigrid's V-cycles (mg_defer_post): every
    // workgroup evaluates the stopping rule for its own gate, workgroup 0 posts the norms to the host first thing -- the host has
    // the length of this kernel to read them and enqueue the next slice -- and a 4.8 us launch is off the slice's chain
    int go_now;
    if (mp.src) {
        go_now = vcycle_active(mp.after) ? 0 : 1;      // (all lanes active: the kernel's first statement)
        if (blockIdx.x == 0) {
            for (int w = threadIdx.x; w < mp.nwords; w += 256) mp.dst[w] = mp.src[w];
            HPS_HOST_STORES_ACKNOWLEDGED();
            __syncthreads();
            if (threadIdx.x == 0) *mp.seq_slot = mp.seq;
        }
    } else go_now = go ? *go : 1;
    int charged = 0;          // IONIZE: does this thread hold an ion of level > 0 after the slice?
    extern __shared__ __attribute__((aligned(16))) double img[];     // [5][R*R]
    bool tail;
    const int4 lrec = tile_record(offsets, tw, pl.n, tail);     // {tile, first, end} of this workgroup (sort.hip)
    const int tile = lrec.x;
    if constexpr (IONIZE) {
        if (!go_now) return;      // (gated: nothing is posted, the launch is repeated ungated)
        // a tile of neutral atoms at rest (no charged ion so far) in a field below the threshold of the first level:
        // nothing to decide, nothing to push -- most tiles of a slice (the field image is not even loaded)
        if (ia.fbound && ia.tile_flag && ia.tile_flag[tile] == 0 && adk_tile_below_threshold(ia, tile % ntx, tile / ntx)) {
            adk_post(ia);
            return;
        }
    }
    const int ox = tail ? TAIL_ORIGIN : (tile % ntx)*TS - TILE_HALO, oy = tail ? TAIL_ORIGIN : (tile / ntx)*TS - TILE_HALO;
    const int tid = threadIdx.x;
    const unsigned pend = (unsigned)lrec.z;
    struct PIn { uint64_t id; double xp, yp, uxh, uyh, psih; };
    auto fetch = [&] (unsigned ip) {
        PIn q;
        const unsigned o = ip*8u;
        if constexpr (!VBP) q.id = ldo(pl.idcpu, o);
        q.xp = ldo(pl.x_prev, o); q.yp = ldo(pl.y_prev, o);
        q.uxh = ldo_nt(pl.ux_half, o); q.uyh = ldo_nt(pl.uy_half, o); q.psih = ldo_nt(pl.psi_half, o);
        if constexpr (VBP) q.id = (q.psih != 0.0) ? HPS_ID_VALID : 0ULL;
        return q;
    };
    // the thread's first particle is requested ahead of the field image: its six values arrive with the image's
    // (HPS_PUSH_PF_LASER=0: the laser variant without the prefetch -- 12 instead of 20 B of scratch per lane under the
    // 168-register cap, and slower: config 5 946 against 968 slices/s)
#ifndef HPS_PUSH_NO_PREFETCH
    constexpr bool PF = HPS_PUSH_PF_LASER || !LASER;
#else
    constexpr bool PF = false;
#endif
    unsigned ip = (unsigned)lrec.y + tid;
    PIn nxt{0, 0.0, 0.0, 0.0, 0.0, 1.0};
    if constexpr (PF) { if (ip < pend) nxt = fetch(ip); }
    const int cc[5] = {cPsi, cEz, cBx, cBy, cBz};
    load_region<R, R, 256, 5>(img, f, cc, 5, ox, oy, tid);
    double* aimg = img + 5*R*R;           // LASER: |a|^2 over the tile region
    if constexpr (LASER) { const int ca[1] = {k.aabs}; load_region<R, R, 256, 1>(aimg, f, ca, 1, ox, oy, tid); }
    __syncthreads();
    if (!go_now) return;
    int nfb = 0;

    // The thread's particles are ip = first + 256 m.  Per particle the kernel needs idcpu, x_prev, y_prev and the three
    // half-step momenta: read one after the other where they are used (the validity bit first), that was three dependent
    // trips to memory per particle against ~1 us of arithmetic, with three waves per SIMD to hide them: the waves of
    // the round-2 kernel lived 29 us for 4-5 particles and issued VALU instructions a fifth of that time.  Now all six
    // values of particle m + 1 are requested before particle m is worked on (14 VGPRs), and every array is addressed
    // as uniform base + one 32-bit byte offset (the file is compiled with -disable-lsr: the loop-strength-reduction
    // pass otherwise keeps a 64-bit pointer per array and iteration in VGPRs, 24 registers here).
    for (; ip < pend; ip += 256) {
        __builtin_assume(ip < (1u << 28));
        PIn cur;
        if constexpr (PF) { cur = nxt; if (ip + 256 < pend) nxt = fetch(ip + 256); }
        else cur = fetch(ip);               // one batch of six loads, one trip to memory per particle
        const unsigned o8 = ip*8u;
        const uint64_t id = cur.id;
        if (!(id & HPS_ID_VALID)) continue;
        double qmc = k.a;
        if (k.can_ionize) qmc *= (double)pl.ion_lev[ip];
        bool dead = false;
        for (int isc = 0; isc < k.n_subcycles && !dead; ++isc) {
            // (sub-cycles after the first re-read what the one before has committed, as the reference does)
            double xp = isc == 0 ? cur.xp : ldo(pl.x_prev, o8);
            double yp = isc == 0 ? cur.yp : ldo(pl.y_prev, o8);
            double sx[NS], dsx[NS], sy[NS], dsy[NS];
            constexpr bool W3 = HPS_PUSH_W3 && ORDER == 2 && HPS_PUSH_SPLIT_GATHER && !HPS_PUSH_GATHER_PIPE;
            double pxw[3] = {0.0, 0.0, 0.0}, pyw[3] = {0.0, 0.0, 0.0}; bool xhw = false, yhw = false;
            int i0, j0;
            if constexpr (W3) {      // the three plain weights and their offset straight from the polynomial (common.h)
                i0 = nodal_weights2_w3((xp - k.xoff)*k.dx_inv, sx, dsx, pxw, xhw);
                j0 = nodal_weights2_w3((yp - k.yoff)*k.dy_inv, sy, dsy, pyw, yhw);
            } else {
                i0 = nodal_weights<ORDER>((xp - k.xoff)*k.dx_inv, sx, dsx);
                j0 = nodal_weights<ORDER>((yp - k.yoff)*k.dy_inv, sy, dsy);
            }
            const int li = i0 - ox, lj = j0 - oy;
            const bool local = (li >= 0 && li + NS <= R && lj >= 0 && lj + NS <= R);
            if (!local && !tail) ++nfb;
            Fld F{0, 0, 0, 0, 0, 0};
            if (local) {
                // tensor-product gather: x sums per stencil row, then one y weight per row and component
                // (reading only the (NS-1) x (NS-1) cells on which the plain weights of Ez, Bx, By, Bz are non-zero -- one
                // of s[0], s[NS-1] is always exactly 0 -- was measured: 52 instead of 80 LDS reads per particle, but the
                // lane-dependent base address costs more than the reads save: 171 against 166 us)
                const double* b = img + lj*R + li;
#if HPS_PUSH_SPLIT_GATHER
                // Psi needs all NS x NS cells (its derivative weights are full); Ez, Bx, By, Bz only the (NS-1) x (NS-1) cells on
                // which the plain weights live: one of s[0], s[NS-1] is exactly 0, which one depends on the particle's half of
                // its cell -- 16 + 4 x 9 = 52 LDS reads per particle instead of 80 (order 2), 416 instead of 640 B through the LDS
                // pipe, 88 instead of 120 FMAs; the skipped terms are exact zeros.  (Round 3 measured this slower, 171 against
                // 166 us, when the kernel still waited for three dependent trips to memory per particle.)
#if HPS_PUSH_GATHER_PIPE
                // the LDS reads of a row are requested one step ahead of the arithmetic that uses the previous ones: Psi's NS x NS
                // cells and the first row of the plain fields are in flight together, then row k + 1 while row k is summed.
                // Measured (round 4, call 18): 20 registers spilled under the 168 cap (11 without the particle prefetch):
                // 168.6 / 135 against 131.5 us -- the kernel has no registers left for more reads in flight.  Off.
                const bool xhi = !(sx[NS - 1] == 0.0), yhi = !(sy[NS - 1] == 0.0);
                const double* bq = b + R*R + (yhi ? R : 0) + (xhi ? 1 : 0);
                double vp[NS][NS];
#pragma unroll
                for (int iy = 0; iy < NS; ++iy)
#pragma unroll
                    for (int ix = 0; ix < NS; ++ix) vp[iy][ix] = lds_get(b + iy*R + ix);
                double vq[2][4][NS - 1];
                auto req = [&] (int ky, int slot) __attribute__((always_inline)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int kx = 0; kx < NS - 1; ++kx) vq[slot][c][kx] = lds_get(bq + c*R*R + ky*R + kx);
                };
                req(0, 0);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int iy = 0; iy < NS; ++iy) {
                    double rp = 0.0, rd = 0.0;
#pragma unroll
                    for (int ix = 0; ix < NS; ++ix) { rp = fma(sx[ix], vp[iy][ix], rp); rd = fma(dsx[ix], vp[iy][ix], rd); }
                    F.ExmBy = fma(sy[iy], rd, F.ExmBy);
                    F.EypBx = fma(dsy[iy], rp, F.EypBx);
                }
                double px[NS - 1], py[NS - 1];
#pragma unroll
                for (int m = 0; m < NS - 1; ++m) { px[m] = xhi ? sx[m + 1] : sx[m]; py[m] = yhi ? sy[m + 1] : sy[m]; }
#pragma unroll
                for (int ky = 0; ky < NS - 1; ++ky) {
                    if (ky + 1 < NS - 1) req(ky + 1, (ky + 1) & 1);
                    asm volatile("" ::: "memory");
                    double rez = 0.0, rbx = 0.0, rby = 0.0, rbz = 0.0;
#pragma unroll
                    for (int kx = 0; kx < NS - 1; ++kx) {
                        rez = fma(px[kx], vq[ky & 1][0][kx], rez);
                        rbx = fma(px[kx], vq[ky & 1][1][kx], rbx);
                        rby = fma(px[kx], vq[ky & 1][2][kx], rby);
                        rbz = fma(px[kx], vq[ky & 1][3][kx], rbz);
                    }
                    F.Ez  = fma(py[ky], rez, F.Ez);
                    F.Bxc = fma(py[ky], rbx, F.Bxc);
                    F.Byc = fma(py[ky], rby, F.Byc);
                    F.Bz  = fma(py[ky], rbz, F.Bz);
                }
#else
#pragma unroll
                for (int iy = 0; iy < NS; ++iy) {
#ifdef HPS_PUSH_PSI_BARRIER
                    asm volatile("" ::: "memory");
#endif
                    double rp = 0.0, rd = 0.0;
#pragma unroll
                    for (int ix = 0; ix < NS; ++ix) {
                        const double psi_c = lds_get(b + iy*R + ix);
                        rp = fma(sx[ix], psi_c, rp);
                        rd = fma(dsx[ix], psi_c, rd);
                    }
                    F.ExmBy = fma(sy[iy], rd, F.ExmBy);
                    F.EypBx = fma(dsy[iy], rp, F.EypBx);
                }
                const bool xhi = W3 ? xhw : !(sx[NS - 1] == 0.0), yhi = W3 ? yhw : !(sy[NS - 1] == 0.0);
                double px[NS - 1], py[NS - 1];
#pragma unroll
                for (int m = 0; m < NS - 1; ++m) {
                    if constexpr (W3) { px[m] = pxw[m < 3 ? m : 2]; py[m] = pyw[m < 3 ? m : 2]; }
                    else { px[m] = xhi ? sx[m + 1] : sx[m]; py[m] = yhi ? sy[m + 1] : sy[m]; }
                }
                const double* bq = b + R*R + (yhi ? R : 0) + (xhi ? 1 : 0);
#pragma unroll
                for (int ky = 0; ky < NS - 1; ++ky) {
                    asm volatile("" ::: "memory");
                    double rez = 0.0, rbx = 0.0, rby = 0.0, rbz = 0.0;
#pragma unroll
                    for (int kx = 0; kx < NS - 1; ++kx) {
                        const int ls = ky*R + kx;
                        rez = fma(px[kx], lds_get(bq + ls), rez);
                        rbx = fma(px[kx], lds_get(bq + R*R + ls), rbx);
                        rby = fma(px[kx], lds_get(bq + 2*R*R + ls), rby);
                        rbz = fma(px[kx], lds_get(bq + 3*R*R + ls), rbz);
                    }
                    F.Ez  = fma(py[ky], rez, F.Ez);
                    F.Bxc = fma(py[ky], rbx, F.Bxc);
                    F.Byc = fma(py[ky], rby, F.Byc);
                    F.Bz  = fma(py[ky], rbz, F.Bz);
                }
#endif
#else
#ifndef HPS_PUSH_ROLLED_ROWS
#pragma unroll
#else
#pragma unroll 1
#endif
                for (int iy = 0; iy < NS; ++iy) {
#ifndef HPS_PUSH_ROLLED_ROWS
                    asm volatile("" ::: "memory");      // one stencil row of LDS reads in flight at a time
#endif
                    double rp = 0.0, rd = 0.0, rez = 0.0, rbx = 0.0, rby = 0.0, rbz = 0.0;
#pragma unroll
                    for (int ix = 0; ix < NS; ++ix) {
                        const int ls = iy*R + ix;
                        const double psi_c = lds_get(b + ls);
                        rp = fma(sx[ix], psi_c, rp);
                        rd = fma(dsx[ix], psi_c, rd);
                        rez = fma(sx[ix], lds_get(b + R*R + ls), rez);
                        rbx = fma(sx[ix], lds_get(b + 2*R*R + ls), rbx);
                        rby = fma(sx[ix], lds_get(b + 3*R*R + ls), rby);
                        rbz = fma(sx[ix], lds_get(b + 4*R*R + ls), rbz);
                    }
                    F.ExmBy = fma(sy[iy], rd, F.ExmBy);
                    F.EypBx = fma(dsy[iy], rp, F.EypBx);
                    F.Ez  = fma(sy[iy], rez, F.Ez);
                    F.Bxc = fma(sy[iy], rbx, F.Bxc);
                    F.Byc = fma(sy[iy], rby, F.Byc);
                    F.Bz  = fma(sy[iy], rbz, F.Bz);
                }
#endif
                F.ExmBy *= k.dx_inv;
                F.EypBx *= k.dy_inv;
            } else {
#ifdef HPS_DIAG_PUSH_NO_FALLBACK      // (diagnostic build: what the kernel's registers are without the slab path; results are wrong)
                F.Ez = 1.0;
#else
#pragma unroll 1
                for (int iy = 0; iy < NS; ++iy) {
#pragma unroll
                    for (int ix = 0; ix < NS; ++ix) {
                        const double* p = f.p + f.off(i0 + ix, j0 + iy);
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
#endif
            }
            F.Bxc *= k.c;
            F.Byc *= k.c;
            if constexpr (IONIZE) {
                if (isc == 0) {
                    int lev = pl.ion_lev[ip];
                    const double uxh = cur.uxh, uyh = cur.uyh;
                    // an ion that has lost all Z electrons cannot ionise (the reference reads past the end of its tables)
                    const bool ionize = lev < ia.Z && adk_decide(ia, F.ExmBy + F.Byc, F.EypBx - F.Bxc, F.Ez, uxh, uyh, cur.psih, lev, id);
                    if (ionize) { ++lev; pl.ion_lev[ip] = lev; qmc = k.a*(double)lev; }
                    adk_emit(ia, ionize, pl.x[ip], pl.y[ip], xp, yp, pl.w[ip]);
                    charged |= (lev > 0);
                    if (lev == 0 && uxh == 0.0 && uyh == 0.0) break;
                }
            }
            LaserFld Lf{0.0, 0.0, 0.0};
            if constexpr (LASER) {
                // |a|^2 and its centred gradient with the plain shape (PlasmaParticleAdvance.cpp:121-131): from the LDS image
                // when the stencil and its ring lie inside it, from the slab otherwise
                double lx[ORDER + 1], ly[ORDER + 1];
                const int ai = shape_weights<ORDER>((xp - k.xoff)*k.dx_inv, lx) - ox, aj = shape_weights<ORDER>((yp - k.yoff)*k.dy_inv, ly) - oy;
                if (ai >= 1 && ai + ORDER + 1 < R && aj >= 1 && aj + ORDER + 1 < R)
                    laser_gather_grad_lds<ORDER>(aimg, R, ai, aj, lx, ly, k.dx_inv, k.dy_inv, Lf.A, Lf.ADx, Lf.ADy);
                else
                    laser_gather_grad<ORDER>(f, k.aabs, (xp - k.xoff)*k.dx_inv, (yp - k.yoff)*k.dy_inv, k.dx_inv, k.dy_inv, Lf.A, Lf.ADx, Lf.ADy);
                const double ln = k.laser_fac*(k.can_ionize ? (double)pl.ion_lev[ip]*(double)pl.ion_lev[ip] : 1.0);
                Lf.A *= 0.5*ln; Lf.ADx *= 0.25*k.c*ln; Lf.ADy *= 0.25*k.c*ln;
            }
            const double dz = k.dz, sdz = dz*0.25;
            double ux = isc == 0 ? cur.uxh : ldo(pl.ux_half, o8), uy = isc == 0 ? cur.uyh : ldo(pl.uy_half, o8), psi = isc == 0 ? cur.psih : ldo(pl.psi_half, o8);
#if HPS_PUSH_ALGEBRA
            const PushForce PFc = push_force(F, Lf, k.c_inv, qmc);
            const double h2 = 0.5*sdz*sdz;
#pragma unroll 1
            for (int s = 0; s < 4; ++s) taylor2_substep_pre<LASER>(ux, uy, psi, PFc, sdz, h2);
#else
            if constexpr (LASER) {
#pragma unroll 1
                for (int s = 0; s < 4; ++s) taylor2_substep_laser(ux, uy, psi, F, Lf, k.c_inv, qmc, sdz);
            } else {
#pragma unroll 1
                for (int s = 0; s < 4; ++s) taylor2_substep(ux, uy, psi, F, k.c_inv, qmc, sdz);
            }
#endif
            const double pinv = fast_rcp(psi);
            xp += dz*k.c_inv*(ux*pinv);
            yp += dz*k.c_inv*(uy*pinv);
            if (apply_particle_bc(k, xp, yp, ux, uy)) {
                sto(pl.w, o8, 0.0);
                sto(pl.idcpu, o8, (uint64_t)((VBP ? ldo(pl.idcpu, o8) : id) & ~HPS_ID_VALID));
                sto(pl.psi_half, o8, 0.0);      // (Tiling::valid_by_psi)
                dead = true;
                break;
            }
            sto(pl.x, o8, xp); sto(pl.y, o8, yp);
            if (!k.temp_slice) {
                sto_nt(pl.ux_half, o8, ux); sto_nt(pl.uy_half, o8, uy); sto_nt(pl.psi_half, o8, psi);
                if (pl.x_prev != pl.x) sto(pl.x_prev, o8, xp);      // (aliased by the engine: already stored)
                if (pl.y_prev != pl.y) sto(pl.y_prev, o8, yp);
            }
#if HPS_PUSH_ALGEBRA
#pragma unroll 1
            for (int s = 0; s < 2; ++s) taylor2_substep_pre<LASER>(ux, uy, psi, PFc, sdz, h2);
#else
            if constexpr (LASER) {
#pragma unroll 1
                for (int s = 0; s < 2; ++s) taylor2_substep_laser(ux, uy, psi, F, Lf, k.c_inv, qmc, sdz);
            } else {
#pragma unroll 1
                for (int s = 0; s < 2; ++s) taylor2_substep(ux, uy, psi, F, k.c_inv, qmc, sdz);
            }
#endif
            sto(pl.ux, o8, ux); sto(pl.uy, o8, uy); sto(pl.psi, o8, psi);
        }
    }
    if constexpr (DEP != 0) {
        // ---- DepositCurrent of the tile's pushed particles into the next slice ----
        constexpr int RR = R*R;
        const int gc[6] = {dt.cm.jx, dt.cm.jy, dt.cm.jz, dt.cm.rho, dt.cm.chi, dt.cm.rhomjz};
        int slot[6]; int na = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) { const bool on = (DEP >> c) & 1; slot[c] = on ? na++ : -1; }
        __syncthreads();                                  // every wave is done with the field image
        double* acc = img;                                // [na][R*R], na <= 5
        {   double2* z = (double2*)acc;
            for (int s2 = tid; s2 < na*RR/2; s2 += 256) z[s2] = make_double2(0.0, 0.0); }
        __syncthreads();
        for (unsigned iq = (unsigned)lrec.y + tid; iq < pend; iq += 256) {
            const unsigned o = iq*8u;
            // (agent-scope loads: past the CU's vector cache, which may still hold the lines as they were before this thread's stores)
            auto ldc2 = [] (const double* base, unsigned off) {
                return __hip_atomic_load(reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            const double w = ldc2(pl.w, o);
            const double x = ldc2(pl.x, o), y = ldc2(pl.y, o), ux = ldc2(pl.ux, o), uy = ldc2(pl.uy, o), psi = ldc2(pl.psi, o);
            if (w == 0.0) continue;                       // (every invalidation zeroes the weight: PartConsts::valid_by_w)
            const double psi_inv = 1.0/psi;               // (IEEE: the QSA test must see inf for psi = 0)
            const double vx_c = ux*psi_inv, vy_c = uy*psi_inv;
            const double q_invvol = dt.a*w;
            const double gamma_psi = 0.5*(psi_inv*psi_inv + vx_c*vx_c*k.c_inv*k.c_inv + vy_c*vy_c*k.c_inv*k.c_inv + 1.0);
            if (gamma_psi < 0.0 || gamma_psi > dt.max_qsa || psi_inv < 0.0) {
                if (dt.n_qsa) atomicAdd(dt.n_qsa, 1);
                sto(pl.w, o, 0.0);
                sto(pl.idcpu, o, (uint64_t)(ldo(pl.idcpu, o) & ~HPS_ID_VALID));
                sto(pl.psi_half, o, 0.0);
                continue;
            }
            double wx[ORDER + 1], wy[ORDER + 1];
            const int i0 = shape_weights<ORDER>((x - k.xoff)*k.dx_inv, wx);
            const int j0 = shape_weights<ORDER>((y - k.yoff)*k.dy_inv, wy);
            const double wv[6] = {vx_c, vy_c, (gamma_psi - 1.0)*k.c, gamma_psi, dt.b*psi_inv, 1.0};
            const int li = i0 - ox, lj = j0 - oy;
            if (li >= 0 && li + ORDER < R && lj >= 0 && lj + ORDER < R) {
#pragma unroll
                for (int iy = 0; iy <= ORDER; ++iy) {
#pragma unroll
                    for (int ix = 0; ix <= ORDER; ++ix) {
                        const double cd = q_invvol*wx[ix]*wy[iy];
                        double* p = acc + (lj + iy)*R + li + ix;
#pragma unroll
                        for (int c = 0; c < 6; ++c) if (slot[c] >= 0) lds_add(p + slot[c]*RR, cd*wv[c]);
                    }
                }
            } else {
                nfb += !tail;
#pragma unroll
                for (int iy = 0; iy <= ORDER; ++iy) {
#pragma unroll
                    for (int ix = 0; ix <= ORDER; ++ix) {
                        const double cd = q_invvol*wx[ix]*wy[iy];
                        double* p = f.p + f.off(i0 + ix, j0 + iy);
#pragma unroll
                        for (int c = 0; c < 6; ++c) if (slot[c] >= 0) atomic_add_f64(p + gc[c]*f.ns, cd*wv[c]);
                    }
                }
            }
        }
        __syncthreads();
        for (int s2 = tid; s2 < RR; s2 += 256) {
            const int lj = s2 / R, li = s2 - lj*R;
            const int i = ox + li, j = oy + lj;
            if (i < -f.ng || i >= f.nx + f.ng || j < -f.ng || j >= f.ny + f.ng) continue;
            double* p = f.p + f.off(i, j);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                if (slot[c] >= 0) {
                    const double v = acc[slot[c]*RR + s2];
                    if (v != 0.0) atomic_add_f64(p + gc[c]*f.ns, v);
                }
            }
        }
    }
    if (n_fallback && nfb) atomicAdd(n_fallback, nfb);
    if constexpr (IONIZE) {
        // tiles without a charged ion are skipped by the species' deposition kernels on the next slice
        const int any = __syncthreads_or(charged);
        if (ia.tile_flag && tid == 0) ia.tile_flag[tile] = any;
        adk_post(ia);
    }
}

// Gather + push of slice k and the current deposition of slice k-1 in ONE pass over the sheet ("fused lower
// bound": the deposition re-reads 56 B per particle that the push has just had in registers).  The workgroup keeps the
// field image of its tile (as k_advance_tiled) AND the deposition accumulators (as k_deposit_tiled) in LDS; a particle
// is pushed (PlasmaParticleAdvance.cpp:92-217) and its new state deposited at once into the components of the NEXT
// slice (PlasmaDepositCurrent.cpp:155-246), which the engine has shifted / zeroed before the launch.  The QSA check of
// the deposition invalidates the particle here, exactly as the stand-alone deposition at the start of the next slice
// would.  No laser, no ionisable species (those keep the two kernels).
template <int ORDER, int TS, int MASK, int NT>
__global__ __launch_bounds__(NT)
void k_advance_deposit_tiled (SlabView f, hps_plasma pl, const int* __restrict__ offsets, int ntx,
                              int cPsi, int cEz, int cBx, int cBy, int cBz, PartConsts k, DepComps cm, PartConsts kd,
                              int* n_qsa, int* n_fallback)
{
    constexpr int R = TS + 2*TILE_HALO;
    constexpr int NS = ORDER + 2;
    extern __shared__ __attribute__((aligned(16))) double img[];     // [5][R*R] field image, then [active comps][R*R] accumulators
    double* acc = img + 5*R*R;
    const int gc[6] = {cm.jx, cm.jy, cm.jz, cm.rho, cm.chi, cm.rhomjz};
    int slot[6]; int na = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) { const bool on = (MASK >> c) & 1; slot[c] = on ? na++ : -1; }
    const int tile = offsets[gridDim.x + 2 + blockIdx.x];     // launch order: heaviest tile first (sort.hip)
    const int ox = (tile % ntx)*TS - TILE_HALO, oy = (tile / ntx)*TS - TILE_HALO;
    const int tid = threadIdx.x;
    const int cc[5] = {cPsi, cEz, cBx, cBy, cBz};
    load_region<R, R, NT, 5>(img, f, cc, 5, ox, oy, tid);
    {
        double2* z = (double2*)acc;
        for (int s = tid; s < na*R*R/2; s += NT) z[s] = make_double2(0.0, 0.0);
    }
    __syncthreads();

    const int pend = offsets[tile + 1];
    int nfb = 0;
    for (int ip = offsets[tile] + tid; ip < pend; ip += NT) {
        const uint64_t id = pl.idcpu[ip];
        if (!(id & HPS_ID_VALID)) continue;
        const double qmc = k.a;
        bool dead = false;
        double xp = 0.0, yp = 0.0, ux = 0.0, uy = 0.0, psi = 1.0;
        for (int isc = 0; isc < k.n_subcycles && !dead; ++isc) {
            xp = pl.x_prev[ip];
            yp = pl.y_prev[ip];
            double sx[NS], dsx[NS], sy[NS], dsy[NS];
            const int i0 = nodal_weights<ORDER>((xp - k.xoff)*k.dx_inv, sx, dsx);
            const int j0 = nodal_weights<ORDER>((yp - k.yoff)*k.dy_inv, sy, dsy);
            const int li = i0 - ox, lj = j0 - oy;
            const bool local = (li >= 0 && li + NS <= R && lj >= 0 && lj + NS <= R);
            if (!local) ++nfb;
            Fld F{0, 0, 0, 0, 0, 0};
            if (local) {
                const double* b = img + lj*R + li;
#pragma unroll 1
                for (int iy = 0; iy < NS; ++iy) {
                    double rp = 0.0, rd = 0.0, rez = 0.0, rbx = 0.0, rby = 0.0, rbz = 0.0;
#pragma unroll
                    for (int ix = 0; ix < NS; ++ix) {
                        const int ls = iy*R + ix;
                        const double psi_c = lds_get(b + ls);
                        rp = fma(sx[ix], psi_c, rp);
                        rd = fma(dsx[ix], psi_c, rd);
                        rez = fma(sx[ix], lds_get(b + R*R + ls), rez);
                        rbx = fma(sx[ix], lds_get(b + 2*R*R + ls), rbx);
                        rby = fma(sx[ix], lds_get(b + 3*R*R + ls), rby);
                        rbz = fma(sx[ix], lds_get(b + 4*R*R + ls), rbz);
                    }
                    F.ExmBy = fma(sy[iy], rd, F.ExmBy);
                    F.EypBx = fma(dsy[iy], rp, F.EypBx);
                    F.Ez  = fma(sy[iy], rez, F.Ez);
                    F.Bxc = fma(sy[iy], rbx, F.Bxc);
                    F.Byc = fma(sy[iy], rby, F.Byc);
                    F.Bz  = fma(sy[iy], rbz, F.Bz);
                }
                F.ExmBy *= k.dx_inv;
                F.EypBx *= k.dy_inv;
            } else {
#pragma unroll 1
                for (int iy = 0; iy < NS; ++iy) {
#pragma unroll
                    for (int ix = 0; ix < NS; ++ix) {
                        const double* p = f.p + f.off(i0 + ix, j0 + iy);
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
            }
            F.Bxc *= k.c;
            F.Byc *= k.c;
            const double dz = k.dz, sdz = dz*0.25;
            ux = pl.ux_half[ip]; uy = pl.uy_half[ip]; psi = pl.psi_half[ip];
#pragma unroll 1
            for (int s = 0; s < 4; ++s) taylor2_substep(ux, uy, psi, F, k.c_inv, qmc, sdz);
            const double pinv = fast_rcp(psi);
            xp += dz*k.c_inv*(ux*pinv);
            yp += dz*k.c_inv*(uy*pinv);
            if (apply_particle_bc(k, xp, yp, ux, uy)) {
                pl.w[ip] = 0.0;
                pl.idcpu[ip] = id & ~HPS_ID_VALID;
                pl.psi_half[ip] = 0.0;
                dead = true;
                break;
            }
            pl.x[ip] = xp; pl.y[ip] = yp;
            pl.ux_half[ip] = ux; pl.uy_half[ip] = uy; pl.psi_half[ip] = psi;
            if (pl.x_prev != pl.x) pl.x_prev[ip] = xp;      // (aliased by the engine: already stored)
            if (pl.y_prev != pl.y) pl.y_prev[ip] = yp;
#pragma unroll 1
            for (int s = 0; s < 2; ++s) taylor2_substep(ux, uy, psi, F, k.c_inv, qmc, sdz);
            pl.ux[ip] = ux; pl.uy[ip] = uy; pl.psi[ip] = psi;
        }
        if (dead) continue;
        // ---- DepositCurrent of the pushed particle into the next slice (same arithmetic as k_deposit_tiled) ----
        const double w = pl.w[ip];
        const double psi_inv = 1.0/psi;
        const double vx_c = ux*psi_inv;
        const double vy_c = uy*psi_inv;
        const double q_invvol = kd.a*w;
        const double gamma_psi = 0.5*(psi_inv*psi_inv + vx_c*vx_c*kd.c_inv*kd.c_inv + vy_c*vy_c*kd.c_inv*kd.c_inv + 1.0);
        if (gamma_psi < 0.0 || gamma_psi > kd.max_qsa || psi_inv < 0.0) {
            if (n_qsa) atomicAdd(n_qsa, 1);
            pl.w[ip] = 0.0;
            pl.idcpu[ip] = id & ~HPS_ID_VALID;
            pl.psi_half[ip] = 0.0;
            continue;
        }
        double wx[ORDER + 1], wy[ORDER + 1];
        const int i0 = shape_weights<ORDER>((xp - kd.xoff)*kd.dx_inv, wx);
        const int j0 = shape_weights<ORDER>((yp - kd.yoff)*kd.dy_inv, wy);
        const double wv[6] = {vx_c, vy_c, (gamma_psi - 1.0)*kd.c, gamma_psi, kd.b*psi_inv, 1.0};
        const int li = i0 - ox, lj = j0 - oy;
        if (li >= 0 && li + ORDER < R && lj >= 0 && lj + ORDER < R) {
#pragma unroll
            for (int iy = 0; iy <= ORDER; ++iy) {
#pragma unroll
                for (int ix = 0; ix <= ORDER; ++ix) {
                    const double cd = q_invvol*wx[ix]*wy[iy];
                    double* p = acc + (lj + iy)*R + li + ix;
#pragma unroll
                    for (int c = 0; c < 6; ++c) if (slot[c] >= 0) lds_add(p + slot[c]*R*R, cd*wv[c]);
                }
            }
        } else {
            ++nfb;
#pragma unroll
            for (int iy = 0; iy <= ORDER; ++iy) {
#pragma unroll
                for (int ix = 0; ix <= ORDER; ++ix) {
                    const double cd = q_invvol*wx[ix]*wy[iy];
                    double* p = f.p + f.off(i0 + ix, j0 + iy);
#pragma unroll
                    for (int c = 0; c < 6; ++c) if (slot[c] >= 0) atomic_add_f64(p + gc[c]*f.ns, cd*wv[c]);
                }
            }
        }
    }
    if (n_fallback && nfb) atomicAdd(n_fallback, nfb);
    __syncthreads();
    for (int s = tid; s < R*R; s += NT) {
        const int lj = s / R, li = s - lj*R;
        const int i = ox + li, j = oy + lj;
        if (i < -f.ng || i >= f.nx + f.ng || j < -f.ng || j >= f.ny + f.ng) continue;
        double* p = f.p + f.off(i, j);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            if (slot[c] >= 0) {
                const double v = acc[slot[c]*R*R + s];
                if (v != 0.0) atomic_add_f64(p + gc[c]*f.ns, v);
            }
        }
    }
}

template <class K>
static int set_lds (K kernel, size_t bytes)
{
    if (bytes > 64*1024) HPS_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return HPS_OK;
}

#define HPS_DISPATCH_ORDER_TS(order, ts, CALL)                                   \
    switch ((order)*100 + (ts)) {                                                \
        case   16: { CALL(0, 16); } break;  case   32: { CALL(0, 32); } break;    \
        case  116: { CALL(1, 16); } break;  case  132: { CALL(1, 32); } break;    \
        case  216: { CALL(2, 16); } break;  case  232: { CALL(2, 32); } break;    \
        case  316: { CALL(3, 16); } break;  case  332: { CALL(3, 32); } break;    \
        default: set_error("tiled kernels: unsupported order / tile size"); return HPS_ERR_ARG; }

int deposit_current_tiled (const hps_slab& slab, const hps_plasma& pl, const hps_geom& g, const int comp[6], double charge,
                           double mass, int order, double max_qsa, int can_ionize, int* n_qsa, Tiling* T, int* n_fallback,
                           hipStream_t st, int aabs_comp, const int* tile_flag, TailWork tw, const BeamPairWork* beam, const int* go,
                           bool valid_by_w)
{
    if (pl.n == 0) return HPS_OK;
    const BeamPairWork bw = beam ? *beam : BeamPairWork{};
    tw.extra = bw.nwg;
    // the per-tile skip indexes the launch order by blockIdx: only valid while the grid holds tile workgroups alone
    HPS_REQUIRE(!(tile_flag && (tw.nwg || bw.nwg)), "deposit_current_tiled: tile flags cannot be combined with tail or beam workgroups");
    HPS_REQUIRE(pl.n + 256L*tw.nwg < (1L << 28), "deposit_current_tiled: the tile kernels address at most 2^28 particles per sheet");
    PartConsts k = base_consts(g);
    k.a = charge*invvol_of(g); k.b = charge*g.mu0/mass; k.max_qsa = max_qsa; k.can_ionize = can_ionize;
    k.aabs = aabs_comp; k.laser_fac = (charge/g.q_e)*(g.m_e/mass)*(charge/g.q_e)*(g.m_e/mass);
    DepComps cm{comp[0], comp[1], comp[2], comp[3], comp[4], comp[5]};
    int na = 0; for (int c = 0; c < 6; ++c) na += comp[c] >= 0;
    {   int m = 0; for (int c = 0; c < 6; ++c) m |= (comp[c] >= 0) << c;
        // (kernel variants exist for the hot component sets without a laser; the tail's released electrons keep idcpu)
        k.valid_by_w = valid_by_w && !tw.nwg && aabs_comp < 0 && (m == 51 || m == 3); }
    const int R = T->g.ts + 2*TILE_HALO;
    const size_t lds = ((size_t)na*R*(R + HPS_DEP_PAD) + (aabs_comp >= 0 ? (size_t)R*R : 0))*sizeof(double);
    SlabView f(slab);
    int mask = 0; for (int c = 0; c < 6; ++c) mask |= (comp[c] >= 0) << c;
#define CALLM(O, S, M) { if (int e = set_lds(k_deposit_tiled<O, S, M>, lds)) return e; \
        hipLaunchKernelGGL((k_deposit_tiled<O, S, M>), dim3(T->g.ntiles + tw.nwg + bw.nwg), dim3(256), lds, st, f, pl, T->offsets, T->g.ntx, cm, k, n_qsa, n_fallback, tile_flag, tw, bw, go); }
#define CALLL(O, S, M) { if (int e = set_lds(k_deposit_tiled<O, S, M, true>, lds)) return e; \
        hipLaunchKernelGGL((k_deposit_tiled<O, S, M, true>), dim3(T->g.ntiles + tw.nwg + bw.nwg), dim3(256), lds, st, f, pl, T->offsets, T->g.ntx, cm, k, n_qsa, n_fallback, tile_flag, tw, bw, go); }
#define CALLV(O, S, M) { if (int e = set_lds(k_deposit_tiled<O, S, M, false, true>, lds)) return e; \
        hipLaunchKernelGGL((k_deposit_tiled<O, S, M, false, true>), dim3(T->g.ntiles + tw.nwg + bw.nwg), dim3(256), lds, st, f, pl, T->offsets, T->g.ntx, cm, k, n_qsa, n_fallback, tile_flag, tw, bw, go); }
#define CALL(O, S) { if (aabs_comp >= 0) { if (mask == 51) CALLL(O, S, 51) else CALLL(O, S, -1) } \
                     else if (k.valid_by_w && mask == 51) CALLV(O, S, 51) else if (k.valid_by_w && mask == 3) CALLV(O, S, 3) \
                     else if (mask == 51) CALLM(O, S, 51) else if (mask == 59) CALLM(O, S, 59) else if (mask == 32) CALLM(O, S, 32) \
                     else if (mask == 3) CALLM(O, S, 3) else if (mask == 39) CALLM(O, S, 39) else if (mask == 47) CALLM(O, S, 47) else CALLM(O, S, -1) }
    HPS_DISPATCH_ORDER_TS(order, T->g.ts, CALL)
#undef CALL
#undef CALLM
#undef CALLL
#undef CALLV
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

int explicit_deposit_tiled (const hps_slab& slab, const hps_plasma& pl, const hps_geom& g, const int cache[4], const int depos[2],
                            double charge, double mass, int order, int dtype, int can_ionize, Tiling* T, int* n_fallback,
                            hipStream_t st, int aabs_comp, const int* tile_flag, TailWork tw, bool valid_by_w)
{
    if (pl.n == 0) return HPS_OK;
    HPS_REQUIRE(!(tile_flag && tw.nwg), "explicit_deposit_tiled: tile flags cannot be combined with tail workgroups");
    HPS_REQUIRE(pl.n + 256L*tw.nwg < (1L << 28), "explicit_deposit_tiled: the tile kernels address at most 2^28 particles per sheet");
    PartConsts k = base_consts(g);
    k.a = charge*invvol_of(g)*g.mu0; k.b = charge/mass; k.can_ionize = can_ionize;
    k.valid_by_w = valid_by_w && !tw.nwg && aabs_comp < 0 && dtype == 2;      // (the kernel variant that exists)
    k.aabs = aabs_comp; k.laser_fac = (g.m_e/g.q_e)*(g.m_e/g.q_e);
    const int R = T->g.ts + 2*TILE_HALO;
    const int pad = expl_pad();
    const size_t lds = (size_t)(aabs_comp >= 0 ? 7 : 6)*R*(R + pad)*sizeof(double);
    SlabView f(slab);
#define HPS_EXPL_LAUNCH(O, D, S, L, P, V) { if (int e = set_lds(k_explicit_tiled<O, D, S, L, P, V>, lds)) return e; \
        hipLaunchKernelGGL((k_explicit_tiled<O, D, S, L, P, V>), dim3(T->g.ntiles + tw.nwg), dim3(256), lds, st, f, pl, T->offsets, T->g.ntx, \
                           cache[0], cache[1], cache[2], cache[3], depos[0], depos[1], k, n_fallback, tile_flag, tw); }
#define HPS_EXPL_PADS(O, D, S, L) { if (pad == 8) HPS_EXPL_LAUNCH(O, D, S, L, 8, false) else if (!L && D == 2 && k.valid_by_w) HPS_EXPL_LAUNCH(O, D, S, false, 2, true) else HPS_EXPL_LAUNCH(O, D, S, L, 2, false) }
    if (dtype == 2) {
#define CALL(O, S) { if (aabs_comp >= 0) HPS_EXPL_PADS(O, 2, S, true) else HPS_EXPL_PADS(O, 2, S, false) }
        HPS_DISPATCH_ORDER_TS(order, T->g.ts, CALL)
#undef CALL
    } else {
#define CALL(O, S) { if (aabs_comp >= 0) HPS_EXPL_PADS(O, 1, S, true) else HPS_EXPL_PADS(O, 1, S, false) }
        HPS_DISPATCH_ORDER_TS(order, T->g.ts, CALL)
#undef CALL
    }
#undef HPS_EXPL_PADS
#undef HPS_EXPL_LAUNCH
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

int advance_plasma_tiled (const hps_slab& slab, const hps_plasma& pl, const hps_geom& g, const int comp[5], double charge,
                          double mass, int order, int temp_slice, int n_subcycles, int can_ionize, Tiling* T,
                          int* n_fallback, hipStream_t st, int aabs_comp, const IonArgs* ion, const int* go, TailWork tw, const MgPost* post)
{
    if (pl.n == 0) return HPS_OK;
    // 32-bit byte offsets into the SoA arrays (ip*8u, __builtin_assume(ip < 2^28) in the kernel); the tail's live count is
    // read on the device but bounded by the workgroups it was given
    HPS_REQUIRE(pl.n + 256L*tw.nwg < (1L << 28), "advance_plasma_tiled: the tile kernels address at most 2^28 particles per sheet");
    HPS_REQUIRE(!(ion && ion->tile_flag && tw.nwg), "advance_plasma_tiled: an ionisable species' tile flags cannot be combined with tail workgroups");
    PartConsts k = base_consts(g);
    k.a = charge/(mass*g.c); k.dz = g.dz/n_subcycles;
    k.aabs = aabs_comp; k.laser_fac = (charge/g.q_e)*(g.m_e/mass)*(charge/g.q_e)*(g.m_e/mass);
    k.temp_slice = temp_slice; k.n_subcycles = n_subcycles; k.can_ionize = can_ionize;
    const int R = T->g.ts + 2*TILE_HALO;
    const size_t lds = (size_t)(aabs_comp >= 0 ? 6 : 5)*R*R*sizeof(double);
    SlabView f(slab);
    const IonArgs ia = ion ? *ion : IonArgs{};
    HPS_REQUIRE(!(post && (tw.nwg || tw.extra)), "advance_plasma_tiled: the launch that posts the multigrid's norms has tile workgroups only");
    const MgPost mp = post ? *post : MgPost{};
    // (the variant without the idcpu read exists for order 2 on 16 x 16 tiles, as the depositions' <.., VBW>)
    const bool vbp = T->valid_by_psi && !ion && !can_ionize && order == 2 && T->g.ts == 16;
#define HPS_ADV(O, S, L, I, V) { if (int e = set_lds(k_advance_tiled<O, S, L, I, V>, lds)) return e; \
        hipLaunchKernelGGL((k_advance_tiled<O, S, L, I, V>), dim3(T->g.ntiles + tw.nwg), dim3(256), lds, st, f, pl, T->offsets, T->g.ntx, \
                           comp[0], comp[1], comp[2], comp[3], comp[4], k, n_fallback, ia, go, tw, mp, DepTail{}); }
#define CALL(O, S) { if (ion) { if (aabs_comp >= 0) HPS_ADV(O, S, true, true, false) else HPS_ADV(O, S, false, true, false) } \
                     else if (vbp && O == 2 && S == 16) { if (aabs_comp >= 0) HPS_ADV(2, 16, true, false, true) else HPS_ADV(2, 16, false, false, true) } \
                     else     { if (aabs_comp >= 0) HPS_ADV(O, S, true, false, false) else HPS_ADV(O, S, false, false, false) } }
    HPS_DISPATCH_ORDER_TS(order, T->g.ts, CALL)
#undef CALL
#undef HPS_ADV
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

// push of this slice + deposition into the next one (k_advance_deposit_tiled); dep_comp as DepositCurrent's comp[6]
int advance_deposit_tiled (const hps_slab& slab, const hps_plasma& pl, const hps_geom& g, const int comp[5], const int dep_comp[6],
                           double charge, double mass, int order, int n_subcycles, double max_qsa, int* n_qsa, Tiling* T,
                           int* n_fallback, hipStream_t st)
{
    if (pl.n == 0) return HPS_OK;
    PartConsts k = base_consts(g);
    k.a = charge/(mass*g.c); k.dz = g.dz/n_subcycles; k.temp_slice = 0; k.n_subcycles = n_subcycles; k.can_ionize = 0;
    PartConsts kd = base_consts(g);
    kd.a = charge*invvol_of(g); kd.b = charge*g.mu0/mass; kd.max_qsa = max_qsa; kd.can_ionize = 0;
    DepComps cm{dep_comp[0], dep_comp[1], dep_comp[2], dep_comp[3], dep_comp[4], dep_comp[5]};
    int mask = 0, na = 0; for (int c = 0; c < 6; ++c) { mask |= (dep_comp[c] >= 0) << c; na += dep_comp[c] >= 0; }
    if (mask != 51 && mask != 59) { set_error("advance_deposit_tiled: deposits jx jy chi rhomjz [rho] only"); return HPS_ERR_UNSUPPORTED; }
    const int R = T->g.ts + 2*TILE_HALO;
    SlabView f(slab);
    // round 5: the push kernel of the separate schedule with the deposition as its tail (k_advance_tiled<.., DEP>); HPS_FUSED_KERNEL=old:
    // round 2's kernel that holds image and accumulators together
    static const bool modern = [] { const char* e = std::getenv("HPS_FUSED_KERNEL"); return !(e && std::string(e) == "old"); }();
    if (modern && (order == 2 && T->g.ts == 16)) {
        const size_t lds5 = (size_t)5*R*R*sizeof(double);
        DepTail dt{cm, kd.a, kd.b, max_qsa, n_qsa};
        const IonArgs ia{}; const TailWork tw0{}; const MgPost mp0{};
        if (mask == 51) { if (int e = set_lds(k_advance_tiled<2, 16, false, false, false, 51>, lds5)) return e;
            hipLaunchKernelGGL((k_advance_tiled<2, 16, false, false, false, 51>), dim3(T->g.ntiles), dim3(256), lds5, st, f, pl, T->offsets, T->g.ntx,
                               comp[0], comp[1], comp[2], comp[3], comp[4], k, n_fallback, ia, (const int*)nullptr, tw0, mp0, dt); }
        else            { if (int e = set_lds(k_advance_tiled<2, 16, false, false, false, 59>, lds5)) return e;
            hipLaunchKernelGGL((k_advance_tiled<2, 16, false, false, false, 59>), dim3(T->g.ntiles), dim3(256), lds5, st, f, pl, T->offsets, T->g.ntx,
                               comp[0], comp[1], comp[2], comp[3], comp[4], k, n_fallback, ia, (const int*)nullptr, tw0, mp0, dt); }
        HPS_HIP_CHECK(hipGetLastError());
        return HPS_OK;
    }
    const size_t lds = (size_t)(5 + na)*R*R*sizeof(double);
    static int nt = 0;
    // (measured at 1024^2 x 4 ppc: 256 threads 285 us, 512 threads 316 us -- against 175 + 77 us for the two kernels)
    if (nt == 0) { nt = 256; if (const char* e = std::getenv("HPS_FUSED_THREADS")) { const int v = std::atoi(e); if (v == 256 || v == 512) nt = v; } }
#define HPS_AD(O, S, M, N) { if (int e = set_lds(k_advance_deposit_tiled<O, S, M, N>, lds)) return e; \
        hipLaunchKernelGGL((k_advance_deposit_tiled<O, S, M, N>), dim3(T->g.ntiles), dim3(N), lds, st, f, pl, T->offsets, T->g.ntx, \
                           comp[0], comp[1], comp[2], comp[3], comp[4], k, cm, kd, n_qsa, n_fallback); }
#define CALL(O, S) { if (nt == 512) { if (mask == 51) HPS_AD(O, S, 51, 512) else HPS_AD(O, S, 59, 512) } \
                     else           { if (mask == 51) HPS_AD(O, S, 51, 256) else HPS_AD(O, S, 59, 256) } }
    HPS_DISPATCH_ORDER_TS(order, T->g.ts, CALL)
#undef CALL
#undef HPS_AD
    HPS_HIP_CHECK(hipGetLastError());
    return HPS_OK;
}

} // namespace hps

using namespace hps;

extern "C" int hps_particles_debug_stamps (long long* stamps8_host)
{
#ifndef HPS_STAMPS
    (void)stamps8_host; set_error("hps_particles_debug_stamps: the library was built without -DHPS_STAMPS (make stamps)"); return HPS_ERR_UNSUPPORTED;
#else
    static long long* d = nullptr;
    if (!d) {
        HPS_HIP_CHECK(hipMalloc(&d, 8*sizeof(long long)));
        HPS_HIP_CHECK(hipMemset(d, 0, 8*sizeof(long long)));
        HPS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_pt_dbg), &d, sizeof(d)));
        return HPS_OK;
    }
    HPS_HIP_CHECK(hipDeviceSynchronize());
    HPS_HIP_CHECK(hipMemcpy(stamps8_host, d, 8*sizeof(long long), hipMemcpyDeviceToHost));
    return HPS_OK;
#endif
}

static int check_tiling (void* tiling, const hps_slab& s, const hps_plasma& pl, const char* what)
{
    if (!tiling) { set_error(std::string(what) + ": null tiling"); return HPS_ERR_ARG; }
    Tiling* T = static_cast<Tiling*>(tiling);
    if (T->g.nx != s.nx || T->g.ny != s.ny) { set_error(std::string(what) + ": tiling was built for another grid"); return HPS_ERR_ARG; }
    if (T->sorted_n != pl.n) { set_error(std::string(what) + ": particle count changed since hps_reorder_particles"); return HPS_ERR_ARG; }
    return HPS_OK;
}

extern "C" int hps_deposit_current_tiled (hps_slab slab, hps_plasma pl, hps_geom g, const int comp[6], double charge,
                                          double mass, int order, double max_qsa, int can_ionize, int* n_qsa,
                                          void* tiling, int* n_fallback, hps_stream stream)
{
    HPS_REQUIRE(order >= 0 && order <= 3, "hps_deposit_current_tiled: depos_order must be 0..3");
    if (int e = check_stencil(slab, (order + 1)/2, "hps_deposit_current_tiled")) return e;
    if (int e = check_tiling(tiling, slab, pl, "hps_deposit_current_tiled")) return e;
    for (int c = 0; c < 6; ++c) HPS_REQUIRE(comp[c] >= -1 && comp[c] < slab.ncomp, "hps_deposit_current_tiled: bad component");
    return deposit_current_tiled(slab, pl, g, comp, charge, mass, order, max_qsa, can_ionize, n_qsa,
                                 static_cast<Tiling*>(tiling), n_fallback, (hipStream_t)stream, -1, nullptr, TailWork{}, nullptr, nullptr, false);
}

extern "C" int hps_explicit_deposit_tiled (hps_slab slab, hps_plasma pl, hps_geom g, const int cache[4], const int depos[2],
                                           double charge, double mass, int order, int dtype, int can_ionize, void* tiling,
                                           int* n_fallback, hps_stream stream)
{
    HPS_REQUIRE(order >= 0 && order <= 3, "hps_explicit_deposit_tiled: depos_order must be 0..3");
    if (dtype != 1 && dtype != 2) { set_error("hps_explicit_deposit_tiled: derivative_type 1 or 2 only"); return HPS_ERR_UNSUPPORTED; }
    if (int e = check_stencil(slab, (order + 1)/2 + 1, "hps_explicit_deposit_tiled")) return e;
    if (int e = check_tiling(tiling, slab, pl, "hps_explicit_deposit_tiled")) return e;
    return explicit_deposit_tiled(slab, pl, g, cache, depos, charge, mass, order, dtype, can_ionize,
                                  static_cast<Tiling*>(tiling), n_fallback, (hipStream_t)stream, -1, nullptr, TailWork{}, false);
}

extern "C" int hps_advance_plasma_tiled (hps_slab slab, hps_plasma pl, hps_geom g, const int comp[5], double charge,
                                         double mass, int order, int temp_slice, int n_subcycles, int can_ionize,
                                         void* tiling, int* n_fallback, hps_stream stream)
{
    HPS_REQUIRE(order >= 0 && order <= 3, "hps_advance_plasma_tiled: depos_order must be 0..3");
    HPS_REQUIRE(n_subcycles >= 1, "hps_advance_plasma_tiled: n_subcycles must be >= 1");
    if (int e = check_stencil(slab, (order + 1)/2 + 1, "hps_advance_plasma_tiled")) return e;
    if (int e = check_tiling(tiling, slab, pl, "hps_advance_plasma_tiled")) return e;
    return advance_plasma_tiled(slab, pl, g, comp, charge, mass, order, temp_slice, n_subcycles, can_ionize,
                                static_cast<Tiling*>(tiling), n_fallback, (hipStream_t)stream, -1, nullptr, nullptr, TailWork{}, nullptr);
}
