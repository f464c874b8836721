// sort.hip -- tile binning of the plasma sheet (the "ReorderParticles" seam of the reference,
// particles/plasma/PlasmaParticleContainer.cpp:196-208, which calls amrex::SortParticlesForDeposition).
//
// The sheet is kept physically sorted by TS x TS-cell transverse tiles so that one workgroup can
// accumulate a whole tile's deposition in LDS and serve its gathers from an LDS copy of the fields.
// Inside a tile the particles are interleaved over the cells: all first particles of the tile's
// cells in cell order, then all second ones, ... so that the 64 lanes of a wave hit 64 different
// cells (consecutive LDS addresses).  Measured at 1024^2 x 4 ppc: particles of one cell next to
// each other cost 200 us per deposition (same-address LDS atomics serialise), a random order
// 125 us, the interleaved order 70 us.
//   pass 1: stable sort by (tile, cell in tile) of the particle's nearest cell, invalid ones last;
//           rank = position within the run of equal keys, capped at RANK_CAP - 1
//   pass 2: stable sort by (tile, rank, cell in tile)
// Both are rocPRIM radix_sort_pairs (stable), so the permutation is reproducible bit for bit by the
// CPU restatement (oracle: orc_tile_sort).  All 11 real arrays + idcpu + ion_lev are then gathered
// through the permutation into a second SoA buffer.
#include "common.h"
#include "tiling.h"

#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace hps {

#ifndef HPS_CELL_BLOCK_W
#define HPS_CELL_BLOCK_W 32     /* >= tile size: row by row.  4 (blocks of 4 x 8 cells) was measured: deposit 75.2 -> 73.9 us, explicit deposit 124.8 -> 133.4 us, 1222 -> 1211 slices/s */
#endif
#ifndef HPS_RANK_CAP
#define HPS_RANK_CAP 16
#endif
constexpr int RANK_CAP = HPS_RANK_CAP;

// Number of cell (x, y) inside its tile.  The particles of a tile are interleaved over its cells in this order, so 32
// consecutive numbers are the cells the lanes of a half-wave work on at the same time, and their LDS words must fall
// into 32 different bank pairs.  Row by row (bw = ts, the default) a half-wave covers two rows of 16 cells: with the row
// pitches of the LDS images (28, 30 doubles) the second row lands on the banks of the first.  In blocks of bw = 4 cells
// across and 8 down, word = x + pitch*y with pitch = 28 or 36 = -4 or +4 (mod 32) would be 32 different bank pairs for
// a wave that starts on a block -- measured (HPS_CELL_BLOCK_W=4, HPS_EXPL_PAD=8) it is slower than row by row: behind the
// driver the cells hold 0 .. 16 particles, the waves do not start on block boundaries, and the taller footprint of a
// wave costs the explicit deposition more than the aligned case gains.
__host__ __device__ __forceinline__ int cell_in_tile (int x, int y, int ts, int bw)
{
    if (bw >= ts) return y*ts + x;
    const int bh = 32/bw;                        // 32 cells per block
    const int bx = x/bw, by = y/bh, nbx = ts/bw;
    return ((by*nbx + bx)*bh + (y - by*bh))*bw + (x - bx*bw);
}

// (tile, cell in tile) of the nearest cell; invalid particles get tile = ntiles, cell 0
__device__ __forceinline__ unsigned int cell_key (double x, double y, uint64_t id, const TileGeom& t)
{
    const int ncell = t.ts*t.ts;
    if (!(id & HPS_ID_VALID)) return (unsigned int)(t.ntiles*ncell);
    int ci = (int)floor((x - t.xoff)*t.dx_inv + 0.5);
    int cj = (int)floor((y - t.yoff)*t.dy_inv + 0.5);
    ci = min(max(ci, 0), t.nx - 1);
    cj = min(max(cj, 0), t.ny - 1);
    const int tile = (cj / t.ts)*t.ntx + (ci / t.ts);
    return (unsigned int)(tile*ncell + cell_in_tile(ci % t.ts, cj % t.ts, t.ts, t.bw));
}

__global__ __launch_bounds__(256)
void k_cell_keys (hps_plasma pl, TileGeom t, unsigned int* keys, unsigned int* idx)
{
    const long p = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (p >= pl.n) return;
    keys[p] = cell_key(pl.x[p], pl.y[p], pl.idcpu[p], t);
    idx[p] = (unsigned int)p;
}

// first[k] = first sorted position with key >= k, k = 0 .. nkeys  (keys sorted ascending, < nkeys)
__global__ __launch_bounds__(256)
void k_run_starts (const unsigned int* keys, long n, int nkeys, int* first)
{
    const long p = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (p > n) return;
    const int prev = (p == 0) ? -1 : (int)keys[p - 1];
    const int cur = (p == n) ? nkeys : (int)keys[p];
    for (int k = prev + 1; k <= cur; ++k) first[k] = (int)p;
}

// second key from the position inside the run of equal cell keys
__global__ __launch_bounds__(256)
void k_rank_keys (const unsigned int* ckeys, const int* first, long n, int ncell, unsigned int* keys2)
{
    const long p = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (p >= n) return;
    const unsigned int ck = ckeys[p];
    const int rank = min((int)p - first[ck], RANK_CAP - 1);
    const unsigned int tile = ck / ncell, cit = ck - tile*ncell;
    keys2[p] = (tile*RANK_CAP + rank)*ncell + cit;
}

// offsets[t] = first sorted position whose tile (= key / per_tile) is >= t, t = 0 .. ntiles+1
__global__ __launch_bounds__(256)
void k_tile_offsets (const unsigned int* keys, long n, int ntiles, int per_tile, int* offsets)
{
    const long p = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (p > n) return;
    const int prev = (p == 0) ? -1 : (int)(keys[p - 1]/per_tile);
    const int cur = (p == n) ? ntiles + 1 : (int)(keys[p]/per_tile);
    for (int t = prev + 1; t <= cur; ++t) offsets[t] = (int)p;
}

// launch order of the tile kernels: workgroup b works on tile order[b], heaviest tile first (longest-processing-time
// first: behind a driver the sheet is very uneven over the tiles and the tail of a launch was a few heavy tiles)
__global__ __launch_bounds__(256)
void k_tile_order_keys (const int* offsets, int ntiles, unsigned int* keys, unsigned int* vals)
{
    const int t = blockIdx.x*blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    keys[t] = 0xFFFFFFFFu - (unsigned int)(offsets[t + 1] - offsets[t]);
    vals[t] = (unsigned int)t;
}
__global__ __launch_bounds__(256)
void k_tile_order_identity (int* order, int ntiles)
{
    const int t = blockIdx.x*blockDim.x + threadIdx.x;
    if (t < ntiles) order[t] = t;
}
// launch[b] = {tile, first particle, end} of workgroup b: ONE 16-byte scalar load at the head of a tile kernel instead of the
// chain order[b] -> offsets[tile], offsets[tile + 1] (two dependent trips to memory before the first particle is requested)
__global__ __launch_bounds__(256)
void k_tile_launch_info (const int* offsets, int ntiles, int4* launch)
{
    const int b = blockIdx.x*blockDim.x + threadIdx.x;
    if (b >= ntiles) return;
    const int t = offsets[ntiles + 2 + b];
    launch[b] = make_int4(t, offsets[t], offsets[t + 1], 0);
}

__global__ __launch_bounds__(256)
void k_permute (hps_plasma src, hps_plasma dst, const unsigned int* perm)
{
    const long p = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (p >= src.n) return;
    const unsigned int q = perm[p];
    dst.x[p] = src.x[q]; dst.y[p] = src.y[q]; dst.w[p] = src.w[q];
    dst.ux[p] = src.ux[q]; dst.uy[p] = src.uy[q]; dst.psi[p] = src.psi[q];
    if (dst.x_prev != dst.x) dst.x_prev[p] = src.x_prev[q];     // the engine aliases x_prev/x, y_prev/y
    if (dst.y_prev != dst.y) dst.y_prev[p] = src.y_prev[q];
    dst.ux_half[p] = src.ux_half[q]; dst.uy_half[p] = src.uy_half[q]; dst.psi_half[p] = src.psi_half[q];
    dst.idcpu[p] = src.idcpu[q]; dst.ion_lev[p] = src.ion_lev[q];
}

Tiling::~Tiling ()
{
    (void)hipFree(offsets); (void)hipFree(keys_a); (void)hipFree(keys_b); (void)hipFree(idx_a); (void)hipFree(idx_b);
    (void)hipFree(temp); (void)hipFree(cell_first); (void)hipFree(okeys); (void)hipFree(otemp);
}

int tiling_create (int nx, int ny, int ts, long capacity, Tiling** out)
{
    if (ts != 16 && ts != 32) { set_error("hps_tiling_create: tile_size must be 16 or 32"); return HPS_ERR_ARG; }
    Tiling* T = new Tiling;
    T->g.nx = nx; T->g.ny = ny; T->g.ts = ts;
    T->g.ntx = (nx + ts - 1)/ts; T->g.nty = (ny + ts - 1)/ts; T->g.ntiles = T->g.ntx*T->g.nty;
    T->g.bw = HPS_CELL_BLOCK_W;
    if (const char* e = std::getenv("HPS_CELL_BLOCK_W")) { const int v = std::atoi(e); if (v == 4 || v == 8 || v == 16 || v == 32) T->g.bw = v; }
    if (T->g.bw > ts) T->g.bw = ts;
    T->capacity = capacity;
    HPS_HIP_CHECK(hipMalloc(&T->offsets, (tile_launch_offset(T->g.ntiles) + 4*T->g.ntiles)*sizeof(int)));
    HPS_HIP_CHECK(hipMemset(T->offsets, 0, (tile_launch_offset(T->g.ntiles) + 4*T->g.ntiles)*sizeof(int)));
    hipLaunchKernelGGL(k_tile_order_identity, dim3(ceil_div(T->g.ntiles, 256)), dim3(256), 0, (hipStream_t)0, T->offsets + T->g.ntiles + 2, T->g.ntiles);
    hipLaunchKernelGGL(k_tile_launch_info, dim3(ceil_div(T->g.ntiles, 256)), dim3(256), 0, (hipStream_t)0, T->offsets, T->g.ntiles, reinterpret_cast<int4*>(T->offsets + tile_launch_offset(T->g.ntiles)));
    HPS_HIP_CHECK(hipDeviceSynchronize());
    HPS_HIP_CHECK(hipMalloc(&T->okeys, 3*(size_t)T->g.ntiles*sizeof(unsigned int)));
    HPS_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, T->otemp_bytes, T->okeys, T->okeys, T->okeys, T->okeys, (size_t)T->g.ntiles, 0, 32, (hipStream_t)0));
    HPS_HIP_CHECK(hipMalloc(&T->otemp, T->otemp_bytes));
    HPS_HIP_CHECK(hipMalloc(&T->keys_a, capacity*sizeof(unsigned int)));
    HPS_HIP_CHECK(hipMalloc(&T->keys_b, capacity*sizeof(unsigned int)));
    HPS_HIP_CHECK(hipMalloc(&T->idx_a, capacity*sizeof(unsigned int)));
    HPS_HIP_CHECK(hipMalloc(&T->idx_b, capacity*sizeof(unsigned int)));
    const long ncell = (long)ts*ts;
    const long nkeys1 = (long)T->g.ntiles*ncell + 1;                 // cell keys, + the invalid key
    const long nkeys2 = ((long)T->g.ntiles*RANK_CAP + 1)*ncell;      // (tile, rank, cell) keys
    if (nkeys2 >= (1L << 31)) { delete T; set_error("hps_tiling_create: grid too large for 32-bit sort keys"); return HPS_ERR_ARG; }
    int bits = 1; while ((1L << bits) < nkeys1) ++bits;
    T->key_bits = bits;
    bits = 1; while ((1L << bits) < nkeys2) ++bits;
    T->key2_bits = bits;
    HPS_HIP_CHECK(hipMalloc(&T->cell_first, (nkeys1 + 1)*sizeof(int)));
    HPS_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, T->temp_bytes, T->keys_a, T->keys_b, T->idx_a, T->idx_b,
                                            (size_t)capacity, 0, T->key2_bits, (hipStream_t)0));
    HPS_HIP_CHECK(hipMalloc(&T->temp, T->temp_bytes));
    *out = T;
    return HPS_OK;
}

int tiling_sort (Tiling* T, const hps_plasma& src, const hps_plasma& dst, const hps_geom& g, hipStream_t st)
{
    if (src.n > T->capacity) { set_error("hps_reorder_particles: more particles than the tiling capacity"); return HPS_ERR_ARG; }
    T->g.xoff = g.xoff; T->g.yoff = g.yoff; T->g.dx_inv = 1.0/g.dx; T->g.dy_inv = 1.0/g.dy;
    const long n = src.n;
    if (n == 0) {
        HPS_HIP_CHECK(hipMemsetAsync(T->offsets, 0, (T->g.ntiles + 2)*sizeof(int), st));
        hipLaunchKernelGGL(k_tile_launch_info, dim3(ceil_div(T->g.ntiles, 256)), dim3(256), 0, st, T->offsets, T->g.ntiles, reinterpret_cast<int4*>(T->offsets + tile_launch_offset(T->g.ntiles)));
        T->sorted_n = 0; return HPS_OK;
    }
    const int ncell = T->g.ts*T->g.ts;
    const int nkeys1 = T->g.ntiles*ncell + 1;
    const dim3 gn(ceil_div(n, 256)), gn1(ceil_div(n + 1, 256)), b256(256);
    hipLaunchKernelGGL(k_cell_keys, gn, b256, 0, st, src, T->g, T->keys_a, T->idx_a);
    size_t tb = T->temp_bytes;
    HPS_HIP_CHECK(rocprim::radix_sort_pairs(T->temp, tb, T->keys_a, T->keys_b, T->idx_a, T->idx_b, (size_t)n, 0,
                                            T->key_bits, st));
    hipLaunchKernelGGL(k_run_starts, gn1, b256, 0, st, T->keys_b, n, nkeys1, T->cell_first);
    hipLaunchKernelGGL(k_rank_keys, gn, b256, 0, st, T->keys_b, T->cell_first, n, ncell, T->keys_a);
    tb = T->temp_bytes;
    HPS_HIP_CHECK(rocprim::radix_sort_pairs(T->temp, tb, T->keys_a, T->keys_b, T->idx_b, T->idx_a, (size_t)n, 0,
                                            T->key2_bits, st));
    hipLaunchKernelGGL(k_tile_offsets, gn1, b256, 0, st, T->keys_b, n, T->g.ntiles, RANK_CAP*ncell, T->offsets);
    hipLaunchKernelGGL(k_permute, dim3(ceil_div(n, 256)), dim3(256), 0, st, src, dst, T->idx_a);
    {   const int nt = T->g.ntiles;
        unsigned int *ka = T->okeys, *kb = T->okeys + nt, *va = T->okeys + 2*nt;
        hipLaunchKernelGGL(k_tile_order_keys, dim3(ceil_div(nt, 256)), b256, 0, st, T->offsets, nt, ka, va);
        size_t ob = T->otemp_bytes;
        HPS_HIP_CHECK(rocprim::radix_sort_pairs(T->otemp, ob, ka, kb, va, reinterpret_cast<unsigned int*>(T->offsets + nt + 2), (size_t)nt, 0, 32, st));
        hipLaunchKernelGGL(k_tile_launch_info, dim3(ceil_div(nt, 256)), b256, 0, st, T->offsets, nt, reinterpret_cast<int4*>(T->offsets + tile_launch_offset(nt))); }
    HPS_HIP_CHECK(hipGetLastError());
    T->sorted_n = n;
    return HPS_OK;
}

// ---- beam particles -> longitudinal boxes (BoxSorter::sortParticlesByBox, particles/sorting/BoxSort.cpp:14-78) ------
// box = int((z - plo_z) * dzi), truncated toward zero as the reference's static_cast; outside [0, num_boxes] -> the
// extra last box num_boxes (:40-43)
__global__ __launch_bounds__(256)
void k_box_keys (const double* __restrict__ z, long n, double plo_z, double dzi, int num_boxes, unsigned int* keys, unsigned int* idx)
{
    const long i = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b = static_cast<int>((z[i] - plo_z)*dzi);
    if (b < 0 || b > num_boxes) b = num_boxes;
    keys[i] = (unsigned int)b; idx[i] = (unsigned int)i;
}

// offsets[b] = first sorted position with key >= b (the exclusive scan of the counts, :47), counts[b] = run length
__global__ __launch_bounds__(256)
void k_box_offsets (const unsigned int* keys, long n, int num_boxes, unsigned long long* counts, unsigned long long* offsets)
{
    const long p = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (p > n) return;
    const int prev = (p == 0) ? -1 : (int)keys[p - 1];
    const int cur = (p == n) ? num_boxes + 1 : (int)keys[p];
    for (int k = prev + 1; k <= cur && k <= num_boxes; ++k) offsets[k] = (unsigned long long)p;
    (void)counts;
}
__global__ __launch_bounds__(256)
void k_box_counts (long n, int num_boxes, const unsigned int* idx, unsigned long long* counts, const unsigned long long* offsets,
                   unsigned long long* perm)
{
    const long p = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (p < n) perm[p] = idx[p];
    if (p <= num_boxes) counts[p] = (p == num_boxes ? (unsigned long long)n : offsets[p + 1]) - offsets[p];
}

} // namespace hps

using namespace hps;

extern "C" int hps_beam_sort_by_box (const double* z_dev, long n, double plo_z, double dz, int num_boxes,
                                     unsigned long long* counts_dev, unsigned long long* offsets_dev,
                                     unsigned long long* perm_dev, hps_stream stream)
{
    HPS_REQUIRE(n >= 0 && num_boxes >= 1 && counts_dev && offsets_dev && (perm_dev || n == 0) && (z_dev || n == 0),
                "hps_beam_sort_by_box: bad argument");
    HPS_REQUIRE(n < (1L << 32), "hps_beam_sort_by_box: more than 2^32 particles");
    hipStream_t st = (hipStream_t)stream;
    const size_t nb1 = (size_t)num_boxes + 1;
    if (n == 0) {
        HPS_HIP_CHECK(hipMemsetAsync(counts_dev, 0, nb1*sizeof(unsigned long long), st));
        HPS_HIP_CHECK(hipMemsetAsync(offsets_dev, 0, nb1*sizeof(unsigned long long), st));
        return HPS_OK;
    }
    unsigned int *ka = nullptr, *kb = nullptr, *ia = nullptr, *ib = nullptr; void* temp = nullptr; size_t tb = 0;
    int bits = 1; while ((1L << bits) < (long)nb1) ++bits;
    HPS_HIP_CHECK(hipMalloc(&ka, 4*(size_t)n*sizeof(unsigned int)));
    kb = ka + n; ia = kb + n; ib = ia + n;
    HPS_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tb, ka, kb, ia, ib, (size_t)n, 0, bits, st));
    HPS_HIP_CHECK(hipMalloc(&temp, tb));
    const dim3 b256(256);
    // the reference takes 1/dz from the geometry (InvCellSize)
    hipLaunchKernelGGL(k_box_keys, dim3(ceil_div(n, 256)), b256, 0, st, z_dev, n, plo_z, 1.0/dz, num_boxes, ka, ia);
    // stable: particles keep their order inside a box, which is what the reference's serial CPU build produces
    HPS_HIP_CHECK(rocprim::radix_sort_pairs(temp, tb, ka, kb, ia, ib, (size_t)n, 0, bits, st));
    const long span = std::max<long>(n + 1, (long)nb1);
    hipLaunchKernelGGL(k_box_offsets, dim3(ceil_div(n + 1, 256)), b256, 0, st, kb, n, num_boxes, counts_dev, offsets_dev);
    hipLaunchKernelGGL(k_box_counts, dim3(ceil_div(span, 256)), b256, 0, st, n, num_boxes, ib, counts_dev, offsets_dev, perm_dev);
    HPS_HIP_CHECK(hipGetLastError());
    HPS_HIP_CHECK(hipStreamSynchronize(st));        // the reference synchronises too (:66-69); frees the scratch
    (void)hipFree(ka); (void)hipFree(temp);
    return HPS_OK;
}

extern "C" int hps_tiling_create (int nx, int ny, int tile_size, long max_particles, void** handle)
{
    HPS_REQUIRE(nx > 0 && ny > 0 && max_particles >= 0 && handle, "hps_tiling_create: bad argument");
    HPS_REQUIRE(max_particles < (1L << 28), "hps_tiling_create: at most 2^28 particles per tile-sorted sheet (32-bit offsets in the tile kernels)");
    Tiling* T = nullptr;
    if (int e = tiling_create(nx, ny, tile_size, std::max(max_particles, 1L), &T)) return e;
    *handle = T;
    return HPS_OK;
}

extern "C" int hps_reorder_particles (void* tiling, hps_plasma src, hps_plasma dst, hps_geom geom, hps_stream stream)
{
    HPS_REQUIRE(tiling, "hps_reorder_particles: null tiling");
    HPS_REQUIRE(dst.n >= src.n || src.n == 0, "hps_reorder_particles: destination too small");
    return tiling_sort(static_cast<Tiling*>(tiling), src, dst, geom, (hipStream_t)stream);
}

extern "C" int hps_tiling_info (void* tiling, int* ntiles, const int** offsets_dev, const unsigned int** perm_dev)
{
    Tiling* T = static_cast<Tiling*>(tiling);
    if (ntiles) *ntiles = T->g.ntiles;
    if (offsets_dev) *offsets_dev = T->offsets;
    if (perm_dev) *perm_dev = T->idx_a;
    return HPS_OK;
}

extern "C" int hps_tiling_destroy (void* tiling) { delete static_cast<Tiling*>(tiling); return HPS_OK; }
