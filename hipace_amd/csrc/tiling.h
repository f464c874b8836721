// tiling.h -- tile binning state of the plasma sheet (sort.hip).
#ifndef HPS_TILING_H_
#define HPS_TILING_H_
#include "common.h"

namespace hps {

struct TileGeom { int nx, ny, ts, ntx, nty, ntiles; double xoff, yoff, dx_inv, dy_inv;
                  int bw;      // cells of a tile are numbered in blocks of bw x (32/bw) cells (bw = ts: row by row)
};

struct Tiling {
    TileGeom g{};
    long capacity = 0, sorted_n = 0;
    int* offsets = nullptr;                       // [ntiles + 2] offsets, then [ntiles] launch order (heaviest tile first), then (16-byte
                                                  // aligned, tile_launch_offset) [ntiles] int4 {tile, first, end, 0} per workgroup, device
    unsigned int *okeys = nullptr; void* otemp = nullptr; size_t otemp_bytes = 0;      // scratch of the launch-order sort
    unsigned int *keys_a = nullptr, *keys_b = nullptr, *idx_a = nullptr, *idx_b = nullptr;
    void* temp = nullptr; size_t temp_bytes = 0; int key_bits = 0, key2_bits = 0;
    int* cell_first = nullptr;                    // [ntiles*ts*ts + 2] run starts of the cell keys
    // the sheet is one whose every invalidation also zeroes psi_half (the engine's own electron sheet: k_init_plasma, the QSA drop of
    // the depositions, the absorbing boundary of the pushes): the tile push then takes "psi_half != 0" for the valid bit and does
    // not read idcpu (HPS_VALID_BY_PSI=0: off).  Never set for a caller's sheet (hps_tiling_create).
    bool valid_by_psi = false;
    ~Tiling ();
};

// particles appended behind the tile-sorted body, worked on by the first `nwg` workgroups of a tile kernel's grid, 256 each
// (particles_tiled.hip::tile_record); live_n: the species' particle count on the device, or null (= hps_plasma::n)
struct TailWork { int first = 0, nwg = 0; const unsigned long long* live_n = nullptr;
                  int extra = 0; };      // extra: workgroups at the HEAD of the grid that are neither (the beam's deposition)

// int index of the per-workgroup launch records inside Tiling::offsets
__host__ __device__ inline int tile_launch_offset (int ntiles) { return (2*ntiles + 2 + 3) & ~3; }

int tiling_create (int nx, int ny, int ts, long capacity, Tiling** out);
int tiling_sort (Tiling* T, const hps_plasma& src, const hps_plasma& dst, const hps_geom& g, hipStream_t st);

} // namespace hps
#endif
