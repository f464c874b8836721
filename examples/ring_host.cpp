// ring_host.cpp -- a C++ host above the C ABI (include/hpslice.h), no Python, no torch: what a maintainer of the reference
// would write in Hipace::Evolve / MultiBuffer (Hipace.cpp:400-471, utils/MultiBuffer.cpp:287-609) with the slice engine and
// the ring transport in place of AMReX's loops.  One rank that is its own ring neighbour (MultiBuffer.cpp:299-308 with the
// transport left in): every pushed beam slice of step m goes through RCCL into the storage step m+1 reads, ordered against
// the engine by events only -- the schedule of hipace_amd/pipeline.py::run_pipeline for a static beam (hipace.dt = 0).
//
//   ring_host <deck.bin> <n_steps> [tile_size [sort_period]]      deck.bin = the bytes of an hps_deck (tests write it with ctypes)
//
// Prints one line per time step: "step <m> <name>=<checksum> ..." with the components in slab order.
#include "hpslice.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(call) do { const int e_ = (call); if (e_ != 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, e_, hps_last_error()); return 1; } } while (0)
#define HIPCHECK(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

int main (int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: ring_host <deck.bin> <n_steps> [tile_size [sort_period]]\n"); return 2; }
    hps_deck deck;
    {   std::FILE* fp = std::fopen(argv[1], "rb");
        if (!fp || std::fread(&deck, 1, sizeof(deck), fp) != sizeof(deck)) { std::fprintf(stderr, "cannot read an hps_deck of %zu bytes from %s\n", sizeof(deck), argv[1]); return 2; }
        std::fclose(fp); }
    const int n_steps = std::atoi(argv[2]);
    const int tile = argc > 3 ? std::atoi(argv[3]) : 16;
    const int sort_period = argc > 4 ? std::atoi(argv[4]) : 128;
    if (deck.dt != 0.0) { std::fprintf(stderr, "ring_host hands a static beam on (hipace.dt = 0)\n"); return 2; }
    const int dev = 0, nz = deck.nz;
    HIPCHECK(hipSetDevice(dev));

    void* eng = nullptr;
    CHECK(hps_engine_create(&deck, dev, &eng));
    CHECK(hps_engine_set_tiling(eng, tile, sort_period));
    CHECK(hps_engine_set_diagnostics(eng, std::getenv("RING_HOST_NO_DIAG") ? 0 : 1));      // (checksums cost a kernel per slice)
    int ncomp = 0, ng = 0; long np = 0;
    CHECK(hps_engine_info(eng, &ncomp, &ng, &np));

    // beam blocks: slice q (from the head) owns [7 off[q], 7 off[q+1]) of the SoA; two steps' worth of storage
    long nbeam = 0;
    std::vector<long> off((size_t)nz + 1);
    CHECK(hps_engine_beam_info(eng, &nbeam, off.data()));
    double* buf[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; ++b) {
        HIPCHECK(hipMalloc(&buf[b], sizeof(double)*(size_t)(7*nbeam > 0 ? 7*nbeam : 1)));
        HIPCHECK(hipMemset(buf[b], 0, sizeof(double)*(size_t)(7*nbeam > 0 ? 7*nbeam : 1)));
    }
    CHECK(hps_engine_initial_beam(eng, buf[0]));          // the head rank injects the beam

    char id[HPS_RING_ID_BYTES];
    void* ring = nullptr;
    CHECK(hps_ring_unique_id(id));
    CHECK(hps_ring_init(0, 1, dev, nullptr, id, &ring));

    // landed[b][q]: event behind the message that fills block q of buf[b] (null: nothing was sent, the block is empty)
    std::vector<void*> landed[2] = {std::vector<void*>((size_t)nz, nullptr), std::vector<void*>((size_t)nz, nullptr)};
    std::vector<double> sums((size_t)ncomp);
    for (int m = 0; m < n_steps; ++m) {
        const int cur = m % 2, nxt = (m + 1) % 2;
        const auto t0 = std::chrono::steady_clock::now();
        CHECK(hps_engine_set_beam_storage(eng, buf[cur]));
        CHECK(hps_engine_assume_initial_beam_support(eng));
        CHECK(hps_engine_begin_step(eng));
        int imported = -1;
        for (int q = 0; q < nz; ++q) {
            // get_data: this slice's beam and the next one's (the source of its jx, jy) must have landed
            const int need = q + 1 < nz ? q + 1 : nz - 1;
            for (; imported < need; ++imported) {
                void* ev = landed[cur][(size_t)imported + 1];
                if (m > 0 && ev) CHECK(hps_engine_wait_event(eng, ev));
            }
            CHECK(hps_engine_solve_slice(eng, nz - 1 - q));
            // put_data: the slice's block goes to the next step's storage behind the engine's event
            const long n = off[(size_t)q + 1] - off[(size_t)q];
            landed[nxt][(size_t)q] = nullptr;
            if (m + 1 < n_steps && n > 0) {
                void* pushed = nullptr;
                CHECK(hps_engine_record_event(eng, q % 64, &pushed));
                CHECK(hps_ring_sendrecv_self(ring, buf[cur] + 7*off[(size_t)q], buf[nxt] + 7*off[(size_t)q], (long)sizeof(double)*7*n,
                                             pushed, cur*nz + q, &landed[nxt][(size_t)q]));
            }
        }
        CHECK(hps_engine_sync(eng));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, "step %d: %.3f ms per slice, %.1f slices/s\n", m, ms/nz, 1e3*nz/ms);
        CHECK(hps_engine_checksums(eng, sums.data()));
        std::printf("step %d", m);
        for (int c = 0; c < ncomp; ++c) std::printf(" %.17g", sums[(size_t)c]);
        std::printf("\n");
    }
    long ns = 0, nr = 0; long long bs = 0, br = 0;
    CHECK(hps_ring_sync(ring));
    CHECK(hps_ring_stats(ring, &ns, &nr, &bs, &br));
    std::printf("ring %ld %ld %lld %lld\n", ns, nr, bs, br);
    CHECK(hps_ring_destroy(ring));
    CHECK(hps_engine_destroy(eng));
    for (int b = 0; b < 2; ++b) (void)hipFree(buf[b]);
    return 0;
}
