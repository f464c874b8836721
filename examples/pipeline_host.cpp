// pipeline_host.cpp -- the multi-rank C++ host above the C ABI (include/hpslice.h; no Python, no torch in the process): the
// time-step pipeline of Hipace::Evolve (Hipace.cpp:393-554: rank r runs steps r, r + N, ...) with the per-slice beam hand-off
// of MultiBuffer (utils/MultiBuffer.cpp:444-609), for a static beam (hipace.dt = 0, as every BASELINE deck).
//
//   pipeline_host <deck.bin> <n_steps> <rank> <world> <id_dir> [stages_per_rank [tile_size [sort_period]]]
//
// One process per GPU (device = rank), `stages_per_rank` engines per process (several time steps in flight per device: the
// reference runs several MPI ranks per GPU for that).  The ring has world x stages_per_rank stages; stage v = rank *
// stages_per_rank + j runs the steps v, v + W, ...  Edges inside a process are device-to-device copies on the sending
// engine's stream, ordered by events; the edge that leaves the process is the C-ABI ring (hps_ring_send_slice /
// hps_ring_recv_slice: peer copies through a shared-memory mailbox by default, RCCL with HPS_RING_EDGE=rccl), its receives
// posted a whole step ahead (MultiBuffer's unlimited max_leading_slices), which is what lets the ring close (more steps than
// stages) without a rank ever waiting for its successor.  With fewer devices than ranks the ranks share devices (rank %
// devices): the ipc edge also connects processes on ONE device.  ONE host thread drives all
// engines of the process and makes every call into the ring: a slice is enqueued up to its Bx/By norm read-back
// (hps_engine_solve_slice_begin), then the other stages get their turn, then the read-backs are awaited in the same order.
// The ids of the ring's edges travel through files in <id_dir> (no MPI in the image): rank r writes edge_<r>.id and reads
// edge_<r-1>.id.  world = 1 needs no ring at all (the last stage hands back to the first in-process).
//
// Prints one line per time step this process ran: "step <s> <checksum of every slab component>".
#include "hpslice.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <deque>
#include <vector>

#define CHECK(call) do { const int e_ = (call); if (e_ != 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, e_, hps_last_error()); return 1; } } while (0)
#define HIPCHECK(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

struct Stage {
    void* eng = nullptr;
    double* buf[2] = {nullptr, nullptr};              // beam storage of the stage's even / odd local step
    std::vector<void*> landed[2];                      // event behind the message that fills block q (null: empty block)
    int have[2] = {0, 0};                              // blocks of buf[b] handed over so far (counting the empty ones) ...
    int have_step[2] = {-1, -1};                       // ... of this time step
    std::vector<int> steps;                            // the time steps this stage runs
    size_t m = 0;                                      // index into steps
    int q = 0;                                         // next slice (from the head) of the current step
    int imported = -1;
    bool begun = false, pending = false;               // begin_step done / a slice sits between its two halves
    long n_local = 0;                                  // messages handed on in-process
};

static bool exchange_id (const std::string& dir, int rank, int world, const char* mine, char* prev)
{
    const std::string fn = dir + "/edge_" + std::to_string(rank) + ".id", tmp = fn + ".tmp";
    std::FILE* fp = std::fopen(tmp.c_str(), "wb");
    if (!fp || std::fwrite(mine, 1, HPS_RING_ID_BYTES, fp) != HPS_RING_ID_BYTES) return false;
    std::fclose(fp);
    if (std::rename(tmp.c_str(), fn.c_str()) != 0) return false;
    const std::string pf = dir + "/edge_" + std::to_string((rank + world - 1) % world) + ".id";
    for (int tries = 0; tries < 6000; ++tries) {       // up to 5 minutes
        if ((fp = std::fopen(pf.c_str(), "rb"))) {
            const size_t n = std::fread(prev, 1, HPS_RING_ID_BYTES, fp);
            std::fclose(fp);
            if (n == HPS_RING_ID_BYTES) return true;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    return false;
}

int main (int argc, char** argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: pipeline_host <deck.bin> <n_steps> <rank> <world> <id_dir> [stages_per_rank [tile [sort_period]]]\n"); return 2; }
    hps_deck deck;
    {   std::FILE* fp = std::fopen(argv[1], "rb");
        if (!fp || std::fread(&deck, 1, sizeof(deck), fp) != sizeof(deck)) { std::fprintf(stderr, "cannot read an hps_deck of %zu bytes from %s\n", sizeof(deck), argv[1]); return 2; }
        std::fclose(fp); }
    const int n_steps = std::atoi(argv[2]), rank = std::atoi(argv[3]), world = std::atoi(argv[4]);
    const std::string id_dir = argv[5];
    const int L = argc > 6 ? std::atoi(argv[6]) : 1;
    const int tile = argc > 7 ? std::atoi(argv[7]) : 16;
    const int sort_period = argc > 8 ? std::atoi(argv[8]) : 128;
    if (deck.dt != 0.0) { std::fprintf(stderr, "pipeline_host hands a static beam on (hipace.dt = 0)\n"); return 2; }
    if (n_steps < 1 || world < 1 || rank < 0 || rank >= world || L < 1) { std::fprintf(stderr, "bad arguments\n"); return 2; }
    if (world > 1) {
        // before the first HIP call: the ring's receive and send streams and the engines' streams on hardware queues of their
        // own (hps_ring_init refuses 2+ ranks otherwise), and two RCCL channels per hand-off (a receive posted ahead is a
        // kernel that holds its workgroups until the data comes).  Values the launcher exported win.
        setenv("GPU_MAX_HW_QUEUES", "8", 0);
        setenv("NCCL_MAX_P2P_NCHANNELS", "2", 0);
    }
    int ndev = 0;
    HIPCHECK(hipGetDeviceCount(&ndev));
    const int dev = rank % (ndev > 0 ? ndev : 1), nz = deck.nz, W = world*L;
    HIPCHECK(hipSetDevice(dev));

    // ---- the stages of this process
    std::vector<Stage> S((size_t)L);
    long nbeam = 0; int ncomp = 0;
    std::vector<long> off((size_t)nz + 1);
    for (int j = 0; j < L; ++j) {
        Stage& s = S[(size_t)j];
        CHECK(hps_engine_create(&deck, dev, &s.eng));
        CHECK(hps_engine_set_tiling(s.eng, tile, sort_period));
        CHECK(hps_engine_set_diagnostics(s.eng, std::getenv("PIPELINE_HOST_NO_DIAG") ? 0 : 1));
        int ng = 0; long np = 0;
        CHECK(hps_engine_info(s.eng, &ncomp, &ng, &np));
        CHECK(hps_engine_beam_info(s.eng, &nbeam, off.data()));
        for (int b = 0; b < 2; ++b) {
            const size_t bytes = sizeof(double)*(size_t)(7*nbeam > 0 ? 7*nbeam : 1);
            HIPCHECK(hipMalloc(&s.buf[b], bytes));
            HIPCHECK(hipMemset(s.buf[b], 0, bytes));
            s.landed[b].assign((size_t)nz, nullptr);
        }
        for (int st = rank*L + j; st < n_steps; st += W) s.steps.push_back(st);
        if (rank*L + j == 0) { CHECK(hps_engine_initial_beam(s.eng, s.buf[0])); s.have[0] = nz; s.have_step[0] = 0; }      // only the head stage injects the beam
    }
    HIPCHECK(hipDeviceSynchronize());

    // ---- the ring (the edge that leaves the process)
    void* ring = nullptr;
    if (world > 1) {
        char mine[HPS_RING_ID_BYTES], prev[HPS_RING_ID_BYTES];
        CHECK(hps_ring_unique_id(mine));
        if (!exchange_id(id_dir, rank, world, mine, prev)) { std::fprintf(stderr, "rank %d: no edge id from rank %d in %s\n", rank, (rank + world - 1) % world, id_dir.c_str()); return 1; }
        CHECK(hps_ring_init(rank, world, dev, prev, mine, &ring));
    }
    auto block_bytes = [&] (int q) { return (long)sizeof(double)*7*(off[(size_t)q + 1] - off[(size_t)q]); };
    // post the receives of one whole step of the first stage (a step ahead of where it is needed)
    size_t posted_m = 0;                               // local steps of stage 0 whose receives have been posted
    auto post_receives = [&] (size_t m) -> int {
        Stage& s = S[0];
        const int b = (int)(m % 2);
        for (int q = 0; q < nz; ++q) {
            s.landed[b][(size_t)q] = nullptr;
            if (block_bytes(q) > 0)
                CHECK(hps_ring_recv_slice(ring, s.buf[b] + 7*off[(size_t)q], block_bytes(q), nullptr, b*nz + q, &s.landed[b][(size_t)q]));
        }
        s.have[b] = nz; s.have_step[b] = s.steps[m];   // the events say when; the host never waits for them
        return 0;
    };

    // put_data towards the next process: on an ipc edge hps_ring_send_slice makes the HOST wait until the matching receive is
    // posted and its buffer free (include/hpslice.h) -- with several stages per thread that would hold up the siblings of the
    // last stage.  So a block that cannot go out at once waits in an outbox; hps_ring_can_send is asked at the top of every
    // round (an RCCL edge always says yes: its sends only enqueue).  The engine event a block waits for comes from a pool of 64
    // slots, so the last stage does not run more than 32 slices ahead of the outbox.
    struct Outgoing { int q, b; void* pushed; Stage* s; };
    std::deque<Outgoing> outbox;
    auto flush_outbox = [&] (bool block) -> int {
        while (!outbox.empty() && (block || hps_ring_edge_kind(ring) == 0 || hps_ring_can_send(ring) == 1)) {
            const Outgoing o = outbox.front();
            void* gone = nullptr;
            CHECK(hps_ring_send_slice(ring, o.s->buf[o.b] + 7*off[(size_t)o.q], block_bytes(o.q), o.pushed, o.b*nz + o.q, &gone));
            outbox.pop_front();
        }
        return 0;
    };

    const auto t0 = std::chrono::steady_clock::now();
    std::vector<double> sums((size_t)ncomp);
    long solved = 0, idle_rounds = 0;
    bool all_done = false;
    auto t_ring_wait = std::chrono::steady_clock::now();
    const double ring_timeout = std::getenv("HPS_RING_TIMEOUT_S") ? std::atof(std::getenv("HPS_RING_TIMEOUT_S")) : 900.0;
    while (!all_done) {
        all_done = true;
        bool worked = false, ring_waits = false;
        if (ring) { const size_t before = outbox.size(); if (flush_outbox(false)) return 1; if (outbox.size() < before) worked = true; if (!outbox.empty()) ring_waits = true; }
        // first halves: every stage that can enqueues its next slice up to the norm read-back
        for (int j = 0; j < L; ++j) {
            Stage& s = S[(size_t)j];
            if (s.m >= s.steps.size()) continue;
            all_done = false;
            const int step = s.steps[s.m], b = (int)(s.m % 2);
            const bool fed = step > 0;
            if (!s.begun) {
                if (j == 0 && world > 1) {
                    // receives a whole step ahead: this step's (if not posted yet) and the next one's
                    while (posted_m <= s.m + 1 && posted_m < s.steps.size()) { if (s.steps[posted_m] > 0) { if (post_receives(posted_m)) return 1; } ++posted_m; }
                }
                CHECK(hps_engine_set_beam_storage(s.eng, s.buf[b]));
                CHECK(hps_engine_assume_initial_beam_support(s.eng));
                CHECK(hps_engine_set_step(s.eng, step));
                CHECK(hps_engine_begin_step(s.eng));
                s.begun = true; s.q = 0; s.imported = -1;
                worked = true;
            }
            if (s.pending) continue;
            if (j == L - 1 && outbox.size() >= 32) { ring_waits = true; continue; }       // (the next process is not taking blocks: do not outrun the event pool)
            // get_data: this slice's beam and the next one's (the source of its jx, jy) must have been handed on
            const int need = s.q + 1 < nz ? s.q + 1 : nz - 1;
            if (fed && (s.have_step[b] != step || s.have[b] < need + 1)) continue;       // (an in-process edge: the stage ahead is not there yet)
            const bool from_ring = fed && j == 0 && world > 1;
            if (from_ring) {
                // an ipc edge has no event the device could wait for: ask whether the messages have landed (an RCCL edge says
                // yes at once and lets the engine's stream wait) and give the other stages their turn if not
                bool there = true;
                for (int k = s.imported + 1; k <= need && there; ++k)
                    if (void* ev = s.landed[b][(size_t)k]) there = hps_ring_recv_landed(ring, ev) == 1 || hps_ring_edge_kind(ring) == 0;
                if (!there) { ring_waits = true; continue; }
            }
            for (; s.imported < need; ++s.imported) {
                void* ev = s.landed[b][(size_t)s.imported + 1];
                if (fed && ev) { if (from_ring) CHECK(hps_ring_engine_wait(ring, s.eng, ev)); else CHECK(hps_engine_wait_event(s.eng, ev)); }
            }
            CHECK(hps_engine_solve_slice_begin(s.eng, nz - 1 - s.q));
            s.pending = true;
            worked = true;
        }
        // second halves, in the same order: wait for the norms, enqueue the rest, hand the slice's block on
        for (int j = 0; j < L; ++j) {
            Stage& s = S[(size_t)j];
            if (!s.pending) continue;
            const int step = s.steps[s.m], b = (int)(s.m % 2), q = s.q;
            CHECK(hps_engine_solve_slice_finish(s.eng, nz - 1 - q));
            s.pending = false;
            ++solved;
            if (step + 1 < n_steps) {
                // put_data: the block goes to the stage that runs step + 1
                const bool to_ring = (j == L - 1) && world > 1;
                if (to_ring) {
                    if (block_bytes(q) > 0) {
                        void* pushed = nullptr;
                        CHECK(hps_engine_record_event(s.eng, q % 64, &pushed));
                        outbox.push_back(Outgoing{q, b, pushed, &s});
                        if (flush_outbox(false)) return 1;
                    }
                } else {
                    Stage& r = S[(size_t)((j + 1) % L)];
                    // the receiver's local step that runs step + 1: the one after its current one if it is this process's
                    // first stage closing the loop, else the one with the same index
                    const size_t mr = (j == L - 1) ? s.m + 1 : s.m;
                    const int rb = (int)(mr % 2);
                    r.landed[rb][(size_t)q] = nullptr;
                    if (block_bytes(q) > 0) {
                        CHECK(hps_engine_copy_async(s.eng, r.buf[rb] + 7*off[(size_t)q], s.buf[b] + 7*off[(size_t)q], block_bytes(q)));
                        CHECK(hps_engine_record_event(s.eng, 4096 + (int)(s.n_local++ % 2048), &r.landed[rb][(size_t)q]));
                    }
                    r.have_step[rb] = step + 1;
                    r.have[rb] = q + 1;
                }
            }
            if (++s.q == nz) {
                CHECK(hps_engine_sync(s.eng));
                CHECK(hps_engine_checksums(s.eng, sums.data()));
                std::printf("step %d", step);
                for (int c = 0; c < ncomp; ++c) std::printf(" %.17g", sums[(size_t)c]);
                std::printf("\n");
                ++s.m; s.begun = false;
            }
        }
        if (worked || !ring_waits) t_ring_wait = std::chrono::steady_clock::now();
        else if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ring_wait).count() > ring_timeout) {
            std::fprintf(stderr, "rank %d: no message from the previous rank for %.0f s\n", rank, ring_timeout); return 1;
        }
        idle_rounds = (worked || ring_waits) ? 0 : idle_rounds + 1;
        if (idle_rounds > 100000) { std::fprintf(stderr, "rank %d: the stages of this process wait for one another\n", rank); return 1; }
    }
    if (ring && flush_outbox(true)) return 1;
    for (auto& s : S) CHECK(hps_engine_sync(s.eng));
    if (ring) CHECK(hps_ring_sync(ring));
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "rank %d: %ld slices in %.3f s = %.1f slices/s (%d stage(s) per rank)\n", rank, solved, sec, solved/sec, L);
    if (ring) {
        long ns = 0, nr = 0; long long bs = 0, br = 0;
        CHECK(hps_ring_stats(ring, &ns, &nr, &bs, &br));
        std::printf("ring %ld %ld %lld %lld\n", ns, nr, bs, br);
        CHECK(hps_ring_destroy(ring));
    }
    for (auto& s : S) { CHECK(hps_engine_destroy(s.eng)); for (int b = 0; b < 2; ++b) (void)hipFree(s.buf[b]); }
    return 0;
}
