// engine.h -- slice-engine state (device resident).
#ifndef HPS_ENGINE_H_
#define HPS_ENGINE_H_

#include "common.h"
#include <vector>
#include "tiling.h"
#include "mg_gate.h"
#include "particle_math.h"
#include "poisson_src.h"
#include "beam_deposit.h"
#include "slab_ops.h"

void mg_rider (void* mg_handle, int** dev_words, const int** host_words);     // multigrid.hip
namespace hps {

constexpr int PC_MAX_SPEC = 64;      // flags / host slots of the device-controlled predictor-corrector loop: iterations 1 .. 62

// hps_mg_solve1 in two halves (multigrid.hip): kernels enqueued between them are gated on mg_gate_after_enqueued
int mg_solve1_begin (void* mg_handle, hps_slab s, int sol_comp, int rhs_comp, int acf_comp, double tol_rel, double tol_abs,
                     int max_iters, hipStream_t st);
void mg_solve1_forget_hierarchy (void* mg_handle);
bool poisson_gateable (void* poisson_handle);
void poisson_set_gate (void* poisson_handle, const int* gate);
bool ring_is_ticket (const void* p);      // ring.hip: is p the receive handle of an ipc ring edge (not a hipEvent_t)?
int mg_solve1_prepare (void* mg_handle, hps_slab s, int sol_comp, int rhs_comp, int acf_comp, int max_iters, hipStream_t st);
// ... in one launch with the -grad Psi / Sx, Sy pass of the slab (multigrid.hip: k_hierarchy_gradpsi); *done = false if this grid's
// hierarchy needs more than that launch (then nothing has been enqueued)
int mg_solve1_prepare_with (void* mg_handle, hps_slab s, int sol_comp, int rhs_comp, int acf_comp, int max_iters, SlabView f, GradPsiSxSy ga,
                            hipStream_t st, bool* done);
const int* mg_gate_after_enqueued (void* mg_handle);
void mg_defer_post (void* mg_handle);
bool mg_take_deferred_post (void* mg_handle, MgPost* out);
bool mg_solve1_ready (void* mg_handle);
int mg_solve1_finish (void* mg_handle, int* iters_out, double* resnorm_out, int* extra, hipStream_t st);
}

namespace hps {

struct LaserState;      // laser.hip

struct BeamSoA { double *x, *y, *z, *ux, *uy, *uz, *w; int* nsub;      // moving beam (beam.hip); nsub < 0: absorbed
                 double *sx, *sy, *sz; };                              // spin (do_spin_tracking), null otherwise

struct Engine {
    hps_deck d{};
    hps_geom gm{};
    int g = 0, ncomp = 0;
    hipStream_t st = nullptr;
    bool st_pooled = false; int st_device = 0;       // the stream came from the per-device pool (engine.hip: create_engine_stream)
    hps_slab slab{};
    hps_plasma pl{};
    double* pl_real = nullptr;
    long np = 0;                   // particles of the first species NOW (grows when the species "ion" releases electrons)
    long np_cap = 0, np_init = 0;  // capacity of its arrays; particles InitParticles creates
    // species "ion" with ADK field ionisation (ionization.hip): tile-sorted once per step (ions hardly move); released
    // electrons are appended to `pl` behind the tile-sorted body and run through the per-particle kernels until the next
    // re-sort.  Counters on the device {electrons, overflow, blocks done, ionised}, the count comes back through mapped
    // host memory (no stream synchronisation).
    struct Ions { hps_plasma pl{}, pl_alt{}; double *real = nullptr, *real_alt = nullptr; Tiling* tiling = nullptr; long n = 0;
                  double* d_adk = nullptr; unsigned long long* d_cnt = nullptr; long long* h_cnt = nullptr; long long* h_cnt_dev = nullptr;
                  long long seq = 0; long n_ionized = 0; bool pending = false; int* d_tile_flag = nullptr;
                  double* d_fbound = nullptr;       // [5][tiles] field maxima of the slice (k_ion_field_bounds; HPS_ION_TILE_SKIP=0: off)
                } ion;
    // tabulated plasma density profile (hps_engine_set_density_profile): radial table on the device, time table on the host
    std::vector<double> prof_r, prof_t, prof_f_t; double* d_prof_r = nullptr; double prof_ft = 1.0;
    // fused push(k) + deposit(k-1) (k_advance_deposit_tiled): ahead_for = slice whose plasma currents are already deposited
    bool fuse_push_deposit = false; int ahead_for = -2;
    long fallback_div = 256;                    // re-sort once more than np / fallback_div particle visits since the last sort left their tile's halo (HPS_SORT_FALLBACK_DIV)
    bool gate_push = true;                      // push enqueued behind the multigrid's V-cycles, gated on its stopping rule (HPS_GATED_PUSH=0: off)
    int step_index = -1;           // physical time step that has begun (density profile's time factor, ionisation draws)
    int next_step = -1;            // hps_engine_set_step: the step the next begin_step starts (-1: the one after step_index)
    IonArgs ion_args (int islice);             // ionization.hip: kernel arguments of this slice's ionisation
    int ionize_slice (int islice);             // ...: launch of the per-particle form
    int ionize_collect ();                     // ...: wait for the electron count of the slice
    int ion_field_bounds ();                   // ...: per-tile field maxima for the tile skip of the ions' push
    hps_plasma tail_of (const hps_plasma& p, long first, long n) const;
    bool fold_tail = true;                     // the particles behind the tile-sorted body ride in the tile kernels' launches (HPS_FOLD_TAIL=0: off)
    TailWork fold_tail_of (const hps_plasma& p, const Tiling* T, long margin, long* covered) const;
    int species_deposit (const hps_plasma& p, Tiling* T, const int comp[6], double charge, double mass, int can_ionize, const BeamPairWork* beam = nullptr);
    bool valid_by_w = true;                    // HPS_VALID_BY_W=0: off.  The depositions take "weight != 0" for the valid bit of idcpu and do not read it (PartConsts::valid_by_w:
                                               // 33.5 MB less per kernel at 1024^2 x 4 ppc -- no time gained with one stage on the GPU, +1 % with three in flight: r04g_ab_inflight_byte_cuts.txt)
    bool post_in_push = true;                  // the gated plasma push posts the Bx/By solve's norms to the host (HPS_POST_IN_PUSH=0: k_post_norms, a launch of its own)
    bool fold_hierarchy = true;                // the multigrid's coefficient hierarchy in the -grad Psi / Sx, Sy launch (HPS_FOLD_HIERARCHY=0: in mg_solve1_begin)
    bool fold_beam = true;                     // the static beam's two deposits of a slice as extra workgroups of the plasma's deposition (HPS_FOLD_BEAM=0: a launch of their own)
    int species_explicit (const hps_plasma& p, Tiling* T, const int cache[4], const int depos[2], double charge, double mass, int can_ionize);
    int species_advance (const hps_plasma& p, Tiling* T, const int comp[5], double charge, double mass, int temp_slice, int can_ionize);
    // tile-sorted sheet (sort.hip): second SoA buffer + tiling state
    Tiling* tiling = nullptr; int tile_size = 16, sort_period = 128, since_sort = 0;
    hps_plasma pl_alt{}; double* pl_real_alt = nullptr;
    int* d_nfallback = nullptr;                 // word 0 of the multigrid norm buffer's header slot (mg_rider)
    const int* h_nfallback = nullptr;           // its pinned image, refreshed by every multigrid solve
    long fb_at_sort = 0; int n_sorts = 0;       // adaptive re-sort: fallbacks at the last sort, number of sorts
    int setup_tiling ();
    int resort ();
    void* ps = nullptr;            // Poisson solver handle
    void* mg = nullptr;            // multigrid handle
    double* staging = nullptr;
    double* d_open_mom = nullptr;      // boundary.field = Open: the multipole moments of up to four sources (Engine::open_boundary)
    int open_boundary (int nbatch, unsigned no_monopole_mask);
    // driver beam: slice-major SoA (head slice first), static because hipace.dt = 0
    // blocks [slice p from the head][7][count_p]; beam_cur = storage in use (own or caller's)
    double* beam_data = nullptr; double* beam_init = nullptr; double* beam_cur = nullptr;
    long nbeam = 0; std::vector<long> beam_off;
    // hipace.dt != 0 (beam.hip): global SoA + device-resident slice boundaries B[nz+1] and slipped-front counts
    bool moving = false; int steps_begun = 0; int beam_rows = 7;     // rows of a hand-off message (10 with spin)
    BeamSoA bm{}, bm_scr{}; double* bm_store = nullptr; int* bm_nsub = nullptr; int* bm_nsub_scr = nullptr;
    long* d_B = nullptr; int* d_nfront = nullptr; std::vector<long> h_B;      // h_B: boundaries as of begin_step
    // ring hand-off: import mode (the slices of the coming step arrive as messages), start offsets of the imported
    // blocks, per-message capacity, overflow counter
    bool beam_import = false; long* d_Bimp = nullptr; long beam_cap = 0; int* d_beam_overflow = nullptr;
    // upper bound of slice p's size during this step.  A particle may slip through SEVERAL slices in one step (it is
    // pushed again on every slice it lands on, BeamParticleAdvance.cpp:131 "IncludingSlipped"), so everything ahead of
    // slice p counts: own + all of slices 0..p-1 as of begin_step.  The kernels loop grid-stride over the device-side
    // count, so the bound only sizes (and caps) the launch.
    long beam_bound (int p) const {
        if (p < 0 || p >= d.nz) return 0;
        if (beam_import) return std::max(nbeam, 1L);
        return h_B[p + 1] - h_B[0];
    }
    // support of the beam currents in padded-array cells (deposit footprint + the centred differences taken of
    // them); the beam planes are identically zero outside, so the slab kernels skip them there
    struct Box { int ilo, ihi, jlo, jhi; };
    Box beam_box{0, -1, 0, -1}, beam_box_init{0, -1, 0, -1}, full_box{0, -1, 0, -1};
    int* d_nqsa = nullptr;
    double* d_checksum = nullptr;
    bool diagnostics = false;
    bool profiling = false; bool prof_light = false; int prof_stride = 1; bool prof_now = false;      // events on every prof_stride-th slice
    std::vector<hipEvent_t> ev; size_t ev_used = 0;      // 11 events per profiled slice
    std::vector<hipEvent_t> hand_ev;                     // hps_engine_record_event pool (no timing)
    void mark ();
    long total_vcycles = 0, slices_done = 0;
    // predictor-corrector Bx/By (hipace.bxby_solver = predictor-corrector): d_pc = {sum |B|, sum |B - B_iter|, halo
    // fallback counter (int), spare}, h_pc its pinned image read back once per iteration
    // device-side control of the loop (HPS_PC_SPECULATE=0: off): per-iteration flags, iterations enqueued ahead of the host
    int* d_pc_dist = nullptr;  // predictor-corrector: device word "something other than rounding residue has been deposited in this sweep" (Engine::create)
    double pc_floor = 0.0;     // sum |B| below this is the exact zero of the serial path (Engine::create)
    int* d_pc_go = nullptr; bool pc_speculate = false; int pc_spec_iters = 1, pc_enqueued = 0, pc_islice = -1; double pc_base_seq = 0.0, pc_last_err = 0.0;
    int solve_slice_pc_begin (int islice); int solve_slice_pc_finish (int islice); int pc_enqueue_iteration (int it); int pc_wait_slot (int slot, double seq);
    bool pc = false; double* d_pc = nullptr; double* d_pc_aux = nullptr; double* h_pc = nullptr; double* h_pc_dev = nullptr; double pc_seq = 0.0; long pc_iterations = 0; double pc_err_sum = 0.0; long pc_zero_b_slices = 0;
    double pc_tol = 4e-2, pc_mix = 0.05; int pc_max_iter = 30;
    int c_aabs = -1; double* d_laser_sum = nullptr;       // laser: slab component of |a|^2, device sum of |a| (diagnostics)
    LaserState* laser = nullptr;                          // envelope arrays + solver (laser.hip)
    // The envelope's advance of a slice needs chi of the slice and the envelope of the slices before it -- nothing else
    // of the slice needs its result.  It runs on a stream of its own beside the explicit deposition, the Bx/By multigrid
    // and the push (HPS_LASER_ASYNC=0: on the engine's stream, in the reference's place, Hipace.cpp:637).
    hipStream_t st_laser = nullptr; hipEvent_t ev_lfork = nullptr, ev_ldone = nullptr;
    bool laser_async = true, laser_pending = false;
    // Small kernels of a slice that wait for nothing on its critical path run on a stream of their own beside it: the beam's
    // deposition (needs the zeroed beam planes; wanted by the Sx/Sy initialisation behind the Poisson solves) and the
    // multigrid's coefficient hierarchy (needs chi; wanted by the Bx/By solve).  Measured: 1453-1458 slices/s against 1452-1470 without -- the two event waits cost what the
    // 14 us of kernels off the chain save -- so this is off unless HPS_AUX_STREAM=1
    hipStream_t st_aux = nullptr; hipEvent_t ev_aux[3] = {nullptr, nullptr, nullptr}; bool aux_on = false, aux_pending = false;
    hipStream_t laser_stream () const { return (laser_async && st_laser) ? st_laser : st; }
    int fork_laser ();          // the laser stream may read what the engine's stream has written so far
    int laser_done ();          // mark the end of the laser stream's work of this slice
    int join_laser ();          // the engine's stream waits for the laser stream's last slice (no-op if none is pending)
    // field diagnostic (Fields::Copy): components, coarsening, device array [ncomps][nzc][nyc][nxc]
    std::vector<int> fd_comps; int fd_c[3] = {1, 1, 1}; double* d_fd = nullptr; int* d_fd_comps = nullptr;
    // geometry of the diagnostic grid (Diagnostic::ResizeFDiagFAB + TrimIOBox, diagnostics/Diagnostic.cpp:300-410): cells, cell
    // size, position of local cell 0 (GetPosOffset of the diagnostic geometry + first index * cell size), real box, slice direction
    int fd_n[3] = {0, 0, 0}; double fd_h[3] = {0, 0, 0}, fd_pos0[3] = {0, 0, 0}, fd_lo[3] = {0, 0, 0}, fd_hi[3] = {0, 0, 0}; int fd_slice_dir = -1;
    size_t fd_cells () const { return (size_t)fd_n[0]*fd_n[1]*fd_n[2]; }
    int fill_field_diagnostic (int islice);
    double* d_insitu_bm = nullptr; double insitu_bm_radius = 0.0;     // [23][nz] raw sums of the beam moments
    void insitu_beam (int islice);
    double* d_insitu_pl = nullptr; double insitu_pl_radius = 0.0;     // [15][nz] raw sums of the plasma moments
    double* d_insitu = nullptr;      // [10][nz] in-situ field reductions (Fields::InSituComputeDiags)

    ~Engine ();
    int create (const hps_deck& deck, int device);
    int init_beam ();
    int install_beam (std::vector<double> (&h)[7]);
    int set_beam_particles (long n, const double* soa_host, long* n_outside);
    int begin_step ();
    int deposit_beam_slice (int islice, int cjx, int cjy, int cjz, const int* go = nullptr);
    void deposit_grid_current (int islice, int cjz);
    int solve_slice (int islice);
    int solve_slice_begin (int islice);      // ... in two halves: everything up to the Bx/By solve's norm read-back is enqueued,
    int solve_slice_finish (int islice);     // then the host waits for the norms and enqueues the rest
    int pending_slice = -1; bool pend_fuse = false, pend_gated = false, pend_gated_ion = false;
    bool gate_ion_push = true; IonArgs pend_ia{}; long pend_covered = 0;      // the ionisable species' push + the electrons' push behind the V-cycles (HPS_GATED_ION_PUSH=0: off)
    int push_with_ionization (int islice, const int comp[5], const int* go, bool first);
    bool lazy_shift = true, shift_pending = false;      // ShiftSlices deferred to the next slice's InitializeSlices pass (HPS_LAZY_SHIFT=0: off)
    void flush_shift ();
    bool fuse_sources = true;       // the Poisson sources formed inside the first transform pass (HPS_FUSE_SOURCES=0: k_rhs_all + staging planes)
    int run_step ();
};

int ion_create (Engine& E);                                                  // ionization.hip
void ion_destroy (Engine& E);
unsigned event_flags (bool timing);                                           // engine.hip
int laser_create (Engine& E);                                                // laser.hip
void laser_destroy (Engine& E);
int laser_begin_step (Engine& E);
int laser_update_aabs (Engine& E, int islice, double* sum_abs);
int laser_advance_slice (Engine& E, int islice);
int laser_copy_envelope (Engine& E, double* out_host);
long laser_mg_vcycles (Engine& E);
int laser_set_import (Engine& E, int on, int step);
int laser_export_slice (Engine& E, int islice, double* msg_dev);
int laser_import_slice (Engine& E, int islice, const double* msg_dev);
int laser_import_from (Engine& E, int islice, Engine& src);
int beam_deposit_moving (Engine& E, int p, int cjx, int cjy, int cjz);      // beam.hip
int beam_push_moving (Engine& E, int islice);
int beam_export_slice (Engine& E, int islice, double* msg_dev, long cap);
int beam_import_slice (Engine& E, int islice, const double* msg_dev, long cap);

} // namespace hps
#endif
