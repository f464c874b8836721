/* hpslice.h -- C ABI of the MI355X-native quasi-static PIC slice engine (libhpslice.so).
 *
 * Drop-in boundary for the per-zeta-slice hot path of HiPACE++ (reference paths relative to
 * /root/reference/src).  The reference has no FFI layer; each entry point below replaces one of
 * its C++ operator seams and takes the same data the reference operator touches, as plain
 * device pointers + sizes:
 *
 *   hps_slab    <-> amrex::MultiFab m_slices[lev] viewed as Array3 (utils/GPUUtil.H:99-147,
 *                   fields/Fields.H:465): ncomp planes of (nx+2ng) x (ny+2ng) doubles, x fastest.
 *   hps_plasma  <-> PlasmaParticleContainer pure-SoA tile (particles/plasma/
 *                   PlasmaParticleContainer.H:21-50): 11 real arrays + idcpu + ion_lev.
 *   hps_geom    <-> amrex::Geometry of the slice + PhysConst (utils/Constants.H:39-81) +
 *                   particle boundary (particles/pusher/GetAndSetPosition.H:29-99).
 *
 * All pointers are DEVICE pointers (HBM) unless a parameter is named *_host.  Every call is
 * asynchronous on the caller's hipStream_t (passed as void*), returns an int status
 * (HPS_OK = 0) instead of aborting, and never allocates unless documented.
 * INTEGRATION.md shows the AMReX-side adaptor for each call.
 */
#ifndef HPSLICE_H_
#define HPSLICE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hps_stream;            /* hipStream_t */

enum { HPS_OK = 0, HPS_ERR_ARG = 1, HPS_ERR_HIP = 2, HPS_ERR_FFT = 3, HPS_ERR_MG_DIVERGED = 4,
       HPS_ERR_MG_MAXITER = 5, HPS_ERR_COMM = 6, HPS_ERR_UNSUPPORTED = 7 };

/* particle boundary, Hipace.H ParticleBoundary */
enum { HPS_BC_REFLECTING = 0, HPS_BC_PERIODIC = 1, HPS_BC_ABSORBING = 2 };

/* Field slab view. element (i,j,n), i in [-ng, nx+ng): p[(i+ng) + (j+ng)*jstride + n*nstride] */
typedef struct {
    double* p;
    int nx, ny, ng, ncomp;
    long jstride, nstride;
} hps_slab;

/* Plasma sheet, pure SoA.  idcpu follows AMReX's packing: bit 63 set = valid particle
 * (ParticleIDWrapper::is_valid), low 24 bits = cpu = mesh-refinement level (0 here). */
typedef struct {
    double *x, *y, *w, *ux, *uy, *psi, *x_prev, *y_prev, *ux_half, *uy_half, *psi_half;
    uint64_t* idcpu;
    int32_t* ion_lev;
    long n;
} hps_plasma;
#define HPS_ID_VALID (1ULL << 63)

typedef struct {
    double dx, dy, dz;           /* cell sizes of the slice geometry                         */
    double xoff, yoff;           /* GetPosOffset(0|1, geom, slab box) (fields/Fields.H:71-77) */
    double c, ep0, mu0, q_e, m_e;/* PhysConst (1 in normalised units)                        */
    double plo[2], phi[2];       /* particle boundary box (Hipace.cpp:219-222)               */
    int bc;                      /* HPS_BC_*                                                 */
    int normalized;              /* hipace.normalized_units                                  */
} hps_geom;

const char* hps_last_error (void);
const char* hps_version (void);

/* ---- particle operators ------------------------------------------------------------------ */

/* DepositCurrent (particles/deposition/PlasmaDepositCurrent.H:28-32, .cpp:22-257).
 * comp = {jx, jy, jz, rho, chi, rhomjz}, -1 = do not deposit.  QSA-violating particles are
 * invalidated (w = 0, idcpu valid bit cleared) and counted into *n_qsa_violation (device int,
 * may be NULL).  `charge` is the species charge (pass -charge for WhichSlice::RhomJzIons). */
int hps_deposit_current (hps_slab slab, hps_plasma plasma, hps_geom geom, const int comp[6],
                         double charge, double mass, int depos_order, double max_qsa_weighting,
                         int can_ionize, int* n_qsa_violation, hps_stream stream);

/* ExplicitDeposition (particles/deposition/ExplicitDeposition.H:20-22, .cpp:20-263).
 * cache = {Bz, Ez, ExmBy, EypBx} (read per cell), depos = {Sy, Sx}. */
int hps_explicit_deposit (hps_slab slab, hps_plasma plasma, hps_geom geom, const int cache[4],
                          const int depos[2], double charge, double mass, int depos_order,
                          int derivative_type, int can_ionize, hps_stream stream);

/* AdvancePlasmaParticles, leapfrog pusher (particles/pusher/PlasmaParticleAdvance.H:23-26,
 * .cpp:29-305).  comp = {Psi, Ez, Bx, By, Bz}. */
int hps_advance_plasma (hps_slab slab, hps_plasma plasma, hps_geom geom, const int comp[5],
                        double charge, double mass, int depos_order, int temp_slice,
                        int n_subcycles, int can_ionize, hps_stream stream);

/* The same three operators with a laser envelope (use_laser = true of the reference's compile-time options):
 * aabs_comp = slab component holding |a|^2 (Comps[This]["aabs"]); -1 = no laser = the calls above. */
int hps_deposit_current_laser (hps_slab slab, hps_plasma plasma, hps_geom geom, const int comp[6], int aabs_comp,
                               double charge, double mass, int depos_order, double max_qsa_weighting,
                               int can_ionize, int* n_qsa_violation, hps_stream stream);
int hps_explicit_deposit_laser (hps_slab slab, hps_plasma plasma, hps_geom geom, const int cache[4], int aabs_comp,
                                const int depos[2], double charge, double mass, int depos_order,
                                int derivative_type, int can_ionize, hps_stream stream);
int hps_advance_plasma_laser (hps_slab slab, hps_plasma plasma, hps_geom geom, const int comp[5], int aabs_comp,
                              double charge, double mass, int depos_order, int temp_slice,
                              int n_subcycles, int can_ionize, hps_stream stream);

/* ---- tile-sorted sheet: LDS-tile variants of the three operators -------------------------- */

/* Reorder hook (PlasmaParticleContainer::ReorderParticles, particles/plasma/
 * PlasmaParticleContainer.cpp:196-208 -> amrex::SortParticlesForDeposition): STABLE sort of the
 * sheet by tile_size x tile_size-cell tiles of the particle's nearest cell (invalid particles
 * last).  src is read, dst (a second SoA of >= src.n entries) receives the permuted sheet; the
 * tiling keeps the per-tile offsets.  tile_size is 16 or 32. */
int hps_tiling_create (int nx, int ny, int tile_size, long max_particles, void** handle);
int hps_reorder_particles (void* tiling, hps_plasma src, hps_plasma dst, hps_geom geom,
                           hps_stream stream);
int hps_tiling_info (void* tiling, int* ntiles, const int** offsets_dev, const unsigned int** perm_dev);
int hps_tiling_destroy (void* tiling);

/* BoxSorter::sortParticlesByBox (particles/sorting/BoxSort.cpp:14-78; index_type = unsigned long long, BoxSort.H:21):
 * beam particles -> longitudinal boxes (slices).  box = (int)((z - plo_z)/dz), out of range -> the extra box
 * num_boxes; counts and offsets have num_boxes + 1 entries (offsets = exclusive scan of counts), perm[new] = old,
 * particles keep their order inside a box (the serial CPU result of the reference).  Allocates scratch and
 * synchronises the stream, like the reference; an initialisation-time operator. */
int hps_beam_sort_by_box (const double* z_dev, long n, double plo_z, double dz, int num_boxes,
                          unsigned long long* counts_dev, unsigned long long* offsets_dev,
                          unsigned long long* perm_dev, hps_stream stream);

/* Same operators, same arithmetic, for a sheet ordered by hps_reorder_particles: one workgroup
 * per tile accumulates / gathers through an LDS image of the tile (+6-cell halo).  Particles
 * that drifted out of the halo since the sort take the global-memory path and are counted into
 * *n_fallback (device int, may be NULL); results do not depend on how stale the ordering is. */
int hps_deposit_current_tiled (hps_slab slab, hps_plasma plasma, hps_geom geom, const int comp[6],
                               double charge, double mass, int depos_order, double max_qsa_weighting,
                               int can_ionize, int* n_qsa_violation, void* tiling, int* n_fallback,
                               hps_stream stream);
int hps_explicit_deposit_tiled (hps_slab slab, hps_plasma plasma, hps_geom geom, const int cache[4],
                                const int depos[2], double charge, double mass, int depos_order,
                                int derivative_type, int can_ionize, void* tiling, int* n_fallback,
                                hps_stream stream);
int hps_advance_plasma_tiled (hps_slab slab, hps_plasma plasma, hps_geom geom, const int comp[5],
                              double charge, double mass, int depos_order, int temp_slice,
                              int n_subcycles, int can_ionize, void* tiling, int* n_fallback,
                              hps_stream stream);

/* ---- transverse field solvers ------------------------------------------------------------ */

/* FFTPoissonSolverDirichletFast (fields/fft_poisson_solver/FFTPoissonSolverDirichletFast.H:30-32,
 * .cpp:195-328): Lap(F) = S on an nx x ny box with F = 0 one cell outside.  `staging` is the
 * caller-filled nx*ny source (x fastest, no guards; FFTPoissonSolver.H:26-57 StagingArea());
 * the solution is written into component dst_comp of dst (valid cells only). Creation allocates
 * device scratch and rocFFT plans. */
int hps_poisson_create (int nx, int ny, double dx, double dy, void** handle);
int hps_poisson_solve (void* handle, const double* staging, hps_slab dst, int dst_comp,
                       hps_stream stream);
/* nbatch (<= 4) independent solves in one go: sources stacked in `staging` as nbatch planes of
 * nx*ny, solutions into components dst_comps[b].  The three solves of a slice (Psi, Ez, Bz) share
 * their launches this way. */
int hps_poisson_solve_batch (void* handle, int nbatch, const double* staging, hps_slab dst,
                             const int* dst_comps, hps_stream stream);
int hps_poisson_destroy (void* handle);

/* hpmg::MultiGrid, system type 1 (mg_solver/HpMultiGrid.H:48,64-66; .cpp:1169-1190,1307-1427):
 * solves -acoef*sol + Lap(sol) = rhs, homogeneous Dirichlet.  sol (in: initial guess, out:
 * solution) and rhs are 2 ADJACENT components starting at sol_comp / rhs_comp; acoef is 1
 * component.  Blocks the host until converged (the stopping rule needs the residual norm);
 * *iters_host receives the number of V-cycles. */
/* slab.nstride must be below 2^27 doubles (1 GB planes: the kernels address cells by 32-bit byte offsets from uniform
 * component bases) and slab.ng at most 8: hps_mg_solve1 / hps_mg_solve1_fabs refuse anything else. */
int hps_mg_create (int nx, int ny, double dx, double dy, void** handle);
int hps_mg_solve1 (void* handle, hps_slab slab, int sol_comp, int rhs_comp, int acoef_comp,
                   double tol_rel, double tol_abs, int max_iters, int* iters_host,
                   double* resnorm_host, hps_stream stream);
/* the same with the reference's own argument list (HpMultiGrid.H:64-66: FArrayBox& sol, FArrayBox const& rhs,
 * FArrayBox const& acoef): three views that need not share a slab -- sol2 and rhs2 with (at least) two components,
 * acoef1 with one; component 0 of each view is used (point .p at the first component wanted) */
int hps_mg_solve1_fabs (void* handle, hps_slab sol2, hps_slab rhs2, hps_slab acoef1, double tol_rel, double tol_abs,
                        int max_iters, int* iters_host, double* resnorm_host, hps_stream stream);
int hps_mg_destroy (void* handle);

/* hpmg::MultiGrid, system type 2 (mg_solver/HpMultiGrid.H:48,84-87 solve2 with an array Re and a scalar Im coefficient;
 * .cpp:296-334 gs2, :192-208 residual2r/2i, :1239-1262), the envelope solve of MultiLaser::AdvanceSliceMG:
 *     -(ar + i ai)(sol_r + i sol_i) + Lap(sol_r + i sol_i) = rhs_r + i rhs_i,   homogeneous Dirichlet,
 * on a cell-centred nx x ny box (even nx, ny).  sol2 / rhs2: device arrays [2][ny][nx] (Re plane, Im plane) without guard
 * cells, sol2 in: initial guess, out: solution; acoef_real_dev [ny][nx]; acoef_imag_dev: ONE double on the device (the
 * caller's kernels produce it).  Blocks the host until converged; HPS_ERR_MG_MAXITER / HPS_ERR_MG_DIVERGED where hpmg aborts. */
int hps_mg2_create (int nx, int ny, double dx, double dy, void** handle);
int hps_mg2_solve2 (void* handle, double* sol2_dev, const double* rhs2_dev, const double* acoef_real_dev,
                    const double* acoef_imag_dev, double tol_rel, double tol_abs, int max_iters, int* iters_host,
                    double* resnorm_host, hps_stream stream);
int hps_mg2_destroy (void* handle);

/* ---- slice engine (Hipace::Evolve / SolveOneSlice, explicit solver; Hipace.cpp:393-728) -- */

#define HPS_MAX_ION_LEVELS 56
typedef struct {
    int nx, ny, nz; double lo[3], hi[3];
    int order; int deriv_type;
    int plasma_ppc[2]; double plasma_density; double plasma_radius;
    double plasma_charge, plasma_mass; double max_qsa; int n_subcycles;
    int beam_profile;                     /* -1 none, 0 gaussian, 1 flattop */
    double beam_zmin, beam_zmax, beam_radius, beam_density;
    double beam_umean[3], beam_pos_mean[3], beam_pos_std[3]; int beam_ppc[3]; double beam_charge;
    int bc; double mg_tol_rel, mg_tol_abs; int deposit_rho; int n_steps;
    /* moving driver beam: hipace.dt (0 = the beam is never pushed, every BASELINE deck),
     * beam.n_subcycles (0 -> 10, BeamParticleContainer.H:222), beam mass (0 -> 1) and a linear focusing field
     * beams.external_E(x,y,z,t) = ext_E_slope[0]*x  ext_E_slope[1]*y  0 (ExternalFields.H:29-56) */
    double dt; int beam_n_subcycles; double beam_mass; double ext_E_slope[2];
    /* hipace.bxby_solver: 0 explicit (Hipace.H:244), 1 predictor-corrector (Hipace.cpp:935-1031) with
     * hipace.predcorr_B_error_tolerance (0 -> 4e-2), predcorr_max_iterations (0 -> 30), predcorr_B_mixing_factor
     * (0 -> 0.05) (Hipace.H:210-222).  field_bc: boundary.field, 0 Dirichlet, 1 Open (Fields::SetBoundaryCondition,
     * fields/Fields.cpp:678-735: Psi, Ez, Bz -- and Bx, By of the predictor-corrector loop -- solved with the free-space potential
     * of the sources, expanded to order 18 about x = y = 0, as Dirichlet values; x = y = 0 must lie inside the box). */
    int bxby_solver; double predcorr_tol; int predcorr_max_iter; double predcorr_mix; int field_bc;
    /* a Gaussian laser envelope (laser/Laser.H:32-45; CEP 0, no propagation angle) drives the wake:
     * |a|^2 goes into the slab component "aabs" (appended last) and enters the deposition, the explicit source and the
     * pusher.  laser_solver = 1 ("fft", MultiLaser::AdvanceSliceFFT) advances the envelope by hipace.dt every step
     * (laser_use_phase = lasers.use_phase); 2 ("multigrid", MultiLaser::AdvanceSliceMG, hpmg system type 2) likewise;
     * 0 keeps it static.  Explicit solver only; LDS-tile and per-particle kernels. */
    int laser_on; double laser_a0, laser_w0, laser_L0, laser_lambda0, laser_pos[3];
    double laser_zfoc; int laser_solver; int laser_use_phase;
    /* hipace.normalized_units = 0: the constants of utils/Constants.H:15-24 (2018 CODATA), charges in C, masses in kg,
     * densities in m^-3, lengths in m; particle weights are then numbers of particles (scale_fac = dx dy dz / ppc) */
    int si_units;
    /* grid_current.* (utils/GridCurrent.cpp:13-23): a Gaussian current density added to jz_beam (explicit solver) or jz
     * (predictor-corrector) of every slice, GridCurrent::DepositCurrentSlice (:25-71, called at Hipace.cpp:629) */
    int grid_current_on; double grid_current_peak, grid_current_mean[3], grid_current_std[3];
    /* laser_solver = 2: lasers.solver_type = multigrid (MultiLaser::AdvanceSliceMG, laser/MultiLaser.cpp:430-608: hpmg
     * system type 2, MG_average_rhs = 1, at most 200 V-cycles); lasers.MG_tolerance_rel (0 -> 1e-4) / MG_tolerance_abs */
    double laser_mg_tol_rel, laser_mg_tol_abs;
    /* <beam>.do_radiation_reaction (classical Landau-Lifshitz force in the beam push, particles/pusher/
     * BeamParticleAdvance.cpp:244-297; in normalised units it needs hipace.background_density_SI to convert the fields)
     * and <beam>.do_z_push (0 = skip z += dt (vz - c), :316; the ABI reads beam_no_z_push so that 0 keeps the default) */
    int beam_radiation_reaction; double background_density_SI; int beam_no_z_push;
    /* SURVEY 8f-2, ionisation: a second plasma species "ion" that can be field-ionised (ADK), its electrons joining the
     * first species -- <ion>.ionization_product = <plasma> (particles/plasma/PlasmaParticleContainer.cpp:61-90, 261-440;
     * InitIonizationModule, PlasmaParticleContainerInit.cpp:382-464).  plasma_no_neutralize: <plasma>.neutralize_background
     * = false (the ions are particles now).  ion_charge: charge of ONE level (+q_e; the deposits and the push weigh it with
     * the particle's ion_lev), ion_energies: the element's ionisation energies in eV (NIST; the reference tabulates them in
     * utils/IonizationEnergiesTable.H), ion_Z of them.  In normalised units background_density_SI must be set.
     * ion_seed: key of the counter-based random number generator (one draw per ion, slice and step; the reference draws
     * from amrex::Random, whose sequence is not reproducible).  Both Bx/By solvers: under the predictor-corrector loop the
     * decisions are taken once per slice on the loop's final fields, ahead of the committing pushes (Hipace.cpp:693-701). */
    int plasma_no_neutralize;
    int ion_on; int ion_ppc[2]; double ion_density, ion_mass, ion_charge; int ion_init_level, ion_Z;
    double ion_energies[HPS_MAX_ION_LEVELS]; unsigned long long ion_seed;
    /* <beam>.do_spin_tracking (moving beam only): every beam particle carries a spin vector, initial_spin (normalised)
     * for all of them, precessing by the Thomas-BMT equation in the beam push (particles/pusher/BeamParticleAdvance.cpp:
     * 218-238; spin_anom = anomalous magnetic moment as given: the electron's is 0.00115965218128, 0 is pure Thomas precession).  The spin travels with the
     * particle through the slipped-particle hand-off and the ring messages (3 more rows). */
    int beam_spin_tracking; double beam_initial_spin[3]; double beam_spin_anom;
} hps_deck;

/* slab component indices of the engine (explicit-solver layout of fields/Fields.cpp:70-122) */
enum { HPS_C_N_JXB = 0, HPS_C_N_JYB, HPS_C_CHI, HPS_C_SY, HPS_C_SX, HPS_C_EXMBY, HPS_C_EYPBX,
       HPS_C_EZ, HPS_C_BX, HPS_C_BY, HPS_C_BZ, HPS_C_PSI, HPS_C_JXB, HPS_C_JYB, HPS_C_JZB,
       HPS_C_JX, HPS_C_JY, HPS_C_RHOMJZ, HPS_C_P_JXB, HPS_C_P_JYB, HPS_C_ION_RHOMJZ, HPS_C_RHO,
       HPS_NCOMP_MAX };

/* slab component indices with bxby_solver = 1 (predictor-corrector layout of fields/Fields.cpp:128-164: beams and
 * plasma share jx jy jz; Previous keeps Bx By jx jy; PCIter / PCPrevIter hold the iterated Bx By) */
enum { HPS_PC_N_JX = 0, HPS_PC_N_JY, HPS_PC_EXMBY, HPS_PC_EYPBX, HPS_PC_EZ, HPS_PC_BX, HPS_PC_BY, HPS_PC_BZ,
       HPS_PC_PSI, HPS_PC_JX, HPS_PC_JY, HPS_PC_JZ, HPS_PC_RHOMJZ, HPS_PC_P_BX, HPS_PC_P_BY, HPS_PC_P_JX,
       HPS_PC_P_JY, HPS_PC_ION_RHOMJZ, HPS_PC_IT_BX, HPS_PC_IT_BY, HPS_PC_PIT_BX, HPS_PC_PIT_BY, HPS_PC_RHO,
       HPS_PC_NCOMP_MAX };

int hps_engine_create (const hps_deck* deck, int device, void** handle);
int hps_engine_destroy (void* handle);
int hps_engine_begin_step (void* handle);                 /* Evolve :401-471: reset, plasma, ions */
/* The physical time step the next hps_engine_begin_step starts (Hipace::m_physical_time = step * dt,
 * PlasmaParticleContainerInit.cpp:90): the density profile's time factor and the key of the ionisation draws follow it.
 * Without a call an engine counts its own begin_step calls -- right for one engine running every step; a pipeline stage
 * that runs steps r, r + N, ... (Hipace.cpp:400-401) must say which step it is about to run. */
int hps_engine_set_step (void* handle, int step);
int hps_engine_solve_slice (void* handle, int islice);    /* SolveOneSlice :556-728               */
/* The slice in two halves, for ONE host thread that keeps several engines (time steps in flight on one device, the
 * stages of a pipeline) busy: begin enqueues everything up to and including the speculated V-cycles of the Bx/By solve
 * and the push gated on them, without waiting for the device; finish waits for that solve's norms (the one host wait of
 * a slice, Hipace.cpp:919-921) and enqueues the rest.  hps_engine_solve_slice(k) = begin(k) + finish(k); between the two
 * halves of one engine only calls on OTHER engines are allowed. */
int hps_engine_solve_slice_begin (void* handle, int islice);
int hps_engine_solve_slice_finish (void* handle, int islice);
/* 1 if hps_engine_solve_slice_finish would not wait for the device (the norms of the slice begun have arrived), else 0: a host
 * that drives several engines finishes whichever is ready first */
int hps_engine_slice_ready (void* handle);
int hps_engine_run_step (void* handle);                   /* begin_step + all slices head->tail   */
int hps_engine_sync (void* handle);                       /* host waits for the engine's stream (and, through it, the laser stream) */
int hps_engine_info (void* handle, int* ncomp, int* nguards, long* nparticles);
/* Deferred ShiftSlices: hps_engine_solve_slice leaves the slab UNSHIFTED (Fields::ShiftSlices, fields/Fields.cpp:588-670, runs
 * in the next slice's InitializeSlices pass); hps_engine_slab and hps_engine_sync enqueue the pending shift before they
 * return, hps_engine_record_event / hps_engine_copy_async / the export calls do NOT.  A host that keeps the hps_slab
 * pointer and reads the Previous / Next planes stream-ordered between two slices without one of those two calls sees
 * them as they were before the shift; HPS_LAZY_SHIFT=0 in the environment restores the shift at the end of every slice. */
hps_slab hps_engine_slab (void* handle);
/* The engine's sheet: raw device pointers.  CONTRACT for a host that invalidates particles itself: clearing the valid bit of
 * idcpu is not enough -- also store 0 to the particle's w AND psi_half.  The engine's tiled depositions take "w != 0" for
 * the valid bit and do not read idcpu (HPS_VALID_BY_W, default on; the push under HPS_VALID_BY_PSI reads psi_half the same
 * way): every path of the engine that invalidates a particle (QSA drop, absorbing boundary) zeroes both.  HPS_VALID_BY_W=0
 * in the environment restores the idcpu read. */
hps_plasma hps_engine_plasma (void* handle);
/* the tiling of the first species as the engine holds it now (NULL with tile_size 0): for hps_tiling_info and for calling
 * the *_tiled operators on the engine's own sheet (diagnostics: scripts/deposit_variants.py); owned by the engine */
int hps_engine_tiling (void* handle, void** tiling);
/* species "ion" (hps_deck.ion_on): its sheet, the electrons it has released since hps_engine_create and the size of the
 * first species now (hps_engine_plasma().n follows it); synchronises the stream */
hps_plasma hps_engine_ions (void* handle);
int hps_engine_ion_stats (void* handle, long* n_ionized_host, long* n_product_host);
hps_stream hps_engine_stream (void* handle);
/* sum |Q| per component over valid cells and all slices of the current step (host array[ncomp]) */
int hps_engine_checksums (void* handle, double* out_host);
int hps_engine_stats (void* handle, long* total_vcycles, long* slices_done);
/* predictor-corrector: iterations so far and the sum over slices of the final relative B-field error
 * (m_predcorr_avg_iterations / m_predcorr_avg_B_error of Hipace.cpp:964,1028 before the division by nz) */
int hps_engine_pc_stats (void* handle, long* iterations, double* error_sum);
/* Slices whose loop left after ONE pass because sum |B| of the guess was 0.  The reference's rule is relative_Bfield_error =
 * norm_B > 0 ? diff/norm_B : 0 (fields/Fields.cpp:1283), and the engine applies it literally.  On the serial CPU path norm_B IS 0
 * ahead of the driver (electron and ion charge cancel term by term); a scatter with atomics leaves 1e-16 residue there.  The
 * engine therefore keeps one device word per sweep, "only the cold plasma's residue has been deposited so far" (cleared by
 * hps_engine_begin_step; set by the beam's deposition when a term is not exactly zero, and from the first slice on with a grid
 * current, a second species or no neutralising background), and while it is clear stores the exact zero the serial path holds
 * into Bx, By at the end of a loop pass -- so norm_B is exactly 0 on the same slices as on the CPU, and a real but tiny norm_B
 * (a moving beam's head) iterates as it does there.  HPS_PC_EXACT_ZERO=0 turns that off; HPS_PC_NOISE_FLOOR=<relative floor>
 * (rounds 4-5: 1e-12 of mu0 c |q n0| nx ny (nx dx); now 0) is kept as a diagnostic.  This counter says how often norm_B was 0. */
int hps_engine_pc_zero_b_slices (void* handle, long* slices);
/* laser: index of the slab component "aabs" (-1 without a laser) and sum |a| over the slices solved in this step
 * (the "laserEnvelope" checksum; needs hps_engine_set_diagnostics; synchronises the stream) */
int hps_engine_laser_info (void* handle, int* aabs_comp, double* envelope_abs_sum_host);
/* the envelope a_n of the step that has begun: [nz][ny][nx] complex (re, im interleaved) to the host; synchronises */
int hps_engine_laser_envelope (void* handle, double* out_host);
/* Ring hand-off of the envelope (MultiBuffer::pack_data / unpack_data, utils/MultiBuffer.cpp:840-852, 913-925): a
 * stage passes {a_{n+1}, a_n} of a slice on (msg_dev = [2][ny][nx] complex = 4 nx ny doubles on the device), the next
 * stage stores them as its {a_n, a_{n-1}}.  Import mode (set before hps_engine_begin_step, with the index of the
 * time step the engine is about to run) keeps begin_step from initialising / rotating the time levels itself. */
int hps_engine_set_laser_import (void* handle, int on, int step);
int hps_engine_export_laser_slice (void* handle, int islice, double* msg_dev);
int hps_engine_import_laser_slice (void* handle, int islice, const double* msg_dev);
/* the same hand-off inside one process (several steps in flight on one device): straight from the engine that ran the
 * previous step, on the receiving engine's stream (make it wait for the sender's slice event first) */
int hps_engine_import_laser_from (void* handle, int islice, void* src_engine);
/* accumulate the per-slice checksums (costs one reduction pass per slice; off by default) */
int hps_engine_set_diagnostics (void* handle, int on);
/* Field diagnostics (Fields::Copy, fields/Fields.cpp:413-533; geometry of Diagnostic::ResizeFDiagFAB,
 * diagnostics/Diagnostic.cpp:300-390; diag_type xyz over the whole box, level 0): a device-resident 3-D array
 * F[ncomps][nz/cz][ny/cy][nx/cx] (x fastest) that every solved slice adds its share to -- linear interpolation of the
 * zero-extended slab components onto the diagnostic grid in x, y and z, exactly as diagnostic.coarsening = cx cy cz
 * does (1 1 1 = plain copy).  Sizes must be divisible by the coarsening.  The array is cleared by
 * hps_engine_begin_step; hps_engine_field_diagnostic copies it to the host (synchronises the stream).
 * Call set before hps_engine_begin_step; ncomps = 0 switches it off. */
int hps_engine_set_field_diagnostic (void* handle, int ncomps, const int* comps, const int coarsening[3]);
int hps_engine_field_diagnostic (void* handle, double* out_host);
/* The other shapes of the reference's field diagnostic (Diagnostic::ResizeFDiagFAB and TrimIOBox,
 * diagnostics/Diagnostic.cpp:300-410): diagnostic.diag_type -- slice_dir -1 = xyz, 0 = yz, 1 = xz (one cell about the middle
 * of the box in that direction: the mean of the two central rows for an even cell count, the central row for an odd one),
 * 2 = xy (one plane that sums the slices of the z range times dz, Fields.cpp:469-479) -- and diagnostic.patch_lo / patch_hi
 * (NULL: the whole box; the patch is rounded to cells per direction, the slice is cut about the middle of the patch).  The
 * array of hps_engine_field_diagnostic is then F[ncomps][n[2]][n[1]][n[0]] with n, and the real box of the diagnostic
 * grid, from hps_engine_field_diagnostic_geometry. */
int hps_engine_set_field_diagnostic_box (void* handle, int ncomps, const int* comps, const int coarsening[3], int slice_dir,
                                         const double* patch_lo /* [3] or NULL */, const double* patch_hi /* [3] or NULL */);
int hps_engine_field_diagnostic_geometry (void* handle, int* n3, double* lo3, double* hi3);

/* In-situ field reductions (Fields::InSituComputeDiags, fields/Fields.cpp:1288-1347; explicit solver only, as the
 * reference): per solved slice, dx dy dz * sum over the valid cells of {Ex^2, Ey^2, Ez^2, Bx^2, By^2, Bz^2, ExmBy^2,
 * EypBx^2, jz_beam, Ez jz_beam} with Ex = ExmBy + c By, Ey = EypBx - c Bx.  out_host[q*nz + islice], q = 0..9, for the
 * step that is being (or has just been) solved; cleared by hps_engine_begin_step; synchronises the stream. */
int hps_engine_set_insitu_fields (void* handle, int on);
int hps_engine_insitu_fields (void* handle, double* out_host /* [10*nz] */);

/* In-situ plasma moments (PlasmaParticleContainer::InSituComputeDiags, particles/plasma/PlasmaParticleContainer.cpp:
 * 443-530), taken at the start of every slice (Hipace.cpp:590) over the valid particles within `radius` of the axis:
 * out_host[q*nz + islice], q = 0..14 = sum(w), [x], [x^2], [y], [y^2], [ux], [ux^2], [uy], [uy^2], [uz], [uz^2], [ga],
 * [ga^2] (averages: divided by sum(w)), [(ga-1)(1-vz)] (sum), Np (count, as a double), with w = weight * gamma/psi.
 * radius <= 0 switches it off; cleared by hps_engine_begin_step; reading synchronises the stream. */
int hps_engine_set_insitu_plasma (void* handle, double radius);
/* In-situ beam moments (BeamParticleContainer::InSituComputeDiags, particles/beam/BeamParticleContainer.cpp:476-556),
 * taken after the field solves and before the beam push (Hipace.cpp:681) over the slice's own valid particles (those
 * that slipped in from the slice ahead are not counted, :494) within `radius` of the axis: out_host[q*nz + islice],
 * q = 0..22 = sum(w), [x], [x^2], [y], [y^2], [z], [z^2], [ux], [ux^2], [uy], [uy^2], [uz], [uz^2], [x*ux], [y*uy],
 * [z*uz], [x*uy], [y*ux], [ux/uz], [uy/uz], [ga], [ga^2] (averages: divided by sum(w)), Np (count, as a double), with
 * u = proper velocity / c.  radius <= 0 switches it off; cleared by hps_engine_begin_step; reading synchronises. */
/* V-cycles of the multigrid envelope solver so far (laser_solver = 2), 0 otherwise */
int hps_engine_laser_vcycles (void* handle, long* vcycles);
int hps_engine_set_insitu_beam (void* handle, double radius);
int hps_engine_insitu_beam (void* handle, double* out_host /* [23*nz] */);
int hps_engine_insitu_plasma (void* handle, double* out_host /* [15*nz] */);

/* particle tiling of the engine: tile_size 0 = per-particle global-atomic kernels, 16 | 32 = LDS
 * tiles, re-sorted after sort_period slices at the latest (plasmas.reorder_period of the reference; see hps_engine_sorts).
 * Call before hps_engine_begin_step. */
int hps_engine_set_tiling (void* handle, int tile_size, int sort_period);
int hps_engine_fallbacks (void* handle, long* n_fallback_host);
/* Plasma density profile (InitParticles evaluates <plasma>.density(x,y,z) per particle with z = c t,
 * PlasmaParticleContainerInit.cpp:246-313; UpdateDensityFunction / density_table_file, PlasmaParticleContainer.cpp:98-117,
 * 211-217).  The input parser is out of scope; the profile is tabulated: n(x, y, ct) = <species>.density * f_r(sqrt(x^2 +
 * y^2)) * f_t(c t), both piecewise linear (constant beyond the ends of a table; n = 0 entries: factor 1), t = hipace.dt *
 * step.  Plasma channels and density ramps are of this form.  A lattice point whose density is <= 0 carries no particle
 * (min_density = 0).  Applies to every plasma species from the next hps_engine_begin_step on. */
int hps_engine_set_density_profile (void* handle, int nr, const double* r_host, const double* fr_host, int nt,
                                    const double* ct_host, const double* ft_host);
/* Fused schedule: the gather + push of slice k also deposits the pushed particles' currents into slice k-1 (one pass over
 * the sheet instead of two; the slab is shifted / cleared before it).  After hps_engine_solve_slice(k) the components jx,
 * jy, chi, rhomjz [, rho] then already belong to slice k-1; everything else, the per-slice checksums and diagnostics are
 * unchanged.  Applies to the explicit solver with a static beam, no laser and one plasma species; off by default. */
int hps_engine_set_fusion (void* handle, int on);
/* number of particle re-sorts so far (periodic + adaptive: the sheet is re-sorted after sort_period slices at the
   latest, earlier once more than 1/256 of it has left the halo of its tile) */
int hps_engine_sorts (void* handle, long* n_sorts_host);
/* HIP-event phase timers on the engine's stream.  phase_times sums, over the slices solved since
 * profiling was switched on (or since the last call), the milliseconds spent in
 * {deposit_current, poisson x3 (+rhs, grad), explicit_deposit, mg_solve1, advance_plasma, other,
 * particle re-sort, empty interval} into ms_host[8] and stores the slice count; it synchronises the stream.
 * "empty interval" = two event records back to back (no kernel between them; 0 when the schedule has no such pair):
 * what every interval carries on top of its kernels. */
/* on = 1: all phases; on = 2: only the deposition kernel and the empty interval (4 event records per slice instead of 11) */
int hps_engine_set_profiling (void* handle, int on);
/* time every stride-th slice only (default 1): the 11 event records of a profiled slice cost about 3.4 us each
 * (4.5 % of a 1024^2 slice at stride 1, measured); phase_times then sums over the profiled slices */
int hps_engine_set_profiling_stride (void* handle, int stride);
int hps_engine_phase_times (void* handle, double* ms_host, long* nslices_host);

/* Driver-beam storage of the engine: particles are kept in slice-major blocks, block p (p-th slice
 * from the head, p = nz-1-islice) = [7][count_p] doubles (x,y,z,ux,uy,uz,w) starting at
 * 7*offsets[p].  A pipeline driver can point the engine at its own buffer (the receive buffer of
 * the ring hand-off); hps_engine_initial_beam copies the injected (step 0) beam into one. */
int hps_engine_beam_info (void* handle, long* nbeam_host, long* offsets_host /* [nz+1] */);
/* A beam the host has initialised itself -- any of the reference's injection types (fixed_weight, fixed_weight_pdf, from_file:
 * InitBeamFixedWeight3D / InitBeamFixedWeightPDFSlice / InitBeamFromFile, particles/beam/BeamParticleContainerInit.cpp:348-477,
 * 479-695, 697-960, which draw from amrex::Random or read an openPMD file) -- in place of the deck's fixed_ppc one
 * (deck.beam_profile = -1: none).  soa_host = [7][n] doubles x y z ux uy uz w on the host, u = gamma beta c, w as the
 * deposition takes it (the weight AddOneBeamParticle stores: in normalised units the injection's total weight over the
 * particle count and the cell volume); any order.  The particles are binned into the box's slices, slice =
 * int((z - lo_z)/dz) as the reference's BoxSorter does (particles/sorting/BoxSort.cpp:34-43), input order kept inside
 * a slice.  Particles outside the box in z are left out and counted in *n_outside (NULL: they are an error).  Call after
 * hps_engine_create and before the first hps_engine_begin_step; works for hipace.dt = 0 and for a moving beam.  In a pipeline
 * every stage's engine is given the same particles (the block layout of hps_engine_beam_info and the hand-off capacity derive
 * from them; only the head stage injects them, MultiBuffer.cpp:809). */
int hps_engine_set_beam_particles (void* handle, long n, const double* soa_host, long* n_outside);
int hps_engine_set_beam_storage (void* handle, double* storage_dev /* [7*nbeam] or NULL = own */);
int hps_engine_initial_beam (void* handle, double* dst_dev);
/* hipace.dt != 0: the beam lives in one SoA over all particles (head slice first) whose slice boundaries move when
 * particles slip (hipace_amd/csrc/beam.hip).  Copies the boundaries [nz+1] and, if soa_host != NULL, the seven
 * arrays x y z ux uy uz w ([7][nbeam], nbeam = hps_engine_beam_info) to the host; synchronises the stream. */
int hps_engine_beam_state (void* handle, long* boundaries_host, double* soa_host);
/* Ring hand-off of the moving beam (MultiBuffer::put_data / get_data, utils/MultiBuffer.cpp:444-609).  A message is
 * 1 + 7*cap doubles on the device: [count | x[cap] y[cap] z[cap] ux[cap] uy[cap] uz[cap] w[cap]], cap =
 * hps_engine_beam_capacity (twice the fullest injected slice).  export: what sits on slice `islice` after its push;
 * import (import mode on, set before hps_engine_begin_step; slices in head-first order, slice k-1 before slice k is
 * solved): the regular particles of that slice for the step that has begun.  All asynchronous on the engine's stream. */
int hps_engine_beam_capacity (void* handle, long* cap_host);
/* A beam whose slices may come to hold more than twice what the fullest one held when it was injected (a hot beam that bunches
 * as it slips): the particles a slice's hand-off message has room for, set before the first hps_engine_begin_step -- the same
 * on every engine of a pipeline (the message size is 1 + rows * cap doubles).  A slice that outgrows it is an error at the end
 * of the step, never a silent loss. */
int hps_engine_set_beam_capacity (void* handle, long cap);
/* rows of a hand-off message: 7, or 10 with spin tracking (sx sy sz behind w); a message is 1 + rows*cap doubles */
int hps_engine_beam_message_rows (void* handle, int* rows_host);
/* spin tracking: the three spin arrays [3][nbeam] in the order of hps_engine_beam_state; synchronises the stream */
int hps_engine_beam_spin (void* handle, double* soa_host);
int hps_engine_set_beam_import (void* handle, int on);
int hps_engine_export_beam_slice (void* handle, int islice, double* msg_dev);
int hps_engine_import_beam_slice (void* handle, int islice, const double* msg_dev);
/* The slab kernels skip the beam-current planes outside the beam's transverse support.  After
 * hps_engine_set_beam_storage the support is the whole plane (caller-owned particles may sit anywhere); a driver
 * that knows its blocks are the injected beam handed along the ring (hipace.dt = 0: the beam does not move) calls
 * this to restore the support box of the injected beam. */
int hps_engine_assume_initial_beam_support (void* handle);

/* Several time steps in flight on ONE device (hipace_amd/pipeline.py::run_local_pipeline): every step runs in its own
 * engine on its own stream, coupled only by the per-slice beam hand-off.  record_event marks the engine's stream
 * (pool slot `slot`, created on first use) and returns the event; wait_event makes the engine's stream wait for an
 * event recorded by another engine -- the device-side form of "slice k of the previous step has been pushed".
 * copy_async is a device-to-device copy on the engine's stream (the in-process hand-off, MultiBuffer.cpp:299-308).
 * The event covers everything the engine has been given so far, the envelope solver's slice on the engine's laser
 * stream included.  It is for streams of the engine's device (another engine, the ring's send stream): it is recorded
 * without the system-scope fence HIP puts behind an event by default (HPS_EVENT_FENCE=0 restores it). */
int hps_engine_record_event (void* handle, int slot, void** event_out);
/* The engines of a device take their streams from a pool made with the first engine (HPS_STREAM_POOL, default 3: streams
 * created back to back land on different pipes of the command front end).  When the pool is made every pair of its streams
 * runs a 100-us kernel side by side once; pairs that took turns are counted and a warning goes to stderr.  Returns that
 * count (0 = every pair overlaps), -1 before the first engine of the device or with HPS_STREAM_POOL_CHECK=0. */
int hps_stream_pool_shared_pairs (int device);
int hps_engine_wait_event (void* handle, void* event);
int hps_engine_copy_async (void* handle, void* dst_dev, const void* src_dev, long bytes);

/* ---- ring pipeline over time steps: the transport (utils/MultiBuffer.H:21-34; MultiBuffer.cpp:287-609) --------
 * One process per GPU; rank r runs time steps r, r+N, ... (Hipace.cpp:400-401) and hands every pushed beam slice (and
 * the laser envelope of the slice) to rank r+1.  The reference does that with MPI_Isend / MPI_Irecv between ring
 * neighbours (MultiBuffer::make_progress :287-442, put_data :444-493, get_data :495-609); here it is RCCL
 * ncclSend / ncclRecv over xGMI on device buffers (HPS_RING_EDGE=rccl), or peer copies through a mailbox (the default,
 * described behind hps_ring_destroy below).  Every edge r -> r+1 of the ring is its own 2-rank communicator
 * with its own stream, so sends and receives of a rank progress independently; ordering against the engine is by
 * events only (hps_engine_record_event / hps_engine_wait_event), never by a host synchronisation.
 *
 * Bootstrap (host driver, e.g. hipace_amd/pipeline.py over torch.distributed's store): rank r makes the id of ITS
 * outgoing edge with hps_ring_unique_id and hands it to rank r+1; hps_ring_init(rank, world, device, id of the edge
 * (r-1 -> r), id of the edge (r -> r+1)) is collective over the ring.  world = 1: id_edge_in may be NULL, the ring is
 * a 1-rank communicator and hps_ring_sendrecv_self is the hand-off (MultiBuffer.cpp:299-308, "send to myself").
 * hps_ring_init also sends one small message over each of the rank's edges (edge colour by edge colour, like the
 * communicator creation): RCCL connects two peers inside their first ncclSend / ncclRecv -- a rendezvous of the two
 * hosts -- and that must not be left to the pipeline's first hand-offs, whose order around the ring is a circle.
 * Afterwards hps_ring_send_slice / recv_slice only enqueue.  A rank must not synchronise its whole device
 * (hipDeviceSynchronize) while receives it has posted ahead are waiting for their data; hps_engine_sync,
 * hps_ring_sync_sends and hps_ring_sync wait for one stream each. */
#define HPS_RING_ID_BYTES 128
int hps_ring_unique_id (char* id_out /* [HPS_RING_ID_BYTES] */);
int hps_ring_init (int rank, int world, int device, const char* id_edge_in, const char* id_edge_out, void** ring);
/* put_data: `bytes` at msg_dev go to rank+1.  The send waits (on the device) for after_event (hipEvent_t, may be NULL);
 * *done_event (pool slot `slot` of the ring's send events; a hipEvent_t on both kinds of edge) fires when msg_dev may be
 * overwritten.  RCCL edge: only enqueues.  ipc edge (the default): the HOST blocks in this call (up to HPS_RING_TIMEOUT_S)
 * until the next rank has posted the matching receive and its buffer is free -- a host that drives several stages from one
 * thread asks hps_ring_can_send first and keeps the block in an outbox meanwhile (examples/pipeline_host.cpp). */
int hps_ring_send_slice (void* ring, const void* msg_dev, long bytes, void* after_event, int slot, void** done_event);
/* get_data: post the receive of the next message from rank-1 into msg_dev (messages arrive in the order they were
 * sent).  The receive waits for after_event (may be NULL); *done_event (slot `slot` of the ring's receive handles) stands
 * for "the data has landed": hps_ring_engine_wait(ring, engine, *done_event) before the slice that reads it.  The handle is
 * a hipEvent_t on the RCCL edge only; on the ipc edge (the default) it is an object of the ring, and
 * hps_engine_wait_event refuses it with HPS_ERR_ARG -- hosts written against rounds 2-4 change that one call. */
int hps_ring_recv_slice (void* ring, void* msg_dev, long bytes, void* after_event, int slot, void** done_event);
int hps_ring_sendrecv_self (void* ring, const void* src_dev, void* dst_dev, long bytes, void* after_event, int slot,
                            void** done_event);
/* the ring's receive (which = 0) or send (which = 1) stream waits for an event of another stream */
int hps_ring_stream_wait (void* ring, int which, void* event);
int hps_ring_sync_sends (void* ring);           /* host waits until everything sent so far has left */
int hps_ring_sync (void* ring);                 /* host waits for both streams of the ring (end of a run) */
/* Both waits poll and give up with HPS_ERR_COMM (the rank's message counters in hps_last_error) after HPS_RING_TIMEOUT_S
 * seconds (environment, default 900) -- hps_ring_sync_timeout with the limit as an argument: a ring whose peer is gone, or
 * whose posted-ahead receives share a hardware queue with the sends they wait for, fails loudly instead of hanging.
 * hps_ring_init refuses to start a ring of 2+ ranks unless GPU_MAX_HW_QUEUES >= 8 is set in the environment -- by the
 * host, BEFORE its first HIP call (the runtime reads it once; the library can only see the string): examples/pipeline_host.cpp
 * does it with setenv at the top of main, hipace_amd/_lib.py at import when WORLD_SIZE > 1.  HPS_RING_ALLOW_SHARED_QUEUES=1
 * turns the refusal off for a host that has arranged its queues some other way. */
int hps_ring_sync_timeout (void* ring, double seconds);
int hps_ring_stats (void* ring, long* n_sent, long* n_received, long long* bytes_sent, long long* bytes_received);
/* what RCCL reports (ncclCommCount / ncclCommUserRank) for the communicators of the incoming and the outgoing edge: 2 and
 * 2 ranks on a ring of 2+ processes (this rank is 1 on the incoming edge, 0 on the outgoing one), 0 and 1 for the one-rank
 * ring -- the evidence that N processes are on RCCL (bench.py's `rccl_ranks_seen`) */
int hps_ring_info (void* ring, int* world, int* comm_in_ranks, int* comm_out_ranks, int* my_rank_in, int* my_rank_out);
int hps_ring_destroy (void* ring);
/* ---- the second kind of edge: peer copies through a shared-memory mailbox (HPS_RING_EDGE=ipc, the default; =rccl for the
 * RCCL edge above).  For the processes of one node, on different devices or on the SAME device (RCCL refuses two ranks on
 * one device).  hps_ring_unique_id makes the edge's mailbox (POSIX shared memory; the id is its name), hps_ring_init maps
 * the mailboxes of both edges and waits for both neighbours (HPS_RING_CONNECT_TIMEOUT_S, default 300).  A posted receive is
 * a descriptor in the mailbox (allocation -- exported once with hipIpcGetMemHandle --, offset, bytes) plus "buffer free",
 * written by the receive stream behind after_event; a send is hipMemcpyAsync into the receiver's buffer plus "landed",
 * written by the send stream: no kernel waits on the device, no requirement on hardware queues (GPU_MAX_HW_QUEUES).  What
 * MultiBuffer does with MPI_Test polls (MultiBuffer.cpp:287-442) the host does with the three calls below.  Differences a
 * host sees: (1) *done_event of hps_ring_recv_slice is a handle of the RING -- order an engine behind it with
 * hps_ring_engine_wait (works for both kinds of edge), not hps_engine_wait_event; (2) hps_ring_send_slice makes the HOST wait
 * (up to HPS_RING_TIMEOUT_S) until the matching receive is posted and free -- ask hps_ring_can_send first where the host
 * must not block; (3) message sizes are checked: a send whose size differs from the posted receive's fails; (4) receive
 * buffers must stay allocated until hps_ring_destroy (their allocations are mapped into the sending process). */
int hps_ring_edge_kind (void* ring);                              /* 0 = RCCL, 1 = ipc */
int hps_ring_can_send (void* ring);                               /* 1: the next hps_ring_send_slice will not wait on the host */
int hps_ring_recv_landed (void* ring, void* done_event);          /* 1: that receive's message is in the buffer; 0: not yet */
/* order the engine behind that receive.  RCCL edge: the engine's STREAM waits for the receive's event (the call only
 * enqueues).  ipc edge: the HOST blocks here (up to HPS_RING_TIMEOUT_S) until the sender's stream has flagged the message
 * as landed -- the engine's later launches are then ordered behind it by program order; a host with several stages per
 * thread polls hps_ring_recv_landed first and gives its other stages their turn (examples/pipeline_host.cpp). */
int hps_ring_engine_wait (void* ring, void* engine, void* done_event);

/* ---- utilities ---------------------------------------------------------------------------- */
int hps_memcpy_d2h (void* dst_host, const void* src_dev, long bytes);
int hps_memcpy_h2d (void* dst_dev, const void* src_host, long bytes);
int hps_device_count (int* n_host);

#ifdef __cplusplus
}
#endif
#endif /* HPSLICE_H_ */
