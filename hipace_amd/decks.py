"""Input decks for the slice engine (normalised units, explicit Bx/By solver).

Each deck is a plain dict mirroring the keys of the reference's input files that matter to the
per-slice hot path.  The named decks restate

* ``linear_wake``     -- /root/reference/examples/linear_wake/inputs_normalized
                         (run as tests/linear_wake.normalized.1Rank.sh:33-36: + ``rho``)
* ``blowout_wake``    -- /root/reference/examples/blowout_wake/inputs_normalized
                         (run as tests/blowout_wake_explicit.2Rank.sh:32-35: max_step=1, dt=0)
* ``beam_in_vacuum``  -- /root/reference/examples/beam_in_vacuum/inputs_normalized
                         (run as tests/beam_in_vacuum.normalized.Serial.sh:30-34: order 0)
* ``beam_in_vacuum_open_boundary`` -- the same deck as run by tests/beam_in_vacuum_open_boundary.normalized.1Rank.sh
                         (predictor-corrector Bx/By; oracle only)
* ``synthetic(n, nz)``-- the BASELINE.md section 3 benchmark deck (blowout deck scaled to n x n,
                         ppc 2x2) used by bench.py.
"""
import copy

_DEFAULT = dict(
    nx=64, ny=64, nz=100,
    lo=(-8.0, -8.0, -6.0), hi=(8.0, 8.0, 6.0),
    order=2,                 # hipace.depos_order_xy
    deriv_type=2,            # hipace.depos_derivative_type (Hipace.H:78)
    plasma_ppc=(1, 1), plasma_density=1.0, plasma_radius=0.0,   # radius 0 -> infinite
    plasma_charge=-1.0, plasma_mass=1.0,                       # electron, normalised units
    max_qsa=35.0,            # plasmas.max_qsa_weighting_factor (PlasmaParticleContainer.H:161)
    n_subcycles=1,
    beam_profile=0,          # -1 none, 0 gaussian, 1 flattop
    beam_zmin=-5.9, beam_zmax=5.9, beam_radius=1.2, beam_density=3.0,
    beam_umean=(0.0, 0.0, 2000.0), beam_pos_mean=(0.0, 0.0, 0.0),
    beam_pos_std=(0.3, 0.3, 1.41), beam_ppc=(1, 1, 1), beam_charge=-1.0,
    bc=1,                    # boundary.particle: 0 Reflecting, 1 Periodic, 2 Absorbing
    mg_tol_rel=1.0e-4,       # hipace.MG_tolerance_rel (Hipace.H:246)
    mg_tol_abs=2.2250738585072014e-308,   # numeric_limits<double>::min() (Hipace.H:248)
    deposit_rho=0,
    n_steps=1,               # max_step + 1
    dt=0.0,                  # hipace.dt; 0 in every BASELINE deck: the beam never moves
    beam_n_subcycles=10,     # beam.n_subcycles (BeamParticleContainer.H:222)
    beam_mass=1.0,
    ext_E_slope=(0.0, 0.0),  # beams.external_E(x,y,z,t) = s0*x s1*y 0.
    bxby_solver=0,           # hipace.bxby_solver: 0 explicit (Hipace.H:244), 1 predictor-corrector
    predcorr_tol=4.0e-2,     # hipace.predcorr_B_error_tolerance (Hipace.H:210)
    predcorr_max_iter=30,    # hipace.predcorr_max_iterations (Hipace.H:213)
    predcorr_mix=0.05,       # hipace.predcorr_B_mixing_factor (Hipace.H:222)
    field_bc=0,              # boundary.field: 0 Dirichlet; 1 Open (multipole expansion of the sources to order 18, fields/Fields.cpp:678-735)
    laser_on=0,              # lasers.names != no_laser: a Gaussian envelope (laser/Laser.H:32-45), static (step 0 only)
    laser_a0=0.0, laser_w0=1.0, laser_L0=1.0, laser_lambda0=0.8e-6, laser_pos=(0.0, 0.0, 0.0),
    laser_zfoc=0.0,          # laser.focal_distance
    laser_solver=0,          # lasers.solver_type: 0 keep the envelope static, 1 "fft" (AdvanceSliceFFT), 2 "multigrid" (AdvanceSliceMG)
    beam_radiation_reaction=0, background_density_SI=0.0, beam_no_z_push=0,      # <beam>.do_radiation_reaction, hipace.background_density_SI, <beam>.do_z_push = 0
    laser_mg_tol_rel=1.0e-4, laser_mg_tol_abs=0.0,      # lasers.MG_tolerance_rel / MG_tolerance_abs
    laser_use_phase=1,       # lasers.use_phase (MultiLaser.H:203)
    grid_current_on=0, grid_current_peak=0.0, grid_current_mean=(0.0, 0.0, 0.0), grid_current_std=(1.0, 1.0, 1.0),   # grid_current.*
    si_units=0,              # hipace.normalized_units = 0: SI constants, charges and masses in C and kg, weights = charges
    # second plasma species "ion" with ADK field ionisation; the released electrons join the first species
    # (<ion>.ionization_product, PlasmaParticleContainer.cpp:61-90, 261-440)
    plasma_no_neutralize=0,  # <plasma>.neutralize_background = false for the first species
    ion_on=0, ion_ppc=(0, 0), ion_density=0.0, ion_mass=0.0, ion_charge=0.0, ion_init_level=0, ion_Z=0,
    ion_energies=(0.0,) * 56, ion_seed=0,
    # <beam>.do_spin_tracking, initial_spin, spin_anom (BeamParticleContainer.cpp:105-109; anomalous magnetic moment of the electron)
    beam_spin_tracking=0, beam_initial_spin=(1.0, 0.0, 0.0), beam_spin_anom=0.00115965218128,
)

# Ionisation energies in eV of a few elements: NIST Atomic Spectra Database (Kramida, Ralchenko, Reader and NIST ASD
# Team, ver. 5.2), the values the reference tabulates in utils/IonizationEnergiesTable.H.
IONIZATION_ENERGIES_EV = {
    "H": (13.59843449,),
    "He": (24.58738880, 54.4177650),
    "Li": (5.39171495, 75.6400964, 122.4543581),
    "N": (14.53413, 29.60125, 47.4453, 77.4735, 97.8901, 552.06732, 667.046116),
    "Ar": (15.7596117, 27.62967, 40.735, 59.58, 74.84, 91.290, 124.41, 143.4567, 422.60, 479.76, 540.4, 619.0, 685.5,
           755.13, 855.5, 918.375, 4120.6656, 4426.2228),
}
M_P_DA = 1.007276466621      # proton mass in Da; <plasma>.mass_Da is converted with m_p / 1.007276466621 (PlasmaParticleContainer.cpp:118-122)


def with_ion_species(d, element, density, ppc=(1, 1), mass_Da=None, initial_level=0, seed=0):
    """Add the species "ion" of `element` to deck d (in place): <ion>.element / mass_Da / initial_ion_level / ppc /
    density, ionization_product = the first species.  Charge +e per level, mass in units of the deck (kg in SI, electron
    masses in normalised units)."""
    en = IONIZATION_ENERGIES_EV[element]
    m_p_SI, m_e_SI, q_e_SI = 1.67262192369e-27, 9.1093837015e-31, 1.602176634e-19
    mass_Da = mass_Da if mass_Da is not None else {"H": 1.008, "He": 4.002602, "Li": 6.94, "N": 14.007, "Ar": 39.948}[element]
    mass_SI = m_p_SI * mass_Da / M_P_DA
    si = bool(d.get("si_units", 0))
    d.update(ion_on=1, ion_ppc=tuple(ppc), ion_density=density, ion_mass=mass_SI if si else mass_SI / m_e_SI,
             ion_charge=q_e_SI if si else 1.0, ion_init_level=initial_level, ion_Z=len(en),
             ion_energies=tuple(en) + (0.0,) * (56 - len(en)), ion_seed=seed)
    return d


def blowout_wake():
    d = copy.deepcopy(_DEFAULT)
    d["n_steps"] = 2
    return d


def linear_wake():
    d = copy.deepcopy(_DEFAULT)
    d.update(nx=32, ny=32, nz=200, lo=(-10.0, -10.0, -7.5), hi=(10.0, 10.0, 2.0),
             beam_profile=1, beam_zmin=-1.0, beam_zmax=1.0, beam_radius=3.0, beam_density=0.01,
             deposit_rho=1)
    return d


def linear_wake_gaussian():
    """linear_wake grid and plasma with a Gaussian driver (smooth in zeta) -- the deck of the predictor-corrector
    tests; the reference's own such test (tests/ion_motion.SI.1Rank.sh) also drives with a Gaussian beam."""
    d = linear_wake()
    d.update(beam_profile=0, beam_pos_std=(1.5, 1.5, 0.7), beam_pos_mean=(0.0, 0.0, 0.0), beam_zmin=-7.4, beam_zmax=1.9,
             beam_radius=6.0, beam_density=0.05)
    return d


def beam_in_vacuum():
    d = copy.deepcopy(_DEFAULT)
    d.update(nx=512, ny=768, nz=4, lo=(-200.0, -200.0, -2.0), hi=(200.0, 200.0, 2.0), order=0,
             plasma_ppc=(0, 0), plasma_density=0.0,
             beam_profile=1, beam_zmin=-10.0, beam_zmax=10.0, beam_radius=1.0, beam_density=1.0,
             beam_umean=(0.0, 0.0, 1.0e3), beam_ppc=(2, 2, 1), deposit_rho=1)
    return d


def synthetic(n=1024, nz=1024, ppc=2):
    """BASELINE.md section 3 / SURVEY 8(d) benchmark deck: blowout deck on n x n x nz, ppc x ppc."""
    d = copy.deepcopy(_DEFAULT)
    d.update(nx=n, ny=n, nz=nz, plasma_ppc=(ppc, ppc))
    return d


def beam_evolution():
    """tests/beam_evolution.1Rank.sh: beam_in_vacuum deck, 32x32x10, 21 steps of dt = 3 in a linear focusing field."""
    d = copy.deepcopy(_DEFAULT)
    d.update(nx=32, ny=32, nz=10, lo=(-2.0, -2.0, -2.0), hi=(2.0, 2.0, 2.0), order=2,
             plasma_ppc=(0, 0), plasma_density=0.0,
             beam_profile=1, beam_zmin=-10.0, beam_zmax=10.0, beam_radius=1.0, beam_density=1.0e-8,
             beam_umean=(0.0, 0.0, 1.0e3), beam_ppc=(4, 4, 1), n_steps=21, dt=3.0, ext_E_slope=(0.5, 0.5))
    return d


def beam_in_vacuum_open_boundary():
    """tests/beam_in_vacuum_open_boundary.normalized.1Rank.sh:31-43 -- the reference's only checksum fixture of the
    predictor-corrector Bx/By solver: beam_in_vacuum deck, order 0, off-centre beam, boundary.field = Open."""
    d = beam_in_vacuum()
    d.update(lo=(-4.0, -4.0, -2.0), hi=(4.0, 4.0, 2.0), beam_pos_mean=(2.0, -1.0, 0.0), bc=2, deposit_rho=1,
             bxby_solver=1, predcorr_mix=0.95, predcorr_max_iter=5, predcorr_tol=4.0e-2, field_bc=1)
    return d


def laser_blowout_wake():
    """tests/laser_blowout_wake_explicit.1Rank.sh:32-45: blowout_wake deck without a beam, driven by a Gaussian laser
    pulse (a0 = 4.5, w0 = 4, L0 = 2) on 128 x 128 x 100 cells, max_step = 0."""
    d = copy.deepcopy(_DEFAULT)
    d.update(nx=128, ny=128, nz=100, lo=(-20.0, -20.0, -7.5), hi=(20.0, 20.0, 6.0), beam_profile=-1, n_steps=1,
             laser_on=1, laser_a0=4.5, laser_w0=4.0, laser_L0=2.0, laser_lambda0=0.8e-6, laser_pos=(0.0, 0.0, 0.0))
    return d


def laser_evolution():
    """tests/laser_evolution.SI.2Rank.sh with lasers.solver_type = fft (examples/laser/inputs_SI): a Gaussian pulse in
    vacuum, 128 x 128 x 50 cells, 31 steps of c dt = 70 um.  The reference runs it in SI units with kp_inv = 10 um;
    here lengths are in units of kp_inv (the envelope equation has no other scale in vacuum)."""
    d = copy.deepcopy(_DEFAULT)
    d.update(nx=128, ny=128, nz=50, lo=(-6.0, -6.0, -8.0), hi=(6.0, 6.0, 6.0), order=0, plasma_ppc=(0, 0), plasma_density=0.0,
             beam_profile=-1, n_steps=31, dt=7.0,
             laser_on=1, laser_a0=1.0, laser_w0=2.0, laser_L0=2.0, laser_lambda0=0.08, laser_pos=(0.0, 0.0, 0.0),
             laser_zfoc=100.0, laser_solver=1, laser_use_phase=1)
    return d


# 2018 CODATA values of utils/Constants.H:15-24
SI = dict(c=299792458.0, ep0=8.8541878128e-12, mu0=1.25663706212e-06, q_e=1.602176634e-19, m_e=9.1093837015e-31)


def laser_blowout_wake_SI():
    """tests/laser_blowout_wake_explicit.SI.1Rank.sh (examples/blowout_wake/inputs_SI): the laser-driven wake in SI
    units, kp_inv = 10 um, n_e = wp^2 m_e eps0 / q_e^2 with wp = c kp -- the deck of BASELINE config 5 at test size."""
    kp_inv = 10.0e-6
    wp = SI["c"] / kp_inv
    ne = wp ** 2 * SI["m_e"] * SI["ep0"] / SI["q_e"] ** 2
    d = laser_blowout_wake()
    d.update(si_units=1, lo=(-20.0 * kp_inv, -20.0 * kp_inv, -7.5 * kp_inv), hi=(20.0 * kp_inv, 20.0 * kp_inv, 6.0 * kp_inv),
             plasma_density=ne, plasma_charge=-SI["q_e"], plasma_mass=SI["m_e"],
             laser_w0=4.0 * kp_inv, laser_L0=2.0 * kp_inv, laser_lambda0=0.8e-6)
    return d


def config5(n=1024, nz=2048, solver=1, si=False, ionize=True):
    """BASELINE configs[4]: laser_blowout_wake on n x n x nz cells, 4 ppc, a Gaussian pulse (a0 = 4.5, w0 = 4 kp^-1, L0 = 2 kp^-1,
    lambda0 = 0.8 um) that drives the wake and is advanced by the envelope solver on every slice (solver 1 = fft, 2 = multigrid),
    c dt = 5 kp^-1, neutral nitrogen at a fifth of the electron density (one macro-atom per cell) field-ionised by the wake.
    si = True: the deck as BASELINE names it (tests/laser_blowout_wake_explicit.SI.1Rank.sh, examples/blowout_wake/inputs_SI:
    hipace.normalized_units = 0, kp_inv = 10 um); si = False: its normalised twin (same kernels, other constants)."""
    if si:
        kp_inv = 10.0e-6
        d = laser_blowout_wake_SI()
        d.update(nx=n, ny=n, nz=nz, plasma_ppc=(2, 2), lo=(-20.0 * kp_inv, -20.0 * kp_inv, -15.0 * kp_inv),
                 hi=(20.0 * kp_inv, 20.0 * kp_inv, 6.0 * kp_inv), laser_solver=solver, dt=5.0 * kp_inv / SI["c"])
        if ionize:
            with_ion_species(d, "N", 0.2 * d["plasma_density"], ppc=(1, 1), initial_level=0, seed=5)
        return d
    d = synthetic(n, nz, 2)
    d.update(beam_profile=-1, lo=(-20.0, -20.0, -15.0), hi=(20.0, 20.0, 6.0), laser_on=1, laser_a0=4.5, laser_w0=4.0,
             laser_L0=2.0, laser_lambda0=0.08, laser_solver=solver, dt=5.0)
    if ionize:
        with_ion_species(d, "N", 0.2, ppc=(1, 1), initial_level=0, seed=5)
        d["background_density_SI"] = 2.8239587008591567e23      # kp_inv = 10 um (hipace.background_density_SI)
    return d


def _ne_SI(kp_inv=10.0e-6):
    wp = SI["c"] / kp_inv
    return wp ** 2 * SI["m_e"] * SI["ep0"] / SI["q_e"] ** 2


def blowout_wake_SI():
    """examples/blowout_wake/inputs_SI (run next to the normalised deck by tests/blowout_wake.2Rank.sh): the same physics
    with kp_inv = 10 um."""
    kp_inv = 10.0e-6
    ne = _ne_SI(kp_inv)
    d = blowout_wake()
    d.update(si_units=1, lo=(-8.0 * kp_inv, -8.0 * kp_inv, -6.0 * kp_inv), hi=(8.0 * kp_inv, 8.0 * kp_inv, 6.0 * kp_inv),
             beam_zmin=-59.0e-6, beam_zmax=59.0e-6, beam_radius=12.0e-6, beam_density=3.0 * ne,
             beam_pos_std=(3.0e-6, 3.0e-6, 14.1e-6), beam_charge=-SI["q_e"], beam_mass=SI["m_e"],
             plasma_density=ne, plasma_charge=-SI["q_e"], plasma_mass=SI["m_e"])
    return d


def linear_wake_SI():
    """tests/linear_wake.SI.1Rank.sh (examples/linear_wake/inputs_SI + rho)."""
    d = linear_wake()
    ne = _ne_SI()
    d.update(si_units=1, lo=(-100.0e-6, -100.0e-6, -75.0e-6), hi=(100.0e-6, 100.0e-6, 20.0e-6),
             beam_zmin=-10.0e-6, beam_zmax=10.0e-6, beam_radius=30.0e-6, beam_density=0.01 * ne,
             beam_charge=-SI["q_e"], beam_mass=SI["m_e"], plasma_density=ne, plasma_charge=-SI["q_e"], plasma_mass=SI["m_e"])
    return d


def gaussian_linear_wake():
    """tests/gaussian_linear_wake.normalized.1Rank.sh: the linear_wake deck with a wide Gaussian beam (+ rho)"""
    d = linear_wake()
    d.update(beam_profile=0, beam_zmin=-5.9, beam_zmax=5.9, beam_radius=10.0, beam_pos_mean=(0.0, 0.0, 0.0), beam_pos_std=(2.0, 2.0, 1.41),
             lo=(-10.0, -10.0, -6.0), hi=(10.0, 10.0, 6.0), deposit_rho=1)
    return d


def gaussian_linear_wake_SI():
    """tests/gaussian_linear_wake.SI.1Rank.sh: the same in SI units (examples/linear_wake/inputs_SI)"""
    d = linear_wake_SI()
    d.update(beam_profile=0, beam_zmin=-59.0e-6, beam_zmax=59.0e-6, beam_radius=100.0e-6, beam_pos_mean=(0.0, 0.0, 0.0),
             beam_pos_std=(20.0e-6, 20.0e-6, 14.1e-6), lo=(-100.0e-6, -100.0e-6, -60.0e-6), hi=(100.0e-6, 100.0e-6, 60.0e-6), deposit_rho=1)
    return d


def radiation_reaction():
    """examples/beam_in_vacuum/inputs_RR (tests/radiation_reaction.1Rank.sh) in normalised units (n0 = 5e24 m^-3): a
    gamma = 2000 beam in the linear focusing field E = (x/2, y/2, 0) of a blowout, no z push, 50 sub-cycles, steps of
    30 / omega_beta; the beam comes from fixed_ppc (flat top of radius 2.5 / kp, at rest transversely) instead of the
    reference's random Gaussian."""
    d = beam_evolution()
    w_beta = 1.0 / (2.0 * 2000.0) ** 0.5
    d.update(nx=32, ny=32, nz=10, lo=(-12.6, -12.6, -4.2), hi=(12.6, 12.6, 4.2), beam_profile=1, beam_zmin=-2.0, beam_zmax=2.0,
             beam_radius=2.5, beam_density=1.0e-10, beam_umean=(0.0, 0.0, 2000.0), beam_ppc=(2, 2, 1), ext_E_slope=(0.5, 0.5),
             dt=30.0 / w_beta, n_steps=6, beam_n_subcycles=50, beam_radiation_reaction=1, background_density_SI=5.0e24,
             beam_no_z_push=1)
    return d


def beam_in_vacuum_1Rank():
    """tests/beam_in_vacuum.normalized.1Rank.sh: as the Serial run with hipace.MG_tolerance_rel = 1e-5."""
    d = beam_in_vacuum()
    d.update(mg_tol_rel=1.0e-5)
    return d


def beam_in_vacuum_SI():
    """tests/beam_in_vacuum.SI.1Rank.sh (examples/beam_in_vacuum/inputs_SI, order 0, MG_tolerance_rel = 1e-5).  The deck
    has two identical beams on top of each other; one beam of twice the density deposits the same currents."""
    d = beam_in_vacuum()
    d.update(si_units=1, lo=(-2000.0e-6, -2000.0e-6, -20.0e-6), hi=(2000.0e-6, 2000.0e-6, 20.0e-6),
             beam_zmin=-100.0e-6, beam_zmax=100.0e-6, beam_radius=10.0e-6, beam_density=2 * 1.4119793504295784e23,
             beam_charge=-SI["q_e"], beam_mass=SI["m_e"], mg_tol_rel=1.0e-5, order=0)
    return d


def grid_current():
    """tests/grid_current.1Rank.sh: the beam_in_vacuum deck at 32^3, order 0, a Gaussian beam of density 0.2 and a grid
    current (grid_current.*, utils/GridCurrent.cpp) of the same shape, which cancels most of the beam's current"""
    d = beam_in_vacuum()
    d.update(nx=32, ny=32, nz=32, lo=(-8.0, -8.0, -6.0), hi=(8.0, 8.0, 6.0), n_steps=2, order=0, beam_profile=0, beam_density=0.2,
             beam_radius=1.0, beam_pos_mean=(0.0, 0.0, 0.0), beam_pos_std=(0.3, 0.3, 1.41), beam_ppc=(1, 1, 1),
             grid_current_on=1, grid_current_peak=0.2, grid_current_mean=(0.0, 0.0, 0.0), grid_current_std=(0.3, 0.3, 1.41))
    return d


def beam_in_vacuum_SI_Serial():
    """tests/beam_in_vacuum.SI.Serial.sh: the SI deck with the multigrid solver's default tolerance."""
    d = beam_in_vacuum_SI()
    d.update(mg_tol_rel=beam_in_vacuum()["mg_tol_rel"])
    return d


def reset():
    """tests/reset.2Rank.sh: the blowout deck for three time steps (max_step = 2, dt = 0) with MG_tolerance_rel = 1e-5"""
    d = blowout_wake()
    d.update(n_steps=3, mg_tol_rel=1.0e-5)
    return d


def blowout_wake_step0():
    """tests/blowout_wake.Serial.sh: examples/blowout_wake/inputs_normalized as it stands (max_step = 0: one time step)."""
    d = blowout_wake()
    d["n_steps"] = 1
    return d


def predictor_corrector(base, tol=1.0e-4, max_iter=7, mix=0.0635):
    """`base` with hipace.bxby_solver = predictor-corrector; the defaults are the settings of the reference's own
    predictor-corrector-vs-explicit test (tests/ion_motion.SI.1Rank.sh:30-34)."""
    d = copy.deepcopy(base)
    d.update(bxby_solver=1, predcorr_tol=tol, predcorr_max_iter=max_iter, predcorr_mix=mix)
    return d


NAMED = dict(radiation_reaction=radiation_reaction, gaussian_linear_wake=gaussian_linear_wake, gaussian_linear_wake_SI=gaussian_linear_wake_SI, reset=reset, grid_current=grid_current, beam_in_vacuum_SI_Serial=beam_in_vacuum_SI_Serial, blowout_wake_step0=blowout_wake_step0, linear_wake_gaussian=linear_wake_gaussian, laser_blowout_wake=laser_blowout_wake, laser_blowout_wake_SI=laser_blowout_wake_SI, linear_wake_SI=linear_wake_SI, blowout_wake_SI=blowout_wake_SI, beam_in_vacuum_SI=beam_in_vacuum_SI, beam_in_vacuum_1Rank=beam_in_vacuum_1Rank, blowout_wake=blowout_wake, linear_wake=linear_wake, beam_in_vacuum=beam_in_vacuum,
             beam_evolution=beam_evolution, beam_in_vacuum_open_boundary=beam_in_vacuum_open_boundary)


def ionization_SI():
    """examples/blowout_wake/inputs_ionization_SI as tests/ionization.2Rank.sh runs it (hipace.dt = 1e-12, max_step = 2):
    neutral hydrogen (one macro-atom per cell), no electrons to begin with (elec.ppc = 0 0), a flat-top driver whose field
    ionises the gas; the released electrons form the wake."""
    ne = 1.25e24
    c, ep0, q_e, m_e = 299792458.0, 8.8541878128e-12, 1.602176634e-19, 9.1093837015e-31
    kp_inv = c / (ne * q_e * q_e / (ep0 * m_e)) ** 0.5
    d = copy.deepcopy(_DEFAULT)
    d.update(nx=64, ny=64, nz=100, lo=(-20.0e-6, -20.0e-6, -30.0e-6), hi=(20.0e-6, 20.0e-6, 30.0e-6), order=2, si_units=1,
             plasma_ppc=(0, 0), plasma_density=ne, plasma_charge=-q_e, plasma_mass=m_e, plasma_no_neutralize=1,
             beam_profile=1, beam_zmin=25.0e-6 - 2.0 * kp_inv, beam_zmax=25.0e-6, beam_radius=kp_inv / 2.0, beam_density=4.0 * ne,
             beam_umean=(0.0, 0.0, 2000.0), beam_ppc=(1, 1, 1), beam_charge=-q_e, beam_mass=m_e,
             n_steps=3, dt=1.0e-12, bc=1)
    return with_ion_species(d, "H", ne, ppc=(1, 1), mass_Da=1.008, initial_level=0)


def ion_motion_SI(nz=200):
    """examples/linear_wake/inputs_ion_motion_SI (tests/ion_motion.SI.1Rank.sh): electrons and MOBILE ions (mass 5 m_e "for
    testing", one particle per cell each, no neutralising background for either) behind an off-axis driver.  The ions are
    the second species at its top level (hydrogen, level 1 of 1: nothing left to ionise).  The reference's driver is a
    fixed_weight Gaussian drawn from amrex::Random; here a flat-top of the same peak density, length and offset (the
    deck pins the two-species push and deposition, not the beam's random positions)."""
    c, ep0, q_e, m_e = 299792458.0, 8.8541878128e-12, 1.602176634e-19, 9.1093837015e-31
    kp_inv = 10.0e-6
    ne = (c / kp_inv) ** 2 * m_e * ep0 / (q_e * q_e)
    d = copy.deepcopy(_DEFAULT)
    d.update(nx=64, ny=64, nz=nz, lo=(-8 * kp_inv, -8 * kp_inv, -6 * kp_inv), hi=(8 * kp_inv, 8 * kp_inv, 6 * kp_inv), order=2, si_units=1,
             plasma_ppc=(1, 1), plasma_density=ne, plasma_charge=-q_e, plasma_mass=m_e, plasma_no_neutralize=1,
             beam_profile=1, beam_zmin=0.6 * kp_inv, beam_zmax=3.4 * kp_inv, beam_radius=0.8 * kp_inv, beam_density=ne,
             beam_pos_mean=(0.25 * kp_inv, 0.0, 2.0 * kp_inv), beam_umean=(10.0, 20.0, 100.0), beam_ppc=(1, 1, 1), beam_charge=-q_e, beam_mass=m_e,
             n_steps=1, dt=0.0, bc=1)
    with_ion_species(d, "H", ne, ppc=(1, 1), initial_level=1)
    d["ion_mass"] = 5.0 * m_e
    return d


def laser_ionization_SI():
    """BASELINE config 5 at test size: the laser-driven wake of tests/laser_blowout_wake_explicit.SI.1Rank.sh in a gas that
    also holds neutral nitrogen (a fifth of the electron density, one macro-atom per cell) -- the wake's field ionises the
    atoms it reaches (ADK), the released electrons join the plasma electrons.  Neutral atoms and electron + ion pairs
    carry no net charge, so the pre-formed plasma stays neutralised by its own background."""
    d = laser_blowout_wake_SI()
    return with_ion_species(d, "N", 0.2 * d["plasma_density"], ppc=(1, 1), initial_level=0, seed=5)


def transverse_benchmark(nxy=1023, nz=1000):
    """examples/benchmarks/inputs_transverse_benchmark as tests/transverse_benchmark.1Rank.sh runs it (my_constants.nxy = 1023):
    the reference's own transverse scaling benchmark -- 2^N - 1 cells per side, 1000 slices, one plasma electron per cell,
    absorbing particle boundary, explicit solver.  Its driver is a fixed_weight_pdf beam of 10 nxy^2 particles drawn from
    amrex::Random: the deck carries no beam (beam_profile = -1), the host hands one in with
    SliceEngine.set_beam_particles(fixed_weight_pdf_beam(deck, **TRANSVERSE_BENCHMARK_BEAM(nxy)))."""
    d = copy.deepcopy(_DEFAULT)
    d.update(nx=nxy, ny=nxy, nz=nz, lo=(-6.0, -6.0, -12.0), hi=(6.0, 6.0, 6.0), order=2, plasma_ppc=(1, 1), plasma_density=1.0,
             beam_profile=-1, bc=2, n_steps=1, dt=0.0)
    return d


def TRANSVERSE_BENCHMARK_BEAM(nxy=1023):
    import numpy as np
    return dict(num_particles=nxy * nxy * 10, density=2.0, pdf=lambda z: np.exp(-0.5 * (z / 1.41) ** 2), pos_std=(0.3, 0.3),
                u_mean=(0.0, 0.0, 2000.0))


def fixed_weight_pdf_beam(deck, num_particles, density, pdf, pos_mean=(0.0, 0.0), pos_std=(1.0, 1.0), u_mean=(0.0, 0.0, 0.0),
                          u_std=(0.0, 0.0, 0.0), seed=0, pdf_ref_ratio=4):
    """beam.injection_type = fixed_weight_pdf with a peak density (InitBeamFixedWeightPDF3D / ...PDFSlice,
    particles/beam/BeamParticleContainerInit.cpp:479-695), on the host: the longitudinal profile `pdf(z)` (vectorised
    callable) is integrated by the trapezoidal rule on nz * pdf_ref_ratio sub-slices (:502-528), the total weight is
    density * integral / max_density with max_density = max local_weight / (dz_sub * sigma_x * sigma_y * 2 pi) (:514-531), in
    normalised units over the cell volume (:540-542); every particle carries total / num_particles (:618).  The particle
    count of a sub-slice is drawn in proportion to its share of the integral (the reference iterates Poisson draws until
    the total is num_particles, :546-579 -- a multinomial draw here), z inside a sub-slice follows the linear profile
    between its two ends (:645-652), x, y and u are normal (:663-670).  numpy's generator stands in for amrex::Random: the
    beam is the reference's in distribution, not particle by particle.
    -> (7, num_particles) array x y z ux uy uz w in the engine's units (u times c; AddOneBeamParticleSlice, :86-116)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    nz = deck["nz"]
    lo, hi = deck["lo"], deck["hi"]
    dx, dy, dz = (hi[0] - lo[0]) / deck["nx"], (hi[1] - lo[1]) / deck["ny"], (hi[2] - lo[2]) / nz
    c = 299792458.0 if deck.get("si_units", 0) else 1.0
    ns = nz * pdf_ref_ratio
    zs = dz / pdf_ref_ratio
    edges = lo[2] + zs * np.arange(ns + 1)
    pe = np.asarray(pdf(edges), dtype=np.float64)
    assert (pe >= 0.0).all(), "PDF must be >= 0 everywhere"
    lw = 0.5 * (pe[:-1] + pe[1:])
    integral = lw.sum()
    max_density = (lw / (zs * pos_std[0] * pos_std[1] * 2.0 * np.pi)).max()
    total_weight = density * integral / max_density
    if not deck.get("si_units", 0):
        total_weight /= dx * dy * dz
    counts = rng.multinomial(num_particles, lw / integral)
    sub = np.repeat(np.arange(ns), counts)
    w01 = rng.random(num_particles)
    lo_w, hi_w = pe[:-1][sub], pe[1:][sub]
    taylor = np.minimum(lo_w, hi_w) * 1.1 > np.maximum(lo_w, hi_w)
    with np.errstate(divide="ignore", invalid="ignore"):
        z_t = w01 - w01 * (w01 - 1.0) * (hi_w - lo_w) / (hi_w + lo_w)
        z_s = (np.sqrt(lo_w * lo_w + w01 * (hi_w * hi_w - lo_w * lo_w)) - lo_w) / (hi_w - lo_w)
    z = edges[:-1][sub] + zs * np.where(taylor, z_t, z_s)
    out = np.empty((7, num_particles))
    out[0] = pos_mean[0] + rng.normal(0.0, pos_std[0], num_particles)
    out[1] = pos_mean[1] + rng.normal(0.0, pos_std[1], num_particles)
    out[2] = z
    for k in range(3):
        out[3 + k] = (u_mean[k] + (rng.normal(0.0, u_std[k], num_particles) if u_std[k] else 0.0)) * c
    out[6] = abs(total_weight / num_particles)
    return out


def fixed_weight_beam(deck, num_particles, density, pos_mean, pos_std, u_mean=(0.0, 0.0, 0.0), u_std=(0.0, 0.0, 0.0),
                      zmin=-float("inf"), zmax=float("inf"), radius=float("inf"), seed=0, total_charge=None, do_symmetrize=False,
                      duz_per_uz0_dzeta=0.0):
    """beam.injection_type = fixed_weight, profile = gaussian, with a peak density (BeamParticleContainer.cpp:137-198,
    InitBeamFixedWeight3D / InitBeamFixedWeightSlice, BeamParticleContainerInit.cpp:350-477), on the host: z is normal about
    pos_mean[2] (:375-377), x and y are normal about pos_mean[0](z), pos_mean[1](z) -- numbers or vectorised callables of z
    (:435-454) --, every particle carries density (2 pi)^(3/2) sigma_x sigma_y sigma_z / num_particles, in normalised units
    over the cell volume (BeamParticleContainer.cpp:179-190, :420).  Particles outside [zmin, zmax] or the radius are invalid
    in the reference (:443-446): left out here; particles outside the box are left out by set_beam_particles.  numpy's
    generator stands in for amrex::Random.  -> (7, n <= num_particles) array x y z ux uy uz w, u times c.
    do_symmetrize: num_particles / 4 draws, each placed four times at (+-x, +-y) about the mean with (+-ux, +-uy) (:457-470);
    duz_per_uz0_dzeta: uz += (z - z_mean) duz_per_uz0_dzeta u_mean[2] (GetInitialMomentum.H:47)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    lo, hi = deck["lo"], deck["hi"]
    dx, dy, dz = (hi[0] - lo[0]) / deck["nx"], (hi[1] - lo[1]) / deck["ny"], (hi[2] - lo[2]) / deck["nz"]
    c = 299792458.0 if deck.get("si_units", 0) else 1.0
    if total_charge is not None:          # <beam>.total_charge (SI units only, BeamParticleContainer.cpp:167-172): density is not read
        assert deck.get("si_units", 0) and density is None
        weight = abs(total_charge / deck["beam_charge"]) / num_particles
    else:
        weight = density * (2.0 * np.pi) ** 1.5 * pos_std[0] * pos_std[1] * pos_std[2] / num_particles
    if not deck.get("si_units", 0):
        weight /= dx * dy * dz
    nd = num_particles // 4 if do_symmetrize else num_particles
    assert not do_symmetrize or num_particles % 4 == 0, "to symmetrize the beam its particle number must be divisible by 4"
    z = rng.normal(pos_mean[2], pos_std[2], nd)
    x = rng.normal(0.0, pos_std[0], nd)
    y = rng.normal(0.0, pos_std[1], nd)
    ok = (z >= zmin) & (z <= zmax) & (x * x + y * y <= radius * radius)
    mx = pos_mean[0](z) if callable(pos_mean[0]) else pos_mean[0]
    my = pos_mean[1](z) if callable(pos_mean[1]) else pos_mean[1]
    u = [u_mean[k] + (rng.normal(0.0, u_std[k], nd) if u_std[k] else np.zeros(nd)) for k in range(3)]
    u[2] = u[2] + (z - pos_mean[2]) * duz_per_uz0_dzeta * u_mean[2]
    parts = []
    for sx, sy in (((1, 1), (-1, 1), (1, -1), (-1, -1)) if do_symmetrize else ((1, 1),)):
        o = np.empty((7, nd))
        o[0], o[1], o[2] = mx + sx * x, my + sy * y, z
        o[3], o[4], o[5] = sx * u[0] * c, sy * u[1] * c, u[2] * c
        o[6] = abs(weight)
        parts.append(o[:, ok])
    return np.ascontiguousarray(np.concatenate(parts, axis=1))


def ion_motion_SI_reference_beam(deck, seed=1):
    """the driver of examples/linear_wake/inputs_ion_motion_SI: fixed_weight, 10^6 particles, peak density ne, a Gaussian of
    (0.4, 0.4, 1.41) kp^-1 about (0.25 kp^-1, 0.2 (z - 2 kp^-1), 2 kp^-1), u = (10, 20, 100)"""
    kp_inv = 10.0e-6
    return fixed_weight_beam(deck, 1000000, deck["plasma_density"], (0.25 * kp_inv, lambda z: (z - 2.0 * kp_inv) * 0.2, 2.0 * kp_inv),
                             (0.4 * kp_inv, 0.4 * kp_inv, 1.41 * kp_inv), u_mean=(10.0, 20.0, 100.0), seed=seed)


def production_lwfa(n=64, nz=100, max_step=10):
    """examples/get_started/inputs_lwfa as tests/production.SI.2Rank.sh runs it (amr.n_cell = 64 64 100, max_step = 10): a laser
    pulse (a0 = 1.9, w0 = 3 kp^-1, L0 = 0.5 kp^-1, multigrid envelope solver, MG_tolerance_rel = 1e-5) enters a parabolic plasma
    channel of radius 23 kp^-1 through a cosine up-ramp of 6 mm, steps of c dt = 10 kp^-1, SI units, ne = 1.0505e23 m^-3.
    The density function n_e (1 + 4 r^2 / (kp^2 Rm^4)) ramp(c t) [c t > 0] is the engine's tabulated form f_r(r) f_t(c t):
    -> (deck with plasma_radius, (r, fr, ct, ft)) for SliceEngine.set_density_profile -- the radial table on the lattice's own
    radii (exact at every particle), the time table on the steps' own times."""
    import numpy as np
    ne = 1.0505e23
    wp = (ne * SI["q_e"] ** 2 / (SI["m_e"] * SI["ep0"])) ** 0.5
    kp_inv = SI["c"] / wp
    kp = wp / SI["c"]
    Rm, Lramp = 3.0 * kp_inv, 6.0e-3
    d = copy.deepcopy(_DEFAULT)
    d.update(nx=n, ny=n, nz=nz, lo=(-18.0 * kp_inv, -18.0 * kp_inv, -7.5 * kp_inv), hi=(18.0 * kp_inv, 18.0 * kp_inv, 1.5 * kp_inv),
             order=2, si_units=1, plasma_ppc=(1, 1), plasma_density=ne, plasma_charge=-SI["q_e"], plasma_mass=SI["m_e"],
             beam_profile=-1, bc=1, n_steps=max_step + 1, dt=10.0 * kp_inv / SI["c"],
             laser_on=1, laser_a0=1.9, laser_w0=3.0 * kp_inv, laser_L0=0.5 * kp_inv, laser_lambda0=800.0e-9, laser_pos=(0.0, 0.0, 0.0),
             laser_solver=2, laser_mg_tol_rel=1.0e-5)
    dx = (d["hi"][0] - d["lo"][0]) / n
    xs = d["lo"][0] + (np.arange(n) + 0.5) * dx
    r2 = np.unique(np.add.outer(xs * xs, xs * xs).round(decimals=22))
    r = np.sqrt(r2)
    d["plasma_radius"] = 23.0 * kp_inv                   # plasma.radius: no particles in the box's corners
    r_tab = r
    fr = 1.0 + 4.0 * r_tab ** 2 / (kp ** 2 * Rm ** 4)
    if r_tab[0] > 0.0:
        r_tab = np.concatenate([[0.0], r_tab]); fr = np.concatenate([[1.0], fr])
    ct = np.array([SI["c"] * d["dt"] * k for k in range(max_step + 2)])
    ft = np.where(ct > 0.0, np.where(ct > Lramp, 1.0, 0.5 * (1.0 - np.cos(np.pi * ct / Lramp))), 0.0)
    return d, (r_tab, fr, ct, ft)


def gaussian_weight_SI():
    """examples/gaussian_weight/inputs_SI as the last run of tests/gaussian_weight.1Rank.sh (the one its checksum file holds): a
    fixed_weight beam of 10^5 particles and 1 nC, a Gaussian of (30, 40, 50) um about (0, 10, 20) um, in vacuum on 64^3 cells,
    absorbing particle boundary.  -> (deck without a beam, the beam's parameters for fixed_weight_beam)"""
    d = copy.deepcopy(_DEFAULT)
    d.update(nx=64, ny=64, nz=64, lo=(-200.0e-6, -200.0e-6, -200.0e-6), hi=(200.0e-6, 200.0e-6, 200.0e-6), order=2, si_units=1,
             plasma_ppc=(0, 0), plasma_density=0.0, plasma_charge=-SI["q_e"], plasma_mass=SI["m_e"],
             beam_profile=-1, beam_charge=-SI["q_e"], beam_mass=SI["m_e"], bc=2, n_steps=1, dt=0.0)
    beam = dict(num_particles=100000, density=None, total_charge=1.0e-9, pos_mean=(0.0, 10.0e-6, 20.0e-6),
                pos_std=(30.0e-6, 40.0e-6, 50.0e-6), u_mean=(0.0, 0.0, 1.0e3), zmin=-1.0, zmax=1.0, radius=1.0)
    return d, beam


def radiation_reaction_SI():
    """examples/beam_in_vacuum/inputs_RR as tests/radiation_reaction.1Rank.sh runs it: a matched gamma = 2000 beam sheet
    (sigma_y = 1e-12 m) of 10^5 fixed_weight particles in the linear focusing field E = (kp E0 / 2)(x, y, 0) of a blowout at
    n0 = 5e24 m^-3, SI units, 16 x 16 x 4 cells, six steps of 30 / omega_beta with 50 sub-cycles, radiation reaction on, no z push.
    -> (deck without a beam, the beam's parameters for fixed_weight_beam)"""
    ne = 5.0e24
    wp = (ne * SI["q_e"] ** 2 / (SI["ep0"] * SI["m_e"])) ** 0.5
    E0 = wp * SI["m_e"] * SI["c"] / SI["q_e"]
    kp = wp / SI["c"]
    K, gamma0, emit = kp / 2.0 ** 0.5, 2000.0, 313.0e-6
    sigma_x = (emit / kp / (gamma0 / 2.0) ** 0.5) ** 0.5
    sigma_ux = emit / sigma_x
    uz = (gamma0 ** 2 - 1.0 - sigma_ux ** 2) ** 0.5
    w_beta = K * SI["c"] / gamma0 ** 0.5
    d = copy.deepcopy(_DEFAULT)
    d.update(nx=16, ny=16, nz=4, lo=(-30.0e-6, -30.0e-6, -10.0e-6), hi=(30.0e-6, 30.0e-6, 10.0e-6), order=2, si_units=1,
             plasma_ppc=(0, 0), plasma_density=0.0, plasma_charge=-SI["q_e"], plasma_mass=SI["m_e"],
             beam_profile=-1, beam_charge=-SI["q_e"], beam_mass=SI["m_e"], bc=1, n_steps=6, dt=30.0 / w_beta,
             ext_E_slope=(0.5 * kp * E0, 0.5 * kp * E0), beam_n_subcycles=50, beam_radiation_reaction=1, beam_no_z_push=1)
    beam = dict(num_particles=100000, density=ne / 1.0e10, pos_mean=(0.0, 0.0, 0.0), pos_std=(sigma_x, 1.0e-12, 1.0e-6),
                u_mean=(0.0, 0.0, uz), u_std=(sigma_ux, 0.0, uz * 0.01))
    return d, beam


def hosing():
    """tests/hosing.2Rank.sh: the blowout deck with hipace.dt = 20, mobile ions (charge 1, mass 1836, one per cell, neither species
    with a neutralising background) and a tilted fixed_weight driver (beam.dx_per_dzeta = 0.2, density 200, sigma 0.1, 0.1, 1.41) --
    the deck of the hosing instability.  (Its checksum file predates the explicit solver's field set: parity is with the oracle.)
    -> (deck without a beam, the beam's parameters for fixed_weight_beam; the reference draws 10^6 particles)"""
    d = blowout_wake()
    d.update(n_steps=11, dt=20.0, beam_profile=-1, plasma_no_neutralize=1, background_density_SI=1.0e23)
    with_ion_species(d, "H", 1.0, ppc=(1, 1), initial_level=1)
    d["ion_mass"] = 1836.0
    beam = dict(num_particles=1000000, density=200.0, pos_mean=(lambda z: 0.2 * z, 0.0, 0.0), pos_std=(0.1, 0.1, 1.41),
                u_mean=(0.0, 0.0, 2000.0))
    return d, beam
