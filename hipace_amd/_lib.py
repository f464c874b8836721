"""ctypes binding of libhpslice.so (include/hpslice.h).  Fails loudly when the HIP library is
missing: there is no CPU fallback for any product path."""
import ctypes as C
import os
import subprocess


def launched_ranks():
    """How many ranks the launcher says there are: torchrun (WORLD_SIZE), Open MPI, PMI / Hydra, Slurm."""
    n = 1
    for k in ("WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "SLURM_NTASKS"):
        try:
            n = max(n, int(os.environ.get(k, "1") or 1))
        except ValueError:
            pass
    return n


def _ring_environment():
    """RCCL edges only (HPS_RING_EDGE=rccl; the default ipc edge has no device-resident waiter and needs none of this).  A ring
    of 2+ ranks posts its receives ahead as RCCL kernels that wait on the device: the ring's two streams and the engine's
    stream must each get a hardware queue of their own (hps_ring_init refuses to start otherwise), and the runtime reads
    GPU_MAX_HW_QUEUES once, when the process first touches HIP.  Any multi-rank launch therefore gets the variable HERE, at
    the import of the binding -- before libhpslice.so is loaded, and before torch initialises the device unless the host has
    already done so (RingTransport checks that and says so)."""
    if launched_ranks() > 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def hip_already_started():
    """True if this process has (or may have) touched HIP: torch has initialised the device, or libhpslice.so has been
    loaded and used (an engine, a solver: hipSetDevice) -- GPU_MAX_HW_QUEUES set now would come too late."""
    import sys
    if _LIB is not None:
        return True
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:      # noqa: BLE001
        return False


_LIB = None
_HWQ_PRESET = "GPU_MAX_HW_QUEUES" in os.environ     # set by the host's environment, before this process started
_HIP_STARTED_AT_IMPORT = hip_already_started()
_ring_environment()
_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO = os.environ.get("HPS_LIB") or os.path.join(CSRC, "libhpslice.so")      # HPS_LIB: a diagnostic build (make stamps)


class Slab(C.Structure):
    _fields_ = [("p", C.c_void_p), ("nx", C.c_int), ("ny", C.c_int), ("ng", C.c_int), ("ncomp", C.c_int),
                ("jstride", C.c_long), ("nstride", C.c_long)]


PL_REAL = ["x", "y", "w", "ux", "uy", "psi", "x_prev", "y_prev", "ux_half", "uy_half", "psi_half"]


class Plasma(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in PL_REAL] + [("idcpu", C.c_void_p), ("ion_lev", C.c_void_p), ("n", C.c_long)]


class Geom(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("dx", "dy", "dz", "xoff", "yoff", "c", "ep0", "mu0", "q_e", "m_e")] + \
               [("plo", C.c_double * 2), ("phi", C.c_double * 2), ("bc", C.c_int), ("normalized", C.c_int)]


class Deck(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
                ("lo", C.c_double * 3), ("hi", C.c_double * 3),
                ("order", C.c_int), ("deriv_type", C.c_int),
                ("plasma_ppc", C.c_int * 2), ("plasma_density", C.c_double), ("plasma_radius", C.c_double),
                ("plasma_charge", C.c_double), ("plasma_mass", C.c_double), ("max_qsa", C.c_double),
                ("n_subcycles", C.c_int),
                ("beam_profile", C.c_int), ("beam_zmin", C.c_double), ("beam_zmax", C.c_double),
                ("beam_radius", C.c_double), ("beam_density", C.c_double),
                ("beam_umean", C.c_double * 3), ("beam_pos_mean", C.c_double * 3),
                ("beam_pos_std", C.c_double * 3), ("beam_ppc", C.c_int * 3), ("beam_charge", C.c_double),
                ("bc", C.c_int), ("mg_tol_rel", C.c_double), ("mg_tol_abs", C.c_double),
                ("deposit_rho", C.c_int), ("n_steps", C.c_int),
                ("dt", C.c_double), ("beam_n_subcycles", C.c_int), ("beam_mass", C.c_double), ("ext_E_slope", C.c_double * 2),
                ("bxby_solver", C.c_int), ("predcorr_tol", C.c_double), ("predcorr_max_iter", C.c_int),
                ("predcorr_mix", C.c_double), ("field_bc", C.c_int),
                ("laser_on", C.c_int), ("laser_a0", C.c_double), ("laser_w0", C.c_double), ("laser_L0", C.c_double),
                ("laser_lambda0", C.c_double), ("laser_pos", C.c_double * 3),
                ("laser_zfoc", C.c_double), ("laser_solver", C.c_int), ("laser_use_phase", C.c_int), ("si_units", C.c_int),
                ("grid_current_on", C.c_int), ("grid_current_peak", C.c_double), ("grid_current_mean", C.c_double * 3),
                ("grid_current_std", C.c_double * 3), ("laser_mg_tol_rel", C.c_double), ("laser_mg_tol_abs", C.c_double),
                ("beam_radiation_reaction", C.c_int), ("background_density_SI", C.c_double), ("beam_no_z_push", C.c_int),
                ("plasma_no_neutralize", C.c_int), ("ion_on", C.c_int), ("ion_ppc", C.c_int * 2), ("ion_density", C.c_double),
                ("ion_mass", C.c_double), ("ion_charge", C.c_double), ("ion_init_level", C.c_int), ("ion_Z", C.c_int),
                ("ion_energies", C.c_double * 56), ("ion_seed", C.c_ulonglong),
                ("beam_spin_tracking", C.c_int), ("beam_initial_spin", C.c_double * 3), ("beam_spin_anom", C.c_double)]


# engine component names, index = value of the HPS_C_* enum in include/hpslice.h
COMPS = ["N_jx_beam", "N_jy_beam", "chi", "Sy", "Sx", "ExmBy", "EypBx", "Ez", "Bx", "By", "Bz",
         "Psi", "jx_beam", "jy_beam", "jz_beam", "jx", "jy", "rhomjz", "P_jx_beam", "P_jy_beam",
         "Ion_rhomjz", "rho"]
CIDX = {n: i for i, n in enumerate(COMPS)}
# predictor-corrector layout, index = value of the HPS_PC_* enum
COMPS_PC = ["N_jx", "N_jy", "ExmBy", "EypBx", "Ez", "Bx", "By", "Bz", "Psi", "jx", "jy", "jz", "rhomjz",
            "P_Bx", "P_By", "P_jx", "P_jy", "Ion_rhomjz", "It_Bx", "It_By", "PIt_Bx", "PIt_By", "rho"]
CIDX_PC = {n: i for i, n in enumerate(COMPS_PC)}
ID_VALID = 1 << 63

_SIGS = {
    "hps_last_error": (C.c_char_p, []),
    "hps_version": (C.c_char_p, []),
    "hps_deposit_current": (C.c_int, [Slab, Plasma, Geom, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_double,
                                      C.c_int, C.c_void_p, C.c_void_p]),
    "hps_explicit_deposit": (C.c_int, [Slab, Plasma, Geom, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p]),
    "hps_advance_plasma": (C.c_int, [Slab, Plasma, Geom, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p]),
    "hps_tiling_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_long, C.POINTER(C.c_void_p)]),
    "hps_reorder_particles": (C.c_int, [C.c_void_p, Plasma, Plasma, Geom, C.c_void_p]),
    "hps_tiling_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "hps_tiling_destroy": (C.c_int, [C.c_void_p]),
    "hps_deposit_current_tiled": (C.c_int, [Slab, Plasma, Geom, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_double,
                                            C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hps_explicit_deposit_tiled": (C.c_int, [Slab, Plasma, Geom, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int,
                                             C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hps_advance_plasma_tiled": (C.c_int, [Slab, Plasma, Geom, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hps_poisson_create": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]),
    "hps_poisson_solve": (C.c_int, [C.c_void_p, C.c_void_p, Slab, C.c_int, C.c_void_p]),
    "hps_poisson_solve_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, Slab, C.c_void_p, C.c_void_p]),
    "hps_poisson_destroy": (C.c_int, [C.c_void_p]),
    "hps_mg_create": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]),
    "hps_mg_solve1": (C.c_int, [C.c_void_p, Slab, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p]),
    "hps_mg_solve1_fabs": (C.c_int, [C.c_void_p, Slab, Slab, Slab, C.c_double, C.c_double, C.c_int,
                                     C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p]),
    "hps_mg_destroy": (C.c_int, [C.c_void_p]),
    "hps_mg2_create": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]),
    "hps_mg2_solve2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int,
                                 C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p]),
    "hps_mg2_destroy": (C.c_int, [C.c_void_p]),
    "hps_engine_laser_vcycles": (C.c_int, [C.c_void_p, C.POINTER(C.c_long)]),
    "hps_engine_create": (C.c_int, [C.POINTER(Deck), C.c_int, C.POINTER(C.c_void_p)]),
    "hps_engine_destroy": (C.c_int, [C.c_void_p]),
    "hps_engine_begin_step": (C.c_int, [C.c_void_p]),
    "hps_engine_solve_slice": (C.c_int, [C.c_void_p, C.c_int]),
    "hps_engine_solve_slice_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "hps_engine_solve_slice_finish": (C.c_int, [C.c_void_p, C.c_int]),
    "hps_engine_slice_ready": (C.c_int, [C.c_void_p]),
    "hps_engine_run_step": (C.c_int, [C.c_void_p]),
    "hps_engine_sync": (C.c_int, [C.c_void_p]),
    "hps_engine_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_long)]),
    "hps_engine_slab": (Slab, [C.c_void_p]),
    "hps_engine_plasma": (Plasma, [C.c_void_p]),
    "hps_engine_stream": (C.c_void_p, [C.c_void_p]),
    "hps_engine_ions": (Plasma, [C.c_void_p]),
    "hps_engine_ion_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "hps_engine_checksums": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hps_beam_sort_by_box": (C.c_int, [C.c_void_p, C.c_long, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "hps_engine_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "hps_engine_set_insitu_plasma": (C.c_int, [C.c_void_p, C.c_double]),
    "hps_engine_set_insitu_beam": (C.c_int, [C.c_void_p, C.c_double]),
    "hps_engine_insitu_beam": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hps_engine_insitu_plasma": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hps_engine_set_insitu_fields": (C.c_int, [C.c_void_p, C.c_int]),
    "hps_engine_insitu_fields": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hps_engine_set_field_diagnostic": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hps_engine_field_diagnostic": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hps_engine_set_field_diagnostic_box": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hps_engine_field_diagnostic_geometry": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hps_engine_record_event": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "hps_engine_wait_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hps_stream_pool_shared_pairs": (C.c_int, [C.c_int]),
    "hps_engine_copy_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]),
    "hps_engine_set_profiling_stride": (C.c_int, [C.c_void_p, C.c_int]),
    "hps_engine_set_laser_import": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "hps_engine_set_step": (C.c_int, [C.c_void_p, C.c_int]),
    "hps_engine_export_laser_slice": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "hps_engine_import_laser_slice": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "hps_engine_import_laser_from": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "hps_engine_laser_envelope": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hps_engine_laser_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "hps_deposit_current_laser": (C.c_int, [Slab, Plasma, Geom, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double,
                                            C.c_int, C.c_void_p, C.c_void_p]),
    "hps_explicit_deposit_laser": (C.c_int, [Slab, Plasma, Geom, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_int,
                                             C.c_int, C.c_int, C.c_void_p]),
    "hps_advance_plasma_laser": (C.c_int, [Slab, Plasma, Geom, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_void_p]),
    "hps_engine_pc_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_double)]),
    "hps_engine_pc_zero_b_slices": (C.c_int, [C.c_void_p, C.POINTER(C.c_long)]),
    "hps_engine_set_diagnostics": (C.c_int, [C.c_void_p, C.c_int]),
    "hps_engine_set_tiling": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "hps_engine_fallbacks": (C.c_int, [C.c_void_p, C.POINTER(C.c_long)]),
    "hps_engine_set_fusion": (C.c_int, [C.c_void_p, C.c_int]),
    "hps_engine_set_density_profile": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hps_engine_sorts": (C.c_int, [C.c_void_p, C.POINTER(C.c_long)]),
    "hps_engine_assume_initial_beam_support": (C.c_int, [C.c_void_p]),
    "hps_engine_beam_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hps_engine_beam_capacity": (C.c_int, [C.c_void_p, C.POINTER(C.c_long)]),
    "hps_engine_set_beam_capacity": (C.c_int, [C.c_void_p, C.c_long]),
    "hps_engine_beam_message_rows": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "hps_engine_beam_spin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hps_engine_set_beam_import": (C.c_int, [C.c_void_p, C.c_int]),
    "hps_engine_export_beam_slice": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "hps_engine_import_beam_slice": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "hps_engine_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "hps_engine_phase_times": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hps_engine_beam_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_long), C.c_void_p]),
    "hps_engine_set_beam_particles": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.POINTER(C.c_long)]),
    "hps_engine_set_beam_storage": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hps_engine_initial_beam": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hps_ring_unique_id": (C.c_int, [C.c_char_p]),
    "hps_ring_init": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "hps_ring_send_slice": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "hps_ring_recv_slice": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "hps_ring_sendrecv_self": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "hps_ring_stream_wait": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "hps_ring_sync_sends": (C.c_int, [C.c_void_p]),
    "hps_ring_sync": (C.c_int, [C.c_void_p]),
    "hps_ring_sync_timeout": (C.c_int, [C.c_void_p, C.c_double]),
    "hps_ring_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "hps_engine_tiling": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "hps_ring_info": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int)] * 5),
    "hps_ring_destroy": (C.c_int, [C.c_void_p]),
    "hps_ring_edge_kind": (C.c_int, [C.c_void_p]),
    "hps_ring_can_send": (C.c_int, [C.c_void_p]),
    "hps_ring_recv_landed": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hps_ring_engine_wait": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hps_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long]),
    "hps_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long]),
    "hps_device_count": (C.c_int, [C.POINTER(C.c_int)]),
}


def build(force=False):
    """Compile libhpslice.so for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args + ["libhpslice.so"], stdout=subprocess.DEVNULL)
    return SO


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO):
            raise RuntimeError(
                f"{SO} is missing: the HIP extension was not built (run __graft_entry__.build()); "
                "hipace_amd has no CPU fallback")
        L = C.CDLL(SO)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)      # AttributeError here = the library does not export the ABI
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


class HpsError(RuntimeError):
    pass


def check(status):
    if status != 0:
        raise HpsError(f"hpslice status {status}: {lib().hps_last_error().decode()}")


def fill_struct(st, d):
    for name, typ in st._fields_:
        if name not in d:
            continue
        v = d[name]
        if hasattr(typ, "_length_"):
            setattr(st, name, typ(*v))
        else:
            setattr(st, name, v)
    return st
