"""ctypes binding of the CPU oracle (oracle/hps_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under hipace_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# component indices of the oracle engine's slab (explicit-solver layout, fields/Fields.cpp:70-122;
# optional "rho" last)
COMPS = ["N_jx_beam", "N_jy_beam", "chi", "Sy", "Sx", "ExmBy", "EypBx", "Ez", "Bx", "By", "Bz",
         "Psi", "jx_beam", "jy_beam", "jz_beam", "jx", "jy", "rhomjz", "P_jx_beam", "P_jy_beam",
         "Ion_rhomjz", "rho"]
CIDX = {n: i for i, n in enumerate(COMPS)}
# predictor-corrector layout (fields/Fields.cpp:128-164; optional "rho" last)
COMPS_PC = ["N_jx", "N_jy", "ExmBy", "EypBx", "Ez", "Bx", "By", "Bz", "Psi", "jx", "jy", "jz", "rhomjz",
            "P_Bx", "P_By", "P_jx", "P_jy", "Ion_rhomjz", "It_Bx", "It_By", "PIt_Bx", "PIt_By", "rho"]
CIDX_PC = {n: i for i, n in enumerate(COMPS_PC)}


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "hps_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def set_threads(n):
    """OpenMP threads of the oracle's hot loops (bench.py's CPU-baseline leg).  1 (the default) = the serial path the golden
    checksums are pinned on; > 1 = 4-colour tiles in the scatter kernels (DepositionUtil.H:204-253), parallel rows
    elsewhere."""
    L = lib()
    L.orc_set_threads.restype = C.c_int
    L.orc_set_threads.argtypes = [C.c_int]
    return L.orc_set_threads(int(n))


class Slab(C.Structure):
    _fields_ = [("p", C.c_void_p), ("nx", C.c_int), ("ny", C.c_int), ("g", C.c_int), ("ncomp", C.c_int)]


class Plasma(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in
                ("x", "y", "w", "ux", "uy", "psi", "x_prev", "y_prev", "ux_half", "uy_half", "psi_half",
                 "valid", "ion_lev")] + [("n", C.c_long)]


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


def fill_struct(st, d):
    """Fill a ctypes Structure from a dict (tuples -> arrays)."""
    for name, typ in st._fields_:
        if name not in d:
            continue
        v = d[name]
        if hasattr(typ, "_length_"):
            setattr(st, name, typ(*v))
        else:
            setattr(st, name, v)
    return st


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_shape_factor.restype = C.c_int
        L.orc_shape_factor.argtypes = [C.c_int, C.c_double, C.c_void_p]
        L.orc_deriv_shape.restype = C.c_int
        L.orc_deriv_shape.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_deposit_current.restype = C.c_long
        L.orc_deposit_current.argtypes = [Slab, Plasma, Geom, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_double]
        L.orc_explicit_deposit.restype = None
        L.orc_explicit_deposit.argtypes = [Slab, Plasma, Geom, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int]
        L.orc_advance_plasma.restype = None
        L.orc_advance_plasma.argtypes = [Slab, Plasma, Geom, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
        L.orc_gather.restype = None
        L.orc_gather.argtypes = [Slab, Geom, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p]
        L.orc_tile_sort.restype = None
        L.orc_tile_sort.argtypes = [Plasma, Geom, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_poisson_create.restype = C.c_void_p
        L.orc_poisson_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_poisson_solve.restype = None
        L.orc_poisson_solve.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_poisson_destroy.restype = None
        L.orc_poisson_destroy.argtypes = [C.c_void_p]
        L.orc_dst1.restype = None
        L.orc_dst1.argtypes = [C.c_int, C.c_void_p, C.c_long]
        L.orc_mg_create.restype = C.c_void_p
        L.orc_mg_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_mg_nlev.restype = C.c_int
        L.orc_mg_nlev.argtypes = [C.c_void_p]
        L.orc_mg_solve1.restype = C.c_int
        L.orc_mg_solve1.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_double, C.c_double, C.c_int, C.c_void_p]
        L.orc_mg_destroy.restype = None
        L.orc_mg_destroy.argtypes = [C.c_void_p]
        L.orc_engine_create.restype = C.c_void_p
        L.orc_engine_create.argtypes = [C.c_void_p]
        for f in ("destroy", "run", "begin_step"):
            getattr(L, "orc_engine_" + f).restype = None
            getattr(L, "orc_engine_" + f).argtypes = [C.c_void_p]
        L.orc_engine_solve_slice.restype = None
        L.orc_engine_solve_slice.argtypes = [C.c_void_p, C.c_int]
        L.orc_engine_ncomp.restype = C.c_int
        L.orc_engine_ncomp.argtypes = [C.c_void_p]
        L.orc_engine_guards.restype = C.c_int
        L.orc_engine_guards.argtypes = [C.c_void_p]
        L.orc_engine_nparticles.restype = C.c_long
        L.orc_engine_nparticles.argtypes = [C.c_void_p]
        L.orc_engine_slab.restype = C.c_void_p
        L.orc_engine_slab.argtypes = [C.c_void_p]
        L.orc_engine_particles.restype = C.c_void_p
        L.orc_engine_particles.argtypes = [C.c_void_p]
        L.orc_engine_valid.restype = C.c_void_p
        L.orc_engine_valid.argtypes = [C.c_void_p]
        L.orc_engine_checksums.restype = None
        L.orc_engine_checksums.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_engine_vcycles.restype = C.c_long
        L.orc_engine_vcycles.argtypes = [C.c_void_p]
        L.orc_engine_times.restype = None
        L.orc_engine_times.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_engine_beam_layout.restype = C.c_long
        L.orc_engine_beam_layout.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_engine_set_external_beam.restype = None
        L.orc_engine_set_external_beam.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_engine_initial_beam.restype = None
        L.orc_engine_initial_beam.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_engine_beam_stats.restype = None
        L.orc_engine_beam_stats.argtypes = [C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------------------------
# numpy-level helpers used by the tests
# --------------------------------------------------------------------------------------------
PL_REAL = ["x", "y", "w", "ux", "uy", "psi", "x_prev", "y_prev", "ux_half", "uy_half", "psi_half"]


def make_geom(nx, ny, lo, hi, dz=0.1, bc=1, normalized=1, consts=(1., 1., 1., 1., 1.)):
    dx = (hi[0] - lo[0]) / nx
    dy = (hi[1] - lo[1]) / ny
    g = Geom()
    g.dx, g.dy, g.dz = dx, dy, dz
    g.xoff = 0.5 * (lo[0] + hi[0] - dx * (nx - 1))
    g.yoff = 0.5 * (lo[1] + hi[1] - dy * (ny - 1))
    g.c, g.ep0, g.mu0, g.q_e, g.m_e = consts
    g.plo = (C.c_double * 2)(lo[0], lo[1])
    g.phi = (C.c_double * 2)(hi[0], hi[1])
    g.bc = bc
    g.normalized = normalized
    return g


def slab_struct(arr, nx, ny, g):
    """arr: float64 C-contiguous (ncomp, ny+2g, nx+2g)."""
    assert arr.dtype == np.float64 and arr.flags.c_contiguous
    assert arr.shape[1:] == (ny + 2 * g, nx + 2 * g)
    s = Slab()
    s.p = arr.ctypes.data
    s.nx, s.ny, s.g, s.ncomp = nx, ny, g, arr.shape[0]
    return s


def plasma_struct(real, valid, ion):
    """real: float64 (11, n) C-contiguous; valid/ion: int32 (n,)."""
    assert real.dtype == np.float64 and real.flags.c_contiguous and real.shape[0] == 11
    p = Plasma()
    n = real.shape[1]
    for k, name in enumerate(PL_REAL):
        setattr(p, name, real[k].ctypes.data)
    p.valid = valid.ctypes.data
    p.ion_lev = ion.ctypes.data
    p.n = n
    return p


def shape_factor(order, xmid):
    s = np.zeros(4)
    cell = lib().orc_shape_factor(order, float(xmid), _ptr(s))
    return cell, s[:order + 1].copy()


def deriv_shape(dtype, order, xmid, ix):
    s = C.c_double()
    ds = C.c_double()
    cell = lib().orc_deriv_shape(dtype, order, float(xmid), ix, C.byref(s), C.byref(ds))
    return cell, s.value, ds.value


def deposit_current(slab, nx, ny, g, real, valid, ion, geom, comp, q, m, order, max_qsa=35.0):
    c = np.asarray(comp, dtype=np.int32)
    return lib().orc_deposit_current(slab_struct(slab, nx, ny, g), plasma_struct(real, valid, ion), geom,
                                     _ptr(c), q, m, order, max_qsa)


def explicit_deposit(slab, nx, ny, g, real, valid, ion, geom, cache, depos, q, m, order, dtype=2):
    c = np.asarray(cache, dtype=np.int32)
    d = np.asarray(depos, dtype=np.int32)
    lib().orc_explicit_deposit(slab_struct(slab, nx, ny, g), plasma_struct(real, valid, ion), geom,
                               _ptr(c), _ptr(d), q, m, order, dtype)


def advance_plasma(slab, nx, ny, g, real, valid, ion, geom, comp, q, m, order, temp_slice=0, n_subcycles=1):
    c = np.asarray(comp, dtype=np.int32)
    lib().orc_advance_plasma(slab_struct(slab, nx, ny, g), plasma_struct(real, valid, ion), geom,
                             _ptr(c), q, m, order, temp_slice, n_subcycles)


def gather(slab, nx, ny, g, geom, comp, order, xp, yp):
    c = np.asarray(comp, dtype=np.int32)
    out = np.zeros(6)
    lib().orc_gather(slab_struct(slab, nx, ny, g), geom, _ptr(c), order, float(xp), float(yp), _ptr(out))
    return out


def tile_sort(real, valid, ion, geom, nx, ny, ts):
    """-> (perm uint32 (n,), offsets int32 (ntiles+2,)) of the stable tile sort."""
    ntiles = ((nx + ts - 1) // ts) * ((ny + ts - 1) // ts)
    perm = np.zeros(real.shape[1], dtype=np.uint32)
    off = np.zeros(ntiles + 2, dtype=np.int32)
    lib().orc_tile_sort(plasma_struct(real, valid, ion), geom, nx, ny, ts, _ptr(perm), _ptr(off))
    return perm, off


def poisson_solve(rhs, dx, dy):
    """rhs: (ny, nx) float64 -> solution of Lap(F) = rhs, F = 0 one cell outside the box."""
    ny, nx = rhs.shape
    h = lib().orc_poisson_create(nx, ny, dx, dy)
    st = np.ascontiguousarray(rhs, dtype=np.float64).copy()
    lib().orc_poisson_solve(h, _ptr(st))
    lib().orc_poisson_destroy(h)
    return st


def dst1(x):
    y = np.ascontiguousarray(x, dtype=np.float64).copy()
    lib().orc_dst1(y.size, _ptr(y), 1)
    return y


def mg_solve1(sol2, rhs2, acf, nx, ny, g, dx, dy, tol_rel=1e-4, tol_abs=2.2250738585072014e-308, maxiter=200):
    """sol2/rhs2: (2, ny+2g, nx+2g); acf: (ny+2g, nx+2g). sol2 updated in place. -> (iters, resnorm)."""
    assert sol2.flags.c_contiguous and rhs2.flags.c_contiguous and acf.flags.c_contiguous
    h = lib().orc_mg_create(nx, ny, dx, dy)
    rn = C.c_double()
    it = lib().orc_mg_solve1(h, _ptr(sol2), _ptr(rhs2), _ptr(acf), nx, ny, g, tol_rel, tol_abs, maxiter, C.byref(rn))
    lib().orc_mg_destroy(h)
    return it, rn.value


def mg_solve2(sol2, rhs2, acf_real, acf_imag, dx, dy, tol_rel=1e-4, tol_abs=0.0, maxiter=200):
    """hpmg system type 2 (the laser envelope solve): sol2 / rhs2 (2, ny, nx) Re and Im planes, acf_real (ny, nx), acf_imag a
    scalar; sol2 (in: initial guess) updated in place. -> (V-cycles, resnorm)."""
    assert sol2.flags.c_contiguous and rhs2.flags.c_contiguous and acf_real.flags.c_contiguous
    ny, nx = acf_real.shape
    L = lib()
    L.orc_mg2_solve2.restype = C.c_int
    L.orc_mg2_solve2.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                 C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double)]
    rn = C.c_double()
    it = L.orc_mg2_solve2(nx, ny, dx, dy, _ptr(sol2), _ptr(rhs2), _ptr(acf_real), acf_imag, tol_rel, tol_abs, maxiter, C.byref(rn))
    return it, rn.value


class FieldDiagnostic:
    """Fields::Copy (fields/Fields.cpp:413-533) on the diagnostic geometry of Diagnostic::ResizeFDiagFAB
    (diagnostics/Diagnostic.cpp:300-390; diag_type xyz, whole box, level 0): every solved slice adds
    rel_z[k] * sum_{iy,ix} sy[iy] sx[ix] slab(i_cell+ix, j_cell+iy, comp) to F(i, j, k, comp), with the order-1 shape
    factors (ShapeFactors.H:56-67) of the diagnostic cell centres in x, y and of the diagnostic planes in z; the slab is
    zero-extended beyond its guard cells (guarded_field_xy, Fields.cpp:331-358).  Feed it the engine's slab after every
    solve_slice: push and ShiftSlices leave the field components untouched."""

    def __init__(self, deck, comps, coarsening=(1, 1, 1)):
        self.d, self.comps, self.c = deck, list(comps), tuple(coarsening)
        nx, ny, nz = deck["nx"], deck["ny"], deck["nz"]
        assert nx % self.c[0] == 0 and ny % self.c[1] == 0 and nz % self.c[2] == 0
        self.n = (nx // self.c[0], ny // self.c[1], nz // self.c[2])
        self.F = np.zeros((len(self.comps), self.n[2], self.n[1], self.n[0]))

    @staticmethod
    def _poff(lo, hi, h, n):          # GetPosOffset (fields/Fields.H:71-77)
        return 0.5 * (lo + hi - h * (n - 1))

    def add_slice(self, islice, slab, g):
        d = self.d
        h = [(d["hi"][q] - d["lo"][q]) / (d["nx"], d["ny"], d["nz"])[q] for q in range(3)]
        hc = [(d["hi"][q] - d["lo"][q]) / self.n[q] for q in range(3)]
        pc = [self._poff(d["lo"][q], d["hi"][q], h[q], (d["nx"], d["ny"], d["nz"])[q]) for q in range(3)]
        pd = [self._poff(d["lo"][q], d["hi"][q], hc[q], self.n[q]) for q in range(3)]

        def round_half_away(v):
            return int(np.floor(abs(v) + 0.5) * np.sign(v))
        k_min = round_half_away(((islice - 1) * h[2] + pc[2] - pd[2]) * (1.0 / hc[2]))
        k_max = round_half_away(((islice + 1) * h[2] + pc[2] - pd[2]) * (1.0 / hc[2]))
        # transverse weights of the diagnostic cell centres
        def weights(q, ncoarse):
            x = np.arange(ncoarse) * hc[q] + pd[q]
            mid = (x - pc[q]) * (1.0 / h[q])
            cell = np.floor(mid).astype(np.int64)
            t = mid - cell
            return cell, np.stack([1.0 - t, t])
        ic, sx = weights(0, self.n[0])
        jc, sy = weights(1, self.n[1])
        ny_p, nx_p = slab.shape[1], slab.shape[2]
        for k in range(max(k_min, 0), min(k_max, self.n[2] - 1) + 1):
            mid = (k * hc[2] + pd[2] - pc[2]) * (1.0 / h[2])
            kc = int(np.floor(mid))
            t = mid - kc
            rel = (1.0 - t) if kc == islice else (t if kc + 1 == islice else 0.0)
            if rel == 0.0:
                continue
            for n, m in enumerate(self.comps):
                A = slab[m]
                v = np.zeros((self.n[1], self.n[0]))
                for iy in range(2):
                    for ix in range(2):
                        jj = jc + iy + g
                        ii = ic + ix + g
                        ok = ((jj >= 0) & (jj < ny_p))[:, None] & ((ii >= 0) & (ii < nx_p))[None, :]
                        val = A[np.clip(jj, 0, ny_p - 1)[:, None], np.clip(ii, 0, nx_p - 1)[None, :]]
                        v = v + (sx[ix][None, :] * sy[iy][:, None]) * np.where(ok, val, 0.0)
                self.F[n, k] += rel * v


def insitu_plasma(real, valid, radius=np.inf, clight=1.0):
    """PlasmaParticleContainer::InSituComputeDiags (particles/plasma/PlasmaParticleContainer.cpp:443-530) of a
    plasma sheet (the 11 real arrays in PlasmaIdx order + validity): the 15 per-slice entries."""
    x, y, wt, ux, uy, psi = real[0], real[1], real[2], real[3] / clight, real[4] / clight, real[5]
    keep = (valid != 0) & (x * x + y * y <= radius * radius)
    x, y, wt, ux, uy, psi = (a[keep] for a in (x, y, wt, ux, uy, psi))
    gamma = (1.0 + ux * ux + uy * uy + psi * psi) / (2.0 * psi)
    uz = gamma - psi
    w = wt * gamma / psi
    raw = np.array([w.sum(), (w * x).sum(), (w * x * x).sum(), (w * y).sum(), (w * y * y).sum(), (w * ux).sum(),
                    (w * ux * ux).sum(), (w * uy).sum(), (w * uy * uy).sum(), (w * uz).sum(), (w * uz * uz).sum(),
                    (w * gamma).sum(), (w * gamma * gamma).sum(), (wt * (gamma - 1.0)).sum(), float(keep.sum())])
    inv = 0.0 if raw[0] <= 0.0 else 1.0 / raw[0]
    raw[1:13] *= inv
    return raw


def insitu_fields(slab, g, deck, clight=1.0):
    """Fields::InSituComputeDiags (fields/Fields.cpp:1288-1347) of one slice from the explicit-solver slab: the ten
    sums over the valid cells, times dx dy dz, in the reference's order."""
    v = lambda name: slab[CIDX[name]][g:-g, g:-g]
    dxdydz = np.prod([(deck["hi"][q] - deck["lo"][q]) / (deck["nx"], deck["ny"], deck["nz"])[q] for q in range(3)])
    ex = v("ExmBy") + v("By") * clight
    ey = v("EypBx") - v("Bx") * clight
    terms = [ex ** 2, ey ** 2, v("Ez") ** 2, v("Bx") ** 2, v("By") ** 2, v("Bz") ** 2, v("ExmBy") ** 2, v("EypBx") ** 2,
             v("jz_beam"), v("Ez") * v("jz_beam")]
    return np.array([t.sum() for t in terms]) * dxdydz


def beam_sort_by_box(z, plo_z, dz, num_boxes):
    """BoxSorter::sortParticlesByBox (particles/sorting/BoxSort.cpp:14-78), serial CPU semantics: box =
    static_cast<int>((z - plo_z) * dzi) (truncation toward zero), out of [0, num_boxes] -> num_boxes; counts,
    offsets = exclusive scan of the counts (both num_boxes + 1 long), perm[offset[box] + k] = k-th particle of the box
    in input order.  -> (counts, offsets, perm) as uint64."""
    z = np.asarray(z, dtype=np.float64)
    dzi = 1.0 / dz
    box = np.trunc((z - plo_z) * dzi)
    box = np.where((box < 0) | (box > num_boxes) | ~np.isfinite(box), num_boxes, box).astype(np.int64)
    counts = np.bincount(box, minlength=num_boxes + 1).astype(np.uint64)
    offsets = np.concatenate(([0], np.cumsum(counts)[:-1])).astype(np.uint64)
    perm = np.argsort(box, kind="stable").astype(np.uint64)
    return counts, offsets, perm


class Engine:
    """The oracle's whole-deck driver (Hipace::Evolve + SolveOneSlice, explicit solver)."""

    def __init__(self, deck):
        self.deck = dict(deck)
        self._dk = fill_struct(Deck(), deck)
        self._h = lib().orc_engine_create(C.byref(self._dk))
        self.ncomp = lib().orc_engine_ncomp(self._h)
        self.g = lib().orc_engine_guards(self._h)

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:     # (_LIB is gone at interpreter shutdown)
            _LIB.orc_engine_destroy(self._h)
            self._h = None

    def run(self):
        lib().orc_engine_run(self._h)

    def begin_step(self):
        lib().orc_engine_begin_step(self._h)

    def solve_slice(self, islice):
        lib().orc_engine_solve_slice(self._h, islice)

    def solve_slice_begin(self, islice):      # the driver's two halves of a slice: the oracle is synchronous
        self.solve_slice(islice)

    def solve_slice_finish(self, islice):
        pass

    def slab(self):
        nx, ny, g = self.deck["nx"], self.deck["ny"], self.g
        n = self.ncomp * (ny + 2 * g) * (nx + 2 * g)
        buf = (C.c_double * n).from_address(lib().orc_engine_slab(self._h))
        return np.frombuffer(buf, dtype=np.float64).reshape(self.ncomp, ny + 2 * g, nx + 2 * g)

    def particles(self):
        L = lib()
        n = L.orc_engine_nparticles(self._h)
        if n == 0:
            return np.zeros((11, 0)), np.zeros(0, dtype=np.int32)
        L.orc_engine_particle_stride.restype = C.c_long
        L.orc_engine_particle_stride.argtypes = [C.c_void_p]
        stride = L.orc_engine_particle_stride(self._h)
        buf = (C.c_double * (11 * stride)).from_address(L.orc_engine_particles(self._h))
        vb = (C.c_int32 * n).from_address(L.orc_engine_valid(self._h))
        return np.frombuffer(buf, dtype=np.float64).reshape(11, stride)[:, :n], np.frombuffer(vb, dtype=np.int32)

    def set_density_profile(self, r=(), fr=(), ct=(), ft=()):
        """n(x, y, ct) = density * f_r(r) * f_t(c t), piecewise linear tables (see hps_engine_set_density_profile)."""
        L = lib()
        L.orc_engine_set_density_profile.restype = None
        L.orc_engine_set_density_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (r, fr, ct, ft)]
        assert len(a[0]) == len(a[1]) and len(a[2]) == len(a[3])
        L.orc_engine_set_density_profile(self._h, len(a[0]), _ptr(a[0]), _ptr(a[1]), len(a[2]), _ptr(a[2]), _ptr(a[3]))

    def ions(self):
        """Species "ion" (ADK ionisation): (real (11, n), valid (n,), ion_lev (n,))."""
        L = lib()
        for f, r in (("orc_engine_nions", C.c_long), ("orc_engine_ions", C.c_void_p), ("orc_engine_ion_valid", C.c_void_p),
                     ("orc_engine_ion_levels", C.c_void_p), ("orc_engine_n_ionized", C.c_long)):
            getattr(L, f).restype = r
            getattr(L, f).argtypes = [C.c_void_p]
        n = L.orc_engine_nions(self._h)
        if n == 0:
            return np.zeros((11, 0)), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32)
        buf = (C.c_double * (11 * n)).from_address(L.orc_engine_ions(self._h))
        vb = (C.c_int32 * n).from_address(L.orc_engine_ion_valid(self._h))
        lb = (C.c_int32 * n).from_address(L.orc_engine_ion_levels(self._h))
        return np.frombuffer(buf, dtype=np.float64).reshape(11, n), np.frombuffer(vb, dtype=np.int32), np.frombuffer(lb, dtype=np.int32)

    def n_ionized(self):
        L = lib()
        L.orc_engine_n_ionized.restype = C.c_long
        L.orc_engine_n_ionized.argtypes = [C.c_void_p]
        return L.orc_engine_n_ionized(self._h)

    def adk_tables(self):
        """(prefactor, exp_prefactor, power) of InitIonizationModule, one entry per ionisation level."""
        z = int(self.deck.get("ion_Z", 0))
        out = [np.zeros(max(z, 1)) for _ in range(3)]
        L = lib()
        L.orc_adk_tables.restype = None
        L.orc_adk_tables.argtypes = [C.c_void_p] * 4
        L.orc_adk_tables(self._h, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]))
        return [o[:z] for o in out]

    # --- beam blocks, same layout and method names as hipace_amd.api.SliceEngine (pipeline tests) ---
    def beam_layout(self):
        off = np.zeros(self.deck["nz"] + 1, dtype=np.int64)
        n = lib().orc_engine_beam_layout(self._h, _ptr(off))
        return n, off

    def set_beam_particles(self, soa, allow_outside=False):
        """A host-initialised beam in place of the deck's, before the first step: soa = (7, n) x y z ux uy uz w; same binning as
        hipace_amd.api.SliceEngine.set_beam_particles.  -> particles outside the box in z (left out)."""
        soa = np.ascontiguousarray(soa, dtype=np.float64)
        assert soa.ndim == 2 and soa.shape[0] == 7
        L = lib()
        L.orc_engine_set_beam_particles.restype = C.c_long
        L.orc_engine_set_beam_particles.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        out = L.orc_engine_set_beam_particles(self._h, soa.shape[1], _ptr(soa))
        assert allow_outside or out == 0, "beam particles outside the box in z"
        return out

    # ---- ring hand-off of a moving beam (same interface as hipace_amd.api.SliceEngine) --------------
    @property
    def moving(self):
        return float(self.deck.get("dt", 0.0)) != 0.0

    def beam_capacity(self):
        """Particles a slice block may hold in the hand-off messages (twice the fullest injected slice)."""
        return 2 * max(self.beam_slice(i).shape[1] for i in range(self.deck["nz"])) if not getattr(self, "_cap", None) else self._cap

    def beam_message_doubles(self):
        return 1 + 7 * self.beam_capacity()

    def set_beam_import(self, on):
        if not getattr(self, "_cap", None):
            self._cap = self.beam_capacity()
        L = lib()
        L.orc_engine_set_beam_import.restype = None
        L.orc_engine_set_beam_import.argtypes = [C.c_void_p, C.c_int]
        L.orc_engine_set_beam_import(self._h, int(on))

    def export_beam_slice(self, islice, msg):
        """msg: float64 tensor/array of 1 + 7*cap: [count | x.. y.. z.. ux.. uy.. uz.. w.. (row stride cap)]."""
        cap = (len(msg) - 1) // 7
        blk = self.beam_slice(islice)
        n = blk.shape[1]
        assert n <= cap, "beam slice exceeds the hand-off capacity"
        m = msg.numpy() if hasattr(msg, "numpy") else msg
        m[0] = float(n)
        for k in range(7):
            m[1 + k * cap:1 + k * cap + n] = blk[k]

    def import_beam_slice(self, islice, msg):
        cap = (len(msg) - 1) // 7
        m = msg.numpy() if hasattr(msg, "numpy") else msg
        n = int(m[0])
        data = np.ascontiguousarray(np.stack([m[1 + k * cap:1 + k * cap + n] for k in range(7)]), dtype=np.float64)
        L = lib()
        L.orc_engine_import_beam_slice.restype = None
        L.orc_engine_import_beam_slice.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_void_p]
        L.orc_engine_import_beam_slice(self._h, islice, n, _ptr(data))

    def beam_slice(self, islice):
        """Moving beam (deck dt != 0): the particles sitting on slice `islice` now, as a (7, count) array x y z ux uy uz w."""
        L = lib()
        L.orc_engine_beam_slice_count.restype = C.c_long
        L.orc_engine_beam_slice_count.argtypes = [C.c_void_p, C.c_int]
        L.orc_engine_beam_slice.restype = None
        L.orc_engine_beam_slice.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        n = L.orc_engine_beam_slice_count(self._h, islice)
        out = np.zeros((7, n), dtype=np.float64)
        if n:
            L.orc_engine_beam_slice(self._h, islice, _ptr(out))
        return out

    def beam_spin(self, islice):
        """<beam>.do_spin_tracking: (3, count) array sx sy sz of the particles sitting on slice `islice` now."""
        L = lib()
        L.orc_engine_beam_spin.restype = None
        L.orc_engine_beam_spin.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        n = self.beam_slice(islice).shape[1]
        out = np.zeros((3, n), dtype=np.float64)
        if n:
            L.orc_engine_beam_spin(self._h, islice, _ptr(out))
        return out

    def laser_vcycles(self):
        L = lib()
        L.orc_engine_laser_vcycles.restype = C.c_long
        L.orc_engine_laser_vcycles.argtypes = [C.c_void_p]
        return L.orc_engine_laser_vcycles(self._h)

    def set_insitu_beam(self, radius=np.inf):
        L = lib()
        L.orc_engine_set_insitu_beam.restype = None
        L.orc_engine_set_insitu_beam.argtypes = [C.c_void_p, C.c_double]
        L.orc_engine_set_insitu_beam(self._h, min(float(radius), 1.0e300))

    def insitu_beam(self):
        """BeamParticleContainer::InSituComputeDiags: (23, nz) array, index = islice, of the step solved last."""
        L = lib()
        L.orc_engine_insitu_beam.restype = None
        L.orc_engine_insitu_beam.argtypes = [C.c_void_p, C.c_void_p]
        out = np.zeros((23, self.deck["nz"]))
        L.orc_engine_insitu_beam(self._h, _ptr(out))
        return out

    def set_beam_storage(self, tensor, injected_beam_support=False):
        """tensor: CPU float64 torch tensor or numpy array of 7*nbeam doubles (kept alive by the caller)."""
        self._beam_keep = tensor
        ptr = tensor.data_ptr() if hasattr(tensor, "data_ptr") else tensor.ctypes.data
        lib().orc_engine_set_external_beam(self._h, C.c_void_p(ptr) if tensor is not None else None)

    def initial_beam_into(self, tensor):
        ptr = tensor.data_ptr() if hasattr(tensor, "data_ptr") else tensor.ctypes.data
        lib().orc_engine_initial_beam(self._h, C.c_void_p(ptr))

    def sync(self):
        pass

    # stream coupling of pipeline.run_local_pipeline: the oracle runs on the calling thread, nothing to order
    def record_event(self, slot):
        return None

    def wait_event(self, event):
        pass

    def copy_async(self, dst, src):
        dst.copy_(src)

    def checksums_xz(self):
        """diag_type = xz: per field the sum of the absolute values on the y = 0 line of every slice of the last step"""
        out = np.zeros(self.ncomp)
        L = lib()
        L.orc_engine_checksums_xz.restype = None
        L.orc_engine_checksums_xz.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_engine_checksums_xz(self._h, _ptr(out))
        names = self.comp_names()
        return {names[i]: out[i] for i in range(self.ncomp)}

    def checksums(self):
        out = np.zeros(self.ncomp)
        lib().orc_engine_checksums(self._h, _ptr(out))
        names = self.comp_names()
        cs = {names[i]: out[i] for i in range(self.ncomp)}
        if "aabs" in cs:
            L = lib()
            L.orc_engine_laser_envelope_sum.restype = C.c_double
            L.orc_engine_laser_envelope_sum.argtypes = [C.c_void_p]
            cs["laserEnvelope"] = L.orc_engine_laser_envelope_sum(self._h)
        return cs

    # ---- ring hand-off of the laser envelope (same interface as hipace_amd.api.SliceEngine) ----------------
    @property
    def has_laser(self):
        return bool(self.deck.get("laser_on", 0))

    def laser_message_doubles(self):
        return 4 * self.deck["nx"] * self.deck["ny"]

    def set_step(self, step):
        L = lib()
        L.orc_engine_set_step.restype = None
        L.orc_engine_set_step.argtypes = [C.c_void_p, C.c_int]
        L.orc_engine_set_step(self._h, int(step))

    def set_laser_import(self, on, step=0):
        L = lib()
        L.orc_engine_set_laser_import.restype = None
        L.orc_engine_set_laser_import.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_engine_set_laser_import(self._h, int(on), int(step))

    def export_laser_slice(self, islice, msg):
        L = lib()
        L.orc_engine_export_laser_slice.restype = None
        L.orc_engine_export_laser_slice.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_engine_export_laser_slice(self._h, islice, C.c_void_p(msg.data_ptr()))

    def import_laser_slice(self, islice, msg):
        L = lib()
        L.orc_engine_import_laser_slice.restype = None
        L.orc_engine_import_laser_slice.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_engine_import_laser_slice(self._h, islice, C.c_void_p(msg.data_ptr()))

    def import_laser_from(self, islice, src):
        L = lib()
        L.orc_engine_import_laser_from.restype = None
        L.orc_engine_import_laser_from.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_engine_import_laser_from(self._h, islice, src._h)

    def laser_envelope(self):
        """a_n of the step that has begun: complex array [nz, ny, nx]."""
        L = lib()
        L.orc_engine_laser_envelope.restype = C.c_void_p
        L.orc_engine_laser_envelope.argtypes = [C.c_void_p]
        ptr = L.orc_engine_laser_envelope(self._h)
        d = self.deck
        n = d["nz"] * d["ny"] * d["nx"]
        buf = (C.c_double * (2 * n)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.complex128).reshape(d["nz"], d["ny"], d["nx"])

    def comp_names(self):
        """Names of the slab components of this engine (rho and aabs are optional and come last)."""
        if self.deck.get("bxby_solver", 0):
            return list(COMPS_PC[:22]) + (["rho"] if self.deck.get("deposit_rho", 0) else [])
        return list(COMPS[:21]) + (["rho"] if self.deck.get("deposit_rho", 0) else []) + \
            (["aabs"] if self.deck.get("laser_on", 0) else [])

    def vcycles(self):
        return lib().orc_engine_vcycles(self._h)

    def pc_stats(self):
        """(predictor-corrector iterations so far, sum over slices of the final relative B error)."""
        L = lib()
        L.orc_engine_pc_iterations.restype = C.c_long
        L.orc_engine_pc_iterations.argtypes = [C.c_void_p]
        L.orc_engine_pc_error_sum.restype = C.c_double
        L.orc_engine_pc_error_sum.argtypes = [C.c_void_p]
        return L.orc_engine_pc_iterations(self._h), L.orc_engine_pc_error_sum(self._h)

    def times(self):
        t = np.zeros(6)
        lib().orc_engine_times(self._h, _ptr(t))
        return dict(zip(["deposit", "explicit_deposit", "push", "poisson", "mg", "other"], t))

    def beam_stats(self):
        t = np.zeros(6)
        lib().orc_engine_beam_stats(self._h, _ptr(t))
        return dict(zip(["n", "w", "x", "y", "z", "uz"], t))
