"""Host-side mirror of the reference's operator interface for the slice hot path, on top of the
C ABI (include/hpslice.h).  Names, argument meaning and error behaviour follow the reference's
C++ seams:

    DepositCurrent(plasma, fields, ...)          particles/deposition/PlasmaDepositCurrent.H:28-32
    ExplicitDeposition(plasma, fields, ...)      particles/deposition/ExplicitDeposition.H:20-22
    AdvancePlasmaParticles(plasma, fields, ...)  particles/pusher/PlasmaParticleAdvance.H:23-26
    FFTPoissonSolver.SolvePoissonEquation(lhs)   fields/fft_poisson_solver/FFTPoissonSolver.H:26-57
    MultiGrid.solve1(sol, rhs, acoef, ...)       mg_solver/HpMultiGrid.H:64-66
    SliceEngine                                  Hipace::Evolve / SolveOneSlice (Hipace.cpp:393-728)

PyTorch is used only to own device memory and streams; all arithmetic happens in libhpslice.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import CIDX, COMPS, ID_VALID, PL_REAL, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _iarr(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


class Geometry:
    """Slice geometry + physical constants + particle boundary (hps_geom)."""

    def __init__(self, nx, ny, lo, hi, dz, bc=1, normalized=True, consts=(1., 1., 1., 1., 1.), ng=None):
        self.nx, self.ny = nx, ny
        g = _lib.Geom()
        g.dx = (hi[0] - lo[0]) / nx
        g.dy = (hi[1] - lo[1]) / ny
        g.dz = dz
        # GetPosOffset (fields/Fields.H:71-77): independent of the number of guard cells
        g.xoff = 0.5 * (lo[0] + hi[0] - g.dx * (nx - 1))
        g.yoff = 0.5 * (lo[1] + hi[1] - g.dy * (ny - 1))
        g.c, g.ep0, g.mu0, g.q_e, g.m_e = consts
        g.plo = (C.c_double * 2)(lo[0], lo[1])
        g.phi = (C.c_double * 2)(hi[0], hi[1])
        g.bc = bc
        g.normalized = int(normalized)
        self.c = g


class Fields:
    """The field slab: ncomp planes of (ny+2ng, nx+2ng) doubles in HBM (fields/Fields.H:465)."""

    def __init__(self, nx, ny, ng, ncomp, device="cuda", data=None):
        self.nx, self.ny, self.ng, self.ncomp = nx, ny, ng, ncomp
        shape = (ncomp, ny + 2 * ng, nx + 2 * ng)
        if data is None:
            self.t = torch.zeros(shape, dtype=torch.float64, device=device)
        else:
            self.t = torch.as_tensor(np.ascontiguousarray(data), dtype=torch.float64).reshape(shape).to(device).contiguous()

    def struct(self):
        s = _lib.Slab()
        s.p = self.t.data_ptr()
        s.nx, s.ny, s.ng, s.ncomp = self.nx, self.ny, self.ng, self.ncomp
        s.jstride = self.nx + 2 * self.ng
        s.nstride = s.jstride * (self.ny + 2 * self.ng)
        return s

    def numpy(self):
        return self.t.cpu().numpy()


class PlasmaSheet:
    """Pure-SoA plasma sheet (particles/plasma/PlasmaParticleContainer.H:21-50)."""

    def __init__(self, real, valid=None, ion_lev=None, device="cuda"):
        real = np.ascontiguousarray(real, dtype=np.float64)
        assert real.shape[0] == 11
        self.n = real.shape[1]
        self.real = torch.as_tensor(real).to(device).contiguous()
        if valid is None:
            valid = np.ones(self.n, dtype=np.int32)
        idcpu = np.where(np.asarray(valid) != 0, np.uint64(ID_VALID | (1 << 24)), np.uint64(1 << 24)).astype(np.uint64)
        self.idcpu = torch.as_tensor(idcpu.view(np.int64)).to(device).contiguous()
        if ion_lev is None:
            ion_lev = np.zeros(self.n, dtype=np.int32)
        self.ion_lev = torch.as_tensor(np.asarray(ion_lev, dtype=np.int32)).to(device).contiguous()

    def struct(self):
        p = _lib.Plasma()
        base = self.real.data_ptr()
        for k, name in enumerate(PL_REAL):
            setattr(p, name, base + 8 * k * self.n)
        p.idcpu = self.idcpu.data_ptr()
        p.ion_lev = self.ion_lev.data_ptr()
        p.n = self.n
        return p

    def numpy(self):
        idc = self.idcpu.cpu().numpy().view(np.uint64)
        return self.real.cpu().numpy(), ((idc >> np.uint64(63)) & np.uint64(1)).astype(np.int32)


class BoxSorter:
    """BoxSorter::sortParticlesByBox (particles/sorting/BoxSort.H:18-48): beam particles -> longitudinal boxes."""

    def sortParticlesByBox(self, z, plo_z, dz, num_boxes):
        """z: float64 device tensor.  Fills boxCounts / boxOffsets (num_boxes + 1 each) and boxPermutations
        (perm[new] = old) as numpy uint64 arrays."""
        z = z.contiguous()
        n = z.numel()
        counts = torch.zeros(num_boxes + 1, dtype=torch.int64, device=z.device)
        offsets = torch.zeros(num_boxes + 1, dtype=torch.int64, device=z.device)
        perm = torch.zeros(max(n, 1), dtype=torch.int64, device=z.device)
        check(_lib.lib().hps_beam_sort_by_box(C.c_void_p(z.data_ptr() if n else 0), n, float(plo_z), float(dz), num_boxes,
                                              C.c_void_p(counts.data_ptr()), C.c_void_p(offsets.data_ptr()),
                                              C.c_void_p(perm.data_ptr()), _stream()))
        torch.cuda.synchronize()
        self.boxCounts = counts.cpu().numpy().astype(np.uint64)
        self.boxOffsets = offsets.cpu().numpy().astype(np.uint64)
        self.boxPermutations = perm[:n].cpu().numpy().astype(np.uint64)
        return self


class Tiling:
    """Tile binning of a plasma sheet (the reference's ReorderParticles hook)."""

    def __init__(self, nx, ny, tile_size, max_particles):
        self._h = C.c_void_p()
        check(_lib.lib().hps_tiling_create(nx, ny, tile_size, max_particles, C.byref(self._h)))
        self.fallback = torch.zeros(1, dtype=torch.int32, device="cuda")

    def reorder(self, plasma, geom):
        """Stable sort of `plasma` by tile; returns the reordered PlasmaSheet (a new SoA buffer)."""
        out = PlasmaSheet.__new__(PlasmaSheet)
        out.n = plasma.n
        out.real = torch.empty_like(plasma.real)
        out.idcpu = torch.empty_like(plasma.idcpu)
        out.ion_lev = torch.empty_like(plasma.ion_lev)
        check(_lib.lib().hps_reorder_particles(self._h, plasma.struct(), out.struct(), geom.c, _stream()))
        return out

    def info(self):
        nt, off, perm = C.c_int(), C.c_void_p(), C.c_void_p()
        check(_lib.lib().hps_tiling_info(self._h, C.byref(nt), C.byref(off), C.byref(perm)))
        return nt.value, off.value, perm.value

    def offsets_and_perm(self, n):
        torch.cuda.synchronize()
        nt, off, perm = self.info()
        o = np.empty(nt + 2, dtype=np.int32)
        p = np.empty(n, dtype=np.uint32)
        check(_lib.lib().hps_memcpy_d2h(o.ctypes.data_as(C.c_void_p), C.c_void_p(off), o.nbytes))
        if n:
            check(_lib.lib().hps_memcpy_d2h(p.ctypes.data_as(C.c_void_p), C.c_void_p(perm), p.nbytes))
        return o, p

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.lib().hps_tiling_destroy(self._h)
            self._h = None


def DepositCurrent(plasma, fields, geom, charge, mass, depos_order, jx=-1, jy=-1, jz=-1, rho=-1, chi=-1,
                   rhomjz=-1, max_qsa_weighting_factor=35.0, can_ionize=False, n_qsa=None, tiling=None):
    comp = _iarr([jx, jy, jz, rho, chi, rhomjz])
    nq = C.c_void_p(n_qsa.data_ptr()) if n_qsa is not None else None
    if tiling is None:
        check(_lib.lib().hps_deposit_current(fields.struct(), plasma.struct(), geom.c, comp, charge, mass,
                                             depos_order, max_qsa_weighting_factor, int(can_ionize), nq, _stream()))
    else:
        check(_lib.lib().hps_deposit_current_tiled(fields.struct(), plasma.struct(), geom.c, comp, charge, mass,
                                                   depos_order, max_qsa_weighting_factor, int(can_ionize), nq,
                                                   tiling._h, C.c_void_p(tiling.fallback.data_ptr()), _stream()))


def ExplicitDeposition(plasma, fields, geom, charge, mass, depos_order, Bz, Ez, ExmBy, EypBx, Sy, Sx,
                       derivative_type=2, can_ionize=False, tiling=None):
    if tiling is None:
        check(_lib.lib().hps_explicit_deposit(fields.struct(), plasma.struct(), geom.c, _iarr([Bz, Ez, ExmBy, EypBx]),
                                              _iarr([Sy, Sx]), charge, mass, depos_order, derivative_type,
                                              int(can_ionize), _stream()))
    else:
        check(_lib.lib().hps_explicit_deposit_tiled(fields.struct(), plasma.struct(), geom.c,
                                                    _iarr([Bz, Ez, ExmBy, EypBx]), _iarr([Sy, Sx]), charge, mass,
                                                    depos_order, derivative_type, int(can_ionize), tiling._h,
                                                    C.c_void_p(tiling.fallback.data_ptr()), _stream()))


def AdvancePlasmaParticles(plasma, fields, geom, charge, mass, depos_order, Psi, Ez, Bx, By, Bz,
                           temp_slice=False, n_subcycles=1, can_ionize=False, tiling=None):
    if tiling is None:
        check(_lib.lib().hps_advance_plasma(fields.struct(), plasma.struct(), geom.c, _iarr([Psi, Ez, Bx, By, Bz]),
                                            charge, mass, depos_order, int(temp_slice), n_subcycles,
                                            int(can_ionize), _stream()))
    else:
        check(_lib.lib().hps_advance_plasma_tiled(fields.struct(), plasma.struct(), geom.c,
                                                  _iarr([Psi, Ez, Bx, By, Bz]), charge, mass, depos_order,
                                                  int(temp_slice), n_subcycles, int(can_ionize), tiling._h,
                                                  C.c_void_p(tiling.fallback.data_ptr()), _stream()))


class FFTPoissonSolver:
    """Lap(F) = S, F = 0 one cell outside the box; S lives in the staging area."""

    def __init__(self, nx, ny, dx, dy, device="cuda"):
        self.nx, self.ny = nx, ny
        self._h = C.c_void_p()
        check(_lib.lib().hps_poisson_create(nx, ny, dx, dy, C.byref(self._h)))
        self.staging = torch.zeros((ny, nx), dtype=torch.float64, device=device)

    def StagingArea(self):
        return self.staging

    def SolvePoissonEquation(self, lhs_fields, comp):
        check(_lib.lib().hps_poisson_solve(self._h, C.c_void_p(self.staging.data_ptr()), lhs_fields.struct(), comp,
                                           _stream()))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.lib().hps_poisson_destroy(self._h)
            self._h = None


class MultiGrid2:
    """hpmg::MultiGrid system type 2 (complex coefficient: the laser envelope solve), solve2 with an array Re and a scalar
    Im coefficient.  sol2 / rhs2: CUDA float64 tensors (2, ny, nx); acoef_real (ny, nx); acoef_imag a 1-element tensor."""

    def __init__(self, nx, ny, dx, dy):
        self._h = C.c_void_p()
        check(_lib.lib().hps_mg2_create(nx, ny, dx, dy, C.byref(self._h)))

    def solve2(self, sol2, rhs2, acoef_real, acoef_imag, tol_rel=1e-4, tol_abs=0.0, nummaxiter=200):
        it = C.c_int()
        rn = C.c_double()
        for t in (sol2, rhs2, acoef_real, acoef_imag):
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float64
        check(_lib.lib().hps_mg2_solve2(self._h, sol2.data_ptr(), rhs2.data_ptr(), acoef_real.data_ptr(), acoef_imag.data_ptr(),
                                        tol_rel, tol_abs, nummaxiter, C.byref(it), C.byref(rn), _stream()))
        return it.value, rn.value

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.lib().hps_mg2_destroy(self._h)
            self._h = None


class MultiGrid:
    """hpmg::MultiGrid system type 1."""

    def __init__(self, nx, ny, dx, dy):
        self._h = C.c_void_p()
        check(_lib.lib().hps_mg_create(nx, ny, dx, dy, C.byref(self._h)))

    def solve1(self, fields, sol_comp, rhs_comp, acoef_comp, tol_rel=1e-4, tol_abs=2.2250738585072014e-308,
               nummaxiter=200):
        it = C.c_int()
        rn = C.c_double()
        check(_lib.lib().hps_mg_solve1(self._h, fields.struct(), sol_comp, rhs_comp, acoef_comp, tol_rel, tol_abs,
                                       nummaxiter, C.byref(it), C.byref(rn), _stream()))
        return it.value, rn.value

    def solve1_fabs(self, sol, rhs, acoef, tol_rel=1e-4, tol_abs=2.2250738585072014e-308, nummaxiter=200):
        """hpmg::MultiGrid::solve1(sol, rhs, acoef, ...) with three separate `Fields` (HpMultiGrid.H:64-66): sol and rhs with
        two components, acoef with one; sol is the initial guess on entry."""
        it = C.c_int()
        rn = C.c_double()
        check(_lib.lib().hps_mg_solve1_fabs(self._h, sol.struct(), rhs.struct(), acoef.struct(), tol_rel, tol_abs, nummaxiter,
                                            C.byref(it), C.byref(rn), _stream()))
        return it.value, rn.value

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.lib().hps_mg_destroy(self._h)
            self._h = None


class SliceEngine:
    """Device-resident slice loop for one deck (see hipace_amd/decks.py)."""

    def __init__(self, deck, device=0, tile_size=None, sort_period=None):
        self.deck = dict(deck)
        self._dk = _lib.fill_struct(_lib.Deck(), deck)
        self._h = C.c_void_p()
        check(_lib.lib().hps_engine_create(C.byref(self._dk), device, C.byref(self._h)))
        if tile_size is not None:
            check(_lib.lib().hps_engine_set_tiling(self._h, tile_size, sort_period or 128))
        nc, ng, npart = C.c_int(), C.c_int(), C.c_long()
        check(_lib.lib().hps_engine_info(self._h, C.byref(nc), C.byref(ng), C.byref(npart)))
        self.ncomp, self.ng, self.nparticles = nc.value, ng.value, npart.value

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.lib().hps_engine_destroy(self._h)
            self._h = None

    def begin_step(self):
        check(_lib.lib().hps_engine_begin_step(self._h))

    def solve_slice(self, islice):
        check(_lib.lib().hps_engine_solve_slice(self._h, islice))

    def solve_slice_begin(self, islice):
        """the slice up to the Bx/By solve's norm read-back, enqueued without waiting (see hps_engine_solve_slice_begin)"""
        check(_lib.lib().hps_engine_solve_slice_begin(self._h, islice))

    def solve_slice_finish(self, islice):
        check(_lib.lib().hps_engine_solve_slice_finish(self._h, islice))

    def slice_ready(self):
        """would solve_slice_finish return without waiting for the device?"""
        return bool(_lib.lib().hps_engine_slice_ready(self._h))

    def run_step(self):
        check(_lib.lib().hps_engine_run_step(self._h))

    def sync(self):
        check(_lib.lib().hps_engine_sync(self._h))

    def set_tiling(self, tile_size=16, sort_period=128):
        check(_lib.lib().hps_engine_set_tiling(self._h, tile_size, sort_period))

    # ---- ring hand-off of a moving beam (hipace.dt != 0) ------------------------------------------
    @property
    def moving(self):
        return float(self.deck.get("dt", 0.0)) != 0.0

    def beam_capacity(self):
        n = C.c_long()
        check(_lib.lib().hps_engine_beam_capacity(self._h, C.byref(n)))
        return n.value

    def set_beam_capacity(self, cap):
        """particles a slice's hand-off message has room for (default: twice the fullest injected slice); before the first step"""
        check(_lib.lib().hps_engine_set_beam_capacity(self._h, int(cap)))

    def beam_message_doubles(self):
        """Length of a hand-off message of the moving beam: 1 + rows*capacity (7 rows, 10 with spin tracking)."""
        r = C.c_int()
        check(_lib.lib().hps_engine_beam_message_rows(self._h, C.byref(r)))
        return 1 + r.value * self.beam_capacity()

    def beam_spin(self):
        """<beam>.do_spin_tracking: (3, nbeam) array sx sy sz in the particle order of beam_state()."""
        nbeam, _ = self.beam_layout()
        out = np.zeros((3, max(nbeam, 1)), dtype=np.float64)
        check(_lib.lib().hps_engine_beam_spin(self._h, out.ctypes.data_as(C.c_void_p) if nbeam else None))
        return out[:, :nbeam]

    def set_beam_import(self, on):
        check(_lib.lib().hps_engine_set_beam_import(self._h, int(on)))

    def export_beam_slice(self, islice, msg):
        """msg: float64 device tensor of 1 + 7*beam_capacity(); filled asynchronously on the engine's stream."""
        check(_lib.lib().hps_engine_export_beam_slice(self._h, islice, C.c_void_p(msg.data_ptr())))

    def import_beam_slice(self, islice, msg):
        check(_lib.lib().hps_engine_import_beam_slice(self._h, islice, C.c_void_p(msg.data_ptr())))

    def beam_state(self):
        """hipace.dt != 0: (boundaries int64 [nz+1], soa float64 [7, nbeam]) of the moving beam; slice p from the head
        is soa[:, boundaries[p]:boundaries[p+1]] (rows x y z ux uy uz w)."""
        nbeam, _ = self.beam_layout()
        nz = self.deck["nz"]
        bnd = np.zeros(nz + 1, dtype=np.int64)
        soa = np.zeros((7, max(nbeam, 1)), dtype=np.float64)
        check(_lib.lib().hps_engine_beam_state(self._h, bnd.ctypes.data_as(C.c_void_p), soa.ctypes.data_as(C.c_void_p) if nbeam else None))
        return bnd, soa[:, :nbeam].reshape(7, nbeam) if nbeam else soa[:, :0]

    def set_density_profile(self, r=(), fr=(), ct=(), ft=()):
        """n(x, y, ct) = density * f_r(r) * f_t(c t): piecewise-linear tables (hps_engine_set_density_profile)."""
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (r, fr, ct, ft)]
        assert len(a[0]) == len(a[1]) and len(a[2]) == len(a[3])
        p = [x.ctypes.data_as(C.c_void_p) if len(x) else None for x in a]
        check(_lib.lib().hps_engine_set_density_profile(self._h, len(a[0]), p[0], p[1], len(a[2]), p[2], p[3]))

    def set_fusion(self, on=True):
        """Push of slice k and deposition of slice k-1 in one pass over the sheet (include/hpslice.h: hps_engine_set_fusion)."""
        check(_lib.lib().hps_engine_set_fusion(self._h, int(on)))

    def fallbacks(self):
        n = C.c_long()
        check(_lib.lib().hps_engine_fallbacks(self._h, C.byref(n)))
        return n.value

    def sorts(self):
        n = C.c_long()
        check(_lib.lib().hps_engine_sorts(self._h, C.byref(n)))
        return n.value

    def set_diagnostics(self, on=True):
        check(_lib.lib().hps_engine_set_diagnostics(self._h, int(on)))

    def set_profiling(self, on=True, stride=1, light=False):
        """HIP-event phase timers; stride > 1 times every stride-th slice only (an event record costs ~3.4 us);
        light: only the deposition kernel and the kernel-free interval (4 records per slice instead of 11)."""
        check(_lib.lib().hps_engine_set_profiling_stride(self._h, int(stride)))
        check(_lib.lib().hps_engine_set_profiling(self._h, (2 if light else 1) if on else 0))

    def beam_layout(self):
        """-> (nbeam, offsets[nz+1]): block p (p-th slice from the head) holds particles offsets[p]:offsets[p+1]."""
        nb = C.c_long()
        off = (C.c_long * (self.deck["nz"] + 1))()
        check(_lib.lib().hps_engine_beam_info(self._h, C.byref(nb), off))
        return nb.value, np.array(off[:], dtype=np.int64)

    def set_beam_particles(self, soa, allow_outside=False):
        """A host-initialised beam in place of the deck's (any of the reference's injection types): soa = (7, n) array
        x y z ux uy uz w.  Before the first begin_step.  -> number of particles outside the box in z (left out)."""
        soa = np.ascontiguousarray(soa, dtype=np.float64)
        assert soa.ndim == 2 and soa.shape[0] == 7
        out = C.c_long(0)
        check(_lib.lib().hps_engine_set_beam_particles(self._h, soa.shape[1], soa.ctypes.data_as(C.c_void_p),
                                                       C.byref(out) if allow_outside else None))
        return out.value

    def set_beam_storage(self, tensor, injected_beam_support=False):
        """Use `tensor` (float64, 7*nbeam, on this device) as the engine's beam blocks; None = own.
        injected_beam_support: the blocks are the injected beam handed along the ring (dt = 0), so the slab
        kernels may keep skipping the beam planes outside its transverse support."""
        check(_lib.lib().hps_engine_set_beam_storage(self._h, C.c_void_p(tensor.data_ptr()) if tensor is not None else None))
        if tensor is not None and injected_beam_support:
            check(_lib.lib().hps_engine_assume_initial_beam_support(self._h))

    def initial_beam_into(self, tensor):
        check(_lib.lib().hps_engine_initial_beam(self._h, C.c_void_p(tensor.data_ptr())))

    # ---- field diagnostics (Fields::Copy) -----------------------------------------------------------------
    def set_field_diagnostic(self, names, coarsening=(1, 1, 1), diag_type="xyz", patch_lo=None, patch_hi=None):
        """diagnostic.field_data = names, diagnostic.coarsening = cx cy cz, diagnostic.diag_type = xyz | xz | yz | xy,
        diagnostic.patch_lo / patch_hi (hps_engine_set_field_diagnostic_box)."""
        idx = _lib.CIDX_PC if self.deck.get("bxby_solver", 0) else _lib.CIDX
        cn = self.comp_names()
        comps = (C.c_int * len(names))(*[idx[n] if n in idx else cn.index(n) for n in names])
        co = (C.c_int * 3)(*coarsening)
        sd = {"xyz": -1, "yz": 0, "xz": 1, "xy": 2}[diag_type]
        plo = (C.c_double * 3)(*patch_lo) if patch_lo is not None else None
        phi = (C.c_double * 3)(*patch_hi) if patch_hi is not None else None
        check(_lib.lib().hps_engine_set_field_diagnostic_box(self._h, len(names), comps, co, sd, plo, phi))
        self._fd = (list(names), tuple(coarsening))

    def field_diagnostic_geometry(self):
        """-> (cells (nx, ny, nz), lo, hi) of the diagnostic grid"""
        n, lo, hi = (C.c_int * 3)(), (C.c_double * 3)(), (C.c_double * 3)()
        check(_lib.lib().hps_engine_field_diagnostic_geometry(self._h, n, lo, hi))
        return tuple(n), tuple(lo), tuple(hi)

    def field_diagnostic(self):
        """-> dict name -> array [nz, ny, nx] of the diagnostic grid, of the step that is being (or has just been) solved."""
        names, _ = self._fd
        n, _, _ = self.field_diagnostic_geometry()
        out = np.empty((len(names), n[2], n[1], n[0]))
        check(_lib.lib().hps_engine_field_diagnostic(self._h, out.ctypes.data_as(C.c_void_p)))
        return {k: out[i] for i, k in enumerate(names)}

    INSITU_FIELDS = ["[Ex^2]", "[Ey^2]", "[Ez^2]", "[Bx^2]", "[By^2]", "[Bz^2]", "[ExmBy^2]", "[EypBx^2]", "[jz_beam]",
                     "[Ez*jz_beam]"]

    INSITU_PLASMA = ["sum(w)", "[x]", "[x^2]", "[y]", "[y^2]", "[ux]", "[ux^2]", "[uy]", "[uy^2]", "[uz]", "[uz^2]", "[ga]",
                     "[ga^2]", "[(ga-1)*(1-vz)]", "Np"]

    INSITU_BEAM = ["sum(w)", "[x]", "[x^2]", "[y]", "[y^2]", "[z]", "[z^2]", "[ux]", "[ux^2]", "[uy]", "[uy^2]", "[uz]", "[uz^2]",
                   "[x*ux]", "[y*uy]", "[z*uz]", "[x*uy]", "[y*ux]", "[ux/uz]", "[uy/uz]", "[ga]", "[ga^2]", "Np"]

    def laser_vcycles(self):
        n = C.c_long()
        check(_lib.lib().hps_engine_laser_vcycles(self._h, C.byref(n)))
        return n.value

    def set_insitu_beam(self, radius=float("inf")):
        """<beam>.insitu_period / insitu_radius: per-slice moments of BeamParticleContainer::InSituComputeDiags."""
        check(_lib.lib().hps_engine_set_insitu_beam(self._h, min(float(radius), 1.0e300)))

    def insitu_beam(self):
        """-> dict name -> array[nz] (index = islice), names as the reference's in-situ file (BeamParticleContainer.cpp:625-650)."""
        out = np.empty((23, self.deck["nz"]))
        check(_lib.lib().hps_engine_insitu_beam(self._h, out.ctypes.data_as(C.c_void_p)))
        return {n: out[i] for i, n in enumerate(self.INSITU_BEAM)}

    def set_insitu_plasma(self, radius=float("inf")):
        """plasma.insitu_period / insitu_radius: per-slice moments of PlasmaParticleContainer::InSituComputeDiags."""
        check(_lib.lib().hps_engine_set_insitu_plasma(self._h, min(float(radius), 1.0e300)))

    def insitu_plasma(self):
        out = np.empty((15, self.deck["nz"]))
        check(_lib.lib().hps_engine_insitu_plasma(self._h, out.ctypes.data_as(C.c_void_p)))
        return {n: out[i] for i, n in enumerate(self.INSITU_PLASMA)}

    def set_insitu_fields(self, on=True):
        """fields.insitu_period: per-slice reductions of Fields::InSituComputeDiags."""
        check(_lib.lib().hps_engine_set_insitu_fields(self._h, int(on)))

    def insitu_fields(self):
        """-> dict name -> array[nz] (index = islice) with the reference's names (Fields.cpp:1380-1389)."""
        out = np.empty((10, self.deck["nz"]))
        check(_lib.lib().hps_engine_insitu_fields(self._h, out.ctypes.data_as(C.c_void_p)))
        return {n: out[i] for i, n in enumerate(self.INSITU_FIELDS)}

    # ---- several steps in flight on one device (pipeline.run_local_pipeline) ----------------------------
    def stream_handle(self):
        """the engine's hipStream_t as an integer (torch.cuda.ExternalStream(handle) for timing events)"""
        return _lib.lib().hps_engine_stream(self._h)

    def record_event(self, slot):
        """Mark this engine's stream; returns the event another engine can wait for."""
        ev = C.c_void_p()
        check(_lib.lib().hps_engine_record_event(self._h, int(slot), C.byref(ev)))
        return ev.value

    def wait_event(self, event):
        """This engine's stream waits (on the device) for an event recorded by another engine."""
        if event is not None:
            check(_lib.lib().hps_engine_wait_event(self._h, C.c_void_p(event)))

    def copy_async(self, dst, src):
        """dst.copy_(src) for two float64 device tensors, on this engine's stream."""
        assert dst.numel() == src.numel()
        check(_lib.lib().hps_engine_copy_async(self._h, C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()),
                                               dst.numel() * 8))

    def phase_times(self):
        ms = (C.c_double * 8)()
        n = C.c_long()
        check(_lib.lib().hps_engine_phase_times(self._h, ms, C.byref(n)))
        names = ["deposit_current", "poisson", "explicit_deposit", "mg_solve1", "advance_plasma", "other", "sort", "empty_interval"]
        return {k: ms[i] for i, k in enumerate(names)}, n.value

    def checksums(self):
        out = (C.c_double * self.ncomp)()
        check(_lib.lib().hps_engine_checksums(self._h, out))
        names = self.comp_names()
        cs = {names[i]: out[i] for i in range(self.ncomp)}
        if "aabs" in cs:
            tot = C.c_double()
            check(_lib.lib().hps_engine_laser_info(self._h, None, C.byref(tot)))
            cs["laserEnvelope"] = tot.value
        return cs

    # ---- ring hand-off of the laser envelope ---------------------------------------------------------------
    @property
    def has_laser(self):
        return bool(self.deck.get("laser_on", 0))

    def laser_message_doubles(self):
        return 4 * self.deck["nx"] * self.deck["ny"]

    def set_step(self, step):
        """physical time step of the next begin_step (pipeline stages run steps r, r + N, ...)"""
        check(_lib.lib().hps_engine_set_step(self._h, int(step)))

    def set_laser_import(self, on, step=0):
        check(_lib.lib().hps_engine_set_laser_import(self._h, int(on), int(step)))

    def export_laser_slice(self, islice, msg):
        check(_lib.lib().hps_engine_export_laser_slice(self._h, islice, C.c_void_p(msg.data_ptr())))

    def import_laser_slice(self, islice, msg):
        check(_lib.lib().hps_engine_import_laser_slice(self._h, islice, C.c_void_p(msg.data_ptr())))

    def import_laser_from(self, islice, src):
        check(_lib.lib().hps_engine_import_laser_from(self._h, islice, src._h))

    def laser_envelope(self):
        """a_n of the step that has begun: complex array [nz, ny, nx]."""
        d = self.deck
        out = np.empty((d["nz"], d["ny"], d["nx"]), dtype=np.complex128)
        check(_lib.lib().hps_engine_laser_envelope(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def comp_names(self):
        """Names of the slab components of this engine (rho and aabs are optional and come last)."""
        if self.deck.get("bxby_solver", 0):
            return list(_lib.COMPS_PC[:22]) + (["rho"] if self.deck.get("deposit_rho", 0) else [])
        return list(COMPS[:21]) + (["rho"] if self.deck.get("deposit_rho", 0) else []) + \
            (["aabs"] if self.deck.get("laser_on", 0) else [])

    def pc_stats(self):
        """(predictor-corrector iterations so far, sum over slices of the final relative B-field error)."""
        its, err = C.c_long(), C.c_double()
        check(_lib.lib().hps_engine_pc_stats(self._h, C.byref(its), C.byref(err)))
        return its.value, err.value

    def pc_zero_b_slices(self):
        """slices on which the predictor-corrector loop left after one pass because sum |B| of its guess was 0 (the serial
        path's exact zeros ahead of the beam, which the engine reproduces: INTEGRATION.md)"""
        n = C.c_long()
        check(_lib.lib().hps_engine_pc_zero_b_slices(self._h, C.byref(n)))
        return n.value

    def stats(self):
        vc, sl = C.c_long(), C.c_long()
        check(_lib.lib().hps_engine_stats(self._h, C.byref(vc), C.byref(sl)))
        return dict(vcycles=vc.value, slices=sl.value)

    def slab(self):
        self.sync()
        s = _lib.lib().hps_engine_slab(self._h)
        nx, ny, g = s.nx, s.ny, s.ng
        out = np.empty((s.ncomp, ny + 2 * g, nx + 2 * g), dtype=np.float64)
        check(_lib.lib().hps_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(s.p), out.nbytes))
        return out

    def ions(self):
        """Species "ion" (deck ion_on): (real (11, n), valid (n,), ion_lev (n,), key (n,)) -- key = the ion's lattice index,
        which the tile sort moves with the particle (the id bits of idcpu)."""
        self.sync()
        p = _lib.lib().hps_engine_ions(self._h)
        n = p.n
        real = np.empty((11, n), dtype=np.float64)
        idc = np.empty(n, dtype=np.uint64)
        lev = np.empty(n, dtype=np.int32)
        if n:
            for k, name in enumerate(PL_REAL):
                check(_lib.lib().hps_memcpy_d2h(real[k].ctypes.data_as(C.c_void_p), C.c_void_p(getattr(p, name)), real[k].nbytes))
            check(_lib.lib().hps_memcpy_d2h(idc.ctypes.data_as(C.c_void_p), C.c_void_p(p.idcpu), idc.nbytes))
            check(_lib.lib().hps_memcpy_d2h(lev.ctypes.data_as(C.c_void_p), C.c_void_p(p.ion_lev), lev.nbytes))
        key = ((idc >> np.uint64(24)) & np.uint64((1 << 39) - 1)).astype(np.int64) - 1
        return real, ((idc >> np.uint64(63)) & np.uint64(1)).astype(np.int32), lev, key

    def ion_stats(self):
        """(electrons released by the species "ion" since the engine was created, particles of the first species now)."""
        a, b = C.c_long(), C.c_long()
        check(_lib.lib().hps_engine_ion_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def particles(self):
        self.sync()
        p = _lib.lib().hps_engine_plasma(self._h)
        n = p.n
        real = np.empty((11, n), dtype=np.float64)
        idc = np.empty(n, dtype=np.uint64)
        if n:
            for k, name in enumerate(PL_REAL):       # x_prev / y_prev may alias x / y inside the engine
                check(_lib.lib().hps_memcpy_d2h(real[k].ctypes.data_as(C.c_void_p), C.c_void_p(getattr(p, name)), real[k].nbytes))
            check(_lib.lib().hps_memcpy_d2h(idc.ctypes.data_as(C.c_void_p), C.c_void_p(p.idcpu), idc.nbytes))
        return real, ((idc >> np.uint64(63)) & np.uint64(1)).astype(np.int32)
