#!/usr/bin/env python
"""Deposition prototypes on a REAL sheath slice: run the headline deck on the engine to slice
--slice (default 715: blown-out sheath, stale tile order), take the engine's tile-sorted sheet and launch records where
they lie on the device, and time the variants of scripts/ubench/depvar.hip on them -- against the shipped kernel through
the C ABI (hps_deposit_current_tiled) on the same data.  Every variant's planes are compared with the shipped kernel's.

    python scripts/deposit_variants.py [--slice 715] [--reps 20] > profiles/r04_deposit_variants.txt
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

from hipace_amd import _lib, api, decks   # noqa: E402

UB = os.path.join(ROOT, "scripts", "ubench")


def build(halo):
    so = os.path.join(UB, f"libdepvar_h{halo}.so")
    src = os.path.join(UB, "depvar.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-w",
                               f"-DDV_HALO={halo}", src, "-o", so])
    lib = C.CDLL(so)
    lib.depvar_run.restype = C.c_int
    lib.depvar_run.argtypes = [C.c_int, _lib.Slab, _lib.Plasma, _lib.Geom, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double,
                               C.c_int, C.POINTER(C.c_float)]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slice", type=int, default=715, help="slices solved (from the head) before the sheet is taken")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--n", type=int, default=1024)
    a = ap.parse_args()
    L = _lib.lib()
    deck = decks.synthetic(a.n, 1024, 2)
    eng = api.SliceEngine(deck, tile_size=16, sort_period=128)
    eng.begin_step()
    for q in range(a.slice):
        eng.solve_slice(deck["nz"] - 1 - q)
    eng.sync()
    L.hps_engine_tiling.restype = C.c_int
    L.hps_engine_tiling.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    til = C.c_void_p()
    _lib.check(L.hps_engine_tiling(eng._h, C.byref(til)))
    ntiles, offs, perm = C.c_int(), C.c_void_p(), C.c_void_p()
    _lib.check(L.hps_tiling_info(til, C.byref(ntiles), C.byref(offs), C.byref(perm)))
    pl = L.hps_engine_plasma(eng._h)
    slab = L.hps_engine_slab(eng._h)
    ntx = (a.n + 15) // 16
    print(f"sheet of slice {a.slice} of the {a.n}^2 x 4 ppc headline deck: {pl.n} particles, {ntiles.value} tiles, "
          f"{eng.sorts()} sorts and {eng.fallbacks()} halo fallbacks so far")
    # occupancy statistics of the sheet (what the lanes of a wave meet)
    real, valid = eng.particles()
    geom = api.Geometry(a.n, a.n, deck["lo"][:2], deck["hi"][:2], (deck["hi"][2] - deck["lo"][2]) / deck["nz"])
    ci = np.floor((real[0] - geom.c.xoff) / geom.c.dx + 0.5).astype(np.int64)
    cj = np.floor((real[1] - geom.c.yoff) / geom.c.dy + 0.5).astype(np.int64)
    live = valid != 0
    occ = np.bincount((cj[live] - cj[live].min()) * (ci.max() - ci.min() + 1) + (ci[live] - ci[live].min()))
    print(f"particles per stencil-base bin: mean {occ[occ > 0].mean():.2f} over occupied bins, max {occ.max()}, "
          f"{(occ == 0).mean() * 100:.1f} % of the bins empty, 99th percentile {np.percentile(occ, 99):.0f}")
    # scratch slab of the same shape: the variants deposit there
    nplane = slab.nstride
    scratch = torch.zeros(slab.ncomp * nplane, dtype=torch.float64, device="cuda")
    sl = _lib.Slab(scratch.data_ptr(), slab.nx, slab.ny, slab.ng, slab.ncomp, slab.jstride, slab.nstride)
    comps = [_lib.CIDX[c] for c in ("jx", "jy", "chi", "rhomjz")]

    def planes():
        torch.cuda.synchronize()
        return scratch.view(slab.ncomp, -1)[comps].clone()

    # the shipped kernel through the C ABI on the same sheet and launch records
    comp6 = (C.c_int * 6)(_lib.CIDX["jx"], _lib.CIDX["jy"], -1, -1, _lib.CIDX["chi"], _lib.CIDX["rhomjz"])
    nfb = torch.zeros(1, dtype=torch.int32, device="cuda")

    def shipped():
        _lib.check(L.hps_deposit_current_tiled(sl, pl, geom.c, comp6, -1.0, 1.0, 2, 35.0, 0, None, til, C.c_void_p(nfb.data_ptr()), None))

    scratch.zero_()
    shipped()
    ref = planes()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    shipped()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.reps):
        shipped()
    e1.record()
    torch.cuda.synchronize()
    t_ship = e0.elapsed_time(e1) / a.reps * 1e3
    print(f"shipped k_deposit_tiled<2,16,51> (C ABI, halo 6):          {t_ship:7.1f} us per launch   [{int(nfb.item()) // (a.reps + 2)} halo fallbacks per launch]")
    nwave_rounds = pl.n / 64.0 / 256.0        # wave-level particle rounds per CU
    names = {0: "mode 0  shipped inner loop in the harness (36 ds_add_f64 per particle)",
             3: "mode 3  the same without its LDS atomics and without flush traffic (loads + arithmetic)",
             5: "mode 5  mode 0 without the flush (loads + arithmetic + LDS atomics)",
             6: "mode 6  mode 3 with every cell of the region flushed (4 x 28 x 28 global atomics per tile)",
             1: "mode 1  neighbouring lanes on the same words merged over DPP",
             2: "mode 2  bins by stencil base, register patch, 36 atomics per BIN (chunks of 1024)",
             4: "mode 4  the same in chunks of 512 particles (3 workgroups per CU)",
             7: "mode 7  768 persistent workgroups, the next tile's particles requested ahead of this tile's atomics",
             8: "mode 8  the same with 1024 workgroups", 9: "mode 9  the same with 512 workgroups"}
    for halo in (6,):
        lib = build(halo)
        print(f"-- harness kernels with a {halo}-cell halo")
        for mode in (0, 3, 5, 6, 1, 2, 4, 7, 8, 9):
            scratch.zero_()
            ms = C.c_float()
            rc = lib.depvar_run(mode, sl, pl, geom.c, offs, ntiles.value, ntx, -1.0, 1.0, 0, C.byref(ms))
            assert rc == 0, (mode, rc)
            got = planes()
            err = ((got - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max().item() if mode not in (3, 5, 6) else float("nan")
            rc = lib.depvar_run(mode, sl, pl, geom.c, offs, ntiles.value, ntx, -1.0, 1.0, a.reps, C.byref(ms))
            assert rc == 0, (mode, rc)
            us = ms.value * 1e3
            clk = us * 1e-6 * 2.4e9 / nwave_rounds
            print(f"{names[mode]:88s} {us:7.1f} us  = {clk:6.0f} clk per wave-level particle round and CU   max rel deviation {err:.1e}")


if __name__ == "__main__":
    main()
