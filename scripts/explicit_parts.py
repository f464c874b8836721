#!/usr/bin/env python
"""What the explicit deposition's time is made of: the shipped kernel (k_explicit_tiled<2,2,16>) timed on the engine's own
sheet and slab at a slice of the headline deck, through whichever library HPS_LIB names -- diagnostic builds with parts of the
kernel compiled out (scripts/build_variant.sh <name> particles_tiled.hip -DHPS_DIAG_EXPL_NO_ATOMICS / _NO_READS / _NO_FLUSH;
their results are wrong, only their time is of interest).   python scripts/explicit_parts.py [--slice 715]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hipace_amd import _lib, api, decks

ap = argparse.ArgumentParser()
ap.add_argument("--slice", type=int, default=715)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
L = _lib.lib()
deck = decks.synthetic(1024, 1024, 2)
eng = api.SliceEngine(deck, tile_size=16, sort_period=128)
eng.begin_step()
for q in range(a.slice):
    eng.solve_slice(deck["nz"] - 1 - q)
eng.sync()
til = C.c_void_p()
_lib.check(L.hps_engine_tiling(eng._h, C.byref(til)))
pl = L.hps_engine_plasma(eng._h)
slab = L.hps_engine_slab(eng._h)
geom = api.Geometry(1024, 1024, deck["lo"][:2], deck["hi"][:2], (deck["hi"][2] - deck["lo"][2]) / deck["nz"])
src = torch.as_tensor(eng.slab()).to("cuda").contiguous().view(-1)       # a copy of the engine's slab: the fields the kernel reads, Sx / Sy to add to
sl = _lib.Slab(src.data_ptr(), slab.nx, slab.ny, slab.ng, slab.ncomp, slab.jstride, slab.nstride)
I = _lib.CIDX
cache = (C.c_int * 4)(I["Bz"], I["Ez"], I["ExmBy"], I["EypBx"]); depos = (C.c_int * 2)(I["Sy"], I["Sx"])
nfb = torch.zeros(1, dtype=torch.int32, device="cuda")
def run():
    _lib.check(L.hps_explicit_deposit_tiled(sl, pl, geom.c, cache, depos, -1.0, 1.0, 2, 2, 0, til, C.c_void_p(nfb.data_ptr()), None))
run(); run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    run()
e1.record(); torch.cuda.synchronize()
print(f"{os.environ.get('HPS_LIB', 'shipped library'):70s} explicit deposition {e0.elapsed_time(e1) / a.reps * 1e3:7.1f} us per launch (slice {a.slice}, {pl.n} particles)")
