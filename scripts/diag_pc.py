"""Per-component error of the predictor-corrector engine against the oracle on the first slices (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipace_amd import decks, api
from hipace_amd._lib import COMPS_PC
from oracle import oracle as O
base = decks.linear_wake_gaussian()
base.update(nz=60, lo=(-10.0, -10.0, -4.0), hi=(10.0, 10.0, 2.0), beam_zmin=-3.9, beam_zmax=2.5)
deck = decks.predictor_corrector(base, 1e-4, 7, 0.0635)
ts = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ge = api.SliceEngine(deck, tile_size=ts, sort_period=5); oe = O.Engine(deck)
ge.begin_step(); oe.begin_step()
for isl in range(59, 52, -1):
    ge.solve_slice(isl); oe.solve_slice(isl)
    gs, os_ = ge.slab(), oe.slab()
    print(isl, ge.pc_stats(), oe.pc_stats())
    for c in range(ge.ncomp):
        d = np.abs(gs[c]-os_[c]).max(); m = np.abs(os_[c]).max()
        if d > 1e-12*max(m,1e-300): print('   ', COMPS_PC[c], d, m, np.abs(gs[c]).max())
