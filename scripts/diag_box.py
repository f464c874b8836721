"""Phase times along the box (blocks of 64 slices) for the synthetic 1024^2 x 4 ppc blowout deck."""
import sys; sys.path.insert(0, '.')
import torch
from hipace_amd import api, decks
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nz = 1024
eng = api.SliceEngine(decks.synthetic(n, nz, 2), device=0, tile_size=16, sort_period=32)
eng.begin_step()
for k in range(32): eng.solve_slice(nz - 1 - k)
eng.sync()
eng.begin_step()
blk = 64
fb0 = 0
for b in range(nz // blk):
    eng.set_profiling(True)
    for k in range(b*blk, (b+1)*blk): eng.solve_slice(nz - 1 - k)
    eng.sync()
    ph, ns = eng.phase_times()
    eng.set_profiling(False)
    fb = eng.fallbacks()
    print(b*blk, {k: round(v/ns, 4) for k, v in ph.items()}, 'total', round(sum(ph.values())/ns, 4), 'fallbacks', fb - fb0, 'sorts', eng.sorts())
    fb0 = fb
