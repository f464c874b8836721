"""Does the particle order inside a tile matter?  Engine state at slice `stop` -> standalone tiled
deposit / explicit / push with (a) engine order, (b) cell-sorted inside the tiles, (c) shuffled."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from hipace_amd import api, decks
n = 1024; nz = 1024
deck = decks.synthetic(n, nz, 2)
eng = api.SliceEngine(deck, device=0, tile_size=16, sort_period=32)
eng.begin_step()
stop = int(sys.argv[1]) if len(sys.argv) > 1 else 700
for k in range(stop): eng.solve_slice(nz - 1 - k)
eng.sync()
real, valid = eng.particles()
slab = eng.slab()
lo, hi = deck['lo'], deck['hi']
geom = api.Geometry(n, n, lo, hi, (hi[2]-lo[2])/nz)
ncomp = slab.shape[0]
print('slab', tuple(slab.shape))
f = api.Fields(n, n, (slab.shape[1] - n)//2, ncomp, data=slab)
dx = (hi[0]-lo[0])/n
ix = np.clip(((real[0]-lo[0])/dx).astype(np.int64), 0, n-1); iy = np.clip(((real[1]-lo[1])/dx).astype(np.int64), 0, n-1)
orders = {'engine': np.arange(real.shape[1]), 'cell-sorted': np.argsort(iy*n+ix, kind='stable'), 'shuffled': np.random.default_rng(0).permutation(real.shape[1])}
from hipace_amd._lib import lib
import hipace_amd._lib as L
C_ = {k: getattr(api, k) for k in dir(api) if k.startswith('C_')} if False else None
comps = dict(jx=15, jy=16, chi=2, rhomjz=17)
for name, o in orders.items():
    pl = api.PlasmaSheet(real[:, o], valid[o])
    til = api.Tiling(n, n, 16, pl.n)
    pls = til.reorder(pl, geom)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(5): api.DepositCurrent(pls, f, geom, -1.0, 1.0, 2, tiling=til, **comps)
        torch.cuda.synchronize(); td = (time.perf_counter()-t0)/5
        t0 = time.perf_counter()
        for _ in range(5): api.ExplicitDeposition(pls, f, geom, -1.0, 1.0, 2, 10, 7, 5, 6, 3, 4, tiling=til)
        torch.cuda.synchronize(); te = (time.perf_counter()-t0)/5
    print(name, 'deposit %.1f us, explicit %.1f us' % (td*1e6, te*1e6), 'fallbacks', int(til.fallback.item()))
