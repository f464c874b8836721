"""Particle-count distribution over 16x16 tiles and over cells in the blowout region."""
import sys; sys.path.insert(0, '.')
import torch
from hipace_amd import api, decks
n = 1024; nz = 1024
deck = decks.synthetic(n, nz, 2)
eng = api.SliceEngine(deck, device=0, tile_size=16, sort_period=32)
eng.begin_step()
stop = int(sys.argv[1]) if len(sys.argv) > 1 else 700
for k in range(stop): eng.solve_slice(nz - 1 - k)
eng.sync()
real, valid = eng.particles()
x = torch.from_numpy(real[0][valid == 1]); y = torch.from_numpy(real[1][valid == 1])
print('valid', x.numel(), 'of', real.shape[1])
print('x range', float(x.min()), float(x.max()))
xmin, xmax = float(x.min()), float(x.max())
dx = (xmax - xmin) / n * (1 + 1e-9)
ix = ((x - xmin) / dx).long().clamp(0, n - 1); iy = ((y - xmin) / dx).long().clamp(0, n - 1)
cell = iy * n + ix
cc = torch.bincount(cell, minlength=n*n)
print('cell counts: mean %.2f max %d; cells > 16: %d, > 64: %d' % (cc.float().mean(), cc.max(), (cc > 16).sum(), (cc > 64).sum()))
tile = (iy // 16) * (n // 16) + ix // 16
tc = torch.bincount(tile, minlength=(n//16)**2)
q = torch.tensor([0.5, 0.9, 0.99, 0.999])
print('tile counts: mean %.0f max %d; quantiles' % (tc.float().mean(), tc.max()), torch.quantile(tc.float(), q.to(tc.device)).tolist())
print('tiles > 2048: %d, > 4096: %d, > 8192: %d; empty: %d' % ((tc > 2048).sum(), (tc > 4096).sum(), (tc > 8192).sum(), (tc == 0).sum()))
