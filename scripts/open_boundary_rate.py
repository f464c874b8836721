"""Slices/s of the headline deck (blowout, 1024^2 x 2 x 2 ppc, explicit solver) with boundary.field = Dirichlet and = Open:
what the two launches per batch of solves and the unfused Poisson sources cost.  python scripts/open_boundary_rate.py [n] [nz]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hipace_amd import decks, api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 512
for bc in ((0, 1) if len(sys.argv) > 3 else (0, 1, 0, 1)):
    deck = dict(decks.synthetic(n, nz, 2), field_bc=bc)
    eng = api.SliceEngine(deck, tile_size=16)
    eng.run_step()
    eng.sync()
    t = time.time()
    eng.run_step()
    eng.sync()
    dt = time.time() - t
    print("boundary.field = %s: %.1f slices/s (%d^2 x %d, V-cycles per slice %.2f)" % (("Dirichlet", "Open")[bc], nz / dt, n, nz, eng.stats()["vcycles"] / (2.0 * nz)), flush=True)
