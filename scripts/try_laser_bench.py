"""Laser-driven wake at the headline size: 1024 x 1024, 4 ppc, Gaussian pulse, FFT envelope solver (config-5-like, no ionisation)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipace_amd import api, decks
n, nz = 1024, 256
deck = decks.synthetic(n, nz, 2)
deck.update(beam_profile=-1, lo=(-20.0, -20.0, -7.5), hi=(20.0, 20.0, 6.0), laser_on=1, laser_a0=4.5, laser_w0=4.0, laser_L0=2.0,
            laser_lambda0=0.08, laser_solver=int(sys.argv[1]) if len(sys.argv) > 1 else 1, dt=5.0)
for ts in (16, 0):
    eng = api.SliceEngine(deck, tile_size=ts, sort_period=128)
    eng.run_step()
    eng.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_step()
    eng.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"tile {ts}: {nz/dt:.1f} slices/s ({1e3*dt/nz:.3f} ms per slice)")
    del eng
