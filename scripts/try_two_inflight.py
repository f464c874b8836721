"""Experiment: two time steps in flight on one GPU (two engines, two host threads, two streams)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipace_amd import api, decks
nz = 1024
deck = decks.synthetic(1024, nz, 2)
ne = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
engs = [api.SliceEngine(deck, device=0, tile_size=16, sort_period=128) for _ in range(ne)]
def run(e, count):
    e.begin_step()
    for k in range(count):
        e.solve_slice(nz - 1 - k)
    e.sync()
for e in engs: run(e, 64)
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(e, K)) for e in engs]
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{ne} in flight: {ne*K/dt:.1f} slices/s, {1e3*dt/(ne*K):.4f} ms per slice")
