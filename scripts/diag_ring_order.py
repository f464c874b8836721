"""RCCL self ring: does it matter when the communicator is created (before / after the engine's warm-up), or that the
engine's phase timers are on?  usage: diag_ring_order.py <early|late> <prof|noprof>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipace_amd import api, decks, pipeline
early, prof = sys.argv[1] == "early", sys.argv[2] == "prof"
deck = decks.synthetic(1024, 1024, 2)
dev = torch.device("cuda", 0)
eng = api.SliceEngine(deck, tile_size=16, sort_period=128)
T = pipeline.RcclSelfRing(0) if early else None
eng.begin_step()
for k in range(64): eng.solve_slice(1023 - k)
eng.sync()
if T is None: T = pipeline.RcclSelfRing(0)
if prof: eng.set_profiling(True, stride=7, light=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = pipeline.run_pipeline(eng, 0, 1, 2, dev, transport=T, handoff_batch=8)
eng.sync(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{sys.argv[1]:6s} {sys.argv[2]:7s} {n/dt:8.1f} slices/s", flush=True)
