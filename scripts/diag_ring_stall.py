"""Find the one-off stall of the RCCL self ring: slowest host calls (send / solve_slice / record_event) with their slice."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipace_amd import api, decks, pipeline
deck = decks.synthetic(1024, 1024, 2)
dev = torch.device("cuda", 0)
eng = api.SliceEngine(deck, tile_size=16, sort_period=128)
eng.begin_step()
for k in range(64): eng.solve_slice(1023 - k)
eng.sync()
T = pipeline.RcclSelfRing(0)
log = []
cur = {"q": -1}
def timed(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); log.append((time.perf_counter() - t, name, cur["q"])); return r
    setattr(obj, name, g)
for n in ("solve_slice", "record_event", "wait_event", "begin_step"): timed(eng, n)
for n in ("send", "sendrecv_self"): timed(T, n)
def on_slice(m, q): cur["q"] = (m, q)
pipeline.run_pipeline(eng, 0, 1, 2, dev, transport=T, handoff_batch=int(os.environ.get("BATCH", "8")), on_slice=on_slice)
log.sort(reverse=True)
for dt, name, q in log[:12]: print(f"{1e3*dt:9.3f} ms  {name:14s} at {q}")
