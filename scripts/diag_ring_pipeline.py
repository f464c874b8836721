"""Where does a slice's host time go in run_pipeline with the RCCL self ring?  (monkey-patched timers)"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipace_amd import api, decks, pipeline
acc = collections.Counter(); cnt = collections.Counter()
def timed(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label or name] += time.perf_counter() - t; cnt[label or name] += 1; return r
    setattr(obj, name, g)
deck = decks.synthetic(1024, 1024, 2)
eng = api.SliceEngine(deck, tile_size=16, sort_period=128)
eng.begin_step()
for k in range(64): eng.solve_slice(1023 - k)
eng.sync()
PLAIN = "--plain" in sys.argv
T = None if PLAIN else pipeline.RcclSelfRing(0)
for n in ("solve_slice", "record_event", "wait_event", "begin_step", "set_beam_storage"): timed(eng, n)
if T is not None:
    for n in ("send", "recv", "engine_wait", "sendrecv_self"): timed(T, n, "T." + n)
t0 = time.perf_counter()
if PLAIN:
    solved = 0
    for s in range(2):
        eng.begin_step()
        for k in range(300): eng.solve_slice(1023 - k); solved += 1
else:
    solved = pipeline.run_pipeline(eng, 0, 1, 2, torch.device("cuda", 0), slices_per_step=300, transport=T)
eng.sync()
dt = time.perf_counter() - t0
print("slices", solved, "wall ms/slice", 1e3*dt/solved)
for k, v in acc.most_common(): print(f"{k:20s} {1e3*v/solved:8.3f} ms/slice  calls {cnt[k]}")
