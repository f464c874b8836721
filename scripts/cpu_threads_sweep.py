"""CPU baseline: slices/s of the oracle's OpenMP leg against the thread count on this host (head 3 slices of the headline
deck).  Writes gpurun_out/cpu_threads.json."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hipace_amd import decks  # noqa: E402
from oracle import oracle as O  # noqa: E402

deck = decks.synthetic(1024, 1024, 2)
out = dict(host_cores=os.cpu_count(), legs=[])
for nt in [1, 8, 16, 32, 64, 128, 256]:
    if nt > (os.cpu_count() or 1):
        break
    O.set_threads(nt)
    e = O.Engine(deck)
    e.begin_step()
    t0 = time.perf_counter()
    for k in range(3):
        e.solve_slice(1023 - k)
    dt = time.perf_counter() - t0
    out["legs"].append(dict(threads=nt, slices_per_s=3 / dt, phases=e.times()))
    print(nt, 3 / dt, flush=True)
    del e
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/cpu_threads.json", "w"), indent=1, default=float)
