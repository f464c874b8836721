"""Cost of a slice along the box of the headline workload (1024^2 x 1024, 4 ppc): wall time and V-cycles per window of
16 slices, and which window of K slices costs what the whole box costs on average -- bench.py's --start-slice default.
Writes gpurun_out/slice_cost_profile.json."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hipace_amd import api, decks  # noqa: E402

nz, win = 1024, 16
eng = api.SliceEngine(decks.synthetic(1024, nz, 2), tile_size=16, sort_period=128)
for rep in range(2):          # first box warms up
    eng.begin_step()
    eng.sync()
    rows = []
    v0 = eng.stats()["vcycles"]
    t_box = time.perf_counter()
    for w in range(nz // win):
        t0 = time.perf_counter()
        for q in range(w * win, (w + 1) * win):
            eng.solve_slice(nz - 1 - q)
        eng.sync()
        v1 = eng.stats()["vcycles"]
        rows.append(dict(first=w * win, ms_per_slice=1e3 * (time.perf_counter() - t0) / win, vcycles_per_slice=(v1 - v0) / win))
        v0 = v1
    t_box = time.perf_counter() - t_box
mean = sum(r["ms_per_slice"] for r in rows) / len(rows)
best = {}
for K in (20, 32, 64, 96):
    # windows of 16 that bracket K slices; choose the start whose mean cost is closest to the box mean, away from the head
    nw = (K + win - 1) // win
    cand = []
    for w in range(8, len(rows) - nw):
        m = sum(r["ms_per_slice"] for r in rows[w:w + nw]) / nw
        cand.append((abs(m - mean), w * win, m))
    cand.sort()
    best[K] = dict(start=cand[0][1], ms_per_slice=cand[0][2])
out = dict(box_ms_per_slice=mean, box_wall_ms_per_slice=1e3 * t_box / nz, windows=rows, closest_window=best)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/slice_cost_profile.json", "w"), indent=1)
print(json.dumps(dict(box_ms_per_slice=mean, closest_window=best)))
