"""Host timeline of the RCCL self ring against the plain loop: wall time per block of 64 slices (the host is held once per
slice by the multigrid's norm read-back, so its clock follows the device's)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipace_amd import api, decks, pipeline
deck = decks.synthetic(1024, 1024, 2)
dev = torch.device("cuda", 0)
res = {}
NS = int(os.environ.get("NSTEPS", "4"))
G = int(os.environ.get("GRAN", "32"))
for mode in ("plain", "ring"):
    eng = api.SliceEngine(deck, tile_size=16, sort_period=128)
    eng.begin_step()
    for k in range(64): eng.solve_slice(1023 - k)
    eng.sync()
    stamps = []
    def on_slice(m, q):
        if q % G == 0: stamps.append((m, q, time.perf_counter()))
    if mode == "plain":
        for m in range(NS):
            eng.begin_step()
            for q in range(1024):
                on_slice(m, q); eng.solve_slice(1023 - q)
            on_slice(m, 1024)
    else:
        T = pipeline.RcclSelfRing(0)
        pipeline.run_pipeline(eng, 0, 1, NS, dev, transport=T, handoff_batch=int(os.environ.get("BATCH", "8")), on_slice=on_slice)
    eng.sync()
    stamps.append((NS, 0, time.perf_counter()))
    res[mode] = stamps
    del eng
for (a, b) in zip(res["plain"], res["ring"]): pass
for i in range(len(res["plain"]) - 1):
    p0, p1 = res["plain"][i], res["plain"][i + 1]
    r0, r1 = res["ring"][i], res["ring"][i + 1]
    print(f"step {p0[0]} slices {p0[1]:4d}..: plain {1e3*(p1[2]-p0[2])/G:.4f}  ring {1e3*(r1[2]-r0[2])/G:.4f} ms/slice  diff {1e3*((r1[2]-r0[2])-(p1[2]-p0[2]))/G*1e3:+.1f} us")
