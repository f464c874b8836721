#!/usr/bin/env python
"""Only the L-stages-in-flight measurement of bench.py (hipace_amd/pipeline.py::run_lanes on L engines, whole boxes of the
headline deck), for profiling that window alone: prints one JSON line {stages, slices, seconds, slices_per_s}.

    python scripts/inflight_run.py --stages 3 --boxes 1            # plain: the window's wall time
    rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- python scripts/inflight_run.py ...   (counters: own passes)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch   # noqa: E402

from hipace_amd import api, decks   # noqa: E402
from hipace_amd.pipeline import run_lanes   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stages", type=int, default=3)
    ap.add_argument("--boxes", type=int, default=1, help="time steps per stage")
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--warm", type=int, default=16, help="untimed slices per stage before the window")
    ap.add_argument("--ring-self", choices=["ipc", "rccl"], default=None,
                    help="the closing edge (last stage -> first stage) goes through the C-ABI ring with this kind of edge "
                         "(pipeline.RcclSelfRing): what a rank of a multi-rank ring with L stages does")
    ap.add_argument("--pre-pipeline", type=int, default=0, help="first run this many steps of ONE stage through the transport (what bench.py's headline run does before its stages-in-flight run)")
    ap.add_argument("--fresh-transport", action="store_true", help="after --pre-pipeline: a new transport for the stages")
    ap.add_argument("--fresh-engines", action="store_true", help="after --pre-pipeline: new engines for the stages")
    a = ap.parse_args()
    nz = 1024
    deck = decks.synthetic(a.n, nz, 2)
    dev = torch.device("cuda", 0)
    L = a.stages
    engines = [api.SliceEngine(deck, device=0, tile_size=16, sort_period=128) for _ in range(L)]
    T = None
    if a.ring_self:
        from hipace_amd.pipeline import RcclSelfRing
        T = RcclSelfRing(0, edge=a.ring_self)
    if a.pre_pipeline:
        from hipace_amd.pipeline import run_pipeline
        run_pipeline(engines[0], 0, 1, a.pre_pipeline, dev, transport=T)
        if a.fresh_transport and T is not None:
            T.close()
            T = RcclSelfRing(0, edge=a.ring_self)
        if a.fresh_engines:
            engines = [api.SliceEngine(deck, device=0, tile_size=16, sort_period=128) for _ in range(L)]
    if a.warm:
        run_lanes(engines, 0, 1, L, dev, slices_per_step=a.warm, transport=T)
    for e in engines:
        e.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if os.environ.get("HPS_DRIVE_TRACE"):
        from hipace_amd import pipeline as _pl
        del _pl._TRACE[:]
    solved = run_lanes(engines, 0, 1, L * a.boxes, dev, transport=T)
    if os.environ.get("HPS_DRIVE_TRACE"):
        _pl.dump_trace()
    for e in engines:
        e.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    vc = sum(e.stats()["vcycles"] for e in engines) / max(sum(e.stats()["slices"] for e in engines), 1)
    print(json.dumps(dict(stages=L, slices=solved, seconds=dt, slices_per_s=solved / dt, vcycles_per_slice=vc, warm_slices_per_stage=a.warm, closing_edge=a.ring_self or "in-process")))


if __name__ == "__main__":
    main()
