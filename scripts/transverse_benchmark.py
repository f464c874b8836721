#!/usr/bin/env python3
"""The reference's own transverse scaling benchmark (examples/benchmarks/inputs_transverse_benchmark, run by
tests/transverse_benchmark.1Rank.sh with nxy = 1023) on the HIP engine: the xz diagnostic's checksums against the reference's
JSON (the beam is drawn by numpy, not amrex::Random: agreement to the shot noise) and the time of the box.

    python scripts/transverse_benchmark.py [nxy]
"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hipace_amd import _lib, api, decks


def main():
    nxy = int(sys.argv[1]) if len(sys.argv) > 1 else 1023
    deck = decks.transverse_benchmark(nxy, 1000)
    t = time.time()
    soa = decks.fixed_weight_pdf_beam(deck, seed=2024, **decks.TRANSVERSE_BENCHMARK_BEAM(nxy))
    print(f"beam: {soa.shape[1]} particles drawn in {time.time() - t:.1f} s")
    gold = None
    if nxy == 1023:
        gold = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "transverse_benchmark.1Rank.json")))
    names = list(gold["lev=0"].keys()) if gold else ["By", "Ez", "Psi"]
    L = _lib.lib()
    nx, nz, jc = deck["nx"], deck["nz"], deck["ny"] // 2
    for rep in range(2):
        eng = api.SliceEngine(deck, tile_size=int(os.environ.get("TILE", "16")))
        eng.set_beam_particles(soa)
        s = L.hps_engine_slab(eng._h)
        rows = torch.zeros((nz, len(names), nx), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        t = time.time()
        eng.begin_step()
        for isl in range(nz - 1, -1, -1):
            eng.solve_slice(isl)
            if rep == 0:
                for m, k in enumerate(names):
                    src = s.p + 8 * (_lib.CIDX[k] * s.nstride + (jc + s.ng) * s.jstride + s.ng)
                    L.hps_engine_copy_async(eng._h, C.c_void_p(rows[isl, m].data_ptr()), C.c_void_p(src), 8 * nx)
        eng.sync()
        dt = time.time() - t
        print(f"{'with' if rep == 0 else 'without'} the diagnostic's row copies: {nz} slices of {nxy}^2, 1 ppc, in {dt:.3f} s = {nz/dt:.0f} slices/s, "
              f"{eng.stats()['vcycles']/nz:.2f} V-cycles per slice")
        if rep == 0:
            tot = rows.abs().sum(dim=(0, 2)).cpu().numpy()
    if gold:
        print(f"{'field':>8} {'this run':>16} {'reference':>16}  rel. difference")
        for m, k in enumerate(names):
            v = gold["lev=0"][k]
            print(f"{k:>8} {tot[m]:16.6f} {v:16.6f}  {(tot[m] - v)/v if v else 0.0:+.2e}")
        gb = gold["beam"]
        for k, r in (("w", 6), ("x", 0), ("y", 1), ("z", 2), ("uz", 5)):
            mine = np.abs(soa[r]).sum()
            print(f"beam {k:>3} {mine:16.6f} {gb[k]:16.6f}  {(mine - gb[k])/gb[k]:+.2e}")


if __name__ == "__main__":
    main()
