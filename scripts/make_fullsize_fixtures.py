#!/usr/bin/env python
"""Whole-box parity fixtures at the BASELINE sizes: the PINNED CPU oracle (oracle/hps_oracle.cpp, checked against the
reference's golden checksums by tests/test_oracle_golden.py) sweeps every slice of the full-size decks, as the reference
does (Hipace.cpp:478-480), and writes what the reference's checksum test reduces a run to
(tests/checksum/checksum.py:82-160, tests/checksum/backend/openpmd_backend.py:40-62: the sum of |F| over the whole box
for every field, the beam block) plus the integer state of the run: V-cycle total, predictor-corrector iterations,
particles still valid at the end (QSA drops and absorbed particles are the difference to the initial count), ionised
count and ion-level sum -- and a trace of the same numbers on every `trace_every`-th slice, so that a deviation can be
located along the box.

    python scripts/make_fullsize_fixtures.py [--threads 8] [--only config4,config3,...] [--out tests/golden]

Run in the build container (no GPU needed; 8 OpenMP threads: config 4 about 25 min, config 3 about 6, config 2 about 2,
config 5 about 25).  The -m gpu tests (tests/test_fullsize_boxes.py) run the same decks -- the deck is stored in the
fixture -- on the HIP engine with the default schedule and compare.

Test infrastructure: this script drives the oracle only; nothing here is on the product path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from hipace_amd import decks  # noqa: E402


def config5_deck(n, nz, solver, ionize=True):
    """bench.py --config5's deck (BASELINE configs[4] in normalised units) on n x n x nz cells over the same box."""
    return decks.config5(n, nz, solver, si=False, ionize=ionize)


def config5_si_deck(n, nz, solver, ionize=True):
    """BASELINE configs[4] as BASELINE names it: the .SI deck (bench.py --config5 --si) on n x n x nz cells over the same box."""
    return decks.config5(n, nz, solver, si=True, ionize=ionize)


def config2_deck():
    d = decks.predictor_corrector(decks.linear_wake(), 4.0e-2, 30, 0.05)
    d.update(nx=256, ny=256, nz=512, plasma_ppc=(2, 2))
    return d


def config2_beam_at_head_deck():
    """config 2 with its flat-top driver reaching the head of the box (beam_zmax = hi_z): no slice ahead of the beam.  On the
    BASELINE deck the driver starts 54 slices into the box; ahead of it the serial CPU path has EXACT zeros (electron and
    ion charge cancel term by term), ComputeRelBFieldError returns 0 for sum|B| = 0 (fields/Fields.cpp:1283) and the loop
    leaves after one pass -- also on the first slice that holds beam, whose previous-iteration B is that exact zero --
    while any atomic scatter leaves 1e-16 residue, sum|B| is tiny but positive, and the loop runs to max_iterations on
    rounding noise (as the reference's own GPU build must).  With a tolerance of 4e-2 and a mixing factor of 0.05 the
    loop's result depends on its path at the per-cent level, so the two runs of the BASELINE deck agree to ~1e-2 only,
    by construction.  With the beam at the head both paths are the same and the box can be held to 1e-6."""
    d = config2_deck()
    d.update(beam_zmax=d["hi"][2])
    return d


BOXES = {
    # name: (deck, trace_every, what)
    "config2": (config2_deck, 32, "BASELINE configs[1]: linear_wake.normalized 256x256x512, 4 ppc, predictor-corrector Bx/By"),
    "config2_beam_at_head": (config2_beam_at_head_deck, 32, "config 2 with the driver reaching the head of the box (see config2_beam_at_head_deck): "
                             "the predictor-corrector loop takes the same path on CPU and GPU"),
    "config3": (lambda: decks.synthetic(512, 1024, 2), 64, "BASELINE configs[2]: blowout_wake 512x512x1024, 4 ppc, explicit solver"),
    "config4": (lambda: decks.synthetic(1024, 1024, 2), 64, "BASELINE configs[3] and the bench's headline deck: blowout_wake 1024x1024x1024, 4 ppc, explicit solver"),
    "config5_fft": (lambda: config5_deck(1024, 512, 1), 32,
                    "BASELINE configs[4] at its transverse size, 512 of its 2048 slices over the same box (the oracle keeps three "
                    "envelope time levels of the whole box in host memory: 26 GB at 512 slices, 103 GB at 2048): laser + N dopant, fft envelope solver"),
    "config5_mg": (lambda: config5_deck(512, 256, 2), 32,
                   "the config-5 deck on 512x512x256 cells with the multigrid envelope solver (the reference's default)"),
    "config5_si_mg": (lambda: config5_si_deck(1024, 512, 2), 32,
                      "BASELINE configs[4] as named -- the .SI deck (hipace.normalized_units = 0) -- at its transverse size, 512 of its "
                      "2048 slices over the same box, with the multigrid envelope solver (lasers.solver_type default): laser + N dopant"),
    "config5_si_mg_full": (lambda: config5_si_deck(1024, 2048, 2), 128,
                           "BASELINE configs[4] as named AND at its own length: laser_blowout_wake_explicit.SI 1024x1024x2048, 4 ppc, laser + N dopant, "
                           "multigrid envelope solver; the oracle holds the envelope's time levels in a rolling window (one step: a_n is "
                           "the Gaussian formed per slice, a_{n+1} is kept for the two slices that read it)"),
    "config5_si_fft": (lambda: config5_si_deck(512, 256, 1), 32,
                       "the .SI config-5 deck on 512x512x256 cells with the fft envelope solver"),
}


def jsonable(d):
    return {k: (list(v) if isinstance(v, tuple) else v) for k, v in d.items()}


def snapshot(eng, have_ions):
    """numbers of the current state that do not depend on the order of the particles or of a sum's terms beyond rounding"""
    slab = eng.slab()
    names = eng.comp_names()
    real, valid = eng.particles()
    live = valid != 0
    out = dict(slab_sum_abs={names[c]: float(np.abs(slab[c]).sum()) for c in range(len(names))},
               vcycles=int(eng.vcycles()), n_valid=int(live.sum()), n_particles=int(valid.size),
               sum_w=float(real[2][live].sum()), sum_abs_x=float(np.abs(real[0][live]).sum()))
    if eng.deck.get("bxby_solver", 0):
        out["pc_iterations"] = int(eng.pc_stats()[0])
    if eng.deck.get("laser_solver", 0) == 2:
        out["laser_vcycles"] = int(eng.laser_vcycles())
    if have_ions:
        _, iv, lev = eng.ions()
        out.update(n_ionized=int(eng.n_ionized()), ion_level_sum=int(lev[iv != 0].sum()))
    return out


def run_box(name, threads, out_dir):
    from oracle import oracle as O
    make, every, what = BOXES[name]
    deck = make()
    O.set_threads(threads)
    eng = O.Engine(deck)
    nz = deck["nz"]
    have_ions = bool(deck.get("ion_on", 0))
    eng.begin_step()
    trace = {}
    t0 = time.perf_counter()
    for q in range(nz):                     # q-th slice from the head (Hipace.cpp:478-480: islice = nz-1 ... 0)
        eng.solve_slice(nz - 1 - q)
        if (q + 1) % every == 0 or q == nz - 1:
            trace[str(q)] = snapshot(eng, have_ions)
            print(f"{name}: slice {q + 1}/{nz}  {time.perf_counter() - t0:7.1f} s  vcycles {eng.vcycles()}", flush=True)
    final = snapshot(eng, have_ions)
    fx = dict(what=what, deck=jsonable(deck), generated_by="scripts/make_fullsize_fixtures.py (CPU oracle, %d OpenMP threads)" % threads,
              oracle_seconds=time.perf_counter() - t0,
              checksums={k: float(v) for k, v in eng.checksums().items()},
              beam=eng.beam_stats() if deck.get("beam_profile", 0) >= 0 else None,
              final=final, trace_every=every, trace=trace)
    if fx["beam"] is not None:
        fx["beam"] = {k: float(v) for k, v in fx["beam"].items()}
    path = os.path.join(out_dir, f"fullsize_{name}.json")
    with open(path, "w") as f:
        json.dump(fx, f, indent=1, sort_keys=True)
    print(f"{name}: wrote {path} after {fx['oracle_seconds']:.0f} s", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=min(os.cpu_count() or 1, 16))
    ap.add_argument("--only", default="config2,config2_beam_at_head,config3,config5_mg,config4,config5_fft")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    for name in a.only.split(","):
        run_box(name, a.threads, a.out)


if __name__ == "__main__":
    main()
