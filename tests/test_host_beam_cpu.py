"""Host-side beam initialisation (decks.fixed_weight_pdf_beam) and the oracle's entry for host-initialised beams -- CPU only."""
import json
import os

import numpy as np

from hipace_amd import decks

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_fixed_weight_pdf_beam_has_the_reference_decks_weight_and_moments():
    """the deterministic entries of tests/checksum/benchmarks_json/transverse_benchmark.1Rank.json: the beam's total weight
    (density * integral / max_density over the cell volume, BeamParticleContainerInit.cpp:497-542) to rounding, whatever the
    particle count; the first moments of |x|, |y|, |z| within the draw's noise"""
    gold = json.load(open(os.path.join(GOLD, "transverse_benchmark.1Rank.json")))["beam"]
    deck = decks.transverse_benchmark(1023, 1000)
    kw = decks.TRANSVERSE_BENCHMARK_BEAM(1023)
    assert kw["num_particles"] == gold["charge"]
    kw["num_particles"] = 400000
    soa = decks.fixed_weight_pdf_beam(deck, seed=1, **kw)
    n = soa.shape[1]
    assert abs(soa[6].sum() - gold["w"]) <= 1e-12 * gold["w"]
    assert np.all(soa[6] == soa[6][0])
    for k, row in (("x", 0), ("y", 1), ("z", 2)):
        assert abs(np.abs(soa[row]).mean() - gold[k] / gold["charge"]) <= 1e-2 * gold[k] / gold["charge"], k
    assert np.all(soa[5] == 2000.0) and not soa[3].any() and not soa[4].any()
    assert soa[2].min() >= deck["lo"][2] and soa[2].max() < deck["hi"][2]
    # the longitudinal profile: a Gaussian of sigma 1.41 about 0 (cut by the box at -12 and 6: 4.3 sigma)
    assert abs(soa[2].mean()) < 0.02 and abs(soa[2].std() - 1.41) < 0.02


def test_oracle_takes_its_own_deck_beam_through_the_host_entry(oracle):
    deck = decks.blowout_wake()
    deck.update(nx=32, ny=32, nz=40, n_steps=1)
    a = oracle.Engine(deck)
    n, off = a.beam_layout()
    assert n > 100
    blk = np.zeros(7 * n)
    a.initial_beam_into(blk)
    soa = np.empty((7, n))
    for p in range(deck["nz"]):
        first, cnt = off[p], off[p + 1] - off[p]
        soa[:, first:first + cnt] = blk[7 * first:7 * (first + cnt)].reshape(7, cnt)
    parts = [soa[:, off[p]:off[p + 1]] for p in range(deck["nz"])]
    b = oracle.Engine(dict(deck, beam_profile=-1))
    assert b.beam_layout()[0] == 0
    assert b.set_beam_particles(np.concatenate(parts[::-1], axis=1)) == 0
    nb, offb = b.beam_layout()
    assert nb == n and np.array_equal(offb, off)
    a.begin_step()
    b.begin_step()
    for isl in range(deck["nz"] - 1, deck["nz"] - 13, -1):
        a.solve_slice(isl)
        b.solve_slice(isl)
    assert np.array_equal(a.slab(), b.slab())
    # particles outside the box in z are counted and left out
    soa2 = soa.copy()
    soa2[2, :7] = deck["hi"][2] + 0.5
    c = oracle.Engine(dict(deck, beam_profile=-1))
    assert c.set_beam_particles(soa2, allow_outside=True) == 7
    assert c.beam_layout()[0] == n - 7
