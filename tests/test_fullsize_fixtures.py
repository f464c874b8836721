"""CPU side of the whole-box fixtures (tests/golden/fullsize_*.json, written by scripts/make_fullsize_fixtures.py from the pinned
oracle): they are what the script would write today -- same decks -- and the script itself runs (on a shrunk deck)."""
import importlib.util
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _script():
    spec = importlib.util.spec_from_file_location("make_fullsize_fixtures", os.path.join(ROOT, "scripts", "make_fullsize_fixtures.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", ["config2", "config2_beam_at_head", "config3", "config4", "config5_fft", "config5_mg", "config5_si_fft", "config5_si_mg", "config5_si_mg_full"])
def test_fixture_is_of_the_deck_the_script_builds(name):
    m = _script()
    if not os.path.exists(os.path.join(GOLD, f"fullsize_{name}.json")):
        pytest.fail("fixture is missing (scripts/make_fullsize_fixtures.py writes it)")
    fx = json.load(open(os.path.join(GOLD, f"fullsize_{name}.json")))
    deck = m.jsonable(m.BOXES[name][0]())
    assert fx["deck"] == json.loads(json.dumps(deck)), "the deck of the fixture is not the deck the script builds now"
    nz = deck["nz"]
    assert str(nz - 1) in fx["trace"] and len(fx["trace"]) >= nz // fx["trace_every"]
    assert set(fx["checksums"]) >= {"Bx", "By", "Ez", "Psi", "jx", "jy", "rhomjz"}
    assert fx["final"]["n_valid"] <= fx["final"]["n_particles"] and fx["final"]["n_particles"] > 0
    for q, s in fx["trace"].items():
        assert s["vcycles"] >= 0 and all(np.isfinite(v) for v in s["slab_sum_abs"].values()), q
    if name.startswith("config5"):
        assert fx["final"]["n_ionized"] == fx["final"]["ion_level_sum"] > 1000
    if name.startswith("config2"):
        assert fx["final"]["pc_iterations"] >= nz and fx["final"]["vcycles"] == 0


def test_fixture_script_runs_on_a_small_box(tmp_path, oracle):
    """the script end to end on a shrunk config-4 deck: what it writes equals what the oracle's own whole-deck driver gives"""
    from hipace_amd import decks
    m = _script()
    m.BOXES["tiny"] = (lambda: decks.synthetic(32, 24, 2), 8, "test box")
    m.run_box("tiny", 2, str(tmp_path))
    fx = json.load(open(tmp_path / "fullsize_tiny.json"))
    ref = oracle.Engine(decks.synthetic(32, 24, 2))
    ref.run()
    for k, v in ref.checksums().items():
        assert abs(fx["checksums"][k] - v) <= 1e-12 * max(abs(v), 1e-300), k
    assert fx["final"]["vcycles"] == ref.vcycles() and sorted(map(int, fx["trace"])) == [7, 15, 23]
