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


def _python_beam_file(n=20000):
    """what tools/write_beam.py writes with openPMD-api (a beam for the from_file tests: positions in units of 1/kp, momenta in
    m_e c, the weights as a constant CHARGE record in units of e n0 / kp^3, no weighting record), as (arrays, attrs)"""
    import math
    n0 = 2.8239587008591567e23
    c, e, m_e, ep0 = 299792458.0, 1.602176634e-19, 9.1093837015e-31, 8.8541878128e-12
    kp_inv = c / e * math.sqrt(ep0 * m_e / n0)
    std = (0.3, 0.3, 1.41)
    single = 3.0 * std[0] * std[1] * std[2] * math.sqrt(2.0 * math.pi) ** 3 / n
    rng = np.random.default_rng(0)
    data = [rng.normal(0.0, s, n) for s in std] + [np.zeros(n), np.zeros(n), np.full(n, 2000.0)]
    p = "/data/0/particles/Electrons"
    arrays, attrs = {}, {p: {"HiPACE++_Plasma_Density": n0}}
    attrs[p + "/position"] = dict(unitDimension=[1.0, 0, 0, 0, 0, 0, 0])
    attrs[p + "/momentum"] = dict(unitDimension=[1.0, 1.0, -1.0, 0, 0, 0, 0])
    attrs[p + "/charge"] = dict(unitDimension=[0, 0, 1.0, 1.0, 0, 0, 0], value=single, shape=[n], unitSI=e * n0 * kp_inv ** 3)
    attrs[p + "/mass"] = dict(unitDimension=[0, 1.0, 0, 0, 0, 0, 0], value=single, shape=[n], unitSI=m_e * n0 * kp_inv ** 3)
    for k, ax in enumerate("xyz"):
        arrays[f"{p}/position/{ax}"] = data[k]
        attrs[f"{p}/position/{ax}"] = dict(unitSI=kp_inv)
        arrays[f"{p}/momentum/{ax}"] = data[3 + k]
        attrs[f"{p}/momentum/{ax}"] = dict(unitSI=m_e * c)
    return (arrays, attrs), data, single, n0


def test_from_file_reads_a_python_written_beam_as_the_reference_does():
    """tests/from_file.normalized.1Rank.sh: a beam written by tools/write_beam.py read into a normalised run on 16 x 16 x 32 cells
    of (-8, 8)^3 -- positions and momenta come back as the numbers the script drew, the weights as the script's fixed-weight
    formula over the cell volume (what examples/beam_in_vacuum/analysis_from_file.py checks between file and output)"""
    from hipace_amd.openpmd_writer import read_beam
    container, data, single, n0 = _python_beam_file()
    deck = dict(decks.beam_in_vacuum(), nx=16, ny=16, nz=32, lo=(-8.0, -8.0, -8.0), hi=(8.0, 8.0, 8.0), beam_profile=-1)
    soa = read_beam(container, deck, species="Electrons", plasma_density=n0)
    for k in range(6):
        assert np.abs(soa[k] - data[k]).max() <= 1e-12 * max(np.abs(data[k]).max(), 1.0), k
    cell = (16.0 / 16) * (16.0 / 16) * (16.0 / 32)
    assert np.abs(soa[6] - single / cell).max() <= 1e-12 * single / cell
    # the density may also come from the file
    assert np.array_equal(read_beam(container, deck), soa)


def test_restart_from_this_writers_file(tmp_path):
    """tests/restart.normalized.1Rank.sh: the beam a first run has written is the from_file beam of a second run on a finer grid --
    the same particles, the weights scaled by the ratio of the cell volumes (the same charge); also in SI units"""
    from hipace_amd import h5lite
    from hipace_amd.openpmd_writer import read_beam, write_iteration
    rng = np.random.default_rng(4)
    n = 5000
    for si in (0, 1):
        scale = 1.0e-5 if si else 1.0
        c = 299792458.0 if si else 1.0
        beam = dict(x=rng.normal(0, 0.3, n) * scale, y=rng.normal(0, 0.3, n) * scale, z=rng.normal(0, 3.0, n) * scale,
                    ux=rng.normal(0, 1, n) * c, uy=rng.normal(0, 1, n) * c, uz=rng.normal(2000, 20, n) * c, w=rng.random(n) + 0.5,
                    charge=-1.602176634e-19 if si else -1.0, mass=9.1093837015e-31 if si else 1.0)
        d1 = dict(decks.beam_in_vacuum(), nx=16, ny=16, nz=32, lo=(-2.0 * scale, -2.0 * scale, -12.0 * scale),
                  hi=(2.0 * scale, 2.0 * scale, 12.0 * scale), si_units=si, beam_charge=beam["charge"], beam_mass=beam["mass"])
        d2 = dict(d1, nx=24, ny=24, nz=48)
        geo = dict(lo=d1["lo"], hi=d1["hi"], cells=(16, 16, 32))
        fn = write_iteration(str(tmp_path / f"run{si}"), 0, 0.0, 0.0, geo, beams={"beam": beam}, normalized=not si)
        src = str(tmp_path / f"run{si}" / "openpmd_000000.h5") if h5lite.available() else None
        if src is None:
            z = np.load(fn)
            import json as _json
            src = ({k: z[k] for k in z.files if k != "__attrs__"}, _json.loads(bytes(z["__attrs__"]).decode()))
        soa = read_beam(src, d2, species="beam")
        ratio = 1.0 if si else (24 * 24 * 48) / (16 * 16 * 32)          # (SI weights count particles: no cell volume in them)
        for k, name in enumerate(("x", "y", "z", "ux", "uy", "uz")):
            assert np.abs(soa[k] - beam[name]).max() <= 1e-12 * np.abs(beam[name]).max(), (si, name)
        assert np.abs(soa[6] - beam["w"] * ratio).max() <= 1e-12 * ratio * 1.5, si


def test_fixed_weight_beam_has_the_moments_the_reference_checks():
    """tests/gaussian_weight.1Rank.sh + examples/gaussian_weight/analysis.py on the host-side restatement of fixed_weight:
    the normalised deck (symmetrised: the transverse means exact, charge to 1e-3, widths to 3 %) and the tilted, chirped beam
    of the test's first run (x and y centroid at z_mean + 1 to 5e-3 of the slope, uz there to 5e-4)."""
    deck = dict(decks.beam_in_vacuum(), nx=64, ny=64, nz=64, lo=(-20.0, -20.0, -20.0), hi=(20.0, 20.0, 20.0), beam_profile=-1)
    b = decks.fixed_weight_beam(deck, 10000, 1.0, (0.0, 1.0, 2.0), (3.0, 4.0, 5.0), u_mean=(0.0, 0.0, 1.0e3), u_std=(3.0, 4.0, 0.0),
                                zmin=-20.0, zmax=20.0, radius=40.0, do_symmetrize=True, seed=4)      # (the z centroid of 2500 draws is within the reference's 5 % for about half of all seeds)
    charge = 1.0 * 3.0 * 4.0 * 5.0 * (2.0 * np.pi) ** 1.5 / (40.0 / 64.0) ** 3
    assert abs(b[6].sum() - charge) / charge < 1.0e-3
    assert abs(b[0].mean()) < 1e-12 and abs(b[1].mean() - 1.0) < 1e-4 and abs(b[3].mean()) < 1e-12 and abs(b[4].mean()) < 1e-12
    assert abs(b[2].mean() - 2.0) / 2.0 < 0.05
    for r, sd in ((0, 3.0), (1, 4.0), (2, 5.0)):
        assert abs(b[r].std() - sd) / sd < 0.03, r
    t = decks.fixed_weight_beam(deck, 1000000, 1.0, (lambda z: (z - 2.0) * 0.1, lambda z: 1.0 + (z - 2.0) * (-0.2), 2.0), (0.1, 0.1, 2.0),
                                u_mean=(0.0, 0.0, 1.0e3), zmin=-20.0, zmax=20.0, radius=40.0, duz_per_uz0_dzeta=0.01, do_symmetrize=True, seed=4)
    at1 = (t[2] > 2.99) & (t[2] < 3.01)
    assert at1.sum() > 1000
    assert abs((t[0][at1] - 0.1).mean() / 0.1) < 5e-3
    assert abs((t[1][at1] + 0.2 - 1.0).mean() / 0.2) < 5e-3
    assert abs(((t[5][at1] - (1000.0 + 1000.0 * 0.01)) / (1000.0 + 1000.0 * 0.01)).mean()) < 5e-4
