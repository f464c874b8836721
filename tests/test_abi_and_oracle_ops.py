"""CPU-only checks: the C-ABI library loads and exports every symbol include/hpslice.h declares
(no compute calls without a GPU); oracle operators against independent references
(scipy DST-I, analytic properties of the B-spline shape factors, residual of the multigrid)."""
import ctypes
import os
import re

import numpy as np
import pytest

from hipace_amd import decks
from tests.util import G2, smooth_slab, thermal_sheet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LO, HI = (-8.0, -8.0), (8.0, 8.0)


def test_library_exports_every_declared_symbol():
    from hipace_amd import _lib
    _lib.build()
    hdr = open(os.path.join(ROOT, "include", "hpslice.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(hps_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 40
    L = ctypes.CDLL(_lib.SO)
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    # and the Python binding table covers them
    assert names <= set(_lib._SIGS) | {"hps_poisson_debug_stamps", "hps_mg_debug_stamps"}, names - set(_lib._SIGS)
    L.hps_version.restype = ctypes.c_char_p
    assert b"hpslice" in L.hps_version()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under hipace_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hipace_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


@pytest.mark.parametrize("order", [0, 1, 2, 3])
def test_shape_factors_partition_of_unity(oracle, order):
    rng = np.random.default_rng(order)
    for x in rng.uniform(-3, 40, 200):
        cell, s = oracle.shape_factor(order, x)
        assert abs(s.sum() - 1.0) < 1e-14 and np.all(s >= -1e-15)
        # first moment reproduces the position for order >= 1
        if order >= 1:
            assert abs((s * (cell + np.arange(order + 1))).sum() - x) < 1e-12
        for dtype in (0, 1, 2):
            n = order + dtype + 1
            vals = [oracle.deriv_shape(dtype, order, x, ix) for ix in range(n)]
            assert abs(sum(v[1] for v in vals) - 1.0) < 1e-13          # shape sums to 1
            assert abs(sum(v[2] for v in vals)) < 1e-13                # derivative weights sum to 0


def test_reference_known_answers_for_shape_and_gather(oracle):
    """Known answers recorded in SURVEY.md section 8(c) from the reference's own headers:
    compute_shape_factor<2>(., 10.3) -> cell 9, weights 0.02/0.66/0.32; doGatherShapeN<2> on
    Psi = x^2/2 + 0.1 y returns ExmBy = -1.7, EypBx = -0.1 at x = 1.7."""
    cell, s = oracle.shape_factor(2, 10.3)
    assert cell == 9 and np.allclose(s, [0.02, 0.66, 0.32], atol=1e-12)
    n, g = 32, G2
    geom = oracle.make_geom(n, n, LO, HI)
    slab = np.zeros((12, n + 2 * g, n + 2 * g))
    jj, ii = np.meshgrid(np.arange(-g, n + g), np.arange(-g, n + g), indexing="ij")
    x = ii * geom.dx + geom.xoff
    y = jj * geom.dy + geom.yoff
    slab[11] = 0.5 * x * x + 0.1 * y
    out = oracle.gather(slab, n, n, g, geom, [11, 7, 8, 9, 10], 2, 1.7, -0.4)
    assert abs(out[0] + 1.7) < 1e-12 and abs(out[1] + 0.1) < 1e-12


@pytest.mark.parametrize("n", [7, 8, 33, 64, 100])
def test_oracle_dst_matches_scipy(oracle, n):
    import scipy.fft
    x = np.random.default_rng(n).standard_normal(n)
    assert np.allclose(oracle.dst1(x), scipy.fft.dst(x, type=1), rtol=1e-12, atol=1e-12)


def test_oracle_poisson_inverts_five_point_laplacian(oracle):
    rng = np.random.default_rng(1)
    ny, nx, dx, dy = 48, 64, 0.3, 0.2
    rhs = rng.standard_normal((ny, nx))
    F = np.pad(oracle.poisson_solve(rhs, dx, dy), 1)
    lap = (F[1:-1, 2:] + F[1:-1, :-2] - 2 * F[1:-1, 1:-1]) / dx ** 2 + (F[2:, 1:-1] + F[:-2, 1:-1] - 2 * F[1:-1, 1:-1]) / dy ** 2
    assert np.abs(lap - rhs).max() < 1e-11


@pytest.mark.parametrize("n", [32, 63])
def test_oracle_multigrid_residual_and_iteration_count(oracle, n):
    """hpmg converges ~50x per V-cycle: 3 V-cycles to 1e-4 from a zero guess (measured
    with the reference's own HpMultiGrid.cpp), and the reported residual is the true residual."""
    rng = np.random.default_rng(n)
    g = G2
    dx = dy = 16.0 / n
    rhs = np.zeros((2, n + 2 * g, n + 2 * g))
    rhs[:, g:-g, g:-g] = rng.standard_normal((2, n, n))
    acf = 0.5 + rng.random((n + 2 * g, n + 2 * g))
    sol = np.zeros_like(rhs)
    it, rn = oracle.mg_solve1(sol, rhs, acf, n, n, g, dx, dy, tol_rel=1e-4)
    assert 2 <= it <= 3
    if n % 2 == 1:      # nodal: plain 5-point stencil with zero walls one node outside
        S = sol[:, g - 1:n + g + 1, g - 1:n + g + 1].copy()
        S[:, 0, :] = S[:, -1, :] = S[:, :, 0] = S[:, :, -1] = 0
        lap = (S[:, 1:-1, 2:] + S[:, 1:-1, :-2] - 2 * S[:, 1:-1, 1:-1]) / dx ** 2 + \
              (S[:, 2:, 1:-1] + S[:, :-2, 1:-1] - 2 * S[:, 1:-1, 1:-1]) / dy ** 2
        res = rhs[:, g:-g, g:-g] + acf[g:-g, g:-g] * sol[:, g:-g, g:-g] - lap
        assert abs(np.abs(res).max() - rn) <= 0.35 * rn       # rn is one smoothing step later
    assert rn <= 1e-4 * np.abs(rhs).max()


def test_oracle_tile_sort_is_the_interleaved_stable_order(oracle):
    """Two stable passes: (tile, cell) -> rank inside the cell run -> (tile, rank, cell); checked against an
    independent numpy construction of the same definition."""
    n, ts = 64, 16
    real, valid, ion = thermal_sheet(n, n, LO, HI, ppc=2, seed=4, jitter=20.0)
    valid[::9] = 0
    geom = oracle.make_geom(n, n, LO, HI)
    perm, off = oracle.tile_sort(real, valid, ion, geom, n, n, ts)
    assert sorted(perm.tolist()) == list(range(real.shape[1]))
    assert off[0] == 0 and off[-1] == real.shape[1] and np.all(np.diff(off) >= 0)
    assert off[-1] - off[-2] == (valid == 0).sum()           # invalid particles last
    dx = (HI[0] - LO[0]) / n
    xoff = 0.5 * (LO[0] + HI[0] - dx * (n - 1))
    ci = np.clip(np.floor((real[0] - xoff) / dx + 0.5).astype(np.int64), 0, n - 1)
    cj = np.clip(np.floor((real[1] - xoff) / dx + 0.5).astype(np.int64), 0, n - 1)
    ntx = n // ts
    tile = (cj // ts) * ntx + ci // ts
    cit = (cj % ts) * ts + ci % ts
    key1 = np.where(valid != 0, tile * ts * ts + cit, ntx * ntx * ts * ts)
    o1 = np.argsort(key1, kind="stable")
    k1s = key1[o1]
    start = np.r_[0, np.flatnonzero(np.diff(k1s)) + 1]
    run_id = np.searchsorted(start, np.arange(k1s.size), side="right") - 1
    rank = np.minimum(np.arange(k1s.size) - start[run_id], 15)
    key2 = ((k1s // (ts * ts)) * 16 + rank) * ts * ts + k1s % (ts * ts)
    o2 = np.argsort(key2, kind="stable")
    assert np.array_equal(perm, o1[o2].astype(np.uint32))
    # what the order is for: inside a tile, the first min(count, cells) particles hit distinct cells
    t0 = int(np.argmax(np.diff(off[:-1])))
    seg = perm[off[t0]:off[t0 + 1]]
    ncells_hit = len(set(cit[seg].tolist()))
    assert len(set(cit[seg[:ncells_hit]].tolist())) == ncells_hit


def test_deck_definitions_match_reference_inputs():
    """Values transcribed from examples/*/inputs_normalized and the test scripts."""
    b = decks.blowout_wake()
    assert (b["nx"], b["ny"], b["nz"]) == (64, 64, 100) and b["n_steps"] == 2 and b["beam_density"] == 3.0
    lw = decks.linear_wake()
    assert (lw["nx"], lw["ny"], lw["nz"]) == (32, 32, 200) and lw["deposit_rho"] == 1 and lw["beam_profile"] == 1
    s = decks.synthetic(1024, 1024, 2)
    assert s["plasma_ppc"] == (2, 2) and s["nx"] == 1024


def test_oracle_beam_box_sort_against_a_serial_loop(oracle):
    """The vectorised restatement of BoxSorter::sortParticlesByBox equals the reference's two serial passes
    (count, exclusive scan, place; particles/sorting/BoxSort.cpp:36-61) written out as a plain loop."""
    import numpy as np
    rng = np.random.default_rng(0)
    for n, nb in [(0, 3), (1, 1), (200, 7), (999, 64)]:
        plo, dz = -3.0, 6.0 / nb
        z = rng.uniform(-3.6, 3.6, n)
        if n > 20:
            z[:10] = plo + dz * rng.integers(0, nb + 1, 10)
            z[10:14] = plo - 0.5 * dz
        c, o, p = oracle.beam_sort_by_box(z, plo, dz, nb)
        dzi = 1.0 / dz
        cnt, box = [0] * (nb + 1), []
        for v in z:
            b = int((v - plo) * dzi)
            if b < 0 or b > nb:
                b = nb
            box.append(b)
            cnt[b] += 1
        off = [0] * (nb + 1)
        for b in range(1, nb + 1):
            off[b] = off[b - 1] + cnt[b - 1]
        perm, fill = [0] * n, [0] * (nb + 1)
        for i, b in enumerate(box):
            perm[off[b] + fill[b]] = i
            fill[b] += 1
        assert np.array_equal(c, np.array(cnt, dtype=np.uint64))
        assert np.array_equal(o, np.array(off, dtype=np.uint64))
        assert np.array_equal(p, np.array(perm, dtype=np.uint64))


def test_oracle_openmp_leg_agrees_with_the_serial_path(oracle):
    """bench.py's CPU baseline times the oracle with 1 thread and with all host cores (4-colour tiles in the scatter
    kernels, DepositionUtil.H:204-253).  The threaded leg must compute the same slices: same V-cycle counts, fields equal
    to summation-order rounding."""
    d = decks.blowout_wake()
    d.update(nx=96, ny=96, nz=10, lo=(-8.0, -8.0, -0.6), hi=(8.0, 8.0, 0.6), plasma_ppc=(2, 2))
    res = []
    try:
        for nt in (1, 4):
            assert oracle.set_threads(nt) == nt
            e = oracle.Engine(d)
            e.run()
            res.append((e.slab().copy(), e.vcycles(), e.checksums()))
    finally:
        oracle.set_threads(1)
    assert res[0][1] == res[1][1] > 0
    for c in range(res[0][0].shape[0]):
        assert np.abs(res[0][0][c] - res[1][0][c]).max() <= 1e-9 * max(np.abs(res[0][0][c]).max(), 1e-300), c


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run with N ranks (one
    per GPU); a launcher that started a different number of ranks is refused.  (--spawn-check: launch path only, gloo.)"""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-check"], capture_output=True,
                         text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d == {"spawn_check": True, "n_gpus": 2, "world_size": 2}
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-check"], capture_output=True,
                         text=True, timeout=120, cwd=ROOT, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "--gpus 2" in bad.stderr


@pytest.mark.parametrize("fault", ["none", "fail_ipc", "hang_rccl", "fail_both"])
def test_bench_runs_one_leg_per_edge_kind_and_survives_a_failed_leg(fault):
    """`bench.py --gpus 2` measures once per kind of ring edge (ipc, rccl), each leg in worker processes of its own
    (bench.py::supervise_legs); a leg whose worker raises, or never comes back, is recorded with its error and does not take
    the other leg's number with it.  --dry-legs: the control flow only (gloo, no GPU)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if fault == "fail_ipc":
        env["BENCH_DRY_FAIL"] = "ipc"
    elif fault == "hang_rccl":
        env["BENCH_DRY_HANG"] = "rccl"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-legs", "--leg-timeout", "15"]
    if fault == "fail_both":
        # both legs fail: one line all the same, value null, exit code 1
        env["BENCH_DRY_FAIL"] = "ipc"
        cmd += ["--edge", "ipc"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (out.stdout[-500:], out.stderr[-2000:])
    d = json.loads(lines[0])
    e = d["ring_edges"]
    if fault == "none":
        assert out.returncode == 0
        assert d["ring_edge"] == d["value_is_of_edge"] == "ipc" and d["value"] == e["ipc"]["value"] > e["rccl"]["value"] > 0
        assert d["rccl_ranks_seen"] == 2 and "error" not in e["ipc"] and "error" not in e["rccl"]
    elif fault == "fail_ipc":
        assert out.returncode == 0
        assert d["ring_edge"] == "rccl" and d["value"] == e["rccl"]["value"] > 0 and d["rccl_ranks_seen"] == 2
        assert e["ipc"]["value"] is None and "exited with code" in e["ipc"]["error"]
    elif fault == "hang_rccl":
        assert out.returncode == 0
        assert d["ring_edge"] == "ipc" and d["value"] == e["ipc"]["value"] > 0
        assert e["rccl"]["value"] is None and "killed" in e["rccl"]["error"]
        assert d["rccl_ranks_seen"] is None and "killed" in d["rccl_ranks_seen_note"]
    else:
        assert out.returncode != 0 and d["value"] is None and "error" in d
        assert e["ipc"]["value"] is None and "skipped" in e["rccl"]


def test_openpmd_json_document(tmp_path):
    """hipace_amd/openpmd_writer.py with json_too: the iteration in the layout of openPMD-api's JSON backend -- groups as nested
    objects, attributes as {"datatype", "value"}, datasets as {"datatype", "data"}, constant components as value + shape --
    holds exactly what the npz container holds (the layout itself could not be opened with openPMD-api here: not installed)."""
    import json
    import numpy as np
    from hipace_amd import openpmd_writer as W
    rng = np.random.default_rng(3)
    fields = {"Ez": rng.standard_normal((4, 3, 5)), "rho": rng.standard_normal((4, 3, 5))}
    beam = dict(x=rng.random(7), y=rng.random(7), z=rng.random(7), ux=rng.random(7), uy=rng.random(7), uz=rng.random(7), w=rng.random(7),
                charge=-1.0, mass=1.0)
    fn = W.write_iteration(str(tmp_path), 12, 0.5, 0.1, dict(lo=(-1.0, -2.0, -3.0), hi=(1.0, 2.0, 3.0)), fields, {"beam": beam}, json_too=True)
    doc = json.load(open(fn.replace(".npz", ".json")))
    assert doc["attributes"]["openPMD"] == {"datatype": "STRING", "value": "1.1.0"}
    assert doc["attributes"]["basePath"]["value"] == "/data/%T/" and doc["platform_byte_widths"]["DOUBLE"] == 8
    it = doc["data"]["12"]
    assert it["attributes"]["time"] == {"datatype": "DOUBLE", "value": 0.5}
    ez = it["fields"]["Ez"]
    assert ez["datatype"] == "DOUBLE" and ez["attributes"]["axisLabels"] == {"datatype": "VEC_STRING", "value": ["z", "y", "x"]}
    assert ez["attributes"]["unitDimension"]["datatype"] == "ARR_DBL_7" and len(ez["attributes"]["gridSpacing"]["value"]) == 3
    sp = it["particles"]["beam"]
    assert sp["charge"]["attributes"]["value"]["value"] == -1.0 and sp["charge"]["attributes"]["shape"] == {"datatype": "VEC_ULONG", "value": [7]}
    assert sp["id"]["datatype"] == "ULONG" and sp["position"]["x"]["datatype"] == "DOUBLE"
    arrays, attrs = W.read_openpmd_json(fn.replace(".npz", ".json"))
    z = np.load(fn)
    for k in z.files:
        if k != "__attrs__":
            assert np.array_equal(arrays[k], z[k]), k
    assert attrs["/data/12/fields/rho"]["gridGlobalOffset"] == [-3.0, -2.0, -1.0]


def test_openpmd_hdf5_file(tmp_path):
    """The reference's container: `openpmd_%06T.h5` written through the HDF5 C library (hipace_amd/h5lite.py: ctypes, the image
    has libhdf5 but neither h5py nor openPMD-api) in the layout of openPMD-api's HDF5 backend -- the hierarchy as groups,
    records as contiguous datasets of doubles / uint64, constant components as groups with `value` and `shape`, strings as
    fixed-length H5T_C_S1, bools as the {FALSE, TRUE} enum.  Read back bit for bit; the structure is also looked at with the
    library's own `h5dump` when the image has it.  (diagnostics/OpenPMDWriter.cpp:55-450)"""
    import shutil
    import subprocess
    import numpy as np
    from hipace_amd import h5lite, openpmd_writer as W
    if not h5lite.available():
        pytest.skip("no HDF5 C library in this image")
    rng = np.random.default_rng(4)
    fields = {"Ez": rng.standard_normal((4, 3, 5)), "rho": rng.standard_normal((4, 3, 5))}
    beam = dict(x=rng.random(7), y=rng.random(7), z=rng.random(7), ux=rng.random(7), uy=rng.random(7), uz=rng.random(7), w=rng.random(7),
                charge=-1.0, mass=1.0)
    fn = W.write_iteration(str(tmp_path), 12, 0.5, 0.1, dict(lo=(-1.0, -2.0, -3.0), hi=(1.0, 2.0, 3.0)), fields, {"beam": beam}, hdf5=True)
    h5 = fn.replace(".npz", ".h5")
    assert open(h5, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    arrays, attrs = W.read_hdf5(h5)
    z = np.load(fn)
    for k in z.files:
        if k != "__attrs__":
            assert arrays[k].dtype == z[k].dtype and np.array_equal(arrays[k], z[k]), k
    assert attrs["/"]["openPMD"] == "1.1.0" and attrs["/"]["basePath"] == "/data/%T/" and attrs["/"]["iterationEncoding"] == "fileBased"
    assert attrs["/data/12"] == {"time": 0.5, "dt": 0.1, "timeUnitSI": 1.0}
    ez = attrs["/data/12/fields/Ez"]
    assert ez["axisLabels"] == ["z", "y", "x"] and ez["geometry"] == "cartesian" and ez["gridGlobalOffset"] == [-3.0, -2.0, -1.0]
    assert ez["unitDimension"] == [0.0] * 7 and ez["unitSI"] == 1.0 and len(ez["gridSpacing"]) == 3
    ch = attrs["/data/12/particles/beam/charge"]
    assert ch["value"] == -1.0 and ch["shape"] == [7] and ch["macroWeighted"] == 0
    assert attrs["/data/12/particles/beam"]["normalized_units"] is True
    assert attrs["/data/12/particles/beam/positionOffset/x"]["value"] == 0.0
    # the same through the shim of the reference's checksum backend
    from tests import openpmd_shim as S
    ts = S.OpenPMDTimeSeries(str(tmp_path))
    assert ts.container == "h5" and ts.avail_fields == ["Ez", "rho"] and ts.avail_species == ["beam"]
    assert S.checksums(str(tmp_path))["lev=0"]["Ez"] == float(np.sum(np.abs(fields["Ez"])))
    hc = S.h5py_checksums(str(tmp_path))          # h5py 3.3 of the image's conda interpreter, openPMD-viewer's access pattern
    if hc is not None:
        assert hc["lev=0"]["rho"] == float(np.sum(np.abs(fields["rho"]))) and hc["meta"]["rho"]["gridGlobalOffset"] == [-3.0, -2.0, -1.0]
        assert abs(hc["beam"]["ux"] - float(np.sum(np.abs(beam["ux"])))) <= 1e-14 and hc["beam"]["id"] == 28 and hc["beam"]["charge"] == 7.0
    tool = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)
    if tool:
        out = subprocess.run([tool, "-H", h5], capture_output=True, text=True).stdout
        for want in ('GROUP "fields"', 'DATASET "Ez"', "H5T_IEEE_F64LE", "SIMPLE { ( 4, 3, 5 ) / ( 4, 3, 5 ) }", 'DATASET "id"', "H5T_STD_U64LE",
                     'GROUP "charge"', 'ATTRIBUTE "openPMD"', "H5T_STD_U32LE", "H5T_ENUM"):
            assert want in out, want
