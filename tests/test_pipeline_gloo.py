"""Ring pipeline over time steps, world_size 2-4, gloo on CPU.

The product driver (hipace_amd/pipeline.py) is run unchanged; the CPU oracle engine stands in for
the HIP engine (same beam-block interface), so this checks the message schedule: the head rank
injects the beam, every other (rank, step) gets it slice by slice through the ring, no deadlock,
and every step reproduces the single-process checksums (hipace.dt = 0: all steps are identical) --
the reference's own 2-rank test asserts exactly that (tests/blowout_wake.2Rank.sh: same JSON as 1 rank).
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from hipace_amd import decks


def _deck(pc=False):
    d = decks.blowout_wake()
    d.update(nx=32, ny=32, nz=24, lo=(-8.0, -8.0, -1.44), hi=(8.0, 8.0, 1.44), n_steps=1,
             beam_zmin=-1.4, beam_zmax=1.4)
    if pc:      # the same hand-off under the predictor-corrector Bx/By solver (beams deposit into the shared jx jy jz)
        d = decks.predictor_corrector(d, 1.0e-3, 5, 0.1)
    return d


def _moving_deck():
    """A slow beam (v_z = 0.77 c) in a focusing field: particles fall back through the slices every step, so the
    hand-off carries blocks of changing size."""
    d = decks.beam_evolution()
    d.update(nz=12, lo=(-2.0, -2.0, -2.4), hi=(2.0, 2.0, 2.4), beam_zmin=-1.0, beam_zmax=1.6, beam_umean=(0.0, 0.0, 1.2),
             beam_density=1.0e-3, n_steps=1, dt=0.9, beam_n_subcycles=4, ext_E_slope=(0.3, 0.2))
    return d


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _host_beam(deck):
    """a warm Gaussian bunch drawn on the host (decks.fixed_weight_beam) for _moving_deck without its own beam"""
    return decks.fixed_weight_beam(deck, 3000, 1.0e-3, (0.1, lambda z: 0.05 * z, 0.3), (0.3, 0.3, 0.6), u_mean=(0.0, 0.0, 1.2),
                                   u_std=(0.02, 0.02, 0.05), seed=9)


def _worker(rank, world, port, n_steps, out, moving=False, pc=False, batch=1, host_beam=False):
    import torch.distributed as dist
    from hipace_amd.pipeline import run_pipeline
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    deck = _moving_deck() if moving else _deck(pc)
    if host_beam:
        deck = dict(deck, beam_profile=-1)
    eng = O.Engine(deck)
    if host_beam:      # every rank's engine is given the particles (layout, capacity); only the head rank injects them
        eng.set_beam_particles(_host_beam(deck), allow_outside=True)
    sums = {}

    def on_step_end(step):
        sums[step] = eng.checksums()

    solved = run_pipeline(eng, rank, world, n_steps, "cpu", on_step_end, handoff_batch=batch)
    out.put((rank, solved, sums))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_steps,pc,batch", [(2, 3, False, 1), (2, 2, False, 1), (1, 2, False, 1), (2, 2, True, 1), (4, 9, False, 1),
                                                    (2, 3, False, 4), (3, 7, False, 8)])
def test_ring_pipeline_matches_single_process(oracle, world, n_steps, pc, batch):
    ref = oracle.Engine(_deck(pc))
    ref.run()
    want = ref.checksums()
    assert want["jz" if pc else "jz_beam"] > 0
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_steps, out, False, pc, batch)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = set()
    nz = _deck()["nz"]
    for rank, solved, sums in results:
        assert solved == nz * len(range(rank, n_steps, world))
        for step, cs in sums.items():
            assert step % world == rank
            seen.add(step)
            for k, v in want.items():
                assert abs(cs[k] - v) <= 1e-12 * max(abs(v), 1e-300), (rank, step, k, cs[k], v)
    assert seen == set(range(n_steps))


@pytest.mark.parametrize("world,n_steps", [(2, 3), (2, 4)])
def test_ring_pipeline_hands_a_moving_beam_on(oracle, world, n_steps):
    """hipace.dt != 0: every step's field checksums equal those of one process stepping the same deck (bit for bit:
    the blocks are moved, not recomputed), with more steps than ranks (closed ring)."""
    deck = _moving_deck()
    ref = oracle.Engine(deck)
    want = {}
    for s in range(n_steps):
        ref.begin_step()
        for k in range(deck["nz"] - 1, -1, -1):
            ref.solve_slice(k)
        want[s] = ref.checksums()
    assert want[n_steps - 1]["jz_beam"] != want[0]["jz_beam"]          # the beam does evolve
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_steps, out, True)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = set()
    for rank, solved, sums in results:
        assert solved == deck["nz"] * len(range(rank, n_steps, world))
        for step, cs in sums.items():
            seen.add(step)
            for k, v in want[step].items():
                assert cs[k] == v, (rank, step, k, cs[k], v)
    assert seen == set(range(n_steps))


def test_ring_pipeline_hands_a_host_initialised_moving_beam_on(oracle):
    """the same with a beam the host has drawn (hps_engine_set_beam_particles' twin in the oracle: a random Gaussian bunch in
    place of the deck's fixed_ppc one): two ranks, three steps, every step's checksums equal one process stepping the same
    particles"""
    world, n_steps = 2, 3
    deck = dict(_moving_deck(), beam_profile=-1)
    ref = oracle.Engine(deck)
    ref.set_beam_particles(_host_beam(deck), allow_outside=True)
    want = {}
    for s in range(n_steps):
        ref.begin_step()
        for k in range(deck["nz"] - 1, -1, -1):
            ref.solve_slice(k)
        want[s] = ref.checksums()
    assert want[0]["jz_beam"] != 0.0 and want[n_steps - 1]["jz_beam"] != want[0]["jz_beam"]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_steps, out, True, False, 1, True)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = set()
    for rank, solved, sums in results:
        for step, cs in sums.items():
            seen.add(step)
            for k, v in want[step].items():
                assert cs[k] == v, (rank, step, k, cs[k], v)
    assert seen == set(range(n_steps))


@pytest.mark.parametrize("lanes,n_steps", [(1, 2), (2, 3), (3, 4), (3, 7)])
def test_local_pipeline_several_steps_in_flight(oracle, lanes, n_steps):
    """pipeline.run_local_pipeline: the ring with all its stages in one process (one engine per step in flight, one
    host thread each, hand-off by copy) gives every step the checksums of a single run -- also when the ring closes
    (more steps than lanes)."""
    from hipace_amd.pipeline import run_local_pipeline
    ref = oracle.Engine(_deck())
    ref.run()
    want = ref.checksums()
    engs = [oracle.Engine(_deck()) for _ in range(lanes)]
    got = {}

    def on_step_end(step, eng):
        got[step] = eng.checksums()

    solved = run_local_pipeline(engs, n_steps, "cpu", on_step_end)
    assert solved == n_steps * _deck()["nz"]
    assert sorted(got) == list(range(n_steps))
    for step, cs in got.items():
        for k, v in want.items():
            assert abs(cs[k] - v) <= 1e-12 * max(abs(v), 1e-300), (step, k, cs[k], v)


@pytest.mark.parametrize("lanes", [1, 2, 3])
def test_local_pipeline_hands_the_laser_envelope_on(oracle, lanes):
    """An evolving laser pulse in a plasma with several steps in flight: every stage receives a_n and a_{n-1} of its step
    slice by slice from the stage that ran the step before (MultiBuffer.cpp:840-852, 913-925) -- envelope and checksums
    of five steps identical to one engine running them in turn."""
    import numpy as np
    from hipace_amd.pipeline import run_local_pipeline
    d = decks.laser_blowout_wake()
    d.update(nx=32, ny=32, nz=16, lo=(-16.0, -16.0, -4.0), hi=(16.0, 16.0, 4.0), laser_a0=1.5, laser_lambda0=0.4,
             laser_solver=1, dt=5.0)
    ref = oracle.Engine(d)
    want = {}
    for s in range(5):
        ref.begin_step()
        for k in range(d["nz"] - 1, -1, -1):
            ref.solve_slice(k)
        want[s] = (ref.checksums(), ref.laser_envelope().copy())
    assert np.abs(want[4][1] - want[0][1]).max() > 1e-2 * np.abs(want[0][1]).max()
    engs = [oracle.Engine(d) for _ in range(lanes)]
    got = {}

    def on_step_end(step, eng):
        got[step] = (eng.checksums(), eng.laser_envelope().copy())

    run_local_pipeline(engs, 5, "cpu", on_step_end)
    assert sorted(got) == list(range(5))
    for s in range(5):
        assert np.array_equal(got[s][1], want[s][1]), s
        for k, v in want[s][0].items():
            assert abs(got[s][0][k] - v) <= 1e-12 * max(abs(v), 1e-300), (s, k)


def _laser_deck():
    d = decks.laser_blowout_wake()
    d.update(nx=32, ny=32, nz=16, lo=(-16.0, -16.0, -4.0), hi=(16.0, 16.0, 4.0), laser_a0=1.5, laser_lambda0=0.4,
             laser_solver=1, dt=5.0)
    return d


def _laser_worker(rank, world, port, n_steps, out):
    import torch.distributed as dist
    from hipace_amd.pipeline import run_pipeline
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = O.Engine(_laser_deck())
    res = {}

    def on_step_end(step):
        res[step] = (eng.checksums(), eng.laser_envelope().copy())

    run_pipeline(eng, rank, world, n_steps, "cpu", on_step_end, laser_lookahead=3)
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_steps", [(2, 2), (2, 5), (3, 4)])
def test_laser_envelope_travels_between_ranks(oracle, world, n_steps):
    """The laser's time levels on the ring: one message per slice ({a_{n+1}, a_n}, packed and unpacked by the engines)
    behind the slice's beam block; every step has the envelope and the checksums of a single engine running the steps in
    turn -- with a short receive look-ahead when the ring does not close (n_steps <= ranks) and a whole step of it when
    it does."""
    import numpy as np
    d = _laser_deck()
    ref = oracle.Engine(d)
    want = {}
    for s in range(n_steps):
        ref.begin_step()
        for k in range(d["nz"] - 1, -1, -1):
            ref.solve_slice(k)
        want[s] = (ref.checksums(), ref.laser_envelope().copy())
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_laser_worker, args=(r, world, port, n_steps, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = {}
    for _, res in results:
        got.update(res)
    assert sorted(got) == list(range(n_steps))
    for s in range(n_steps):
        assert np.array_equal(got[s][1], want[s][1]), s
        for k, v in want[s][0].items():
            assert abs(got[s][0][k] - v) <= 1e-12 * max(abs(v), 1e-300), (s, k)


def _prefilled_worker(rank, world, port, counts, batch, out):
    import torch.distributed as dist
    from hipace_amd.pipeline import run_pipeline
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = O.Engine(_deck())
    seen = []
    marks = {}

    def on_slice(m, q):
        seen.append((m, q))
        if q == counts[rank] - 4:          # every rank pauses 4 slices before ITS end: rank r is 2r slices behind rank 0
            dist.barrier()
            marks["paused_at"] = q

    solved = run_pipeline(eng, rank, world, world, "cpu", slices_per_step=counts, on_slice=on_slice, handoff_batch=batch)
    out.put((rank, solved, seen, marks, eng.checksums()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,batch", [(2, 1), (3, 1), (2, 3), (3, 2)])
def test_prefilled_pipeline_with_rank_dependent_slice_counts(oracle, world, batch):
    """bench.py's short timed runs: rank r solves lag*r slices fewer than rank 0 (lag = 2, or the hand-off group + 2), so
    that all ranks can pause at a barrier with the pipeline filled (rank r-1 that far ahead of rank r) and then time the
    same number of slices each.  The ranks stop at different slices, the barrier inside the run does not deadlock, and
    what a later rank computed for its slices equals the single-process result for those slices."""
    n0 = 16
    lag = 2 if batch == 1 else batch + 2
    counts = [n0 - lag * r for r in range(world)]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prefilled_worker, args=(r, world, port, counts, batch, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, solved, seen, marks, cs in results:
        assert solved == counts[rank]
        assert seen == [(0, q) for q in range(counts[rank] + 1)]
        assert marks["paused_at"] == counts[rank] - 4
        ref = oracle.Engine(_deck())
        ref.begin_step()
        for q in range(counts[rank]):
            ref.solve_slice(_deck()["nz"] - 1 - q)
        want = ref.checksums()
        for k, v in want.items():
            assert abs(cs[k] - v) <= 1e-12 * max(abs(v), 1e-300), (rank, k, cs[k], v)


def _lanes_worker(rank, world, port, lanes, n_steps, kind, out):
    import torch.distributed as dist
    from hipace_amd.pipeline import run_lanes
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    deck = {"static": _deck, "moving": _moving_deck, "laser": _laser_deck}[kind]()
    engs = [O.Engine(deck) for _ in range(lanes)]
    res = {}

    def on_step_end(step, eng):
        res[step] = (eng.checksums(), eng.laser_envelope().copy() if kind == "laser" else None)

    solved = run_lanes(engs, rank, world, n_steps, "cpu", on_step_end, laser_lookahead=3)
    out.put((rank, solved, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,lanes,n_steps,kind", [(2, 2, 4, "static"), (2, 2, 9, "static"), (2, 2, 6, "moving"), (2, 2, 5, "laser"),
                                                      (3, 2, 7, "static"), (2, 3, 8, "static")])
def test_several_stages_per_rank_on_the_ring(oracle, world, lanes, n_steps, kind):
    """pipeline.run_lanes: `lanes` engines per process are consecutive stages of a ring of world x lanes stages (several
    time steps in flight per device AND several devices: Hipace.cpp:400-401 with several ranks per GPU).  The edges inside
    a process are device copies, the edge that leaves it is the transport; ONE host thread per process drives its engines
    and makes every transport call.  Every step has exactly the checksums (and envelope) of one engine running the steps
    in turn -- open and closed ring, static beam, moving beam, evolving laser pulse."""
    import numpy as np
    deck = {"static": _deck, "moving": _moving_deck, "laser": _laser_deck}[kind]()
    ref = oracle.Engine(deck)
    want = {}
    for s in range(n_steps):
        ref.begin_step()
        for k in range(deck["nz"] - 1, -1, -1):
            ref.solve_slice(k)
        want[s] = (ref.checksums(), ref.laser_envelope().copy() if kind == "laser" else None)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lanes_worker, args=(r, world, port, lanes, n_steps, kind, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = {}
    W = world * lanes
    for rank, solved, res in results:
        mine = [s for s in range(n_steps) if (s % W) // lanes == rank]
        assert sorted(res) == mine, (rank, sorted(res), mine)
        assert solved == deck["nz"] * len(mine)
        got.update(res)
    assert sorted(got) == list(range(n_steps))
    for s in range(n_steps):
        if kind == "laser":
            assert np.array_equal(got[s][1], want[s][1]), s
        for k, v in want[s][0].items():
            if kind == "static":
                assert abs(got[s][0][k] - v) <= 1e-12 * max(abs(v), 1e-300), (s, k)
            else:
                assert got[s][0][k] == v, (s, k, got[s][0][k], v)
