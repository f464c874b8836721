"""The between-PROCESS hand-off on a device: two or three processes, each driving its own HIP engine(s), connected by the
C-ABI ring's ipc edge (include/hpslice.h hps_ring_*, HPS_RING_EDGE=ipc: peer copies into the next rank's buffers through
hipIpc memory handles, ordered through a shared-memory mailbox).  RCCL refuses two ranks on one device, so on a one-GPU
box this is the edge that can run `run_pipeline` / `run_lanes` with world >= 2 on real streams -- MultiBuffer's
make_progress / get_data / put_data between ranks (utils/MultiBuffer.cpp:287-609) -- and examples/pipeline_host.cpp's
between-process branch.  All ranks use device 0; torch.distributed (gloo) only carries the edge ids.

Every step of every run is compared with the reference's golden checksums (static beam: the reference's own 2-rank
test asserts the same JSON as one rank, tests/blowout_wake_explicit.2Rank.sh) or with the CPU oracle stepping the same
deck (moving beam, evolving laser pulse)."""
import json
import os
import socket
import subprocess

import numpy as np
import pytest

from hipace_amd import decks

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from hipace_amd import _lib, api as A
    _lib.lib()      # raises if libhpslice.so is missing: no fallback
    return A


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _moving_deck():
    d = decks.beam_evolution()
    d.update(nz=12, lo=(-2.0, -2.0, -2.4), hi=(2.0, 2.0, 2.4), beam_zmin=-1.0, beam_zmax=1.6, beam_umean=(0.0, 0.0, 1.2),
             beam_density=1.0e-3, n_steps=1, dt=0.9, beam_n_subcycles=4, ext_E_slope=(0.3, 0.2))
    return d


def _laser_deck():
    d = decks.laser_blowout_wake()
    d.update(nx=32, ny=32, nz=16, lo=(-16.0, -16.0, -4.0), hi=(16.0, 16.0, 4.0), laser_a0=1.5, laser_lambda0=0.4, laser_solver=1, dt=5.0)
    return d


def _deck(kind):
    if kind == "static":
        d = decks.blowout_wake()
        d["n_steps"] = 1
        return d
    return _moving_deck() if kind == "moving" else _laser_deck()


def _worker(rank, world, port, lanes, n_steps, kind, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HPS_RING_EDGE"] = "ipc"
    os.environ["HPS_RING_TIMEOUT_S"] = "120"
    os.environ["HPS_RING_CONNECT_TIMEOUT_S"] = "120"
    try:
        import torch
        import torch.distributed as dist
        from hipace_amd import api
        from hipace_amd.pipeline import RingTransport, run_lanes
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        deck = _deck(kind)
        engs = [api.SliceEngine(deck, device=0, tile_size=16 if kind == "static" else 0, sort_period=16) for _ in range(lanes)]
        for e in engs:
            e.set_diagnostics(True)
        T = RingTransport(rank, world, 0)
        assert T.kind == "ipc"
        res = {}

        def on_step_end(step, eng):
            eng.sync()
            res[step] = (eng.checksums(), eng.laser_envelope().copy() if kind == "laser" else None)

        solved = run_lanes(engs, rank, world, n_steps, torch.device("cuda", 0), on_step_end, transport=T, laser_lookahead=3)
        st, inf = T.stats(), T.info()
        dist.barrier()
        T.close()
        out.put((rank, solved, res, st, inf, None))
        dist.destroy_process_group()
    except Exception as exc:      # noqa: BLE001
        import traceback
        out.put((rank, -1, {}, None, None, traceback.format_exc() + str(exc)))


CASES = [(2, 1, 2, "static"), (2, 1, 5, "static"), (3, 1, 7, "static"), (2, 2, 4, "static"), (2, 2, 9, "static"), (3, 2, 8, "static"),
         (4, 1, 6, "static"),
         (2, 1, 4, "moving"), (2, 2, 6, "moving"), (2, 1, 4, "laser"), (2, 2, 5, "laser")]


@pytest.mark.gpu
@pytest.mark.parametrize("world,lanes,n_steps,kind", CASES)
def test_processes_on_one_device_hand_over_through_the_ipc_edge(oracle, world, lanes, n_steps, kind):
    """world processes x lanes engines each = a ring of world*lanes stages on ONE device: open ring (n_steps = stages) and
    closed ring (more steps than stages), static beam (received in place into the engine's beam storage), moving beam
    (packed messages, rotating send slots) and the laser envelope's 2 x 2 planes per slice.  Every step has the reference's
    checksums / the oracle's; every rank's two edges report both ends attached and carried messages."""
    import torch.multiprocessing as mp
    deck = _deck(kind)
    want = {}
    if kind == "static":
        gold = json.load(open(os.path.join(GOLD, "blowout_wake_explicit.2Rank.json")))["lev=0"]
    else:
        ref = oracle.Engine(deck)
        for s in range(n_steps):
            ref.begin_step()
            for k in range(deck["nz"] - 1, -1, -1):
                ref.solve_slice(k)
            want[s] = (ref.checksums(), ref.laser_envelope().copy() if kind == "laser" else None)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, lanes, n_steps, kind, out)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = [out.get(timeout=420) for _ in range(world)]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for r in results:
        assert r[5] is None, f"rank {r[0]}: {r[5]}"
    got = {}
    W = world * lanes
    for rank, solved, res, st, inf, _ in results:
        mine = [s for s in range(n_steps) if (s % W) // lanes == rank]
        assert sorted(res) == mine, (rank, sorted(res), mine)
        assert solved == deck["nz"] * len(mine)
        got.update(res)
        assert inf["comm_in_ranks"] == 2 and inf["comm_out_ranks"] == 2, inf
        # a rank sends whenever one of its steps has a successor, receives whenever one has a predecessor
        sends = any(s + 1 < n_steps and (s % W) % lanes == lanes - 1 for s in mine)
        recvs = any(s > 0 and (s % W) % lanes == 0 for s in mine)
        assert (st["sent"] > 0) == sends and (st["received"] > 0) == recvs, (rank, st)
    assert sorted(got) == list(range(n_steps))
    for s in range(n_steps):
        if kind == "static":
            for k, v in gold.items():
                assert abs(got[s][0][k] - v) <= 1e-9 * abs(v), (s, k, got[s][0][k], v)
        else:
            if kind == "laser":
                assert np.abs(got[s][1] - want[s][1]).max() <= 1e-9 * np.abs(want[s][1]).max(), s
            for k, v in want[s][0].items():
                assert abs(got[s][0][k] - v) <= 1e-9 * max(abs(v), 1e-300), (s, k, got[s][0][k], v)


@pytest.mark.gpu
@pytest.mark.parametrize("world,stages,n_steps", [(2, 1, 2), (2, 1, 5), (2, 2, 7), (3, 1, 4), (2, 3, 8)])
def test_cpp_pipeline_host_as_several_processes(api, tmp_path, world, stages, n_steps):
    """examples/pipeline_host.cpp run as `world` PROCESSES on the one device (rank % devices): the between-process branch of
    the C++ host -- edge ids through files, receives of a whole step posted ahead, hps_ring_send_slice behind the engine's
    event, hps_ring_recv_landed / hps_ring_engine_wait before the slice that reads a block -- open and closed ring.  Every
    step reproduces the reference's checksums; the ring's counters say the blocks went between the processes."""
    exe = os.path.join(ROOT, "examples", "pipeline_host")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build_cpp_host()
    gold = json.load(open(os.path.join(GOLD, "blowout_wake_explicit.2Rank.json")))["lev=0"]
    eng = api.SliceEngine(decks.blowout_wake())
    names = eng.comp_names()
    nbeam, off = eng.beam_layout()
    nonempty = int((np.diff(off) > 0).sum())
    path = tmp_path / "deck.bin"
    path.write_bytes(bytes(eng._dk))
    del eng
    env = dict(os.environ, HPS_RING_EDGE="ipc", HPS_RING_TIMEOUT_S="120", HPS_RING_CONNECT_TIMEOUT_S="120")
    procs = [subprocess.Popen([exe, str(path), str(n_steps), str(r), str(world), str(tmp_path), str(stages)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            p.kill()
            o, e = p.communicate()
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, e[-2000:]
    steps = [l.split() for _, o, _ in outs for l in o.splitlines() if l.startswith("step ")]
    assert sorted(int(s[1]) for s in steps) == list(range(n_steps))
    for s in steps:
        cs = dict(zip(names, map(float, s[2:])))
        for k, v in gold.items():
            assert abs(cs[k] - v) <= 1e-9 * abs(v), (s[1], k, cs[k], v)
    W = world * stages
    sent = received = 0
    for r, (_, o, _) in enumerate(outs):
        ring = [l.split() for l in o.splitlines() if l.startswith("ring ")][0]
        sent += int(ring[1])
        received += int(ring[2])
    # every step but the last hands its non-empty blocks on; the hand-offs that cross a process boundary go through the ring
    crossing = sum(1 for s in range(n_steps - 1) if (s % W) % stages == stages - 1)
    assert sent == received == crossing * nonempty, (sent, received, crossing, nonempty)


def _fullsize_worker(rank, world, port, lanes, n_steps, deck, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HPS_RING_EDGE"] = "ipc"
    os.environ["HPS_RING_TIMEOUT_S"] = "180"
    try:
        import torch
        import torch.distributed as dist
        from hipace_amd import api
        from hipace_amd.pipeline import RingTransport, run_lanes
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        engs = [api.SliceEngine(deck, device=0, tile_size=16, sort_period=128) for _ in range(lanes)]
        for e in engs:
            e.set_diagnostics(True)
        T = RingTransport(rank, world, 0)
        res = {}

        def on_step_end(step, eng):
            eng.sync()
            res[step] = (eng.checksums(), eng.stats()["vcycles"])

        prev = {id(e): 0 for e in engs}
        solved = run_lanes(engs, rank, world, n_steps, torch.device("cuda", 0), on_step_end, transport=T)
        st = T.stats()
        dist.barrier()
        T.close()
        out.put((rank, solved, res, st, None))
        dist.destroy_process_group()
    except Exception as exc:      # noqa: BLE001
        import traceback
        out.put((rank, -1, {}, None, traceback.format_exc() + str(exc)))


@pytest.mark.gpu
def test_headline_box_through_two_processes_of_two_stages():
    """The bench's multi-rank configuration at FULL size, checked: the 1024 x 1024 x 1024, 4 ppc box (tests/golden/fullsize_config4.json, the
    oracle's whole box) run as six time steps by a closed ring of 2 processes x 2 stages on the one device -- every hand-off
    between the processes a peer copy through the ipc edge, the receives of a whole step posted ahead.  hipace.dt = 0: every step
    is the box of the fixture -- its checksums to 1e-6 (measured 4e-11) and its V-cycle total."""
    import torch.multiprocessing as mp
    path = os.path.join(GOLD, "fullsize_config4.json")
    if not os.path.exists(path):
        pytest.fail("tests/golden/fullsize_config4.json is missing (scripts/make_fullsize_fixtures.py --only config4 writes it)")
    fx = json.load(open(path))
    deck = {k: (tuple(v) if isinstance(v, list) else v) for k, v in fx["deck"].items()}
    world, lanes, n_steps = 2, 2, 6
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fullsize_worker, args=(r, world, port, lanes, n_steps, deck, out)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = [out.get(timeout=600) for _ in range(world)]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for r in results:
        assert r[4] is None, f"rank {r[0]}: {r[4]}"
    got = {}
    for rank, solved, res, st, _ in results:
        got.update(res)
        assert st["sent"] > 0 and st["received"] > 0, (rank, st)
    assert sorted(got) == list(range(n_steps))
    worst = 0.0
    per_stage = {}
    for s in range(n_steps):
        cs, vc_total = got[s]
        for k, v in fx["checksums"].items():
            if v == 0.0:
                assert cs[k] == 0.0, (s, k)
            else:
                worst = max(worst, abs(cs[k] - v) / abs(v))
                assert abs(cs[k] - v) <= 1e-6 * abs(v), (s, k, cs[k], v)
        per_stage.setdefault(s % (world * lanes), []).append(vc_total)
    # an engine's V-cycle counter runs over its steps: every step adds the box's total
    for stage, totals in per_stage.items():
        for n, t in enumerate(sorted(totals)):
            assert abs(t - (n + 1) * fx["final"]["vcycles"]) <= max(2, 2e-3 * t), (stage, totals)
    print(f"headline box through 2 processes x 2 stages, 6 steps: worst checksum deviation {worst:.2e}")


def _failure_worker(rank, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HPS_RING_EDGE"] = "ipc"
    os.environ["HPS_RING_TIMEOUT_S"] = "8"
    os.environ["HPS_RING_NO_PROBE"] = "1"
    try:
        import torch
        import torch.distributed as dist
        from hipace_amd import _lib
        from hipace_amd.pipeline import RingTransport
        dist.init_process_group("gloo", rank=rank, world_size=2)
        torch.cuda.set_device(0)
        T = RingTransport(rank, 2, 0)
        msgs = []
        a = torch.zeros(1024, dtype=torch.float64, device="cuda")
        b = torch.zeros(512, dtype=torch.float64, device="cuda")
        if rank == 1:
            T.recv(b, None, 0)                     # posts 4096 bytes; rank 0 sends 8192
            dist.barrier()
            dist.barrier()                         # rank 0 has seen the size mismatch
            T.close()                              # ... and now this rank leaves with rank 0 about to send again
            dist.barrier()
        else:
            dist.barrier()
            for what in ("size", "fits", "gone"):
                try:
                    T.send(a if what == "size" else b, None, 1)
                    msgs.append((what, "no error"))
                except _lib.HpsError as exc:
                    msgs.append((what, str(exc)))
                if what == "fits":
                    T.sync_sends()
                    dist.barrier()
                    dist.barrier()                 # rank 1 has closed its ring: nobody will post another receive
        out.put((rank, msgs, None))
        if rank == 0:
            T.close()
        dist.destroy_process_group()
    except Exception as exc:      # noqa: BLE001
        import traceback
        out.put((rank, [], traceback.format_exc() + str(exc)))


@pytest.mark.gpu
def test_ipc_edge_fails_loudly():
    """The ipc edge's error behaviour between two processes: a send whose size differs from the posted receive's is an error
    (RCCL would hang or corrupt), and a send to a rank that has left the ring -- no receive will ever be posted -- gives up
    with the edge's counters in the message instead of waiting for HPS_RING_TIMEOUT_S."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failure_worker, args=(r, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        results = dict((r[0], r) for r in (out.get(timeout=240) for _ in range(2)))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for r in results.values():
        assert r[2] is None, r[2]
    msgs = dict(results[0][1])
    assert "8192 bytes" in msgs["size"] and "4096" in msgs["size"], msgs
    assert msgs["fits"] == "no error", msgs
    assert "left the ring" in msgs["gone"] or "still waiting" in msgs["gone"], msgs
