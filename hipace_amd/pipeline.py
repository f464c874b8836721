"""Time-step ring pipeline over the GPUs of one node.

The reference does not decompose a slice or the zeta axis; it pipelines *time steps* over ranks
(src/Hipace.cpp:400-401: rank r runs steps r, r+N, ...) and hands the freshly pushed beam slice of
step s to the rank that runs step s+1 (src/utils/MultiBuffer.cpp:444-609; in-process "send to
myself" when there is one rank, :299-308).  This module is that schedule for `SliceEngine`.

Transport: torch.distributed point-to-point -- backend "nccl" (= RCCL over xGMI) between GPUs,
"gloo" in the CPU tests.  One message per slice: the slice's beam block, `7*count` doubles
(x, y, z, ux, uy, uz, w), contiguous in the engine's beam storage (include/hpslice.h,
hps_engine_beam_info).

Deadlock freedom by construction: all ranks advance through the same global pipeline *ticks*.
Rank r is 2r ticks behind rank 0 (solving slice k needs the beam of slice k AND of slice k-1, whose
jx/jy feed the explicit source term: Hipace.cpp:639-657, so a rank trails its predecessor by two
slices).  In tick t every rank posts, in one batch (= one RCCL group),
  * the send of the slice it solved in tick t-1 (if a later step exists), and
  * the receive of the slice its ring predecessor solved in tick t-1 (if that feeds one of its steps).
For r > 0 that is slice k-1 while it solves slice k in the same tick; rank 0 receives the slices of
its next step early (its predecessor, rank N-1, is only 2(N-1) ticks behind; requires nz >= 2N)
into the other of two beam buffers.  Every posted send therefore has a matching receive posted in
the same tick on the peer, and a rank only ever waits for messages of earlier-or-equal ticks.
"""
import torch
import torch.distributed as dist


def steps_of_rank(rank, world, n_steps):
    return list(range(rank, n_steps, world))


class _Sched:
    """Who solves what in which tick."""

    def __init__(self, world, n_steps, nz, per_step=None):
        self.world, self.n_steps, self.nz = world, n_steps, nz
        self.per_step = per_step or nz        # slices solved per step (head slices first)

    def work(self, rank, tick):
        """(step, islice, local_step_index) solved by `rank` in `tick`, or None."""
        loc = tick - 2 * rank
        if loc < 0:
            return None
        m, q = divmod(loc, self.per_step)
        step = rank + m * self.world
        if step >= self.n_steps:
            return None
        return step, self.nz - 1 - q, m

    def n_ticks(self):
        last = 0
        for r in range(self.world):
            n = len(steps_of_rank(r, self.world, self.n_steps))
            last = max(last, 2 * r + n * self.per_step)
        return last + 1      # one more tick to flush the final sends (there are none, but keep symmetric)


def run_pipeline(engine, rank, world, n_steps, device, on_step_end=None, slices_per_step=None):
    """Run steps rank, rank+world, ... < n_steps of `engine` with the per-slice ring hand-off.

    engine: SliceEngine-like (begin_step, solve_slice, sync, beam_layout, set_beam_storage,
    initial_beam_into).  slices_per_step < nz solves only the head slices of every step
    (benchmark runs shorter than one box).  Returns the number of slices this rank solved.
    """
    nz = engine.deck["nz"]
    per_step = slices_per_step or nz
    assert per_step >= 2 * world, "the ring pipeline needs at least 2 slices per rank"
    if getattr(engine, "moving", False):
        assert per_step == nz, "a moving beam (hipace.dt != 0) needs whole steps"
        return _run_pipeline_moving(engine, rank, world, n_steps, device, on_step_end)
    sched = _Sched(world, n_steps, nz, per_step)
    nbeam, off = engine.beam_layout()
    bufs = [torch.zeros(max(7 * nbeam, 1), dtype=torch.float64, device=device) for _ in range(2)]
    if rank == 0:
        engine.initial_beam_into(bufs[0])       # only the head rank injects the beam (as the reference)
    prev, nxt = (rank - 1) % world, (rank + 1) % world

    def block(buf, islice):
        p = nz - 1 - islice
        return buf[7 * off[p]:7 * off[p + 1]]

    pending_recv = {}     # (local step index, islice) -> request
    pending_send = []
    solved = 0
    for tick in range(sched.n_ticks()):
        ops = []
        # what my predecessor solved in the previous tick feeds my step (its step + 1)
        pw = sched.work(prev, tick - 1)
        if pw is not None and pw[0] + 1 < n_steps and (pw[0] + 1) % world == rank:
            m_target = (pw[0] + 1 - rank) // world
            t = block(bufs[m_target % 2], pw[1])
            if t.numel() > 0:
                if world == 1:
                    pass                      # in-process hand-off below
                else:
                    ops.append(("recv", dist.P2POp(dist.irecv, t, prev), (m_target, pw[1])))
        # what I solved in the previous tick goes to my successor
        mw = sched.work(rank, tick - 1)
        if mw is not None and mw[0] + 1 < n_steps:
            t = block(bufs[mw[2] % 2], mw[1])
            if t.numel() > 0:
                if world == 1:
                    block(bufs[(mw[2] + 1) % 2], mw[1]).copy_(t)      # MultiBuffer.cpp:299-308
                else:
                    ops.append(("send", dist.P2POp(dist.isend, t, nxt), None))
        # one batch per tick (one RCCL group: the send and the receive of a rank progress together, so a
        # closed ring -- more steps than ranks -- cannot park every rank in a receive whose matching send is
        # queued behind another receive).  Every send posted in tick t has its receive posted in tick t of
        # the successor, and a rank posts its batch before it waits for anything of that tick.
        if ops:
            reqs = dist.batch_isend_irecv([o[1] for o in ops])
            for i, o in enumerate(ops):
                rq = reqs[i] if len(reqs) == len(ops) else reqs[-1]     # coalescing back-ends: one request per batch
                if o[0] == "recv":
                    pending_recv[o[2]] = rq
                else:
                    pending_send.append(rq)

        w = sched.work(rank, tick)
        if w is None:
            continue
        step, islice, m = w
        if islice == nz - 1:
            engine.set_beam_storage(bufs[m % 2], injected_beam_support=True)     # hipace.dt = 0: the beam never moves
            engine.begin_step()
        waited = False
        for key in ((m, islice), (m, islice - 1)):            # this slice's beam and the next one's (jx/jy source)
            rq = pending_recv.pop(key, None)
            if rq is not None:
                rq.wait()
                waited = True
        if waited and str(device) != "cpu":
            torch.cuda.current_stream().synchronize()         # data landed before the engine's stream reads it
        engine.solve_slice(islice)
        engine.sync()                                         # the slice's beam block is final before it is sent
        solved += 1
        while len(pending_send) > 4:
            pending_send.pop(0).wait()
        if islice == nz - per_step and on_step_end is not None:
            on_step_end(step)
    for rq in pending_send:
        rq.wait()
    return solved


def _run_pipeline_moving(engine, rank, world, n_steps, device, on_step_end):
    """hipace.dt != 0: what sits on a slice after its push travels as one fixed-size message
    [count | 7 rows of `cap` doubles] (engine.export_beam_slice) and becomes the next step's slice
    (engine.import_beam_slice) -- MultiBuffer::put_data / get_data (utils/MultiBuffer.cpp:444-609).  Same tick
    schedule as the static path; the head rank injects the beam (its first step only)."""
    nz = engine.deck["nz"]
    sched = _Sched(world, n_steps, nz, nz)
    cap = engine.beam_capacity()
    mlen = 1 + 7 * cap
    prev, nxt = (rank - 1) % world, (rank + 1) % world
    # receive slots: two steps' worth (a rank may hold the early slices of its next step); send slots rotate
    rpool = [[torch.zeros(mlen, dtype=torch.float64, device=device) for _ in range(nz)] for _ in range(2)]
    spool = [torch.zeros(mlen, dtype=torch.float64, device=device) for _ in range(8)]
    pending_recv, pending_send, have = {}, [], set()
    exported = {}          # tick -> send slot holding the slice solved in that tick
    solved = 0
    for tick in range(sched.n_ticks()):
        ops = []
        pw = sched.work(prev, tick - 1)
        if world > 1 and pw is not None and pw[0] + 1 < n_steps and (pw[0] + 1) % world == rank:
            m_target = (pw[0] + 1 - rank) // world
            ops.append(("recv", dist.P2POp(dist.irecv, rpool[m_target % 2][nz - 1 - pw[1]], prev), (m_target, pw[1])))
        mw = sched.work(rank, tick - 1)
        if world > 1 and mw is not None and mw[0] + 1 < n_steps:
            ops.append(("send", dist.P2POp(dist.isend, exported.pop(tick - 1), nxt), None))
        if ops:
            reqs = dist.batch_isend_irecv([o[1] for o in ops])
            for i, o in enumerate(ops):
                rq = reqs[i] if len(reqs) == len(ops) else reqs[-1]
                if o[0] == "recv":
                    pending_recv[o[2]] = rq
                else:
                    pending_send.append(rq)

        w = sched.work(rank, tick)
        if w is None:
            continue
        step, islice, m = w
        from_ring = step > 0                                    # step 0 is the injected beam of the head rank
        if islice == nz - 1:
            engine.set_beam_import(from_ring)
            engine.begin_step()
        if from_ring:
            waited = False
            for k in (islice, islice - 1):                      # this slice and the next one (jx/jy source)
                if k < 0 or (m, k) in have:
                    continue
                rq = pending_recv.pop((m, k), None)
                if rq is not None:
                    rq.wait()
                    waited = True
                if waited and str(device) != "cpu":
                    torch.cuda.current_stream().synchronize()
                engine.import_beam_slice(k, rpool[m % 2][nz - 1 - k])
                have.add((m, k))
        engine.solve_slice(islice)
        if step + 1 < n_steps:
            if world == 1:
                engine.export_beam_slice(islice, rpool[(m + 1) % 2][nz - 1 - islice])      # in-process hand-off
            else:
                slot = spool[tick % len(spool)]
                engine.export_beam_slice(islice, slot)
                exported[tick] = slot
        engine.sync()                                         # the message is complete before it is sent
        solved += 1
        while len(pending_send) > 4:
            pending_send.pop(0).wait()
        if islice == 0:
            have = {h for h in have if h[0] != m}
            if on_step_end is not None:
                on_step_end(step)
    for rq in pending_send:
        rq.wait()
    return solved


def make_edge_groups(world):
    """Process groups for the ring edges of `run_local_pipeline` with world > 1 (collective: every rank calls it, in
    the same order).  Edge r -> r+1 uses group colour(r); a rank's incoming and outgoing edge never share a group, so
    its receiving thread and its sending thread each own one communicator (a proper edge colouring of the ring: two
    colours, three when the ring is odd)."""
    if world == 1:
        return None
    return [dist.new_group(list(range(world))) for _ in range(3)]


def _edge_colour(r, world):
    return 2 if (world % 2 == 1 and r == world - 1) else r % 2


def run_local_pipeline(engines, n_steps, device, on_step_end=None, slices_per_step=None, rank=0, world=1,
                       groups=None):
    """Several time steps in flight on ONE device: the ring pipeline with L = len(engines) of its stages in this process.

    Stage g = rank*L + j (engine j of this rank, one stream each, all on `device`) runs steps g, g+G, ... < n_steps,
    G = world*L, exactly as rank g of `run_pipeline` would.  What couples the stages is the same per-slice beam
    hand-off: stage g solves slice k of step s only after stage g-1 has pushed slices k and k-1 of step s-1.  Between
    two engines of this process that is a device-to-device copy (MultiBuffer.cpp:299-308, the reference's in-process
    "send to myself") ordered by stream events; between the last engine of a rank and the first engine of the next
    rank (world > 1) it is one point-to-point message per slice on that edge's own process group (`make_edge_groups`:
    the sending and the receiving thread of a rank never share a communicator, every edge carries one ordered sequence
    of messages posted in the same order on both sides).  One host thread per engine (the multigrid's stopping rule
    holds its host thread once per slice; ctypes releases the GIL), so while one step sits in a latency-bound phase --
    the lower multigrid levels, a DST pass, a launch gap -- the kernels of the others fill the device.  Static beam
    only (hipace.dt = 0, as every BASELINE deck).

    Returns the number of slices this process solved.
    """
    import threading
    L = len(engines)
    G = world * L
    nz = engines[0].deck["nz"]
    per_step = slices_per_step or nz
    assert per_step >= 2, "a step needs at least two slices"
    assert world == 1 or groups is not None, "world > 1 needs make_edge_groups(world)"
    laser = bool(getattr(engines[0], "has_laser", False))
    on_gpu = str(device) != "cpu"
    nbeam, off = engines[0].beam_layout()
    assert nbeam == 0 or not getattr(engines[0], "moving", False), "run_local_pipeline hands a static beam on (hipace.dt = 0)"
    bufs = [[torch.zeros(max(7 * nbeam, 1), dtype=torch.float64, device=device) for _ in range(2)] for _ in range(L)]
    if rank == 0:
        engines[0].initial_beam_into(bufs[0][0])      # only the head of the ring injects the beam
        engines[0].sync()
    prev_rank, next_rank = (rank - 1) % world, (rank + 1) % world
    g_in = groups[_edge_colour(prev_rank, world)] if world > 1 else None
    g_out = groups[_edge_colour(rank, world)] if world > 1 else None

    def block(buf, q):
        p = q                                          # q-th slice from the head = block q
        return buf[7 * off[p]:7 * off[p + 1]]

    cond = threading.Condition()
    progress = [0] * L                                 # slices enqueued so far by each engine
    events = [dict() for _ in range(L)]                # (local step, q) -> event recorded after that slice
    errors = []
    solved = [0] * L

    def lane(j):
        try:
            if on_gpu:
                torch.cuda.set_device(device)
            eng, pj = engines[j], (j - 1) % L
            stage = rank * L + j
            remote_in = world > 1 and j == 0           # my predecessor stage lives on the previous rank
            remote_out = world > 1 and j == L - 1      # my successor stage lives on the next rank
            sends = []
            # laser messages on the rank-to-rank edges: {a_{n+1}, a_n} of a slice, packed / unpacked on the device
            lmsg_in = torch.zeros(eng.laser_message_doubles(), dtype=torch.float64, device=device) if (laser and remote_in) else None
            lpool = []                                  # (buffer, request) of the laser sends in flight
            for m, step in enumerate(range(stage, n_steps, G)):
                buf = bufs[j][m % 2]
                fed = step > 0                          # step 0 starts from the injected beam
                mp = (step - 1 - (rank * L + pj)) // G if (fed and not remote_in) else None
                if nbeam > 0:
                    eng.set_beam_storage(buf, injected_beam_support=True)
                if laser:
                    # the stage that runs step 0 evaluates the initial envelope; every later step receives a_n, a_{n-1}
                    # slice by slice from the stage that ran the step before (MultiBuffer.cpp:840-852, 913-925)
                    eng.set_laser_import(fed, step)
                eng.begin_step()
                copied = 0
                lcopied = 0
                for q in range(per_step):
                    if fed:
                        need = min(q + 1, per_step - 1)             # this slice's beam and the next one's (jx/jy source)
                        if remote_in:
                            # same order as the sender posts them: per slice the beam block (if any), then the laser
                            while copied <= need:
                                d = block(buf, copied)
                                if d.numel() > 0:
                                    dist.irecv(d, src=prev_rank, group=g_in).wait()
                                    if on_gpu:
                                        torch.cuda.current_stream().synchronize()   # landed before the engine's stream reads it
                                if laser:
                                    dist.irecv(lmsg_in, src=prev_rank, group=g_in).wait()
                                    if on_gpu:
                                        torch.cuda.current_stream().synchronize()
                                    eng.import_laser_slice(nz - 1 - copied, lmsg_in)
                                    eng.sync()                      # lmsg_in is free for the next slice
                                copied += 1
                        else:
                            with cond:
                                while progress[pj] < mp * per_step + need + 1 and not errors:
                                    cond.wait(timeout=1.0)
                                ev = events[pj].get((mp, need))
                            if errors:
                                return
                            eng.wait_event(ev)
                            src = bufs[pj][mp % 2]
                            while copied <= need:
                                d, s_ = block(buf, copied), block(src, copied)
                                if d.numel() > 0:
                                    eng.copy_async(d, s_)
                                copied += 1
                            if laser:
                                while lcopied <= need:
                                    eng.import_laser_from(nz - 1 - lcopied, engines[pj])
                                    lcopied += 1
                    eng.solve_slice(nz - 1 - q)
                    solved[j] += 1
                    if remote_out:
                        if step + 1 < n_steps:
                            t = block(buf, q)
                            if t.numel() > 0:
                                eng.sync()                          # the slice's beam block is final before it is sent
                                sends.append(dist.isend(t, dst=next_rank, group=g_out))
                                # never block on a send here: with one stage per rank this thread also posts the
                                # receives its peer's sends are waiting for (a whole step of blocks may be in flight)
                                while sends and sends[0].is_completed():
                                    sends.pop(0)
                            if laser:
                                free = [k for k, (_, rq) in enumerate(lpool) if rq.is_completed()]
                                if free:
                                    lb = lpool.pop(free[0])[0]
                                else:                               # never wait for a send (see above): take a new buffer
                                    lb = torch.zeros(eng.laser_message_doubles(), dtype=torch.float64, device=device)
                                eng.export_laser_slice(nz - 1 - q, lb)
                                eng.sync()
                                lpool.append((lb, dist.isend(lb, dst=next_rank, group=g_out)))
                    ev = eng.record_event((m % 2) * per_step + q)
                    with cond:
                        events[j][(m, q)] = ev
                        events[j].pop((m - 2, q), None)
                        progress[j] = m * per_step + q + 1
                        cond.notify_all()
                if on_step_end is not None:
                    on_step_end(step, eng)
            for rq in sends:
                rq.wait()
            for _, rq in lpool:
                rq.wait()
        except BaseException as e:      # noqa: BLE001 -- re-raised on the caller's thread
            with cond:
                errors.append(e)
                cond.notify_all()

    if L == 1:
        lane(0)
    else:
        threads = [threading.Thread(target=lane, args=(j,), name=f"hps-step-lane-{j}") for j in range(L)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if errors:
        raise errors[0]
    for e in engines:
        e.sync()
    return sum(solved)
