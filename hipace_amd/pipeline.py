"""Time-step ring pipeline over the GPUs of one node.

The reference does not decompose a slice or the zeta axis; it pipelines *time steps* over ranks
(src/Hipace.cpp:400-401: rank r runs steps r, r+N, ...) and hands the freshly pushed beam slice (and the laser
envelope of that slice) of step s to the rank that runs step s+1 (src/utils/MultiBuffer.cpp:287-609; in-process "send
to myself" when there is one rank, :299-308).  This module is the host driver of that schedule for `SliceEngine`.

Transport: `RcclTransport` = the C-ABI ring of include/hpslice.h (hps_ring_*: ncclSend / ncclRecv over xGMI, one
2-rank communicator and one stream per ring edge); `GlooTransport` = torch.distributed point-to-point on host tensors
for the CPU tests (the oracle engine stands in for the HIP engine).  One message per slice and kind, in a fixed order
per edge: for every slice (head first) the beam message, then the laser message:
  * static beam (hipace.dt = 0): the slice's beam block, `7*count` doubles, straight out of / into the engine's beam
    storage (hps_engine_beam_info / set_beam_storage); empty blocks are not sent;
  * moving beam: `1 + 7*cap` doubles packed / unpacked on the device (hps_engine_export_beam_slice / import_beam_slice);
  * laser: `4 nx ny` doubles ({a_{n+1}, a_n} of the slice, hps_engine_export_laser_slice / import_laser_slice).

No host synchronisation per slice.  The engine records an event behind every slice; the send waits for it on the
ring's send stream; the engine's stream waits for the event the ring records behind a receive.  Receives are posted a
whole step ahead (two steps' worth of beam buffers: the slot of (step m+1, slice j) is free once step m-1 is done), as
the reference's MultiBuffer does with its default unlimited `max_leading_slices` -- that is what lets the ring close
(more steps than ranks) without a rank ever waiting for its successor.  The laser messages are 33 MB per slice at
1024^2: they are posted `laser_lookahead` slices ahead when the ring does not close (n_steps <= ranks) and a whole step
ahead when it does (the reference's requirement max_trailing_slices * n_ranks > nslices, MultiBuffer.cpp:87-91).

Dependencies: solving slice k of step s+1 needs the beam of slices k and k-1 of step s (the next slice's jx / jy feed
the explicit source term, Hipace.cpp:639-657), so a rank trails its predecessor by two slices.
"""
import gc
import os
import time
import ctypes as C

import torch
import torch.distributed as dist


def steps_of_rank(rank, world, n_steps):
    return list(range(rank, n_steps, world))


class GlooTransport:
    """torch.distributed point-to-point on host tensors (CPU tests; the oracle engine is synchronous).  What the ring
    calls an event is a request here, and waiting happens on the host."""

    def __init__(self, rank, world, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.prev, self.next = (rank - 1) % world, (rank + 1) % world
        self._sends, self._recvs = [], []

    def send(self, t, after_event=None, slot=0):
        rq = dist.isend(t, self.next, group=self.group)
        self._sends.append(rq)
        if len(self._sends) > 64:
            self._sends = [r for r in self._sends if not r.is_completed()]
        return rq

    def recv(self, t, after_event=None, slot=0):
        if after_event is not None and not after_event.is_completed():
            after_event.wait()
        rq = dist.irecv(t, self.prev, group=self.group)
        self._recvs.append(rq)
        if len(self._recvs) > 64:
            self._recvs = [r for r in self._recvs if not r.is_completed()]
        return rq

    def engine_wait(self, engine, ev):
        if ev is not None and not ev.is_completed():
            ev.wait()

    engine_wait_sent = engine_wait      # (a send's request: the message buffer may be reused)

    def engine_wait_ordered(self, engine, evs):
        for ev in evs:
            self.engine_wait(engine, ev)

    def ready(self, ev):
        return True

    def can_send(self):
        return True

    def recv_after(self, ev):
        if ev is not None and hasattr(ev, "wait") and not ev.is_completed():
            ev.wait()

    def sync_sends(self):
        for rq in self._sends:
            if not rq.is_completed():       # (a second wait on a finished gloo request never returns)
                rq.wait()
        self._sends = []

    def finish(self):
        self.sync_sends()
        for rq in self._recvs:
            if not rq.is_completed():
                rq.wait()
        self._recvs = []

    def close(self):
        pass


def host_identity():
    """what tells two ranks that they share a node (and with it /dev/shm and the devices' IPC handles)"""
    import socket
    try:
        boot = open("/proc/sys/kernel/random/boot_id").read().strip()
    except OSError:
        boot = ""
    return f"{socket.gethostname()}|{boot}"


def ring_edge():
    """Kind of edge the C-ABI ring makes (HPS_RING_EDGE, read by hps_ring_unique_id): "ipc" (default: peer copies ordered
    through a shared-memory mailbox; one node, any device assignment -- also several ranks on ONE device) or "rccl"."""
    return "rccl" if os.environ.get("HPS_RING_EDGE", "") == "rccl" else "ipc"


class RingTransport:
    """The C-ABI ring (include/hpslice.h hps_ring_*), one edge object and one stream per ring edge.  Two kinds of edge under
    the same calls: RCCL ncclSend / ncclRecv on device buffers (one 2-rank communicator per edge), or -- HPS_RING_EDGE=ipc,
    the default -- a peer copy into the receiver's buffer (hipIpc memory handles) ordered through a mailbox in shared memory.
    Bootstrap: every rank makes the id of its outgoing edge, all ids are exchanged through the torch.distributed group (any
    backend), hps_ring_init is collective."""

    def __init__(self, rank, world, device_index, group=None, edge=None):
        from . import _lib
        if edge is not None:                                    # (the library reads the variable when the edge's id is made)
            assert edge in ("ipc", "rccl")
            os.environ["HPS_RING_EDGE"] = edge
        if world > 1 and ring_edge() == "rccl":
            # hardware queues (see _lib._ring_environment): the variable must have been in the environment before HIP started
            override = os.environ.get("HPS_RING_ALLOW_SHARED_QUEUES", "0") not in ("", "0")
            if int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < 8 and not override:
                if _lib.hip_already_started():
                    raise RuntimeError("RingTransport: an RCCL ring of %d ranks needs GPU_MAX_HW_QUEUES >= 8 in the environment BEFORE the "
                                       "process touches the GPU (export it in the launcher; importing hipace_amd._lib sets it under a "
                                       "multi-rank launcher, but this process had already created an engine or initialised torch.cuda); "
                                       "HPS_RING_ALLOW_SHARED_QUEUES=1 overrides, HPS_RING_EDGE=ipc needs none of this" % world)
                os.environ["GPU_MAX_HW_QUEUES"] = "8"
            elif not _lib._HWQ_PRESET and _lib._HIP_STARTED_AT_IMPORT and not override:
                raise RuntimeError("RingTransport: GPU_MAX_HW_QUEUES was set after torch had initialised the device -- the runtime did "
                                   "not see it; export it in the launcher's environment (HPS_RING_ALLOW_SHARED_QUEUES=1 overrides)")
        self._lib, self._check = _lib.lib(), _lib.check
        self.rank, self.world = rank, world
        self._h = None
        self.fell_back = None                   # why a ring that was to be ipc is RCCL (text), else None
        chosen_here = edge is None and not os.environ.get("HPS_RING_EDGE")
        if world > 1 and chosen_here:
            # the default edge is ipc: POSIX shared memory + hipIpc handles, i.e. ONE node.  A ring that spans hosts takes RCCL
            # (decided by all ranks together, before any rank makes a mailbox its neighbour on another host could never open)
            ids = [None] * world
            dist.all_gather_object(ids, host_identity(), group=group)
            if len(set(ids)) > 1:
                os.environ["HPS_RING_EDGE"] = "rccl"
                self.fell_back = f"the ring spans {len(set(ids))} hosts"
        if world > 1 and ring_edge() == "ipc" and edge is None and os.environ.get("HPS_RING_NO_PROBE", "0") in ("", "0"):
            # One small message around the ring before anything is timed: it opens the next rank's allocation in this process
            # (hipIpcOpenMemHandle + peer access, first use) and proves that a peer copy and its flags arrive.  If any rank
            # cannot connect or cannot pass the message (a node whose devices have no peer access, a runtime without IPC, a
            # neighbour whose shared memory is not ours) ALL ranks fall back to the RCCL edge together -- decided through the
            # torch.distributed group, loudly.
            ok, why = True, ""
            try:
                self._connect(rank, world, device_index, group)
            except Exception as exc:      # noqa: BLE001
                ok, why = False, f"hps_ring_init: {type(exc).__name__}: {exc}"
            if ok:
                ok = self._probe(device_index)
                why = self._probe_error
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 0:
                import sys
                print(f"hipace_amd.pipeline: rank {rank}: the ipc edge could not be set up on some rank ({'here: ' + why if not ok else 'not here'}); "
                      "all ranks fall back to HPS_RING_EDGE=rccl", file=sys.stderr, flush=True)
                self.close()
                self.fell_back = "the ipc edge's connection or probe message failed" + (f" on this rank: {why}" if not ok else " on another rank")
                os.environ["HPS_RING_EDGE"] = "rccl"
                self._connect(rank, world, device_index, group)
        else:
            self._connect(rank, world, device_index, group)

    def _connect(self, rank, world, device_index, group):
        my = C.create_string_buffer(128)
        self._check(self._lib.hps_ring_unique_id(my))
        ids = [None] * world
        if world > 1:
            dist.all_gather_object(ids, my.raw, group=group)
        else:
            ids[0] = my.raw
        h = C.c_void_p()
        self._check(self._lib.hps_ring_init(rank, world, int(device_index), ids[(rank - 1) % world] if world > 1 else None,
                                            ids[rank], C.byref(h)))
        self._h = h
        self.kind = "ipc" if self._lib.hps_ring_edge_kind(h) == 1 else "rccl"
        self._polls = self.kind == "ipc" and world > 1       # the host asks before it waits (several stages per thread)

    def _probe(self, device_index, seconds=90.0):
        """one 4 KB message to the next rank and one from the previous rank, checked; False (and self._probe_error) on any failure"""
        old = os.environ.get("HPS_RING_TIMEOUT_S")
        os.environ["HPS_RING_TIMEOUT_S"] = str(int(seconds))
        self._probe_error = ""
        try:
            dev = torch.device("cuda", int(device_index))
            rx = torch.zeros(512, dtype=torch.float64, device=dev)
            tx = torch.arange(512, dtype=torch.float64, device=dev) + 1000.0 * self.rank
            torch.cuda.synchronize(dev)
            ev = self.recv(rx, None, slot=0)
            self.send(tx, None, slot=0)
            t0 = time.perf_counter()
            while not self.ready(ev):
                if time.perf_counter() - t0 > seconds:
                    raise RuntimeError(f"no probe message from rank {(self.rank - 1) % self.world} after {seconds:.0f} s")
            self.sync_sends()
            torch.cuda.synchronize(dev)
            want = torch.arange(512, dtype=torch.float64, device=dev) + 1000.0 * ((self.rank - 1) % self.world)
            if not bool((rx == want).all()):
                raise RuntimeError("the probe message arrived with the wrong contents")
            self._probe_buffers = (rx, tx)        # (the allocation stays mapped in the neighbour until the ring is destroyed)
            return True
        except Exception as exc:      # noqa: BLE001
            self._probe_error = f"{type(exc).__name__}: {exc}"
            return False
        finally:
            if old is None:
                os.environ.pop("HPS_RING_TIMEOUT_S", None)
            else:
                os.environ["HPS_RING_TIMEOUT_S"] = old

    def send(self, t, after_event=None, slot=0):
        done = C.c_void_p()
        self._check(self._lib.hps_ring_send_slice(self._h, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(),
                                                  C.c_void_p(after_event) if after_event else None, int(slot), C.byref(done)))
        return done.value

    def recv(self, t, after_event=None, slot=0):
        done = C.c_void_p()
        self._check(self._lib.hps_ring_recv_slice(self._h, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(),
                                                  C.c_void_p(after_event) if after_event else None, int(slot), C.byref(done)))
        return done.value

    def sendrecv_self(self, src, dst, after_event=None, slot=0):
        done = C.c_void_p()
        self._check(self._lib.hps_ring_sendrecv_self(self._h, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()),
                                                     src.numel() * src.element_size(),
                                                     C.c_void_p(after_event) if after_event else None, int(slot), C.byref(done)))
        return done.value

    def engine_wait(self, engine, ev):
        """order the engine's stream behind a received message (RCCL: a device-side wait for the receive's event; ipc: the
        host waits -- not at all in steady state -- until the sender's stream has flagged the message as landed)"""
        if ev:
            self._check(self._lib.hps_ring_engine_wait(self._h, engine._h, C.c_void_p(ev)))

    def engine_wait_sent(self, engine, ev):
        """the engine's stream waits for a send's done-event (a device event with either kind of edge): the buffer may be rewritten"""
        engine.wait_event(ev)

    def engine_wait_ordered(self, engine, evs):
        """events of receives in the order they were posted: they complete in that order (one stream), the last one says it all"""
        if evs:
            self.engine_wait(engine, evs[-1])

    def ready(self, ev):
        """has the message behind this receive landed?  (RCCL edge: the engine's stream waits on the device -- always go on)"""
        if not self._polls:
            return True
        r = self._lib.hps_ring_recv_landed(self._h, C.c_void_p(ev))
        if r < 0:
            raise RuntimeError("hps_ring_recv_landed: not a receive of this ring")
        return r == 1

    def can_send(self):
        """would the next send go out without the host waiting for the receiver (its receive posted, the buffer free)?"""
        return (not self._polls) or self._lib.hps_ring_can_send(self._h) == 1

    def recv_after(self, ev):
        if ev:
            self._check(self._lib.hps_ring_stream_wait(self._h, 0, C.c_void_p(ev)))

    def sync_sends(self):
        self._check(self._lib.hps_ring_sync_sends(self._h))

    def finish(self):
        self._check(self._lib.hps_ring_sync(self._h))

    def stats(self):
        """messages and bytes of the run (the probe message of the constructor, one each way, is not counted)"""
        ns, nr, bs, br = C.c_long(), C.c_long(), C.c_longlong(), C.c_longlong()
        self._check(self._lib.hps_ring_stats(self._h, C.byref(ns), C.byref(nr), C.byref(bs), C.byref(br)))
        p = 1 if getattr(self, "_probe_buffers", None) is not None else 0
        return dict(sent=ns.value - p, received=nr.value - p, bytes_sent=bs.value - 4096 * p, bytes_received=br.value - 4096 * p,
                    probe_messages=p)

    def info(self):
        """What RCCL reports for this rank's two edge communicators (hps_ring_info)."""
        v = [C.c_int() for _ in range(5)]
        self._check(self._lib.hps_ring_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("world", "comm_in_ranks", "comm_out_ranks", "my_rank_in", "my_rank_out"), (x.value for x in v)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hps_ring_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


RcclTransport = RingTransport          # the name of rounds 2-4


class RcclSelfRing(RingTransport):
    """One rank that is its own ring neighbour, with the hand-off going through RCCL all the same (MultiBuffer.cpp:299-308
    with the transport left in): `run_pipeline(..., transport=RcclSelfRing(device))` then runs the multi-rank code path --
    receives posted a step ahead, sends behind the engine's events, the engine waiting for receive events, buffers reused
    behind send / import events -- on a single GPU.  A receive is only remembered when it is posted; the message's
    ncclSend + ncclRecv are issued as one group when the matching send comes (messages are matched in order, as on an
    edge between two ranks), and the event the engine later waits for is looked up then."""
    self_ring = True

    def __init__(self, device_index, edge="rccl"):
        old = os.environ.get("HPS_RING_EDGE")
        try:
            super().__init__(0, 1, device_index, edge=edge)
        finally:
            if old is None:
                os.environ.pop("HPS_RING_EDGE", None)
            else:
                os.environ["HPS_RING_EDGE"] = old
        self._posted, self._done, self._n_posted, self._n_sent = [], {}, 0, 0

    def recv(self, t, after_event=None, slot=0):
        # the wait for the buffer's previous use is enqueued NOW, on the stream that will carry the message (as
        # hps_ring_recv_slice does between ranks): an enqueued wait keeps the record it saw.  Kept as a handle until the send --
        # a whole step later when receives are posted ahead -- it named an event slot of the sending engine's rotating pool
        # that had been re-recorded since: the message then waited for a RECENT slice of the first stage, which made the
        # first and the last stage of a rank take turns (three stages: 1552 instead of 1990 slices/s)
        if after_event:
            self._check(self._lib.hps_ring_stream_wait(self._h, 1, C.c_void_p(after_event)))
        self._posted.append((t, None))
        self._n_posted += 1
        return ("self", self._n_posted - 1)

    def send(self, t, after_event=None, slot=0):
        dst, free_ev = self._posted.pop(0)
        assert dst.numel() == t.numel(), "self ring: message sizes of send and posted receive differ"
        if free_ev:                       # the receive buffer's previous contents are no longer needed
            self._check(self._lib.hps_ring_stream_wait(self._h, 1, C.c_void_p(free_ev)))
        ev = self.sendrecv_self(t, dst, after_event, slot)
        self._done[self._n_sent] = ev
        self._n_sent += 1
        return ev

    def engine_wait(self, engine, ev):
        if isinstance(ev, tuple):         # a receive: its message has been sent by now (the previous step is over)
            ev = self._done.pop(ev[1])
        engine.wait_event(ev)

    def engine_wait_ordered(self, engine, evs):
        evs = [self._done.pop(ev[1]) if isinstance(ev, tuple) else ev for ev in evs]
        if evs:
            engine.wait_event(evs[-1])

    def recv_after(self, ev):
        if ev:
            self._check(self._lib.hps_ring_stream_wait(self._h, 1, C.c_void_p(ev)))

    def finish(self):
        assert not self._posted, "self ring: receives without a matching send"
        super().finish()


def _wait_generic(engine, ev):
    """wait for an event of any transport: None, a gloo request (host wait), or a device event handle (the engine's stream
    waits for it)"""
    if ev is None:
        return
    if hasattr(ev, "is_completed"):
        if not ev.is_completed():
            ev.wait()
        return
    engine.wait_event(ev)


class LocalEdge:
    """The edge stage v -> stage v + 1 when both live in this process (several time steps in flight on one device, driven
    by one host thread): MultiBuffer.cpp:299-308's "send to myself" between two engines.  A message is a device-to-device
    copy on the SENDING engine's stream (behind the slice that produced it, ahead of the event the receiver waits for);
    receives are only remembered when they are posted and matched in order when the send comes, as on an edge between two
    ranks.  `_EV_LOCAL` slots of the sending engine's event pool rotate (an enqueued wait keeps the record it saw)."""

    def __init__(self):
        self.posted, self.done, self.n_posted, self.n_sent = [], {}, 0, 0
        self.pending_after = None
        self.tx_engine = None
        self.messages = 0

    def post(self, t, after_event):
        if after_event is None:
            after_event, self.pending_after = self.pending_after, None
        self.posted.append((t, after_event))
        self.n_posted += 1
        return ("loc", self, self.n_posted - 1)

    def can_send(self):
        return bool(self.posted)

    def send(self, engine, t):
        dst, free_ev = self.posted.pop(0)
        assert dst.numel() == t.numel(), "local edge: message sizes of send and posted receive differ"
        _wait_generic(engine, free_ev)            # the receive buffer's previous contents are no longer needed
        engine.copy_async(dst, t)
        ev = engine.record_event(_EV_LOCAL + self.n_sent % 2048)
        self.done[self.n_sent] = ev
        self.n_sent += 1
        self.messages += 1
        return ev


class StageTransport:
    """What one pipeline stage sees of its two edges: receives come from `rx` (a LocalEdge, or a ring / gloo transport),
    sends go to `tx`.  Events are device event handles (or gloo requests on the CPU) whichever side made them, so a buffer
    that is received into and later sent from can be ordered across the two."""

    def __init__(self, engine, rx, tx):
        self.engine, self.rx, self.tx = engine, rx, tx
        self.self_ring = True                      # always hand on through the transport (never the one-engine in-process path)

    # ---- receive side
    def recv(self, t, after_event=None, slot=0):
        if isinstance(self.rx, LocalEdge):
            return self.rx.post(t, after_event)
        return self.rx.recv(t, after_event, slot)

    def recv_after(self, ev):
        if isinstance(self.rx, LocalEdge):
            self.rx.pending_after = ev
        else:
            self.rx.recv_after(ev)

    def ready(self, ev):
        """has the message behind this receive ticket been sent (a local edge's tickets), has it landed (an ipc edge of the
        ring)?  Anything else is an event the engine's stream waits for on the device."""
        if isinstance(ev, tuple) and ev and ev[0] == "loc":
            return ev[2] in ev[1].done
        if isinstance(self.rx, LocalEdge) or isinstance(ev, tuple) or ev is None:
            return True
        return self.rx.ready(ev)

    @property
    def wait_in_tag(self):
        return "wait" if isinstance(self.rx, LocalEdge) else "wait-ring"

    @property
    def wait_out_tag(self):
        return "wait" if isinstance(self.tx, LocalEdge) else "wait-ring"

    def _resolve(self, ev):
        if isinstance(ev, tuple) and ev and ev[0] == "loc":
            return ev[1].done.pop(ev[2])
        return ev

    def engine_wait(self, engine, ev):
        """order the engine behind a RECEIVED message (`ev` = what recv returned)"""
        if isinstance(self.rx, LocalEdge):
            _wait_generic(engine, self._resolve(ev))
        else:                                      # the ring transport behind rx: only it knows what its receive handles are
            self.rx.engine_wait(engine, ev)

    def engine_wait_sent(self, engine, ev):
        """order the engine behind a SEND's completion (`ev` = what send returned): the message buffer may be rewritten"""
        if isinstance(self.tx, LocalEdge) or self.tx is None:
            _wait_generic(engine, ev)
        else:
            self.tx.engine_wait_sent(engine, ev)

    def engine_wait_ordered(self, engine, evs):
        if isinstance(self.rx, LocalEdge):
            for ev in evs:
                self.engine_wait(engine, ev)
        else:
            self.rx.engine_wait_ordered(engine, evs)

    # ---- send side
    def can_send(self):
        return self.tx.can_send() if self.tx is not None else True

    def send(self, t, after_event=None, slot=0):
        if isinstance(self.tx, LocalEdge):
            return self.tx.send(self.engine, t)
        return self.tx.send(t, after_event, slot)

    def sync_sends(self):
        if self.tx is not None and not isinstance(self.tx, LocalEdge):
            self.tx.sync_sends()

    def finish(self):
        done = set()
        for side in (self.rx, self.tx):
            if side is not None and not isinstance(side, LocalEdge) and id(side) not in done:
                done.add(id(side))
                side.finish()

    def close(self):
        pass


def make_transport(rank, world, device):
    """The transport `run_pipeline` uses by default: none for one rank (in-process hand-off), gloo point-to-point on
    the CPU, the RCCL ring on a GPU."""
    if world == 1:
        return None
    if str(device) == "cpu":
        return GlooTransport(rank, world)
    dev = torch.device(device)
    return RcclTransport(rank, world, dev.index if dev.index is not None else torch.cuda.current_device())


# event-pool slots of the engine (hps_engine_record_event) used by the driver
_EV_SLICE, _EV_STEP, _EV_LFREE, _EV_LOCAL = 0, 64, 80, 4096


_TRACE = [] if os.environ.get("HPS_DRIVE_TRACE") else None      # diagnostic: (seconds, stage, what, slice) per host action


def _trace(stage, what, q):
    if _TRACE is not None:
        _TRACE.append((time.perf_counter(), stage, what, q))


def dump_trace():
    """HPS_DRIVE_TRACE=<file>: write the host-side timeline of the stages (begin / finish of every slice, waits on local
    edges) recorded so far and clear it."""
    if _TRACE:
        with open(os.environ["HPS_DRIVE_TRACE"], "w") as f:
            t0 = _TRACE[0][0]
            for t, st, what, q in _TRACE:
                f.write(f"{(t - t0)*1e6:12.1f} us  stage {st}  {'      '*st}{what} {q}\n")
        del _TRACE[:]


def _drive(gens, engines=None):
    """One host thread, several stages: every generator runs to its next yield in turn ("work": a slice has been enqueued
    and its norm read-back is pending; "wait": a local edge has nothing for it yet).  With `engines` (HPS_DRIVE_READY=1) a
    stage whose slice is pending is only resumed once its norms have arrived (engine.slice_ready()), so that the stage whose
    multigrid finishes first gets its stream refilled first -- measured on MI355X: whole boxes the same (1786 against 1779
    slices/s with three stages), a 20-slice window slower (1546-1673 against 1748-1768: the fixed order keeps the stages
    evenly staggered), so the default is the fixed round-robin order.  Returns the generators' results."""
    live = list(range(len(gens)))
    results = [None] * len(gens)
    tags = [None] * len(gens)
    idle_rounds = 0
    spins = 0
    ring_wait_since = None
    ring_timeout = float(os.environ.get("HPS_RING_TIMEOUT_S", "900") or 900)
    while live:
        worked = False
        stepped = False
        for k in list(live):
            if tags[k] == "work" and engines is not None and len(live) > 1 and not engines[k].slice_ready():
                continue
            try:
                stepped = True
                tags[k] = next(gens[k])
                worked = worked or tags[k] not in ("wait", "wait-ring")
            except StopIteration as stop:
                results[k] = stop.value
                live.remove(k)
                worked = True
        if not stepped:
            # every stage waits for its device norms: poll on; after a long while take the first one (its finish waits and
            # notices a failed launch)
            spins += 1
            if spins < 2000000:
                continue
            k = live[0]
            try:
                tags[k] = next(gens[k])
            except StopIteration as stop:
                results[k] = stop.value
                live.remove(k)
            worked = True
        spins = 0
        # (a stage whose slice is pending on the device is progress to come: only rounds in which every stage waits for a
        #  local edge count towards the deadlock check)
        idle_rounds = 0 if (worked or any(tags[k] in ("work", "wait-ring") for k in live)) else idle_rounds + 1
        if idle_rounds > 1000:
            raise RuntimeError("pipeline stages of this process wait for one another (local edges): deadlock")
        # every stage waits, at least one of them for another process (ipc edge): fine for as long as the neighbour may take,
        # loud when it takes longer than the ring's timeout
        if live and not worked and all(tags[k] in ("wait", "wait-ring") for k in live):
            now = time.perf_counter()
            if ring_wait_since is None:
                ring_wait_since = now
            elif now - ring_wait_since > ring_timeout:
                raise RuntimeError(f"pipeline stages of this process have waited {now - ring_wait_since:.0f} s for a neighbouring rank "
                                   "(HPS_RING_TIMEOUT_S): the ring makes no progress")
        else:
            ring_wait_since = None
    return results


def run_pipeline(*args, **kwargs):
    """One stage per process: `_stage` driven to its end, with Python's cyclic garbage collector kept out of the way: the
    driver allocates small objects per slice (views, tuples, events), and with torch imported one full collection over the
    heap takes 30-50 ms -- measured as one stall of that length in the middle of the first step, with the device running dry
    (the whole difference between the ring and the plain slice loop on one GPU).  What exists now is parked in the permanent
    generation for the run.
    Two or more ranks on RCCL: GPU_MAX_HW_QUEUES >= 8 must be in the environment before the process touches the GPU
    (hipace_amd._lib sets it at import when WORLD_SIZE > 1; RcclTransport refuses a host that initialised the device first)."""
    gc.freeze()                     # (no collection first: that is the 30-50 ms pass this is here to avoid)
    try:
        return _drive([_stage(*args, **kwargs)])[0]
    finally:
        gc.unfreeze()


def run_lanes(engines, rank, world, n_steps, device, on_step_end=None, slices_per_step=None, transport=None,
              laser_lookahead=8, on_slice=None, handoff_batch=1):
    """Several pipeline stages per process: L = len(engines) engines (one stream each) on this rank's device are the stages
    rank*L .. rank*L + L - 1 of a ring of world*L stages; stage v runs the time steps v, v + world*L, ... (Hipace.cpp:400-401
    with several ranks per device).  The edges between the stages of a process are `LocalEdge`s (device copies ordered by
    events), the edge that leaves the process is `transport` (RcclTransport / GlooTransport; none with one process, where the
    last stage hands back to the first).  ONE host thread drives all stages (`_drive`): each engine's slice is enqueued up
    to its Bx/By norm read-back (hps_engine_solve_slice_begin), then the next stage gets its turn, then the read-backs are
    awaited in the same order -- the device always holds L slices' worth of kernels on L streams, and every call into the
    ring comes from this thread.
    on_step_end(step, engine), on_slice(stage_in_process, m, q): as `_stage`'s, with the stage's engine / index added.
    slices_per_step: as `_stage`'s; a list has one entry per STAGE of the whole ring (world*L).
    world > 1: the same requirement on GPU_MAX_HW_QUEUES as `run_pipeline`.
    Returns the number of slices solved by this process."""
    L = len(engines)
    W = world * L
    own = transport is None and world > 1
    ringT = make_transport(rank, world, device) if own else transport
    edges = [LocalEdge() for _ in range(L)]        # edges[j]: stage j -> stage j + 1 of this process (the last one closes the loop when world == 1)
    gens = []
    # one process whose closing edge (last stage -> first stage) goes through the RCCL ring all the same (RcclSelfRing): the
    # configuration of a rank of a multi-rank ring with L stages -- receives of the first stage posted a step ahead as RCCL
    # kernels, sends of the last stage behind its engine's events -- on the one rank a 1-GPU box has
    closing = ringT if (world == 1 and getattr(ringT, "self_ring", False) and not isinstance(ringT, StageTransport)) else None
    for j, eng in enumerate(engines):
        rx = edges[j - 1] if (j > 0 or (world == 1 and closing is None)) else ringT
        tx = edges[j] if (j < L - 1 or (world == 1 and closing is None)) else ringT
        T = StageTransport(eng, rx, tx)
        ose = (lambda step, e=eng: on_step_end(step, e)) if on_step_end is not None else None
        osl = (lambda m, q, jj=j: on_slice(jj, m, q)) if on_slice is not None else None
        gens.append(_stage(eng, rank * L + j, W, n_steps, device, ose, slices_per_step, T, laser_lookahead, osl, handoff_batch))
    gc.freeze()
    try:
        import os
        ready_first = os.environ.get("HPS_DRIVE_READY", "0") == "1" and all(hasattr(e, "slice_ready") for e in engines)
        solved = _drive(gens, engines if ready_first else None)
    finally:
        gc.unfreeze()
    if own and ringT is not None:
        ringT.close()
    return sum(solved)


def _stage(engine, rank, world, n_steps, device, on_step_end=None, slices_per_step=None, transport=None,
           laser_lookahead=8, on_slice=None, handoff_batch=1):
    """Stage `rank` of a ring of `world` stages (a generator, see `_drive`): run steps rank, rank+world, ... < n_steps of
    `engine` with the per-slice hand-off.  Yields "work" between the two halves of a slice (solve_slice_begin /
    solve_slice_finish) and "wait" while a local edge has no message / no posted receive for it yet.

    engine: SliceEngine-like (begin_step, solve_slice, sync, record_event, wait_event, copy_async, beam_layout,
    set_beam_storage, initial_beam_into, ...; the oracle's Engine has the same interface).
    slices_per_step: None = whole steps; an int = only the head slices of every step (benchmark runs shorter than one
    box); a list with one entry per rank (non-increasing, n_steps <= world) lets the ranks stop at different slices --
    the pre-filled pipeline of bench.py.  on_slice(m, q): called before slice q (from the head) of this rank's m-th
    step is solved, and once more with q = number of slices after the last one.
    handoff_batch: 1 = one hand-off per slice, as the reference.  > 1: a static beam (hipace.dt = 0, no laser) is handed on
    in groups of that many slices -- one event on the engine's stream and one wait per group -- and the rank behind runs
    that many slices later (a longer pipeline fill).  Measured on the RCCL ring of one GPU: no difference in the rate
    (the events are not what a hand-off costs), so the default stays 1.  A moving beam and a laser are always handed on
    slice by slice.
    Returns the number of slices this rank solved.
    """
    nz = engine.deck["nz"]
    if slices_per_step is None:
        counts = [nz] * world
    elif isinstance(slices_per_step, int):
        counts = [slices_per_step] * world
    else:
        counts = [int(c) for c in slices_per_step]
        assert len(counts) == world and all(counts[r] >= counts[r + 1] for r in range(world - 1))
        assert len(set(counts)) == 1 or n_steps <= world, "rank-dependent slice counts need a ring that does not close"
    per, per_prev = counts[rank], counts[(rank - 1) % world]
    assert 1 <= per <= nz and per_prev <= nz
    moving = bool(getattr(engine, "moving", False))
    laser = bool(getattr(engine, "has_laser", False))
    assert not (moving and (per != nz or per_prev != nz)), "a moving beam (hipace.dt != 0) needs whole steps"
    my_steps = steps_of_rank(rank, world, n_steps)
    own_transport = transport is None and world > 1
    T = make_transport(rank, world, device) if own_transport else transport
    assert world == 1 or T is not None
    ring = world > 1 or bool(getattr(T, "self_ring", False))      # hand-off through the transport (else: in-process copies)
    # what a stage yields while it waits for a neighbour: "wait" = a stage of this process (local edge; counts towards the
    # driver's deadlock check), "wait-ring" = another process (ipc edge; bounded by HPS_RING_TIMEOUT_S)
    tag_in, tag_out = getattr(T, "wait_in_tag", "wait-ring"), getattr(T, "wait_out_tag", "wait-ring")
    closes = n_steps > world
    f64 = dict(dtype=torch.float64, device=device)

    nbeam, off = engine.beam_layout()
    bufs = rpool = spool = None
    if moving:
        mlen = engine.beam_message_doubles()
        # receive slots: two steps' worth (a rank holds the early slices of its next step); send slots rotate
        rpool = [[torch.zeros(mlen, **f64) for _ in range(nz)] for _ in range(2)]
        spool = [torch.zeros(mlen, **f64) for _ in range(16)] if ring else []
    else:
        bufs = [torch.zeros(max(7 * nbeam, 1), **f64) for _ in range(2)]
        if rank == 0:
            engine.initial_beam_into(bufs[0])        # only the head rank injects the beam (as the reference)
    spool_done = [None] * (len(spool) if spool else 0)
    ring_laser = laser and ring                      # one rank without a transport: the engine rotates its own time levels
    batch = 1 if (moving or ring_laser) else max(1, int(handoff_batch))
    # receives are posted a whole step ahead; an ipc edge's mailbox holds 65536 descriptors (ring.hip: kBoxDescs), two per
    # slice with a laser: boxes beyond 16 k slices post 16 k slices ahead
    lookahead = min(per_prev, 16384)
    lpool, lspool = [], []
    if ring_laser:
        llen = engine.laser_message_doubles()
        if not closes:
            lookahead = max(2, min(laser_lookahead, per_prev))
        lpool = [torch.zeros(llen, **f64) for _ in range(lookahead + 3)]
        lspool = [torch.zeros(llen, **f64) for _ in range(4)]
    lpool_free = [None] * len(lpool)
    lspool_done = [None] * len(lspool)

    def block(buf, q):                                # q-th slice from the head = block q
        return buf[7 * off[q]:7 * off[q + 1]]

    # ---- receives, in the order the previous rank sends: per fed step, per slice: beam message, laser message ----
    fed_steps = [(m, s) for m, s in enumerate(my_steps) if s > 0] if ring else []
    fed_index = {m: f for f, (m, _) in enumerate(fed_steps)}
    n_incoming = len(fed_steps) * per_prev
    recv_ev = {}                                      # (m, j, kind) -> event [, laser pool slot]
    send_done_static = {}                             # (m % 2, j) -> event of the last send out of that block
    step_done = [None, None]
    state = dict(posted=0, nl=0, ns=0, nls=0)

    def post_until(frontier):
        """post the receives of the incoming slices with linear index < frontier"""
        frontier = min(frontier, n_incoming)
        while state["posted"] < frontier:
            f, j = divmod(state["posted"], per_prev)
            m = fed_steps[f][0]
            if moving:
                if j == 0 and step_done[m % 2] is not None:
                    T.recv_after(step_done[m % 2])    # the slots' previous contents (step m-2) have been imported
                recv_ev[(m, j, 0)] = (T.recv(rpool[m % 2][j], None, ((m % 2) * nz + j) * 2), None)
            else:
                blk = block(bufs[m % 2], j)
                if blk.numel() > 0:
                    recv_ev[(m, j, 0)] = (T.recv(blk, send_done_static.pop((m % 2, j), None), ((m % 2) * nz + j) * 2), None)
            if ring_laser:
                k = state["nl"] % len(lpool)
                state["nl"] += 1
                recv_ev[(m, j, 1)] = (T.recv(lpool[k], lpool_free[k], ((m % 2) * nz + j) * 2 + 1), k)
                lpool_free[k] = None
            state["posted"] += 1

    solved = 0
    held = []                                         # static beam blocks waiting for their group's hand-off
    for m, step in enumerate(my_steps):
        fed = step > 0
        from_ring = fed and ring
        if moving:
            engine.set_beam_import(fed)
        else:
            engine.set_beam_storage(bufs[m % 2], injected_beam_support=True)     # hipace.dt = 0: the beam never moves
        if ring_laser:
            # the rank that runs step 0 evaluates the initial envelope; every later step receives a_n, a_{n-1} slice by
            # slice from the rank that ran the step before (MultiBuffer.cpp:840-852, 913-925)
            engine.set_laser_import(fed, step)
        if hasattr(engine, "set_step"):
            engine.set_step(step)                      # the physical step: density profile's time factor, ionisation draws
        engine.begin_step()
        imported = -1
        for q in range(per):
            if on_slice is not None:
                on_slice(m, q)
            islice = nz - 1 - q
            if ring:
                if from_ring:
                    pos = fed_index[m] * per_prev + q
                else:                                  # step 0 of the head rank: the next fed step starts `per - q` slices on
                    nxt = fed_index.get(m + 1)
                    pos = (nxt * per_prev - (per - q)) if nxt is not None else n_incoming
                post_until(pos + lookahead + 1)
            if from_ring:
                need = min(q + 1, per_prev - 1)        # this slice's beam and the next one's (jx/jy source)
                if batch > 1:                          # ... which arrive with the rest of their group
                    need = min((need // batch + 1) * batch - 1, per_prev - 1)
                post_until(fed_index[m] * per_prev + need + 1)
                if batch > 1 and imported < need:
                    evs = [recv_ev.pop((m, j, 0), None) for j in range(imported + 1, need + 1)]
                    for ev in evs:
                        while ev is not None and not T.ready(ev[0]):
                            yield tag_in
                    T.engine_wait_ordered(engine, [ev[0] for ev in evs if ev is not None])
                    imported = need
                while imported < need:
                    imported += 1
                    ev = recv_ev.pop((m, imported, 0), None)
                    if ev is not None:
                        while not T.ready(ev[0]):      # (a local edge: the stage ahead has not sent it yet; an ipc edge: not landed)
                            _trace(rank, "wait-in", q)
                            yield tag_in
                        T.engine_wait(engine, ev[0])
                        if moving:
                            engine.import_beam_slice(nz - 1 - imported, rpool[m % 2][imported])
                    if ring_laser:
                        ev, k = recv_ev.pop((m, imported, 1))
                        while not T.ready(ev):
                            yield tag_in
                        T.engine_wait(engine, ev)
                        engine.import_laser_slice(nz - 1 - imported, lpool[k])
                        lpool_free[k] = engine.record_event(_EV_LFREE + k)
            elif fed and moving and not ring:
                for k in (q, q + 1):                   # in-process hand-off: the blocks were exported by the previous step
                    if k < nz and imported < k:
                        engine.import_beam_slice(nz - 1 - k, rpool[m % 2][k])
                        imported = k
            # the slice in two halves: everything up to the Bx/By solve's norm read-back is enqueued, the other stages of
            # this process (if any) get their turn, then the host waits for the norms and enqueues the rest
            _trace(rank, "begin", q)
            engine.solve_slice_begin(islice)
            _trace(rank, "begun", q)
            yield "work"
            _trace(rank, "finish", q)
            engine.solve_slice_finish(islice)
            _trace(rank, "finished", q)
            solved += 1
            if step + 1 < n_steps:
                if not ring:                           # MultiBuffer.cpp:299-308: send to myself
                    if moving:
                        engine.export_beam_slice(islice, rpool[(m + 1) % 2][q])
                    else:
                        src, dst = block(bufs[m % 2], q), block(bufs[(m + 1) % 2], q)
                        if src.numel() > 0:
                            engine.copy_async(dst, src)
                else:
                    out = []
                    if moving:
                        k = state["ns"] % len(spool)
                        state["ns"] += 1
                        T.engine_wait_sent(engine, spool_done[k])     # the slot's previous message has left
                        engine.export_beam_slice(islice, spool[k])
                        out.append(("b", k, spool[k]))
                    else:
                        blk = block(bufs[m % 2], q)
                        if blk.numel() > 0:
                            held.append(("s", q, blk))
                        if (q + 1) % batch == 0 or q == per - 1:
                            out, held = held, []
                    if ring_laser:
                        k = state["nls"] % len(lspool)
                        state["nls"] += 1
                        T.engine_wait_sent(engine, lspool_done[k])
                        engine.export_laser_slice(islice, lspool[k])
                        out.append(("l", k, lspool[k]))
                    if out:
                        ev = engine.record_event(_EV_SLICE + q % 64)  # the slice (its push, the exports) is done
                        for kind, k, t in out:
                            while not T.can_send():    # (the stage / rank behind has not posted its receive yet, or its buffer is in use)
                                _trace(rank, "wait-out", q)
                                yield tag_out
                            if kind == "b":
                                spool_done[k] = T.send(t, ev, 4 * nz + k)
                            elif kind == "l":
                                lspool_done[k] = T.send(t, ev, 4 * nz + 32 + k)
                            else:
                                send_done_static[(m % 2, k)] = T.send(t, ev, (m % 2) * nz + k)
        if moving:
            step_done[m % 2] = engine.record_event(_EV_STEP + m % 2)
        if on_slice is not None:
            on_slice(m, per)
        if on_step_end is not None:
            on_step_end(step)
    if ring:
        post_until(n_incoming)                         # what the previous rank sends beyond my last slice is still received
        engine.sync()
        T.finish()
        if own_transport:
            T.close()
    return solved


def run_local_pipeline(engines, n_steps, device, on_step_end=None, slices_per_step=None):
    """Several time steps in flight on ONE device: the ring pipeline with all L = len(engines) of its stages in this process
    (`run_lanes` with one process).  Stage j (engine j, one stream each, all on `device`) runs steps j, j+L, ... < n_steps,
    exactly as rank j of `run_pipeline` would; what couples the stages is the same per-slice hand-off -- stage j solves
    slice k of step s only after stage j-1 has pushed slices k and k-1 of step s-1 -- here device-to-device copies ordered
    by stream events (`LocalEdge`; MultiBuffer.cpp:299-308, the reference's in-process "send to myself").  While one step
    sits in a latency-bound phase -- the lower multigrid levels, a DST pass, a launch gap -- the kernels of the others fill
    the device.  One host thread drives all engines (round 2 had one thread per engine).  Returns the number of slices solved."""
    return run_lanes(engines, 0, 1, n_steps, device, on_step_end, slices_per_step)
