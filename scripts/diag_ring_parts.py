"""Which part of the ring hand-off costs the engine time?  run_pipeline over whole boxes with transports that leave
parts out: nothing at all (the driver's own Python), engine events only, copies on a second stream only, everything."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipace_amd import api, decks, pipeline


class Parts:
    self_ring = True

    def __init__(self, events, copies):
        self.events, self.copies = events, copies
        self.st = torch.cuda.Stream(priority=-1)
        self.posted = []
        self.done = {}
        self.n = 0

    def recv(self, t, after_event=None, slot=0):
        self.posted.append(t)
        self.n += 1
        return ("self", self.n - 1)

    def send(self, t, after_event=None, slot=0):
        dst = self.posted.pop(0)
        k = self.n_sent = getattr(self, "n_sent", -1) + 1
        if self.copies:
            with torch.cuda.stream(self.st):
                dst.copy_(t, non_blocking=True)
        self.done[k] = None
        return None

    def engine_wait(self, engine, ev):
        if isinstance(ev, tuple):
            self.done.pop(ev[1], None)

    def engine_wait_ordered(self, engine, evs):
        for ev in evs:
            self.engine_wait(engine, ev)

    def recv_after(self, ev): pass
    def sync_sends(self): pass
    def finish(self): torch.cuda.synchronize()
    def close(self): pass


deck = decks.synthetic(1024, 1024, 2)
dev = torch.device("cuda", 0)
for name, mk in (("plain loop", None), ("driver only", lambda: Parts(False, False)), ("driver + copies", lambda: Parts(False, True)),
                 ("rccl self ring", lambda: pipeline.RcclSelfRing(0)), ("plain loop", None)):
    eng = api.SliceEngine(deck, tile_size=16, sort_period=128)
    eng.begin_step()
    for k in range(64): eng.solve_slice(1023 - k)
    eng.sync()
    T = mk() if mk else None
    if T is not None and not isinstance(T, Parts):
        pass
    rec = eng.record_event
    if isinstance(T, Parts) and not T.events:
        eng.record_event = lambda slot: None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if T is None:
        n = 0
        for s in range(2):
            eng.begin_step()
            for k in range(1024): eng.solve_slice(1023 - k); n += 1
    else:
        n = pipeline.run_pipeline(eng, 0, 1, 2, dev, transport=T, handoff_batch=int(os.environ.get("BATCH", "1")))
    eng.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:18s} {n/dt:8.1f} slices/s  {1e3*dt/n:.4f} ms/slice", flush=True)
    del eng, T
