"""Host and device cost of one hand-off through the RCCL self ring (hps_ring_sendrecv_self), by message size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipace_amd.pipeline import RcclSelfRing
T = RcclSelfRing(0)
for n in (8, 131072, 1 << 20, 4 << 20):
    a = torch.ones(n, dtype=torch.float64, device="cuda"); b = torch.zeros(n, dtype=torch.float64, device="cuda")
    for _ in range(5):
        T.sendrecv_self(a, b, None, 0)
    T.finish(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        T.sendrecv_self(a, b, None, i % 8)
    t1 = time.perf_counter()
    T.finish()
    t2 = time.perf_counter()
    print(f"{n*8/1e6:8.3f} MB: host {1e6*(t1-t0)/200:7.1f} us per call, total {1e6*(t2-t0)/200:7.1f} us per message", flush=True)

# does the RCCL enqueue block the host while the ring's stream waits for an event of a busy stream?
busy = torch.cuda.Stream()
big = torch.ones(64 << 20, dtype=torch.float64, device="cuda")
a = torch.ones(1 << 17, dtype=torch.float64, device="cuda"); b = torch.zeros(1 << 17, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
with torch.cuda.stream(busy):
    t0 = time.perf_counter()
    for _ in range(40):
        big.mul_(1.0000001)          # ~40 x 0.25 ms of work on the busy stream
    ev = torch.cuda.Event()
    ev.record(busy)
t1 = time.perf_counter()
done = T.sendrecv_self(a, b, ev.cuda_event, 0)      # waits (on the device) for ev
t2 = time.perf_counter()
torch.cuda.synchronize(); T.finish()
t3 = time.perf_counter()
print(f"enqueue busy work {1e3*(t1-t0):.2f} ms; RCCL call behind a pending event: host {1e3*(t2-t1):.3f} ms; all done after {1e3*(t3-t0):.2f} ms", flush=True)
