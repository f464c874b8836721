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
