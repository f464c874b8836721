"""Can two ranks share the one GPU of the box?  (RCCL normally refuses duplicate devices.)  Process group on gloo (only the
bootstrap), the ring of hps_ring_* on RCCL with both ranks on device 0."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from hipace_amd.pipeline import RcclTransport
try:
    T = RcclTransport(rank, world, 0)
    a = torch.full((1000,), float(rank + 1), dtype=torch.float64, device="cuda")
    b = torch.zeros(1000, dtype=torch.float64, device="cuda")
    ev_r = T.recv(b, None, 0)
    ev_s = T.send(a, None, 0)
    T.finish()
    torch.cuda.synchronize()
    print("rank", rank, "received", b[0].item(), "stats", T.stats(), flush=True)
    T.close()
except Exception as e:      # noqa: BLE001
    print("rank", rank, "ring failed:", repr(e)[:300], flush=True)
dist.barrier()
dist.destroy_process_group()
