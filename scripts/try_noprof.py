import sys, time
sys.path.insert(0,'/root/repo')
import torch
from hipace_amd import api, decks
nz=1024; deck=decks.synthetic(1024,nz,2)
eng=api.SliceEngine(deck, device=0, tile_size=16, sort_period=128)
def run(count):
    eng.begin_step()
    for k in range(count): eng.solve_slice(nz-1-k)
run(64)
for prof in (False, True, False, True):
    eng.set_profiling(prof); eng.sync(); torch.cuda.synchronize()
    t0=time.perf_counter(); run(1024); eng.sync(); dt=time.perf_counter()-t0
    if prof: eng.phase_times()
    print('profiling', prof, round(1024/dt,1))
