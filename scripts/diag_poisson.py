import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import oracle as O
from hipace_amd import api
rng=np.random.default_rng(0)
for n in (255, 256, 512, 1023, 1024):
    rhs=rng.standard_normal((n,n))
    dx=16/n
    ref=O.poisson_solve(rhs, dx, dx)
    ps=api.FFTPoissonSolver(n,n,dx,dx)
    ps.StagingArea().copy_(torch.as_tensor(rhs))
    f=api.Fields(n,n,2,1)
    ps.SolvePoissonEquation(f,0)
    torch.cuda.synchronize()
    t=time.time()
    for _ in range(20): ps.SolvePoissonEquation(f,0)
    torch.cuda.synchronize(); dt=(time.time()-t)/20
    out=f.numpy()[0,2:-2,2:-2]
    print(n, 'rel err vs oracle', np.abs(out-ref).max()/np.abs(ref).max(), 'ms/solve', dt*1e3)
import ctypes as C
from hipace_amd import _lib
L=_lib.lib()
L.hps_poisson_debug_stamps.argtypes=[C.c_void_p, C.c_void_p]
n=1024
ps=api.FFTPoissonSolver(n,n,16/n,16/n)
st=(C.c_longlong*6)()
L.hps_poisson_debug_stamps(ps._h, st)
f=api.Fields(n,n,2,1)
ps.SolvePoissonEquation(f,0); torch.cuda.synchronize()
L.hps_poisson_debug_stamps(ps._h, st)
v=list(st); print('stamps (last pass) deltas:', [v[i+1]-v[i] for i in range(5)])
