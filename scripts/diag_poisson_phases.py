"""Phases of the 1025-point DST kernel inside a batch of three solves (the engine's Psi / Ez / Bz call), from the shader
clock stamps of four workgroups of the LAST launch (the x pass back): build `make -C hipace_amd/csrc stamps`, run with
HPS_LIB=hipace_amd/csrc/libhpslice_stamps.so.  Stamps: 0 start, 1 rows loaded, 2 pre-processing done, 3 stage A done,
4 stage B done, 5 stored (s_memtime ticks of 10 ns)."""
import ctypes as C
import sys
import time
sys.path.insert(0, '.')
import numpy as np
import torch
from hipace_amd import _lib, api

L = _lib.lib()
L.hps_poisson_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ps = api.FFTPoissonSolver(n, n, 16/n, 16/n)
f = api.Fields(n, n, 2, 6)
staging = torch.randn(3*n*n, dtype=torch.float64, device='cuda')
comps = (C.c_int*3)(0, 1, 2)
st = (C.c_longlong*24)()
L.hps_poisson_debug_stamps(ps._h, st)          # allocates the stamp buffer
def solve():
    e = L.hps_poisson_solve_batch(ps._h, 3, C.c_void_p(staging.data_ptr()), f.struct(), comps, None)
    assert e == 0, e
for _ in range(5):
    solve()
torch.cuda.synchronize()
t = time.time()
for _ in range(50):
    solve()
torch.cuda.synchronize()
print(f"n = {n}: {1e6*(time.time() - t)/50:.1f} us per batch of three solves (host clock, back to back)")
L.hps_poisson_debug_stamps(ps._h, st)
v = np.array(list(st)).reshape(4, 6)
# (the shader clock's zero differs from XCD to XCD: only the differences inside a workgroup mean anything)
print("workgroup   load    pre     stage A stage B store   total  (ticks; share of the workgroup's life)")
for k, name in enumerate(("0", "g/3", "2g/3", "g-1")):
    d = np.diff(v[k]).astype(float)
    tot = d.sum()
    if tot <= 0:
        continue
    print(f"{name:>9} " + " ".join(f"{x:7.0f}" for x in d) + f" {tot:7.0f}   " + " ".join(f"{100*x/tot:4.1f}%" for x in d))
