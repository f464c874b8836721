"""A/B of the y direction of the Poisson solve: tridiagonal solves (HPS_POISSON_TRIDIAG=1, round 6) against the two DST passes;
error against the oracle and time per batch of three solves (the slice's Psi, Ez, Bz)."""
import os, sys, time; sys.path.insert(0, '.')
import ctypes as C
import numpy as np, torch
from oracle import oracle as O
from hipace_amd import api, _lib
rng = np.random.default_rng(0)
sizes = [(64, 64), (256, 256), (255, 255), (511, 511), (512, 512), (1023, 1023), (1024, 1024), (1024, 512), (96, 48), (2047, 2047)]
if len(sys.argv) > 1: sizes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for nx, ny in sizes:
    rhs = rng.standard_normal((3, ny, nx))
    dx, dy = 16/nx, 12/ny
    ref = [O.poisson_solve(rhs[b], dx, dy) for b in range(3)] if nx*ny <= 1100*1100 else None
    for tri in ("0", "1"):
        os.environ["HPS_POISSON_TRIDIAG"] = tri
        ps = api.FFTPoissonSolver(nx, ny, dx, dy)
        st = torch.as_tensor(rhs).cuda().contiguous()
        f = api.Fields(nx, ny, 2, 3)
        comps = (C.c_int*3)(0, 1, 2)
        L = _lib.lib()
        def go(): _lib.check(L.hps_poisson_solve_batch(ps._h, 3, C.c_void_p(st.data_ptr()), f.struct(), comps, None))
        go(); torch.cuda.synchronize()
        out = f.numpy()[:, 2:-2, 2:-2].copy()
        for _ in range(20): go()
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(50): go()
        torch.cuda.synchronize(); dt = (time.time() - t)/50
        err = max(np.abs(out[b] - ref[b]).max()/np.abs(ref[b]).max() for b in range(3)) if ref else float('nan')
        if tri == "0": base = out
        else: err2 = np.abs(out - base).max()/np.abs(base).max()
        print(f"{nx}x{ny} tridiag={tri}: rel err vs oracle {err:.2e}  us/batch {dt*1e6:.1f}" + (f"  vs DST path {err2:.2e}" if tri == "1" else ""), flush=True)
