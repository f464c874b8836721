import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipace_amd import api, decks
for n, nz in ((128, 256), (512, 256), (1024, 256), (1024, 2048)):
    d = decks.synthetic(n, nz, 2)
    d.update(beam_profile=-1, lo=(-20.0, -20.0, -15.0), hi=(20.0, 20.0, 6.0), laser_on=1, laser_a0=4.5, laser_w0=4.0,
             laser_L0=2.0, laser_lambda0=0.08, laser_solver=1, dt=5.0)
    decks.with_ion_species(d, "N", 0.2, ppc=(1, 1), initial_level=0, seed=5)
    d["background_density_SI"] = 2.8239587008591567e23
    e = api.SliceEngine(d, tile_size=16)
    e.begin_step()
    stop = int(nz * (1.0 - 8.0 / 21.0))          # down to z = -2: behind the pulse
    for k in range(nz - 1, stop, -1):
        e.solve_slice(k)
    s = e.slab()
    names = e.comp_names()
    mx = {nm: float(np.abs(s[i]).max()) for i, nm in enumerate(names) if nm in ("aabs", "Ez", "ExmBy", "Psi", "chi", "By")}
    print(n, nz, "slices", nz - 1 - stop, "ion_stats", e.ion_stats(), "fallbacks", e.fallbacks(), mx, flush=True)
    del e
