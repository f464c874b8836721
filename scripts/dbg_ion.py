import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipace_amd import api, decks
for n, nz in ((64, 60), (256, 240), (512, 480)):
    d = decks.laser_blowout_wake()
    d.update(nx=n, ny=n, nz=nz, laser_lambda0=0.08, plasma_ppc=(2, 2))
    decks.with_ion_species(d, "N", 0.2, ppc=(1, 1), initial_level=0, seed=5)
    d["background_density_SI"] = 2.8239587008591567e23
    e = api.SliceEngine(d, tile_size=16)
    e.begin_step()
    for k in range(nz - 1, -1, -1):
        e.solve_slice(k)
    print(n, nz, "ion_stats", e.ion_stats(), "levels", np.bincount(e.ions()[2], minlength=4), flush=True)
