"""Timing of the exact sweep kernel and its phases (ablate: 1 = A1 only, 2 = A1 + A2) at config-2 size."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, model as M
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=11)
g = binding.TcGpu(0, options={"timing": 1})
g.set_model(m); g.upload(pos, ids)
g.Regularise_sph_particles(max_iter=4)
for shift in (0, -1, 1):
    for ab in (0, 1, 2):
        g.set_option("ablate", ab); g.set_option("xsweep_shift", shift)
        g.phase_times(reset=True)
        for _ in range(3):
            g.wvt_step(0.0085, move=False, fetch=False)
        ph = g.phase_times(reset=True)
        print("xsweep_shift %2d ablate %d: wvt_sweep %.3f ms" % (shift, ab, ph["wvt_sweep"][0] / ph["wvt_sweep"][1] * 1e3), flush=True)
g.close()
