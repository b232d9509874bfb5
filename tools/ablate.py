"""Profiling helper (not part of the product): time the density kernel with stages disabled."""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from toycluster_amd import binding, model as M
n = 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
g = binding.TcGpu(0)
g.set_model(m); g.upload(pos, ids)
g.density_error(); g.wvt_step(0.0085, fetch=False)
g.set_option("stats", 1)
g.Find_sph_quantities()
print("stats", g.density_stats())
g.set_option("stats", 0)
p = g.particles()
g.close()
for lmax, shift in ((9, 0), (9, 1)):
    g = binding.TcGpu(0, options={"lmax": lmax, "level_shift": shift})
    g.set_model(m)
    for ab in (0, 3, 2, 1):
        g.set_option("ablate", ab)
        g.upload(p["pos"], p["id"], p["hsml"])
        g.phase_times(reset=True)
        for _ in range(3):
            g.Find_sph_quantities()
        t = g.phase_times()
        print("lmax", lmax, "shift", shift, "ablate", ab, "density ms %.2f" % (1e3 * t["density"][0] / t["density"][1]),
              "cells ms %.2f" % (1e3 * t["cell_index"][0] / t["cell_index"][1]))
    g.close()
