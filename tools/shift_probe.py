"""Profiling helper (not part of the product): k_iter time for different cell-level shifts."""
import sys
sys.path.insert(0, '.')
import numpy as np
from toycluster_amd import binding, model as M
n = 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
for lmax in (0, 8, 0, 8):
    g = binding.TcGpu(0)
    g.set_option("timing", 1)
    if lmax: g.set_option("lmax", lmax)
    g.set_model(m); g.upload(pos, ids)
    for _ in range(3):
        g.density_error(); g.wvt_step(0.0085, fetch=False)
    g.phase_times(reset=True)
    for _ in range(4):
        g.density_error(); g.wvt_step(0.0085, fetch=False)
    t = g.phase_times()
    print("lmax", lmax or "default", {k: round(1e3 * v[0] / max(1, v[1]), 3) for k, v in t.items() if k in ("density", "mirror", "cell_index")}, "sum %.3f" % sum(1e3 * v[0] / max(1, v[1]) for k, v in t.items() if k in ("density", "mirror", "cell_index")), flush=True)
    g.close()
