"""Profiling helper: k_iter time against the number of co-resident blocks per CU (= waves per SIMD), option "blocks_per_cu"."""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from toycluster_amd import binding, model as M
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
g = binding.TcGpu(0)
g.set_model(m); g.upload(pos, ids)
for _ in range(4):
    g.density_error(); g.wvt_step(0.0085, fetch=False)
for b in (4, 3, 2, 1, 4):
    g.set_option("blocks_per_cu", b)
    g.density_error()
    g.phase_times(reset=True)
    g.density_error()
    t = g.phase_times()
    print("blocks per CU %d: k_iter %.2f ms" % (b, 1e3 * t["density"][0] / max(1, t["density"][1])), flush=True)
