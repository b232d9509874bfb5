"""Profiling helper: work counters and timings of the fused vs plain density path."""
import sys
sys.path.insert(0, '.')
import numpy as np
from toycluster_amd import binding, model as M
n = 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
for fuse, shift in ((0, 0), (1, 0), (1, 1), (0, 1)):
    g = binding.TcGpu(0, options={"fuse": fuse, "level_shift": shift})
    g.set_model(m); g.upload(pos, ids)
    for _ in range(2):
        g.density_error(); g.wvt_step(0.0085, fetch=False)
    g.set_option("stats", 1)
    g.phase_times(reset=True)
    for _ in range(3):
        e = g.density_error(); g.wvt_step(0.0085, fetch=False)
    t = g.phase_times()
    g.Find_sph_quantities()
    print("fuse", fuse, "shift", shift, "err %.6g" % e[0], "stats", {k: round(v, 1) for k, v in g.density_stats().items()},
          "density ms %.2f wvt ms %.2f" % (1e3 * t["density"][0] / 3, 1e3 * t["wvt_sweep"][0] / 3))
    g.close()
