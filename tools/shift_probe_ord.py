"""Profiling helper (not part of the product): the default iteration (ordered gather) against the cell-size knobs
(level_shift, level_scale): finer leaves = fewer candidates to test, more runs to walk and stream."""
import sys
sys.path.insert(0, '.')
import numpy as np
from toycluster_amd import binding, model as M
n = 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
for shift, scale in ((1, 2 ** 0.25), (1, 1.3), (1, 2 ** 0.5), (1, 1.54), (1, 2 ** 0.25), (1, 2 ** 0.5)):
    g = binding.TcGpu(0)
    g.set_option("timing", 1)
    g.set_option("level_shift", shift); g.set_option("level_scale", scale)
    g.set_model(m); g.upload(pos, ids)
    for _ in range(3):
        g.density_error(); g.wvt_step(0.0085, fetch=False)
    g.phase_times(reset=True)
    for _ in range(4):
        g.density_error(); g.wvt_step(0.0085, fetch=False)
    t = g.phase_times()
    keys = ("density", "query_records", "wvt_sweep", "cell_index")
    print("shift %d scale %.3f" % (shift, scale), {k: round(1e3 * v[0] / max(1, v[1]), 3) for k, v in t.items() if k in keys},
          "all %.3f" % sum(1e3 * v[0] / 4 for v in t.values()), flush=True)
    g.close()
