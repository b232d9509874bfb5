"""Profiling helper: work counters of the COLD density pass (hsml = 0, first-pass guess) against a warm one."""
import sys
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, model as M
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
g = binding.TcGpu(0, options={"timing": 1, "stats": 1})
g.set_model(m); g.upload(pos, ids)
for it in range(3):
    g.phase_times(reset=True)
    g.density_error()
    st = g.density_stats()
    ph = g.phase_times()
    print("pass %d (%s): density %.2f ms" % (it, "cold" if it == 0 else "warm", 1e3 * ph["density"][0]), st, flush=True)
    g.wvt_step(0.0085, fetch=False)
g.close()
