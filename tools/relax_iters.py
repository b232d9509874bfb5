"""Profiling helper: per-iteration phase times and neighbour-list coverage over the first iterations of a relaxation
from a cold start (hsml = 0) -- where the unlisted particles of the ordered gather are, iteration by iteration."""
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, model as M
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
g = binding.TcGpu(0, options={"timing": 1})
g.set_model(m); g.upload(pos, ids)
g._L.tcgpu_debug_xlist_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
out = (C.c_double * 7)()
tot = 0.0
for it in range(iters):
    g.phase_times(reset=True)
    em, ex = g.density_error()
    listed = float("nan")
    if g._L.tcgpu_debug_xlist_stats(g._h, out) == 0 and out[0] > 0:
        listed = out[2] / out[0]
    g.wvt_step(0.0085, fetch=False)
    ph = g.phase_times()
    ms = {k: 1e3 * v[0] for k, v in ph.items() if v[1]}
    s = sum(ms.values()); tot += s
    print("it %2d err_mean %.4f  listed %.4f  device %.2f ms: density %.2f  sweep %.2f  query_records %.2f  rest %.2f" %
          (it, em, listed, s, ms.get("density", 0), ms.get("wvt_sweep", 0), ms.get("query_records", 0),
           s - ms.get("density", 0) - ms.get("wvt_sweep", 0) - ms.get("query_records", 0)), flush=True)
    if it == 0:
        print("      cold pass, every phase:", {k: round(v, 2) for k, v in ms.items()}, flush=True)
print("device total %.1f ms over %d iterations" % (tot, iters))
g.close()
