"""Profiling helper (not part of the product): per-particle cost of k_iter at 1.6e7 particles on one GPU
for different deepest table levels; candidates per particle; level histogram."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from toycluster_amd import binding, model as M
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 16_000_000
lmaxes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 10]
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
print("sampled", flush=True)
for lmax in lmaxes:
    g = binding.TcGpu(0)
    g.set_option("timing", 1)
    g.set_option("stats", 1)
    if lmax:
        g.set_option("lmax", lmax)
    g.set_model(m); g.upload(pos, ids)
    for _ in range(3):
        g.density_error(); g.wvt_step(0.0085, fetch=False)
    g.phase_times(reset=True)
    for _ in range(3):
        g.density_error(); g.wvt_step(0.0085, fetch=False)
    t = g.phase_times()
    st = g.density_stats()
    print("lmax", lmax or "default", {k: round(1e3 * v[0] / max(1, v[1]), 3) for k, v in t.items() if v[1]}, st, flush=True)
    if lmax == lmaxes[0]:
        p = g.particles()
        h = p["hsml"].astype(np.float64)
        L = np.floor(np.log2(m.boxsize / (1.23 * h))).astype(int) + 2
        print("hsml/box min %.5f median %.5f" % (h.min() / m.boxsize, np.median(h) / m.boxsize))
        print("level histogram (unclamped):", dict(zip(*np.unique(L, return_counts=True))), flush=True)
        del p
    g.close()
