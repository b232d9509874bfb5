"""Profiling helper (not part of the product): which cell-table levels the queries of a relaxed
2e6-particle merger use, candidates per particle, and the step time with the deepest level capped."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from toycluster_amd import binding, model as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
for lmax in (0,):
    g = binding.TcGpu(0)
    g.set_option("timing", 1)
    if lmax:
        g.set_option("lmax", lmax)
    g.set_model(m); g.upload(pos, ids)
    for _ in range(4):
        g.density_error(); g.wvt_step(0.0085, fetch=False)
    g.phase_times(reset=True)
    t0 = time.time()
    for _ in range(5):
        g.density_error(); g.wvt_step(0.0085, fetch=False)
    g.density_error()
    dt = (time.time() - t0) / 5
    t = g.phase_times()
    print("lmax", lmax or "default", "ms/step %.2f" % (1e3 * dt),
          {k: round(1e3 * v[0] / max(1, v[1]), 3) for k, v in t.items() if v[1]})
    if not lmax:
        p = g.particles()
        h = p["hsml"].astype(np.float64)
        R = 1.23 * h
        L = np.floor(np.log2(m.boxsize / R)).astype(int) + 1 + 1
        print("hsml/box min %.5f median %.5f max %.4f" % (h.min() / m.boxsize, np.median(h) / m.boxsize, h.max() / m.boxsize))
        print("level histogram (unclamped):", dict(zip(*np.unique(L, return_counts=True))))
        Lc = np.minimum(L, 9)
        edge = m.boxsize / 2.0 ** Lc
        ext = R * (1 + 1e-5) + m.boxsize * 1.2e-5 + edge
        P = p["pos"].astype(np.float64)
        wrap = ~((P >= ext[:, None]) & (P <= m.boxsize - ext[:, None])).all(axis=1)
        print("wrap fraction %.4f, level>8 fraction %.4f, either %.4f" % (wrap.mean(), (L > 8).mean(), (wrap | (L > 8)).mean()))
    g.close()
