"""Profiling helper: how the sweep's ball (hsml_wvt * box) compares with the density balls (hsml, 1.23 hsml) on a warm state."""
import sys
sys.path.insert(0, '.')
import numpy as np
from toycluster_amd import binding, model as M
n = 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
g = binding.TcGpu(0)
g.set_model(m); g.upload(pos, ids)
for _ in range(6):
    g.density_error(); g.wvt_step(0.0085, fetch=False)
g.density_error()
hs, de = g.wvt_step(0.0085, move=False, fetch=True)
p = g.particles()
box = float(m["boxsize"]) if isinstance(m, dict) else float(m.boxsize)
r = hs * box / p["hsml"]
print("hw / h0 percentiles 1 5 25 50 75 95 99:", np.percentile(r, [1, 5, 25, 50, 75, 95, 99]).round(3))
print("fraction hw > 1.23 h0: %.3f   hw < h0: %.3f   hw < 0.9 h0: %.3f" % ((r > 1.23).mean(), (r < 1).mean(), (r < 0.9).mean()))
v = (np.minimum(r, 1.23) / 1.23) ** 3
print("mean (sweep hits / density hits) ~ (min(hw, hb)/hb)^3: %.3f;  mean (max(hw,hb)/hb)^3: %.3f" % (v.mean(), ((np.maximum(r, 1.23) / 1.23) ** 3).mean()))
