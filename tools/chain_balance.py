"""Profiling helper: lane utilisation of k_wvt_chain4 -- a quad per particle, 16 consecutive particles per wave pass, every
quad of a pass waits for the longest list: mean / max of the list lengths over groups of 16 consecutive local slots."""
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, model as M
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
g = binding.TcGpu(0)
g.set_model(m); g.upload(pos, ids)
for _ in range(5):
    g.density_error(); g.wvt_step(0.0085, fetch=False)
g.density_error()
cnt = np.zeros(n, np.uint32)
g._L.tcgpu_debug_xlcnt.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
print("rc", g._L.tcgpu_debug_xlcnt(g._h, cnt.ctypes.data_as(C.c_void_p), n))
g.close()
c = cnt.astype(np.int64); c[cnt == 0xffffffff] = 0
steps = (c + 3) // 4
for grp in (16, 64):
    k = n // grp * grp
    s = steps[:k].reshape(-1, grp)
    print("groups of %d consecutive particles: mean steps %.1f, mean of group maxima %.1f -> utilisation %.3f" %
          (grp, s.mean(), s.max(axis=1).mean(), s.mean() / s.max(axis=1).mean()))
o = np.sort(steps)
k = n // 16 * 16
s = o[:k].reshape(-1, 16)
print("if grouped by length (sorted): utilisation %.3f" % (s.mean() / s.max(axis=1).mean()))
