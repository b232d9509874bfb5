"""How the ordered gather of k_iter went: particles with run lists / neighbour lists, list lengths."""
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, model as M
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=11)
g = binding.TcGpu(0, options={"timing": 1})
g.set_model(m); g.upload(pos, ids)
g.Regularise_sph_particles(max_iter=4)
out = (C.c_double * 7)()
g._L.tcgpu_debug_xlist_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
print("rc", g._L.tcgpu_debug_xlist_stats(g._h, out))
print("own %d, with runs %.4f, with list %.4f, mean runs %.1f (max %d), mean listed %.1f (max %d)" %
      (out[0], out[1] / out[0], out[2] / out[0], out[3], out[5], out[4], out[6]))
g.phase_times(reset=True)
g.Regularise_sph_particles(max_iter=3)
ph = g.phase_times()
print({k: round(v[0] / max(v[1], 1) * 1e3, 3) for k, v in ph.items() if v[1]})
g.close()
