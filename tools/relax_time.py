import sys, time
sys.path.insert(0, '.')
import numpy as np
from toycluster_amd import binding, model as M
n = 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
for rep in range(2):
    g = binding.TcGpu(0)
    g.set_model(m); g.upload(pos, ids)
    t0 = time.perf_counter(); log = g.Regularise_sph_particles(); t = time.perf_counter() - t0
    print("iterations", len(log), "seconds %.4f" % t, "errmean %.8f" % log[-1]["err_mean"], flush=True)
    g.close()
