"""Candidates / queries per particle of the density pass in the ordered-run mode and in the mirror mode."""
import sys
sys.path.insert(0, ".")
from toycluster_amd import binding, model as M
n = 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=11)
for mode in (0, 2):
    g = binding.TcGpu(0, options={"sweep": mode, "stats": 1})
    g.set_model(m); g.upload(pos, ids)
    g.Regularise_sph_particles(max_iter=3)
    g.density_error()
    print("sweep mode", mode, g.density_stats())
    g.close()
