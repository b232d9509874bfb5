"""Driver for profiling the exact sweep: set-up at config-2 size, then a few launches.  usage: xsweep_run.py [n] [ablate]"""
import sys
sys.path.insert(0, ".")
from toycluster_amd import binding, model as M
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
ab = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=11)
g = binding.TcGpu(0)
g.set_model(m); g.upload(pos, ids)
g.Regularise_sph_particles(max_iter=3)
g.set_option("ablate", ab)
for _ in range(2):
    g.wvt_step(0.0085, move=False, fetch=False)
g.close()
