"""Driver: a few warm WVT iterations at config-2 size with a given sweep mode (0 lists from k_iter, 2 stand-alone exact kernel,
1 round 2's fused f64 sweep).  usage: iter_modes.py mode [n]"""
import sys
sys.path.insert(0, ".")
from toycluster_amd import binding, model as M
mode = int(sys.argv[1]); n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=11)
g = binding.TcGpu(0, options={"sweep": mode})
g.set_model(m); g.upload(pos, ids)
g.Regularise_sph_particles(max_iter=4)
g.close()
