"""Profiling helper: time of k_curl at N particles (2-cluster merger) after a warm density pass."""
import sys
sys.path.insert(0, '.')
import numpy as np
from toycluster_amd import binding, hostio
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
s = hostio.setup_system("tests/golden/cluster.par", {"ntotal": 2 * n, "mass_ratio": 0.3125})
pos, ids = hostio.sample_gas(s, nthreads=8)
m = hostio.setup_to_model(s)
g = binding.TcGpu(0)
g.set_model(m); g.upload(pos, ids)
g.Regularise_sph_particles(max_iter=2)
g.Find_sph_quantities()
rm = g.Global_density_model().astype(np.float64)
a = ((rm / max(h.rho0 for h in m.halos)) ** 0.5).astype(np.float32)
apot = np.repeat(a[:, None], 3, axis=1)
g.Bfld_from_rotA_SPH(apot)
for ab in (0, 0):
    g.phase_times(reset=True)
    for _ in range(3):
        b = g.Bfld_from_rotA_SPH(apot)
    t = g.phase_times()
    print("n=%d %s k_curl %.3f ms  |B| max %.3e finite %s" % (n, "old" if ab else "new", 1e3 * t["curl"][0] / t["curl"][1], np.abs(b).max(), np.isfinite(b).all()), flush=True)
