"""Profiling helper: warm WVT iteration at N particles on one GPU (native C sampler), phase by phase, plus the curl."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, hostio
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 16_000_000
s = hostio.setup_system("tests/golden/cluster.par", {"ntotal": 2 * n, "mass_ratio": 0.3125})
t0 = time.perf_counter()
pos, ids = hostio.sample_gas(s, nthreads=16)
m = hostio.setup_to_model(s)
print("sampled %d particles in %.1f s" % (len(ids), time.perf_counter() - t0), flush=True)
g = binding.TcGpu(0)
g.set_model(m); g.upload(pos, ids)
for _ in range(3):
    g.density_error(); g.wvt_step(0.0085, fetch=False)
g.phase_times(reset=True)
t0 = time.perf_counter()
for _ in range(3):
    e = g.density_error(); g.wvt_step(0.0085, fetch=False)
wall = (time.perf_counter() - t0) / 3
t = g.phase_times()
print("n=%d: %.1f ms per warm iteration (wall), err_mean %.5f" % (n, 1e3 * wall, e[0]))
print("  ", {k: round(1e3 * v[0] / max(1, v[1]), 3) for k, v in t.items() if v[1]}, flush=True)
print("   k_iter %.2f ns per particle" % (1e9 * t["density"][0] / t["density"][1] / n))
g.Find_sph_quantities()
rm = g.Global_density_model().astype(np.float64)
a = ((rm / rm.max()) ** 0.5).astype(np.float32)
del rm
apot = np.repeat(a[:, None], 3, axis=1); del a
g.Bfld_from_rotA_SPH(apot)
g.phase_times(reset=True)
g.Bfld_from_rotA_SPH(apot)
t = g.phase_times()
print("   k_curl %.1f ms" % (1e3 * t["curl"][0] / t["curl"][1]), flush=True)
