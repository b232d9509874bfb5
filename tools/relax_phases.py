"""Profiling helper: device time per phase over one whole relaxation (reference's stop rule) at N particles."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, hostio
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
s = hostio.setup_system("tests/golden/cluster.par", {"ntotal": 2 * n, "mass_ratio": 0.3125})
pos, ids = hostio.sample_gas(s, nthreads=8)
m = hostio.setup_to_model(s)
g = binding.TcGpu(0)
g.set_model(m)
g.phase_times(reset=True)
t0 = time.perf_counter(); g.upload(pos, ids); t1 = time.perf_counter()
log = g.Regularise_sph_particles(); t2 = time.perf_counter()
ph = g.phase_times()
print("upload %.1f ms, relaxation %.1f ms wall, %d iterations" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), len(log)))
tot = sum(v[0] for v in ph.values())
for k, v in sorted(ph.items(), key=lambda kv: -kv[1][0]):
    if v[1]: print("  %-14s %8.2f ms  %5.1f %%  (%d launches, %.3f ms each)" % (k, 1e3 * v[0], 100 * v[0] / tot, v[1], 1e3 * v[0] / v[1]))
print("  device total %.1f ms" % (1e3 * tot))
