"""Profiling helper (not part of the product): k_iter with stages disabled (results invalid).
Run under `rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace` to get instruction counts per stage."""
import sys
sys.path.insert(0, '.')
import numpy as np
import os
from toycluster_amd import binding, model as M
# the hooks live in a separate build: make -C toycluster_amd/csrc ablate
binding.LIB_PATH = os.path.join(os.path.dirname(binding.LIB_PATH), "libtcgpu_ablate.so")
n = 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
g = binding.TcGpu(0)
g.set_model(m); g.upload(pos, ids)
for _ in range(3):
    g.density_error(); g.wvt_step(0.0085, fetch=False)
p = g.particles()
for ab in (0, 4, 3, 2, 1):
    g.set_option("ablate", ab)
    g.upload(p["pos"], p["id"], p["hsml"])
    g.phase_times(reset=True)
    for _ in range(2):
        try:
            g.density_error()
        except Exception as e:
            pass
    t = g.phase_times()
    print("ablate", ab, "k_iter ms %.2f" % (1e3 * t["density"][0] / t["density"][1]))
g.set_option("ablate", 0)
