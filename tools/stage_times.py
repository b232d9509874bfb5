"""Profiling helper (not part of the product; needs `make -C toycluster_amd/csrc stages`): where the waves of a warm
k_iter launch spend their shader cycles, stage by stage (s_memtime accumulators compiled into libtcgpu_stages.so).
The four waves of a SIMD interleave, so a stage's share of the waves' lives is its share of the launch."""
import ctypes as C, json, os, sys
sys.path.insert(0, '.')
import numpy as np
from toycluster_amd import binding, model as M
binding.LIB_PATH = os.path.join(os.path.dirname(binding.LIB_PATH), os.environ.get("TC_STAGES_LIB", "libtcgpu_stages.so"))
NAMES = ["prologue", "producer", "window+loads", "candidate test+staging", "convert_d (f64 r)", "convert_w (sweep pairs)",
         "solver pair loop", "solver uniform part", "epilogue", "work queue"]
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
g = binding.TcGpu(0)
g.set_model(m); g.upload(pos, ids)
for _ in range(4):
    g.density_error(); g.wvt_step(0.0085, fetch=False)
g.phase_times(reset=True)
g.density_error()
t = g.phase_times()
ms = 1e3 * t["density"][0] / max(1, t["density"][1])
nw = 4096
ns = len(NAMES)
buf = np.zeros(ns * nw, np.uint32)
L = C.CDLL(binding.LIB_PATH)
L.tcgpu_debug_stage_cycles(buf.ctypes.data_as(C.c_void_p), nw)
a = buf.reshape(nw, ns).astype(np.float64)
a = a[a.sum(1) > 0]
tot = a.sum()
out = {"k_iter_ms_with_timers": ms, "waves": int(len(a)), "mean_wave_cycles": tot / len(a), "stages": {}}
print("k_iter %.2f ms with timers, %d waves, mean wave life %.3g cycles" % (ms, len(a), tot / len(a)))
for s, nm in enumerate(NAMES):
    fr = a[:, s].sum() / tot
    out["stages"][nm] = {"frac": fr, "cycles_per_particle": a[:, s].sum() / n}
    print("  %-26s %5.1f %%   %8.0f wave-cycles per particle" % (nm, 100 * fr, a[:, s].sum() / n))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
