"""Profiling helper (not part of the product; needs `make -C toycluster_amd/csrc ablate`): life span of every
wave of a warm k_iter launch -- how well the static particle assignment balances the persistent grid."""
import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np
from toycluster_amd import binding, model as M
binding.LIB_PATH = os.path.join(os.path.dirname(binding.LIB_PATH), "libtcgpu_ablate.so")
n = 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
g = binding.TcGpu(0)
g.set_model(m); g.upload(pos, ids)
for _ in range(4):
    g.density_error(); g.wvt_step(0.0085, fetch=False)
g.density_error()
nw = 4096
buf = np.zeros(2 * nw, np.uint64)
L = C.CDLL(binding.LIB_PATH)
L.tcgpu_debug_wave_spans(buf.ctypes.data_as(C.c_void_p), nw)
t0, t1 = buf[0::2].astype(np.int64), buf[1::2].astype(np.int64)
ok = t1 > 0
t0, t1 = t0[ok], t1[ok]
start = t0.min()
end = (t1 - start) / 100.0          # microseconds (100 MHz)
print("waves", ok.sum(), "kernel span us %.0f" % end.max())
print("wave end time percentiles us:", np.percentile(end, [0, 5, 25, 50, 75, 95, 100]).round(0))
print("mean wave life / kernel span = %.3f" % (((t1 - t0) / 100.0).mean() / end.max()))
blk = np.arange(len(end)) // 4
for grp in range(8):
    sel = (blk % 8) == grp
    print("XCD group", grp, "end us: min %.0f mean %.0f max %.0f" % (end[sel].min(), end[sel].mean(), end[sel].max()))
