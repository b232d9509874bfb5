import sys
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, model as M
from oracle import oracle as O
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(want + 1):
    n = int(rng.integers(2000, 26000)); iters = int(rng.integers(1, 5))
    name = "merger" if rng.random() < 0.7 else "single"
    m = M.preset(name, n)
    if rng.random() < 0.3:
        m = M.with_subhalos(m, int(rng.integers(2, 7)), n, seed=int(rng.integers(1, 100)))
    pos, ids = M.sample_gas(m, n, seed=int(rng.integers(1, 10**6)))
print("case", want, name, n, iters)
rel = lambda a, b: np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.abs(b.astype(np.float64))
for opt in ({}, {"fuse": 0}, {"no_records": 1}):
    g = binding.TcGpu(0, options=opt)
    o = O.Oracle(m, pos, ids, nthreads=16)
    g.set_model(m); g.upload(pos, ids)
    print("options", opt)
    for it in range(iters + 1):
        eo = o.find_sph_quantities(); po = o.particles()
        g.Find_sph_quantities(); pg = g.particles()
        same = np.array_equal(pg["id"], po["id"])
        dp = (np.abs(pg["pos"] - po["pos"]).max(axis=1) / po["hsml"])
        k = int(np.argmax(dp))
        rh = rel(pg["hsml"], po["hsml"]); rr = rel(pg["rho"], po["rho"])
        print(" it %d ids %s  max dpos/h %.3g (particle %d id %d)  hsml rel max %.3g (n>1e-5: %d)  rho rel max %.3g" % (it, same, dp.max(), k, pg["id"][k], rh.max(), int((rh > 1e-5).sum()), rr.max()))
        if it < iters:
            o.wvt_step(0.0085, move=True); g.wvt_step(0.0085, move=True, fetch=False)
    g.close()
