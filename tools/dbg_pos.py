"""Where do GPU and oracle positions part?  usage: dbg_pos.py seed case"""
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
g = binding.TcGpu(0)
for it in range(iters + 1):
    o = O.Oracle(m, pos, ids, nthreads=16)
    lo = o.regularise(max_iter=it); po = o.particles()
    g.set_model(m); g.upload(pos, ids)
    lg = g.Regularise_sph_particles(max_iter=it); pg = g.particles()
    step = lo[-1]["step"]
    ohs, ode = o.wvt_step(step, move=False)
    hs, de = g.wvt_step(step, move=False)
    om = o.particles()["rho_model"]; gm = g.particles()["rho_model"]
    print("after %d sweeps: positions differing %d | next sweep: rho_model differing %d (max rel %.3g), hsml_wvt differing %d (max rel %.3g), delta rows differing %d" %
          (it, int((pg["pos"] != po["pos"]).any(axis=1).sum()), int((om != gm).sum()), (np.abs(om - gm) / om).max(), int((hs != ohs).sum()),
           (np.abs(hs - ohs) / ohs).max(), int((de != ode).any(axis=1).sum())), flush=True)
    if (hs != ohs).any():
        k = np.where(hs != ohs)[0][:5]
        print("   first differing hsml_wvt:", [(int(i), float(hs[i]), float(ohs[i])) for i in k], " rho_model there:", [(float(gm[i]), float(om[i])) for i in k])
        print("   ratio hs/ohs min/max over all:", (hs / ohs).min(), (hs / ohs).max())
g.close()
