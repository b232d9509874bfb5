"""Second sweep of a fuzz case: which particles get a different displacement, and what is special about them."""
import sys
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, model as M
from oracle import oracle as O
seed, want, nit = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
for case in range(want + 1):
    n = int(rng.integers(2000, 26000)); iters = int(rng.integers(1, 5))
    name = "merger" if rng.random() < 0.7 else "single"
    m = M.preset(name, n)
    if rng.random() < 0.3:
        m = M.with_subhalos(m, int(rng.integers(2, 7)), n, seed=int(rng.integers(1, 100)))
    pos, ids = M.sample_gas(m, n, seed=int(rng.integers(1, 10**6)))
g = binding.TcGpu(0)
o = O.Oracle(m, pos, ids, nthreads=16)
lo = o.regularise(max_iter=nit)
g.set_model(m); g.upload(pos, ids); lg = g.Regularise_sph_particles(max_iter=nit)
print("positions equal after %d sweeps:" % (nit + 1), np.array_equal(o.particles()["pos"], g.particles()["pos"]))
o.find_sph_quantities(); g.Find_sph_quantities()
po, pg = o.particles(), g.particles()
print("ids equal", np.array_equal(po["id"], pg["id"]), "pos equal", np.array_equal(po["pos"], pg["pos"]))
step = lo[-1]["step"]
ohs, ode = o.wvt_step(step, move=False)
for xk in (0, 1):
    g.set_option("xsweep_kernel", xk)
    hs, de = g.wvt_step(step, move=False)
    bad = np.where((de != ode).any(axis=1))[0]
    print("xsweep_kernel", xk, ": hsml_wvt equal", np.array_equal(hs, ohs), " delta rows differing", len(bad))
P = po["pos"].astype(np.float64); box = m.boxsize
print("coordinates == box:", int((po["pos"] == np.float32(box)).sum()), " == 0:", int((po["pos"] == 0).sum()))
for i in bad[:8]:
    d = P - P[i]; d -= box * np.round(d / box)
    r2 = (d * d).sum(axis=1)
    hq = float(np.float32(ohs[i]) * np.float32(box)) if False else float(ohs[i]) * box
    nn = int((r2 < hq * hq).sum())
    lst = o.find_ngb_tree(int(i), np.float32(np.float64(ohs[i]) * box))
    lst2 = o.find_ngb_simple(int(i), np.float32(np.float64(ohs[i]) * box))
    print(" particle", int(i), "pos", po["pos"][i], "hq/box %.4f" % float(ohs[i]), "brute count", nn, "tree list", len(lst), "simple list", len(lst2),
          "tree==simple", np.array_equal(lst, lst2), "delta gpu", de[i], "oracle", ode[i])
g.close()
