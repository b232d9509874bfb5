"""Validation helper: random small configurations (particle number, rank count, exchange mode, iteration count, model)
on loopback ranks against the single-rank run; every rank must reproduce it bit for bit."""
import sys, threading
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, model as M
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for case in range(ncase):
    R = int(rng.integers(2, 9))
    n = int(rng.integers(3000, 120000)) | 1
    mode = int(rng.integers(0, 3))
    iters = int(rng.integers(1, 7))
    name = "merger" if rng.random() < 0.7 else "single"
    m = M.preset(name, n)
    if rng.random() < 0.3:
        m = M.with_subhalos(m, int(rng.integers(2, 9)), n, seed=int(rng.integers(1, 100)))
    pos, ids = M.sample_gas(m, n, seed=int(rng.integers(1, 10**6)))
    g1 = binding.TcGpu(0); g1.set_model(m); g1.upload(pos, ids)
    log1 = g1.Regularise_sph_particles(max_iter=iters); g1.Find_sph_quantities(); p1 = g1.particles(); g1.close()
    ctxs = [binding.TcGpu(0, options={"ghost_exchange": mode}) for _ in range(R)]
    binding.loopback_group(ctxs)
    out = [None] * R
    def run(r):
        try:
            g = ctxs[r]; g.set_model(m); g.upload(pos, ids)
            log = g.Regularise_sph_particles(max_iter=iters); g.Find_sph_quantities()
            out[r] = (log, g.particles())
        except Exception as e:
            out[r] = e
    th = [threading.Thread(target=run, args=(r,)) for r in range(R)]
    [t.start() for t in th]; [t.join() for t in th]
    [c.close() for c in ctxs]
    ok = True
    for r in range(R):
        if isinstance(out[r], Exception) or out[r] is None:
            ok = False; print("  rank", r, "failed:", out[r]); continue
        log, p = out[r]
        ok = ok and len(log) == len(log1) and all(a == b for a, b in zip(log, log1))
        ok = ok and all(np.array_equal(p[k], p1[k]) for k in ("id", "pos", "hsml", "rho", "varhsmlfac", "rho_model"))
    print("case %2d: %s n=%6d ranks=%d mode=%d iters=%d halos=%d -> %s" % (case, name, n, R, mode, iters, len(m.halos), "equal" if ok else "MISMATCH"), flush=True)
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
