"""Validation helper: a whole relaxation (the reference's stop rule) on R loopback ranks against the single-rank run --
log, ids, positions, hsml, rho, varHsmlFac, rho_model must be equal bit for bit on every rank."""
import sys, threading
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, model as M
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 400003
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=99)
g1 = binding.TcGpu(0); g1.set_model(m); g1.upload(pos, ids)
log1 = g1.Regularise_sph_particles(); g1.Find_sph_quantities(); p1 = g1.particles(); g1.close()
ctxs = [binding.TcGpu(0, options={"ghost_exchange": mode}) for _ in range(R)]
binding.loopback_group(ctxs)
out = [None] * R
def run(r):
    try:
        g = ctxs[r]; g.set_model(m); g.upload(pos, ids)
        log = g.Regularise_sph_particles(); info = g.local_set_info(); nb = g.comm_bytes(); g.Find_sph_quantities()
        out[r] = (log, g.particles(), info, nb)
    except Exception as e:
        out[r] = e
th = [threading.Thread(target=run, args=(r,)) for r in range(R)]
[t.start() for t in th]; [t.join() for t in th]
[c.close() for c in ctxs]
ok = True
for r in range(R):
    if isinstance(out[r], Exception) or out[r] is None:
        print("rank", r, "failed:", out[r]); ok = False; continue
    log, p, info, nb = out[r]
    same_log = len(log) == len(log1) and all(a == b for a, b in zip(log, log1))
    same = all(np.array_equal(p[k], p1[k]) for k in ("id", "pos", "hsml", "rho", "varhsmlfac", "rho_model"))
    print("rank %d: iterations %d, log equal %s, particles equal %s, nloc/nown %.2f, retries %d, received %.1f MB"
          % (r, len(log), same_log, same, info["nloc"] / info["nown"], info["retries"], nb / 1e6))
    ok = ok and same_log and same
print("single rank: %d iterations, errMean %.6f" % (len(log1), log1[-1]["err_mean"]))
print("ALL EQUAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
