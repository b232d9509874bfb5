"""Validation helper: BASELINE config 2 at FULL size (2e6 gas particles, 2-cluster merger) -- a whole relaxation under the
reference's stop rule on the GPU against the CPU oracle (test infrastructure) on the box's host cores: the log, ids,
positions (bit for bit), hsml, rho after the final density pass.  The oracle answers its ball queries exactly
(DEV_EXACT_BALL, what the reference's Find_ngb_simple returns); its faithful tree search misses the particles of misplaced
nodes now and then (DESIGN.md sections 2.1 and 5).  About 27 iterations x 5-10 s of oracle time.

    python3 tools/config2_vs_oracle.py [n = 2e6] [max_iter = -1: the stop rule]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, hostio
from oracle import oracle as O
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
max_iter = int(sys.argv[2]) if len(sys.argv) > 2 else -1
s = hostio.setup_system("tests/golden/cluster.par", {"ntotal": 2 * n, "mass_ratio": 0.3125})
pos, ids = hostio.sample_gas(s, nthreads=8)
m = hostio.setup_to_model(s)
g = binding.TcGpu(0)
g.set_model(m); g.upload(pos, ids)
t0 = time.perf_counter()
lg = g.Regularise_sph_particles(max_iter=max_iter); g.Find_sph_quantities(); pg = g.particles()
tg = time.perf_counter() - t0
print("GPU: %d iterations, %.2f s, errMean %.6f at stop" % (len(lg), tg, lg[-1]["err_mean"]), flush=True)
O.set_deviation(O.DEV_EXACT_BALL)
o = O.Oracle(m, pos, ids, nthreads=16)
t0 = time.perf_counter()
import threading
done = threading.Event()
def heartbeat():                                   # the oracle call is one long C call: keep the log alive
    while not done.wait(60):
        print("  ... oracle running, %.0f s" % (time.perf_counter() - t0), flush=True)
threading.Thread(target=heartbeat, daemon=True).start()
lo = o.regularise(max_iter=max_iter); o.find_sph_quantities(); po = o.particles()
done.set()
to = time.perf_counter() - t0
print("oracle (exact ball queries, 16 threads): %d iterations, %.1f s, errMean %.6f at stop" % (len(lo), to, lo[-1]["err_mean"]), flush=True)
bad = 0
if len(lg) != len(lo):
    print("log length %d vs %d" % (len(lg), len(lo))); bad += 1
for a, b in zip(lg, lo):
    same = a["step"] == b["step"] and abs(a["err_mean"] - b["err_mean"]) <= 1e-6 * b["err_mean"] and a["err_max"] == b["err_max"]
    print("it %2d step %.6g err_mean %.9g | %.9g err_max %.7g | %.7g  %s" % (a["it"], a["step"], a["err_mean"], b["err_mean"],
                                                                          a["err_max"], b["err_max"], "" if same else "<-- differs"))
    bad += not same
rel = lambda a, b: np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.abs(b.astype(np.float64))
ids_equal = np.array_equal(pg["id"], po["id"])
print("ids (Peano order) equal:", ids_equal)
if ids_equal:
    npos = int((pg["pos"] != po["pos"]).any(axis=1).sum())
    print("positions differing: %d of %d" % (npos, n))
    if npos:
        print("  max |dpos| / hsml: %.3g" % (np.abs(pg["pos"] - po["pos"]).max(axis=1) / po["hsml"]).max())
    print("hsml max rel %.3g, rho max rel %.3g, rho median rel %.3g, varHsmlFac max rel %.3g" %
          (rel(pg["hsml"], po["hsml"]).max(), rel(pg["rho"], po["rho"]).max(), np.median(rel(pg["rho"], po["rho"])),
           rel(pg["varhsmlfac"], po["varhsmlfac"]).max()))
    bad += npos > 0
else:
    bad += 1
# the curl of a smooth vector potential on the relaxed state (K11, src/sph.c:216-300)
if ids_equal:
    a = ((po["rho_model"].astype(np.float64) / po["rho_model"].max()) ** 0.5).astype(np.float32)
    apot = np.stack([a, a, a], axis=1)
    o.set_apot(apot); bo = o.bfld_from_rotA()
    bg = g.Bfld_from_rotA_SPH(apot)                  # on the GPU's own relaxed state
    db = np.abs(bg - bo).max() / np.abs(bo).max()
    print("curl: max |dB| / max |B| = %.3g" % db)
    bad += db > 1e-5
g.close()
print("RESULT:", "bit-equal positions and log" if not bad else "DIFFERENCES (%d)" % bad)
sys.exit(1 if bad else 0)
