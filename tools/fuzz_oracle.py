"""Validation helper: random small configurations, GPU library against the CPU oracle (test infrastructure): ids,
positions, relaxation log, densities after a few WVT iterations, plus the curl.

Tolerances (round 3; the story is in DESIGN.md section 2 and tools/attribute_tail.py): the sweep reproduces the
reference's f32 accumulation, so POSITIONS must be equal bit for bit, rho to 2e-6, the curl to 1e-5 of max|B|, hsml to the
solver's own band (NNGBDEV / DESNNGB / 3 = 6e-5: a carried hsml that differs in its last bit can take a raw-count
guard of src/sph.c:49-54 the other way; seen once in 4e6 particle-passes at N = 2e5: 5.6e-5, rho 1.6e-6); observed
over 150 cases at N < 26 000: 3.2e-7, 2.2e-7, 2.7e-7 -- against the oracle with exact ball queries (DEV_EXACT_BALL: what the
reference's brute-force Find_ngb_simple returns); its tree search now and then misses the particles of a mis-placed node,
an artefact the library does not reproduce.  Fixed cases of this generator are committed as
tests/test_gpu_parity.py::test_fuzz_cases_of_round_2."""
import sys
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, model as M
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nlo, nhi = (int(float(sys.argv[3])), int(float(sys.argv[4]))) if len(sys.argv) > 4 else (2000, 26000)   # particle numbers (default: the committed cases)
bad = 0
worst = [0.0, 0.0, 0.0]
g = binding.TcGpu(0)
for case in range(ncase):
    n = int(rng.integers(nlo, nhi))
    iters = int(rng.integers(1, 5))
    name = "merger" if rng.random() < 0.7 else "single"
    m = M.preset(name, n)
    if rng.random() < 0.3:
        m = M.with_subhalos(m, int(rng.integers(2, 7)), n, seed=int(rng.integers(1, 100)))
    pos, ids = M.sample_gas(m, n, seed=int(rng.integers(1, 10**6)))
    # the oracle with exact ball queries (Find_ngb_simple's answer): the reference's tree search now and then misses the
    # particles of a mis-placed node, which the library does not reproduce (DESIGN.md sections 2 and 5)
    O.set_deviation(O.DEV_EXACT_BALL)
    o = O.Oracle(m, pos, ids, nthreads=16)
    lo = o.regularise(max_iter=iters); o.find_sph_quantities(); po = o.particles()
    g.set_model(m); g.upload(pos, ids)
    lg = g.Regularise_sph_particles(max_iter=iters); g.Find_sph_quantities(); pg = g.particles()
    rel = lambda a, b: np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.abs(b.astype(np.float64))
    msg = []
    if len(lg) != len(lo): msg.append("log length %d vs %d" % (len(lg), len(lo)))
    else:
        for a, b in zip(lg, lo):
            if abs(a["err_mean"] - b["err_mean"]) > 1e-5 * b["err_mean"] or a["step"] != b["step"]:
                msg.append("log it %d: %.9g vs %.9g" % (a["it"], a["err_mean"], b["err_mean"])); break
    if not np.array_equal(pg["id"], po["id"]): msg.append("ids differ at %d places" % int((pg["id"] != po["id"]).sum()))
    else:
        npos = int((pg["pos"] != po["pos"]).any(axis=1).sum())
        if npos: msg.append("positions differ at %d places (max %.3g hsml)" % (npos, (np.abs(pg["pos"] - po["pos"]).max(axis=1) / po["hsml"]).max()))
        if np.median(rel(pg["rho"], po["rho"])) > 1e-6: msg.append("rho median %.3g" % np.median(rel(pg["rho"], po["rho"])))
        if rel(pg["hsml"], po["hsml"]).max() > 1e-4: msg.append("hsml max %.3g" % rel(pg["hsml"], po["hsml"]).max())
        if rel(pg["rho"], po["rho"]).max() > 2e-6: msg.append("rho max %.3g" % rel(pg["rho"], po["rho"]).max())
        a = ((po["rho_model"].astype(np.float64) / po["rho_model"].max()) ** 0.5).astype(np.float32)
        apot = np.stack([a, a, a], axis=1)
        o.set_apot(apot); bo = o.bfld_from_rotA(); bg = g.Bfld_from_rotA_SPH(apot)
        O.set_deviation(0)
        db = np.abs(bg - bo).max() / np.abs(bo).max()
        if db > 1e-5: msg.append("curl %.3g" % db)
    worst = [max(worst[0], rel(pg["hsml"], po["hsml"]).max()), max(worst[1], rel(pg["rho"], po["rho"]).max()), max(worst[2], db)] if not msg or "ids" not in msg[0] else worst
    print("case %2d: %s n=%5d iters=%d halos=%d -> %s" % (case, name, n, iters, len(m.halos), "ok" if not msg else "; ".join(msg)), flush=True)
    bad += bool(msg)
g.close()
print("worst over all cases: hsml %.3g rho %.3g curl %.3g" % tuple(worst))
print("failures:", bad)
sys.exit(1 if bad else 0)
