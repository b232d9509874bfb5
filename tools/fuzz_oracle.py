"""Validation helper: random small configurations, GPU library against the CPU oracle (test infrastructure): ids,
positions, relaxation log, densities after a few WVT iterations, plus the curl.

Tolerances: the sweep's single rounding (DESIGN.md "Numerics") moves positions by ~1e-6 hsml per iteration; now and then
that flips a borderline neighbour, a raw count crosses 295 and the reference's control flow takes its other branch --
hsml then differs at the solver's own tolerance (NNGBDEV / DESNNGB ~ 1.7e-4) and the difference feeds the next move.
tools/dbg_case.py shows the fused, the plain (fuse = 0) and the record-less (no_records) paths agreeing with each
other to the last digit in such a case: it is sensitivity of the algorithm, not a path-specific fault.  Hence 2e-2 hsml
on positions after up to four iterations here; the committed tests use cases that stay below 1e-3."""
import sys
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, model as M
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
g = binding.TcGpu(0)
for case in range(ncase):
    n = int(rng.integers(2000, 26000))
    iters = int(rng.integers(1, 5))
    name = "merger" if rng.random() < 0.7 else "single"
    m = M.preset(name, n)
    if rng.random() < 0.3:
        m = M.with_subhalos(m, int(rng.integers(2, 7)), n, seed=int(rng.integers(1, 100)))
    pos, ids = M.sample_gas(m, n, seed=int(rng.integers(1, 10**6)))
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
        dp = (np.abs(pg["pos"] - po["pos"]).max(axis=1) / po["hsml"]).max()
        if dp > 2e-2: msg.append("pos %.3g hsml" % dp)
        if np.median(rel(pg["rho"], po["rho"])) > 1e-6: msg.append("rho median %.3g" % np.median(rel(pg["rho"], po["rho"])))
        if rel(pg["hsml"], po["hsml"]).max() > 2e-3: msg.append("hsml max %.3g" % rel(pg["hsml"], po["hsml"]).max())
        a = ((po["rho_model"].astype(np.float64) / po["rho_model"].max()) ** 0.5).astype(np.float32)
        apot = np.stack([a, a, a], axis=1)
        o.set_apot(apot); bo = o.bfld_from_rotA(); bg = g.Bfld_from_rotA_SPH(apot)
        db = np.abs(bg - bo).max() / np.abs(bo).max()
        if db > 2e-4: msg.append("curl %.3g" % db)
    print("case %2d: %s n=%5d iters=%d halos=%d -> %s" % (case, name, n, iters, len(m.halos), "ok" if not msg else "; ".join(msg)), flush=True)
    bad += bool(msg)
g.close()
print("failures:", bad)
sys.exit(1 if bad else 0)
