"""GPU check of the exact WVT sweep (k_wvt_exact4 / k_wvt_exact): displacement and positions against the oracle with ==,
and the cost of the exact sweep against round 2's fused f64 sweep at BASELINE config 2's size.
usage: python tools/exact_sweep_check.py [n_big]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, model as M
from oracle import oracle as O


def cmp_delta(tag, de, ode, hs, ohs):
    bad = np.where((de != ode).any(axis=1))[0]
    print("%s: hsml_wvt equal %s (%d differ); delta rows differing %d of %d; max |d-o|/max|o| %.3g" %
          (tag, np.array_equal(hs, ohs), int((hs != ohs).sum()), len(bad), len(de), np.abs(de - ode).max() / np.abs(ode).max()), flush=True)


def fuzz_case(seed, want):
    rng = np.random.default_rng(seed)
    for case in range(want + 1):
        n = int(rng.integers(2000, 26000)); iters = int(rng.integers(1, 5))
        name = "merger" if rng.random() < 0.7 else "single"
        m = M.preset(name, n)
        if rng.random() < 0.3:
            m = M.with_subhalos(m, int(rng.integers(2, 7)), n, seed=int(rng.integers(1, 100)))
        pos, ids = M.sample_gas(m, n, seed=int(rng.integers(1, 10**6)))
    return name, n, iters, m, pos, ids


def small_checks(opts):
    print("options", opts, flush=True)
    g = binding.TcGpu(0, options=opts)
    for name, n, seed in (("single", 3000, 5), ("merger", 5000, 6), ("merger", 21497, 7)):
        m = M.preset(name, n)
        pos, ids = M.sample_gas(m, n, seed=seed)
        o = O.Oracle(m, pos, ids, nthreads=16); o.find_sph_quantities()
        ohs, ode = o.wvt_step(0.0085, move=False)
        g.set_model(m); g.upload(pos, ids); g.Find_sph_quantities()
        hs, de = g.wvt_step(0.0085, move=False)
        cmp_delta("sweep %s n=%d" % (name, n), de, ode, hs, ohs)
    n = 20000
    m = M.preset("single", n)
    pos, ids = M.sample_gas(m, n, seed=12)
    rng = np.random.default_rng(3)
    c = np.float32(m.boxsize) * np.float32([0.80, 0.78, 0.76])
    pos[:3200] = (c + rng.normal(0, 0.004 * m.boxsize, (3200, 3))).astype(np.float32)
    o = O.Oracle(m, pos, ids, nthreads=16); o.find_sph_quantities(); ohs, ode = o.wvt_step(0.0085, move=False)
    g.set_model(m); g.upload(pos, ids); g.Find_sph_quantities(); hs, de = g.wvt_step(0.0085, move=False)
    cmp_delta("sweep NGBMAX clump", de, ode, hs, ohs)
    for seed, want in ((3, 16), (101, 74), (101, 129)):
        name, n, iters, m, pos, ids = fuzz_case(seed, want)
        o = O.Oracle(m, pos, ids, nthreads=16)
        lo = o.regularise(max_iter=iters); o.find_sph_quantities(); po = o.particles()
        g.set_model(m); g.upload(pos, ids)
        lg = g.Regularise_sph_particles(max_iter=iters); g.Find_sph_quantities(); pg = g.particles()
        ids_eq = np.array_equal(pg["id"], po["id"])
        npos = int((pg["pos"] != po["pos"]).any(axis=1).sum()) if ids_eq else -1
        rh = np.abs(pg["hsml"].astype(np.float64) - po["hsml"]) / po["hsml"]
        rr = np.abs(pg["rho"].astype(np.float64) - po["rho"]) / po["rho"]
        a = ((po["rho_model"].astype(np.float64) / po["rho_model"].max()) ** 0.5).astype(np.float32)
        apot = np.stack([a, a, a], axis=1)
        o.set_apot(apot); bo = o.bfld_from_rotA(); bg = g.Bfld_from_rotA_SPH(apot)
        db = np.abs(bg - bo).max() / np.abs(bo).max()
        print("fuzz seed %d case %d (%s n=%d iters=%d): ids equal %s, positions differing %d, hsml rel max %.3g, rho rel max %.3g, curl %.3g, log equal %s" %
              (seed, want, name, n, iters, ids_eq, npos, rh.max(), rr.max(), db,
               all(x["step"] == y["step"] and abs(x["err_mean"] - y["err_mean"]) <= 1e-6 * y["err_mean"] for x, y in zip(lg, lo))), flush=True)
    g.close()


def main():
    for opts in ({}, {"sweep": 2}, {"xsweep_kernel": 1}):
        small_checks(opts)
    nbig = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
    m = M.preset("merger", nbig)
    pos, ids = M.sample_gas(m, nbig, seed=11)
    res = {}
    for mode, xk in ((0, 0), (2, 0), (1, 0)):
        g = binding.TcGpu(0, options={"sweep": mode, "timing": 1, "xsweep_kernel": xk})
        g.set_model(m); g.upload(pos, ids)
        g.Regularise_sph_particles(max_iter=3)
        g.phase_times(reset=True)
        t0 = time.time(); g.Regularise_sph_particles(max_iter=4); dt = time.time() - t0
        ph = g.phase_times(reset=True)
        res[(mode, xk)] = g.particles()
        print("sweep=%d xsweep_kernel=%d: %.2f ms per iteration over 5 passes; phases ms/launch: %s" %
              (mode, xk, dt / 5 * 1e3, {k: round(v[0] / max(v[1], 1) * 1e3, 3) for k, v in ph.items() if v[1]}), flush=True)
        g.close()
    a, b, c = res[(0, 0)], res[(2, 0)], res[(1, 0)]
    print("lists from k_iter vs stand-alone kernel after 9 iterations: positions equal %s" % np.array_equal(a["pos"], b["pos"]))
    o0, o1 = np.argsort(a["id"]), np.argsort(c["id"])
    dp = np.abs(a["pos"][o0] - c["pos"][o1]).max(axis=1) / a["hsml"][o0]
    print("exact vs fused after 9 iterations at n=%d (joined on id): max dpos/h %.3g, mean %.3g" % (nbig, dp.max(), dp.mean()))


if __name__ == "__main__":
    main()
