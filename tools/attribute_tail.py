"""CPU-only attribution of the GPU-vs-oracle parity tail.

Runs the oracle against ITSELF on exactly the cases of tools/fuzz_oracle.py (same seeds, same generator), with ONE
deviation injected at a time (oracle/tc_oracle.h ORC_DEV_*, or 1-ulp noise on the input positions as in BASELINE.md
section 2), and tabulates the tail of the differences.  No GPU involved: whatever tail shows up here is a property of the
reference algorithm under that perturbation, not of a GPU code path.

usage: python tools/attribute_tail.py [seed:ncase ...] [--json out.json]
"""
import sys, json, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import model as M
from oracle import oracle as O

L = O.lib()
L.orc_set_deviation.argtypes = [C.c_int]
L.orc_truncations.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long), C.c_int]

DEVS = [("sweep rounded once (f64 sum, unit step)", 1, False),
        ("solver sums in 64-lane tree order", 2, False),
        ("pow() +-1 f64 ulp", 4, False),
        ("W, W' +-1 f64 ulp before f32 rounding", 8, False),
        ("all four arithmetic deviations", 15, False),
        ("tree search -> exact ball query (Find_ngb_simple's answer)", 16, False),
        ("input positions +-1 f32 ulp (BASELINE.md s2 probe)", 0, True)]


def cases(seed, ncase):
    rng = np.random.default_rng(seed)
    for case in range(ncase):
        n = int(rng.integers(2000, 26000)); iters = int(rng.integers(1, 5))
        name = "merger" if rng.random() < 0.7 else "single"
        m = M.preset(name, n)
        if rng.random() < 0.3:
            m = M.with_subhalos(m, int(rng.integers(2, 7)), n, seed=int(rng.integers(1, 100)))
        pos, ids = M.sample_gas(m, n, seed=int(rng.integers(1, 10**6)))
        yield case, name, n, iters, m, pos, ids


def run(m, pos, ids, iters, dev):
    L.orc_set_deviation(dev)
    try:
        o = O.Oracle(m, pos, ids, nthreads=8)
        log = o.regularise(max_iter=iters); o.find_sph_quantities(); p = o.particles()
        a = ((p["rho_model"].astype(np.float64) / p["rho_model"].max()) ** 0.5).astype(np.float32)
        o.set_apot(np.stack([a, a, a], axis=1)); b = o.bfld_from_rotA()
        td, ts = C.c_long(), C.c_long()
        L.orc_truncations(o._h, C.byref(td), C.byref(ts), 0)
    finally:
        L.orc_set_deviation(0)
    return log, p, b, td.value, ts.value


def main():
    out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    args = [a for a in sys.argv[1:] if ":" in a and a != out]
    runs = [(int(a.split(":")[0]), int(a.split(":")[1])) for a in args] or [(3, 30), (11, 50), (101, 150)]
    acc = {d[0]: dict(dpos=[], dh=[], drho=[], curl=[], ids=0, log=0, ncase=0, npart=0, worst=[]) for d in DEVS}
    trunc_cases = 0; tot_cases = 0
    for seed, ncase in runs:
        for case, name, n, iters, m, pos, ids in cases(seed, ncase):
            lo, po, bo, td, ts = run(m, pos, ids, iters, 0)
            tot_cases += 1; trunc_cases += (ts > 0 or td > 0)
            order_o = np.argsort(po["id"])
            for label, dev, noise in DEVS:
                p_in = pos
                if noise:
                    r = np.random.default_rng(seed * 1000 + case)
                    p_in = np.nextafter(pos, np.where(r.random(pos.shape) < 0.5, np.float32(-1e30), np.float32(1e30))).astype(np.float32)
                    p_in = np.clip(p_in, 0, np.float32(m.boxsize))
                ld, pd, bd, _, _ = run(m, p_in, ids, iters, dev)
                A = acc[label]; A["ncase"] += 1; A["npart"] += n
                if len(ld) != len(lo) or any(a["step"] != b["step"] or abs(a["err_mean"] - b["err_mean"]) > 1e-5 * b["err_mean"] for a, b in zip(ld, lo)):
                    A["log"] += 1
                # join on id: with input noise the Peano order itself may differ
                order_d = np.argsort(pd["id"])
                if not np.array_equal(pd["id"], po["id"]): A["ids"] += 1
                h = po["hsml"][order_o].astype(np.float64)
                dp = np.abs(pd["pos"][order_d].astype(np.float64) - po["pos"][order_o]).max(axis=1)
                dp = np.minimum(dp, m.boxsize - dp) / h
                dh = np.abs(pd["hsml"][order_d] - h) / h
                dr = np.abs(pd["rho"][order_d].astype(np.float64) - po["rho"][order_o]) / po["rho"][order_o]
                dc = np.abs(bd[order_d].astype(np.float64) - bo[order_o]).max(axis=1) / np.abs(bo).max()
                for k, v in (("dpos", dp), ("dh", dh), ("drho", dr), ("curl", dc)): A[k].append(v)
                A["worst"].append((float(dp.max()), float(dh.max()), float(dr.max()), float(dc.max()), seed, case, n, iters, ts, td))
            print("seed %d case %3d %s n=%5d iters=%d halos=%d  truncated sweep/density queries %d/%d" %
                  (seed, case, name, n, iters, len(m.halos), ts, td), flush=True)
    res = {"cases": tot_cases, "cases_with_truncated_lists": trunc_cases, "rows": []}
    print("\n%d cases, %d with at least one NGBMAX-truncated list" % (tot_cases, trunc_cases))
    print("%-52s %28s %28s %28s %28s  ids log" % ("deviation", "dpos/h max 99.9% >1e-3", "dh/h max 99.9% >1e-3", "drho/rho max 99.9% >1e-3", "curl max 99.9% >1e-5"))
    for label, dev, noise in DEVS:
        A = acc[label]; row = {"deviation": label, "particles": A["npart"], "cases": A["ncase"], "cases_order_differs": A["ids"], "cases_log_differs": A["log"]}
        txt = "%-52s" % label
        for k, tol in (("dpos", 1e-3), ("dh", 1e-3), ("drho", 1e-3), ("curl", 1e-5)):
            v = np.concatenate(A[k])
            row[k] = dict(max=float(v.max()), p999=float(np.quantile(v, 0.999)), mean=float(v.mean()), n_over=int((v > tol).sum()), tol=tol)
            txt += " %9.2e %9.2e %8d" % (v.max(), np.quantile(v, 0.999), int((v > tol).sum()))
        w = sorted(A["worst"], reverse=True)[:3]
        row["worst_cases"] = [dict(dpos=a, dh=b, drho=c, curl=d, seed=s, case=cs, n=n, iters=it, trunc_sweep=ts, trunc_density=td) for a, b, c, d, s, cs, n, it, ts, td in w]
        res["rows"].append(row)
        print(txt + "  %3d %3d" % (A["ids"], A["log"]))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
