"""GPU parity tests (run with -m gpu on an MI355X).  Every call goes through the C ABI of
libtcgpu.so (toycluster_amd.binding -> include/tcgpu.h); the oracle is only the checker.

Parity tiers (DESIGN.md section 2 for what round 3 changed):
  T0 bit-exact : Peano keys, sort permutation / ids, iteration count, step schedule -- and, since round 3, the WVT
                 displacement and therefore the POSITIONS after any number of iterations (the sweep reproduces the
                 reference's per-neighbour f32 accumulation in ascending index, src/wvt_relax.c:167-169)
  T1 set-exact : neighbour sets for identical f32 inputs
  T2 tolerance : |drho|/rho <= 1e-3, |dhsml|/hsml <= 1e-3 (observed: the solver's own +-0.05/295 band, <= 2e-4 / 6e-4,
                 when a carried hsml differs in its last bit and a raw-count guard of src/sph.c:49-54 goes the other
                 way), errMean to 5 digits
The only differences expected from the oracle are f64 summation order (1e-16 relative) and the
device libm's pow() (<= 1 ulp f64), so single-pass results are compared far tighter than T2.
"""
import json
import os

import numpy as np
import pytest

from toycluster_amd import binding, model as M
from oracle import oracle as O

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TOL_POS = 1e-3     # in units of hsml
TOL_RHO = 1e-3
TOL_HSML = 1e-3


def rel(a, b):
    return np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.abs(b.astype(np.float64))


# ------------------------------------------------------------------ T0: keys / sort

def test_peano_known_answers_on_device(gpu):
    kat = json.load(open(os.path.join(GOLDEN, "peano_kat.json")))["exact"]
    xyz = np.array([e["xyz"] for e in kat], dtype=np.float64)
    keys = gpu.Peano_Key(xyz)
    for e, k in zip(kat, keys):
        assert k == int(e["key"], 16), (e, hex(k))


def test_peano_keys_random_vs_oracle(gpu):
    rng = np.random.default_rng(1)
    xyz = rng.random((20000, 3))
    xyz[:50] = np.round(xyz[:50] * 8) / 8          # cell corners, incl. exact 0 and 1
    keys = gpu.Peano_Key(xyz)
    for p, k in zip(xyz[:3000], keys[:3000]):
        assert k == O.peano_key(*p)


def test_sort_is_bit_exact(gpu, golden_case):
    c = golden_case
    gpu.set_model(c["model"])
    gpu.upload(c["pos"], c["ids"])
    hi, lo = gpu.Sort_Particles_By_Peano_Key()
    assert np.array_equal(hi, c["key_hi"]) and np.array_equal(lo, c["key_lo"])
    p = gpu.particles()
    assert np.array_equal(p["id"], c["ids"][c["perm"]])
    assert np.array_equal(p["pos"], c["pos"][c["perm"]])


def test_sort_with_duplicates_and_box_faces(gpu):
    """Edge cases: duplicated positions (equal keys), coordinates exactly 0 and exactly boxsize."""
    n = 5000
    m = M.preset("single", n)
    pos, ids = M.sample_gas(m, n, seed=21)
    pos[100:110] = pos[100]                         # ties
    pos[200, 0] = 0.0
    pos[201, 1] = np.float32(m.boxsize)             # "orphan": X = 2^63 (peano.c:134-136)
    pos[202] = np.float32(m.boxsize)
    gpu.set_model(m)
    gpu.upload(pos, ids)
    hi, lo = gpu.Sort_Particles_By_Peano_Key()
    k = [(int(h) << 64) | int(l) for h, l in zip(hi, lo)]
    assert all(a <= b for a, b in zip(k[:-1], k[1:]))
    o = O.Oracle(m, pos, ids)
    ohi, olo, _ = o.sort_by_peano_key()
    assert np.array_equal(hi, ohi) and np.array_equal(lo, olo)
    p = gpu.particles()
    assert sorted(p["id"]) == sorted(ids)
    # neighbour queries stay exact with orphans present (brute-force oracle: wvt_relax.c:296-340)
    gpu.build_neighbour_index()
    og = O.Oracle(m, p["pos"], p["id"])
    for i in list(range(0, n, 250)) + [int(np.where(p["id"] == ids[202])[0][0])]:
        for h in (300.0, 900.0, 2500.0):
            a = gpu.Find_ngb_tree(i, h)
            b = og.find_ngb_simple(i, h)
            assert np.array_equal(a, b), (i, h, len(a), len(b))


def test_sort_large_set_with_close_pairs():
    """Above 2^20 particles the radix sort looks at fewer leading key bits and the fix-up orders what ties
    (csrc/sort.hip).  Half a million close pairs (offsets of 1e-6 .. 1e-4 of the box: equal in the sorted bits,
    different below) must still come out in full 128-bit key order, ids carried along."""
    n = 1_300_000
    m = M.preset("single", n)
    rng = np.random.default_rng(17)
    box = np.float32(m.boxsize)
    pos = (rng.random((n, 3)) * 0.98 + 0.01).astype(np.float32) * box
    k = 500_000
    pos[n - k:] = pos[:k] + (rng.random((k, 3)) * 1e-4 + 1e-6).astype(np.float32) * box
    ids = np.arange(1, n + 1, dtype=np.int32)
    g = binding.TcGpu(0)
    g.set_model(m)
    g.upload(pos, ids)
    hi, lo = g.Sort_Particles_By_Peano_Key()
    p = g.particles()
    g.close()
    assert np.all(hi[1:] >= hi[:-1])
    same = hi[1:] == hi[:-1]
    assert np.all(lo[1:][same] >= lo[:-1][same])
    assert np.array_equal(np.sort(p["id"]), ids)
    assert np.array_equal(p["pos"], pos[p["id"] - 1])                  # whole particles moved
    # keys belong to the particles they sit next to
    for i in rng.integers(0, n, 50):
        x, y, z = (float(v) / float(box) for v in p["pos"][i].astype(np.float64))
        key = O.peano_key(x, y, z)
        assert (key >> 64, key & ((1 << 64) - 1)) == (int(hi[i]), int(lo[i]))


def test_sort_keys_tying_in_the_high_half(gpu):
    """Particles a few f32 ulps apart share the high 64 key bits (21 Hilbert levels) and differ only in the
    low half: the one-pass radix sort + tie fix-up must still give the full 128-bit order.  Includes a run
    of 40 such particles (heapsort branch of the fix-up) and exact duplicates."""
    n = 6000
    m = M.preset("single", n)
    pos, ids = M.sample_gas(m, n, seed=77)
    base = pos[1000].copy()
    base[:] = np.float32(m.boxsize) * np.float32([0.731, 0.642, 0.553])
    for k in range(40):                                  # consecutive floats in x, scrambled order
        pos[2000 + (k * 7) % 40] = base
        pos[2000 + (k * 7) % 40, 0] = np.nextafter(base[0], np.float32(np.inf), dtype=np.float32) if k == 0 else \
            np.float32(base[0]) + np.float32(k) * np.spacing(np.float32(base[0]))
    pos[3000] = pos[3001] = pos[3002]                    # exact triple
    pos[3100, :] = pos[3101, :]
    pos[3101, 2] = np.nextafter(pos[3100, 2], np.float32(0), dtype=np.float32)
    gpu.set_model(m)
    gpu.upload(pos, ids)
    hi, lo = gpu.Sort_Particles_By_Peano_Key()
    o = O.Oracle(m, pos, ids)
    ohi, olo, operm = o.sort_by_peano_key()
    assert np.array_equal(hi, ohi) and np.array_equal(lo, olo)
    assert (np.diff(hi.astype(np.float64)) == 0).sum() >= 40          # the case really has high-half ties
    p, q = gpu.particles(), o.particles()
    k = (hi.astype(object) << 64) | lo.astype(object)
    distinct = np.r_[True, k[1:] != k[:-1]] & np.r_[k[1:] != k[:-1], True]
    assert np.array_equal(p["id"][distinct], q["id"][distinct])       # unique keys: unique order
    # identical keys keep their upload order (stable sort)
    trip = np.flatnonzero(np.isin(p["id"], ids[3000:3003]))
    assert list(p["id"][trip]) == list(ids[3000:3003])


# ------------------------------------------------------------------ T1: neighbour sets

def test_neighbour_sets_exact(gpu, golden_case):
    c = golden_case
    gpu.set_model(c["model"])
    gpu.upload(c["pos"], c["ids"])
    gpu.build_neighbour_index()
    for k, (i, h) in enumerate(zip(c["ngb_probe"], c["ngb_hsml"])):
        got = gpu.Find_ngb_tree(int(i), float(h))
        want = c["ngb_lists"][c["ngb_offsets"][k]:c["ngb_offsets"][k + 1]]
        assert np.array_equal(got, want), (i, h, len(got), len(want))


def test_neighbour_sets_all_radii(gpu):
    """From 0.2% of the box to beyond the half box (periodic wrap, whole-box coverage)."""
    n = 8000
    m = M.preset("merger", n)
    pos, ids = M.sample_gas(m, n, seed=4)
    gpu.set_model(m)
    gpu.upload(pos, ids)
    gpu.build_neighbour_index()
    p = gpu.particles()
    o = O.Oracle(m, p["pos"], p["id"])
    rng = np.random.default_rng(2)
    for i in rng.integers(0, n, 25):
        for f in (0.002, 0.01, 0.03, 0.1, 0.3, 0.49, 0.55, 0.9):
            a = gpu.Find_ngb_tree(int(i), f * m.boxsize)
            b = o.find_ngb_simple(int(i), f * m.boxsize)
            assert np.array_equal(a, b), (i, f, len(a), len(b))


def test_guess_hsml_matches_reference_tree(gpu, golden_case):
    """tree.c:113-121 evaluated without building the tree == the serial tree's answer."""
    c = golden_case
    gpu.set_model(c["model"])
    gpu.upload(c["pos"], c["ids"])
    g = gpu.Guess_hsml()
    bad = np.flatnonzero(g != c["guess"])
    assert len(bad) == 0, (len(bad), bad[:10], g[bad[:10]], c["guess"][bad[:10]])


def test_guess_hsml_larger_case(gpu):
    n = 60000
    m = M.preset("merger", n)
    pos, ids = M.sample_gas(m, n, seed=17)
    gpu.set_model(m)
    gpu.upload(pos, ids)
    g = gpu.Guess_hsml()
    o = O.Oracle(m, pos, ids)
    o.sort_by_peano_key()
    o.build_tree()
    want = np.array([o.guess_hsml(i) for i in range(n)], np.float32)
    assert np.array_equal(g, want), int((g != want).sum())


# ------------------------------------------------------------------ T2: density pass

def test_density_model(gpu, golden_case):
    c = golden_case
    gpu.set_model(c["model"])
    gpu.upload(c["d_pos"], c["d_ids"])
    got = gpu.Global_density_model()
    assert rel(got, c["d_rho_model"]).max() < 2e-7          # device pow() vs glibc pow(), f32 result


def test_double_beta_cool_core_variant(gpu):
    """The reference's -DDOUBLE_BETA_COOL_CORES build (src/setup.c:604-612; a run-time switch here: rho0_cc / rc_cc of
    tcgpu_halo): the cool-core component of a cuspy halo in the density model, the model smoothing lengths and the
    relaxation they steer, against the oracle's restatement of that build (parity unpinned like the default build)."""
    import copy
    n = 20000
    m = copy.deepcopy(M.preset("merger", n))
    m.halos[0].have_cuspy = 1                              # Param.Cuspy = 1: halo 0 is a cool core, halo 1 is not
    m.rho0_fac, m.rc_fac = 8.0, 6.0                        # Param.Rho0_Fac, Param.Rc_Fac
    pos, ids = M.sample_gas(m, n, seed=5)
    O.set_double_beta(m.rho0_fac, m.rc_fac)
    try:
        o = O.Oracle(m, pos, ids, nthreads=8)
        want_log = o.regularise(max_iter=3)
        want_rm = o.global_density_model()              # of the oracle's final positions
        want = o.particles()
    finally:
        O.set_double_beta(0, 0)
    gpu.set_model(m)
    gpu.upload(pos, ids)
    log = gpu.Regularise_sph_particles(max_iter=3)
    got = gpu.particles()
    assert len(log) == len(want_log)
    for a, b in zip(log, want_log):
        assert a["err_mean"] == pytest.approx(b["err_mean"], rel=1e-5) and a["step"] == b["step"]
    assert np.array_equal(got["id"], want["id"])
    assert rel(got["rho_model"], want["rho_model"]).max() < 5e-5     # of positions that differ by the sweep's 1e-6
    # the component really is in the model: without it the same positions give a lower central density
    gpu.set_model(m)
    gpu.upload(pos, ids)
    rm_cc = gpu.Global_density_model()
    m0 = copy.deepcopy(m)
    m0.rho0_fac = m0.rc_fac = 0.0
    gpu.set_model(m0)
    gpu.upload(pos, ids)
    rm_0 = gpu.Global_density_model()
    assert rm_cc.max() > 2 * rm_0.max() and (rm_cc >= rm_0).all()
    O.set_double_beta(m.rho0_fac, m.rc_fac)
    try:
        o = O.Oracle(m, pos, ids, nthreads=8)
        o.sort_particles_by_peano_key() if hasattr(o, "sort_particles_by_peano_key") else None
        want_rm0 = o.global_density_model()
        want_id = o.particles()["id"]
    finally:
        O.set_double_beta(0, 0)
    # compare particle by particle through the ids (the oracle object may or may not have been sorted)
    gpu.set_model(m)
    gpu.upload(pos, ids)
    rm = gpu.Global_density_model()
    gid = gpu.particles()["id"]
    a = np.empty(n + 1, np.float64); a[gid] = rm
    b = np.empty(n + 1, np.float64); b[want_id] = want_rm0
    assert rel(a[1:], b[1:]).max() < 2e-7


def test_cubic_spline_variant_vs_oracle():
    """libtcgpu_m4.so, the library build that restates the reference's -DSPH_CUBIC_SPLINE build (Makefile:25: M4 kernel,
    50 neighbours, NGBMAX 400, no bias correction, varHsmlFac = 1, WVT step 0.035), against the oracle built the same
    way (libtcoracle_m4.so): cold density pass, three relaxation iterations, the SPH curl.  Parity unpinned."""
    n = 30000
    m = M.preset("merger", n)
    pos, ids = M.sample_gas(m, n, seed=11)
    o = O.Oracle(m, pos, ids, nthreads=8, variant="m4")
    o.find_sph_quantities()
    po = o.particles()
    g = binding.TcGpu(0, variant="m4")
    try:
        assert b"CUBIC_SPLINE" in g._L.tcgpu_version()
        g.set_model(m)
        g.upload(pos, ids)
        g.set_option("stats", 1)
        g.Find_sph_quantities()
        pg = g.particles()
        assert np.array_equal(pg["id"], po["id"])
        assert rel(pg["hsml"], po["hsml"]).max() < 1e-6 and rel(pg["rho"], po["rho"]).max() < 1e-6
        assert (pg["varhsmlfac"] == 1.0).all() and (po["varhsmlfac"] == 1.0).all()       # sph.c:201: dRhodHsml stays 0
        # 50 +- 0.05 kernel-weighted neighbours (globals.h:42-43), by the reference's own formula on a sample
        g.build_neighbour_index()
        for i in range(0, n, n // 40):
            ngb = g.Find_ngb_tree(i, pg["hsml"][i])
            d = pg["pos"][ngb].astype(np.float64) - pg["pos"][i].astype(np.float64)
            d -= m.boxsize * np.round(d / m.boxsize)
            r = np.sqrt((d * d).sum(axis=1))
            u = (r.astype(np.float32) / pg["hsml"][i]).astype(np.float64)
            wk = np.where(u < 0.5, 2.546479089470 + 15.278874536822 * (u - 1) * u * u, 5.092958178941 * (1 - u) ** 3)
            wk = np.where(u > 1, 0.0, wk)
            assert abs(4.18879032135009765 * wk.sum() - 50) < 0.06, (i, 4.18879032135009765 * wk.sum())
        # relaxation: same log (step 0.035, wvt_relax.c:48-49), same particles
        g.upload(pos, ids)
        log = g.Regularise_sph_particles(max_iter=3)
        o = O.Oracle(m, pos, ids, nthreads=8, variant="m4")
        want = o.regularise(max_iter=3)
        assert len(log) == len(want) and all(l["step"] == 0.035 for l in log)
        for a, b in zip(log, want):
            assert a["err_mean"] == pytest.approx(b["err_mean"], rel=1e-5) and a["err_max"] == pytest.approx(b["err_max"], rel=1e-4)
        g.Find_sph_quantities()
        o.find_sph_quantities()
        pg, po = g.particles(), o.particles()
        assert np.array_equal(pg["id"], po["id"])
        assert (np.abs(pg["pos"] - po["pos"]).max(axis=1) / po["hsml"]).max() < TOL_POS
        assert np.median(rel(pg["rho"], po["rho"])) < 1e-6
        # curl of a smooth vector potential
        rm = g.Global_density_model().astype(np.float64)
        a = ((rm / max(h.rho0 for h in m.halos)) ** 0.5).astype(np.float32)
        apot = np.stack([a, 0.5 * a, -a], axis=1)
        b = g.Bfld_from_rotA_SPH(apot)
        o.set_apot(apot)
        bo = o.bfld_from_rotA()
        assert np.abs(b - bo).max() < 1e-4 * np.abs(bo).max()
    finally:
        g.close()


def test_unusable_query_records_fall_back(golden_case):
    """k_prec / k_cprec hand the solver kernels a 64-byte record per particle; one that cannot describe its query (a cell
    box wider than 127 cells -- never seen in practice) must send the particle down the plain per-query code, the curl
    down its literal path.  Forced for every particle here: same relaxation log, same particles (to summation order),
    same field as the default run."""
    c = golden_case
    out = []
    for flag in (0, 1):
        g = binding.TcGpu(0, options={"no_records": flag})
        try:
            g.set_model(c["model"])
            g.upload(c["pos"], c["ids"])
            log = g.Regularise_sph_particles(max_iter=2)
            g.Find_sph_quantities()
            p = g.particles()
            a = (p["rho_model"] / p["rho_model"].max()) ** 0.5
            b = g.Bfld_from_rotA_SPH(np.stack([a, a, a], axis=1).astype(np.float32))
            out.append((log, p, b))
        finally:
            g.close()
    (l0, p0, b0), (l1, p1, b1) = out
    assert len(l0) == len(l1)
    for x, y in zip(l0, l1):
        assert x["err_mean"] == pytest.approx(y["err_mean"], rel=1e-6) and x["step"] == y["step"]
    assert np.array_equal(p0["id"], p1["id"])
    assert (np.abs(p0["pos"] - p1["pos"]).max(axis=1) / p0["hsml"]).max() < TOL_POS
    assert np.median(rel(p1["hsml"], p0["hsml"])) < 1e-6 and rel(p1["rho"], p0["rho"]).max() < 1e-4
    assert np.abs(b1 - b0).max() < 1e-5 * np.abs(b0).max()


def test_density_pass_cold(gpu, golden_case):
    c = golden_case
    gpu.set_model(c["model"])
    gpu.upload(c["pos"], c["ids"])                          # hsml = 0: first-pass guess path
    gpu.Find_sph_quantities()
    p = gpu.particles()
    assert np.array_equal(p["id"], c["d_ids"])
    assert rel(p["hsml"], c["d_hsml"]).max() < 1e-6
    assert rel(p["rho"], c["d_rho"]).max() < 1e-6
    assert np.abs(p["varhsmlfac"] - c["d_vhf"]).max() < 1e-5
    # nearly all values are bit-identical (only f64 summation order differs)
    assert (p["hsml"] == c["d_hsml"]).mean() > 0.99
    assert (p["rho"] == c["d_rho"]).mean() > 0.99


def test_density_pass_tiny_input(gpu):
    """Barely more particles than DESNNGB: smoothing lengths comparable to the box (whole-box queries,
    periodic images), many x1.23 retries.  The reference would hang below 256 particles per thread
    (SURVEY.md section 5); the library has no such limit."""
    n = 700
    m = M.preset("single", n)
    pos, ids = M.sample_gas(m, n, seed=13)
    gpu.set_model(m)
    gpu.upload(pos, ids)
    gpu.Find_sph_quantities()
    p = gpu.particles()
    o = O.Oracle(m, pos, ids, nthreads=1)
    o.find_sph_quantities()
    q = o.particles()
    assert np.array_equal(p["id"], q["id"])
    assert rel(p["hsml"], q["hsml"]).max() < 1e-6 and rel(p["rho"], q["rho"]).max() < 1e-6
    assert p["hsml"].max() > 0.3 * m.boxsize
    log = gpu.Regularise_sph_particles(max_iter=2)
    olog = o.regularise(max_iter=2)
    assert [l["err_mean"] for l in log] == pytest.approx([l["err_mean"] for l in olog], rel=1e-5)


def test_density_pass_warm_and_idempotent(gpu):
    n = 30000
    m = M.preset("merger", n)
    pos, ids = M.sample_gas(m, n, seed=8)
    gpu.set_model(m)
    gpu.upload(pos, ids)
    gpu.set_option("stats", 1)
    gpu.Find_sph_quantities()
    cold = gpu.density_stats()
    p1 = gpu.particles()
    gpu.Find_sph_quantities()                               # warm start from the carried hsml
    warm = gpu.density_stats()
    gpu.set_option("stats", 0)
    p2 = gpu.particles()
    o = O.Oracle(m, pos, ids)
    o.find_sph_quantities()
    so_cold = o.last_stats()
    q1 = o.particles()
    o.find_sph_quantities()
    so_warm = o.last_stats()
    q2 = o.particles()
    for p, q in ((p1, q1), (p2, q2)):
        assert np.array_equal(p["id"], q["id"])
        assert rel(p["hsml"], q["hsml"]).max() < 1e-6 and rel(p["rho"], q["rho"]).max() < 1e-6
    # same control flow as the reference: same number of queries and solver iterations per particle
    for a, b in ((cold, so_cold), (warm, so_warm)):
        assert a["queries"] == pytest.approx(b["queries"], rel=1e-3)
        assert a["solver_iters"] == pytest.approx(b["solver_iters"], rel=1e-3)
        assert a["pair_evals"] == pytest.approx(b["pair_evals"], rel=1e-3)
    # idempotence: a warm pass moves hsml only inside the +-0.05-neighbour band
    assert rel(p2["hsml"], p1["hsml"]).max() < 1e-3


# ------------------------------------------------------------------ T2: WVT sweep

def test_wvt_sweep_single_step(gpu, golden_case):
    c = golden_case
    gpu.set_model(c["model"])
    gpu.upload(c["pos"], c["ids"])
    gpu.Find_sph_quantities()
    hs, de = gpu.wvt_step(0.0085, move=False)
    # the reference's order and roundings (f32 accumulator, one rounding per neighbour, ascending index): bit for bit
    assert np.array_equal(hs, c["w_hsml"])
    assert np.array_equal(de, c["w_delta"])


def test_wvt_sweep_second_implementation_and_f64_mode(golden_case):
    """The one-lane-per-particle kernel on the (x, y, z) cell table (option xsweep_kernel = 1) is an independent second
    implementation of the exact sweep: same bits.  Round 2's sweep (option sweep = 1: f64 sums over 64 lanes, rounded
    once) stays available and stays within 2e-6 of the scale -- the deviation that owned round 2's parity tail."""
    c = golden_case
    scale = np.abs(c["w_delta"]).max()
    for opts, exact in (({"xsweep_kernel": 1}, True), ({"sweep": 1}, False), ({"sweep": 1, "fuse": 0}, False)):
        g = binding.TcGpu(0, options=opts)
        g.set_model(c["model"]); g.upload(c["pos"], c["ids"])
        g.Find_sph_quantities()
        if not exact: g.density_error()                     # warm pass: the fused kernel sums the sweep on the way
        hs, de = g.wvt_step(0.0085, move=False)
        g.close()
        if exact:
            assert np.array_equal(de, c["w_delta"])
        else:
            assert rel(hs, c["w_hsml"]).max() < 3e-7 and np.abs(de - c["w_delta"]).max() < 2e-6 * scale


def test_cell_start_table_built_either_way_gives_the_same_runs():
    """The curve-ordered cell starts (pf) are made by one scan over every level, or -- for the thin local set of a
    sharded rank -- with the three deepest levels filled block by block under occupied cells only (k_pf_deep); the
    ordered walks must read the same index runs from both: whole relaxations equal bit for bit, with either sweep
    implementation that walks the table."""
    n = 60013
    m = M.preset("merger", n)
    pos, ids = M.sample_gas(m, n, seed=77)
    for sweep in (0, 2):                                   # the list path (k_xruns) and the stand-alone kernel (k_wvt_exact4)
        out = []
        for mode in (1, 2):
            g = binding.TcGpu(0, options={"pf_mode": mode, "sweep": sweep})
            g.set_model(m); g.upload(pos, ids)
            log = g.Regularise_sph_particles(max_iter=5); g.Find_sph_quantities()
            out.append((log, g.particles()))
            g.close()
        (la, pa), (lb, pb) = out
        assert [(a["step"], a["err_mean"], a["err_max"]) for a in la] == [(b["step"], b["err_mean"], b["err_max"]) for b in lb]
        for k in ("id", "pos", "hsml", "rho", "varhsmlfac"):
            assert np.array_equal(pa[k], pb[k]), (sweep, k)


def test_relaxation_short(gpu, golden_case):
    c = golden_case
    gpu.set_model(c["model"])
    gpu.upload(c["pos"], c["ids"])
    log = gpu.Regularise_sph_particles(max_iter=3)
    want = c["r_log"]
    assert len(log) == len(want)
    for l, w in zip(log, want):
        assert l["it"] == int(w[0]) and l["step"] == w[4]
        assert l["err_mean"] == pytest.approx(w[2], rel=1e-5)
        assert l["err_max"] == pytest.approx(w[1], rel=1e-3)
    p = gpu.particles()
    assert np.array_equal(p["id"], c["r_ids"])
    assert np.array_equal(p["pos"], c["r_pos"])                      # T0 since round 3
    assert rel(p["hsml"], c["r_hsml"]).max() < TOL_HSML and rel(p["rho"], c["r_rho"]).max() < TOL_RHO


def test_relaxation_to_convergence_matches_oracle(gpu):
    """Config-1-shaped run at a size the oracle finishes in seconds: identical iteration count, step
    schedule and log to 5 digits; positions/densities inside T2 after the whole loop."""
    n = 20000
    m = M.preset("single", n)
    pos, ids = M.sample_gas(m, n, seed=1)
    gpu.set_model(m)
    gpu.upload(pos, ids)
    log = gpu.Regularise_sph_particles()
    o = O.Oracle(m, pos, ids)
    olog = o.regularise()
    assert len(log) == len(olog)
    for a, b in zip(log, olog):
        assert a["it"] == b["it"] and a["step"] == b["step"]
        assert a["err_mean"] == pytest.approx(b["err_mean"], rel=1e-5)
    p, q = gpu.particles(), o.particles()
    assert np.array_equal(p["id"], q["id"])
    assert np.array_equal(p["pos"], q["pos"])                        # T0 since round 3
    assert rel(p["hsml"], q["hsml"]).max() < TOL_HSML and rel(p["rho"], q["rho"]).max() < TOL_RHO


FUZZ_CASES = [(3, 16), (101, 74), (101, 129), (3, 4), (11, 40), (101, 30)]


@pytest.mark.parametrize("seed,want", FUZZ_CASES)
def test_fuzz_cases_of_round_2(gpu, seed, want):
    """Fixed cases of tools/fuzz_oracle.py, first of all the three that FAILED at the end of round 2 (seed 3 case 16:
    positions off by 6.8e-3 hsml; seed 101 case 74: curl 3.1e-4; seed 101 case 129: hsml 2.2e-3), then the worst cases
    of the CPU-only attribution run (tools/attribute_tail.py, profiles/round3_parity_tail_attribution.json).  That run
    reproduced those very numbers with ONE deviation injected into the oracle -- the sweep rounded once instead of after
    every neighbour -- and zeros for the other three.  With the sweep exact, the positions are equal bit for bit and
    hsml / rho / curl agree to a few f32 ulp -- against the algorithm with exact ball queries (a).  Against the
    faithful tree search (b) a second, unrelated effect remains: the reference's tree misses neighbours near a mis-placed
    node.  Small N on purpose: hundreds of NGBMAX-truncated lists per case (src/tree.c:91-92), the discontinuity that
    turned 1e-6 into 7e-3."""
    rng = np.random.default_rng(seed)
    for case in range(want + 1):
        n = int(rng.integers(2000, 26000)); iters = int(rng.integers(1, 5))
        name = "merger" if rng.random() < 0.7 else "single"
        m = M.preset(name, n)
        if rng.random() < 0.3:
            m = M.with_subhalos(m, int(rng.integers(2, 7)), n, seed=int(rng.integers(1, 100)))
        pos, ids = M.sample_gas(m, n, seed=int(rng.integers(1, 10**6)))
    gpu.set_model(m); gpu.upload(pos, ids)
    lg = gpu.Regularise_sph_particles(max_iter=iters); gpu.Find_sph_quantities(); pg = gpu.particles()
    # (a) against the oracle with every ball query answered exactly (the reference's Find_ngb_simple, wvt_relax.c:296-340)
    O.set_deviation(O.DEV_EXACT_BALL)
    try:
        o = O.Oracle(m, pos, ids)
        lo = o.regularise(max_iter=iters); o.find_sph_quantities(); po = o.particles()
        a = ((po["rho_model"].astype(np.float64) / po["rho_model"].max()) ** 0.5).astype(np.float32)
        apot = np.stack([a, a, a], axis=1)
        o.set_apot(apot); bo = o.bfld_from_rotA()
    finally:
        O.set_deviation(0)
    assert len(lg) == len(lo)
    for x, y in zip(lg, lo):
        assert x["step"] == y["step"] and x["err_mean"] == pytest.approx(y["err_mean"], rel=1e-6)
    assert np.array_equal(pg["id"], po["id"])
    assert np.array_equal(pg["pos"], po["pos"])
    assert rel(pg["hsml"], po["hsml"]).max() < 2e-6 and rel(pg["rho"], po["rho"]).max() < 2e-6
    assert np.median(rel(pg["rho"], po["rho"])) == 0
    bg = gpu.Bfld_from_rotA_SPH(apot)
    assert np.abs(bg - bo).max() < 1e-5 * np.abs(bo).max()
    # (b) against the faithful restatement, tree search and all: the same, but for the neighbourhood of the occasional
    # node that Build_Tree puts under a parent of the wrong level (tree.c:201-226, :297-306; ~1e-5 of the leaves): its
    # particles are missed by queries that should find them.  Not reproduced.
    o = O.Oracle(m, pos, ids)
    o.regularise(max_iter=iters); o.find_sph_quantities(); pf = o.particles()
    assert np.array_equal(pg["id"], pf["id"])
    dp = np.abs(pg["pos"] - pf["pos"]).max(axis=1) / pf["hsml"]
    assert dp.max() < TOL_POS and (dp > 0).mean() < 0.15


def test_config1_full_relaxation(gpu):
    """BASELINE config 1: stock cluster.par shape, Ntotal 200000 -> 1e5 gas particles, single cluster,
    relaxed until the reference's stop rule fires; compared with the oracle run on the host cores."""
    n = 100_000
    m = M.preset("single", n)
    pos, ids = M.sample_gas(m, n, seed=14041981)
    gpu.set_model(m)
    gpu.upload(pos, ids)
    log = gpu.Regularise_sph_particles()
    gpu.Find_sph_quantities()                                  # main.c:54
    o = O.Oracle(m, pos, ids)
    olog = o.regularise()
    o.find_sph_quantities()
    assert len(log) == len(olog)
    for a, b in zip(log, olog):
        assert a["it"] == b["it"] and a["step"] == b["step"]
        assert a["err_mean"] == pytest.approx(b["err_mean"], rel=1e-5)
        assert a["err_max"] == pytest.approx(b["err_max"], rel=1e-4)
        assert binding.format_log_line(a)[:17] == O.format_log_line(b)[:17]      # '   #NN: Err max=N'
    p, q = gpu.particles(), o.particles()
    assert np.array_equal(p["id"], q["id"])
    dpos = np.abs(p["pos"] - q["pos"]).max(axis=1) / q["hsml"]
    assert dpos.max() < TOL_POS and dpos.mean() < 1e-5
    assert rel(p["hsml"], q["hsml"]).max() < TOL_HSML and rel(p["rho"], q["rho"]).max() < TOL_RHO
    # with the oracle's ball queries answered exactly (no neighbours lost near misplaced tree nodes) the
    # whole relaxation is reproduced bit for bit; the same at config 2's full size (2e6 particles, 27 iterations, 176 s of
    # oracle time): profiles/round3_config2_full_relaxation_vs_oracle.txt, tools/config2_vs_oracle.py
    O.set_deviation(O.DEV_EXACT_BALL)
    try:
        o = O.Oracle(m, pos, ids)
        xlog = o.regularise()
        o.find_sph_quantities()
        x = o.particles()
    finally:
        O.set_deviation(0)
    assert [(a["it"], a["step"], a["err_max"]) for a in log] == [(b["it"], b["step"], b["err_max"]) for b in xlog]
    assert np.array_equal(p["id"], x["id"]) and np.array_equal(p["pos"], x["pos"])
    assert rel(p["hsml"], x["hsml"]).max() < 2e-6 and rel(p["rho"], x["rho"]).max() < 2e-6


def test_many_halos_substructure_shape(gpu):
    """BASELINE config 4 shape: two clusters + a population of small gas halos (Nhalos = 14); stresses the
    density model loop and the density contrast handled by the cell hierarchy."""
    n = 40_000
    m = M.with_subhalos(M.preset("merger", n), 12, n)
    pos, ids = M.sample_gas(m, n, seed=31)
    gpu.set_model(m)
    gpu.upload(pos, ids)
    log = gpu.Regularise_sph_particles(max_iter=5)
    o = O.Oracle(m, pos, ids)
    olog = o.regularise(max_iter=5)
    assert len(log) == len(olog) == 6
    for a, b in zip(log, olog):
        assert a["step"] == b["step"] and a["err_mean"] == pytest.approx(b["err_mean"], rel=1e-5)
        assert a["err_max"] == pytest.approx(b["err_max"], rel=1e-3)
    p, q = gpu.particles(), o.particles()
    assert np.array_equal(p["id"], q["id"])
    assert (np.abs(p["pos"] - q["pos"]).max(axis=1) / q["hsml"]).max() < TOL_POS
    assert rel(p["rho_model"], q["rho_model"]).max() < 1e-4      # evaluated at positions that agree to T2


# ------------------------------------------------------------------ T2: curl

def test_curl_of_vector_potential(gpu, golden_case):
    c = golden_case
    gpu.set_model(c["model"])
    gpu.upload(c["pos"], c["ids"])
    gpu.Find_sph_quantities()
    b = gpu.Bfld_from_rotA_SPH(c["c_apot"])
    scale = np.abs(c["c_bfld"]).max()
    assert np.abs(b - c["c_bfld"]).max() < 1e-5 * scale


# ------------------------------------------------------------------ collectives (1-rank communicator)

def test_rccl_code_path_single_rank(golden_case):
    """The RCCL all-gather / all-reduce calls of the sharded path, exercised with a 1-rank
    communicator on the one GPU of this box: results must equal the communicator-free path."""
    c = golden_case
    g = binding.TcGpu(0, options={"force_comm": 1})
    try:
        # every RCCL entry point the library binds, incl. the ghost exchange's grouped ncclSend / ncclRecv (to the own
        # rank here), with the payload checked
        assert g._L.tcgpu_debug_comm_selftest(g._h) == 0, g.last_error() if hasattr(g, "last_error") else "self-test"
        g.set_model(c["model"])
        g.upload(c["pos"], c["ids"])
        log = g.Regularise_sph_particles(max_iter=3)
        p = g.particles()
    finally:
        g.close()
    assert len(log) == len(c["r_log"])
    for l, w in zip(log, c["r_log"]):
        assert l["err_mean"] == pytest.approx(w[2], rel=1e-5)
    assert np.array_equal(p["id"], c["r_ids"])
    assert np.array_equal(p["pos"], c["r_pos"])


@pytest.mark.parametrize("nranks,n,ghosts", [(2, 20011, 2), (3, 20011, 1), (3, 20011, 0), (8, 60013, 2), (4, 2_000_003, 1)])
def test_sharded_path_with_loopback_ranks(nranks, n, ghosts):
    """The multi-GPU control flow -- Peano-range shards of the global order, per-rank local sets (own range +
    ghost shell from the interest mask) with their own sort / cell table / mirror, the ghost exchange (pyramids
    all-gathered, only the particles inside a receiver's pyramid sent: ghosts = 2 always, 1 when it moves fewer bytes
    than every position to every rank, 0 never), exact all-reduced sums -- run by `nranks` host threads on this one GPU through the loopback
    communicator: every rank must end with the single-rank result, bit for bit, log included."""
    import threading
    # n is not a multiple of nranks: padded tail shard
    m = M.preset("merger", n)
    pos, ids = M.sample_gas(m, n, seed=23)
    g1 = binding.TcGpu(0)
    g1.set_model(m)
    g1.upload(pos, ids)
    log1 = g1.Regularise_sph_particles(max_iter=4)
    g1.Find_sph_quantities()
    p1 = g1.particles()
    g1.close()

    ctxs = [binding.TcGpu(0, options={"ghost_exchange": ghosts}) for _ in range(nranks)]
    binding.loopback_group(ctxs)
    out = [None] * nranks

    def run(r):
        try:
            g = ctxs[r]
            g.set_model(m)
            g.upload(pos, ids)
            log = g.Regularise_sph_particles(max_iter=4)
            info = g.local_set_info()
            nb = g.comm_bytes()
            g.Find_sph_quantities()
            out[r] = (log, g.particles(), info, nb)
        except Exception as e:                      # pragma: no cover
            out[r] = e
    th = [threading.Thread(target=run, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    for c in ctxs:
        c.close()
    for r in range(nranks):
        assert not isinstance(out[r], Exception) and out[r] is not None, out[r]
        log, p, info, nbytes = out[r]
        assert len(log) == len(log1)
        for a, b in zip(log, log1):
            assert a["step"] == b["step"] and a["err_mean"] == b["err_mean"] and a["err_max"] == b["err_max"]
        for k in ("id", "pos", "hsml", "rho", "varhsmlfac", "rho_model"):
            assert np.array_equal(p[k], p1[k]), (r, k)
        # the warm passes ran on a local set, not on everything: own range < local set < all particles
        assert info["nown"] == min((r + 1) * -(-n // nranks), n) - r * -(-n // nranks)
        assert info["nown"] < info["nloc"] <= n and info["retries"] <= 2
        assert nbytes > 0
    if nranks == 8:
        assert min(o[2]["nloc"] for o in out) < 0.8 * n          # at least the core ranks work on a compact set
    if n > 1_000_000:
        # BASELINE config 2 size split four ways (a strong-scaling split: 5e5 particles per rank).  The ghost shell
        # (1.86 x hsml, hsml up to a fifth of the box in the outskirts at this N) is thick compared with such a shard:
        # measured 0.83 n on the most compact rank, everything on the ranks owning the outskirts
        assert min(o[2]["nloc"] for o in out) < 0.9 * n


def test_sharded_rank_memory_follows_the_local_set():
    """Weak-scaling split (4 loopback ranks x 1e6 particles): what a rank HOLDS once it is in steady state -- lists,
    tables and mirror sized by its local set; the global arrays are still sized by all N (memory is
    not sharded) -- stays within 2.5 x what one rank alone holds for 1e6 particles (measured: 20.1 against 8.2 GB at
    8 x 2e6, 2.4 x; 34.7 GB before the lists of a repeated pass, the full-set mirror and two unused arrays were dropped)."""
    import threading
    per, R = 1_000_000, 4

    def held(nranks, n, seed):
        m = M.preset("merger", n)
        pos, ids = M.sample_gas(m, n, seed=seed)
        base = binding.device_memory_used()
        ctxs = [binding.TcGpu(0) for _ in range(nranks)]
        if nranks > 1:
            binding.loopback_group(ctxs)
        err = []

        def run(g):
            try:
                g.set_model(m); g.upload(pos, ids)
                g.Regularise_sph_particles(max_iter=3)
            except Exception as e:                  # pragma: no cover
                err.append(e)
        th = [threading.Thread(target=run, args=(g,)) for g in ctxs]
        [t.start() for t in th]
        [t.join(timeout=600) for t in th]
        used = binding.device_memory_used() - base
        info = [g.local_set_info() for g in ctxs]
        [g.close() for g in ctxs]
        assert not err, err
        return used / nranks, info

    one, _ = held(1, per, 5)
    each, info = held(R, R * per, 5)
    assert max(i["nloc"] for i in info) < 0.6 * R * per          # steady state really runs on local sets
    print("held per rank: %.2f GB at %d x %d, %.2f GB at 1 x %d: %.2f x" % (each / 1e9, R, per, one / 1e9, per, each / one))
    assert each < 2.5 * one, (each / 1e9, one / 1e9)


def test_failure_of_one_rank_before_the_ghost_exchange_fails_all(golden_case):
    """A failure that only ONE rank sees while it prepares the ghost exchange (an allocation sized by its own counts,
    its table layout, a launch) used to make that rank return alone while the others posted sends and receives to it:
    a hang.  The ranks now agree on the status before any point-to-point traffic (agree_status, api.hip): with a failure
    injected on rank 1 every rank must return an error -- and return at all."""
    import threading
    n, nranks = 20011, 3
    m = M.preset("merger", n)
    pos, ids = M.sample_gas(m, n, seed=23)
    ctxs = [binding.TcGpu(0, options={"ghost_exchange": 2}) for _ in range(nranks)]
    binding.loopback_group(ctxs)
    out = [None] * nranks

    def run(r):
        try:
            g = ctxs[r]
            g.set_model(m)
            g.upload(pos, ids)
            g.Regularise_sph_particles(max_iter=1)            # cold pass + the first warm passes: shards become compact
            g.set_option("debug_fail_rank", 2)                # rank 1 fails in its next ghost exchange
            g.Regularise_sph_particles(max_iter=1)
            out[r] = "no error"
        except Exception as e:
            out[r] = e
    th = [threading.Thread(target=run, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a rank is still waiting for the one that failed"
    for c in ctxs:
        c.close()
    for r in range(nranks):
        assert isinstance(out[r], Exception), (r, out[r])
    assert "injected failure" in str(out[1])
    assert "another rank failed" in str(out[0]) and "another rank failed" in str(out[2])


def test_curl_larger_case_vs_oracle(gpu):
    """BASELINE config 5 shape (curl of the Bonafede vector potential) at a size the oracle does in seconds."""
    n = 60_000
    m = M.preset("merger", n)
    pos, ids = M.sample_gas(m, n, seed=41)
    gpu.set_model(m)
    gpu.upload(pos, ids)
    gpu.Find_sph_quantities()
    p = gpu.particles()
    rm = gpu.Global_density_model()
    a = (rm / np.float32(m.halos[0].rho0)) ** np.float32(0.5)          # magnetic_field.c:57, eta = 0.5
    apot = np.stack([a, a, a], axis=1).astype(np.float32)
    b = gpu.Bfld_from_rotA_SPH(apot)
    o = O.Oracle(m, pos, ids)
    o.find_sph_quantities()
    q = o.particles()
    assert np.array_equal(p["id"], q["id"])
    o.set_apot(apot)
    bo = o.bfld_from_rotA()
    assert np.abs(b - bo).max() < 1e-5 * np.abs(bo).max()


# ------------------------------------------------------------------ error behaviour

def test_curl_literal_path_matches_staged_path():
    """k_curl sets particles whose ball holds >= NGBMAX neighbours aside for k_curl_slow, the literal pair-by-pair
    form with the reference's list truncation (tree.c:91-92); converged smoothing lengths never get there, so the
    option "curl_literal" sends EVERY particle through it: same B as the staged kernel, A in the w lane or not."""
    n = 40000
    m = M.preset("merger", n)
    pos, ids = M.sample_gas(m, n, seed=12)
    g = binding.TcGpu(0)
    try:
        g.set_model(m)
        g.upload(pos, ids)
        g.Find_sph_quantities()
        rng = np.random.default_rng(0)
        a1 = rng.random(n).astype(np.float32)
        equal = np.repeat(a1[:, None], 3, axis=1)                       # Ax = Ay = Az: rides in the w lane
        general = rng.random((n, 3)).astype(np.float32)                # the 3-component path
        for apot in (equal, general):
            g.set_option("curl_literal", 0)
            b_fast = g.Bfld_from_rotA_SPH(apot)
            g.set_option("curl_literal", 1)
            b_lit = g.Bfld_from_rotA_SPH(apot)
            scale = np.abs(b_lit).max()
            assert np.isfinite(b_fast).all() and np.abs(b_fast - b_lit).max() < 1e-5 * scale
    finally:
        g.close()


def test_out_of_box_coordinate_is_reported(gpu):
    n = 2000
    m = M.preset("single", n)
    pos, ids = M.sample_gas(m, n, seed=2)
    pos[5, 0] = -1.0
    gpu.set_model(m)
    gpu.upload(pos, ids)
    with pytest.raises(binding.TcGpuError, match="COORD_RANGE"):
        gpu.Find_sph_quantities()
    pos[5, 0] = np.nan
    gpu.upload(pos, ids)
    with pytest.raises(binding.TcGpuError, match="COORD_RANGE"):
        gpu.Find_sph_quantities()


# ------------------------------------------------------------------ full-size properties (config 2)

def test_ngbmax_overflow_paths(gpu):
    """A tight clump where the model density is low: thousands of particles have more than NGBMAX = 2360
    neighbours inside their sweep radius.  The reference truncates such a list to the first 2360 hits in ascending
    index (tree.c:91-92) without noticing (wvt_relax.c:135), and its density pass divides hsml by 1.24 when a
    query fills the list (sph.c:42-47).  Plain kernels and the fused kernel's fall-backs against the oracle."""
    n = 20000
    m = M.preset("single", n)
    pos, ids = M.sample_gas(m, n, seed=12)
    rng = np.random.default_rng(3)
    c = np.float32(m.boxsize) * np.float32([0.80, 0.78, 0.76])
    k = 3200
    pos[:k] = (c + rng.normal(0, 0.004 * m.boxsize, (k, 3))).astype(np.float32)
    o = O.Oracle(m, pos, ids)
    o.find_sph_quantities()
    q1 = o.particles()
    ohs, ode = o.wvt_step(0.0085, move=False)
    # the case does what it is meant to do: count sweep neighbours of the clump by brute force
    P = q1["pos"].astype(np.float64)
    far = 0
    for i in np.where(np.abs(P - np.float64(c)).max(axis=1) < 0.004 * m.boxsize)[0][:20]:
        d = P - P[i]
        d -= m.boxsize * np.round(d / m.boxsize)
        far += ((d * d).sum(axis=1) < (float(ohs[i]) * m.boxsize) ** 2).sum() >= 2360
    assert far >= 10

    gpu.set_model(m)
    gpu.upload(pos, ids)
    gpu.Find_sph_quantities()                                   # cold: plain density kernel
    p1 = gpu.particles()
    assert np.array_equal(p1["id"], q1["id"])
    assert rel(p1["hsml"], q1["hsml"]).max() < 1e-6 and rel(p1["rho"], q1["rho"]).max() < 1e-6
    hs, de = gpu.wvt_step(0.0085, move=False)                   # the exact sweep: truncated lists and all
    assert np.array_equal(hs, ohs) and np.array_equal(de, ode)

    o.find_sph_quantities()                                     # warm pass
    q2 = o.particles()
    ohs2, ode2 = o.wvt_step(0.0085, move=False)
    gpu.density_error()                                         # warm: fused kernel, sweep sums included
    p2 = gpu.particles()
    assert rel(p2["hsml"], q2["hsml"]).max() < 1e-6 and rel(p2["rho"], q2["rho"]).max() < 1e-6
    hs2, de2 = gpu.wvt_step(0.0085, move=False)
    assert np.array_equal(hs2, ohs2) and np.array_equal(de2, ode2)
    g1 = binding.TcGpu(0, options={"sweep": 1})                 # round 2's sweep: the index-threshold bisection of k_wvt / wvt_sum
    g1.set_model(m); g1.upload(pos, ids); g1.Find_sph_quantities(); g1.density_error()
    hs3, de3 = g1.wvt_step(0.0085, move=False)
    g1.close()
    assert rel(hs3, ohs2).max() < 3e-7 and np.abs(de3 - ode2).max() < 2e-6 * np.abs(ode2).max()


def test_particles_on_the_box_faces(gpu):
    """Coordinates exactly 0 and exactly boxsize are legal (the wrap of wvt_relax.c:190-212 uses `>`).  A
    coordinate == boxsize is keyed like 0 (X = 2^63, peano.c:134-136) while the position sits on the far face.
    The reference's tree places a node by the *position* of the particle that creates it but fills it by *key*
    (tree.c:297-306, :273-280), so such a particle mis-centres its node and the reference's own ball queries then
    lose ordinary neighbours stored in it.  That artefact is NOT reproduced: the HIP path returns the exact
    predicate set (the particle is kept out of the cell table / mirror and tested by brute force), which is what
    the reference's brute-force twin (wvt_relax.c:296-340) returns.  DESIGN.md section 5."""
    n = 30000
    m = M.preset("single", n)
    pos, ids = M.sample_gas(m, n, seed=31)
    rng = np.random.default_rng(5)
    box = np.float32(m.boxsize)
    for k, i in enumerate(rng.choice(n, 60, replace=False)):
        pos[i, k % 3] = box if k % 2 else np.float32(0)
    pos[7] = box                                            # a corner
    gpu.set_model(m)
    gpu.upload(pos, ids)
    gpu.Find_sph_quantities()
    gpu.Find_sph_quantities()                               # second pass: warm, i.e. the fused kernel
    p = gpu.particles()
    assert np.all(np.isfinite(p["hsml"])) and p["hsml"].min() > 0 and p["rho"].min() > 0
    P = p["pos"]
    face = np.where(((P == box) | (P == 0)).any(axis=1))[0]
    assert len(face) >= 50
    o = O.Oracle(m, P, p["id"])
    o.build_tree()
    tree_differs = 0
    Pd = P.astype(np.float64)
    for i in face[:40]:
        d = Pd - Pd[i]
        d -= m.boxsize * np.round(d / m.boxsize)
        r = np.sqrt((d * d).sum(axis=1))
        for j in [int(i)] + [int(x) for x in np.argsort(r)[1:3]]:        # the face particle and two neighbours
            h = float(p["hsml"][j])
            a = gpu.Find_ngb_tree(j, h)
            assert np.array_equal(a, o.find_ngb_simple(j, h))
            tree_differs += not np.array_equal(a, o.find_ngb_tree(j, h))
        # kernel-weighted neighbour number of the face particle itself: 295 +- 0.05 over the true neighbour set
        h = float(p["hsml"][i])
        u = np.minimum(r / h, 1.0)
        w = 1365.0 / (64 * np.pi) / h ** 3 * (1 - u) ** 8 * (1 + 8 * u + 25 * u * u + 32 * u ** 3)
        assert abs((4.18879032135009765 * w * h ** 3).sum() - 295) < 0.06
    assert tree_differs > 0          # the reference artefact exists (else this docstring is out of date)
    log = gpu.Regularise_sph_particles(max_iter=2)          # the whole relaxation path runs with them present
    assert len(log) == 3 and all(np.isfinite(l["err_mean"]) for l in log) and log[2]["err_mean"] < log[0]["err_mean"]


def test_plain_kernels_match_oracle():
    """Option "fuse" = 0: one plain kernel per reference loop (k_density for sph.c:19-72, k_wvt for
    wvt_relax.c:126-171) instead of the fused kernel -- same results, same control flow."""
    n = 20000
    m = M.preset("merger", n)
    pos, ids = M.sample_gas(m, n, seed=41)
    g = binding.TcGpu(0)
    g.set_option("fuse", 0)
    g.set_option("stats", 1)
    g.set_model(m)
    g.upload(pos, ids)
    o = O.Oracle(m, pos, ids)
    for _ in range(2):                                       # cold pass, then warm pass
        g.Find_sph_quantities()
        o.find_sph_quantities()
        p, q = g.particles(), o.particles()
        a, b = g.density_stats(), o.last_stats()
        assert np.array_equal(p["id"], q["id"])
        assert rel(p["hsml"], q["hsml"]).max() < 1e-6 and rel(p["rho"], q["rho"]).max() < 1e-6
        assert a["queries"] == pytest.approx(b["queries"], rel=1e-3)
        assert a["solver_iters"] == pytest.approx(b["solver_iters"], rel=1e-3)
    log = g.Regularise_sph_particles(max_iter=3)
    olog = o.regularise(max_iter=3)
    g.Find_sph_quantities()
    o.find_sph_quantities()
    assert len(log) == len(olog)
    for x, y in zip(log, olog):
        assert x["it"] == y["it"] and x["step"] == y["step"]
        assert x["err_mean"] == pytest.approx(y["err_mean"], rel=1e-5)
    p, q = g.particles(), o.particles()
    g.close()
    assert np.array_equal(p["id"], q["id"])
    assert (np.abs(p["pos"] - q["pos"]).max(axis=1) / q["hsml"]).max() < TOL_POS
    assert rel(p["hsml"], q["hsml"]).max() < TOL_HSML and rel(p["rho"], q["rho"]).max() < TOL_RHO


def test_row_run_path_matches_cell_path():
    """The fused kernel has two candidate producers: cell by cell over the Peano-ordered table (any ball) and
    run by run over the row-major mirror (interior balls at a mirrored level, the default where it applies).
    Both visit the same cells, so candidate counts and control flow are identical; only the order of the hit
    lists -- hence the f64 summation order -- differs."""
    n = 200_000
    m = M.preset("merger", n)
    pos, ids = M.sample_gas(m, n, seed=77)
    out = []
    for rows in (1, 0):
        g = binding.TcGpu(0)
        g.set_option("rows", rows)
        g.set_option("stats", 1)
        g.set_option("timing", 1)
        g.set_model(m)
        g.upload(pos, ids)
        log = g.Regularise_sph_particles(max_iter=3)
        g.Find_sph_quantities()
        st = g.density_stats()
        t = g.phase_times()
        out.append((log, g.particles(), st, t))
        g.close()
    (la, pa, sa, ta), (lb, pb, sb, tb) = out
    assert ta["mirror"][1] > 0 and tb.get("mirror", (0, 0))[1] == 0          # the mirror was built / was not
    assert sa["candidates"] == sb["candidates"] and sa["queries"] == sb["queries"]
    assert sa["solver_iters"] == pytest.approx(sb["solver_iters"], rel=1e-5)
    assert len(la) == len(lb)
    for a, b in zip(la, lb):
        assert a["err_mean"] == pytest.approx(b["err_mean"], rel=1e-6)
    assert np.array_equal(pa["id"], pb["id"])
    assert (pa["hsml"] == pb["hsml"]).mean() > 0.99 and rel(pa["hsml"], pb["hsml"]).max() < 1e-5
    assert rel(pa["rho"], pb["rho"]).max() < 1e-5
    assert (np.abs(pa["pos"] - pb["pos"]).max(axis=1) / pb["hsml"]).max() < 1e-4


def test_full_size_properties(gpu):
    """BASELINE config 2 size (2e6 gas, 2-cluster merger): size-independent properties."""
    n = 2_000_000
    m = M.preset("merger", n)
    pos, ids = M.sample_gas(m, n, seed=14041981)
    gpu.set_model(m)
    gpu.upload(pos, ids)
    log = gpu.Regularise_sph_particles(max_iter=2)
    assert len(log) == 3 and log[2]["err_mean"] < log[0]["err_mean"]
    hi, lo = gpu.Sort_Particles_By_Peano_Key()
    assert np.all(hi[1:] >= hi[:-1])                                   # sortedness
    p = gpu.particles()
    assert np.array_equal(np.sort(p["id"]), np.sort(ids))               # ids are a permutation
    assert p["pos"].min() >= 0 and p["pos"].max() <= m.boxsize
    assert np.all(np.isfinite(p["hsml"])) and p["hsml"].min() > 0 and p["rho"].min() > 0
    # kernel-weighted neighbour number of sampled particles is 295 +- 0.05 (brute force, numpy)
    gpu.Find_sph_quantities()
    p = gpu.particles()
    P = p["pos"].astype(np.float64)
    for i in range(0, n, n // 16):
        d = P - P[i]
        d -= m.boxsize * np.round(d / m.boxsize)
        r = np.sqrt((d * d).sum(axis=1))
        h = float(p["hsml"][i])
        u = np.minimum(r / h, 1.0)
        w = 1365.0 / (64 * np.pi) / h ** 3 * (1 - u) ** 8 * (1 + 8 * u + 25 * u * u + 32 * u ** 3)
        assert abs((4.18879032135009765 * w * h ** 3).sum() - 295) < 0.06
    # oracle spot check at full size: neighbour sets of a few particles (brute force)
    o = O.Oracle(m, p["pos"], p["id"])
    for i in (0, n // 3, n - 1):
        a = gpu.Find_ngb_tree(i, float(p["hsml"][i]))
        b = o.find_ngb_simple(i, float(p["hsml"][i]))
        assert np.array_equal(a, b)
