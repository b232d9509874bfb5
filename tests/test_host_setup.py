"""CPU tests of the native set-up stages (toycluster_amd/host/tc_setup.c), pinned by the values the survey
recorded from the running reference (closed-form scalars to 6 digits; Rho0 / Mpart to
the reference integrator's 1e-6) and by statistical properties of the sampled particles."""
import os

import numpy as np
import pytest

from toycluster_amd import hostio, model as M

PAR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cluster.par")


def test_config1_scalars_match_survey_probe():
    s = hostio.setup_system(PAR, {"ntotal": 200000})            # BASELINE config 1
    h = s.halo[0]
    assert s.nhalos == 1 and s.boxsize == 13923.0
    assert h.r200 == pytest.approx(1856.45, rel=5e-6)
    assert h.a_hernq == pytest.approx(783.333, rel=5e-6)
    assert h.c_nfw == pytest.approx(2.42309, rel=5e-6)
    assert h.beta == 0.54
    assert h.rcore == pytest.approx(255.384, rel=5e-6)
    assert h.rcut == pytest.approx(2599.04, rel=5e-6)
    assert h.rho0 == pytest.approx(7.44228e-6, rel=2e-5)        # depends on the integrator to ~1e-6
    assert s.mpart[0] == pytest.approx(0.317534, rel=2e-5)
    assert s.npart[0] == 100000
    assert s.mtotal == pytest.approx(171379, rel=2e-5)          # Param.Mtotal > 1e5: step stays 0.0085


def test_config2_scalars_match_survey_probe():
    s = hostio.setup_system(PAR, {"ntotal": 2000000, "mass_ratio": 0.3125})
    h0, h1 = s.halo[0], s.halo[1]
    assert s.nhalos == 2 and s.boxsize == 12716.0
    assert h0.r200 == pytest.approx(1695.58, rel=5e-6) and h1.r200 == pytest.approx(1150.63, rel=5e-6)
    assert h0.rcore == pytest.approx(227.181, rel=5e-6) and h1.rcore == pytest.approx(137.717, rel=5e-6)
    assert h0.rcut == pytest.approx(2373.81, rel=5e-6) and h1.rcut == pytest.approx(1610.88, rel=5e-6)
    assert h0.rho0 == pytest.approx(7.73628e-6, rel=2e-5) and h1.rho0 == pytest.approx(9.14311e-6, rel=2e-5)
    assert tuple(h0.d_com) == pytest.approx((-609.901, -11.9048, 0.0), rel=5e-6)
    assert tuple(h1.d_com) == pytest.approx((1951.68, 38.0952, 0.0), rel=5e-6)
    assert s.mpart[0] == pytest.approx(0.03014, rel=2e-4)
    assert s.npart[0] == h0.npart[0] + h1.npart[0] == 1000000
    # the python presets used for synthetic inputs carry the same scalars
    m = M.preset("merger", 1000000)
    assert m.mpart_gas == pytest.approx(s.mpart[0], rel=2e-5)
    assert m.halos[1].mass_gas == pytest.approx(h1.mass[0], rel=2e-5)


def test_sampled_gas_follows_the_model():
    s = hostio.setup_system(PAR, {"ntotal": 400000})
    pos, ids = hostio.sample_gas(s, nthreads=4)
    n = len(ids)
    assert n == s.npart[0] == 200000
    assert pos.min() >= 0 and pos.max() <= s.boxsize
    # ids: strided permutation of 1..n (ids.c:16-39)
    assert sorted(ids) == list(range(1, n + 1)) and ids[1] - ids[0] >= 128 and n % (ids[1] - ids[0]) == 0
    assert np.array_equal(ids, M.make_ids(n))
    # radial mass distribution inside the box-inscribed sphere = the model's M(<r) (KS distance)
    m = hostio.setup_to_model(s)
    h = m.halos[0]
    r = np.sqrt(((pos.astype(np.float64) - s.boxsize / 2) ** 2).sum(axis=1))
    inside = r < s.boxsize / 2
    rt, mt = M._mass_table(h, s.boxsize / 2)
    cdf_model = np.interp(np.sort(r[inside]), rt, mt) / mt[-1]
    cdf_emp = (np.arange(inside.sum()) + 0.5) / inside.sum()
    assert np.abs(cdf_model - cdf_emp).max() < 0.01
    # isotropy
    u = (pos[inside].astype(np.float64) - s.boxsize / 2) / r[inside, None]
    assert np.abs(u.mean(axis=0)).max() < 0.01
    # reproducible for a given thread count, different streams for another (main.c:15-26)
    pos2, _ = hostio.sample_gas(s, nthreads=4)
    pos3, _ = hostio.sample_gas(s, nthreads=3)
    assert np.array_equal(pos, pos2) and not np.array_equal(pos, pos3)


def test_double_beta_cool_core_setup():
    """The reference's -DDOUBLE_BETA_COOL_CORES build as the run-time switch of the native set-up (double_beta,
    Rho0_Fac, Rc_Fac; src/setup.c:567-615): a cuspy halo keeps rc = rs / 3 (not rs / 9), its profile gains the
    component rho0 * Rho0_Fac / (1 + (r Rc_Fac / rc)^2) / (1 + (r / rcut)^4), and Rho0 is normalised so that the
    whole profile -- component included -- holds the gas mass inside r200.  Parity unpinned (no reference run of
    that build exists); checked against an independent quadrature and the default build."""
    base = {"ntotal": 200000, "cuspy": 1}
    s0 = hostio.setup_system(PAR, base)                                   # default build: cool core = rc = rs / 9
    s1 = hostio.setup_system(PAR, dict(base, double_beta=1, rho0_fac=8.0, rc_fac=6.0))
    h0, h1 = s0.halo[0], s1.halo[0]
    assert h0.have_cuspy == 1 and h1.have_cuspy == 1
    assert h0.rcore == pytest.approx(h0.rs / 9) and h1.rcore == pytest.approx(h1.rs / 3)
    assert h0.rho0_cc == 0 and h0.rc_cc == 0
    assert h1.rho0_cc == h1.rho0 * 8.0 and h1.rc_cc == h1.rcore / 6.0
    m = hostio.setup_to_model(s1)
    assert m.rho0_fac == 8.0 and m.rc_fac == 6.0 and m.halos[0].have_cuspy == 1
    # gas mass inside r200 by an independent quadrature of the python restatement of the profile
    r = np.concatenate([[0.0], np.logspace(np.log10(h1.rcore * 1e-5), np.log10(h1.r200), 400001)])
    f = 4 * np.pi * r * r * M.gas_density_profile(r, m.halos[0], m.rho0_fac, m.rc_fac)
    mass = float((0.5 * (f[1:] + f[:-1]) * np.diff(r)).sum())
    assert mass == pytest.approx(h1.mass200[0], rel=2e-5)
    # the component is a sizeable part of that mass, so Rho0 is lower than it would be without it
    f0 = 4 * np.pi * r * r * M.gas_density_profile(r, m.halos[0])
    assert float((0.5 * (f0[1:] + f0[:-1]) * np.diff(r)).sum()) < 0.98 * mass
    # the sampler draws from the whole profile: more particles inside the cool core than the plain beta model gives
    pos, ids = hostio.sample_gas(s1, nthreads=4)
    rr = np.sqrt(((pos.astype(np.float64) - s1.boxsize / 2) ** 2).sum(axis=1))
    inside = (rr < h1.rc_cc).mean()
    mt = np.concatenate([[0.0], np.cumsum(0.5 * (f[1:] + f[:-1]) * np.diff(r))])
    f_s = 4 * np.pi * r * r * M.gas_density_profile(r, m.halos[0], m.rho0_fac, m.rc_fac)
    rs_grid = np.concatenate([[0.0], np.logspace(np.log10(h1.rcore * 1e-5), np.log10(h1.r_sample[0]), 400001)])
    fs = 4 * np.pi * rs_grid ** 2 * M.gas_density_profile(rs_grid, m.halos[0], m.rho0_fac, m.rc_fac)
    ms = np.concatenate([[0.0], np.cumsum(0.5 * (fs[1:] + fs[:-1]) * np.diff(rs_grid))])
    expect = np.interp(h1.rc_cc, rs_grid, ms) / ms[-1]
    assert inside == pytest.approx(expect, rel=0.1, abs=2e-4)


def test_double_beta_tags_are_mandatory_only_in_that_build(tmp_path, monkeypatch):
    """src/io.c:435-443: the -DDOUBLE_BETA_COOL_CORES build reads Rho0_Fac and Rc_Fac and stops when they are missing;
    the default build does not know them.  The reference's sample parameter file carries both (50 and 40)."""
    txt = open(PAR).read()
    p1 = tmp_path / "plain.par"
    p1.write_text("\n".join(l for l in txt.splitlines() if not l.startswith(("Rho0_Fac", "Rc_Fac"))) + "\n")
    monkeypatch.delenv("TC_DOUBLE_BETA", raising=False)
    s = hostio.setup_system(PAR)
    assert s.par.double_beta == 0 and s.par.rho0_fac == 0 and s.halo[0].rho0_cc == 0
    assert hostio.setup_system(str(p1)).par.double_beta == 0
    monkeypatch.setenv("TC_DOUBLE_BETA", "1")
    with pytest.raises(RuntimeError, match="Rho0_Fac"):
        hostio.setup_system(str(p1))
    s = hostio.setup_system(PAR)
    assert s.par.double_beta == 1 and s.par.rho0_fac == 50 and s.par.rc_fac == 40
    assert s.halo[0].rho0_cc == 0                                         # Cuspy = 0 in the sample file: no cool core
    s = hostio.setup_system(PAR, {"cuspy": 1})
    assert s.halo[0].rho0_cc == s.halo[0].rho0 * 50 and s.halo[0].rc_cc == s.halo[0].rcore / 40
