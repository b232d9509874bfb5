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
