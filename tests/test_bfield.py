"""CPU tests of the wrapper around the SPH curl (src/magnetic_field.c:33-131): vector potential,
normalisation, and the limiter with its subhalo case -- C host library (host/tc_bfield.c) against the oracle's
literal restatement (oracle/tc_oracle.c).  Parity unpinned: the reference holds no fixture for this step."""
import numpy as np
import pytest

from oracle import oracle as O
from toycluster_amd import hostio, model as M


def _case(n=20000, nsub=6, seed=11):
    m = M.with_subhalos(M.preset("merger", n), nsub, n)
    pos, _ = M.sample_gas(m, n, seed=seed)
    rng = np.random.default_rng(seed)
    # dark-matter sampling radii (Halo[i].R_Sample[1]): cluster-sized for the two clusters, small for the subhalos
    rdm = np.array([1.2 * h.r_sample if k < 2 else 0.6 * h.r_sample for k, h in enumerate(m.halos)])
    bf = (rng.standard_normal((n, 3)) * np.exp(2.5 * rng.standard_normal((n, 1)))).astype(np.float32)
    return m, pos, rdm, bf


def test_vector_potential_matches_oracle():
    m, pos, _, _ = _case(5000)
    a = hostio.set_magnetic_vector_potential(m, pos, 0.5)
    b = O.set_vector_potential(m, pos, 0.5)
    assert np.array_equal(a, b)
    assert np.all(a[:, 0] == a[:, 1]) and np.all(a[:, 1] == a[:, 2])            # magnetic_field.c:63-65
    assert 0 < a.min() and a.max() <= 1.0


@pytest.mark.parametrize("sub_first", [2, 1])
def test_normalisation_and_limits_match_oracle(sub_first):
    """Three or more halos: particles the reference's Halo_containing(ipart, ...) call (index passed as type ->
    dark-matter branch, magnetic_field.c:109, positions.c:343-362) puts into a halo with index > 1 are limited to
    2e-6 G, the others to 18e-6 G."""
    m, pos, rdm, bf = _case()
    got, norm, cnt = hostio.normalise_magnetic_field(m, pos, bf, 2e-4, r_sample_dm=rdm, sub_first=sub_first)
    want, onorm, ocnt = O.normalise_magnetic_field(m, pos, bf, 2e-4, rdm, sub_first=sub_first)
    assert norm == onorm and cnt == ocnt
    assert np.array_equal(got, want)
    mag = np.sqrt((got.astype(np.float64) ** 2).sum(axis=1))
    assert mag.max() <= 18e-6 * (1 + 1e-6)
    # some particles sit in a subhalo's DM sampling radius and are held to 2e-6
    low = (mag > 1.99e-6) & (mag < 2.01e-6)
    assert cnt > 0 and low.sum() > 0
    # default build (no dark-matter radii known, e.g. from a state file): only the 18e-6 limit
    got0, _, cnt0 = hostio.normalise_magnetic_field(m, pos, bf, 2e-4)
    mag0 = np.sqrt((got0.astype(np.float64) ** 2).sum(axis=1))
    assert cnt0 < cnt and mag0.max() <= 18e-6 * (1 + 1e-6) and (mag0 > 2.01e-6).sum() > (mag > 2.01e-6).sum()


def test_two_halo_default_build_has_only_the_cluster_limit():
    n = 8000
    m = M.preset("merger", n)
    pos, _ = M.sample_gas(m, n, seed=2)
    rng = np.random.default_rng(0)
    bf = (rng.standard_normal((n, 3)) * np.exp(2.0 * rng.standard_normal((n, 1)))).astype(np.float32)
    rdm = np.array([1.2 * h.r_sample for h in m.halos])
    got, norm, cnt = hostio.normalise_magnetic_field(m, pos, bf, 5e-6, r_sample_dm=rdm, sub_first=2)
    want, onorm, ocnt = O.normalise_magnetic_field(m, pos, bf, 5e-6, rdm, sub_first=2)
    assert np.array_equal(got, want) and norm == onorm and cnt == ocnt
    bmax = np.sqrt((bf.astype(np.float64) ** 2).sum(axis=1)).max()
    assert norm == pytest.approx(5e-6 / bmax / np.sqrt(3), rel=1e-6)              # magnetic_field.c:87-89
