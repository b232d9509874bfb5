"""Cluster model (the scalars of the reference's `Param` / `Halo[]` that the SPH/WVT path reads)
and a seeded sampler of synthetic gas-particle initial positions.

Reference facts restated here (reference = /root/reference):
  * what the hot path reads: SURVEY.md Appendix A "hot-path state file"
    (globals.h:94-159, wvt_relax.c:227-256, setup.c:598-615);
  * how gas positions are drawn: positions.c:90-133 (isotropic direction,
    r = M^-1(u * M_gas), reject outside the box / where another halo is denser,
    positions.c:333-388), origin shift + wrap setup.c:427-500, ids ids.c:8-44.

The halo scalars of the two presets are the survey's probe values (SURVEY.md
Appendix A); Rho0 / Mpart depend on the reference's GSL integrator to ~1e-6,
which is why inputs are exchanged as a state (model + positions), not as cluster.par.
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np

DESNNGB = 295
FOURPITHIRD = 4.18879032135009765


@dataclass
class Halo:
    rho0: float
    beta: float
    rcore: float
    rcut: float
    d_com: tuple = (0.0, 0.0, 0.0)
    r_sample: float = 0.0      # Halo[i].R_Sample[0] (gas)
    mass_gas: float = 1.0      # Halo[i].Mass[0]; 0 => halo skipped by the density model
    have_cuspy: int = 0


@dataclass
class ClusterModel:
    boxsize: float
    halos: List[Halo]
    mpart_gas: float = 0.0     # Param.Mpart[0]
    mtotal: float = 2.0e5      # Param.Mtotal (only the `< 1e5` test of wvt_relax.c:53 reads it)
    bfld_eta: float = 0.5
    name: str = ""
    rho0_fac: float = 0.0      # Param.Rho0_Fac / Param.Rc_Fac of the reference's -DDOUBLE_BETA_COOL_CORES build
    rc_fac: float = 0.0        # (0 = the default build: no cool-core component)

    @property
    def nhalos(self):
        return len(self.halos)


def gas_density_profile(r, h: Halo, rho0_fac=0.0, rc_fac=0.0):
    """setup.c:598-615 (beta model with r^4 cut-off; the cool-core term of -DDOUBLE_BETA_COOL_CORES only when the
    factors are given and the halo is cuspy)."""
    r = np.asarray(r, dtype=np.float64)
    rho = h.rho0 * (1 + (r / h.rcore) ** 2) ** (-1.5 * h.beta) / (1 + (r / h.rcut) ** 4)
    if rho0_fac and rc_fac and h.have_cuspy:
        rho = rho + h.rho0 * rho0_fac / (1 + (r / (h.rcore / rc_fac)) ** 2) / (1 + (r / h.rcut) ** 4)
    return rho


def _mass_table(h: Halo, rmax, npts=20000):
    r = np.concatenate([[0.0], np.logspace(np.log10(h.rcore * 1e-4), np.log10(rmax), npts)])
    f = 4 * np.pi * r * r * gas_density_profile(r, h)
    m = np.concatenate([[0.0], np.cumsum(0.5 * (f[1:] + f[:-1]) * np.diff(r))])
    return r, m


def preset(name, npart):
    """Model scalars for BASELINE.json configs 1 and 2 (SURVEY.md Appendix A sanity values)."""
    if name in ("single", "cfg1"):
        box = 13923.0
        h0 = Halo(rho0=7.44228e-6, beta=0.54, rcore=255.384, rcut=2599.04, d_com=(0, 0, 0),
                  r_sample=np.sqrt(3) * box / 2)
        m = ClusterModel(boxsize=box, halos=[h0], mtotal=171379.0, name="single")
    elif name in ("merger", "cfg2"):
        box = 12716.0
        h0 = Halo(rho0=7.73628e-6, beta=0.54, rcore=227.181, rcut=2373.81,
                  d_com=(-609.901, -11.9048, 0.0), r_sample=np.sqrt(3) * box / 2)
        h1 = Halo(rho0=9.14311e-6, beta=0.54, rcore=137.717, rcut=1610.88,
                  d_com=(1951.68, 38.0952, 0.0), r_sample=1.8 * 1150.63)
        m = ClusterModel(boxsize=box, halos=[h0, h1], mtotal=2.0e5, name="merger")
    else:
        raise ValueError("unknown preset %r" % name)
    for h in m.halos:                      # setup.c:97-101: Mass[0] = M_gas(<R_Sample[0])
        r, mt = _mass_table(h, h.r_sample)
        h.mass_gas = float(mt[-1])
    m.mpart_gas = sum(h.mass_gas for h in m.halos) / npart      # setup.c:190-193
    return m


def with_subhalos(model: ClusterModel, nsub, npart, seed=4):
    """A BASELINE-config-4-shaped model: the preset plus `nsub` small gas halos inside the main cluster
    (stand-in for the Giocoli substructure population of src/substructure.c, which is out of scope: the hot
    path only sees more entries in Halo[] and a clumpier density field)."""
    rng = np.random.default_rng(seed)
    h0 = model.halos[0]
    halos = list(model.halos)
    for _ in range(nsub):
        r = 1500.0 * rng.random() ** (1 / 3)
        cth, phi = 2 * rng.random() - 1, 2 * np.pi * rng.random()
        sth = np.sqrt(1 - cth * cth)
        d = (h0.d_com[0] + r * sth * np.cos(phi), h0.d_com[1] + r * sth * np.sin(phi), h0.d_com[2] + r * cth)
        rc = 20.0 + 40.0 * rng.random()
        halos.append(Halo(rho0=h0.rho0 * (2 + 6 * rng.random()), beta=h0.beta, rcore=rc, rcut=6 * rc, d_com=d,
                          r_sample=10 * rc))
    m = ClusterModel(boxsize=model.boxsize, halos=halos, mtotal=model.mtotal, name=model.name + "+sub%d" % nsub)
    for h in m.halos:
        r, mt = _mass_table(h, h.r_sample)
        h.mass_gas = float(mt[-1])
    m.mpart_gas = sum(h.mass_gas for h in m.halos) / npart
    return m


def make_ids(n):
    """ids.c:16-39: gas ids strided by the smallest divisor >= 128 of n."""
    delta = 127
    while True:
        delta += 1
        if n % delta == 0:
            break
    ids = np.empty(n, np.int64)
    k, start, idv = 0, 1, 1 - delta
    # vectorised form of the reference loop: runs start, start+delta, ... <= n, then start+1, ...
    out = []
    s = 1
    while k < n:
        run = np.arange(s, n + 1, delta, dtype=np.int64)
        take = min(len(run), n - k)
        out.append(run[:take])
        k += take
        s += 1
    return np.concatenate(out).astype(np.int32)


def sample_gas(model: ClusterModel, npart, seed=14041981):
    """Synthetic gas positions in [0, box], distributed like the reference's sampler
    (positions.c:90-133 + setup.c:427-500) but drawn from numpy's PCG64 with a stated seed
    (the reference's per-thread erand48 streams are not reproduced)."""
    rng = np.random.default_rng(seed)
    box = model.boxsize
    half = box / 2
    mtot = sum(h.mass_gas for h in model.halos)
    counts = [int(round(h.mass_gas / (mtot / npart))) for h in model.halos]   # setup.c:197
    counts[0] += npart - sum(counts)
    chunks = []
    for i, (h, cnt) in enumerate(zip(model.halos, counts)):
        r_tab, m_tab = _mass_table(h, h.r_sample)
        got = []
        need = cnt
        while need > 0:
            k = int(need * 1.5) + 1024
            cth = 2 * rng.random(k) - 1
            phi = 2 * np.pi * rng.random(k)
            r = np.interp(rng.random(k) * m_tab[-1], m_tab, r_tab)
            sth = np.sqrt(np.maximum(0.0, 1 - cth * cth))
            p = np.stack([r * sth * np.cos(phi), r * sth * np.sin(phi), r * cth], axis=1)
            ok = np.all(np.abs(p) <= half, axis=1)
            pg = p + np.asarray(h.d_com)
            # positions.c:366-385: the particle belongs to the halo of maximum model density
            best = np.full(k, -1)
            rho_best = np.zeros(k)
            for j, hj in enumerate(model.halos):
                rj = np.sqrt(((pg - np.asarray(hj.d_com)) ** 2).sum(axis=1)).astype(np.float32)
                rho_j = gas_density_profile(rj, hj)
                upd = (rho_j > rho_best) & (rj < hj.r_sample)
                best[upd] = j
                rho_best[upd] = rho_j[upd]
            ok &= best == i
            p = p[ok][:need]
            got.append(p.astype(np.float32))
            need -= len(p)
        pos = np.concatenate(got)[:cnt]
        d = np.asarray(h.d_com, dtype=np.float32)
        pos = pos + d                                   # setup.c:452-454 (f32 add)
        chunks.append(pos)
    pos = np.concatenate(chunks).astype(np.float32)
    bh = np.float32(half)
    bs = np.float32(box)
    pos = pos + bh                                       # setup.c:474-476
    for _ in range(2):
        pos = np.where(pos > bs, pos - bs, pos)
        pos = np.where(pos < 0, pos + bs, pos)
    return np.ascontiguousarray(pos, dtype=np.float32), make_ids(npart)
