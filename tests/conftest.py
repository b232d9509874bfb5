import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Built artefacts are not part of the source tree: build whatever is missing (hipcc cross-compiles
    without a GPU; everything else is gcc).  A no-op when __graft_entry__.build() has run already."""
    need = [os.path.join(ROOT, "toycluster_amd", "lib", n) for n in ("libtcgpu.so", "libtchost.so", "libtcshim.so")]
    need += [os.path.join(ROOT, "toycluster_amd", "host", "toycluster_hip"), os.path.join(ROOT, "oracle", "libtcoracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


class FixtureModel:
    """Model scalars as stored in a golden .npz (same attributes as toycluster_amd.model.ClusterModel)."""

    def __init__(self, d):
        from toycluster_amd.model import Halo
        self.boxsize, self.mpart_gas, self.mtotal, self.name = d["boxsize"], d["mpart_gas"], d["mtotal"], d["name"]
        self.bfld_eta = 0.5
        self.halos = [Halo(rho0=h["rho0"], beta=h["beta"], rcore=h["rcore"], rcut=h["rcut"], d_com=tuple(h["d_com"]),
                           r_sample=h["r_sample"], mass_gas=h["mass_gas"], have_cuspy=h["have_cuspy"])
                      for h in d["halos"]]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    case = {k: z[k] for k in z.files}
    case["model"] = FixtureModel(json.loads(bytes(case["model"]).decode()))
    return case


@pytest.fixture(scope="session", params=["case_single_3000", "case_merger_5000"])
def golden_case(request):
    return load_case(request.param)


@pytest.fixture(scope="session")
def gpu():
    """One libtcgpu context on device 0 for the whole session (the HIP path; no fallback)."""
    from toycluster_amd import binding
    g = binding.TcGpu(0)
    yield g
    g.close()
