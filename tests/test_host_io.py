"""CPU tests of the C host routines: the reference's parameter-file semantics (io.c:298-507) and
its Gadget-2 format-2 layout (io.c:13-287, io.h:1-41)."""
import os
import struct

import numpy as np

from toycluster_amd import hostio

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_stock_cluster_par():
    """tests/golden/cluster.par is the reference's sample parameter file (data, copied verbatim)."""
    rc, p, msg = hostio.read_param_file(os.path.join(GOLDEN, "cluster.par"))
    assert rc == 0, msg
    assert p["output_file"] == "./IC_single_0"
    assert p["ntotal"] == 1000000 and p["mtot200"] == 1e5 and p["mass_ratio"] == 0
    assert p["impact_param"] == 50 and p["zero_e_orbit_frac"] == 0.8 and p["cuspy"] == 0
    assert p["redshift"] == 0.87 and p["bfld_norm"] == 20e-6 and p["bfld_eta"] == 0.5
    assert p["baryon_fraction"] == 0.17
    assert p["unit_length"] == 3.085678e21 and p["unit_mass"] == 1.989e43 and p["unit_vel"] == 1e5


def test_param_file_rules(tmp_path):
    base = open(os.path.join(GOLDEN, "cluster.par")).read()
    # comments, unknown tags and repeated tags: first occurrence wins, unknown ignored (io.c:459-470)
    f = tmp_path / "a.par"
    f.write_text("% comment line\nNtotal 200000 % trailing\nSomeUnknownTag 7\n" + base)
    rc, p, _ = hostio.read_param_file(str(f))
    assert rc == 0 and p["ntotal"] == 200000
    # a missing tag is fatal with the reference's message (io.c:489-496)
    g = tmp_path / "b.par"
    g.write_text("\n".join(l for l in base.splitlines() if not l.startswith("Redshift")))
    rc, p, msg = hostio.read_param_file(str(g))
    assert rc == 2 and msg == "Value for tag 'Redshift' missing in parameter file '%s'." % g
    rc, p, msg = hostio.read_param_file(str(tmp_path / "nope.par"))
    assert rc == 1 and "not found" in msg


def test_snapshot_layout(tmp_path):
    ngas, ndm = 5, 3
    n = ngas + ndm
    rng = np.random.default_rng(0)
    pos, vel = rng.random((n, 3), dtype=np.float32), rng.random((n, 3), dtype=np.float32)
    ids = np.arange(1, n + 1, dtype=np.int32)
    u, rho, hsml, rhom = (rng.random(ngas, dtype=np.float32) for _ in range(4))
    bfld = rng.random((ngas, 3), dtype=np.float32)
    path = str(tmp_path / "snap")
    rc = hostio.write_snapshot(path, [ngas, ndm, 0, 0, 0, 0], [0.25, 1.5, 0, 0, 0, 0], 1234.0, pos, vel, ids, u, rho,
                               hsml, bfld, rhom)
    assert rc == 0
    header, blocks, order = hostio.read_snapshot(path)
    assert order == ["HEAD", "POS ", "VEL ", "ID  ", "U   ", "RHO ", "HSML", "BFLD", "RHOM"]     # io.h:31-41
    assert header["npart"] == (ngas, ndm, 0, 0, 0, 0) and header["npartTotal"] == (ngas, ndm, 0, 0, 0, 0)
    assert header["mass"][:2] == (0.25, 1.5) and header["BoxSize"] == 1234.0
    assert header["Omega0"] == 1 and header["OmegaLambda"] == 0.7 and header["HubbleParam"] == 0.7
    assert header["num_files"] == 1 and header["time"] == 0 and header["redshift"] == 0
    assert blocks["POS "] == pos.tobytes() and blocks["VEL "] == vel.tobytes() and blocks["ID  "] == ids.tobytes()
    assert blocks["U   "] == u.tobytes() and blocks["RHO "] == rho.tobytes() and blocks["HSML"] == hsml.tobytes()
    assert blocks["BFLD"] == bfld.tobytes() and blocks["RHOM"] == rhom.tobytes()
    size = os.path.getsize(path)
    assert size == sum(16 + 8 + len(b) for b in blocks.values())
    # first bytes: [8]["HEAD"][256+8][8][256]
    raw = open(path, "rb").read(24)
    assert struct.unpack("<i4sii i", raw[:20]) == (8, b"HEAD", 264, 8, 256)
