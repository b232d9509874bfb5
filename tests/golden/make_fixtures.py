"""Regenerates tests/golden/case_*.npz from the CPU oracle (oracle/tc_oracle.c).

Provenance: the reference cannot be built in this image (needs libgsl) and ships no golden data,
so these vectors are outputs of this repo's own restatement -- "parity unpinned" beyond the
Peano known answers in peano_kat.json.  They pin the oracle against regressions and give the GPU
tests committed expected values.

    python tests/golden/make_fixtures.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from toycluster_amd import model as M   # noqa: E402
from oracle import oracle as O          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def model_json(m):
    return json.dumps(dict(boxsize=m.boxsize, mpart_gas=m.mpart_gas, mtotal=m.mtotal, name=m.name,
                           halos=[dict(rho0=h.rho0, beta=h.beta, rcore=h.rcore, rcut=h.rcut, d_com=list(h.d_com),
                                       r_sample=h.r_sample, mass_gas=h.mass_gas, have_cuspy=h.have_cuspy)
                                  for h in m.halos]))


def make(name, preset, n, seed, relax_iters):
    m = M.preset(preset, n)
    pos, ids = M.sample_gas(m, n, seed=seed)
    out = dict(model=np.frombuffer(model_json(m).encode(), dtype=np.uint8), pos=pos, ids=ids)

    o = O.Oracle(m, pos, ids)
    hi, lo, perm = o.sort_by_peano_key()
    out.update(key_hi=hi, key_lo=lo, perm=perm.astype(np.int32))
    o.build_tree()
    out["guess"] = np.array([o.guess_hsml(i) for i in range(n)], np.float32)
    probe = np.linspace(0, n - 1, 12).astype(np.int32)
    out["ngb_probe"] = probe
    out["ngb_hsml"] = (out["guess"][probe] * np.float32(0.9)).astype(np.float32)
    lists = [o.find_ngb_tree(int(i), float(h)) for i, h in zip(probe, out["ngb_hsml"])]
    out["ngb_offsets"] = np.cumsum([0] + [len(l) for l in lists]).astype(np.int32)
    out["ngb_lists"] = np.concatenate(lists).astype(np.int32)

    o = O.Oracle(m, pos, ids)
    o.find_sph_quantities()
    p = o.particles()
    out.update(d_ids=p["id"], d_pos=p["pos"], d_hsml=p["hsml"], d_rho=p["rho"], d_vhf=p["varhsmlfac"])
    out["d_rho_model"] = o.global_density_model()
    hs, de = o.wvt_step(0.0085, move=False)
    out.update(w_hsml=hs, w_delta=de)
    eta = 0.5
    rm = out["d_rho_model"]
    a = (rm / np.float32(m.halos[0].rho0)) ** np.float32(eta)
    apot = np.stack([a, a, a], axis=1).astype(np.float32)
    o.set_apot(apot)
    out["c_apot"] = apot
    out["c_bfld"] = o.bfld_from_rotA()

    o = O.Oracle(m, pos, ids)
    log = o.regularise(max_iter=relax_iters)
    p = o.particles()
    out.update(r_ids=p["id"], r_pos=p["pos"], r_hsml=p["hsml"], r_rho=p["rho"],
               r_log=np.array([[l["it"], l["err_max"], l["err_mean"], l["err_diff"], l["step"]] for l in log]))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "written:", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    make("case_single_3000", "single", 3000, 11, 3)
    make("case_merger_5000", "merger", 5000, 12, 3)
