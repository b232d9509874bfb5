"""Two PROCESSES, two GPUs, real RCCL: the sharded path end to end against the single-GPU result.

Needs two visible devices, so it is skipped on the one-GPU boxes of the pool (the in-process loopback tests of
test_gpu_parity.py cover the sharded control flow there, and test_rccl_code_path_single_rank the RCCL calls with one
rank).  Each rank is a FRESH process that touches no GPU before tcgpu_comm_init (RCCL wants to initialise the
device itself); rank 0 creates the 128-byte unique id and hands it over through a file."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, time
    sys.path.insert(0, %r)
    import numpy as np
    from toycluster_amd import binding, model as M
    rank, world, tmp, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    idfile = os.path.join(tmp, "uid.npy")
    if world > 1:
        if rank == 0:
            uid = binding.comm_unique_id()
            np.save(idfile + ".tmp.npy", uid); os.replace(idfile + ".tmp.npy", idfile)
        else:
            t0 = time.time()
            while not os.path.exists(idfile):
                if time.time() - t0 > 120: raise SystemExit("no unique id from rank 0")
                time.sleep(0.05)
            uid = np.load(idfile)
    else:
        uid = None
    m = M.preset("merger", n)
    pos, ids = M.sample_gas(m, n, seed=23)
    # ghost_exchange = 2: always the pyramids + grouped ncclSend / ncclRecv (at this size the automatic choice would
    # be the position all-gather)
    g = binding.TcGpu(rank, rank=rank, nranks=world, unique_id=uid, options={"ghost_exchange": 2})
    g.set_model(m)
    g.upload(pos, ids)
    log = g.Regularise_sph_particles(max_iter=4)
    info = g.local_set_info()
    g.Find_sph_quantities()
    p = g.particles()
    np.savez(os.path.join(tmp, "out_%%d_of_%%d.npz" %% (rank, world)), log=np.array([[l["err_mean"], l["err_max"], l["step"]] for l in log]),
             nloc=info["nloc"], nown=info["nown"], recv=g.comm_bytes(), **p)
    g.close()
''') % ROOT


def _ndev():
    import torch
    return torch.cuda.device_count()            # does not initialise the GPU


@pytest.mark.gpu
@pytest.mark.skipif(_ndev() < 2, reason="needs two GPUs (one process per GPU over RCCL)")
def test_two_processes_two_gpus_match_one_gpu(tmp_path):
    n = 40009
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, str(script), "0", "1", str(tmp_path), str(n)], env=env, capture_output=True,
                         text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(tmp_path), str(n)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    ref = np.load(tmp_path / "out_0_of_1.npz")
    for r in range(2):
        z = np.load(tmp_path / ("out_%d_of_2.npz" % r))
        assert np.array_equal(z["log"], ref["log"])                      # exact sums: the same decisions, bit for bit
        for k in ("id", "pos", "hsml", "rho", "varhsmlfac", "rho_model"):
            assert np.array_equal(z[k], ref[k]), (r, k)
        assert z["nown"] < z["nloc"] <= n and z["recv"] > 0
