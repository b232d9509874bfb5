"""world_size-2 gloo test (CPU) of the N>1 host logic: rendezvous + unique-id bootstrap, shard
bounds, padded equal-block all-gather, and the all-reduced error sums -- the same steps the library
performs with RCCL (csrc/api.hip), here with the oracle doing each rank's arithmetic."""
import os
import socket

import numpy as np
import pytest

from toycluster_amd import shard


def test_shard_bounds_cover_and_balance():
    for n in (1, 2, 7, 1000, 1001, 2_000_000):
        for r in (1, 2, 3, 8):
            s = shard.shard_len(n, r)
            b = [shard.shard_bounds(n, r, k) for k in range(r)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[k][1] == b[k + 1][0] for k in range(r - 1))
            assert all(hi - lo <= s for lo, hi in b)
            assert shard.padded_len(n, r) >= n and shard.padded_len(n, r) % r == 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_q):
    import torch.distributed as dist
    from oracle import oracle as O
    from toycluster_amd import model as M
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        uid = shard.bootstrap_unique_id(dist, rank, lambda: np.arange(128, dtype=np.uint8)[::-1].copy())
        assert uid.tolist() == list(range(127, -1, -1))

        m = M.preset("single", n)
        pos, ids = M.sample_gas(m, n, seed=5)          # same seed on every rank, as in bench.py
        o = O.Oracle(m, pos, ids, nthreads=2)
        o.find_sph_quantities()                        # replicated sort + index, as in the library
        p = o.particles()
        rm = o.global_density_model()
        lo, hi = shard.shard_bounds(n, world, rank)
        s = shard.shard_len(n, world)

        # each rank "owns" its range of the density result; others are wiped, then all-gathered back
        def block(a):
            b = np.zeros((s,) + a.shape[1:], a.dtype)
            b[:hi - lo] = a[lo:hi]
            return b
        hs = shard.allgather_shards(dist, block(p["hsml"]), n, world)
        ps = shard.allgather_shards(dist, block(p["pos"]), n, world)
        assert np.array_equal(hs, p["hsml"]) and np.array_equal(ps, p["pos"])

        # error sums: local partial sums + all-reduce == global sums (wvt_relax.c:73-87)
        import torch
        err = (np.abs(p["rho"] - rm) / rm).astype(np.float32)
        part = torch.tensor([float(err[lo:hi].astype(np.float64).sum()), float(hi - lo)], dtype=torch.float64)
        mx = torch.tensor([float(err[lo:hi].max())], dtype=torch.float64)
        dist.all_reduce(part, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        assert part[1].item() == n
        assert abs(part[0].item() / n - err.astype(np.float64).mean()) < 1e-12
        assert mx.item() == float(err.max())
        out_q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        out_q.put((rank, "FAIL %r" % (e,)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharded_iteration():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 3001, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
