"""world_size-2 gloo test (CPU) of the N > 1 design of csrc/api.hip: Peano-range shards, the
interest pyramid a rank marks from its own particles' permitted query radii, the SENDER-side ghost selection against the
all-gathered pyramids, and the coverage that makes a sharded pass exact -- every neighbour a permitted query of an own
particle can return lies in own range + ghosts.  The geometry below is a host restatement of tc_margin_radius (tc_ctx.h),
k_mark_interest's per-ball marking and tc_in_mask (kernels_stream.hip); the oracle supplies hsml and the model hsml."""
import os
import socket

import numpy as np
import pytest

from toycluster_amd import shard

LP_MAX = 8            # TC_LP_MAX: deepest pyramid level


def test_shard_bounds_cover_and_balance():
    for n in (1, 2, 7, 1000, 1001, 2_000_000):
        for r in (1, 2, 3, 8):
            s = shard.shard_len(n, r)
            b = [shard.shard_bounds(n, r, k) for k in range(r)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[k][1] == b[k + 1][0] for k in range(r - 1))
            assert all(hi - lo <= s for lo, hi in b)


def margin_radius(h0, w, box, widen=0):
    """tc_margin_radius: first query, the reference's two x1.23 retries (sph.c:49-54), the sweep's ball w*box (f32 steps)."""
    hb = np.float32(np.float64(h0) * 1.23)
    h3 = np.float32(np.float64(hb) * 1.23)
    for _ in range(widen):
        h3 = np.float32(np.float64(h3) * 1.23)
    hw = np.float32(np.float64(w) * box)
    return np.float32(max(h3, hw)) * np.float32(1.000001)


def mark_pyramid(pos, h0, w, box):
    """Bits of the cells (levels 1..LP_MAX, one bool array per level) that the permitted queries of these particles can
    touch: ball of the margin radius, padded like the cell-table query, at the level whose cell edge s has r/2 <= s < r."""
    pyr = [None] + [np.zeros((1 << L,) * 3, bool) for L in range(1, LP_MAX + 1)]
    for k in range(len(pos)):
        rg = float(margin_radius(h0[k], w[k], box))
        rp = rg * (1.0 + 1e-5) + box * 2e-6
        L = 1 if rp >= box else int(np.floor(np.log2(box / rp))) + 1
        L = min(max(L, 1), LP_MAX)
        nL = 1 << L
        idx = []
        for d in range(3):
            a, b = int(np.floor((float(pos[k, d]) - rp) * nL / box)), int(np.floor((float(pos[k, d]) + rp) * nL / box))
            idx.append(np.arange(nL) if b - a + 1 >= nL else np.arange(a, b + 1) % nL)
        pyr[L][np.ix_(*idx)] = True
    return pyr


def in_pyramid(pyr, pos, box):
    """tc_in_mask: is the particle's cell marked at any level?  (cell = top L bits of trunc(x / box * 2^63))"""
    X = np.floor(pos.astype(np.float64) / box * 9223372036854775808.0)
    hit = (X >= 9223372036854775808.0).any(axis=1)              # a coordinate == boxsize: always taken
    Xi = np.minimum(X, 9223372036854775807.0).astype(np.uint64)
    for L in range(1, LP_MAX + 1):
        c = (Xi >> np.uint64(63 - L)).astype(np.int64)
        hit |= pyr[L][c[:, 0], c[:, 1], c[:, 2]]
    return hit


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

        m = M.preset("merger", n)
        pos, ids = M.sample_gas(m, n, seed=5)          # same seed on every rank, as in bench.py
        box = m.boxsize
        o = O.Oracle(m, pos, ids, nthreads=2)
        o.find_sph_quantities()                        # the global Peano order, carried hsml
        p = o.particles()
        w, _ = o.wvt_step(0.0085, move=False)          # model hsml of these positions (box units)
        P, h0 = p["pos"], p["hsml"]
        lo, hi = shard.shard_bounds(n, world, rank)
        other = 1 - rank
        olo, ohi = shard.shard_bounds(n, world, other)

        # 1. every rank marks its pyramid from its OWN particles; the pyramids are all-gathered (2.4 MB each on the GPU)
        mine = mark_pyramid(P[lo:hi], h0[lo:hi], w[lo:hi], box)
        packed = [None if a is None else np.packbits(a) for a in mine]
        allp = [None] * world
        dist.all_gather_object(allp, packed)
        theirs = [None] + [np.unpackbits(allp[other][L]).astype(bool).reshape((1 << L,) * 3) for L in range(1, LP_MAX + 1)]

        # 2. the SENDER tests its own particles against the receiver's pyramid; the counts are all-gathered
        send = lo + np.where(in_pyramid(theirs, P[lo:hi], box))[0]
        counts = [None] * world
        dist.all_gather_object(counts, len(send))
        # 3. the ghosts travel (20 B each on the GPU: index + position + model hsml), grouped by sender, ascending index
        box_ = [None] * world
        dist.all_gather_object(box_, (send.astype(np.int64), P[send].copy()))
        gidx, gpos = box_[other]
        assert len(gidx) == counts[other] and np.all(np.diff(gidx) > 0)
        assert np.array_equal(gpos, P[gidx])
        # sender-side selection == what the receiver would have selected from every position (the path it replaced)
        assert np.array_equal(gidx, olo + np.where(in_pyramid(mine, P[olo:ohi], box))[0])

        # 4. coverage: every particle within the permitted radius of an own particle is own or ghost
        local = np.zeros(n, bool)
        local[lo:hi] = True
        local[gidx] = True
        Pd = P.astype(np.float64)
        rng = np.random.default_rng(17 + rank)
        sample = np.concatenate([rng.choice(np.arange(lo, hi), 250, replace=False),
                                 lo + np.argsort(-w[lo:hi])[:50]])          # and the widest balls (the outskirts)
        worst = 0
        for i in sample:
            rg = float(margin_radius(h0[i], w[i], box))
            d = Pd - Pd[i]
            d -= box * np.round(d / box)
            ngb = np.where((d * d).sum(axis=1) < (rg * (1 + 1e-6)) ** 2)[0]
            assert local[ngb].all(), (rank, int(i), int((~local[ngb]).sum()))
            worst = max(worst, len(ngb))
            # and the reference's own lists at the radii the density pass may query (sph.c:36-64) are inside
            for r in (h0[i], np.float32(np.float64(h0[i]) * 1.23), np.float32(np.float64(w[i]) * box)):
                assert local[o.find_ngb_simple(int(i), r)].all()
        # at 3e4 particles split two ways the shell is nearly everything (hsml up to a third of the box in the outskirts);
        # it thins out with N -- 0.83 n at 2e6 split four ways, 1.4-1.9 x the own range at 8 x 2e6
        assert 0 < len(gidx) <= ohi - olo
        out_q.put((rank, "ok", int(len(gidx)), int(worst)))
    except Exception as e:  # pragma: no cover
        import traceback
        out_q.put((rank, "FAIL %r %s" % (e, traceback.format_exc()), 0, 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_ghost_selection_covers_every_permitted_query():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 30011, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[:2] for r in res) == [(0, "ok"), (1, "ok")], res
