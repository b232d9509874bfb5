#!/usr/bin/env python3
"""Per-rank cost of the sharded path on ONE GPU (profiling helper, not part of the product).

    TCGPU_LOOPBACK_EXCLUSIVE=1 python3 tools/shard_probe.py <nranks> <particles_per_rank> [iterations]

Runs `nranks` loopback ranks (host threads, tcgpu_comm_init_loopback) over nranks x particles_per_rank particles
of the 2-cluster merger.  With TCGPU_LOOPBACK_EXCLUSIVE set only one rank computes at a time, so the HIP-event
phase times of a rank are those it would see alone on its own GPU; the loopback "comm" phase (device-to-device
copies) is NOT an xGMI time -- the bytes each rank receives are reported instead.  Prints one JSON object:
per-rank local-set size, phase milliseconds per iteration, received bytes per iteration, and the projected
weak-scaling efficiency  t(1 rank) / max_r (compute_r + bytes_r / BW)  for the stated all-gather bandwidth.
"""
import json
import sys
import threading
import time

sys.path.insert(0, ".")
import numpy as np
from toycluster_amd import binding, hostio

R = int(sys.argv[1])
per = int(float(sys.argv[2]))
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
BW = 300e9                                   # B/s a rank receives in an 8-GPU xGMI all-gather (assumed, not measured here)
n = R * per
s = hostio.setup_system("tests/golden/cluster.par", {"ntotal": 2 * n, "mass_ratio": 0.3125})
pos, ids = hostio.sample_gas(s, nthreads=8)
m = hostio.setup_to_model(s)


def run_group(nranks):
    ctxs = [binding.TcGpu(0) for _ in range(nranks)]
    if nranks > 1:
        binding.loopback_group(ctxs)
    out = [None] * nranks

    def work(r):
        try:
            g = ctxs[r]
            g.set_model(m)
            g.upload(pos, ids)
            for _ in range(3):
                g.density_error(); g.wvt_step(0.0085, fetch=False)
            g.phase_times(reset=True); g.comm_bytes(reset=True)
            t0 = time.perf_counter()
            for _ in range(iters):
                e = g.density_error(); g.wvt_step(0.0085, fetch=False)
            wall = time.perf_counter() - t0
            ph = g.phase_times()
            out[r] = dict(rank=r, info=g.local_set_info(), err_mean=e[0], wall_ms_per_iter=1e3 * wall / iters,
                          phase_ms={k: 1e3 * v[0] / iters for k, v in ph.items() if v[1]},
                          recv_bytes_per_iter=g.comm_bytes() / iters)
        except Exception as ex:              # pragma: no cover
            out[r] = repr(ex)
    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join() for t in th]
    [c.close() for c in ctxs]
    return out


res = run_group(R)
for o in res:
    if isinstance(o, str):
        print(o)
        sys.exit(1)
    o["compute_ms"] = sum(v for k, v in o["phase_ms"].items() if k != "comm")
    o["comm_ms_at_bw"] = 1e3 * o["recv_bytes_per_iter"] / BW
summary = dict(nranks=R, particles_total=n, particles_per_rank=per, iterations=iters, assumed_allgather_bw_GBs=BW / 1e9,
               ranks=res)
if R > 1:
    # single-rank reference at the per-rank size
    pos, ids = pos[:0], ids[:0]
    s1 = hostio.setup_system("tests/golden/cluster.par", {"ntotal": 2 * per, "mass_ratio": 0.3125})
    pos, ids = hostio.sample_gas(s1, nthreads=8)
    m = hostio.setup_to_model(s1)
    one = run_group(1)[0]
    one["compute_ms"] = sum(v for k, v in one["phase_ms"].items() if k != "comm")
    summary["single_rank_at_per_rank_size"] = one
    slow = max(o["compute_ms"] + o["comm_ms_at_bw"] for o in res)
    summary["projected_weak_scaling_efficiency"] = one["compute_ms"] / slow
    summary["max_nloc_over_nown"] = max(o["info"]["nloc"] / max(1, o["info"]["nown"]) for o in res)
print(json.dumps(summary, indent=1))
