#!/usr/bin/env python3
"""Per-rank cost of the sharded path on ONE GPU (profiling helper, not part of the product).

    TCGPU_LOOPBACK_EXCLUSIVE=1 python3 tools/shard_probe.py <nranks> <particles_per_rank> [iterations]

Runs `nranks` loopback ranks (host threads, tcgpu_comm_init_loopback) over nranks x particles_per_rank particles
of the 2-cluster merger.  With TCGPU_LOOPBACK_EXCLUSIVE set only one rank computes at a time, so the HIP-event
phase times of a rank are those it would see alone on its own GPU; the loopback "comm" phase (device-to-device
copies) is NOT an xGMI time -- the bytes each rank receives are reported instead.  Prints one JSON object:
per-rank local-set size, phase milliseconds per iteration, received bytes per iteration, and the projected
weak-scaling efficiency  t(1 rank) / max_r (compute_r + bytes_r / BW + latency)  as a RANGE: nothing about the fabric is
measured here, so the receive rate is taken between BW_LO (about one xGMI link's worth, 153 GB/s) and BW_HI (several links in
parallel), and every pass is charged LAT_MS for its small collectives -- the pyramid all-gather, the count-matrix
all-gather with its host read-back, the grouped ghost send / receive, the status agreement and the two scalar all-reduces
with their synchronisation: six collectives at ~30 us plus two stream synchronisations at ~20 us.  The first hardware run
replaces all of this.
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
BW_LO, BW_HI = 150e9, 400e9                  # B/s a rank receives over xGMI: ASSUMED range, not measured here
BW = BW_LO
LAT_MS = 6 * 0.030 + 2 * 0.020               # small collectives and synchronisations of one pass: ASSUMED
n = R * per
s = hostio.setup_system("tests/golden/cluster.par", {"ntotal": 2 * n, "mass_ratio": 0.3125})
pos, ids = hostio.sample_gas(s, nthreads=8)
m = hostio.setup_to_model(s)


def device_bytes_used():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    fr, tot = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(fr), C.byref(tot))
    return tot.value - fr.value


BASE_USED = None


def run_group(nranks):
    global BASE_USED
    if BASE_USED is None:
        binding.TcGpu(0).close()             # runtime initialised: what is in use before any particle array exists
        BASE_USED = device_bytes_used()
    ctxs = [binding.TcGpu(0) for _ in range(nranks)]
    if nranks > 1:
        binding.loopback_group(ctxs)
    out = [None] * nranks

    def work(r):
        try:
            g = ctxs[r]
            g.set_model(m)
            g.phase_times(reset=True)                # timing on from the start: what the cold start costs a rank
            g.upload(pos, ids)
            for _ in range(3):
                g.density_error(); g.wvt_step(0.0085, fetch=False)
            ph0 = g.phase_times()
            g.phase_times(reset=True); g.comm_bytes(reset=True)
            t0 = time.perf_counter()
            for _ in range(iters):
                e = g.density_error(); g.wvt_step(0.0085, fetch=False)
            wall = time.perf_counter() - t0
            ph = g.phase_times()
            out[r] = dict(rank=r, info=g.local_set_info(), err_mean=e[0], wall_ms_per_iter=1e3 * wall / iters,
                          phase_ms={k: 1e3 * v[0] / iters for k, v in ph.items() if v[1]},
                          startup_phase_ms_total={k: 1e3 * v[0] for k, v in ph0.items() if v[1]},
                          recv_bytes_per_iter=g.comm_bytes() / iters)
        except Exception as ex:              # pragma: no cover
            out[r] = repr(ex)
    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join() for t in th]
    used = device_bytes_used()               # every context of the group still holds its allocations
    [c.close() for c in ctxs]
    for o in out:
        if isinstance(o, dict):
            o["device_bytes_per_rank_mean"] = (used - BASE_USED) / nranks
    return out


res = run_group(R)
for o in res:
    if isinstance(o, str):
        print(o)
        sys.exit(1)
    o["compute_ms"] = sum(v for k, v in o["phase_ms"].items() if k != "comm")
    o["startup_compute_ms"] = sum(v for k, v in o["startup_phase_ms_total"].items() if k != "comm")   # cold pass + 3 iterations
    o["comm_ms_at_bw"] = 1e3 * o["recv_bytes_per_iter"] / BW
    o["comm_ms_range"] = [1e3 * o["recv_bytes_per_iter"] / BW_HI + LAT_MS, 1e3 * o["recv_bytes_per_iter"] / BW_LO + LAT_MS]
summary = dict(nranks=R, particles_total=n, particles_per_rank=per, iterations=iters,
               assumed_receive_bw_GBs=[BW_LO / 1e9, BW_HI / 1e9], assumed_collective_latency_ms_per_pass=LAT_MS, ranks=res)
if R > 1:
    # single-rank reference at the per-rank size
    pos, ids = pos[:0], ids[:0]
    s1 = hostio.setup_system("tests/golden/cluster.par", {"ntotal": 2 * per, "mass_ratio": 0.3125})
    pos, ids = hostio.sample_gas(s1, nthreads=8)
    m = hostio.setup_to_model(s1)
    one = run_group(1)[0]
    one["compute_ms"] = sum(v for k, v in one["phase_ms"].items() if k != "comm")
    one["startup_compute_ms"] = sum(v for k, v in one["startup_phase_ms_total"].items() if k != "comm")
    summary["single_rank_at_per_rank_size"] = one
    slow_lo = max(o["compute_ms"] + o["comm_ms_range"][1] for o in res)
    slow_hi = max(o["compute_ms"] + o["comm_ms_range"][0] for o in res)
    summary["projected_weak_scaling_efficiency_range"] = [one["compute_ms"] / slow_lo, one["compute_ms"] / slow_hi]
    summary["projected_weak_scaling_efficiency"] = one["compute_ms"] / slow_lo
    summary["max_nloc_over_nown"] = max(o["info"]["nloc"] / max(1, o["info"]["nown"]) for o in res)
print(json.dumps(summary, indent=1))
