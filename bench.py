#!/usr/bin/env python3
"""bench.py -- WVT relaxation throughput of libtcgpu on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one WVT iteration of the reference loop (src/wvt_relax.c:61-218) over all particles:
Peano sort, neighbour index, hsml/density solve, error sums, model hsml, WVT sweep, move.
Workload at N=1: BASELINE.json configs[1] -- 2-cluster merger, 2e6 SPH particles, synthetic
positions drawn from the beta-model (seeded), already resident in HBM when timing starts.
Weak scaling: 2e6 particles per GPU; every rank solves its contiguous Peano range and the ranks
exchange positions / smoothing lengths with RCCL all-gathers each iteration.

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     -- dominant kernel (k_iter = fused density solve + WVT sweep): algorithmic bytes /
                  measured kernel time vs HBM peak (the kernel is VALU-bound)
  cpu_baseline -- the CPU oracle ("port" of the reference algorithm, OpenMP) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
BYTES_DENSITY_PER_PARTICLE = 56  # K5 (12+4 R, 12 W) + K9 (12+4 R, 12 W), fused in k_iter
BYTES_ITER_PER_PARTICLE = 868    # whole iteration incl. 128-bit radix sort
PER_GPU_PARTICLES = 2_000_000


def cpu_baseline(nsample, iters):
    """Time the oracle (CPU restatement, OpenMP) on a bounded sample of the same workload."""
    from toycluster_amd import model as M
    from oracle import oracle as O
    cores = len(os.sched_getaffinity(0))
    try:                                          # container CPU share (cgroup v2), e.g. "1600000 100000"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception:
        pass
    m = M.preset("merger", nsample)
    pos, ids = M.sample_gas(m, nsample, seed=14041981)
    o = O.Oracle(m, pos, ids, nthreads=cores)
    o.find_sph_quantities()                       # warm-up pass: the timed iterations start warm
    t0 = time.time()
    o.regularise(max_iter=iters - 1)
    dt = time.time() - t0
    return {"value": nsample * iters / dt, "unit": "particle-iterations/s", "cores": cores, "kind": "port",
            "sample": "oracle/tc_oracle.c (OpenMP restatement of the reference path), 2-cluster merger, "
                      "%d particles, %d warm WVT iterations, %.1f s" % (nsample, iters, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--particles-per-gpu", type=int, default=PER_GPU_PARTICLES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=200_000)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--force-comm", action="store_true",
                    help="testing: run the RCCL collectives through a 1-rank communicator")
    ap.add_argument("--force-dist", action="store_true",
                    help="testing: initialise torch.distributed (nccl) and bootstrap the id even with one rank")
    args = ap.parse_args()

    # RCCL prints a version banner on stdout when a communicator is created; keep stdout clean for
    # the one JSON line by pointing fd 1 at stderr until the result is ready.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from toycluster_amd import binding, model as M, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    torch.cuda.set_device(local_rank)
    uid = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        uid = shard.bootstrap_unique_id(dist, rank, binding.comm_unique_id)

    n_total = args.particles_per_gpu * world
    m = M.preset("merger", n_total)
    pos, ids = M.sample_gas(m, n_total, seed=14041981)       # same seed => same particles on every rank

    g = binding.TcGpu(local_rank, rank=rank, nranks=world, unique_id=uid,
                      options={"force_comm": 1} if args.force_comm else None)
    g.set_model(m)
    g.upload(pos, ids)
    del pos, ids

    step_size = 0.0085                                       # wvt_relax.c:51

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        err_mean, err_max = g.density_error()                # density pass + error sums (syncs)
        g.wvt_step(step_size, move=True, fetch=False)        # model hsml + sweep + move (+ all-gather)
        return err_mean, err_max

    for _ in range(args.warmup):
        one_step()
    g.phase_times(reset=True)

    barrier()
    t0 = time.perf_counter()
    errs = [one_step() for _ in range(args.steps)]
    barrier()
    dt = time.perf_counter() - t0

    if use_dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    phases = g.phase_times()
    dens_s, dens_launch = phases["density"]
    dens_avg = dens_s / max(1, dens_launch)
    n_local = (n_total + world - 1) // world
    achieved = BYTES_DENSITY_PER_PARTICLE * n_local / dens_avg / 1e9 if dens_avg > 0 else 0.0

    # HBM/fabric traffic of the dominant kernel: PMC counters cannot be collected from inside the
    # process, so the figure measured by `tools/profile_round.sh` (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE passes of this same command) is read from profiles/ when present.
    traffic = None
    valu_insts = None
    tpath = os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")
    if os.path.exists(tpath) and world == 1 and args.particles_per_gpu == PER_GPU_PARTICLES:
        try:
            tj = json.load(open(tpath))
            traffic = tj["bytes_per_launch"]
            valu_insts = tj.get("valu_insts_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(args.cpu_sample, args.cpu_iters)
        value = n_total * args.steps / dt
        out = {
            "metric": "WVT-relaxed particles/sec (whole node)",
            "value": value,
            "unit": "particle-iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "2-cluster merger (Mass_Ratio 0.3125), %d SPH particles per GPU, "
                                   "WVT iterations (sort + density solve + sweep + move)" % args.particles_per_gpu,
                       "particles_total": n_total, "parallelism": "peano-range shards x%d, RCCL all-gather" % world,
                       "err_mean_last": errs[-1][0], "err_max_last": errs[-1][1]},
            "roofline": {"bound": "hbm", "kernel": "k_iter (fused density solve K5 + WVT sweep K9)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": 1e3 * dens_avg, "launches": dens_launch,
                         "algorithmic_bytes_per_particle": BYTES_DENSITY_PER_PARTICLE,
                         "whole_iteration_GBs": BYTES_ITER_PER_PARTICLE * n_total * args.steps / dt / 1e9 / world,
                         # the kernel is VALU-bound, not HBM-bound: wave instructions of one
                         # launch (rocprofv3 SQ_INSTS_VALU, profiles/) x 4 cycles / (1024 SIMDs x 2.4 GHz) / launch time
                         "valu_issue_frac": (valu_insts * 4 / (1024 * 2.4e9) / dens_avg) if (valu_insts and dens_avg > 0) else None},
            "cpu_baseline": cpu,
            "phase_ms_per_step": {k: 1e3 * v[0] / args.steps for k, v in phases.items() if v[1]},
        }
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    g.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
