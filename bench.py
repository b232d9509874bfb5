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
  roofline     -- dominant kernel (k_iter = hsml / density solve; it also lists the sweep's neighbours in index order):
                  counted vector flops / measured kernel time vs the f64 vector peak (the kernel is VALU-bound; DESIGN.md
                  section 4), HBM figure beside it; the sweep's own kernels (k_wvt_chain4: the lists evaluated in the
                  reference's summation order) in `sweep_kernel`
  cpu_baseline -- the CPU oracle ("port" of the reference algorithm, OpenMP) on the SAME workload and state:
                  warm WVT iterations started from the GPU's positions and smoothing lengths
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
F64_VECTOR_PEAK_TFLOPS = 78.6    # MI355X vector f64 peak (half the 157.3 TF f32 vector peak of MI355X_MICROARCH.md)
# SURVEY.md 8(d) flop model of one particle-iteration: every list entry the hsml solver visits costs ~60 flop
# (f64 separation already paid, kernel + derivative polynomials, 1 sqrt, 1 f32 divide, ~2 f64 divides), the WVT
# sweep ~350 pairs x ~40 flop = 14 kflop.  The number of solver visits is COUNTED by a stats pass of this run.
FLOP_PER_SOLVER_PAIR = 60.0
FLOP_SWEEP_PER_PARTICLE = 14.0e3
BYTES_DENSITY_PER_PARTICLE = 28  # K5 (12+4 R, 12 W)
BYTES_SWEEP_PER_PARTICLE = 28    # K9 (12+4 R, 12 W)
BYTES_ITER_PER_PARTICLE = 868    # whole iteration incl. 128-bit radix sort
PER_GPU_PARTICLES = 2_000_000


def workload(n_gas):
    """BASELINE config 2 shape at n_gas SPH particles: the reference's sample parameter file with Mass_Ratio 0.3125
    and Ntotal = 2 n_gas, set up and sampled by the C host code exactly as the executable does it (host/tc_setup.c:
    the reference's positions.c:90-133 with its per-thread erand48 streams; 8 streams on every rank whatever the
    core count, so every rank draws the same particles).  1.6e7 particles take ~2 s (numpy: a minute)."""
    from toycluster_amd import hostio
    par = os.path.join(ROOT, "tests", "golden", "cluster.par")
    s = hostio.setup_system(par, {"ntotal": 2 * n_gas, "mass_ratio": 0.3125})
    pos, ids = hostio.sample_gas(s, nthreads=8)
    return hostio.setup_to_model(s), pos, ids


def cpu_baseline(m, state, iters):
    """Time the oracle (CPU restatement, OpenMP) on the benchmark's own workload: `iters` warm WVT iterations started
    from the state the GPU left behind (positions, ids, carried smoothing lengths), i.e. the same kind of iteration
    the GPU figure is about (no 2e5 sample any more)."""
    from oracle import oracle as O
    cores = len(os.sched_getaffinity(0))
    try:                                          # container CPU share (cgroup v2), e.g. "1600000 100000"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception:
        pass
    n = len(state["id"])
    o = O.Oracle(m, state["pos"], state["id"], hsml=state["hsml"], nthreads=cores)
    t0 = time.time()
    o.regularise(max_iter=iters - 1)              # `iters` loop bodies: sort, tree, density solve, error, sweep, move
    dt = time.time() - t0
    return {"value": n * iters / dt, "unit": "particle-iterations/s", "cores": cores, "kind": "port",
            "sample_particles": n, "sample_iterations": iters, "sample_seconds": dt,
            "sample": "oracle/tc_oracle.c (OpenMP restatement of the reference path) on the benchmark's own particles: "
                      "%d particles, %d warm WVT iterations from the GPU's state, %.1f s" % (n, iters, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--particles-per-gpu", type=int, default=PER_GPU_PARTICLES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-relax", action="store_true",
                    help="profiling runs: skip the stats pass and the whole-relaxation run behind the timed steps")
    ap.add_argument("--cpu-iters", type=int, default=2)
    ap.add_argument("--sweep", type=int, default=0,
                    help="0: the sweep in the reference's summation order, neighbour lists from k_iter (default, what the "
                         "library does); 2: the same sums by the stand-alone kernel; 1: round 2's fused f64 sweep (faster, "
                         "~1e-6 |delta| off per iteration) -- 1 and 2 for A/B runs only")
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
    m, pos, ids = workload(n_total)                          # deterministic => the same particles on every rank

    opts = {}
    if args.force_comm: opts["force_comm"] = 1
    if args.sweep: opts["sweep"] = args.sweep
    g = binding.TcGpu(local_rank, rank=rank, nranks=world, unique_id=uid, options=opts or None)
    g.set_model(m)
    g.upload(pos, ids)

    step_size = 0.0085                                       # wvt_relax.c:51

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        err_mean, err_max = g.density_error()                # density pass + error sums (syncs)
        g.wvt_step(step_size, move=True, fetch=False)        # model hsml + sweep + move (+ all-gather)
        return err_mean, err_max

    # Priming (part of set-up, like the upload): the first pass starts from hsml = 0 -- the reference's zero-initialised
    # SphP -- and costs several warm passes (cold tree guess, 4x the pair evaluations); on sharded runs the second pass
    # also makes the shards compact (one presentation).  The benchmark measures WARM iterations, the state the loop is
    # in for all but its first two passes, so two untimed iterations precede the W warm-up steps whatever W is.
    for _ in range(2):
        one_step()
    for _ in range(args.warmup):
        one_step()
    g.comm_bytes(reset=True)

    # The timed region runs the library as shipped: per-phase event records OFF (option "timing" = 0, the product default).
    barrier()
    t0 = time.perf_counter()
    errs = [one_step() for _ in range(args.steps)]
    barrier()
    dt = time.perf_counter() - t0

    if use_dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    recv_bytes = g.comm_bytes() / max(1, args.steps)
    lset = g.local_set_info()

    # Kernel times: the same K steps once more with HIP events around every phase on the library's stream (not timed
    # by the wall clock above).
    g.phase_times(reset=True)                                # switches "timing" on
    for _ in range(args.steps):
        one_step()
    phases = g.phase_times()
    g.set_option("timing", 0)
    dens_s, dens_launch = phases["density"]
    dens_avg = dens_s / max(1, dens_launch)
    sw_s, sw_launch = phases.get("wvt_sweep", (0.0, 0))
    sw_avg = sw_s / max(1, sw_launch)
    n_local = (n_total + world - 1) // world
    achieved = BYTES_DENSITY_PER_PARTICLE * n_local / dens_avg / 1e9 if dens_avg > 0 else 0.0
    state = g.particles() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    # Work counters of one density pass (stats instantiation of the same kernel, outside the timed region):
    # list entries visited by the hsml solver per particle -> counted flops of the launch.
    st = {"pair_evals": float("nan"), "queries": float("nan"), "candidates": float("nan")}
    if not args.no_relax:
        g.set_option("stats", 1)
        g.density_error()
        st = g.density_stats()
        g.set_option("stats", 0)
    flop_per_particle = st["pair_evals"] * FLOP_PER_SOLVER_PAIR + (FLOP_SWEEP_PER_PARTICLE if args.sweep == 1 else 0.0)
    tflops = flop_per_particle * n_local / dens_avg / 1e12 if dens_avg > 0 else 0.0
    sweep_tflops = FLOP_SWEEP_PER_PARTICLE * n_local / sw_avg / 1e12 if (sw_avg > 0 and args.sweep != 1) else 0.0

    # One whole relaxation under the reference's stop rule (wvt_relax.c:94-98) from the same initial positions:
    # the second half of BASELINE.json's metric ("iterations to <1% rho-error" = stop-rule count).
    relax_log, relax_s = None, float("nan")
    if not args.no_relax:
        g.upload(pos, ids)
        barrier()
        t0 = time.perf_counter()
        relax_log = g.Regularise_sph_particles()
        barrier()
        relax_s = time.perf_counter() - t0
    del pos, ids

    # Counter-derived figures of the dominant kernel cannot be collected from inside the process: they come from
    # the rocprofv3 --pmc passes of tools/profile_round2.sh over this same command, summarised by
    # tools/roofline_valu.py into profiles/dominant_kernel.json -- FILE values, labelled as such, and refused
    # when the kernel sources have changed since the profile was taken.
    prof = None
    ppath = os.path.join(ROOT, "profiles", "dominant_kernel.json")
    if os.path.exists(ppath) and world == 1 and args.particles_per_gpu == PER_GPU_PARTICLES:
        try:
            from tools import roofline_valu
            pj = json.load(open(ppath))
            if pj.get("kernel_sources_sha256") == roofline_valu.kernel_sources_sha256(ROOT):
                prof = pj
        except Exception:
            prof = None

    # every rank's view of the step, for the driver's 1 -> 8 curve (rank 0 prints them)
    mine = {"rank": rank, "phase_ms_per_step": {k: 1e3 * v[0] / args.steps for k, v in phases.items() if v[1]},
            "recv_bytes_per_step": recv_bytes, "local_set": lset["nloc"], "own": lset["nown"],
            "local_set_over_own": lset["nloc"] / max(1, lset["nown"]), "passes_repeated": lset["retries"],
            "device_memory_used_gb": (lambda fr_tot: (fr_tot[1] - fr_tot[0]) / 1e9)(torch.cuda.mem_get_info())}
    per_rank = [mine]
    if use_dist:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline:
            if state is None:                                # sharded run: the baseline is quoted on one GPU's share
                cpu = {"value": None, "unit": "particle-iterations/s", "cores": None, "kind": "port",
                       "sample": "not run at N > 1 (the N = 1 line carries it)"}
            else:
                cpu = cpu_baseline(m, state, args.cpu_iters)
        value = n_total * args.steps / dt
        out = {
            "metric": "WVT-relaxed particles/sec (whole node) + iterations to <1% rho-error",
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
            "data": "synthetic (beta-model positions sampled by the C host code from the reference's cluster.par, "
                    "Mass_Ratio 0.3125; random-seeded like the reference: erand48, 8 streams)",
            "config": {"workload": "2-cluster merger (Mass_Ratio 0.3125), %d SPH particles per GPU, "
                                   "WVT iterations (sort + density solve + sweep + move)" % args.particles_per_gpu,
                       "particles_total": n_total, "priming": "2 untimed iterations before the warm-up (cold start from hsml = 0)",
                       "parallelism": "peano-range shards x%d: per-rank local set (own range + ghost shell) with its own "
                                      "sort / cell table / mirror; per iteration the ghost exchange over RCCL (interest pyramids "
                                      "all-gathered, 20 B per ghost sent to the ranks that need it) and two exact scalar "
                                      "all-reduces" % world,
                       "sweep": {0: "reference order and roundings; neighbour lists in index order from k_iter, evaluated by k_wvt_chain4",
                                 2: "reference order and roundings, stand-alone kernel k_wvt_exact4 (option sweep = 2)",
                                 1: "round 2's fused f64 sweep (option sweep = 1)"}[args.sweep],
                       "ranks": per_rank,
                       "err_mean_last": errs[-1][0], "err_max_last": errs[-1][1]},
            # k_iter is a gather/stencil kernel bound by the vector ALU, not by HBM: the
            # headline fraction is counted vector flops against the f64 vector peak; the HBM fraction of its
            # compulsory 56 B/particle is reported beside it.
            "roofline": {"bound": "valu", "kernel": "k_iter (hsml / density solve K5%s)" % (
                             " + fused f64 sweep K9" if args.sweep == 1 else
                             "; candidates from per-particle ordered runs, sweep neighbours listed on the way" if args.sweep == 0 else ""),
                         "achieved": tflops, "peak": F64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tflops / F64_VECTOR_PEAK_TFLOPS,
                         "flop_model": "counted solver list visits/particle (stats pass of this run) x %g flop (SURVEY.md 8d), "
                                       "all counted as f64%s" % (FLOP_PER_SOLVER_PAIR, " + %g flop for the fused sweep" % FLOP_SWEEP_PER_PARTICLE if args.sweep == 1 else ""),
                         "solver_pair_evals_per_particle": st["pair_evals"], "queries_per_particle": st["queries"],
                         "candidates_per_particle": st["candidates"], "flop_per_particle": flop_per_particle,
                         "avg_launch_ms": 1e3 * dens_avg, "launches": dens_launch,
                         "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_particle": BYTES_DENSITY_PER_PARTICLE,
                                 "whole_iteration_GBs": BYTES_ITER_PER_PARTICLE * n_total * args.steps / dt / 1e9 / world},
                         "sweep_kernel": (None if args.sweep == 1 else
                                          {"kernel": ("k_wvt_chain4 + k_wvt_exact_w" if args.sweep == 0 else "k_wvt_exact4") +
                                                     " (WVT sweep K9 in the reference's summation order: f32 accumulator, "
                                                     "ascending index, one rounding per neighbour)",
                                           "run_list_prepass_ms": mine["phase_ms_per_step"].get("query_records"),
                                           "avg_launch_ms": 1e3 * sw_avg, "launches": sw_launch, "bound": "valu",
                                           "achieved": sweep_tflops, "peak": F64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": sweep_tflops / F64_VECTOR_PEAK_TFLOPS,
                                           "flop_model": "%g flop per particle (SURVEY.md 8d: ~350 pairs x ~40 flop)" % FLOP_SWEEP_PER_PARTICLE,
                                           "hbm_GBs": BYTES_SWEEP_PER_PARTICLE * n_local / sw_avg / 1e9 if sw_avg > 0 else 0.0}),
                         "traffic": prof["traffic_bytes_per_launch"] if prof else None,
                         "profile": ({"source": "profiles/dominant_kernel.json (rocprofv3 --pmc passes, tools/profile_round2.sh); "
                                                "file values from the profiling run, not measured by this run",
                                      "commit": prof.get("commit"), "valu_insts_per_launch": prof.get("valu_insts_per_launch"),
                                      "valu_insts_per_particle": prof.get("valu_insts_per_particle"),
                                      "valu_issue_frac": prof.get("valu_issue_frac"),
                                      "clock_ghz": prof.get("clock_ghz"), "l2_hit_rate": prof.get("l2_hit_rate")}
                                     if prof else None)},
            "relaxation": ({"iterations_to_stop": len(relax_log), "err_mean_at_stop": relax_log[-1]["err_mean"],
                            "err_max_at_stop": relax_log[-1]["err_max"], "relax_wall_s": relax_s,
                            "particles_per_s_to_convergence": n_total / relax_s,
                            "stop_rule": "reference's own (wvt_relax.c:94-98); an absolute mean error < 1 % is not reached "
                                         "by the reference either (SURVEY.md 6)"} if relax_log else None),
            "cpu_baseline": cpu,
            "phase_ms_per_step": mine["phase_ms_per_step"],
            "phase_ms_note": "HIP events per phase in a second run of the same K steps; the wall-clock figure above was "
                             "taken with the event records off",
        }
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    g.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
