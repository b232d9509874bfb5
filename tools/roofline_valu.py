#!/usr/bin/env python3
"""Turn one tools/profile_round2.sh run (gpurun_out/prof_<tag>) into the committed evidence under profiles/:

    python3 tools/roofline_valu.py gpurun_out/prof_r2a --round 2

  profiles/round<N>_kernel_stats.csv          rocprofv3 --kernel-trace --stats summary of the bench command
  profiles/round<N>_bench_under_trace.json    the bench line printed under that trace
  profiles/round<N>_k_iter_dispatches.txt     per-dispatch durations of the dominant kernel, vs bench.py's HIP events
  profiles/round<N>_pmc_summary.txt           every counter of every pass, per kernel (warm = last dispatch)
  profiles/round<N>_valu_cost.json            measured issue cost per VALU class in shader cycles (tools/ubench/valu_cost)
  profiles/dominant_kernel.json               what bench.py quotes (labelled as file values): traffic, instruction mix,
                                              clock, VALU issue fraction -- with the commit and a hash of the kernel
                                              sources, so that a stale profile is refused

VALU issue fraction of k_iter (all quantities in shader cycles, no clock assumed):
    sum over classes c of  N_c (rocprofv3 SQ_INSTS_VALU_<c>, wave-level instructions of the warm dispatch)
                         x cost_c (cycles one such instruction occupies a SIMD's issue port: valu_cost, 4 waves/SIMD)
    divided by  1024 SIMDs x kernel cycles,  kernel cycles = GRBM_GUI_ACTIVE / 8  (the counter sums the 8 XCDs).
Instructions the class counters do not cover (moves, compares, selects, DPP, readlane: total - sum of classes) are
priced at the mean measured cost of those instructions.
"""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NSIMD = 1024
DOMINANT = "k_iter"

# rocprofv3 class counter -> the micro-benchmarked instructions that price it
CLASS_COST = {
    "SQ_INSTS_VALU_FMA_F64": ["v_fma_f64"], "SQ_INSTS_VALU_ADD_F64": ["v_add_f64"], "SQ_INSTS_VALU_MUL_F64": ["v_mul_f64"],
    "SQ_INSTS_VALU_TRANS_F64": ["v_rcp_f64", "v_rsq_f64"],
    "SQ_INSTS_VALU_FMA_F32": ["v_fma_f32"], "SQ_INSTS_VALU_ADD_F32": ["v_add_f32"], "SQ_INSTS_VALU_MUL_F32": ["v_mul_f32"],
    "SQ_INSTS_VALU_TRANS_F32": ["v_sqrt_f32", "v_rcp_f32"],
    "SQ_INSTS_VALU_CVT": ["v_cvt_f32_f64", "v_cvt_f64_f32"],
    "SQ_INSTS_VALU_INT32": ["v_add_u32", "v_mul_lo_u32"], "SQ_INSTS_VALU_INT64": ["v_lshlrev_b64"],
}
OTHER_COST = ["v_mov_b32", "v_cndmask_b32", "v_cmp_lt_f32", "v_cmp_lt_f64", "v_add_u32_dpp", "v_mov_b32_dpp",
              "v_readlane_b32", "v_mbcnt_lo_u32_b32", "v_min_f64"]


def kernel_sources_sha256(root=ROOT):
    h = hashlib.sha256()
    d = os.path.join(root, "toycluster_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def short(name):
    return name.split("(")[0].replace("void ", "").split("<")[0].strip()


def load_pass(d):
    """{kernel: {counter: [value per dispatch]}}, {kernel: [duration ns per dispatch]} of one pass directory"""
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(dict))
    dur = collections.defaultdict(dict)
    if not files:
        return {}, {}
    for r in csv.DictReader(open(files[0])):
        k, disp = short(r["Kernel_Name"]), int(r["Dispatch_Id"])
        agg[k][r["Counter_Name"]][disp] = agg[k][r["Counter_Name"]].get(disp, 0.0) + float(r["Counter_Value"])
        dur[k][disp] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {k: {c: [v[i] for i in sorted(v)] for c, v in cs.items()} for k, cs in agg.items()}
    return out, {k: [v[i] for i in sorted(v)] for k, v in dur.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("prof")
    ap.add_argument("--round", type=int, default=2)
    ap.add_argument("--particles", type=int, default=2_000_000)
    a = ap.parse_args()
    P, R = a.prof, "round%d" % a.round
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    dst = lambda n: os.path.join(ROOT, "profiles", n)

    # ---- kernel trace ----
    shutil.copy(glob.glob(P + "/trace/*/*_kernel_stats.csv")[0], dst(R + "_kernel_stats.csv"))
    shutil.copy(P + "/bench_under_trace.json", dst(R + "_bench_under_trace.json"))
    rows = [r for r in csv.DictReader(open(glob.glob(P + "/trace/*/*_kernel_trace.csv")[0])) if short(r["Kernel_Name"]) == DOMINANT]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    b = json.loads([l for l in open(P + "/bench_under_trace.json") if l.startswith("{")][-1])
    steps = b["steps"]
    txt = ["%s dispatches of `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps %d --warmup %d --no-cpu-baseline --no-relax`"
           % (DOMINANT, steps, b["warmup"]),
           "(kernel_trace.csv, End-Start, ms). The dispatches before the last %d are untimed: priming (1 = cold start) and %d warm-up steps." % (steps, b["warmup"])]
    txt += ["  dispatch %d: %.3f ms" % (i + 1, x) for i, x in enumerate(d)]
    txt.append("mean of all %d (the --stats AverageNs): %.3f ms" % (len(d), sum(d) / len(d)))
    txt.append("mean of the %d timed dispatches:        %.3f ms" % (steps, sum(d[-steps:]) / steps))
    txt.append("bench.py roofline.avg_launch_ms (HIP events on the library stream, same run): %.3f ms" % b["roofline"]["avg_launch_ms"])
    open(dst(R + "_k_iter_dispatches.txt"), "w").write("\n".join(txt) + "\n")
    print("\n".join(txt[-3:]))

    # ---- PMC passes ----
    passes = {}
    summ = []
    for pd in sorted(glob.glob(P + "/pmc_*")):
        if not os.path.isdir(pd):
            continue
        name = os.path.basename(pd)[4:]
        passes[name] = load_pass(pd)
        cnt, dur = passes[name]
        for k in sorted(cnt):
            if not k.startswith("k_") and "radix" not in k.lower() and "scan" not in k.lower():
                continue
            summ.append("pmc_%s %s  (dispatches=%d, warm dispatch %.4f ms)" % (name, k, len(dur[k]), dur[k][-1] / 1e6))
            for c, v in sorted(cnt[k].items()):
                summ.append("   %-28s last=%.6g mean=%.6g" % (c, v[-1], sum(v) / len(v)))
    open(dst(R + "_pmc_summary.txt"), "w").write("\n".join(summ) + "\n")

    def warm(pass_name, counter, kernel=DOMINANT):
        return passes[pass_name][0][kernel][counter][-1]

    def warm_ns(pass_name, kernel=DOMINANT):
        return passes[pass_name][1][kernel][-1]

    cost = json.load(open(P + "/valu_cost_w4.json"))
    cost8 = json.load(open(P + "/valu_cost_w8.json")) if os.path.exists(P + "/valu_cost_w8.json") else None
    json.dump({"waves_per_simd_4": cost, "waves_per_simd_8": cost8,
               "method": "tools/ubench/valu_cost.hip: median over waves of d(s_memtime)/(instructions x waves per SIMD); "
                         "clock = d(s_memtime)/d(s_memrealtime) x 100 MHz"}, open(dst(R + "_valu_cost.json"), "w"), indent=1)
    cyc = {k: v["cycles_per_wave_instr_per_simd"] for k, v in cost["instr"].items()}
    mean = lambda names: sum(cyc[n] for n in names) / len(names)

    total = warm("f64", "SQ_INSTS_VALU")
    mix, issue_cycles, covered = {}, 0.0, 0.0
    for c, names in CLASS_COST.items():
        n = warm("f64" if c in passes["f64"][0][DOMINANT] else "f32", c)
        mix[c.replace("SQ_INSTS_VALU_", "")] = {"insts": n, "cycles_each": mean(names)}
        issue_cycles += n * mean(names)
        covered += n
    other = max(0.0, total - covered)
    mix["OTHER (moves, compares, selects, DPP, readlane, f64 min/max)"] = {"insts": other, "cycles_each": mean(OTHER_COST)}
    issue_cycles += other * mean(OTHER_COST)

    gui = warm("grbm", "GRBM_GUI_ACTIVE")
    kernel_cycles = gui / 8.0
    t_ns = warm_ns("grbm")
    clock_ghz = kernel_cycles / t_ns
    fetch, write = warm("fetch", "FETCH_SIZE"), warm("write", "WRITE_SIZE")
    hit, miss = warm("write", "TCC_HIT_sum"), warm("write", "TCC_MISS_sum")

    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "status", "--porcelain", "toycluster_amd/csrc"], cwd=ROOT, capture_output=True,
                                    text=True).stdout.strip())
    except Exception:
        commit, dirty = "unknown", True
    out = {
        "kernel": DOMINANT, "n_particles": a.particles,
        "commit": commit + ("+uncommitted csrc changes" if dirty else ""),
        "kernel_sources_sha256": kernel_sources_sha256(),
        "profile_dir": P,
        "warm_dispatch_ms_under_pmc": t_ns / 1e6,
        "traffic_bytes_per_launch": (2 * fetch + write) * 1024,
        "traffic_note": "FETCH_SIZE x 2 (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM) + WRITE_SIZE, KB x 1024; "
                        "separate --pmc passes; the gather is 16 B/lane but scattered, so the x2 is an upper bound",
        "fetch_size_kb": fetch, "write_size_kb": write, "l2_hit_rate": hit / (hit + miss),
        "valu_insts_per_launch": total, "valu_insts_per_particle": total / a.particles,
        "valu_mix": mix,
        "valu_issue_cycles_per_simd": issue_cycles / NSIMD,
        "kernel_cycles": kernel_cycles, "clock_ghz": clock_ghz,
        "clock_note": "GRBM_GUI_ACTIVE / 8 / kernel duration of the same dispatch (MI355X_MICROARCH.md, DVFS give-back)",
        "valu_issue_frac": issue_cycles / NSIMD / kernel_cycles,
        "salu_insts_per_launch": warm("f32", "SQ_INSTS_SALU"), "lds_insts_per_launch": warm("f32", "SQ_INSTS_LDS"),
        "vmem_insts_per_launch": warm("f32", "SQ_INSTS_VMEM"),
        "sq": {c: warm("sq", c) for c in passes["sq"][0][DOMINANT]},
    }
    json.dump(out, open(dst("dominant_kernel.json"), "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("commit", "valu_insts_per_particle", "valu_issue_frac", "clock_ghz",
                                          "traffic_bytes_per_launch", "l2_hit_rate")}, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
