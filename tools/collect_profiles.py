"""Copy the summaries of one tools/profile_round.sh run (gpurun_out/prof_<tag>) into profiles/ under the
round's names and refresh profiles/dominant_kernel_traffic.json (read by bench.py).
Usage: python tools/collect_profiles.py gpurun_out/prof_r1f"""
import csv, glob, json, re, shutil, sys

P = sys.argv[1]
shutil.copy(glob.glob(P + '/trace/*/*_kernel_stats.csv')[0], 'profiles/round1_final_kernel_stats.csv')
shutil.copy(P + '/pmc_summary.txt', 'profiles/round1_final_pmc_summary.txt')
shutil.copy(P + '/bench_under_trace.json', 'profiles/round1_final_bench_under_trace.json')
f = glob.glob(P + '/trace/*/*_kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_iter' in r['Kernel_Name'].split('(')[0]]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
b = json.loads([l for l in open(P + '/bench_under_trace.json') if l.startswith('{')][-1])
out = ["k_iter dispatches of `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline`",
       "(kernel_trace.csv, End-Start, ms). Dispatches 1-2 are the untimed warm-up steps (1 = cold start)."]
out += ["  dispatch %d: %.3f ms" % (i + 1, x) for i, x in enumerate(d)]
out.append("mean of all %d (the --stats AverageNs): %.3f ms" % (len(d), sum(d) / len(d)))
out.append("mean of the 5 timed dispatches:        %.3f ms" % (sum(d[-5:]) / 5))
out.append("bench.py roofline.avg_launch_ms (HIP events on the library stream, same run): %.3f ms" % b['roofline']['avg_launch_ms'])
open('profiles/round1_final_k_iter_dispatches.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
s = open(P + '/pmc_summary.txt').read()


def last(sec, name):
    m = re.search(sec + r' k_iter\n(?:.*\n)*?\s+' + name + r'\s+dispatches=\d+ last=([0-9.e+]+)', s)
    return float(m.group(1))


fetch, write = last('pmc_fetch', 'FETCH_SIZE'), last('pmc_write', 'WRITE_SIZE')
hit, miss = last('pmc_write', 'TCC_HIT_sum'), last('pmc_write', 'TCC_MISS_sum')
t = json.load(open('profiles/dominant_kernel_traffic.json'))
t.update(bytes_per_launch=(fetch + write) * 1024, fetch_size_kb=fetch, write_size_kb=write, l2_hit_rate=hit / (hit + miss),
         valu_insts_per_launch=last('pmc_sq', 'SQ_INSTS_VALU'))
json.dump(t, open('profiles/dominant_kernel_traffic.json', 'w'), indent=1)
print({k: t[k] for k in ('bytes_per_launch', 'l2_hit_rate', 'valu_insts_per_launch')})
