"""Summarise rocprofv3 --pmc CSVs per kernel (mean over dispatches and the last, warm, dispatch)."""
import collections
import csv
import glob
import sys


def load(d):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]      # "void k_iter<false>(...)" -> k_iter
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg


if __name__ == "__main__":
    for d in sys.argv[1:]:
        agg = load(d)
        for k in ('k_iter', 'k_density', 'k_wvt', 'k_cells', 'k_keys', 'k_permute'):
            if k in agg:
                print(d, k)
                for c, v in sorted(agg[k].items()):
                    print('   %-28s dispatches=%d last=%.5g mean=%.5g' % (c, len(v), v[-1], sum(v) / len(v)))
