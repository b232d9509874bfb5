import csv, sys, glob, collections
def load(d):
    f=glob.glob(d+'/*/*counter_collection.csv')[0]
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k=r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg
for d in sys.argv[1:]:
    agg=load(d)
    for k in ('k_density','k_wvt'):
        if k in agg:
            print(d,k)
            for c,v in agg[k].items():
                print('   %-28s n=%d last=%.4g mean=%.4g'%(c,len(v),v[-1],sum(v)/len(v)))
