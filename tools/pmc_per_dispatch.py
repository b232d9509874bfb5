import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
name = sys.argv[2] if len(sys.argv) > 2 else "k_iter"
rows = [r for r in csv.DictReader(open(f)) if name in r["Kernel_Name"].split("(")[0]]
d = collections.defaultdict(dict)
for r in rows:
    d[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(d):
    print(k, {c: "%.3e" % v for c, v in sorted(d[k].items())})
