"""Per-kernel averages of every counter in a rocprofv3 counter_collection.csv (our kernels only)."""
import csv, glob, os, sys, collections
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "freddy" not in k: continue
    k = k.split("(")[0].replace("void freddy::", "").replace("freddy::", "")[:40]
    a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    print(k)
    for c, (n, v) in sorted(agg[k].items()):
        print("   %-28s n=%4d avg=%18.1f" % (c, n, v / max(n, 1)))
