"""Condense rocprofv3 CSV output (kernel stats / counter collection) to the kernels of this
library; run on the GPU box, writes small text files that travel back in gpurun_out/."""
import csv, glob, os, sys, collections

OURS = ("freddy",)


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def stats(d, out):
    f = find(d, "kernel_stats.csv")
    if not f:
        out.write("no kernel_stats.csv under %s\n" % d); return
    rows = list(csv.DictReader(open(f)))
    out.write("# rocprofv3 --kernel-trace --stats : kernels of libfreddy_gpu.so (full file: %d kernels)\n" % len(rows))
    out.write("%-72s %8s %14s %12s %8s\n" % ("Name", "Calls", "TotalDur(ns)", "Avg(ns)", "Pct"))
    for r in rows:
        if any(o in r["Name"] for o in OURS):
            out.write("%-72s %8s %14s %12s %8s\n" % (r["Name"][:72], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))


def counters(d, out, name):
    f = find(d, "counter_collection.csv")
    if not f:
        out.write("no counter_collection.csv under %s\n" % d); return
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != name: continue
        k = r["Kernel_Name"]
        if not any(o in k for o in OURS): continue
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    out.write("# rocprofv3 --pmc %s : per-dispatch average (raw counter units)\n" % name)
    for k, (n, v) in sorted(agg.items()):
        out.write("%-72s dispatches=%6d avg=%16.1f\n" % (k[:72], n, v / max(n, 1)))


if __name__ == "__main__":
    mode, d, outp = sys.argv[1], sys.argv[2], sys.argv[3]
    with open(outp, "w") as out:
        if mode == "stats": stats(d, out)
        else: counters(d, out, mode)
    print(open(outp).read())
