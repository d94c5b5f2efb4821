"""GPU-box helper: idle time between consecutive kernels of the bench step, from a rocprofv3 kernel trace
(usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o r -- python3 bench.py ... ; python3 tools/gap_stats.py DIR)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    name = r.get("Kernel_Name") or r.get("kernel_name")
    rows.append((int(r.get("Start_Timestamp") or r.get("start_timestamp")), int(r.get("End_Timestamp") or r.get("end_timestamp")), name))
rows.sort()
short = lambda n: n.split("(")[0].replace("void ", "").replace("freddy::", "").split("<")[0]
gaps = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    if "freddy" in n0 and "freddy" in n1:
        gaps[(short(n0), short(n1))].append(s1 - e0)
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 10:
        v2 = sorted(v)
        print(f"{k[0]:28s} -> {k[1]:28s} n={len(v):4d} median gap {v2[len(v2)//2]/1000:8.2f} us  mean {sum(v)/len(v)/1000:8.2f} us")
