"""GPU-box helper: the multi-stream part of a bench.py kernel trace (warm-up + timed steps on the in-flight streams):
per step the start / end of its chain and of its scan, the idle time of the chip between chains.
usage: python tools/trace_streams.py DIR"""
import csv, glob, sys, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ours = [r for r in rows if "freddy" in r["Kernel_Name"] and "row_term" not in r["Kernel_Name"]]
short = lambda r: r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]
# chains: group by stream, split at query_codebook5 (first kernel of a step)
by = collections.defaultdict(list)
for r in ours:
    by[r["Stream_Id"]].append(r)
multi = [s for s, v in by.items() if sum(short(x).startswith("merge_refine") for x in v) >= 5]
print("streams with searches:", {s: len(by[s]) for s in multi})
first = min(int(by[s][0]["Start_Timestamp"]) for s in multi)
steps = []
for s in multi:
    cur = None
    for r in by[s]:
        n = short(r)
        if n.startswith("query_codebook"):
            cur = {"stream": s, "start": int(r["Start_Timestamp"]), "scan": None, "end": None}
            steps.append(cur)
        if cur is None:
            continue
        if n.startswith("ivf_filter"):
            cur["scan"] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
        if n.startswith("merge_refine"):
            cur["end"] = int(r["End_Timestamp"])
steps = [s for s in steps if s["end"]]
steps.sort(key=lambda s: s["start"])
for i, s in enumerate(steps[:60]):
    print(f"step {i:3d} stream {s['stream']:>3}  start {(s['start']-first)/1e3:9.1f}  scan {(s['scan'][0]-first)/1e3:9.1f}..{(s['scan'][1]-first)/1e3:9.1f} ({(s['scan'][1]-s['scan'][0])/1e3:6.1f})  end {(s['end']-first)/1e3:9.1f}  chain {(s['end']-s['start'])/1e3:6.1f}")
