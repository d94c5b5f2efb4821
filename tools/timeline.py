"""GPU-box helper: print the kernel timeline (start / end in us, queue) of the last steps of a rocprofv3 kernel trace.
usage: python tools/timeline.py DIR [n_kernels] [--overlapped]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 40
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append(r)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if "--overlapped" in sys.argv:
    # bench.py ends with one-batch-at-a-time runs on ONE stream: show the last n kernels of the part before, where
    # the library's kernels come from several streams
    ours = [i for i, r in enumerate(rows) if "freddy" in r["Kernel_Name"]]
    last_stream = rows[ours[-1]].get("Stream_Id")
    end = max(i for i in ours if rows[i].get("Stream_Id") != last_stream and "row_term" not in rows[i]["Kernel_Name"])
    rows = rows[:end + 1]
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1][:28]
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:9.1f}  q={r.get("Queue_Id", "?"):>3} st={r.get("Stream_Id", "?"):>3} {name}')
