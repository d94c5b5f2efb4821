"""GPU-box helper: the last kernels of a rocprofv3 kernel trace before / at a marker kernel, as a timeline in us.
usage: python tools/trace_tail.py DIR MARKER_SUBSTRING N"""
import csv, glob, sys
d, marker, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda r: r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1][:44]
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
end = idx[-1]
seg = rows[max(0, end - n + 1):end + 1]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    print("%9.1f %9.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, short(r)))
