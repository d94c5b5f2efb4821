"""GPU-box helper: kernels AND memory copies of a rocprofv3 trace (--kernel-trace --memory-copy-trace) on one time line -- what the
one-rank RCCL all_gather (a device-to-device copy) does to the four searching streams.   usage: coll_trace.py DIR [n_events]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1][:30]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f'q={r.get("Queue_Id", "?"):>3} st={r.get("Stream_Id", "?"):>3}', name))
ncopy = 0
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f'st={r.get("Stream_Id", "?"):>3}', "COPY " + r.get("Direction", "?") + " " + r.get("Size", r.get("Bytes", "?"))))
        ncopy += 1
ev.sort()
print(f"# {len(ev)} events, {ncopy} copies")
# a device-to-device copy of the collective = a __amd_rocclr_copyBuffer blit kernel in the neighbourhood of scan kernels
names = [e[3] for e in ev]
idx = [i for i, nm in enumerate(names) if "copyBuffer" in nm and any("ivf_filter5" in x for x in names[max(0, i - 25):i + 25])]
print(f"# {len(idx)} copies among the search kernels")
mid = idx[len(idx) * 3 // 4] if idx else len(ev) // 2
lo = max(0, mid - n // 2)
t0 = ev[lo][0]
for s, e, where, name in ev[lo:lo + n]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {where:14s} {name}")
