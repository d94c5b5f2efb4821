#!/bin/bash
# GPU-box helper: the three rocprofv3 passes behind profiles/<tag>_* (kernel trace + stats, and one
# PMC pass each for FETCH_SIZE and WRITE_SIZE -- they do not fit one pass on gfx950), condensed to
# small text/JSON files under gpurun_out/prof/.
#   usage: tools/profile_round.sh <tag> [extra bench.py args]
set -u
TAG=${1:-rXX}; shift || true
export TMPDIR=/tmp
T=/tmp/prof_$TAG; rm -rf $T; mkdir -p $T gpurun_out/prof
# (--in-flight 1: one batch at a time, so that a kernel's duration in the trace is the kernel alone -- what bench.py's
#  instrumented re-run and its roofline object report; the default command's overlapped timeline is the last pass)
B="python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-recall --no-other-configs --no-host-abi --in-flight 1 $*"
B2="python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-recall --no-other-configs --no-host-abi $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $T/trace -o r -- $B > gpurun_out/prof/${TAG}_bench_under_trace.json 2> $T/trace.err
python3 tools/prof_summary.py stats $T/trace gpurun_out/prof/${TAG}_kernel_stats.txt > /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $T/fetch -o r -- $B > /dev/null 2> $T/fetch.err
python3 tools/prof_summary.py FETCH_SIZE $T/fetch gpurun_out/prof/${TAG}_pmc_fetch_size.txt > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $T/write -o r -- $B > /dev/null 2> $T/write.err
python3 tools/prof_summary.py WRITE_SIZE $T/write gpurun_out/prof/${TAG}_pmc_write_size.txt > /dev/null
python3 - "$TAG" "$@" <<'PY'
import json, re, sys
tag = sys.argv[1]
extra = sys.argv[2:]
out = {}
for name, key in (("fetch_size", "fetch_kib"), ("write_size", "write_kib")):
    for line in open(f"gpurun_out/prof/{tag}_pmc_{name}.txt"):
        m = re.match(r"(.*?)\s+dispatches=\s*(\d+)\s+avg=\s*([0-9.]+)", line)
        if m:
            k = m.group(1).split("(")[0].replace("void ", "").replace("freddy::", "").strip()
            k = re.sub(r"<.*", "", k)
            # (template instantiations share a key: the one that moves the most bytes is the kernel a roofline is written for --
            #  exf_filter_kernel<2, false> reads the table, <2, true> only the threshold sample)
            if float(m.group(3)) >= out.get(k, {}).get(key, -1.0):
                out.setdefault(k, {})[key] = float(m.group(3))
# the workload shape these passes were taken on: bench.py's pmc_traffic() refuses the record for any other shape
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="ivfadc"); ap.add_argument("--N", type=int); ap.add_argument("--Q", type=int)
ap.add_argument("--C", type=int, default=1000); ap.add_argument("--nprobe", type=int, default=10); ap.add_argument("--K", type=int, default=1024)
a, _ = ap.parse_known_args(extra)
dflt = {"ivfadc": (3_000_000, 1024), "pq": (1_000_000, 64), "join": (1_000_000, 5000), "exact": (3_000_000, 64)}[a.config]
shape = {"N": a.N or dflt[0], "Q": a.Q or dflt[1]}
if a.config == "ivfadc":
    shape.update({"C": a.C, "nprobe": a.nprobe})
    if a.K != 1024:
        shape["K"] = a.K
out["_shape"] = shape
json.dump(out, open(f"gpurun_out/prof/{tag}_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $T/trace2 -o r -- $B2 > gpurun_out/prof/${TAG}_bench_under_trace_in_flight.json 2> $T/trace2.err
python3 tools/prof_summary.py stats $T/trace2 gpurun_out/prof/${TAG}_kernel_stats_in_flight.txt > /dev/null
python3 tools/timeline.py $T/trace2 60 --overlapped > gpurun_out/prof/${TAG}_timeline_in_flight.txt 2>/dev/null
cat gpurun_out/prof/${TAG}_kernel_stats.txt
