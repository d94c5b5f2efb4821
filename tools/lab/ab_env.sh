#!/bin/bash
# GPU-box helper: ONE library, an environment variable (a tuning option's FREDDY_GPU_* name) at two values, alternating in one
# gpurun call.  usage: tools/lab/ab_env.sh NAME VALUE_A VALUE_B [rounds] [bench args]
NAME=$1; A=$2; B=$3; N=${4:-2}; shift 4
for r in $(seq $N); do
  for v in $A $B; do
    env $NAME=$v python bench.py --no-other-configs --no-host-abi --no-recall --cpu-sample 0 "$@" > /tmp/ab.out 2>/dev/null
    python - "$NAME=$v" <<'P'
import json, sys
o = json.load(open("bench_details.json"))
k = {n: v["avg_us"] for n, v in o["kernels"].items()}
ko = {n: v["avg_us"] for n, v in o["kernels_overlapped"].items()}
print(f"{sys.argv[1]:32s} {o['value']/1e6:6.3f} M q/s  {o['ms_per_step']:.4f} ms  serial {o['pipelining']['serial_ms_per_step']:.4f}  scan {k.get('ivf_filter')} / {ko.get('ivf_filter')}  plan {k.get('probe_plan')} / {ko.get('probe_plan')}  wt {k.get('work_table')}")
P
  done
done
