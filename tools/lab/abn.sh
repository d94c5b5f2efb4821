#!/bin/bash
# GPU-box helper: several builds of the library alternating in ONE gpurun call (box-to-box variation is larger than most kernel changes).
# usage: tools/lab/abn.sh <rounds> "<so paths, space separated>" [bench args]
N=${1:-2}; SOS=$2; shift 2
for r in $(seq $N); do
  for so in $SOS; do
    export FREDDY_GPU_SO=$PWD/$so
    python bench.py --no-other-configs --no-host-abi --no-recall --cpu-sample 0 --no-collective-child "$@" > /tmp/ab.out 2>/dev/null
    python - "$so" <<'P'
import json, sys
o = json.load(open("bench_details.json"))
k = {n: v["avg_us"] for n, v in o["kernels"].items()}
ko = {n: v["avg_us"] for n, v in o["kernels_overlapped"].items()}
print(f"{sys.argv[1].split('libfreddy_gpu')[-1]:10s} {o['value']/1e6:6.3f} M q/s  {o['ms_per_step']:.4f} ms  serial {o['pipelining']['serial_ms_per_step']:.4f}  scan {k.get('ivf_filter')} / {ko.get('ivf_filter')}  merge {k.get('merge_refine')} / {ko.get('merge_refine')}  coarse {k.get('coarse_table')} / {ko.get('coarse_table')}  plan {k.get('probe_plan')} / {ko.get('probe_plan')}")
P
  done
done
