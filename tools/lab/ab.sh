#!/bin/bash
# GPU-box helper: the library in the tree against postgres-word2vec_amd/libfreddy_gpu_base.so (a build of an earlier commit),
# alternating in ONE gpurun call: box-to-box variation (+-5 %) is larger than most kernel changes.
# usage: tools/lab/ab.sh [rounds] [bench args]
N=${1:-2}; shift
for r in $(seq $N); do
  for v in new base; do
    if [ $v = base ]; then export FREDDY_GPU_SO=$PWD/postgres-word2vec_amd/libfreddy_gpu_base.so; else unset FREDDY_GPU_SO; fi
    python bench.py --no-other-configs --no-host-abi --no-recall --cpu-sample 0 "$@" > /tmp/ab.out 2>/dev/null
    python - "$v" <<'P'
import json, sys
o = json.load(open("bench_details.json"))
k = {n: v["avg_us"] for n, v in o["kernels"].items()}
ko = {n: v["avg_us"] for n, v in o["kernels_overlapped"].items()}
print(f"{sys.argv[1]:5s} {o['value']/1e6:6.3f} M q/s  {o['ms_per_step']:.4f} ms  serial {o['pipelining']['serial_ms_per_step']:.4f}  scan {k.get('ivf_filter')} / {ko.get('ivf_filter')}  coarse {k.get('coarse_table')} / {ko.get('coarse_table')}  merge {k.get('merge_refine')} / {ko.get('merge_refine')}  plan {k.get('probe_plan')}  wt {k.get('work_table')}  rec {k.get('entry_records')}")
P
  done
done
