#!/bin/bash
# GPU-box helper: the library in the tree against the builds under tools/ab_*/ in ONE call, alternating, the driver's short run
# and a long one.  usage: tools/ab3.sh [rounds]
L=postgres-word2vec_amd/libfreddy_gpu.so
cp $L /tmp/new.so
for r in $(seq ${1:-2}); do
  for which in new $(ls -d tools/ab_*/ | xargs -n1 basename); do
    if [ $which = new ]; then cp /tmp/new.so $L; else cp tools/$which/libfreddy_gpu.so $L; fi
    for args in "--steps 20 --warmup 5" "--steps 300 --warmup 10"; do
      python3 bench.py --gpus 1 $args --cpu-sample 0 --no-recall --no-other-configs --no-host-abi 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())
ko = j['kernels_overlapped']
print('$which', '$args', j['value'], j['ms_per_step'], 'serial', j['pipelining']['serial_ms_per_step'], 'scan', j['kernels']['ivf_filter']['avg_us'], ko['ivf_filter']['avg_us'], 'merge', ko['merge_refine']['avg_us'], j['filter_bound_violations'])"
    done
  done
done
cp /tmp/new.so $L
