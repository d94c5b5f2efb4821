#!/bin/bash
# GPU-box helper: tools/ab.sh for the driver's short run (--steps 20 --warmup 5) and the long one, alternating builds
L=postgres-word2vec_amd/libfreddy_gpu.so
cp $L /tmp/new.so
for r in 1 2; do
  for which in new base; do
    if [ $which = base ]; then cp tools/ab_head/libfreddy_gpu.so $L; else cp /tmp/new.so $L; fi
    for args in "--steps 20 --warmup 5" "--steps 300 --warmup 10"; do
      python3 bench.py --gpus 1 $args --cpu-sample 0 --no-recall 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$which', '$args', j['value'], j['ms_per_step'], j['pipelining']['serial_ms_per_step'], j['kernels']['query_codebook']['avg_us'], j['kernels_overlapped']['query_codebook']['avg_us'], j['filter_bound_violations'])"
    done
  done
done
cp /tmp/new.so $L
