#!/bin/bash
# GPU-box helper: per-entry cycle breakdown of the scan (FREDDY_GPU_FUSED_PROF) for the library in the tree and tools/ab_head's
L=postgres-word2vec_amd/libfreddy_gpu.so
cp $L /tmp/new.so
for which in new base new base; do
  if [ $which = base ]; then cp tools/ab_head/libfreddy_gpu.so $L; else cp /tmp/new.so $L; fi
  echo "== $which"
  FREDDY_GPU_FUSED_PROF=1 python3 bench.py --steps 3 --warmup 2 --in-flight 1 --cpu-sample 0 --no-recall --no-other-configs --no-host-abi 2>&1 | grep "scan prof" | tail -1
  python3 bench.py --steps 50 --warmup 5 --in-flight 1 --cpu-sample 0 --no-recall --no-other-configs --no-host-abi 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['kernels']['ivf_filter'], d['kernels']['merge_refine'], d['filter_bound_violations'])"
done
cp /tmp/new.so $L
