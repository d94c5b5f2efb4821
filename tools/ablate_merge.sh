#!/bin/bash
# GPU-box helper: merge_refine_kernel time with parts switched off (timing only, results are wrong).
# 1 = skip the exact stage, 2 = no query staging / E, 8 = return after pass 1 (selection of the L smallest lower bounds)
# usage: tools/ablate_merge.sh [--config pq]
for ab in ${ABLATES:-0 1 8 10}; do
  FREDDY_GPU_MERGE_ABLATE=$ab python bench.py "$@" --cpu-sample 0 --no-recall --steps 30 --in-flight 1 --no-other-configs --no-host-abi 2>/dev/null | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernels']; print('merge ablate', $ab, 'merge_us', k['merge_refine']['avg_us'], 'step_ms', j['ms_per_step'])"
done
