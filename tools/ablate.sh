#!/bin/bash
# GPU-box helper: fused-kernel time with parts switched off (timing only, results are wrong).
# 1 = no slab build, 2 = no gather, 4 = no selection, 8 = no codebook/code loads
for ab in ${ABLATES:-0 1 2 4 3 6 7 15}; do
  FREDDY_GPU_FUSED_ABLATE=$ab python bench.py --cpu-sample 0 --no-recall --steps 30 2>/dev/null | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('ablate', $ab, 'fused_us', j['kernels']['ivf_fused']['avg_us'], 'step_ms', j['ms_per_step'])"
done
