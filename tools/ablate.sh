#!/bin/bash
# GPU-box helper: scan-kernel time with parts switched off (timing only, results are wrong).
# 2 = no gathers, 4 = no selection, 8 = keep every row, 16 = no slab writes, 32 = no table loads, 64 = no code reloads
for ab in ${ABLATES:-0 2 4 16 32 48 50 54}; do
  FREDDY_GPU_FUSED_ABLATE=$ab python bench.py --cpu-sample 0 --no-recall --steps 30 2>/dev/null | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernels']; print('ablate', $ab, 'scan_us', (k.get('ivf_filter') or k.get('ivf_fused'))['avg_us'], 'step_ms', j['ms_per_step'])"
done
