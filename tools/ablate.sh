#!/bin/bash
# GPU-box helper: scan-kernel time with parts switched off (timing only, results are wrong).
# 2 = no gathers, 4 = no selection, 8 = keep every row, 16 = no slab writes, 32 = no table loads, 64 = no code reloads
for ab in ${ABLATES:-0 2 4 6 16 32 48 50 54 118}; do
  FREDDY_GPU_FUSED_ABLATE=$ab FREDDY_GPU_FUSED_PROF=1 python bench.py --cpu-sample 0 --no-recall --no-other-configs --no-host-abi --in-flight 1 --steps 20 --warmup 3 2>/tmp/ab.err | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernels']; print('ablate', $ab, 'scan_us', (k.get('ivf_filter') or k.get('ivf_fused'))['avg_us'])"
  grep "scan prof" /tmp/ab.err | tail -1 | cut -c1-250
done
