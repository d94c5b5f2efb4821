#!/bin/bash
# GPU-box helper: the pipelined bench N times (box-to-box and run-to-run noise is ~2 %): tools/rep.sh [N] [bench args]
N=${1:-3}; shift
for i in $(seq $N); do
python bench.py --cpu-sample 0 --no-recall --steps 400 "$@" | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['pipelining']['serial_ms_per_step'], {k: v['avg_us'] for k, v in j['kernels'].items()}, j['filter_bound_violations'], j['gather_verified'])"
done
