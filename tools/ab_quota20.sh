#!/bin/bash
# GPU-box helper: the driver's short run (--steps 20 --warmup 5) against quota-limited scan workgroups, several repetitions
# in ONE call (box-to-box variation is larger than the effect).   tools/ab_quota20.sh "<quota:wgs ...>" [reps]
CFG=${1:-"0:0 8:128 8:192 16:64"}; REPS=${2:-3}
for rep in $(seq $REPS); do for c in $CFG; do q=${c%%:*}; w=${c##*:}
  FREDDY_GPU_SCAN_QUOTA=$q FREDDY_GPU_SCAN_QUOTA_WGS=$w python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-recall --no-other-configs --no-host-abi 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rep $rep quota=$q wgs=$w  value %.2f M  ms/step %.4f' % (d['value']/1e6, d['ms_per_step']))"
done; done
