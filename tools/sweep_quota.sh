#!/bin/bash
# GPU-box helper: throughput of bench.py against the quota-limited scan workgroups.
#   tools/sweep_quota.sh "<quota list>" "<quota_wgs list>" [steps]
QS=${1:-"0 2 4 8"}; WS=${2:-"0"}; STEPS=${3:-100}
for q in $QS; do for w in $WS; do
  FREDDY_GPU_SCAN_QUOTA=$q FREDDY_GPU_SCAN_QUOTA_WGS=$w python3 bench.py --steps $STEPS --warmup 8 --cpu-sample 0 --no-recall --no-other-configs --no-host-abi 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); ko=d['kernels_overlapped']; k=d['kernels']
print('quota=$q wgs=$w  value %.2f M  ms/step %.4f  serial %.4f  scan alone %.1f overlapped %.1f  merge ov %.1f' % (d['value']/1e6, d['ms_per_step'], d['pipelining']['serial_ms_per_step'], k['ivf_filter']['avg_us'], ko['ivf_filter']['avg_us'], ko['merge_refine']['avg_us']))"
  [ "$q" = "0" ] && break
done; done
