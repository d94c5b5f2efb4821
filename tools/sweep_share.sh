# GPU-box helper: pipelined throughput against the batches in flight, the scan's share of the chip and the hardware queues
for HQ in ${HQS:-6}; do
for F in ${FS:-4}; do
  for SH in ${SHS:-0}; do
  GPU_MAX_HW_QUEUES=$HQ python3 bench.py --steps 300 --warmup 10 --cpu-sample 0 --no-recall --no-other-configs --no-host-abi --in-flight $F --scan-share $SH 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())
print('HQ=$HQ F=$F share=$SH', j['value'], j['ms_per_step'], j.get('timed_region_parity'), j['kernels_overlapped']['ivf_filter']['avg_us'])"
  done
done
done
