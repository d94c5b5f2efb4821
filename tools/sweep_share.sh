# GPU-box helper: pipelined throughput against the scan's share of the chip (0 = auto), the batches in flight and the hardware queues
for HQ in ${HQS:-6}; do
for SH in ${SHS:-0 1 2 3 4 5}; do
  for F in ${FS:-4}; do
  GPU_MAX_HW_QUEUES=$HQ FREDDY_GPU_SCAN_SHARE=$SH python3 bench.py --steps 300 --warmup 10 --cpu-sample 0 --no-recall --in-flight $F 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())
print('HQ=$HQ share=$SH F=$F', j['value'], j['ms_per_step'], j.get('gather_verified'), j['pipelining']['serial_ms_per_step'], j['kernels']['ivf_filter']['avg_us'])"
  done
done
done
