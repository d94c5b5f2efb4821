# GPU-box helper: pipelined throughput against the number of CUs the persistent scan leaves free (no masks),
# the batches in flight and the runtime's hardware-queue count
for HQ in ${HQS:-6}; do
for R in ${RS:-0 8 16 24 32 48 64}; do
  for F in ${FS:-4}; do
  GPU_MAX_HW_QUEUES=$HQ FREDDY_GPU_RESERVE_CUS=$R python3 bench.py --steps 300 --warmup 10 --cpu-sample 0 --no-recall --in-flight $F 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())
print('HQ=$HQ reserve=$R F=$F', j['value'], j['ms_per_step'], j.get('gather_verified'), j['pipelining']['serial_ms_per_step'], j['kernels']['ivf_filter']['avg_us'])"
  done
done
done
