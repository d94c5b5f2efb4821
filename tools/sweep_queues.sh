# GPU-box helper: pipelined throughput against the runtime's hardware-queue count, the number of batches in flight and
# the position of the bench's streams in the runtime's round-robin queue assignment
for HQ in ${HQS:-4 6 8}; do
  for F in ${FS:-3 4}; do
   for SK in ${SKS:-0 1 2 3}; do
  GPU_MAX_HW_QUEUES=$HQ python3 bench.py --steps 300 --warmup 10 --cpu-sample 0 --no-recall --in-flight $F --stream-skip $SK 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())
print('HQ=$HQ F=$F skip=$SK', j['value'], j['ms_per_step'], j.get('gather_verified'))"
   done
  done
done
