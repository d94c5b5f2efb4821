# GPU-box helper: pipelined throughput against the CU partition (R > 0: small kernels confined to R CUs and the scan to the rest;
# R < 0: only the scan masked, to n_cus - |R| CUs)
for R in ${RS:-0 -8 -16 -32 0}; do
  for F in ${FS:-3 4}; do
  FREDDY_GPU_PARTITION_CUS=$R python3 bench.py --steps 300 --warmup 10 --cpu-sample 0 --no-recall --in-flight $F 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())
print('R=$R F=$F', j['value'], j['ms_per_step'], j.get('gather_verified'), j.get('filter_bound_violations'), j['pipelining']['serial_ms_per_step'], j['kernels']['ivf_filter']['avg_us'])"
  done
done
