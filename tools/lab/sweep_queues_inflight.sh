for hq in ${HQS:-4 5 6 8}; do for nf in ${NFS:-3 4 5 6}; do
GPU_MAX_HW_QUEUES=$hq python bench.py --steps 40 --warmup 8 --in-flight $nf --no-other-configs --no-host-abi --no-recall --cpu-sample 0 > /dev/null 2>&1
python - "$hq" "$nf" <<'P'
import json,sys
o=json.load(open("bench_details.json")); print("hwq",sys.argv[1],"in_flight",sys.argv[2], round(o["value"]/1e6,3), o["ms_per_step"])
P
done; done
