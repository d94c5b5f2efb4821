for oc in "" "--one-comm"; do for i in 1 2; do
  python bench.py --force-collective $oc --steps 200 --warmup 16 --no-other-configs --no-host-abi --no-recall --cpu-sample 0 > /dev/null 2>/tmp/e.txt || tail -5 /tmp/e.txt
  python - <<P
import json
d=json.load(open("bench_details_collective.json")); c=d["collective_1rank"]
print("one_comm='$oc'", d["config"].get("communicators"), c["with_collective_qps"], c["without_qps"], c["ratio"], c["gather_verified"])
P
done; done
