#!/bin/bash
mkdir -p gpurun_out/s5
for fake in none copy ""; do
  FREDDY_LAB_GATHER_FAKE=$fake python bench.py --force-collective --gather-stream own --gather-every 1 --steps 200 --warmup 16 --no-other-configs --no-host-abi --no-recall --cpu-sample 0 > gpurun_out/s5/coll_fake_$fake.out 2>gpurun_out/s5/coll_fake_$fake.err
  python - <<P
import json
d=json.load(open("bench_details_collective.json"))["collective_1rank"]
print("fake='$fake'", d["with_collective_qps"], d["without_qps"], d["ratio"], d["rounds"])
P
done 2>&1 | tee gpurun_out/s5/fake_sweep.txt
