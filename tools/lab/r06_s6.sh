#!/bin/bash
mkdir -p gpurun_out/s6
python -m pytest tests/test_gpu_collective.py -x -q 2>&1 | tail -5
for cfg in "rccl 1" "rccl 4" "c10d 1"; do
  set -- $cfg
  python bench.py --force-collective --gather-path $1 --gather-every $2 --steps 200 --warmup 16 --no-other-configs --no-host-abi --no-recall --cpu-sample 0 > gpurun_out/s6/coll_$1_$2.out 2>gpurun_out/s6/coll_$1_$2.err
  tail -3 gpurun_out/s6/coll_$1_$2.err
  python - <<P
import json
d=json.load(open("bench_details_collective.json"))["collective_1rank"]
print("gather_path=$1 gather_every=$2", d["with_collective_qps"], d["without_qps"], d["ratio"], d["rounds"])
P
done 2>&1 | tee gpurun_out/s6/path_sweep.txt
