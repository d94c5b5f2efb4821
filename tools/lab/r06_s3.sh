#!/bin/bash
# round 6, GPU session 3: what the one-rank collective costs and where (host time per step; SDMA copy of RCCL's one-rank shortcut)
mkdir -p gpurun_out/s3
run() { # name, env..., -- args
  name=$1; shift
  env "$@" > /dev/null 2>&1 || true
}
for cfg in "own 1 1" "own 4 1" "own 16 1" "own 1 0" "own 4 0" "rccl 4 0" "rccl 1 0"; do
  set -- $cfg; gs=$1; ge=$2; sdma=$3
  HSA_ENABLE_SDMA=$sdma python bench.py --force-collective --gather-stream $gs --gather-every $ge --steps 200 --warmup 16 --no-other-configs --no-host-abi --no-recall --cpu-sample 0 > gpurun_out/s3/coll_${gs}_ge${ge}_sdma${sdma}.out 2>gpurun_out/s3/coll_${gs}_ge${ge}_sdma${sdma}.err
  python - <<P
import json
d=json.load(open("bench_details_collective.json"))["collective_1rank"]
print("gather_stream=$gs gather_every=$ge HSA_ENABLE_SDMA=$sdma", d["with_collective_qps"], d["without_qps"], d["ratio"], d["rounds"])
P
done 2>&1 | tee gpurun_out/s3/collective_sweep.txt
# larger steps: the host's share of a step shrinks
for Q in 4096; do
  python bench.py --force-collective --gather-stream own --gather-every 1 --Q $Q --steps 100 --warmup 16 --no-other-configs --no-host-abi --no-recall --cpu-sample 0 > gpurun_out/s3/coll_Q$Q.out 2>gpurun_out/s3/coll_Q$Q.err
  python - <<P
import json
d=json.load(open("bench_details_collective.json"))["collective_1rank"]
print("Q=$Q gather_stream=own gather_every=1", d["with_collective_qps"], d["without_qps"], d["ratio"], d["rounds"])
P
done 2>&1 | tee -a gpurun_out/s3/collective_sweep.txt
