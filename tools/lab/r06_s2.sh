#!/bin/bash
# round 6, GPU session 2
mkdir -p gpurun_out/s2
python -m pytest tests -m gpu -x -q > gpurun_out/s2/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/s2/pytest.txt; tail -5 gpurun_out/s2/pytest.txt
for own in 0 1; do
  echo "== FREDDY_GPU_LANE0_OWN=$own"
  FREDDY_GPU_LANE0_OWN=$own GPU_MAX_HW_QUEUES=6 python tools/lab/host_abi_ab.py 60 2>&1 | grep -v amdgpu.ids
done > gpurun_out/s2/host_abi_ab.txt 2>&1
cat gpurun_out/s2/host_abi_ab.txt
python tools/backends.py 3 > gpurun_out/s2/backends.txt 2>&1
cat gpurun_out/s2/backends.txt
for gs in own rccl; do for hwq in 8 12; do for ge in 4 1; do
  GPU_MAX_HW_QUEUES=$hwq python bench.py --force-collective --gather-stream $gs --gather-every $ge --steps 200 --warmup 10 --no-other-configs --no-host-abi --no-recall --cpu-sample 0 > gpurun_out/s2/coll_${gs}_q${hwq}_ge${ge}.out 2>gpurun_out/s2/coll_${gs}_q${hwq}_ge${ge}.err
  python - <<P
import json
d=json.load(open("bench_details_collective.json"))["collective_1rank"]
print("gather_stream=$gs hw_queues=$hwq gather_every=$ge", d["with_collective_qps"], d["without_qps"], d["ratio"], d["rounds"])
P
done; done; done 2>&1 | tee gpurun_out/s2/collective_sweep.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s2/bench20.out 2> gpurun_out/s2/bench20.err; cp bench_details.json gpurun_out/s2/bench20_details.json
tail -c 1200 gpurun_out/s2/bench20.out
