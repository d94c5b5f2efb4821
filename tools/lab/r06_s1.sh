#!/bin/bash
# round 6, GPU session 1: the suite, the driver's bench command, host-buffer options A/B, backends with the library's own defaults,
# the one-rank collective in both gather modes
mkdir -p gpurun_out/s1
python -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/s1/pytest.txt; tail -5 gpurun_out/s1/pytest.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s1/bench20.out 2> gpurun_out/s1/bench20.err; cp bench_details.json gpurun_out/s1/bench20_details.json; cp bench_details_collective.json gpurun_out/s1/bench20_collective.json 2>/dev/null
tail -c 1500 gpurun_out/s1/bench20.out
GPU_MAX_HW_QUEUES=6 python tools/lab/host_abi_ab.py 30 > gpurun_out/s1/host_abi_ab.txt 2>&1
tail -40 gpurun_out/s1/host_abi_ab.txt
python tools/backends.py 3 > gpurun_out/s1/backends.txt 2>&1
cat gpurun_out/s1/backends.txt
for ge in 1 4; do for rc in 0 2; do
  python bench.py --force-collective --gather-every $ge --reserve-cus $rc --steps 200 --warmup 10 --no-other-configs --no-host-abi --no-recall --cpu-sample 0 > gpurun_out/s1/coll_ge${ge}_rc${rc}.out 2>gpurun_out/s1/coll_ge${ge}_rc${rc}.err
  python - <<P
import json
d=json.load(open("bench_details_collective.json"))["collective_1rank"]
print("gather_every=$ge reserve_cus=$rc", d["with_collective_qps"], d["without_qps"], d["ratio"], d["rounds"])
P
done; done 2>&1 | tee gpurun_out/s1/collective_sweep.txt
