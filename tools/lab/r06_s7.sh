#!/bin/bash
mkdir -p gpurun_out/s7
for i in 1 2 3; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-host-abi --no-recall --cpu-sample 0 --no-collective-child > gpurun_out/s7/b20_$i.out 2>/dev/null
python - <<P
import json
d=json.load(open("bench_details.json")); print("20 steps:", d["value"], d["ms_per_step"], d["pipelining"]["serial_ms_per_step"])
P
done
python bench.py --gpus 1 --steps 300 --warmup 10 --no-other-configs --no-host-abi --no-recall --cpu-sample 0 --no-collective-child > gpurun_out/s7/b300.out 2>/dev/null
python - <<P
import json
d=json.load(open("bench_details.json")); print("300 steps:", d["value"], d["ms_per_step"], d["pipelining"]["serial_ms_per_step"])
P
python tools/backends.py 3 3000000 "1:0:0,2:0:0,4:0:0,8:0:0,1:1:6,4:2:2" 2>&1 | grep -v amdgpu | tee gpurun_out/s7/backends.txt
