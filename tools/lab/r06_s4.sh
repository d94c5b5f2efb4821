#!/bin/bash
export TMPDIR=/tmp
cd /tmp 2>/dev/null; cd - >/dev/null
mkdir -p gpurun_out/s4
T=/tmp/prof_coll; rm -rf $T; mkdir -p $T
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $T/t1 -o r -- python3 bench.py --force-collective --gather-stream own --gather-every 1 --steps 60 --warmup 8 --ab-rounds 1 --no-other-configs --no-host-abi --no-recall --cpu-sample 0 > gpurun_out/s4/trace_run.out 2> gpurun_out/s4/trace_run.err
python3 tools/lab/coll_trace.py $T/t1 220 > gpurun_out/s4/coll_timeline.txt
head -5 gpurun_out/s4/coll_timeline.txt
ls $T/t1/* | head
