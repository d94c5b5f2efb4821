export TMPDIR=/tmp
T=/tmp/prof_tl; rm -rf $T; mkdir -p $T gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $T/trace2 -o r -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-recall --no-other-configs --no-host-abi > gpurun_out/prof/tl_bench.json 2> $T/trace2.err
python3 tools/timeline.py $T/trace2 90 --overlapped > gpurun_out/prof/tl_timeline_in_flight.txt
tail -3 $T/trace2.err
