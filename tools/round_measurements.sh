#!/bin/bash
# GPU-box helper: everything profiles/<tag>_* holds for one round.
#   tools/round_measurements.sh <tag>
set -u
TAG=${1:-rXX}
mkdir -p gpurun_out/prof
# bench.py prints a digest line and the short headline; the full record goes to bench_details.json: kept per run
bench() { local out=$1; shift; timeout ${TMO:-3000} python3 bench.py "$@" > gpurun_out/prof/${out}.json 2> gpurun_out/prof/${out}.err; cp bench_details.json gpurun_out/prof/${out}_details.json 2>/dev/null; }
tools/profile_round.sh $TAG > gpurun_out/prof/${TAG}_profile.log 2>&1
cp gpurun_out/prof/${TAG}_pmc.json profiles/latest_pmc.json
bench ${TAG}_bench
# configs 2 and 4 (BASELINE configs[1], configs[3]) with their own kernel stats + PMC passes
for cfg in pq join exact; do
  tools/profile_round.sh ${TAG}_${cfg} --config $cfg > gpurun_out/prof/${TAG}_${cfg}_profile.log 2>&1
  cp gpurun_out/prof/${TAG}_${cfg}_pmc.json profiles/latest_pmc_${cfg}.json 2>/dev/null
  python3 - <<PY
import json
a = json.load(open("profiles/latest_pmc.json")) if __import__("os").path.exists("profiles/latest_pmc.json") else {}
try:
    b = json.load(open("gpurun_out/prof/${TAG}_${cfg}_pmc.json"))
    for k, v in b.items():
        a.setdefault(k, v)
    json.dump(a, open("profiles/latest_pmc.json", "w"), indent=1)
except Exception as e:
    print("pmc merge:", e)
PY
  bench ${TAG}_config_${cfg} --config $cfg --steps 20 --warmup 3
done
cp profiles/latest_pmc.json gpurun_out/prof/${TAG}_latest_pmc.json
# larger batches on one GPU (SURVEY 8e asks for a Q >= 8192 variant)
bench ${TAG}_bench_Q8192 --steps 20 --warmup 4 --Q 8192 --cpu-sample 0 --no-recall --no-host-abi
bench ${TAG}_bench_300steps --steps 300 --warmup 10 --no-other-configs
# the same index shape with the reference's default codebook size K = 256 (one byte per code: ivf_filter8_kernel)
bench ${TAG}_bench_K256 --steps 300 --warmup 10 --K 256 --cpu-sample 64 --no-recall --no-host-abi --no-other-configs --no-collective-child
# a corpus that does NOT fit the 256 MiB Infinity Cache: N = 40 M rows (1.1 GB of lists), same list length
# with its OWN FETCH / WRITE passes (bench.py's pmc_traffic() refuses the 3 M-row record for this shape)
cp profiles/latest_pmc.json gpurun_out/prof/${TAG}_latest_pmc_3M.json
timeout 1200 tools/profile_round.sh ${TAG}_N40M --N 40000000 --C 13000 > gpurun_out/prof/${TAG}_N40M_profile.log 2>&1
if [ -s gpurun_out/prof/${TAG}_N40M_pmc.json ]; then cp gpurun_out/prof/${TAG}_N40M_pmc.json profiles/latest_pmc.json; fi
TMO=1500 bench ${TAG}_bench_N40M --steps 20 --warmup 4 --N 40000000 --C 13000 --cpu-sample 64 --no-recall --no-host-abi
cp gpurun_out/prof/${TAG}_latest_pmc_3M.json profiles/latest_pmc.json
# the same non-resident corpus with the reference's default codebook size K = 256: one byte per code (16 B per row)
TMO=1500 bench ${TAG}_bench_N40M_K256 --steps 20 --warmup 4 --N 40000000 --C 13000 --K 256 --cpu-sample 64 --no-recall --no-host-abi
FREDDY_GPU_CODES_U8=0 TMO=1500 bench ${TAG}_bench_N40M_K256_int16 --steps 20 --warmup 4 --N 40000000 --C 13000 --K 256 --cpu-sample 0 --no-recall --no-host-abi
tail -2 gpurun_out/prof/${TAG}_bench_N40M.err
ls -la gpurun_out/prof | tail -30
# the driver's own invocation (20 steps), LDS counters of the scan, the reference's other index shape, small-batch latencies
bench ${TAG}_bench_20steps --steps 20 --warmup 5
bash tools/lds_pmc.sh > gpurun_out/prof/${TAG}_lds_counters.txt 2>&1
python3 tools/other_shape.py > gpurun_out/prof/${TAG}_other_shape.json 2> /dev/null
python3 tools/latency.py > gpurun_out/prof/${TAG}_latency.txt 2> /dev/null
python3 tools/single_query.py > gpurun_out/prof/${TAG}_single_query.txt 2> /dev/null
# ONE query per call (the reference's own call shape): the one-launch kernels under the kernel trace + their in-kernel phases
export TMPDIR=/tmp
rm -rf /tmp/prof_${TAG}_lat; mkdir -p /tmp/prof_${TAG}_lat
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_lat -o r -- python3 tools/latency.py > /dev/null 2> /dev/null
python3 tools/prof_summary.py stats /tmp/prof_${TAG}_lat gpurun_out/prof/${TAG}_latency_kernel_stats.txt > /dev/null
FREDDY_GPU_ONE_PROF=1 python3 tools/latency.py 2>&1 | grep "ivf_one\]" | tail -3 > gpurun_out/prof/${TAG}_one_launch_phases.txt
