export TMPDIR=/tmp
mkdir -p gpurun_out/prof
python3 tools/work_stats.py > gpurun_out/prof/work_stats.json 2>/dev/null
cd /tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d /tmp/p1 -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-recall --no-other-configs --no-host-abi --in-flight 1 > /tmp/p1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d /tmp/p2 -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-recall --no-other-configs --no-host-abi --in-flight 1 > /tmp/p2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for d in ("/tmp/p1", "/tmp/p2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in agg.items():
            if "filter5" in k or "merge_refine" in k:
                print(k, {c: (len(v), sum(v) / len(v)) for c, v in cs.items()})
PY
tail -2 /tmp/p1.log | cut -c1-300
