#!/bin/bash
# K = 256 (the reference's shipped default): the whole-entry-slab kernel (codes_u8 = 1, fused8.h) against fused5.h's one-byte
# instantiation (2) and the int16 layout (0), alternating on one box
for rnd in 1 2; do for u8 in 1 2 0; do
  FREDDY_GPU_CODES_U8=$u8 python bench.py --K 256 --steps ${1:-200} --warmup 10 --no-other-configs --no-host-abi --no-recall --cpu-sample 1024 --no-collective-child > /dev/null 2>/tmp/e.txt || tail -3 /tmp/e.txt
  python - <<P
import json
d=json.load(open("bench_details.json"))
k={n:v["avg_us"] for n,v in d["kernels"].items()}; ko={n:v["avg_us"] for n,v in d["kernels_overlapped"].items()}
print("codes_u8=$u8: %.3f M q/s  %.4f ms  serial %.4f  scan %s / %s  merge %s / %s  parity %s violations %s" % (d["value"]/1e6, d["ms_per_step"], d["pipelining"]["serial_ms_per_step"], k.get("ivf_filter"), ko.get("ivf_filter"), k.get("merge_refine"), ko.get("merge_refine"), d["cpu_baseline"]["parity_with_gpu_on_sample"], d["filter_bound_violations"]))
P
done; done
