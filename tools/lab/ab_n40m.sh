# GPU-box helper: the library in the tree against postgres-word2vec_amd/libfreddy_gpu_prev.so (a build of the commit before) on the 40 M-row corpus, alternating in one call
for r in 1 2; do for v in new prev; do if [ $v = prev ]; then export FREDDY_GPU_SO=$PWD/postgres-word2vec_amd/libfreddy_gpu_prev.so; else unset FREDDY_GPU_SO; fi
timeout 600 python bench.py --steps 20 --warmup 4 --N 40000000 --C 13000 --cpu-sample 0 --no-recall --no-host-abi > /tmp/n.out 2>/dev/null
python - "$v" <<'P'
import json, sys
o = json.load(open("bench_details.json"))
k = {n: v["avg_us"] for n, v in o["kernels"].items()}
ko = {n: v["avg_us"] for n, v in o["kernels_overlapped"].items()}
print(f"N40M {sys.argv[1]:5s} {o['value']/1e6:6.3f} M q/s  {o['ms_per_step']:.4f} ms  serial {o['pipelining']['serial_ms_per_step']:.4f}  coarse {k.get('coarse_table')} / {ko.get('coarse_table')}  plan {k.get('probe_plan')}  viol {o.get('filter_bound_violations')} gather {o.get('gather_verified')}")
P
done; done
